"""Developer probe: time the device pipeline on a synthetic operand (not the bench)."""
import os
import sys
import time

import torch

from spsparse_amd import capi


def galerkin(ctx, N, sink, reps, dev):
    """cfg5: T = R*A, C = T*R^T on the 3-D 7-point Laplacian."""
    nc = N // 2
    na, nr = 7 * N ** 3 - 6 * N ** 2, N ** 3

    def bufs(n):
        return (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                torch.empty(n, dtype=torch.float64, device=dev))
    a, r = bufs(na), bufs(nr)
    ctx.gen_laplace3d(N, *[x.data_ptr() for x in a])
    ctx.gen_aggregation3d(N, *[x.data_ptr() for x in r])
    torch.cuda.synchronize()
    A = capi.device_coo(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), na, (N ** 3, N ** 3), sort0=0)
    R = capi.device_coo(r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), nr, (nc ** 3, N ** 3), sort0=0)
    for rep in range(reps):
        t0 = time.time()
        rt = ctx.multiply(R, A, sink=capi.SINK_COO)
        nt = int(rt.nnz)
        t = bufs(nt)
        ctx.memcpy(t[0].data_ptr(), rt.idx0, nt * 4)
        ctx.memcpy(t[1].data_ptr(), rt.idx1, nt * 4)
        ctx.memcpy(t[2].data_ptr(), rt.val, nt * 8)
        T = capi.device_coo(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), nt, (nc ** 3, N ** 3), sort0=0)
        rc = ctx.multiply(T, R, tB='T', sink=sink)
        dt = time.time() - t0
        print("galerkin %d rep %d: wall %.1f ms | R*A total %.2f ms (P %d nnzT %d) | T*Rt total %.2f ms (P %d nnzC %d; closed form %d) | "
              "alg-read %.0f GB/s" % (N, rep, dt * 1e3, rt.ms_total, rt.products, rt.nnz, rc.ms_total, rc.products, rc.nnz,
                                      7 * nc ** 3 - 6 * nc ** 2,
                                      (16 * (rt.nnz_a + rc.nnz_a) + 12 * (rt.products + rc.products)) / ((rt.ms_total + rc.ms_total) * 1e-3) / 1e9),
              flush=True)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "rmat"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    sink = capi.SINK_COO if (len(sys.argv) > 3 and sys.argv[3] == "coo") else capi.SINK_DIGEST
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dev = torch.device("cuda:0")
    ctx = capi.Context()
    if kind == "rmat":
        n, ne = 1 << size, 16 << size
    elif kind == "poisson":
        n, ne = size * size, 5 * size * size - 4 * size
    elif kind == "galerkin":
        return galerkin(ctx, size, sink, reps, dev)
    t0 = torch.empty(ne, dtype=torch.int32, device=dev)
    t1 = torch.empty(ne, dtype=torch.int32, device=dev)
    tv = torch.empty(ne, dtype=torch.float64, device=dev)
    if kind == "rmat":
        ctx.gen_rmat(size, 1, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    else:
        ctx.gen_poisson2d(size, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    torch.cuda.synchronize()
    A = capi.device_coo(t0.data_ptr(), t1.data_ptr(), tv.data_ptr(), ne, (n, n))
    for rep in range(reps):
        t = time.time()
        r = ctx.multiply(A, A, sink=sink, flags=int(os.environ.get("PROBE_FLAGS", "0")))
        dt = time.time() - t
        print("%s %d rep %d: total %.1f ms (cons %.1f symb %.1f light %.2f mid %.2f hash %.1f dense %.1f) "
              "P %.3g nnzC %.3g | prods mid %.3g hash %.3g dense %.3g | cells hash %d dense %d | ws %.2f GB | %.3g prod/s, %.0f GB/s alg-read" % (
                  kind, size, rep, r.ms_total, r.ms_consolidate, r.ms_symbolic, r.ms_light, r.ms_mid, r.ms_heavy - r.ms_dense,
                  r.ms_dense, r.products, r.nnz, r.products_mid, r.products_heavy - r.products_dense, r.products_dense,
                  r.cells_hash, r.cells_dense, r.workspace_bytes / 1e9, r.products / (r.ms_total * 1e-3),
                  (16 * r.nnz_a + 12 * r.products) / (r.ms_total * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    main()
