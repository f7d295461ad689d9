"""Test tool (GPU box; not collected by pytest): R-MAT A*A at several seeds and scales against the oracle's streaming digest -- whole-product
count / index hash / per-row counts and hashes in the digest sink and in both passes of the COO sink.

    python tests/soak_rmat.py [scale] [seeds, e.g. 2,3,4]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import binding as orc          # noqa: E402  (test infrastructure: this script is a checker, like tests/)
from spsparse_amd import capi              # noqa: E402
from spsparse_amd import workloads as wl  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 4]
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    n, ne = 1 << scale, 16 << scale
    for seed in seeds:
        raw = (torch.empty(ne, dtype=torch.int32, device=dev), torch.empty(ne, dtype=torch.int32, device=dev), torch.empty(ne, dtype=torch.float64, device=dev))
        ctx.gen_rmat(scale, seed, 0, ne, *[x.data_ptr() for x in raw])
        A = capi.device_coo(raw[0].data_ptr(), raw[1].data_ptr(), raw[2].data_ptr(), ne, (n, n))
        a = wl.rmat(scale, seed)
        w = orc.multiply_digest(orc.Mat(*a), orc.Mat(*a), nthreads=orc.host_threads(), rowstats=True)
        for xcd in (2, 0):
            ctx.set_tuning("xcd", xcd)
            d = ctx.multiply(A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
            assert (d.nnz, d.hash, d.products) == (w.nnz, w.hash, w.products), (seed, xcd)
            assert np.array_equal(ctx.to_host(d.row_nnz, n, np.int64), w.row_nnz)
            assert np.array_equal(ctx.to_host(d.row_hash, n, np.uint64), w.row_hash)
            assert abs(d.sum - w.sum) <= 1e-10 * abs(w.sum)
            r = ctx.multiply(A, A, sink=capi.SINK_COO)
            assert r.nnz == w.nnz
            ci = ctx.to_host(r.idx0, int(r.nnz), np.int32)
            cj = ctx.to_host(r.idx1, int(r.nnz), np.int32)
            assert np.array_equal(np.bincount(ci, minlength=n), w.row_nnz)
            with np.errstate(over="ignore"):
                assert int(np.sum(orc.mix64(ci, cj), dtype=np.uint64)) == w.hash
            order = ci.astype(np.int64) * n + cj
            assert np.all(order[1:] > order[:-1]), "COO tuples not in ascending (i, j)"
        ctx.set_tuning("xcd", 2)
        print("scale %d seed %d: nnz(C) %d products %d dense cells %d ok" % (scale, seed, w.nnz, w.products, d.cells_dense), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
