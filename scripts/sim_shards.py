"""Developer tool: run every row block of an N-way sharded multiply one after the other on ONE
GPU and report the per-block device time (what each rank would spend in spsamd_multiply)."""
import sys

import torch

from spsparse_amd import capi
from spsparse_amd import dist as sd


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "cost"
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    n, ne = 1 << scale, 16 << scale
    t0 = torch.empty(ne, dtype=torch.int32, device=dev)
    t1 = torch.empty(ne, dtype=torch.int32, device=dev)
    tv = torch.empty(ne, dtype=torch.float64, device=dev)
    ctx.gen_rmat(scale, 1, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    r = ctx.consolidate(capi.device_coo(t0.data_ptr(), t1.data_ptr(), tv.data_ptr(), ne, (n, n)), 0)
    m = int(r.nnz)
    c0 = torch.empty(m, dtype=torch.int32, device=dev)
    c1 = torch.empty(m, dtype=torch.int32, device=dev)
    cv = torch.empty(m, dtype=torch.float64, device=dev)
    ctx.memcpy(c0.data_ptr(), r.idx0, m * 4)
    ctx.memcpy(c1.data_ptr(), r.idx1, m * 4)
    ctx.memcpy(cv.data_ptr(), r.val, m * 8)
    rowlen = torch.bincount(c0.long(), minlength=n)
    P = sd.row_products(c0, c1, rowlen, n)
    B = capi.device_coo(c0.data_ptr(), c1.data_ptr(), cv.data_ptr(), m, (n, n), sort0=0)
    full = None
    for _ in range(2):
        full = ctx.multiply(B, B, sink=capi.SINK_DIGEST)
    print("full: %.1f ms" % full.ms_total)
    cost = sd.row_cost(P) if mode == "cost" else P
    cost_prefix = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(cost, 0)])
    for world in (2, 4, 8):
        bounds = sd.product_balanced_bounds(cost, world)
        for rnd in range(4):                      # round 0: the estimate alone; then measure -> rebalance
            times, prods = [], []
            for rank in range(world):
                keep = (c0 >= bounds[rank]) & (c0 < bounds[rank + 1])
                a0, a1, av = c0[keep].contiguous(), c1[keep].contiguous(), cv[keep].contiguous()
                A = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), a0.numel(), (n, n), sort0=0)
                res = ctx.multiply(A, B, sink=capi.SINK_DIGEST)
                best = min(ctx.multiply(A, B, sink=capi.SINK_DIGEST).ms_total for _ in range(3))
                times.append(best)
                prods.append(res.products)
            print("world %d (%s) round %d: block ms %s | max %.1f -> speed-up %.2fx (compute only) | products %s" % (
                world, mode, rnd, " ".join("%.1f" % t for t in times), max(times), full.ms_total / max(times),
                " ".join("%.2g" % p for p in prods)), flush=True)
            bounds = sd.rebalance_bounds(bounds, cost_prefix, times, min_gain=0.015)


def fixed_cost():
    """Device time of a multiply whose A block is a single light row: the per-call fixed cost."""
    scale = 20
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    n, ne = 1 << scale, 16 << scale
    t0 = torch.empty(ne, dtype=torch.int32, device=dev)
    t1 = torch.empty(ne, dtype=torch.int32, device=dev)
    tv = torch.empty(ne, dtype=torch.float64, device=dev)
    ctx.gen_rmat(scale, 1, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    r = ctx.consolidate(capi.device_coo(t0.data_ptr(), t1.data_ptr(), tv.data_ptr(), ne, (n, n)), 0)
    m = int(r.nnz)
    c0 = torch.empty(m, dtype=torch.int32, device=dev)
    c1 = torch.empty(m, dtype=torch.int32, device=dev)
    cv = torch.empty(m, dtype=torch.float64, device=dev)
    ctx.memcpy(c0.data_ptr(), r.idx0, m * 4)
    ctx.memcpy(c1.data_ptr(), r.idx1, m * 4)
    ctx.memcpy(cv.data_ptr(), r.val, m * 8)
    B = capi.device_coo(c0.data_ptr(), c1.data_ptr(), cv.data_ptr(), m, (n, n), sort0=0)
    for rows in (1, 1000, 100000):
        keep = c0 >= (n - rows)
        a0, a1, av = c0[keep].contiguous(), c1[keep].contiguous(), cv[keep].contiguous()
        A = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), a0.numel(), (n, n), sort0=0)
        import time
        for _ in range(3):
            t = time.time()
            res = ctx.multiply(A, B, sink=capi.SINK_DIGEST)
            wall = (time.time() - t) * 1e3
        print("last %d rows: tuples %d products %d: device %.2f ms (cons %.2f symb %.2f num %.2f) wall %.2f ms" % (
            rows, a0.numel(), res.products, res.ms_total, res.ms_consolidate, res.ms_symbolic, res.ms_numeric, wall))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fixed":
        fixed_cost()
    else:
        main()
