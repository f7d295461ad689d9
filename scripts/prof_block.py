"""Developer tool: one 1/8 row block of the R-MAT scale-20 product (what a rank of an 8-way sharded run multiplies), five
times, for a kernel trace:  rocprofv3 --kernel-trace --stats -d out -o blk -- python3 scripts/prof_block.py"""
import sys

import torch

from spsparse_amd import capi


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    n, ne = 1 << scale, 16 << scale
    t0 = torch.empty(ne, dtype=torch.int32, device=dev); t1 = torch.empty_like(t0); tv = torch.empty(ne, dtype=torch.float64, device=dev)
    ctx.gen_rmat(scale, 1, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    r = ctx.consolidate(capi.device_coo(t0.data_ptr(), t1.data_ptr(), tv.data_ptr(), ne, (n, n)), 0)
    m = int(r.nnz)
    c0 = torch.empty(m, dtype=torch.int32, device=dev); c1 = torch.empty_like(c0); cv = torch.empty(m, dtype=torch.float64, device=dev)
    ctx.memcpy(c0.data_ptr(), r.idx0, m * 4); ctx.memcpy(c1.data_ptr(), r.idx1, m * 4); ctx.memcpy(cv.data_ptr(), r.val, m * 8)
    B = capi.device_coo(c0.data_ptr(), c1.data_ptr(), cv.data_ptr(), m, (n, n), sort0=0)
    # a block with about 1/8 of the products: rows from the middle of the matrix
    rowlen = torch.bincount(c0.long(), minlength=n)
    P = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, c0.long(), rowlen[c1.long()])
    pref = torch.cumsum(P, 0)
    lo = int(torch.searchsorted(pref, pref[-1] * 4 // 8)); hi = int(torch.searchsorted(pref, pref[-1] * 5 // 8))
    sel = (c0 >= lo) & (c0 < hi)
    a0, a1, av = c0[sel].contiguous(), c1[sel].contiguous(), cv[sel].contiguous()
    A = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), a0.numel(), (n, n), sort0=0)
    for _ in range(6):
        res = ctx.multiply(A, B, sink=capi.SINK_DIGEST)
    print("block rows [%d, %d): products %.3g of %.3g: %.2f ms (cons %.2f symb %.2f num %.2f)" % (
        lo, hi, res.products, float(pref[-1]), res.ms_total, res.ms_consolidate, res.ms_symbolic, res.ms_numeric))


if __name__ == "__main__":
    main()
