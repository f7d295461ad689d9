"""Developer tool: one 1/N row block of R-MAT scale-20 against the whole B (what one rank of N multiplies),
for profiling the per-rank fixed costs."""
import sys
import time

import torch

from spsparse_amd import capi
from spsparse_amd import dist as sd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
rowlen = torch.bincount(c0.long(), minlength=n)
P = sd.row_products(c0, c1, rowlen, n)
bounds = sd.product_balanced_bounds(sd.row_cost(P), N)
keep = (c0 >= bounds[which]) & (c0 < bounds[which + 1])
a0, a1, av = c0[keep].contiguous(), c1[keep].contiguous(), cv[keep].contiguous()
A = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), a0.numel(), (n, n), sort0=0)
B = capi.device_coo(c0.data_ptr(), c1.data_ptr(), cv.data_ptr(), m, (n, n), sort0=0)
ctx.reserve(int(ne * 220) + (512 << 20))
for rep in range(reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = ctx.multiply(A, B, sink=capi.SINK_DIGEST)
    wall = (time.perf_counter() - t) * 1e3
    print("block %d/%d rep %d: wall %.2f ms device %.2f (cons %.2f symb %.2f num %.2f: light %.2f mid %.2f hash %.2f dense %.2f) products %.3g" % (
        which, N, rep, wall, res.ms_total, res.ms_consolidate, res.ms_symbolic, res.ms_numeric, res.ms_light, res.ms_mid,
        res.ms_heavy - res.ms_dense, res.ms_dense, res.products), flush=True)
