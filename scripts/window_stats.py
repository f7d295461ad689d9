"""Developer tool: how the scalar products of the heavy rows of R-MAT scale-S A*A spread over (row, window)
pairs, by pair size -- decides where the dense / hash threshold belongs and what a cheaper per-cell fixed
cost would buy."""
import sys

import torch

from spsparse_amd import capi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WSH = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dev = torch.device("cuda:0")
ctx = capi.Context()
n, ne = 1 << S, 16 << S
t0 = torch.empty(ne, dtype=torch.int32, device=dev)
t1 = torch.empty(ne, dtype=torch.int32, device=dev)
tv = torch.empty(ne, dtype=torch.float64, device=dev)
ctx.gen_rmat(S, 1, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
torch.cuda.synchronize()
key = torch.unique(t0.long() * n + t1.long())
row, col = key // n, key % n
blen = torch.bincount(row, minlength=n)
nwin = n >> WSH
P = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row, blen[col])
heavy = P > 4096
hid = torch.cumsum(heavy.long(), 0) - 1
nh = int(heavy.sum())
hv_t = heavy[row]
cnt_kw = torch.bincount(row * nwin + (col >> WSH), minlength=n * nwin).view(n, nwin).to(torch.float32)
wp = torch.zeros(nh, nwin, dtype=torch.float32, device=dev)      # products per (heavy row, window)
ns = torch.zeros(nh, nwin, dtype=torch.float32, device=dev)      # non-empty segments per (heavy row, window)
a_r, a_k = hid[row[hv_t]], col[hv_t]
CH = 1 << 20
ne_kw = (cnt_kw > 0).to(torch.float32)
for s in range(0, a_r.numel(), CH):
    wp.index_add_(0, a_r[s:s + CH], cnt_kw[a_k[s:s + CH]])
    ns.index_add_(0, a_r[s:s + CH], ne_kw[a_k[s:s + CH]])
Lh = blen[heavy].float()
print("heavy rows %d, products %.4g, A tuples %.4g" % (nh, float(wp.sum()), float(Lh.sum())))
edges = [0, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 1e9]
print("pair size     pairs     products   share   segs     prod/seg   sum(La) over pairs")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (wp > lo) & (wp <= hi)
    print("%6g-%-6g %9d  %10.4g  %5.1f%%  %9.4g  %6.2f  %10.4g" % (lo, hi, int(m.sum()), float(wp[m].sum()), 100 * float(wp[m].sum()) / float(wp.sum()),
          float(ns[m].sum()), float(wp[m].sum()) / max(1.0, float(ns[m].sum())), float((m.float() * Lh[:, None]).sum())))
print("La of heavy rows: quantiles", [float(torch.quantile(Lh, q)) for q in (0.1, 0.5, 0.9, 0.99, 0.999)], "max", float(Lh.max()))
# segment length distribution over all (k, w) pairs weighted by use (column degree of k among heavy rows)
use = torch.bincount(a_k, minlength=n).float()                   # times B row k is used by heavy rows
seg_len = cnt_kw
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 64), (64, 1e9)):
    m = (seg_len > lo) & (seg_len <= hi)
    w = (m.float() * use[:, None])
    print("segments of length (%g, %g]: uses %.4g, products %.4g" % (lo, hi, float(w.sum()), float((w * seg_len).sum())))
