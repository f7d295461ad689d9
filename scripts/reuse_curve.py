"""Developer tool: how much of the heavy rows' product traffic an ideal cache of C bytes per column
window would serve (R-MAT scale-S A*A): segments B[k, w] ranked by reference count."""
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
P = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row, blen[col])
heavy = P > 4096
refs = torch.bincount(col[heavy[row]], minlength=n)             # heavy rows referencing B row k
nwin = n >> WSH
seg_key, seg_len = torch.unique(row * nwin + (col >> WSH), return_counts=True)
seg_k, seg_w = seg_key // nwin, seg_key % nwin
seg_refs = refs[seg_k]
seg_prod = seg_refs * seg_len
tot = int(seg_prod.sum())
print("segments", seg_key.numel(), "products", tot)
# by segment length
for lo, hi in [(1, 1), (2, 4), (5, 16), (17, 64), (65, 256), (257, 1 << 30)]:
    m = (seg_len >= lo) & (seg_len <= hi)
    print("len %d..%d: segment refs %.3g products %.1f%%" % (lo, hi, float(seg_refs[m].sum()), 100.0 * float(seg_prod[m].sum()) / tot))
# ideal per-window cache
order = torch.argsort(seg_w * (1 << 32) + ((1 << 31) - seg_refs.clamp(max=(1 << 31) - 1)))   # window, then refs descending
w_s, len_s, prod_s = seg_w[order], seg_len[order], seg_prod[order]
bytes_cum = torch.cumsum(len_s * 12, 0)
first = torch.zeros(nwin + 1, dtype=torch.int64, device=dev)
first[1:] = torch.cumsum(torch.bincount(w_s, minlength=nwin), 0)
base = torch.zeros_like(bytes_cum)
wb = torch.where(first[:-1] > 0, bytes_cum[(first[:-1] - 1).clamp(min=0)], torch.zeros_like(first[:-1]))
inwin = bytes_cum - wb[w_s]
for C in [0.5, 1, 2, 3, 4, 8, 16, 32]:
    m = inwin <= C * (1 << 20)
    print("ideal cache %4.1f MB per window: %.1f%% of products" % (C, 100.0 * float(prod_s[m].sum()) / tot))
wprod = torch.zeros(nwin, dtype=torch.int64, device=dev).index_add_(0, seg_w, seg_prod)
wbytes = torch.zeros(nwin, dtype=torch.int64, device=dev).index_add_(0, seg_w, seg_len * 12)
top = torch.argsort(wprod, descending=True)[:12]
for w in top.tolist():
    print("window %3d: %.1f%% of products, slice %.1f MB" % (w, 100.0 * int(wprod[w]) / tot, int(wbytes[w]) / 1e6))
