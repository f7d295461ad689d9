"""Developer tool: how many window-index lookups / non-empty segments / products the heavy rows of
R-MAT scale-S A*A have (decides whether the per-(cell, A tuple) index lookups matter)."""
import sys

import torch

from spsparse_amd import capi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WSH = 13
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
nnz = key.numel()
blen = torch.bincount(row, minlength=n)
nwin = n >> WSH
kw = torch.unique(row * nwin + (col >> WSH))
nw_k = torch.bincount(kw // nwin, minlength=n)                  # distinct windows per B row
print("nnz", nnz, "nonempty (k,w) pairs", kw.numel(), "avg seg len over B", nnz / kw.numel())
P = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row, blen[col])
L = blen
heavy = P > 4096
print("heavy rows", int(heavy.sum()), "A tuples in heavy rows", int(L[heavy].sum()), "products heavy", int(P[heavy].sum()))
hv_t = heavy[row]
segs = int(nw_k[col[hv_t]].sum())
print("non-empty segments processed by heavy rows", segs, "products/segment", int(P[heavy].sum()) / segs)
# per heavy row window histogram
hid = torch.cumsum(heavy.long(), 0) - 1
nh = int(heavy.sum())
cnt_kw = torch.bincount(row * nwin + (col >> WSH), minlength=n * nwin).view(n, nwin).to(torch.float32)
wp = torch.zeros(nh, nwin, dtype=torch.float32, device=dev)
a_r, a_k = hid[row[hv_t]], col[hv_t]
CH = 1 << 20
for s in range(0, a_r.numel(), CH):
    wp.index_add_(0, a_r[s:s + CH], cnt_kw[a_k[s:s + CH]])
ncell = torch.zeros(nh, dtype=torch.int64, device=dev)
ndense = torch.zeros(nh, dtype=torch.int64, device=dev)
acc = torch.zeros(nh, dtype=torch.float32, device=dev)
for w in range(nwin):
    p = wp[:, w]
    dense = p > 2048
    close = (dense & (acc > 0)) | (~dense & (acc + p > 2048) & (acc > 0))
    ncell += close.long() + dense.long()
    ndense += dense.long()
    acc = torch.where(dense, torch.zeros_like(acc), torch.where(close, p, acc + p))
ncell += (acc > 0).long()
Lh = L[heavy]
print("cells", int(ncell.sum()), "dense", int(ndense.sum()), "lookups sum(ncell*L)", int((ncell * Lh).sum()),
      "dense-cell lookups", int((ndense * Lh).sum()))
print("products per lookup", int(P[heavy].sum()) / int((ncell * Lh).sum()))

# how many of the (cell, A tuple) lookups find a non-empty B segment
ne_kw = (cnt_kw > 0).to(torch.float32)
ne = torch.zeros(nh, nwin, dtype=torch.float32, device=dev)
for s in range(0, a_r.numel(), CH):
    ne.index_add_(0, a_r[s:s + CH], ne_kw[a_k[s:s + CH]])
dense_mask = wp > 2048
print("dense cells: lookups %.4g, of them non-empty %.4g (%.1f%%)" % (
    float((dense_mask.float() * Lh[:, None].float()).sum()), float((ne * dense_mask.float()).sum()),
    100.0 * float((ne * dense_mask.float()).sum()) / float((dense_mask.float() * Lh[:, None].float()).sum())))
hash_mask = (wp > 0) & ~dense_mask
print("hash windows: (row, window) pairs with products %.4g, non-empty segments in them %.4g, lookups if one per window %.4g" % (
    float(hash_mask.sum()), float((ne * hash_mask.float()).sum()), float((hash_mask.float() * Lh[:, None].float()).sum())))
