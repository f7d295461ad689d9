"""Test tool (not collected by pytest; was scripts/compare_big.py): one-off full tuple-by-tuple comparison against the row-wise oracle at a larger R-MAT scale
than the test suite uses (needs the host cores and memory of the GPU box)."""
import sys
import time

import numpy as np

from oracle import binding as orc
from spsparse_amd import capi, workloads as wl

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 17
a = wl.rmat(scale, seed=5)
A = orc.Mat(*a)
t = time.time()
wi, wj, wv, _ = orc.multiply(A, A, rowwise=True, nthreads=16)
print("oracle: %d tuples in %.1f s" % (len(wv), time.time() - t), flush=True)
ctx = capi.Context(0)
s, keep = capi.host_coo(*a)
for flags, name in ((0, "default"), (capi.SINK_ORDERED, "ordered")):
    res = ctx.multiply(s, s, sink=capi.SINK_COO, flags=flags)
    gi, gj, gv = ctx.fetch(res)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj), name
    err = float(np.max(np.abs(gv - wv) / np.abs(wv)))
    print("%s: index set identical (%d tuples), max relative value error %.3g, bit-identical values: %s; rows l/m/h %d/%d/%d cells hash/dense %d/%d" % (
        name, len(gv), err, bool(np.array_equal(gv, wv)), res.rows_light, res.rows_mid, res.rows_heavy, res.cells_hash, res.cells_dense), flush=True)
    assert err <= 1e-12
    if flags:
        assert np.array_equal(gv, wv)
ctx.close()
