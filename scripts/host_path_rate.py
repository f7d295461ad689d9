"""Developer tool: the PCIe-inclusive rate of the host-pointer boundary (what a caller of the C++ shim
pays): host COO operands -> spsamd_multiply (H2D inside) -> spsamd_result_fetch (D2H in chunks) -> host arrays."""
import sys
import time

import numpy as np

from spsparse_amd import capi, workloads as wl

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 17
a = wl.rmat(scale, seed=1)
ctx = capi.Context(0)
s, keep = capi.host_coo(*a)
for rep in range(3):
    t0 = time.perf_counter()
    res = ctx.multiply(s, s, sink=capi.SINK_COO)
    t1 = time.perf_counter()
    i, j, v = ctx.fetch(res)
    t2 = time.perf_counter()
    print("scale %d rep %d: multiply (H2D + device) %.1f ms [device %.1f] | fetch (D2H + host copy) %.1f ms | nnz(C) %d -> %.3g nnz(C)/s PCIe-inclusive, "
          "%.1f GB/s of output over the fetch" % (scale, rep, (t1 - t0) * 1e3, res.ms_total, (t2 - t1) * 1e3, res.nnz,
                                               res.nnz / (t2 - t0), 16 * res.nnz / (t2 - t1) / 1e9), flush=True)
ctx.close()
