"""Developer tool: where the time of one N>1 bench step goes, rehearsed on a 1-rank NCCL group."""
import os
import time

import torch
import torch.distributed as dist

from spsparse_amd import capi
from spsparse_amd import dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
scale = 20
n, ne = 1 << scale, 16 << scale
raw0 = torch.empty(ne, dtype=torch.int32, device=dev)
raw1 = torch.empty(ne, dtype=torch.int32, device=dev)
rawv = torch.empty(ne, dtype=torch.float64, device=dev)
ctx.gen_rmat(scale, 1, 0, ne, raw0.data_ptr(), raw1.data_ptr(), rawv.data_ptr())
ctx.reserve(int(ne * 220) + (512 << 20))
bounds = [0, n]


def tick(label, t, acc):
    torch.cuda.synchronize()
    now = time.perf_counter()
    acc[label] = acc.get(label, 0.0) + (now - t) * 1e3
    return now


for rep in range(4):
    acc = {}
    t = time.perf_counter()
    r = ctx.consolidate(capi.device_coo(raw0.data_ptr(), raw1.data_ptr(), rawv.data_ptr(), ne, (n, n)), 0)
    t = tick("consolidate call", t, acc)
    m = int(r.nnz)
    a0 = torch.empty(m, dtype=torch.int32, device=dev)
    a1 = torch.empty(m, dtype=torch.int32, device=dev)
    av = torch.empty(m, dtype=torch.float64, device=dev)
    ctx.memcpy(a0.data_ptr(), r.idx0, m * 4)
    ctx.memcpy(a1.data_ptr(), r.idx1, m * 4)
    ctx.memcpy(av.data_ptr(), r.val, m * 8)
    t = tick("copy out", t, acc)
    p0, p1, pv, remote = sd.exchange_b_panels(a1, a0, a1, av, bounds, n)
    t = tick("exchange", t, acc)
    Ab = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), m, (n, n), sort0=0)
    Bp = capi.device_coo(p0.data_ptr(), p1.data_ptr(), pv.data_ptr(), p0.numel(), (n, n), sort0=0)
    res = ctx.multiply(Ab, Bp, sink=capi.SINK_DIGEST)
    t = tick("multiply call", t, acc)
    print("rep %d: %s | multiply device ms %.1f (cons %.2f symb %.2f num %.2f)" % (
        rep, " ".join("%s %.2f" % kv for kv in acc.items()), res.ms_total, res.ms_consolidate, res.ms_symbolic, res.ms_numeric), flush=True)
ctx.close()
dist.destroy_process_group()
