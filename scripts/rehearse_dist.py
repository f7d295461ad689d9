"""Developer tool: the N-way sharded step of bench.py rehearsed on ONE GPU, one rank at a time.

    python scripts/rehearse_dist.py [scale] [worlds, e.g. 2,4,8] [--trace RANK]

Every rank's spsamd_dist_multiply is run ALONE on the GPU, from its block of the raw tuples, through a REPLAY transport:
what the peers would send in each exchange round (status headers, need masks, row lengths, packed panel tuples) is computed
beforehand with torch from the consolidated matrix and copied into the step's receive buffers device-to-device when the
step asks for it.  So the figure per rank is everything a rank does in a step -- block consolidation, round-1 kernels, the
host synchronisation for the totals, pack kernel, panel row pointer, block product -- with a wire time of about zero (and,
unlike the RCCL path, a stream synchronisation per round and no overlap of the panel transfer).  Add the wire estimate it
prints (bytes received / 7 xGMI links x 50 GB/s effective) for a pessimistic total; with the overlap of the RCCL path the
transfer sits beside the symbolic phase.  Block boundaries: cost estimate, then measure / rebalance rounds as in bench.py.
"""
import ctypes
import sys
import time

import numpy as np
import torch

from spsparse_amd import capi
from spsparse_amd import dist as sd


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    scale = int(argv[0]) if argv else 20
    worlds = [int(x) for x in argv[1].split(",")] if len(argv) > 1 else [2, 4, 8]
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    hip = ctypes.CDLL("libamdhip64.so.7" if True else None)
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]

    def d2d(dst, src, nbytes):
        if nbytes:
            rc = hip.hipMemcpyAsync(dst, src, nbytes, 3, stream.cuda_stream)
            assert rc == 0, rc

    n, ne = 1 << scale, 16 << scale
    raw = (torch.empty(ne, dtype=torch.int32, device=dev), torch.empty(ne, dtype=torch.int32, device=dev), torch.empty(ne, dtype=torch.float64, device=dev))
    ctx.gen_rmat(scale, 1, 0, ne, *[x.data_ptr() for x in raw])
    ctx.reserve(int(ne * 220) + (512 << 20))
    A = capi.device_coo(raw[0].data_ptr(), raw[1].data_ptr(), raw[2].data_ptr(), ne, (n, n))
    r = ctx.consolidate(A, 0)
    m = int(r.nnz)
    c0 = torch.empty(m, dtype=torch.int32, device=dev); c1 = torch.empty_like(c0); cv = torch.empty(m, dtype=torch.float64, device=dev)
    ctx.memcpy(c0.data_ptr(), r.idx0, m * 4); ctx.memcpy(c1.data_ptr(), r.idx1, m * 4); ctx.memcpy(cv.data_ptr(), r.val, m * 8)
    rowlen = torch.bincount(c0.long(), minlength=n).to(torch.int32)
    full = min((ctx.multiply(A, A, sink=capi.SINK_DIGEST) for _ in range(3)), key=lambda x: x.ms_total)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ctx.multiply(A, A, sink=capi.SINK_DIGEST)
    torch.cuda.synchronize()
    full_wall = (time.perf_counter() - t) / 3 * 1e3
    print("whole product from raw tuples on one GPU: %.2f ms wall per step (device %.2f)" % (full_wall, full.ms_total), flush=True)
    P = sd.row_products(c0, c1, rowlen.long(), n)
    cost = sd.row_cost(P)
    cost_prefix = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(cost, 0)])
    crow = c0.long()

    def rank_step(world, bounds, rank, reps=3):
        """(best wall ms, device ms of the block product, bytes received from peers) of `rank`'s step."""
        lo, hi = bounds[rank], bounds[rank + 1]
        keep = (raw[0] >= lo) & (raw[0] < hi)
        blk = tuple(x[keep].contiguous() for x in raw)
        # what the peers send: need masks of every rank (for my rows), row lengths, my panel's tuples from every owner
        mine = (c0 >= lo) & (c0 < hi)
        need_me = torch.zeros(n, dtype=torch.uint8, device=dev)
        need_me[c1[mine].long()] = 1
        their = []
        for p in range(world):
            np_ = torch.zeros(n, dtype=torch.uint8, device=dev)
            sel = (c0 >= bounds[p]) & (c0 < bounds[p + 1])
            np_[c1[sel].long()] = 1
            their.append(np_[lo:hi].contiguous())
        panel_sel = need_me[crow] == 1
        pcols, pvals = [], []
        for p in range(world):
            s = panel_sel & (c0 >= bounds[p]) & (c0 < bounds[p + 1])
            pcols.append(c1[s].contiguous()); pvals.append(cv[s].contiguous())
        hdr0 = torch.zeros(2, dtype=torch.int32, device=dev)
        state = {"call": 0, "bytes": 0}

        def transport(send, sendb, recv, recvb, _stream):
            k = state["call"] % 5
            state["call"] += 1
            for p in range(world):
                if p == rank:
                    d2d(recv[p], send[p], sendb[p])
                    continue
                if k == 0:
                    src, nb = hdr0.data_ptr(), 8
                elif k == 1:
                    src, nb = their[p].data_ptr(), their[p].numel()
                elif k == 2:
                    src, nb = rowlen.data_ptr() + 4 * bounds[p], 4 * (bounds[p + 1] - bounds[p])
                elif k == 3:
                    src, nb = pcols[p].data_ptr(), 4 * pcols[p].numel()
                else:
                    src, nb = pvals[p].data_ptr(), 8 * pvals[p].numel()
                assert nb == recvb[p], (k, p, nb, recvb[p])
                state["bytes"] += nb if k >= 3 else 0
                d2d(recv[p], src, nb)

        dd = capi.Dist(ctx, rank, world, transport=transport)
        Ab = capi.device_coo(blk[0].data_ptr(), blk[1].data_ptr(), blk[2].data_ptr(), blk[0].numel(), (n, n))
        best, dev_ms, exch = 1e9, 0.0, 0.0
        res = None
        for _ in range(reps + 1):
            state["bytes"] = 0
            torch.cuda.synchronize()
            t = time.perf_counter()
            res, st = dd.multiply(Ab, None, bounds, sink=capi.SINK_DIGEST)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t) * 1e3
            if w < best:
                best, dev_ms, exch = w, float(res.ms_total), float(st.ms_exchange)
        dd.close()
        return best, dev_ms, exch, state["bytes"], res

    trace_rank = None
    for a in sys.argv[1:]:
        if a.startswith("--trace"):
            trace_rank = int(a.split("=")[1]) if "=" in a else 4
    for world in worlds:
        bounds = sd.product_balanced_bounds(cost, world)
        for rnd in range(4):
            out = [rank_step(world, bounds, q) for q in range(world)]
            walls = [o[0] for o in out]
            wire = [o[3] / (7 * 50e9) * 1e3 for o in out]
            nnz = sum(int(o[4].nnz) for o in out)
            hsh = sum(int(o[4].hash) for o in out) % (1 << 64)
            assert (nnz, hsh) == (int(full.nnz), int(full.hash)), "the blocks' digests do not add up to the whole product's"
            print("world %d round %d: step wall ms %s | max %.2f -> %.2fx of %.2f | block product device ms %s | before the product ms %s | wire est ms %s" % (
                world, rnd, " ".join("%.2f" % w for w in walls), max(walls), full_wall / max(walls), full_wall,
                " ".join("%.2f" % o[1] for o in out), " ".join("%.2f" % o[2] for o in out), " ".join("%.2f" % w for w in wire)), flush=True)
            bounds = sd.rebalance_bounds(bounds, cost_prefix, walls, min_gain=0.015)
        if trace_rank is not None and world == worlds[-1]:
            rank_step(world, bounds, min(trace_rank, world - 1), reps=5)
    ctx.close()


if __name__ == "__main__":
    main()
