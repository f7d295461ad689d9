"""Multi-GPU SpGEMM, the harness side: row-block boundaries (cost model, measured rebalancing), the reduction of the
ranks' digests and a test transport.  The STEP itself -- consolidate the own block, exchange the needed B row panels,
multiply -- is spsamd_dist_multiply behind the C ABI (csrc/dist.hip, grouped ncclSend / ncclRecv over RCCL).  No
reduction of C anywhere: it stays row partitioned.

The reference is single threaded and has no counterpart of this module
(SURVEY.md section 8e).  Output row i of C depends only on row i of op(A) and
the B rows {k : A(i,k) != 0} (the reference's own loop structure,
multiply_sparse.hpp:192), so the only exchange step is fetching those B rows.

Everything here is tensor plumbing on whatever device the tensors live on;
the product itself is spsamd_multiply on each rank's GPU.
"""
import torch
import torch.distributed as dist


def product_balanced_bounds(row_products, nparts):
    """Contiguous row-block boundaries [b0=0, b1, ..., bN=n] such that every
    block holds ~1/nparts of the given per-row weights: the scalar products, or
    row_cost() of them (equal-size blocks put 40% of an un-permuted R-MAT's
    products on shard 0).  row_products: int64 [n]."""
    n = row_products.numel()
    pref = torch.cumsum(row_products.to(torch.int64), 0)
    total = int(pref[-1]) if n else 0
    targets = torch.tensor([total * p // nparts for p in range(1, nparts)], dtype=torch.int64, device=pref.device)
    cuts = torch.searchsorted(pref, targets, right=False) + 1 if nparts > 1 else targets
    cuts = torch.clamp(cuts, 0, n)
    b = [0] + [int(c) for c in cuts.tolist()] + [n]
    for q in range(1, len(b)):          # monotone
        b[q] = max(b[q], b[q - 1])
    return b


# Measured device time per scalar product by the size of its output row (MI355X, R-MAT scale-20
# sharded 8 ways, scripts/sim_shards.py): light rows (<= 64 products) ~16 ps, rows of one hash
# cell (<= 4096) ~10 ps, the bulk ~4.8 ps; rows beyond a million products cost more again
# (~7 ps): their largest dense cells are single work items that finish last on a small shard.
_COST_STEPS = ((64, 16.0), (4096, 10.0), (1 << 20, 4.8))
_COST_TOP = 7.0


def row_cost(row_products_):
    """Estimated device time (ps) of every output row from its product count: the weights that
    make the contiguous row blocks equal in TIME, not just in products."""
    P = row_products_.to(torch.float64)
    w = torch.full_like(P, _COST_TOP)
    for limit, cost in reversed(_COST_STEPS):
        w = torch.where(P <= limit, torch.full_like(P, cost), w)
    return (P * w).round().to(torch.int64)


def rebalance_bounds(bounds, cost_prefix, block_ms, fixed_ms=0.0, min_gain=0.0):
    """New contiguous row-block boundaries from MEASURED per-block times.

    bounds       current boundaries [0, b1, ..., n]
    cost_prefix  exclusive prefix sum of the per-row cost estimate, [n+1] (cost_prefix[n] = total)
    block_ms     measured local time of every block under `bounds` (same list on every rank)
    fixed_ms     part of every block's time that does not move with its rows

    Inside a block the measured time (less fixed_ms) is spread over its rows in proportion to the
    cost estimate, which gives a piecewise-linear cumulative time over the rows; the new
    boundaries cut it into equal parts.  One or two rounds (measure, rebalance) make the blocks
    equal in time even where the estimate is off by a block-dependent factor.  With min_gain > 0 the
    boundaries are left alone when the slowest block is less than that fraction above the mean (a
    rebalance on timer noise can only make things worse)."""
    nparts = len(bounds) - 1
    n = bounds[-1]
    mean = sum(float(t) for t in block_ms) / max(nparts, 1)
    if min_gain > 0.0 and mean > 0.0 and max(float(t) for t in block_ms) / mean - 1.0 < min_gain:
        return list(bounds)                      # already balanced within the measurement noise: leave it
    cp = cost_prefix.to(torch.float64)
    var = [max(float(t) - fixed_ms, 1e-9) for t in block_ms]
    total = sum(var)
    cum = [0.0]
    for v in var:
        cum.append(cum[-1] + v)
    new = [0]
    q = 0
    for j in range(1, nparts):
        target = total * j / nparts
        while q + 1 < nparts and cum[q + 1] <= target:
            q += 1
        lo, hi = bounds[q], bounds[q + 1]
        c_lo, c_hi = float(cp[lo]), float(cp[hi])
        frac = (target - cum[q]) / var[q]
        want = c_lo + frac * (c_hi - c_lo)
        # first row boundary whose cost prefix reaches `want`, inside the block
        r = int(torch.searchsorted(cp[lo:hi + 1].contiguous(), torch.tensor([want], dtype=torch.float64, device=cp.device))[0]) + lo
        new.append(min(max(r, new[-1]), n))
    new.append(n)
    return new


def row_products(a_row, a_col, b_rowlen, n_rows):
    """P_r = sum over tuples (r,k) of A of the length of B row k. int64 [n_rows]."""
    out = torch.zeros(n_rows, dtype=torch.int64, device=a_row.device)
    out.index_add_(0, a_row.long(), b_rowlen[a_col.long()].to(torch.int64))
    return out


def reduce_digest(count, digest_sum, digest_hash, device, group=None):
    """Whole-job digest from the per-rank ones (count and the mod-2^64 index
    hash add as integers, the value sums as doubles)."""
    lo, hi = digest_hash & 0xFFFFFFFF, digest_hash >> 32          # 32-bit halves: the int64 sums cannot overflow
    ints = torch.tensor([count, lo, hi], dtype=torch.int64, device=device)
    flt = torch.tensor([digest_sum], dtype=torch.float64, device=device)
    dist.all_reduce(ints, group=group)
    dist.all_reduce(flt, group=group)
    h = ((int(ints[2]) << 32) + int(ints[1])) % (1 << 64)
    return int(ints[0]), float(flt[0]), h


def host_transport(ctx, world, group=None):
    """An spsamd_alltoallv_fn for capi.Dist that moves the device buffers through host memory and a
    torch.distributed all-to-all (gloo): lets several ranks share ONE GPU in the tests, where RCCL refuses
    two ranks on a device.  Not a measurement path."""
    import numpy as np

    def xfer(send, sendb, recv, recvb, _stream):
        ins = []
        for p in range(world):
            buf = np.empty(max(int(sendb[p]), 1), np.uint8)
            if sendb[p]:
                ctx.memcpy(buf.ctypes.data, send[p], int(sendb[p]))
            ins.append(torch.from_numpy(buf[:int(sendb[p])]))
        flat_in = torch.cat(ins) if ins else torch.empty(0, dtype=torch.uint8)
        flat_out = torch.empty(int(sum(recvb)), dtype=torch.uint8)
        dist.all_to_all_single(flat_out, flat_in, output_split_sizes=[int(x) for x in recvb],
                               input_split_sizes=[int(x) for x in sendb], group=group)
        out = flat_out.numpy()
        o = 0
        for p in range(world):
            if recvb[p]:
                chunk = np.ascontiguousarray(out[o:o + int(recvb[p])])
                ctx.memcpy(recv[p], chunk.ctypes.data, int(recvb[p]))
            o += int(recvb[p])
    return xfer
