"""The multi-GPU path's host logic on CPU: world_size 2 over gloo.

Row-block partition (spsparse_amd/dist.py: cost-balanced bounds, digest reduction) + a torch model of the all-to-allv of
needed B row panels (exchange_b_panels below: what csrc/dist.hip does on the device, restated here because no GPU is in
this container -- the C code itself is driven by two real ranks in tests/dist_worker.py on the GPU box); each rank's
block product is computed by the ORACLE and the concatenation must equal the oracle's product of the whole matrices --
the property the sharding relies on: C's rows are independent (multiply_sparse.hpp:192), no reduction.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import binding as orc
from spsparse_amd import dist as sd
from spsparse_amd import workloads as wl


def exchange_b_panels(a_col, b_row, b_col, b_val, bounds, n_inner, group=None, whole_block_fraction=0.5):
    """All-to-allv of the B row panels this rank's A block needs.

    a_col           inner indices k of this rank's A block tuples
    b_row/col/val   this rank's own block of B (rows bounds[rank]..bounds[rank+1]), row-major sorted
    bounds          row-block boundaries of B over the inner dimension, len world+1
    Returns (row, col, val) of the received panel: the tuples of every B row
    this rank needs (plus, from owners it needs more than `whole_block_fraction`
    of, their whole block), sorted row-major (owner blocks arrive in rank
    order), and the number of tuples received from other ranks.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = b_row.device
    sizes = [bounds[q + 1] - bounds[q] for q in range(world)]
    my_lo, my_n = bounds[rank], sizes[rank]

    # 1. which rows of each owner do I need?  one byte per row of the inner dimension
    need = torch.zeros(n_inner, dtype=torch.uint8, device=dev)
    need[a_col.long()] = 1
    their_need = torch.empty(my_n * world, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(their_need, need, output_split_sizes=[my_n] * world, input_split_sizes=sizes, group=group)

    # 2. pack, per requester, the tuples of my rows it asked for.  A requester that needs most of
    #    my rows (R-MAT blocks need 80-98 % of B) gets the whole block: no per-tuple selection, and
    #    rows it did not ask for are simply never referenced by its A block.
    local_row = (b_row.long() - my_lo)
    masks = their_need.view(world, my_n) if my_n else their_need.view(world, 0)
    frac = (masks.sum(dim=1).to(torch.float64) / max(my_n, 1)).tolist()
    sel = []
    for p in range(world):
        if frac[p] >= whole_block_fraction:
            sel.append(None)
        else:
            sel.append(torch.nonzero(masks[p][local_row], as_tuple=False).flatten())
    counts = [b_row.numel() if s_ is None else s_.numel() for s_ in sel]
    send_counts = torch.tensor(counts, dtype=torch.int64, device=dev)
    recv_counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_counts, send_counts, group=group)

    def pack(x):
        return torch.cat([x if s_ is None else x[s_] for s_ in sel]) if world > 1 else (x if sel[0] is None else x[sel[0]])
    s_row, s_col, s_val = pack(b_row), pack(b_col), pack(b_val)
    in_splits = counts
    out_splits = [int(x) for x in recv_counts.tolist()]
    total = sum(out_splits)

    # 3. the all-to-allv proper: (row, col, val) panels
    r_row = torch.empty(total, dtype=b_row.dtype, device=dev)
    r_col = torch.empty(total, dtype=b_col.dtype, device=dev)
    r_val = torch.empty(total, dtype=b_val.dtype, device=dev)
    dist.all_to_all_single(r_row, s_row, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    dist.all_to_all_single(r_col, s_col, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    dist.all_to_all_single(r_val, s_val, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    remote = total - out_splits[rank]
    return r_row, r_col, r_val, remote



def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if kind == "rmat":
            a = wl.rmat(9, seed=3)
            b = a
        else:                       # rectangular, B != A: R * A of the Galerkin product
            a = wl.aggregation3d(8)
            b = wl.laplace3d(8)
        n_rows, n_inner = a[3]
        # setup (untimed in the bench): consolidated operands, product-balanced row blocks
        a0, a1, av = orc.consolidate(a[0], a[1], a[2], 0)
        b0, b1, bv = orc.consolidate(b[0], b[1], b[2], 0)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        b_rowlen = torch.bincount(t(b0).long(), minlength=n_inner)
        P = sd.row_products(t(a0), t(a1), b_rowlen, n_rows)
        a_bounds = sd.product_balanced_bounds(P, world)
        b_bounds = sd.product_balanced_bounds(b_rowlen, world)         # B's own distribution over the inner dim
        assert a_bounds[0] == 0 and a_bounds[-1] == n_rows and all(x <= y for x, y in zip(a_bounds, a_bounds[1:]))
        shares = [int(P[a_bounds[q]:a_bounds[q + 1]].sum()) for q in range(world)]
        assert max(shares) <= 0.75 * int(P.sum()) + int(P.max())          # balanced up to one row

        ma = (a0 >= a_bounds[rank]) & (a0 < a_bounds[rank + 1])
        mb = (b0 >= b_bounds[rank]) & (b0 < b_bounds[rank + 1])
        # exact panels (whole_block_fraction > 1): exactly the B rows this block needs, row-major sorted
        r_row, r_col, r_val, remote = exchange_b_panels(t(a1[ma]), t(b0[mb]), t(b1[mb]), t(bv[mb]), b_bounds, n_inner,
                                                           whole_block_fraction=2.0)
        needed = np.unique(a1[ma])
        want_mask = np.isin(b0, needed)
        assert np.array_equal(r_row.numpy(), b0[want_mask])
        assert np.array_equal(r_col.numpy(), b1[want_mask]) and np.array_equal(r_val.numpy(), bv[want_mask])
        assert remote == int(np.sum(want_mask & ~mb))
        # default: owners this block needs most of send their whole block -- a sorted superset
        r_row, r_col, r_val, remote = exchange_b_panels(t(a1[ma]), t(b0[mb]), t(b1[mb]), t(bv[mb]), b_bounds, n_inner)
        got_rows = r_row.numpy().astype(np.int64)
        assert np.all(np.diff(got_rows) >= 0)
        ncolb = b[3][1]
        got_keys = got_rows * ncolb + r_col.numpy()
        want_keys = b0[want_mask].astype(np.int64) * ncolb + b1[want_mask]
        assert np.all(np.isin(want_keys, got_keys)) and got_keys.size >= want_keys.size

        A_blk = orc.Mat(a0[ma], a1[ma], av[ma], a[3], sort0=0)
        B_pan = orc.Mat(r_row.numpy(), r_col.numpy(), r_val.numpy(), b[3], sort0=0)
        ci, cj, cv, _ = orc.multiply(A_blk, B_pan, rowwise=True)
        cnt, s, h = orc.digest(ci, cj, cv)
        tot = sd.reduce_digest(cnt, s, h, torch.device("cpu"))
        np.savez(os.path.join(out_dir, "part%d.npz" % rank), i=ci, j=cj, v=cv, tot=np.array([tot[0], tot[2]], dtype=np.uint64),
                 tots=np.array([tot[1]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["rmat", "galerkin"])
def test_row_block_sharding_world2(tmp_path, kind):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "part%d.npz" % r)) for r in range(world)]
    if kind == "rmat":
        a = wl.rmat(9, seed=3)
        b = a
    else:
        a, b = wl.aggregation3d(8), wl.laplace3d(8)
    wi, wj, wv, _ = orc.multiply(orc.Mat(*a), orc.Mat(*b), rowwise=True)
    gi = np.concatenate([p["i"] for p in parts])
    gj = np.concatenate([p["j"] for p in parts])
    gv = np.concatenate([p["v"] for p in parts])
    # blocks are contiguous and ordered: plain concatenation is the row-major product
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gv, wv)
    cnt, s, h = orc.digest(wi, wj, wv)
    for p in parts:
        assert int(p["tot"][0]) == cnt and int(p["tot"][1]) == h
        assert abs(float(p["tots"][0]) - s) <= 1e-12 * abs(s)


def test_row_cost_weights():
    P = torch.tensor([0, 10, 64, 65, 4096, 4097, 70000, 2000000], dtype=torch.int64)
    c = sd.row_cost(P)
    assert c.tolist() == [0, 160, 1024, 650, 40960, 19666, 336000, 14000000]
    b = sd.product_balanced_bounds(c, 2)
    assert b[0] == 0 and b[-1] == 8 and 0 < b[1] <= 8


def test_rebalance_from_measured_times():
    """Blocks whose true cost deviates from the estimate by a block-dependent factor (plus a fixed
    part) become equal in time after a few measure/rebalance rounds."""
    g = torch.Generator().manual_seed(5)
    n, world = 20000, 8
    est = torch.randint(1, 1000, (n,), generator=g, dtype=torch.int64)
    est[:50] *= 100                                            # a few hub rows, like an un-permuted R-MAT
    skew = 0.5 + 4.0 * torch.linspace(1, 0, n, dtype=torch.float64) ** 2      # true cost per estimated unit varies along the rows
    true = est.to(torch.float64) * skew
    tp = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(true, 0)])
    cp = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(est, 0)])
    fixed = 0.02 * float(tp[-1]) / world

    def measure(b):
        return [fixed + float(tp[b[q + 1]] - tp[b[q]]) for q in range(world)]
    b = sd.product_balanced_bounds(est, world)
    t0 = measure(b)
    assert max(t0) / (sum(t0) / world) > 1.2                    # the estimate alone is badly off
    for _ in range(3):
        b = sd.rebalance_bounds(b, cp, measure(b), fixed_ms=0.0)
        assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
    t2 = measure(b)
    assert max(t2) / (sum(t2) / world) < 1.06
    # equal measured times leave the boundaries where they are (up to one row)
    b2 = sd.rebalance_bounds(b, cp, [1.0] * world)
    assert all(abs(x - y) <= 1 for x, y in zip(b2, b))
    # within the noise threshold nothing moves at all
    assert sd.rebalance_bounds(b, cp, [1.0, 1.02] + [1.0] * (world - 2), min_gain=0.03) == b


def test_balanced_bounds_edge_cases():
    P = torch.tensor([0, 0, 10, 0, 5, 5, 0], dtype=torch.int64)
    b = sd.product_balanced_bounds(P, 2)
    assert b[0] == 0 and b[-1] == 7 and int(P[b[0]:b[1]].sum()) == 10
    b = sd.product_balanced_bounds(P, 4)
    assert len(b) == 5 and all(x <= y for x, y in zip(b, b[1:]))
    b = sd.product_balanced_bounds(torch.zeros(5, dtype=torch.int64), 3)
    assert b[0] == 0 and b[-1] == 5
    assert sd.product_balanced_bounds(P, 1) == [0, 7]
