"""Worker of tests/test_gpu_parity.py::test_dist_multiply_two_ranks: two REAL ranks (torch.distributed.run, gloo), both on
cuda:0 -- RCCL refuses two ranks on one device, so the exchange rounds of spsamd_dist_multiply go through its transport
callback (spsparse_amd.dist.host_transport: device buffers -> host -> gloo all-to-all -> device).  Everything else is the
C code a multi-GPU run executes: block consolidation, status / mask / row-length round, pack kernel, panel, block product.

Every scenario is checked on every rank against the single-GPU product of the WHOLE matrices on the same context (itself
pinned to the oracle by the other tests) restricted to the rank's rows, and the first ones against the oracle directly.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from oracle import binding as orc
    from spsparse_amd import capi, dist as sd, workloads as wl

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    dd = capi.Dist(ctx, rank, world, transport=sd.host_transport(ctx, world))
    rng = np.random.default_rng(12345)                   # the same stream on every rank

    def rand_mat(shape, nnz, signed=True, nan=0):
        i0 = rng.integers(0, shape[0], nnz).astype(np.int32)
        i1 = rng.integers(0, shape[1], nnz).astype(np.int32)
        v = rng.uniform(-1 if signed else 0.1, 1, nnz)
        v[rng.integers(0, nnz, max(1, nnz // 12))] = 0.0
        if nan:
            v[rng.integers(0, nnz, nan)] = np.nan
        return i0, i1, v, shape

    def rand_vec(n, density=0.8):
        idx = np.flatnonzero(rng.uniform(size=n) < density).astype(np.int32)
        if idx.size == 0:
            idx = np.array([0], np.int32)
        v = rng.uniform(0.5, 2.0, idx.size)
        if idx.size > 3:
            v[2] = 0.0
        return idx, v

    def cut(n, parts):
        b = sorted(int(x) for x in rng.integers(0, n + 1, parts - 1))
        return [0] + b + [n]

    def block(m, dim, lo, hi):
        keep = (m[dim] >= lo) & (m[dim] < hi)
        return m[0][keep], m[1][keep], m[2][keep], m[3]

    def whole(a, b, tA, tB, scales, C_, dup, zn, flags=0):
        sa, ka = capi.host_coo(a[0], a[1], a[2], a[3])
        sb, kb = capi.host_coo(b[0], b[1], b[2], b[3])
        sv = [None if s is None else capi.host_vec(s[0], s[1], s[2]) for s in scales]
        r = ctx.multiply(sa, sb, C_, None if sv[0] is None else sv[0][0], tA, None if sv[1] is None else sv[1][0], tB,
                         None if sv[2] is None else sv[2][0], dup, zn, capi.SINK_COO, flags)
        return ctx.fetch(r)

    def sharded(a, b, tA, tB, scales, C_, dup, zn, a_bounds, b_bounds, same=False, flags=0, sink=capi.SINK_COO):
        adim = 1 if tA == "T" else 0
        bdim = 1 if tB == "T" else 0
        ab = block(a, adim, a_bounds[rank], a_bounds[rank + 1])
        sa, ka = capi.host_coo(ab[0], ab[1], ab[2], ab[3])
        sb = None
        if not same:
            bb = block(b, bdim, b_bounds[rank], b_bounds[rank + 1])
            sb, kb = capi.host_coo(bb[0], bb[1], bb[2], bb[3])
        sv = [None if s is None else capi.host_vec(s[0], s[1], s[2]) for s in scales]
        res, st = dd.multiply(sa, sb, b_bounds, C_, None if sv[0] is None else sv[0][0], tA, None if sv[1] is None else sv[1][0], tB,
                              None if sv[2] is None else sv[2][0], dup, zn, sink, flags)
        return res, st

    def rows_of(got, lo, hi):
        keep = (got[0] >= lo) & (got[0] < hi)
        return got[0][keep], got[1][keep], got[2][keep]

    def same_tuples(x, y, exact):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), "index sets differ"
        if exact:
            assert np.array_equal(x[2], y[2], equal_nan=True)
        else:
            ok = np.isclose(x[2], y[2], rtol=1e-12, atol=0, equal_nan=True)
            assert ok.all(), "values differ"

    n_cases = 0
    # ---- 1. every argument of the reference's multiply, rectangular operands, B != A, random cuts
    for trial in range(16):
        m, k, n = [int(x) for x in rng.integers(3, 70, 3)]
        tA, tB = ".T"[trial % 2], ".T"[(trial // 2) % 2]
        a = rand_mat((k, m) if tA == "T" else (m, k), int(rng.integers(1, 900)))
        b = rand_mat((n, k) if tB == "T" else (k, n), int(rng.integers(1, 900)))
        si = (*rand_vec(m), m) if trial % 2 else None
        sj = (*rand_vec(k), k) if trial % 3 else None
        sk = (*rand_vec(n), n) if trial % 4 == 1 else None
        dup = [capi.ADD, capi.LEAVE_ALONE, capi.REPLACE][trial % 3]
        a_bounds, b_bounds = cut(m, world), cut(k, world)
        want = whole(a, b, tA, tB, (si, sj, sk), 3.0, dup, False, flags=capi.SINK_ORDERED)
        res, st = sharded(a, b, tA, tB, (si, sj, sk), 3.0, dup, False, a_bounds, b_bounds, flags=capi.SINK_ORDERED)
        got = ctx.fetch(res)
        same_tuples(got, rows_of(want, a_bounds[rank], a_bounds[rank + 1]), exact=True)
        assert (res.shape0, res.shape1) == (m, n)
        if trial < 6:                                     # ... and against the oracle itself
            A = orc.Mat(a[0], a[1], a[2], a[3]); B = orc.Mat(b[0], b[1], b[2], b[3])
            vec = lambda s: None if s is None else orc.Vec(s[0], s[1], s[2])
            w = orc.multiply(A, B, 3.0, vec(si), tA, vec(sj), tB, vec(sk), dup, rowwise=True)
            same_tuples(got, rows_of(w, a_bounds[rank], a_bounds[rank + 1]), exact=True)
        n_cases += 1

    # ---- 2. zero_nan with NaNs: the leading run is a property of the whole matrix; A*A through B_block = NULL too
    for trial in range(12):
        m = int(rng.integers(4, 40))
        tA = tB = ".T"[trial % 2]
        a = rand_mat((m, m), int(rng.integers(5, 400)), nan=int(rng.integers(1, 30)))
        same = trial % 3 != 2
        b = a if same else rand_mat((m, m), int(rng.integers(5, 400)), nan=int(rng.integers(1, 30)))
        bounds = cut(m, world)
        for zn in (True, False):
            want = whole(a, b, tA, tB, (None, None, None), 1.0, capi.ADD, zn, flags=capi.SINK_ORDERED)
            res, st = sharded(a, b, tA, tB, (None, None, None), 1.0, capi.ADD, zn, bounds, bounds, same=same, flags=capi.SINK_ORDERED)
            same_tuples(ctx.fetch(res), rows_of(want, bounds[rank], bounds[rank + 1]), exact=True)
            n_cases += 1

    # ---- 3. R-MAT A*A with heavy rows (whole-block panels), digest sink: the ranks' digests add up to the whole product's
    a = wl.rmat(13, seed=5)
    P = sd.row_products(torch.from_numpy(a[0]), torch.from_numpy(a[1]), torch.bincount(torch.from_numpy(a[0]).long(), minlength=a[3][0]), a[3][0])
    bounds = sd.product_balanced_bounds(P, world)
    sa, ka = capi.host_coo(a[0], a[1], a[2], a[3])
    d1 = ctx.multiply(sa, sa, sink=capi.SINK_DIGEST)
    res, st = sharded(a, a, ".", ".", (None, None, None), 1.0, capi.ADD, False, bounds, bounds, same=True, sink=capi.SINK_DIGEST)
    tot = sd.reduce_digest(int(res.nnz), float(res.sum), int(res.hash), torch.device("cpu"))
    assert tot[0] == d1.nnz and tot[2] == d1.hash and abs(tot[1] - d1.sum) <= 1e-12 * abs(d1.sum)
    assert st.remote_tuples > 0 and st.panel_tuples >= st.remote_tuples
    n_cases += 1

    # ---- 4. the Galerkin chain R*A*R^T cut at z planes: T chained in place, 'T' on the second product's B, a one-plane halo
    g = 16
    nc = g // 2
    A3, R3 = wl.laplace3d(g), wl.aggregation3d(g)
    zc = [nc * q // world for q in range(world + 1)]
    bc, bf = [z * nc * nc for z in zc], [2 * z * g * g for z in zc]
    Rr = block(R3, 0, bc[rank], bc[rank + 1]); Ar = block(A3, 0, bf[rank], bf[rank + 1]); Rc = block(R3, 1, bf[rank], bf[rank + 1])
    sRr, k1 = capi.host_coo(*Rr); sAr, k2 = capi.host_coo(*Ar); sRc, k3 = capi.host_coo(*Rc)
    rt, st1 = dd.multiply(sRr, sAr, bf, sink=capi.SINK_COO)
    assert st1.remote_tuples == 0                        # the fine cells of an aggregate lie in the rank's own planes
    rc, st2 = dd.multiply(capi.result_operand(rt), sRc, bf, tB="T", sink=capi.SINK_COO)
    assert (st2.remote_tuples > 0) == (world > 1)
    got = ctx.fetch(rc)
    sR, k4 = capi.host_coo(*R3); sA, k5 = capi.host_coo(*A3)
    wt = ctx.multiply(sR, sA, sink=capi.SINK_COO)
    wc = ctx.fetch(ctx.multiply(capi.result_operand(wt), sR, tB="T", sink=capi.SINK_COO))
    same_tuples(got, rows_of(wc, bc[rank], bc[rank + 1]), exact=True)
    assert set(np.unique(got[2]).tolist()) <= {24.0, -4.0}
    n_cases += 1

    # ---- 5. errors are agreed on: a mis-cut B block on ONE rank, an index out of bounds on ONE rank -- every rank returns an
    # error (its own, or EPEER) instead of one returning and the other waiting for ever; the communicator stays usable
    m = 30
    a = rand_mat((m, m), 300)
    bounds = cut(m, world)
    bad_rank = world - 1
    for kind in (("miscut", "index") if world > 1 else ("index",)):
        ab = block(a, 0, bounds[rank], bounds[rank + 1])
        bb = block(a, 0, bounds[rank], bounds[rank + 1])
        if rank == bad_rank and kind == "miscut":
            bb = a                                        # the whole matrix: rows outside this rank's bounds
        if rank == bad_rank and kind == "index":
            if ab[0].size == 0:
                ab = (np.array([0], np.int32), np.array([0], np.int32), np.array([1.0]), ab[3])
            i0 = ab[0].copy()
            i0[0] = m + 5                                 # VectorCooArray::add would reject it (VectorCooArray.hpp:246-262)
            ab = (i0, ab[1], ab[2], ab[3])
        sa, ka = capi.host_coo(*ab); sb, kb = capi.host_coo(*bb)
        try:
            dd.multiply(sa, sb, bounds, sink=capi.SINK_COO)
            raise AssertionError("rank %d: the step should have failed (%s)" % (rank, kind))
        except capi.SpsamdError as e:
            want_code = -2 if rank == bad_rank else -7
            if world == 1:
                want_code = -2
            assert e.code == want_code, (rank, kind, e.code, e.msg)
        n_cases += 1
    # ... and still works
    want = whole(a, a, ".", ".", (None, None, None), 1.0, capi.ADD, False, flags=capi.SINK_ORDERED)
    res, st = sharded(a, a, ".", ".", (None, None, None), 1.0, capi.ADD, False, bounds, bounds, flags=capi.SINK_ORDERED)
    same_tuples(ctx.fetch(res), rows_of(want, bounds[rank], bounds[rank + 1]), exact=True)
    n_cases += 1

    dist.barrier()
    dd.close()
    ctx.close()
    dist.destroy_process_group()
    print("rank %d: %d sharded cases ok" % (rank, n_cases), flush=True)


if __name__ == "__main__":
    main()
