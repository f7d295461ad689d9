"""Parity of the HIP product path (through the C ABI, include/spsparse_amd.h)
with the CPU oracle.  GPU only.

Bars: index sets identical; values bit-exact where the device sums in the
reference's ascending-k order (rows with <= 64 products), within 1e-12
relative (BASELINE.json north_star) where the LDS accumulators add in
arrival order.
"""
import os

import numpy as np
import pytest

from oracle import binding as orc
from spsparse_amd import workloads as wl

pytestmark = pytest.mark.gpu

REL = 1e-12


@pytest.fixture(scope="module")
def ctx():
    from spsparse_amd import capi
    c = capi.Context()
    yield c
    c.close()


def _dev(ctx, A, B, **kw):
    """multiply through the C ABI with host operands; returns (i, j, v, res)."""
    from spsparse_amd import capi
    keep = []

    def coo(M):
        s, k = capi.host_coo(M.idx0, M.idx1, M.val, M.shape, M.sort0)
        keep.append(k)
        return s

    def vec(V):
        if V is None:
            return None
        s, k = capi.host_vec(V.idx, V.val, V.shape0)
        keep.append(k)
        return s

    res = ctx.multiply(coo(A), coo(B), kw.get("C_", 1.0), vec(kw.get("scalei")), kw.get("tA", "."),
                       vec(kw.get("scalej")), kw.get("tB", "."), vec(kw.get("scalek")),
                       kw.get("duplicate_policy", capi.ADD), kw.get("zero_nan", False),
                       kw.get("sink", capi.SINK_COO), kw.get("flags", 0))
    if kw.get("sink", capi.SINK_COO) == capi.SINK_COO:
        i, j, v = ctx.fetch(res)
        return i, j, v, res
    return None, None, None, res


def _check(got, want, exact=False, scale=None):
    gi, gj, gv = got[:3]
    wi, wj, wv = want[:3]
    assert len(gi) == len(wi), (len(gi), len(wi))
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj)
    if exact:
        assert np.array_equal(gv, wv)
    elif len(wv):
        denom = np.abs(wv) if scale is None else scale
        assert np.max(np.abs(gv - wv) / denom) <= REL
    # ascending (i, j), each at most once
    if len(gi) > 1:
        key = gi.astype(np.int64) * (int(gj.max()) + 1) + gj
        assert np.all(np.diff(key) > 0)


def test_known_answer(ctx):
    """tests/test_multiply_sparse.cpp:45-78 -> {(0,0):128, (1,0):60}"""
    row = orc.Mat([0, 0, 0, 0, 1], [8, 4, 0, 3, 8], [6., 4., 2., 3., 3.], (2, 10))
    scale = orc.Vec([0, 4, 8], [2., 4., 4.], 10)
    col = orc.Mat([0, 3, 8], [0, 0, 0], [2., 3., 5.], (10, 1))
    eye = orc.Vec(list(range(10)), [1.] * 10, 10)
    i, j, v, res = _dev(ctx, row, col, scalei=eye, scalej=scale, scalek=eye)
    assert (res.shape0, res.shape1) == (2, 1)
    assert i.tolist() == [0, 1] and j.tolist() == [0, 0] and v.tolist() == [128., 60.]


def test_random_mm_property(ctx):
    """tests/test_multiply_sparse.cpp:84-136 through the device path: the 999 seeded 5x5 cases with
    scalej = eye.  Bit-exact against the oracle (which is itself pinned to the dense triple loop)."""
    from tests.test_oracle_pins import RANDOM5
    mm, _ = RANDOM5
    eye = orc.Vec(list(range(5)), [1.] * 5, 5)
    total = 0
    for seed, ((ai, av), (bi, bv)) in mm:
        A = orc.Mat(ai[0], ai[1], av, (5, 5))
        B = orc.Mat(bi[0], bi[1], bv, (5, 5))
        want = orc.multiply(A, B, 1.0, scalej=eye)
        got = _dev(ctx, A, B, scalej=eye)
        assert (got[3].shape0, got[3].shape1) == (5, 5)
        _check(got, want, exact=True)
        total += len(want[2])
    assert total > 5000


def _dev_mv(ctx, A, V, **kw):
    from spsparse_amd import capi
    a, k1 = capi.host_coo(A.idx0, A.idx1, A.val, A.shape, A.sort0)
    v, k2 = capi.host_vec(V.idx, V.val, V.shape0, V.sort0)
    keep = [k1, k2]

    def vec(S):
        if S is None:
            return None
        s, k = capi.host_vec(S.idx, S.val, S.shape0)
        keep.append(k)
        return s

    res = ctx.multiply_mv(a, v, kw.get("C_", 1.0), vec(kw.get("scalei")), kw.get("tA", "."), vec(kw.get("scalej")),
                          kw.get("duplicate_policy", capi.ADD), kw.get("zero_nan", False))
    i, j, val = ctx.fetch(res)
    assert not j.any()
    return i, val, res


def test_random_mv_property(ctx):
    """tests/test_multiply_sparse.cpp:138-203 through spsamd_multiply_mv: 999 seeded 5x5 cases,
    exact equality like the reference's `sum != Cd(i)` check."""
    from tests.test_oracle_pins import RANDOM5
    _, mv = RANDOM5
    total = 0
    for seed, ((ai, av), (vi, vv)) in mv:
        A = orc.Mat(ai[0], ai[1], av, (5, 5))
        V = orc.Vec(vi[0], vv, 5)
        wi, _, wv, wshape = orc.multiply_mv(A, V)
        gi, gv, res = _dev_mv(ctx, A, V)
        assert (res.shape0, res.shape1) == (5, 0)
        assert np.array_equal(gi, wi) and np.array_equal(gv, wv), seed
        total += len(wv)
    assert total > 500


def test_mv_flags_scales(ctx):
    """MV with 'T', scalei/scalej, C != 1 and duplicate policies against the oracle; error text for V."""
    from spsparse_amd import capi
    rng = np.random.default_rng(9)
    for trial in range(16):
        m, k = rng.integers(1, 80, 2)
        tA = "T" if trial % 2 else "."
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 900)), zeros=True)
        nv = int(rng.integers(1, 2 * k + 1))
        V = orc.Vec(rng.integers(0, k, nv), rng.uniform(0.1, 1, nv), k)
        kw = dict(C_=3.0, tA=tA, scalei=_rand_vec(rng, m) if trial % 3 == 0 else None,
                  scalej=_rand_vec(rng, k) if trial % 3 == 1 else None,
                  duplicate_policy=[orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3])
        wi, _, wv, wshape = orc.multiply_mv(A, V, **kw)
        gi, gv, res = _dev_mv(ctx, A, V, **kw)
        assert res.shape0 == wshape[0]
        assert np.array_equal(gi, wi)
        if len(wv):
            assert np.max(np.abs(gv - wv) / np.abs(wv)) <= REL
    with pytest.raises(capi.SpsamdError, match=r"Inner dimensions for A \(3\) and V \(4\) must match!"):
        _dev_mv(ctx, orc.Mat([0], [1], [2.], (2, 3)), orc.Vec([0], [1.], 4))


def _rand_mat(rng, shape, nnz, zeros=False, positive=True):
    i0 = rng.integers(0, shape[0], nnz)
    i1 = rng.integers(0, shape[1], nnz)
    v = rng.uniform(0.1, 1, nnz) if positive else rng.uniform(-1, 1, nnz)
    if zeros:
        v[rng.integers(0, nnz, max(1, nnz // 10))] = 0.0
    return orc.Mat(i0, i1, v, shape)


def _rand_vec(rng, n, density=0.7):
    idx = np.flatnonzero(rng.uniform(size=n) < density)
    if idx.size == 0:
        idx = np.array([0])
    v = rng.uniform(0.5, 2, idx.size)
    if idx.size > 3:
        v[1] = 0.0
    return orc.Vec(idx, v, n)


@pytest.mark.parametrize("tA", [".", "T"])
@pytest.mark.parametrize("tB", [".", "T"])
def test_flags_scales_policies(ctx, tA, tB):
    """'T' flags x scale vectors x C != 1 x each DuplicatePolicy on <= 60x60 inputs
    (SURVEY 8c golden-vector class iv; the reference's own tests never vary these)."""
    rng = np.random.default_rng(42)
    for trial in range(12):
        m, k, n = rng.integers(1, 60, 3)
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 600)), zeros=True)
        B = _rand_mat(rng, (n, k) if tB == "T" else (k, n), int(rng.integers(1, 600)), zeros=True)
        kw = dict(C_=17.0, tA=tA, tB=tB,
                  scalei=_rand_vec(rng, m) if trial % 2 else None,
                  scalej=_rand_vec(rng, k) if trial % 3 else None,
                  scalek=_rand_vec(rng, n) if trial % 4 == 1 else None,
                  duplicate_policy=[orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3])
        want = orc.multiply(A, B, **kw)
        got = _dev(ctx, A, B, **kw)
        assert (got[3].shape0, got[3].shape1) == want[3]
        _check(got, want)


def test_cfg1_1k(ctx):
    """BASELINE cfg1: 1k x 1k, 10 per row, fp64 (duplicates exercise consolidate-ADD)."""
    for seed in (1, 2):
        A = orc.Mat(*wl.random_rows(1000, 10, seed, 0))
        B = orc.Mat(*wl.random_rows(1000, 10, seed, 8))
        want = orc.multiply(A, B)
        got = _dev(ctx, A, B)
        _check(got, want)
        assert got[3].products > 90000 and 90000 < got[3].nnz < 100000
        # digest sink agrees with the oracle's digest of the same product
        from spsparse_amd import capi
        _, _, _, d = _dev(ctx, A, B, sink=capi.SINK_DIGEST)
        cnt, s, h = orc.digest(*want[:3])
        assert d.nnz == cnt and d.hash == h and abs(d.sum - s) <= REL * abs(s)


@pytest.mark.parametrize("scale", [8, 11, 13, 15])
def test_rmat_full_compare(ctx, scale):
    """R-MAT A*A (cfg2's generator at small scale): all three row classes
    (light / LDS hash / dense windows) against the row-wise oracle, tuple by tuple."""
    a = wl.rmat(scale, seed=1)
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    got = _dev(ctx, A, A)
    _check(got, want)
    res = got[3]
    assert res.products == res.products_light + res.products_mid + res.products_heavy
    if scale >= 11:
        assert res.rows_heavy > 0 and res.rows_mid > 0 and res.rows_light > 0
    if scale >= 15:                      # several 8192-column windows: hash cells, tiles and dense cells
        assert res.cells_hash > 0 and res.cells_dense > 0
    from spsparse_amd import capi
    _, _, _, d = _dev(ctx, A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    cnt, s, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s) <= REL * abs(s)


def test_fuzz_shapes_flags_and_sinks(ctx):
    """Seeded random cases over everything the boundary takes at once: rectangular and degenerate
    shapes (1 x n, n x 1, empty rows and columns), duplicate tuples, explicit zeros, the three
    scale vectors, C != 1, 'T' flags, duplicate policies, and the COO / permuted / digest / ordered
    sinks -- with row products on both sides of the 64 / 4096 class boundaries and column counts on
    both sides of one 8192-column window.  Index sets identical, values <= 1e-12 relative."""
    from spsparse_amd import capi
    rng = np.random.default_rng(int(os.environ.get("SPSAMD_FUZZ_SEED", "20240917")))      # other seeds: soak runs
    shapes = [(1, 1, 1), (1, 40, 1), (5, 1, 7), (3, 300, 9000), (60, 60, 60), (200, 17, 20000), (17, 400, 12000),
              (2, 3000, 9000), (64, 64, 70000), (9, 2500, 40000), (30, 2000, 3000000), (4, 5000, 2500000)]   # last two: 16384-column windows
    seen = dict(light=0, mid=0, heavy=0, hash_cells=0, dense_cells=0, empty=0)
    for case in range(120):
        m, k, n = shapes[(case * 7 + case // 10) % len(shapes)]
        dens_a = rng.choice([0.02, 0.2, 0.9])
        dens_b = rng.choice([0.002, 0.05, 0.4]) if n > 5000 else rng.choice([0.05, 0.5])
        nnz_a = max(1, int(m * k * dens_a))
        nnz_b = max(1, min(int(k * n * dens_b), 600000))
        tA, tB = rng.choice([".", "T"]), rng.choice([".", "T"])
        sa = (k, m) if tA == "T" else (m, k)
        sb = (n, k) if tB == "T" else (k, n)
        A = _rand_mat(rng, sa, nnz_a, zeros=bool(case % 3 == 0), positive=bool(case % 4))
        B = _rand_mat(rng, sb, nnz_b, zeros=bool(case % 5 == 0), positive=bool(case % 4))
        kw = dict(tA=tA, tB=tB, C_=float(rng.choice([1.0, -0.5, 3.0])),
                  duplicate_policy=int(rng.choice([capi.ADD, capi.LEAVE_ALONE, capi.REPLACE])))
        if case % 2:
            kw["scalei"] = _rand_vec(rng, m)
        if case % 3 == 1:
            kw["scalej"] = _rand_vec(rng, k)
        if case % 4 == 2:
            kw["scalek"] = _rand_vec(rng, n)
        okw = dict(kw)
        want = orc.multiply(A, B, rowwise=True, nthreads=8, **okw)
        mixed = not bool(case % 4)                              # values of both signs: sums cancel
        ordered = mixed or case % 4 == 3
        got = _dev(ctx, A, B, flags=capi.SINK_ORDERED if ordered else 0, **kw)
        _check(got, want, exact=ordered)
        if mixed:                                               # default (arrival-order) sums: same index set, values to rounding
            g2 = _dev(ctx, A, B, **kw)
            assert np.array_equal(g2[0], want[0]) and np.array_equal(g2[1], want[1])
            assert np.allclose(g2[2], want[2], rtol=1e-9, atol=1e-9)
        assert (got[3].shape0, got[3].shape1) == (m, n)
        r_ = got[3]
        seen["light"] += r_.rows_light > 0; seen["mid"] += r_.rows_mid > 0; seen["heavy"] += r_.rows_heavy > 0
        seen["hash_cells"] += r_.cells_hash > 0; seen["dense_cells"] += r_.cells_dense > 0; seen["empty"] += r_.nnz == 0
        pi, pj, pv, pres = _dev(ctx, A, B, flags=capi.SINK_PERMUTE, **kw)
        assert np.array_equal(pi, got[1]) and np.array_equal(pj, got[0]) and (pres.shape0, pres.shape1) == (n, m)
        _, _, _, d = _dev(ctx, A, B, sink=capi.SINK_DIGEST, **kw)
        cnt, ssum, h = orc.digest(*want[:3])
        assert d.nnz == cnt and d.hash == h
    print("fuzz coverage:", seen)
    assert seen["light"] >= 5 and seen["mid"] >= 5 and seen["heavy"] >= 5 and seen["hash_cells"] >= 3 and seen["dense_cells"] >= 3


def test_fuzz_nan_values_and_zero_nan(ctx):
    """NaN operand values with zero_nan on and off (consolidate's literal behaviour: a NaN is dropped only
    where it leads a run of equal indices, algorithm.hpp:272-275 vs :284-292; surviving NaNs poison their
    sums).  Index sets identical; values identical including the NaN positions (ordered mode)."""
    from spsparse_amd import capi
    rng = np.random.default_rng(77)
    for case in range(24):
        m, k, n = [(6, 9, 7), (40, 300, 9000), (3, 2500, 12000), (120, 50, 60)][case % 4]
        A = _rand_mat(rng, (m, k), max(1, int(m * k * 0.3)), zeros=True)
        B = _rand_mat(rng, (k, n), max(1, min(int(k * n * 0.05), 200000)), zeros=bool(case % 2))
        for M in (A, B):
            M.val[rng.integers(0, M.val.size, max(1, M.val.size // 50))] = np.nan
        kw = dict(zero_nan=bool(case % 3), duplicate_policy=int(rng.choice([capi.ADD, capi.LEAVE_ALONE, capi.REPLACE])))
        want = orc.multiply(A, B, rowwise=True, nthreads=8, **kw)
        got = _dev(ctx, A, B, flags=capi.SINK_ORDERED, **kw)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[2], want[2], equal_nan=True)
        g2 = _dev(ctx, A, B, **kw)                              # arrival-order sums: NaN stays NaN, the rest to rounding
        assert np.array_equal(g2[0], want[0]) and np.array_equal(g2[1], want[1])
        assert np.array_equal(np.isnan(g2[2]), np.isnan(want[2]))
        ok = ~np.isnan(want[2])
        assert np.allclose(g2[2][ok], want[2][ok], rtol=1e-12, atol=0)


def test_zero_nan_follows_the_references_sequence_for_b(ctx):
    """which NaNs zero_nan drops from B depends on the reference's COLUMN-major sequence of B
    (multiply_sparse.hpp:168, algorithm.hpp:272-275).  Small shapes against the FAITHFUL restatement
    (inner-product loops, rowwise=False), all transposes and policies."""
    from spsparse_amd import capi
    A = orc.Mat([0, 0], [0, 1], [1., 1.], (1, 2))
    B = orc.Mat([0, 1], [1, 0], [np.nan, 5.], (2, 2))
    got = _dev(ctx, A, B, zero_nan=True, flags=capi.SINK_ORDERED)
    assert got[0].tolist() == [0, 0] and got[1].tolist() == [0, 1] and got[2][0] == 5.0 and np.isnan(got[2][1])
    rng = np.random.default_rng(12)
    for trial in range(40):
        m, k, n = rng.integers(1, 14, 3)
        tA, tB = ".T"[trial % 2], ".T"[(trial // 2) % 2]
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 80)), zeros=True)
        B = _rand_mat(rng, (n, k) if tB == "T" else (k, n), int(rng.integers(1, 80)), zeros=True)
        for M in (A, B):
            M.val[rng.integers(0, M.val.size, max(1, M.val.size // 4))] = np.nan
        dup = [capi.ADD, capi.LEAVE_ALONE, capi.REPLACE][trial % 3]
        for zn in (False, True):
            want = orc.multiply(A, B, 1.0, None, tA, None, tB, None, dup, zero_nan=zn)          # faithful loops
            got = _dev(ctx, A, B, tA=tA, tB=tB, duplicate_policy=dup, zero_nan=zn, flags=capi.SINK_ORDERED)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.array_equal(got[2], want[2], equal_nan=True)
        if trial % 4 == 0:                                       # the same matrix on both sides, row-major = A's order
            S = _rand_mat(rng, (9, 9), 40, zeros=True)
            S.val[rng.integers(0, S.val.size, 8)] = np.nan
            want = orc.multiply(S, S, zero_nan=True)
            got = _dev(ctx, S, S, zero_nan=True, flags=capi.SINK_ORDERED)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.array_equal(got[2], want[2], equal_nan=True)


def test_bench_dist_path_matches_plain_path():
    """bench.py's N>1 code (row block, calibration rounds, all-to-allv of B panels over RCCL, digest
    reduction) rehearsed on a 1-rank NCCL group: same digest as the plain single-GPU path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
              "--no-cpu-baseline", "--scale", "16"]
    outs = []
    for extra in ([], ["--dist-path", "--calibrate", "1"]):
        p = subprocess.run(common + extra, cwd=root, env=env, capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    a, b = outs
    assert a["config"]["nnz_c"] == b["config"]["nnz_c"] and a["config"]["digest"]["hash"] == b["config"]["digest"]["hash"]
    assert abs(a["config"]["digest"]["sum"] - b["config"]["digest"]["sum"]) <= 1e-12 * abs(a["config"]["digest"]["sum"])
    assert b["config"]["calibration_local_ms"] and len(b["config"]["calibration_local_ms"][0]) == 1
    for d in outs:
        assert d["unit"] == "nnz(C)/s" and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1


@pytest.mark.parametrize("wl_args", [["--scale", "16"], ["--workload", "poisson", "--grid", "512"], ["--workload", "galerkin", "--grid", "64"]],
                         ids=["rmat16", "poisson512", "galerkin64"])
def test_bench_two_ranks_on_one_gpu(wl_args):
    """The multi-rank control flow of bench.py with a REAL second rank, started the way the driver starts a scaling run --
    the bare `python bench.py --gpus 2` (bench.py launches its own ranks) -- both on cuda:0, collectives over gloo on host
    copies (--rehearse-gloo).  Distinct row blocks per rank, calibration + rebalancing, panel exchange between different
    owners, digest reduction: the whole-job digest must equal the single-GPU one.  galerkin: BASELINE cfg5's two chained
    products, the second with 'T' on B, cut at z planes (no remote rows for R*A, a one-plane halo for T*R^T)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    base = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-configs"] + wl_args       # R-MAT: whole-block panels; stencil: exact panels
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + base, cwd=root, env=env,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    one = json.loads(p.stdout.strip().splitlines()[-1])
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--calibrate", "2"] + base,
                       cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    two = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["config"]["remote_panel_tuples"] > 0
    assert two["config"]["nnz_c"] == one["config"]["nnz_c"] and two["config"]["digest"]["hash"] == one["config"]["digest"]["hash"]
    assert abs(two["config"]["digest"]["sum"] - one["config"]["digest"]["sum"]) <= 1e-12 * abs(one["config"]["digest"]["sum"])
    if "galerkin" in wl_args:
        g = 64
        assert two["config"]["nnz_c"] == 7 * (g // 2) ** 3 - 6 * (g // 2) ** 2
        assert two["config"]["remote_panel_tuples"] == 2 * g * g          # one fine plane of R^T from the neighbour, each way
        return
    assert two["config"]["products"] == one["config"]["products"] and two["config"]["nnz_a"] == one["config"]["nnz_a"]
    assert len(two["config"]["calibration_local_ms"]) == 2 and len(two["config"]["calibration_local_ms"][0]) == 2


def test_dist_multiply_two_ranks():
    """spsamd_dist_multiply between two real ranks sharing this GPU (tests/dist_worker.py: the transport callback over gloo):
    all of the reference's arguments ('T' flags, three scale vectors, C, DuplicatePolicy) on rectangular operands with
    random cuts, zero_nan with NaNs, R-MAT whole-block panels, the Galerkin chain with 'T', and the agreement on errors
    (a mis-cut block or a bad index on one rank: EINVAL there, EPEER on the other, and the communicator keeps working)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29591", os.path.join(root, "tests", "dist_worker.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "rank 0:" in p.stdout and "rank 1:" in p.stdout and "sharded cases ok" in p.stdout


def test_dist_step_one_rank_rccl(ctx):
    """spsamd_dist_multiply (C ABI) on a 1-rank RCCL communicator created from a unique id: the whole step --
    consolidate the block, masks, pack, grouped ncclSend/ncclRecv to itself, panel, block product -- must give the
    single-GPU digest; raw tuples with duplicates in, B = A and B != A."""
    from spsparse_amd import capi
    a = wl.rmat(13, seed=4)
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    cnt, ssum, h = orc.digest(*want[:3])
    d = capi.Dist(ctx, 0, 1, unique_id=capi.Dist.unique_id())
    try:
        s, keep = capi.host_coo(*a)
        res, st = d.multiply(s, None, [0, a[3][1]])
        assert (res.nnz, res.hash) == (cnt, h) and abs(res.sum - ssum) <= REL * abs(ssum)
        assert st.remote_tuples == 0 and st.panel_tuples == res.nnz_b and st.block_nnz_a == res.nnz_a
        b = wl.rmat(13, seed=9)
        B = orc.Mat(*b)
        want2 = orc.multiply(A, B, rowwise=True, nthreads=8)
        cnt2, ssum2, h2 = orc.digest(*want2[:3])
        sb, keepb = capi.host_coo(*b)
        res2, st2 = d.multiply(s, sb, [0, b[3][0]])
        assert (res2.nnz, res2.hash) == (cnt2, h2) and abs(res2.sum - ssum2) <= REL * abs(ssum2)
        # the COO sink through the same entry point: tuples with global indices, in order
        res3, _ = d.multiply(s, sb, [0, b[3][0]], sink=capi.SINK_COO)
        gi, gj, gv = ctx.fetch(res3)
        _check((gi, gj, gv), want2)
    finally:
        d.close()


def test_prepared_operands_equal_plain_ones(ctx):
    """spsamd_operand_prepare: an operand consolidated once, its row structure / packed tuples / window indices built by the
    first product that needs them and kept (the reference's Consolidate<> fast path and lazy dim_beginnings cache,
    algorithm.hpp:360, VectorCooArray.hpp:325-335).  Results equal the plain operand's in every sink, on both sides, with
    scale vectors and 'T'; the second product with the same handle does not rebuild what the first one built."""
    from spsparse_amd import capi
    a = wl.rmat(14, seed=6)                                   # duplicates, heavy rows, dense and hash cells
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    s, keep = capi.host_coo(*a)
    op = capi.Operand(ctx, s, '.', capi.AS_A | capi.AS_B)
    try:
        assert op.coo.mem == capi.MEM_PREPARED and op.coo.nnz < a[0].size        # duplicates were merged once, here
        b0 = op.bytes
        r1 = ctx.multiply(op.coo, op.coo, sink=capi.SINK_COO, flags=capi.SINK_ORDERED)
        _check(ctx.fetch(r1), want, exact=True)
        b1 = op.bytes
        assert b1 >= b0 > 0                                   # the heavy rows' indices were built into the handle (its memory was taken when it was prepared) ...
        d1 = ctx.multiply(op.coo, op.coo, sink=capi.SINK_DIGEST)
        assert op.bytes == b1                                 # ... once
        cnt, ssum, h = orc.digest(*want[:3])
        assert (d1.nnz, d1.hash) == (cnt, h) and abs(d1.sum - ssum) <= REL * abs(ssum)
        assert d1.ms_consolidate < 0.5 and d1.nnz_a == op.coo.nnz
        plain = ctx.multiply(s, s, sink=capi.SINK_DIGEST)
        assert d1.ms_symbolic < plain.ms_symbolic             # nothing of B's indices is rebuilt
        # one side prepared, the other plain; scale vectors; C != 1
        rng = np.random.default_rng(3)
        n = a[3][0]
        si, sj, sk = _rand_vec(rng, n), _rand_vec(rng, n), _rand_vec(rng, n)
        w2 = orc.multiply(A, A, 2.5, si, '.', sj, '.', sk, rowwise=True, nthreads=8)
        vs = [capi.host_vec(v.idx, v.val, v.shape0) for v in (si, sj, sk)]
        for X, Y in ((op.coo, s), (s, op.coo)):
            r = ctx.multiply(X, Y, 2.5, vs[0][0], '.', vs[1][0], '.', vs[2][0], sink=capi.SINK_COO, flags=capi.SINK_ORDERED)
            _check(ctx.fetch(r), w2, exact=True)
        # used with the OTHER transpose flag than it was prepared for: read as a device operand sorted the other way
        w3 = orc.multiply(A, A, 1.0, None, 'T', None, '.', None, rowwise=True, nthreads=8)
        r3 = ctx.multiply(op.coo, op.coo, tA='T', sink=capi.SINK_COO, flags=capi.SINK_ORDERED)
        _check(ctx.fetch(r3), w3, exact=True)
        # the stand-alone algorithms take it too
        assert np.array_equal(ctx.dim_beginnings(op.coo, 0), orc.dim_beginnings(orc.consolidate(a[0], a[1], a[2], 0)[0]))
    finally:
        op.close()
    # rectangular, prepared for 'T' on the right: the Galerkin product's R^T (cfg5) -- no sort of R per product
    g = 16
    R3, A3 = wl.aggregation3d(g), wl.laplace3d(g)
    sR, k1 = capi.host_coo(*R3, sort0=0); sA, k2 = capi.host_coo(*A3, sort0=0)
    wt = orc.multiply(orc.Mat(*R3), orc.Mat(*A3), rowwise=True)
    wc = orc.multiply(orc.Mat(wt[0], wt[1], wt[2], wt[3]), orc.Mat(*R3), tB='T', rowwise=True)
    Rt = capi.Operand(ctx, sR, 'T', capi.AS_B)
    try:
        rt = ctx.multiply(sR, sA, sink=capi.SINK_COO)
        rc = ctx.multiply(capi.result_operand(rt), Rt.coo, tB='T', sink=capi.SINK_COO)
        _check(ctx.fetch(rc), wc, exact=True)
    finally:
        Rt.close()
    # under zero_nan an operand is prepared for ONE side (the two sides drop different NaNs)
    with pytest.raises(capi.SpsamdError):
        capi.Operand(ctx, s, '.', capi.AS_A | capi.AS_B, zero_nan=True)


def test_two_contexts_on_two_threads(ctx):
    """SURVEY 8b 'Threading': the library must be callable concurrently on different handles.
    Two host threads, one context (HIP stream, arena) each, multiply different operands at once."""
    import threading
    from spsparse_amd import capi
    cases = []
    for seed in (3, 4):
        a = wl.rmat(13, seed=seed)
        A = orc.Mat(*a)
        cases.append((A, orc.multiply(A, A, rowwise=True, nthreads=4)))
    errors = []

    def work(A, want):
        try:
            c = capi.Context(0)
            for _ in range(5):
                _check(_dev(c, A, A), want)
                _, _, _, d = _dev(c, A, A, sink=capi.SINK_DIGEST)
                cnt, _, h = orc.digest(*want[:3])
                assert d.nnz == cnt and d.hash == h
            c.close()
        except Exception as e:          # noqa: BLE001 - reported below
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=c) for c in cases]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_heavy_rows_beyond_2_25_columns(ctx):
    """the windowed heavy-row path indexes at most 2048 column windows of 16384 (ncol <= 2^25).  A wider
    op(B) with a heavy row used to be refused (SPSAMD_EINVAL); it is now multiplied by column blocks of B (2^25 columns
    at a time through the ordinary path, the blocks' outputs interleaved row by row) -- the row-wise oracle's tuples in
    the oracle's order, in the COO sink, the digest sink, with scale vectors and a 'T' flag.  Products whose rows all
    stay at or below 4096 scalar products never needed windows and take the ordinary path at any width."""
    from spsparse_amd import capi
    ncol = 3 * (1 << 25) - 12345                     # three column blocks, the last one partial
    rng = np.random.default_rng(5)
    k = 6000
    # B: k rows, a few tuples each, columns spread over the whole (huge) range
    per = 3
    brow = np.repeat(np.arange(k, dtype=np.int32), per)
    bcol = rng.integers(0, ncol, size=k * per).astype(np.int32)
    bcol[:8] = [0, (1 << 25) - 1, 1 << 25, (1 << 25) + 1, (1 << 26) - 1, 1 << 26, ncol - 1, 5]      # block edges
    B = orc.Mat(brow, bcol, rng.standard_normal(k * per), (k, ncol))
    # A: row 0 a mid row (3000 products at most), rows 1..3 heavy (all of B's rows, half of them, ...), row 5 light
    ai = np.concatenate([np.zeros(1000, np.int32), np.ones(k, np.int32), np.full(k // 2, 2, np.int32), np.full(2500, 3, np.int32), np.full(3, 5, np.int32)])
    ak = np.concatenate([np.arange(1000), np.arange(k), np.arange(0, k, 2), rng.choice(k, 2500, replace=False), [1, 2, 3]]).astype(np.int32)
    A = orc.Mat(ai, ak, rng.standard_normal(ai.size), (7, k))
    light = orc.Mat(ai[:1000], ak[:1000], A.val[:1000], (7, k))
    got = _dev(ctx, light, B)
    _check(got, orc.multiply(light, B, rowwise=True))
    assert got[3].rows_heavy == 0 and got[3].rows_mid == 1
    want = orc.multiply(A, B, rowwise=True)
    got = _dev(ctx, A, B)
    _check(got, want)
    assert got[3].rows_heavy >= 1 and got[3].window == 16384       # (per column block: a third of a row's products)
    _, _, _, d = _dev(ctx, A, B, sink=capi.SINK_DIGEST)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s_) <= 1e-11 * np.abs(want[2]).sum()
    # scale vectors (scalek over the wide dimension), C != 1, EXACT_PATTERN, and B given as its transpose
    kcols = np.unique(bcol)
    kcols = kcols[rng.uniform(size=kcols.size) < 0.7]              # (columns absent from scalek are skipped)
    kw = dict(C_=-1.5, scalei=_rand_vec(rng, 7, 0.9), scalej=_rand_vec(rng, k, 0.9), scalek=orc.Vec(kcols, rng.uniform(0.5, 2, kcols.size), ncol))
    want = orc.multiply(A, B, rowwise=True, **kw)
    _check(_dev(ctx, A, B, flags=capi.SINK_EXACT_PATTERN, **kw), want)
    Bt = orc.Mat(B.idx1, B.idx0, B.val, (ncol, k))
    _check(_dev(ctx, A, Bt, tB="T", **kw), want)
    # the same heavy rows at 2^25 columns exactly are inside the windowed path's range
    ncol2 = 1 << 25
    B2 = orc.Mat(brow, (bcol.astype(np.int64) % ncol2).astype(np.int32), B.val, (k, ncol2))
    got2 = _dev(ctx, A, B2)
    _check(got2, orc.multiply(A, B2, rowwise=True))
    assert got2[3].rows_heavy >= 2 and got2[3].window == 16384


@pytest.mark.parametrize("budget_mb", [1, 8])
def test_window_indices_over_budget_go_by_column_blocks(ctx, budget_mb):
    """ADVICE r1 #2, second half: the heavy rows' window indices take 12 bytes per row of op(B) and column window.  Where
    they exceed what the device has left the product goes by column blocks narrow enough for them to fit; the knob
    `index_budget_mb` sets the cap so that the path runs on a small matrix: same tuples, same digest as the oracle."""
    from spsparse_amd import capi
    a = wl.rmat(15, seed=9)                         # 32768 columns = 4 windows; 12 B x 32768 rows per window = 0.4 MB
    A = orc.Mat(*a)
    b = wl.rmat(15, seed=10)
    B = orc.Mat(b[0], b[1], -b[2], b[3])
    want = orc.multiply(A, B, rowwise=True, nthreads=8)
    plain = _dev(ctx, A, B)
    ctx.set_tuning("index_budget_mb", budget_mb)
    try:
        got = _dev(ctx, A, B)
        _, _, _, d = _dev(ctx, A, B, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    finally:
        ctx.set_tuning("index_budget_mb", 0)
    _check(got, want)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s_) <= 1e-11 * np.abs(want[2]).sum()
    if budget_mb == 1:
        assert got[3].products == plain[3].products and got[3].rows_heavy > 0     # (two windows per block: every product once)


def test_ablation_switch_is_not_in_the_shipped_library(monkeypatch):
    """SPSAMD_DBG (ablations that skip work and give wrong results on purpose) exists only in
    -DSPSAMD_ABLATIONS profiling builds.  With it set in the environment a fresh context still
    returns the oracle's product."""
    from spsparse_amd import capi
    monkeypatch.setenv("SPSAMD_DBG", "31")
    c = capi.Context()
    try:
        a = wl.rmat(12, seed=5)
        A = orc.Mat(*a)
        want = orc.multiply(A, A, rowwise=True, nthreads=8)
        got = _dev(c, A, A)
        _check(got, want)
        assert got[3].rows_heavy > 0 and got[3].cells_dense > 0        # the kernels that carried the switches ran
    finally:
        c.close()


@pytest.mark.parametrize("path", [0, 1, 2])
def test_coo_emission_paths_agree(ctx, path):
    """The three ways a hash cell is emitted in column order -- bitmap rank (narrow cells), LDS radix
    sort, bitonic network (keys wider than 32 bits) -- forced in turn through the tuning knob:
    the same tuples in the same order, against the oracle."""
    a = wl.rmat(15, seed=2)
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    ctx.set_tuning("emit_path", path)
    try:
        got = _dev(ctx, A, A)
    finally:
        ctx.set_tuning("emit_path", 0)
    _check(got, want)
    assert got[3].cells_hash > 0 and got[3].rows_mid > 0


@pytest.mark.parametrize("xcd", [0, 1, 2])
def test_dense_cell_walks_agree(ctx, xcd):
    """The three ways the dense cells' list is dealt to the workgroups -- one list with a grid stride, eight static XCD
    parts, eight parts claimed from counters (the default; cells are claimed three ahead and stolen from other parts at
    the end) -- give the oracle's tuples in the oracle's order, in both passes of the COO sink and in the digest."""
    from spsparse_amd import capi
    a = wl.rmat(16, seed=3)
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    ctx.set_tuning("xcd", xcd)
    try:
        got = _dev(ctx, A, A)
        _, _, _, d = _dev(ctx, A, A, sink=capi.SINK_DIGEST)
    finally:
        ctx.set_tuning("xcd", 2)
    _check(got, want)
    assert got[3].cells_dense >= 4096                         # (below that the list is not cut into parts)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s_) <= 1e-11 * abs(s_)


@pytest.mark.parametrize("knobs", [
    {"tiles_v1": 1}, {"tiles_v1": 2}, {"tiles_v1": 3}, {"no_tiles": 1},
    {"direct_min": 1024}, {"direct_min": 1024, "tiles_v1": 2},
    {"dense_min": 1024, "long_dense_min": 64}, {"dense_min": 4096, "long_dense_min": 4096},
], ids=lambda k: ",".join("%s=%d" % kv for kv in k.items()))
def test_every_cell_scheme_against_the_oracle(ctx, knobs):
    """The kernels a default call no longer picks on this matrix stay covered: each tile scheme (first-generation hash
    tiles, hash tiles v2, bitmap-rank tiles, no tiles at all), the direct cells (off by default since round 2) and the
    extremes of the dense thresholds, forced through the tuning knobs -- the same tuples in the same order as the
    row-wise oracle, and the same digest."""
    from spsparse_amd import capi
    a = wl.rmat(15, seed=4)
    A = orc.Mat(*a)
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    for k, v in knobs.items():
        ctx.set_tuning(k, v)
    try:
        got = _dev(ctx, A, A)
        _, _, _, d = _dev(ctx, A, A, sink=capi.SINK_DIGEST)
    finally:
        for k in knobs:
            ctx.set_tuning(k, 0)
    _check(got, want)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s_) <= 1e-11 * abs(s_)
    res = got[3]
    assert res.rows_heavy > 0
    if "direct_min" in knobs:
        assert res.products_direct > 0
    else:
        assert res.products_direct == 0                     # off by default
    if knobs.get("no_tiles"):
        assert res.products_tiles == 0
    elif "dense_min" not in knobs:
        assert res.products_tiles > 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPSAMD_STRESS_SEEDS", "30"))))
def test_random_knobs_signs_and_flags(ctx, seed):
    """Differential stress of the heavy-row kernels: R-MAT operands of random scale (A*A and A*B), random signs, a random
    subset of the tuning knobs, the default and the EXACT_PATTERN flag, 'T' flags and C != 1 -- every draw against the
    row-wise oracle (index set exact, values to 1e-12 of the sum of |terms|)."""
    from spsparse_amd import capi
    rng = np.random.default_rng(1000 + seed)
    scale = int(rng.choice([12, 13, 14, 15]))
    a = wl.rmat(scale, seed=20 + seed)
    va = a[2] * rng.choice([-1.0, 1.0], size=a[2].size) if seed % 2 else a[2]
    A = orc.Mat(a[0], a[1], va, a[3])
    if seed % 3 == 0:
        B = A
    else:
        b = wl.rmat(scale, seed=40 + seed)
        B = orc.Mat(b[0], b[1], b[2] * rng.choice([-1.0, 1.0], size=b[2].size), b[3])
    pool = {"tiles_v1": [1, 2, 3], "dense_min": [256, 1024, 2048, 4096], "long_dense_min": [64, 512, 4096], "cell_cap": [512, 1024, 4096],
            "direct_min": [512, 1536], "no_wmajor": [1], "window": [16384], "no_tiles": [1], "long_cap": [4096], "emit_path": [1, 2]}
    knobs = {k: int(rng.choice(v)) for k, v in pool.items() if rng.uniform() < 0.3}
    kw = dict(tA=str(rng.choice([".", "T"])), tB=str(rng.choice([".", "T"])), C_=float(rng.choice([1.0, -0.5])))
    flags = capi.SINK_EXACT_PATTERN if seed % 2 else 0
    want = orc.multiply(A, B, rowwise=True, nthreads=8, **kw)
    absA, absB = orc.Mat(A.idx0, A.idx1, np.abs(A.val), A.shape), orc.Mat(B.idx0, B.idx1, np.abs(B.val), B.shape)
    bound = orc.multiply(absA, absB, rowwise=True, nthreads=8, tA=kw["tA"], tB=kw["tB"], C_=abs(kw["C_"]))
    for k, v in knobs.items():
        ctx.set_tuning(k, v)
    try:
        got = _dev(ctx, A, B, flags=flags, **kw)
        _, _, _, d = _dev(ctx, A, B, flags=flags, sink=capi.SINK_DIGEST, **kw)
    finally:
        for k in knobs:
            ctx.set_tuning(k, 0)
    ncol = int(want[3][1]) if len(want) > 3 else int(max(A.shape[1], B.shape[1], A.shape[0], B.shape[0]))
    key = lambda r: r[0].astype(np.int64) * (1 << 32) + r[1]
    if flags:
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), knobs
        pos = np.searchsorted(key(bound), key(want))
        assert np.all(np.abs(got[2] - want[2]) <= 1e-12 * bound[2][pos]), knobs
        assert d.nnz == len(want[0])
    else:
        # default mode: a sum that cancels to rounding may differ in its zero test; with random real values none does
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), knobs
        pos = np.searchsorted(key(bound), key(want))
        assert np.all(np.abs(got[2] - want[2]) <= 1e-12 * bound[2][pos]), knobs
        cnt, s_, h = orc.digest(*want[:3])
        assert d.nnz == cnt and d.hash == h, knobs
    assert got[3].rows_heavy > 0


@pytest.mark.parametrize("tA,tB", [(".", "."), ("T", "."), (".", "T"), ("T", "T")])
def test_heavy_rows_with_scales_and_flags(ctx, tA, tB):
    """Scale vectors (absent indices, zero scales), C != 1 and 'T' flags on an R-MAT product whose rows
    reach the LDS-hash, tile and dense-window kernels (the small random cases only reach the light ones)."""
    rng = np.random.default_rng(5)
    a = wl.rmat(14, seed=2)          # 16384 columns = two 8192-column windows
    b = wl.rmat(14, seed=3)
    n = a[3][0]
    A, B = orc.Mat(*a), orc.Mat(*b)
    kw = dict(C_=-2.5, tA=tA, tB=tB, scalei=_rand_vec(rng, n, 0.9), scalej=_rand_vec(rng, n, 0.8), scalek=_rand_vec(rng, n, 0.9))
    want = orc.multiply(A, B, rowwise=True, nthreads=8, **kw)
    got = _dev(ctx, A, B, **kw)
    _check(got, want)
    assert got[3].rows_heavy > 0 and got[3].cells_hash > 0 and got[3].cells_dense > 0
    from spsparse_amd import capi
    _, _, _, d = _dev(ctx, A, B, sink=capi.SINK_DIGEST, **kw)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s_) <= 1e-11 * abs(s_)


def test_wide_matrix_16k_windows(ctx):
    """ncol > 2^21 switches the heavy rows to 16384-column windows (k_dense<16384,1024>): dense cells
    in a crowded column range, hash cells over the sparse remainder, against the row-wise oracle."""
    rng = np.random.default_rng(8)
    m, k, ncol = 6, 96, 1 << 22
    ai0 = np.repeat(np.arange(m), k)
    ai1 = np.tile(np.arange(k), m)
    keep = rng.uniform(size=ai0.size) < 0.8
    A = orc.Mat(ai0[keep], ai1[keep], rng.uniform(0.1, 1, int(keep.sum())), (m, k))
    rows, cols = [], []
    for r in range(k):
        crowded = rng.integers(0, 90000, 1500)
        spread = rng.integers(0, ncol, 400)
        c = np.unique(np.concatenate([crowded, spread]))
        rows.append(np.full(c.size, r))
        cols.append(c)
    bi0, bi1 = np.concatenate(rows), np.concatenate(cols)
    B = orc.Mat(bi0, bi1, rng.uniform(0.1, 1, bi0.size), (k, ncol))
    want = orc.multiply(A, B, rowwise=True, nthreads=4)
    got = _dev(ctx, A, B)
    _check(got, want)
    assert got[3].cells_dense > 0 and got[3].cells_hash > 0
    assert got[3].shape1 == ncol


@pytest.mark.parametrize("scale", [11, 14])
def test_ordered_mode_is_bit_exact(ctx, scale):
    """SPSAMD_SINK_ORDERED: every sum in ascending k like multiply_sparse.hpp:219-236, so hash,
    tile and dense-window rows are bit-identical to the oracle too -- on mixed-sign values, where
    the arrival-order atomics of the default mode would only be within tolerance."""
    from spsparse_amd import capi
    rng = np.random.default_rng(scale)
    a = wl.rmat(scale, seed=4)
    vals = rng.uniform(-1, 1, a[2].size)
    A = orc.Mat(a[0], a[1], vals, a[3])
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    got = _dev(ctx, A, A, flags=capi.SINK_ORDERED)
    _check(got, want, exact=True)
    assert got[3].rows_heavy > 0 and got[3].rows_mid > 0
    _, _, _, d = _dev(ctx, A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ORDERED)
    cnt, s_, h = orc.digest(*want[:3])
    assert d.nnz == cnt and d.hash == h
    # the default mode on the same input: same index set here, values within tolerance of the scale
    got2 = _dev(ctx, A, A)
    assert np.array_equal(got2[0], want[0]) and np.array_equal(got2[1], want[1])


def test_ordered_mode_exact_cancellation(ctx):
    """Sums that cancel to exactly 0 only in the reference's summation order are dropped in ordered
    mode exactly as the reference drops them (multiply_sparse.hpp:238)."""
    from spsparse_amd import capi
    # row 0: a_k * b_kj with terms 1e16, 1, -1e16 in ascending k: ((1e16 + 1) - 1e16) == 0 in this order
    k, ncol = 3, 3000
    A = orc.Mat([0] * k, list(range(k)), [1.0, 1.0, 1.0], (1, k))
    bi0 = np.repeat(np.arange(k), ncol)
    bi1 = np.tile(np.arange(ncol), k)
    bv = np.repeat(np.array([1e16, 1.0, -1e16]), ncol)
    B = orc.Mat(bi0, bi1, bv, (k, ncol))
    want = orc.multiply(A, B, rowwise=True)
    assert len(want[2]) == 0                                # the reference order cancels every column
    got = _dev(ctx, A, B, flags=capi.SINK_ORDERED)
    assert got[3].nnz == 0 and got[3].products == k * ncol


@pytest.mark.parametrize("spread", [1, 64], ids=["dense_cells", "hash_cells"])
def test_exact_pattern_mode_cancellation(ctx, spread):
    """SPSAMD_SINK_EXACT_PATTERN: the reference's index set at arrival-order speed.  Terms that cancel to exactly 0
    only in ascending k (1e16 + 1 - 1e16) must be dropped like the reference drops them (multiply_sparse.hpp:238),
    terms that cancel only in SOME other order (1e16 - 1e16 + 1 = 1) must be kept with the reference's value -- in the
    dense-window cells (spread 1: 9000 products in one window) and in the hash cells (spread 64: 24 windows)."""
    from spsparse_amd import capi
    k, ncol = 3, 3000
    cols = np.arange(ncol) * spread
    A = orc.Mat([0] * k, list(range(k)), [1.0, 1.0, 1.0], (1, k))
    bi0 = np.repeat(np.arange(k), ncol)
    bi1 = np.tile(cols, k)
    for vals, expect in (([1e16, 1.0, -1e16], None), ([1e16, -1e16, 1.0], 1.0)):
        B = orc.Mat(bi0, bi1, np.repeat(np.array(vals), ncol), (k, ncol * spread))
        want = orc.multiply(A, B, rowwise=True)
        assert len(want[2]) == (0 if expect is None else ncol)
        got = _dev(ctx, A, B, flags=capi.SINK_EXACT_PATTERN)
        assert got[3].rows_heavy == 1 and got[3].products == k * ncol
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        d = _dev(ctx, A, B, sink=capi.SINK_DIGEST, flags=capi.SINK_EXACT_PATTERN)[3]
        cnt, ssum, h = orc.digest(*want[:3])
        assert (d.nnz, d.hash) == (cnt, h) and d.sum == ssum


def test_exact_pattern_mode_mixed_signs(ctx):
    """Mixed-sign values on a matrix with light, hash-cell and dense-cell rows: with EXACT_PATTERN the index set is the
    oracle's (ascending-k sums); values agree to the rounding of an arrival-order sum (relative to the sum of |terms|),
    and where the flag re-evaluated a sum it is the oracle's value bit for bit.  Integer-valued operands make exact
    cancellation common."""
    from spsparse_amd import capi
    rng = np.random.default_rng(21)
    a = wl.rmat(15, seed=6)
    vals = rng.integers(-3, 4, size=a[2].size).astype(np.float64)
    vals[vals == 0] = 1.0
    A = orc.Mat(a[0], a[1], vals, a[3])
    want = orc.multiply(A, A, rowwise=True, nthreads=8)
    got = _dev(ctx, A, A, flags=capi.SINK_EXACT_PATTERN)
    assert got[3].cells_dense > 0 and got[3].cells_hash > 0 and got[3].rows_mid > 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2])                  # small integers: every order is exact
    plain = _dev(ctx, A, A)                                  # the default mode has the same pattern here: sums of integers are exact
    assert np.array_equal(plain[0], want[0])
    # real-valued mixed signs: pattern identical, values to rounding
    vals2 = rng.standard_normal(a[2].size)
    A2 = orc.Mat(a[0], a[1], vals2, a[3])
    want2 = orc.multiply(A2, A2, rowwise=True, nthreads=8)
    got2 = _dev(ctx, A2, A2, flags=capi.SINK_EXACT_PATTERN)
    assert np.array_equal(got2[0], want2[0]) and np.array_equal(got2[1], want2[1])
    absA = orc.Mat(a[0], a[1], np.abs(vals2), a[3])
    bound = orc.multiply(absA, absA, rowwise=True, nthreads=8)     # sum of |terms| per output, same pattern superset
    key = lambda r: r[0].astype(np.int64) * a[3][1] + r[1]
    pos = np.searchsorted(key(bound), key(want2))
    assert np.all(np.abs(got2[2] - want2[2]) <= 1e-12 * bound[2][pos])
    # ... at a sane price: a dense cell with both signs once re-evaluated every EMPTY slot of its window from the
    # operands (170x the default time on this size); only touched slots whose sum the bound cannot decide are candidates
    plain2 = _dev(ctx, A2, A2)
    got2 = _dev(ctx, A2, A2, flags=capi.SINK_EXACT_PATTERN)   # (second call: no workspace growth in the time)
    assert got2[3].ms_total <= 5.0 * plain2[3].ms_total + 2.0, (got2[3].ms_total, plain2[3].ms_total)


def test_poisson_exact(ctx):
    """cfg3 at N=64: values are small integers, so every summation order is exact;
    closed forms nnz(A)=5N^2-4N, P=25N^2-36N+8, nnz(C)=13N^2-20N+4."""
    N = 64
    a = wl.poisson2d(N)
    A = orc.Mat(*a, sort0=0)
    want = orc.multiply(A, A, rowwise=True)
    got = _dev(ctx, A, A)
    _check(got, want, exact=True)
    assert A.nnz == 5 * N * N - 4 * N
    assert got[3].products == 25 * N * N - 36 * N + 8
    assert got[3].nnz == 13 * N * N - 20 * N + 4
    assert set(np.unique(got[2]).tolist()) <= {-8., 1., 2., 18., 19., 20.}


@pytest.mark.parametrize("two_pass", [0, 1], ids=["one_pass_and_gather", "count_scan_store"])
def test_all_light_coo_sink_both_ways(ctx, two_pass):
    """The COO sink of a product in which every row is light: one compute pass into per-row slots plus a gather (default), or
    count / scan / store (the `light_two_pass` knob; also what a product too large for the sparse buffer takes).  Random
    values with cancellations and explicit zeros, scale vectors, 'T', empty rows, rows of every slot class (S = 8 .. 64)."""
    rng = np.random.default_rng(41)
    ctx.set_tuning("light_two_pass", two_pass)
    try:
        for trial in range(24):
            n = int(rng.integers(3, 400))
            per = [1, 2, 3, 5, 7][trial % 5]                       # tuples per row of either operand: products per row up to per^2 <= 49
            def band(m, k, per_):
                rows = np.repeat(np.arange(m), per_)
                cols = rng.integers(0, k, rows.size)
                vals = rng.integers(-2, 3, rows.size).astype(np.float64)      # small integers: exact sums, real cancellations, zeros
                keep = rng.uniform(size=rows.size) < 0.85                     # some rows shorter, some empty
                return orc.Mat(rows[keep], cols[keep], vals[keep], (m, k))
            tA, tB = ".T"[trial % 2], ".T"[(trial // 2) % 2]
            m, k, nn = n, int(rng.integers(3, 400)), int(rng.integers(3, 400))
            A = band(k, m, per) if tA == "T" else band(m, k, per)
            B = band(nn, k, per) if tB == "T" else band(k, nn, per)
            if tA == "T" or tB == "T":
                # a transposed band has no bound on its row lengths: only keep the case if it is still all-light, else it simply takes the binned path
                pass
            si = _rand_vec(rng, m) if trial % 3 == 0 else None
            sk = _rand_vec(rng, nn) if trial % 4 == 1 else None
            want = orc.multiply(A, B, 2.0, si, tA, None, tB, sk, rowwise=True)
            got = _dev(ctx, A, B, C_=2.0, scalei=si, tA=tA, tB=tB, scalek=sk)
            _check(got, want, exact=True)
        # cfg3 / cfg5 shapes
        N = 48
        a = wl.poisson2d(N)
        A = orc.Mat(*a, sort0=0)
        _check(_dev(ctx, A, A), orc.multiply(A, A, rowwise=True), exact=True)
        R3, A3 = orc.Mat(*wl.aggregation3d(12), sort0=0), orc.Mat(*wl.laplace3d(12), sort0=0)
        wt = orc.multiply(R3, A3, rowwise=True)
        gt = _dev(ctx, R3, A3)
        _check(gt, wt, exact=True)
        assert gt[3].rows_light == R3.shape[0] and gt[3].rows_heavy == 0     # the all-light direct kernel took it
    finally:
        ctx.set_tuning("light_two_pass", 0)


def test_galerkin_exact(ctx):
    """cfg5 at N=16: T = R*A, C = T*R^T ('T' flag, multiply_sparse.hpp:168); closed forms
    nnz(T)=32nc^3-24nc^2, nnz(C)=7nc^3-6nc^2, C values in {24,-4}."""
    N, nc = 16, 8
    A = orc.Mat(*wl.laplace3d(N), sort0=0)
    R = orc.Mat(*wl.aggregation3d(N), sort0=0)
    wantT = orc.multiply(R, A, rowwise=True)
    gotT = _dev(ctx, R, A)
    _check(gotT, wantT, exact=True)
    assert gotT[3].nnz == 32 * nc ** 3 - 24 * nc ** 2
    T = orc.Mat(gotT[0], gotT[1], gotT[2], (nc ** 3, N ** 3))
    wantC = orc.multiply(T, R, tB="T", rowwise=True)
    gotC = _dev(ctx, T, R, tB="T")
    _check(gotC, wantC, exact=True)
    assert gotC[3].nnz == 7 * nc ** 3 - 6 * nc ** 2
    assert set(np.unique(gotC[2]).tolist()) == {24., -4.}


def test_cancellation_closes_holes(ctx):
    """Sums that are exactly 0 are dropped (multiply_sparse.hpp:238) in every row class, including
    where the reserve pass is structural (LDS hash / dense windows) and leaves holes to close."""
    for ncol, reps in ((300, 1), (3000, 1), (9000, 2)):
        # row 0 of A = [+1, -1, +1 ...] against B rows that agree on even columns and differ on odd ones
        k = 2 * reps
        ai0 = np.zeros(k, np.int32)
        ai1 = np.arange(k, dtype=np.int32)
        av = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
        cols = np.arange(ncol, dtype=np.int32)
        bi0 = np.repeat(np.arange(k, dtype=np.int32), ncol)
        bi1 = np.tile(cols, k)
        bv = np.ones(k * ncol)
        odd = (bi1 % 2 == 1) & (bi0 % 2 == 1)
        bv[odd] = 3.0
        A = orc.Mat(ai0, ai1, av, (1, k))
        B = orc.Mat(bi0, bi1, bv, (k, ncol))
        want = orc.multiply(A, B, rowwise=True)
        got = _dev(ctx, A, B)
        _check(got, want, exact=True)
        assert len(want[2]) == ncol // 2 and np.all(got[1] % 2 == 1)


def test_short_circuits_and_errors(ctx):
    from spsparse_amd import capi
    A = orc.Mat([0], [1], [2.], (2, 3))
    B = orc.Mat([1], [0], [4.], (3, 2))
    E = orc.Mat([], [], [], (3, 2))
    i, j, v, res = _dev(ctx, A, B)
    assert v.tolist() == [8.] and (res.shape0, res.shape1) == (2, 2)
    assert _dev(ctx, A, B, C_=0.0)[3].nnz == 0
    r = _dev(ctx, A, E)[3]
    assert r.nnz == 0 and (r.shape0, r.shape1) == (2, 2)          # shape set before the short-circuit (:169)
    assert _dev(ctx, A, B, scalej=orc.Vec([], [], 3))[3].nnz == 0
    Z = orc.Mat([0], [1], [0.], (2, 3))                             # only explicit zeros (Appendix A.3)
    assert _dev(ctx, Z, B)[3].nnz == 0
    with pytest.raises(capi.SpsamdError, match=r"Inner dimensions for A \(3\) and B \(2\) must match!") as e:
        _dev(ctx, A, orc.Mat([0], [0], [1.], (2, 2)))
    assert e.value.code == -1
    with pytest.raises(capi.SpsamdError, match="not strictly ascending"):
        _dev(ctx, A, B, scalej=orc.Vec([2, 1], [1., 1.], 3))
    with pytest.raises(capi.SpsamdError, match="out of bounds"):
        _dev(ctx, orc.Mat([5], [1], [2.], (2, 3)), B)
    # the context stays usable after an error
    assert _dev(ctx, A, B)[2].tolist() == [8.]


def test_presorted_operands(ctx):
    """algorithm.hpp:360: a matching sort_order skips the consolidation."""
    rng = np.random.default_rng(11)
    A, B = _rand_mat(rng, (30, 30), 200), _rand_mat(rng, (30, 30), 200)
    a0, a1, av = orc.consolidate(A.idx0, A.idx1, A.val, 0)
    b0, b1, bv = orc.consolidate(B.idx0, B.idx1, B.val, 0)
    As, Bs = orc.Mat(a0, a1, av, (30, 30), sort0=0), orc.Mat(b0, b1, bv, (30, 30), sort0=0)
    want = orc.multiply(A, B)
    _check(_dev(ctx, As, Bs), want)
    _check(_dev(ctx, A, B), want)
    # a sort_order that is CLAIMED but not there (set_sorted() on unsorted tuples): the reference would quietly mis-compute;
    # here the dense row pointer is built from the tuples' side and would be left with entries nobody wrote -- rejected
    # (ADVICE r2).  Rows ascending but columns not, or duplicates, stay "trusted" like Consolidate<> trusts them.
    from spsparse_amd import capi
    lie = orc.Mat(a0[::-1].copy(), a1[::-1].copy(), av[::-1].copy(), (30, 30), sort0=0)
    with pytest.raises(capi.SpsamdError) as e:
        _dev(ctx, lie, Bs)
    assert e.value.code == -2 and "sort_order" in e.value.msg
    with pytest.raises(capi.SpsamdError):
        _dev(ctx, As, orc.Mat(b0[::-1].copy(), b1[::-1].copy(), bv[::-1].copy(), (30, 30), sort0=0))


@pytest.mark.parametrize("dups,zeros", [(True, True), (False, True), (False, False)])
@pytest.mark.parametrize("policy", [0, 1, 2])
def test_row_sorted_operand_used_transposed(ctx, policy, dups, zeros):
    """An operand stored in (row, column) order, not declared sorted, and used with 'T'.  Strictly ascending: its
    consolidation by columns sorts on the major digits only (the LSD passes over the low digits are skipped: ties stay in
    input order, which is the minor order already) and, with no value to drop, the sorted tuples are the result; with
    explicit zeros the flag / compact / merge passes follow; with duplicate tuples the full sort.  Both as A and as B,
    every duplicate policy, against the oracle; and the stand-alone consolidate by {1, 0} of the same tuples."""
    rng = np.random.default_rng(31 + policy)
    n, m = 3000, 70000
    i0 = np.sort(rng.integers(0, 300, n)).astype(np.int32)                  # rows ascending, many per row
    i1 = rng.integers(0, m, n).astype(np.int32)
    order = np.lexsort((i1, i0))
    i0, i1 = i0[order], i1[order]
    if dups:
        dup = rng.integers(0, n - 1, 400)
        i0[dup + 1], i1[dup + 1] = i0[dup], i1[dup]                          # duplicate tuples, adjacent (still in order)
        order = np.lexsort((i1, i0), )
        i0, i1 = i0[order], i1[order]
    else:                                                                   # strictly ascending: sorted tuples = consolidated operand
        keep = np.concatenate([[True], (np.diff(i0) != 0) | (np.diff(i1) != 0)])
        i0, i1 = i0[keep], i1[keep]
        n = i0.size
    v = rng.standard_normal(n)
    if zeros:
        v[rng.integers(0, n, 60)] = 0.0
    X = orc.Mat(i0, i1, v, (300, m))                                       # sort0 = -1: nothing declared
    Y = _rand_mat(rng, (300, 40), 2000)
    kw = dict(duplicate_policy=policy)
    _check(_dev(ctx, X, Y, tA="T", **kw), orc.multiply(X, Y, rowwise=True, tA="T", **kw))        # (m x 300) * (300 x 40)
    Z = _rand_mat(rng, (25, m), 4000)
    _check(_dev(ctx, Z, X, tB="T", **kw), orc.multiply(Z, X, rowwise=True, tB="T", **kw))        # (25 x m) * (m x 300)
    from spsparse_amd import capi
    sx, keep = capi.host_coo(X.idx0, X.idx1, X.val, X.shape)
    r = ctx.consolidate(sx, 1, policy)
    gi, gj, gv = ctx.fetch(r)
    w0, w1, wv = orc.consolidate(X.idx0, X.idx1, X.val, 1, policy)
    assert np.array_equal(gi, w0) and np.array_equal(gj, w1) and np.array_equal(gv, wv)


def test_consolidate_known_answer(ctx):
    """tests/test_array.cpp:135-168 through spsamd_consolidate"""
    from spsparse_amd import capi
    s, keep = capi.host_coo([1, 1, 0, 0, 1], [3, 2, 3, 1, 2], [5., 3., 17., 14., 15.], (2, 4))
    r = ctx.consolidate(s, 0)
    i, j, v = ctx.fetch(r)
    assert i.tolist() == [0, 0, 1, 1] and j.tolist() == [1, 3, 2, 3] and v.tolist() == [14., 17., 18., 5.]
    r = ctx.consolidate(s, 1)
    i, j, v = ctx.fetch(r)
    assert i.tolist() == [0, 1, 0, 1] and j.tolist() == [1, 2, 3, 3] and v.tolist() == [14., 18., 17., 5.]
    # policies and the zero / NaN quirks against the oracle
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = 2000
        i0, i1 = rng.integers(0, 40, n), rng.integers(0, 50, n)
        v = rng.uniform(-1, 1, n)
        v[rng.integers(0, n, 100)] = 0.0
        if trial % 2:
            v[rng.integers(0, n, 50)] = np.nan
        pol = [orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3]
        for so0 in (0, 1):
            w0, w1, wv = orc.consolidate(i0, i1, v, so0, pol, bool(trial % 2))
            s, keep = capi.host_coo(i0, i1, v, (40, 50))
            r = ctx.consolidate(s, so0, pol, bool(trial % 2))
            g0, g1, gv = ctx.fetch(r)
            assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
            assert np.array_equal(gv, wv, equal_nan=True)


def test_permutation_and_dim_beginnings_known_answers(ctx):
    """tests/test_array.cpp:67-79 (sorted_permutation) and :146-166 (dim_beginnings) through the C ABI,
    then both against the oracle on random inputs (stable order of duplicates included)."""
    from spsparse_amd import capi
    s, keep = capi.host_coo([1, 1, 0], [3, 2, 3], [5., 3., 17.], (2, 4))
    assert ctx.sorted_permutation(s, 0).tolist() == [2, 1, 0]
    assert ctx.sorted_permutation(s, 1).tolist() == [1, 2, 0]
    s, keep = capi.host_coo([0, 0, 1, 1], [1, 3, 2, 3], [14., 17., 18., 5.], (2, 4), sort0=0)
    assert ctx.dim_beginnings(s, 0).tolist() == [0, 2, 4]
    s, keep = capi.host_coo([0, 1, 0, 1], [1, 2, 3, 3], [14., 18., 17., 5.], (2, 4), sort0=1)
    assert ctx.dim_beginnings(s, 1).tolist() == [0, 1, 2, 4]
    with pytest.raises(capi.SpsamdError, match="sorted first"):
        ctx.dim_beginnings(capi.host_coo([0], [0], [1.], (1, 1))[0], 0)
    rng = np.random.default_rng(2)
    i0, i1 = rng.integers(0, 300, 20000), rng.integers(0, 40, 20000)
    v = rng.uniform(size=20000)
    for so0 in (0, 1):
        s, keep = capi.host_coo(i0, i1, v, (300, 40))
        assert np.array_equal(ctx.sorted_permutation(s, so0), orc.sorted_permutation(i0, i1, so0))
        c0, c1, cv = orc.consolidate(i0, i1, v, so0)
        s, keep = capi.host_coo(c0, c1, cv, (300, 40), sort0=so0)
        assert np.array_equal(ctx.dim_beginnings(s, so0), orc.dim_beginnings(c0 if so0 == 0 else c1))


def test_dense_accumulator_sink(ctx):
    """DenseAccum analogue (accum.hpp:110-140): the COO result scattered into a device dense matrix,
    ADD twice (the sink is appended to, never cleared) then REPLACE, against the oracle's tuples."""
    import torch
    from spsparse_amd import capi
    rng = np.random.default_rng(4)
    A, B = _rand_mat(rng, (37, 50), 400), _rand_mat(rng, (50, 41), 500)
    wi, wj, wv, _ = orc.multiply(A, B)
    want = np.zeros((37, 41))
    want[wi, wj] = wv
    a, ka = capi.host_coo(A.idx0, A.idx1, A.val, A.shape)
    b, kb = capi.host_coo(B.idx0, B.idx1, B.val, B.shape)
    res = ctx.multiply(a, b, sink=capi.SINK_COO)
    dense = torch.zeros((37, 48), dtype=torch.float64, device="cuda:0")     # ld = 48 > 41 columns
    ctx.scatter_dense(res, dense.data_ptr(), 48, capi.ADD)
    ctx.scatter_dense(res, dense.data_ptr(), 48, capi.ADD)
    got = dense.cpu().numpy()
    assert np.allclose(got[:, :41], 2 * want, rtol=1e-12, atol=0) and not got[:, 41:].any()
    ctx.scatter_dense(res, dense.data_ptr(), 48, capi.REPLACE)
    assert np.allclose(dense.cpu().numpy()[:, :41], want, rtol=1e-12, atol=0)
    # LEAVE_ALONE as the reference spells it (accum.hpp:128-130: `if (!std::isnan(oval)) oval = val`), the same on
    # the device and in the host mirror: an entry is overwritten unless it holds a NaN
    dense.fill_(7.0)
    dense[0, :] = float("nan")
    ctx.scatter_dense(res, dense.data_ptr(), 48, capi.LEAVE_ALONE)
    got = dense.cpu().numpy()[:, :41]
    hit = np.zeros((37, 41), dtype=bool)
    hit[wi, wj] = True
    assert np.all(np.isnan(got[0])) and np.allclose(got[1:][hit[1:]], want[1:][hit[1:]], rtol=1e-12, atol=0) and np.all(got[1:][~hit[1:]] == 7.0)
    with pytest.raises(capi.SpsamdError, match="leading dimension"):
        ctx.scatter_dense(res, dense.data_ptr(), 40)


def test_permute_sink_flag(ctx):
    """PermuteAccum {1,0} analogue (accum.hpp:73-101): SPSAMD_SINK_PERMUTE emits (j, i, v) with the shape
    swapped; scattered into a dense matrix it is the transpose of the plain result.  MV ignores the flag."""
    import torch
    from spsparse_amd import capi
    rng = np.random.default_rng(9)
    A, B = _rand_mat(rng, (37, 50), 400), _rand_mat(rng, (50, 41), 500)
    wi, wj, wv, _ = orc.multiply(A, B)
    i, j, v, res = _dev(ctx, A, B, flags=capi.SINK_PERMUTE)
    assert (res.shape0, res.shape1) == (41, 37)
    assert np.array_equal(i, wj) and np.array_equal(j, wi)
    assert np.allclose(v, wv, rtol=1e-12, atol=0)
    dense = torch.zeros((41, 37), dtype=torch.float64, device="cuda:0")
    ctx.scatter_dense(res, dense.data_ptr(), 37, capi.ADD)
    want = np.zeros((37, 41))
    want[wi, wj] = wv
    assert np.allclose(dense.cpu().numpy(), want.T, rtol=1e-12, atol=0)
    # shape is set (permuted) before the dimension check, as the sink's set_shape would be (:169)
    bad = _rand_mat(rng, (49, 41), 10)
    a, ka = capi.host_coo(A.idx0, A.idx1, A.val, A.shape)
    b, kb = capi.host_coo(bad.idx0, bad.idx1, bad.val, bad.shape)
    with pytest.raises(capi.SpsamdError, match="Inner dimensions"):
        ctx.multiply(a, b, flags=capi.SINK_PERMUTE)


def test_device_generators_match_numpy(ctx):
    """csrc/workload.hip == spsparse_amd/workloads.py, tuple for tuple."""
    import torch
    dev = torch.device("cuda:0")

    def bufs(n):
        return (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                torch.empty(n, dtype=torch.float64, device=dev))

    def same(t, ref):
        torch.cuda.synchronize()        # device-wide: also drains the library's own stream
        return all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(t, ref[:3]))

    t = bufs(16 << 10)
    ctx.gen_rmat(10, 7, 0, 16 << 10, *[x.data_ptr() for x in t])
    assert same(t, wl.rmat(10, 7))
    t = bufs(1000)
    ctx.gen_rmat(10, 7, 5000, 1000, *[x.data_ptr() for x in t])
    assert same(t, wl.rmat(10, 7, first_edge=5000, n_edges=1000))
    t = bufs(5000)
    ctx.gen_random_rows(500, 10, 3, 8, *[x.data_ptr() for x in t])
    assert same(t, wl.random_rows(500, 10, 3, 8))
    for N, gen, ref in ((20, ctx.gen_poisson2d, wl.poisson2d), (8, ctx.gen_laplace3d, wl.laplace3d),
                        (8, ctx.gen_aggregation3d, wl.aggregation3d)):
        r = ref(N)
        t = bufs(len(r[2]))
        gen(N, *[x.data_ptr() for x in t])
        assert same(t, r)


def test_device_resident_operands_rmat15(ctx):
    """Device-resident operands (the bench's path): R-MAT scale 15 A*A, digest + row statistics
    against the row-wise oracle (nnz(C) ~ 6e7: compared through count, index hash and row sums)."""
    import torch
    from spsparse_amd import capi
    scale, seed = 15, 1
    n, ne = 1 << scale, 16 << scale
    dev = torch.device("cuda:0")
    t0 = torch.empty(ne, dtype=torch.int32, device=dev)
    t1 = torch.empty(ne, dtype=torch.int32, device=dev)
    tv = torch.empty(ne, dtype=torch.float64, device=dev)
    ctx.gen_rmat(scale, seed, 0, ne, t0.data_ptr(), t1.data_ptr(), tv.data_ptr())
    Ad = capi.device_coo(t0.data_ptr(), t1.data_ptr(), tv.data_ptr(), ne, (n, n))
    d = ctx.multiply(Ad, Ad, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    A = orc.Mat(*wl.rmat(scale, seed))
    wi, wj, wv, _ = orc.multiply(A, A, rowwise=True, nthreads=16)
    cnt, s, h = orc.digest(wi, wj, wv)
    assert d.nnz == cnt and d.hash == h and abs(d.sum - s) <= REL * abs(s)
    rn = ctx.to_host(d.row_nnz, n, np.int64)
    rs = ctx.to_host(d.row_sum, n, np.float64)
    want_n = np.bincount(wi, minlength=n)
    want_s = np.bincount(wi, weights=wv, minlength=n)
    assert np.array_equal(rn, want_n)
    nz = want_s != 0
    assert np.max(np.abs(rs[nz] - want_s[nz]) / np.abs(want_s[nz])) <= 1e-11
    # the COO sink on the same operands: sorted, same count
    r = ctx.multiply(Ad, Ad, sink=capi.SINK_COO)
    assert r.nnz == cnt


# ----------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: too large for a tuple-by-tuple oracle, checked through
# size-independent properties (linearity: C 1 = A (B 1); closed-form sizes of the stencils;
# agreement of the two sinks).

def _device_operand(ctx, gen, n_tuples, shape, sort0=-1):
    import torch
    from spsparse_amd import capi
    dev = torch.device("cuda:0")
    t = (torch.empty(n_tuples, dtype=torch.int32, device=dev), torch.empty(n_tuples, dtype=torch.int32, device=dev),
         torch.empty(n_tuples, dtype=torch.float64, device=dev))
    gen(*[x.data_ptr() for x in t])
    torch.cuda.synchronize()
    return capi.device_coo(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n_tuples, shape, sort0), t


def _row_sums_by_linearity(a, b, n_rows, n_inner):
    """(A B) 1 = A (B 1): per-row sums of C from O(nnz) host arithmetic."""
    b1 = np.bincount(b[0], weights=b[2], minlength=n_inner)
    return np.bincount(a[0], weights=a[2] * b1[a[1]], minlength=n_rows)


def test_cfg2_rmat20_full_size_properties(ctx):
    """BASELINE cfg2: R-MAT scale-20 A*A (P = 2.09e10, nnz(C) = 9.7e9) -- every row sum and the
    grand sum follow from linearity; nnz(C) is bounded by the products and reproducible."""
    from spsparse_amd import capi
    scale, seed = 20, 1
    n, ne = 1 << scale, 16 << scale
    A, keep = _device_operand(ctx, lambda *p: ctx.gen_rmat(scale, seed, 0, ne, *p), ne, (n, n))
    d = ctx.multiply(A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    a = wl.rmat(scale, seed)
    want = _row_sums_by_linearity(a, a, n, n)
    got = ctx.to_host(d.row_sum, n, np.float64)
    nz = want != 0
    assert np.array_equal(got != 0, nz)                  # values are positive: a row is empty iff its sum is 0
    assert np.max(np.abs(got[nz] - want[nz]) / want[nz]) <= 1e-10
    assert abs(d.sum - want.sum()) <= 1e-10 * want.sum()
    rn = ctx.to_host(d.row_nnz, n, np.int64)
    assert int(rn.sum()) == d.nnz and 9.0e9 < d.nnz < 1.1e10 and d.nnz <= d.products
    assert d.products == 20924218068 and d.nnz_a == 16086131
    d2 = ctx.multiply(A, A, sink=capi.SINK_DIGEST)
    assert (d2.nnz, d2.hash) == (d.nnz, d.hash)          # index set independent of the atomics' order
    # ... and pinned to the oracle: the checker's streaming digest of the WHOLE product (2.1e10 scalar products on the
    # host cores) -- tuple count, index hash, and per output row its tuple count and its own index hash.  This is where
    # the kernels only this size selects by default meet the oracle (tests/test_multiply_sparse.cpp:119-128's per-cell
    # check, as far as 9.7e9 cells can be held: per row).
    w = orc.multiply_digest(orc.Mat(*a), orc.Mat(*a), nthreads=orc.host_threads(), rowstats=True)
    assert (w.products, w.nnz_a) == (d.products, d.nnz_a)
    assert (d.nnz, d.hash) == (w.nnz, w.hash)
    assert np.array_equal(rn, w.row_nnz)
    assert np.array_equal(ctx.to_host(d.row_hash, n, np.uint64), w.row_hash)
    assert abs(d.sum - w.sum) <= 1e-10 * abs(w.sum)


@pytest.mark.parametrize("flags", [0, 8], ids=["arrival_order", "exact_pattern"])
def test_rmat18_whole_product_against_the_oracle_digest(ctx, flags):
    """R-MAT scale-18 A*A (2.9e9 products, nnz(C) = 1.28e9) from raw device-resident tuples: digest and per-row
    statistics equal the oracle's streaming digest; the COO sink delivers the same count."""
    from spsparse_amd import capi
    scale, seed = 18, 1
    n, ne = 1 << scale, 16 << scale
    A, keep = _device_operand(ctx, lambda *p: ctx.gen_rmat(scale, seed, 0, ne, *p), ne, (n, n))
    d = ctx.multiply(A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS | flags)
    a = wl.rmat(scale, seed)
    w = orc.multiply_digest(orc.Mat(*a), orc.Mat(*a), nthreads=orc.host_threads(), rowstats=True)
    assert (d.nnz, d.hash, d.products, d.nnz_a) == (w.nnz, w.hash, w.products, w.nnz_a)
    assert np.array_equal(ctx.to_host(d.row_nnz, n, np.int64), w.row_nnz)
    assert np.array_equal(ctx.to_host(d.row_hash, n, np.uint64), w.row_hash)
    assert abs(d.sum - w.sum) <= 1e-10 * abs(w.sum)
    if flags == 0:
        r = ctx.multiply(A, A, sink=capi.SINK_COO)
        assert r.nnz == w.nnz
        # the stored tuples' own digest, from the device arrays (row-major order: the row of the last tuple is the last non-empty one)
        ci = ctx.to_host(r.idx0, int(r.nnz), np.int32)
        assert np.array_equal(np.bincount(ci, minlength=n), w.row_nnz)
        cj = ctx.to_host(r.idx1, int(r.nnz), np.int32)
        with np.errstate(over="ignore"):
            assert int(np.sum(orc.mix64(ci, cj), dtype=np.uint64)) == w.hash
    del keep


def test_cfg3_poisson4096_full_size_properties(ctx):
    """BASELINE cfg3: 5-point Poisson on a 4096^2 grid, A*A, both sinks; closed forms."""
    from spsparse_amd import capi
    N = 4096
    na = 5 * N * N - 4 * N
    A, keep = _device_operand(ctx, lambda *p: ctx.gen_poisson2d(N, *p), na, (N * N, N * N), sort0=0)
    d = ctx.multiply(A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    assert d.products == 25 * N * N - 36 * N + 8 and d.nnz == 13 * N * N - 20 * N + 4
    # row sums of A are 0 in the interior: (A A) 1 = A (A 1) is exact in integers
    i = np.arange(N * N)
    y, x = i // N, i % N
    a1 = 4.0 - (y > 0) - (y < N - 1) - (x > 0) - (x < N - 1)
    up = np.where(y > 0, np.roll(a1, N), 0.0)
    dn = np.where(y < N - 1, np.roll(a1, -N), 0.0)
    lf = np.where(x > 0, np.roll(a1, 1), 0.0)
    rt = np.where(x < N - 1, np.roll(a1, -1), 0.0)
    want = 4.0 * a1 - up - dn - lf - rt
    got = ctx.to_host(d.row_sum, N * N, np.float64)
    assert np.array_equal(got, want)
    r = ctx.multiply(A, A, sink=capi.SINK_COO)
    assert r.nnz == d.nnz
    # spot-check the COO result's first and last rows on the host
    head = ctx.to_host(r.val, 3, np.float64)
    assert head.tolist() == [18.0, -8.0, 1.0]


def test_cfg5_galerkin256_full_size_properties(ctx):
    """BASELINE cfg5: R*A*R^T on the 256^3 Laplacian; closed forms nnz(T)=32nc^3-24nc^2,
    nnz(C)=7nc^3-6nc^2 and the value set {24,-4} (every entry an integer: exact)."""
    from spsparse_amd import capi
    N, nc = 256, 128
    A, k1 = _device_operand(ctx, lambda *p: ctx.gen_laplace3d(N, *p), 7 * N ** 3 - 6 * N ** 2, (N ** 3, N ** 3), sort0=0)
    R, k2 = _device_operand(ctx, lambda *p: ctx.gen_aggregation3d(N, *p), N ** 3, (nc ** 3, N ** 3), sort0=0)
    rt = ctx.multiply(R, A, sink=capi.SINK_COO)
    nt = int(rt.nnz)
    assert nt == 32 * nc ** 3 - 24 * nc ** 2 and rt.products == 7 * N ** 3 - 6 * N ** 2
    # T is chained as it stands: the result buffers are the operand of the next call (no copy, no
    # re-consolidation), whose own result goes to the context's other output buffer
    T = capi.result_operand(rt)
    rc = ctx.multiply(T, R, tB="T", sink=capi.SINK_COO)
    assert rc.idx0 != rt.idx0 and rc.val != rt.val
    assert rc.ms_consolidate < 5.0                          # T (66.7M tuples) was inspected, not sorted
    assert rc.nnz == 7 * nc ** 3 - 6 * nc ** 2 and rc.products == nt
    vals = np.unique(ctx.to_host(rc.val, int(rc.nnz), np.float64))
    assert vals.tolist() == [-4.0, 24.0]
    ci = ctx.to_host(rc.idx0, int(rc.nnz), np.int32).astype(np.int64)
    cj = ctx.to_host(rc.idx1, int(rc.nnz), np.int32)
    assert np.all(np.diff(ci * nc ** 3 + cj) > 0)         # ascending (i, j), each once


def test_row_block_of_a_sharded_product_costs_its_share(ctx):
    """A row BLOCK of A times the whole of B (what one rank of a sharded product runs): the rows before and after the block
    are empty.  The dense row pointer once filled such a gap from ONE thread (10 ms per call at scale 20: the 8-block
    rehearsal fell from 5.7x to 3.0x); the device time of a 1/8 block must stay well under half of the whole product's,
    and its digest must add up with the other blocks' to the whole."""
    import torch
    from spsparse_amd import capi
    scale = 19
    n, ne = 1 << scale, 16 << scale
    A, keep = _device_operand(ctx, lambda *p: ctx.gen_rmat(scale, 3, 0, ne, *p), ne, (n, n))
    r = ctx.consolidate(A, 0)
    m = int(r.nnz)
    c0 = torch.empty(m, dtype=torch.int32, device="cuda:0"); c1 = torch.empty_like(c0); cv = torch.empty(m, dtype=torch.float64, device="cuda:0")
    ctx.memcpy(c0.data_ptr(), r.idx0, m * 4); ctx.memcpy(c1.data_ptr(), r.idx1, m * 4); ctx.memcpy(cv.data_ptr(), r.val, m * 8)
    torch.cuda.synchronize()
    B = capi.device_coo(c0.data_ptr(), c1.data_ptr(), cv.data_ptr(), m, (n, n), sort0=0)
    whole = min((ctx.multiply(B, B, sink=capi.SINK_DIGEST) for _ in range(3)), key=lambda x: x.ms_total)
    nnz = hsh = 0
    worst = 0.0
    for lo, hi in ((0, n // 64), (n // 64, n // 8), (n // 8, n // 2), (n // 2, n - 1000), (n - 1000, n)):
        sel = (c0 >= lo) & (c0 < hi)
        a0, a1, av = c0[sel].contiguous(), c1[sel].contiguous(), cv[sel].contiguous()
        blk = capi.device_coo(a0.data_ptr(), a1.data_ptr(), av.data_ptr(), a0.numel(), (n, n), sort0=0)
        d = min((ctx.multiply(blk, B, sink=capi.SINK_DIGEST) for _ in range(3)), key=lambda x: x.ms_total)
        nnz += d.nnz; hsh = (hsh + d.hash) % (1 << 64)
        # a block's time: its share of the products plus a fixed part (B's indices) that the whole product pays once
        share = d.products / max(1, whole.products)
        worst = max(worst, d.ms_total - share * whole.ms_total)
    assert nnz == whole.nnz and hsh == whole.hash
    print("row-block fixed part: %.2f ms of a whole %.2f ms" % (worst, whole.ms_total))
    assert worst < 0.2 * whole.ms_total, (worst, whole.ms_total)
    del keep


def test_cfg4_rmat23_single_gpu_properties(ctx):
    """BASELINE cfg4 on ONE GPU ("the same input must also run on 1 GPU"): R-MAT scale-23
    A*A, 134M raw tuples, P ~ 3.7e11, digest + row statistics.  Too large for the oracle: every row
    sum and the grand sum follow from linearity ((A A) 1 = A (A 1), duplicates included), the product
    count from the consolidated operand, and the 16384-column window path is the one taken."""
    import torch
    from spsparse_amd import capi
    scale, seed = 23, 1
    n, ne = 1 << scale, 16 << scale
    A, keep = _device_operand(ctx, lambda *p: ctx.gen_rmat(scale, seed, 0, ne, *p), ne, (n, n))
    d = ctx.multiply(A, A, sink=capi.SINK_DIGEST, flags=capi.SINK_ROWSTATS)
    assert d.window == 16384 and d.rows_heavy > 0 and d.cells_dense > 0 and d.cells_hash > 0
    # host side of the check: the raw tuples (the device generator is bit-identical to workloads.rmat,
    # test_device_generators_match_numpy) copied out instead of regenerated
    a0, a1, av = keep[0].cpu().numpy(), keep[1].cpu().numpy(), keep[2].cpu().numpy()
    b1 = np.bincount(a0, weights=av, minlength=n)
    want = np.bincount(a0, weights=av * b1[a1], minlength=n)
    got = ctx.to_host(d.row_sum, n, np.float64)
    nz = want != 0
    assert np.array_equal(got != 0, nz)
    assert np.max(np.abs(got[nz] - want[nz]) / want[nz]) <= 1e-10
    assert abs(d.sum - want.sum()) <= 1e-10 * want.sum()
    rn = ctx.to_host(d.row_nnz, n, np.int64)
    assert int(rn.sum()) == d.nnz and d.nnz <= d.products
    # products and nnz(A) from the distinct (row, col) pairs
    key = np.unique(a0.astype(np.int64) * n + a1)
    rowlen = np.bincount((key // n).astype(np.int64), minlength=n)
    assert d.nnz_a == key.size == d.nnz_b
    assert d.products == int(rowlen[(key % n).astype(np.int64)].sum())
    assert 1.0e11 < d.nnz < d.products and 3.0e11 < d.products < 4.5e11
    # pinned to the oracle on a fixed sample of the output rows (every 16th row starting at 5: hub rows, whose dense
    # cells only this size cuts with 16384-column windows, are among them): per row the tuple count and the index hash
    mask = np.zeros(n, np.uint8)
    mask[5::16] = 1
    mask[:64] = 1                                 # the heaviest rows of an un-permuted R-MAT are the first ones
    Ah = orc.Mat(a0, a1, av, (n, n))
    w = orc.multiply_digest(Ah, Ah, nthreads=orc.host_threads(), row_mask=mask, rowstats=True)
    assert w.nnz_a == d.nnz_a
    sel = mask == 1
    assert np.array_equal(rn[sel], w.row_nnz[sel])
    assert np.array_equal(ctx.to_host(d.row_hash, n, np.uint64)[sel], w.row_hash[sel])
    assert w.nnz == int(rn[sel].sum())
    del keep
    torch.cuda.empty_cache()
