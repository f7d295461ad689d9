"""Pins the CPU oracle (oracle/spsparse_oracle.c) to the reference's own test
vectors.  CPU only.  Each test names the reference test it restates
(paths relative to /root/reference).
"""
import gzip
import os

import numpy as np
import pytest

from oracle import binding as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------- test_array.cpp

def test_permutation():
    """tests/test_array.cpp:67-79"""
    i0, i1 = [1, 1, 0], [3, 2, 3]
    assert orc.sorted_permutation(i0, i1, 0).tolist() == [2, 1, 0]
    assert orc.sorted_permutation(i0, i1, 1).tolist() == [1, 2, 0]


def test_consolidate_known_answer():
    """tests/test_array.cpp:135-168"""
    i0 = [1, 1, 0, 0, 1]
    i1 = [3, 2, 3, 1, 2]
    v = [5., 3., 17., 14., 15.]
    o0, o1, ov = orc.consolidate(i0, i1, v, 0)
    assert o0.tolist() == [0, 0, 1, 1]
    assert o1.tolist() == [1, 3, 2, 3]
    assert ov.tolist() == [14., 17., 18., 5.]
    assert orc.dim_beginnings(o0).tolist() == [0, 2, 4]
    o0, o1, ov = orc.consolidate(i0, i1, v, 1)
    assert o0.tolist() == [0, 1, 0, 1]
    assert o1.tolist() == [1, 2, 3, 3]
    assert ov.tolist() == [14., 18., 17., 5.]
    assert orc.dim_beginnings(o1).tolist() == [0, 1, 2, 4]


def test_dim_beginnings_rows():
    """tests/test_array.cpp:170-218: non-empty rows 1, 2, 6 and their ranges"""
    i0, i1, v = [1, 1, 2, 6], [0, 3, 4, 4], [15., 17., 17., 10.]
    o0, o1, ov = orc.consolidate(i0, i1, v, 0)
    beg = orc.dim_beginnings(o0)
    assert beg.tolist() == [0, 2, 3, 4]
    assert [int(o0[b]) for b in beg[:-1]] == [1, 2, 6]
    assert o1[beg[0]:beg[1]].tolist() == [0, 3] and ov[beg[0]:beg[1]].tolist() == [15., 17.]
    assert o1[beg[1]:beg[2]].tolist() == [4]
    assert o1[beg[2]:beg[3]].tolist() == [4]
    assert orc.dim_beginnings([]).tolist() == []


def test_duplicate_policies():
    """algorithm.hpp:307-310 with the 'first added stays first' stable order (:404-406)"""
    i0, i1, v = [1, 0, 1, 1], [2, 0, 2, 2], [3., 9., 15., 7.]
    assert orc.consolidate(i0, i1, v, 0, orc.ADD)[2].tolist() == [9., 25.]
    assert orc.consolidate(i0, i1, v, 0, orc.LEAVE_ALONE)[2].tolist() == [9., 3.]
    assert orc.consolidate(i0, i1, v, 0, orc.REPLACE)[2].tolist() == [9., 7.]


def test_consolidate_zero_and_nan_quirks():
    """algorithm.hpp:272-275 vs :284-292"""
    nan = float("nan")
    # zeros are dropped before merging: +1 and -1 survive as an explicit 0.0
    o0, o1, ov = orc.consolidate([0, 0, 0], [1, 1, 2], [1., -1., 0.], 0)
    assert o1.tolist() == [1] and ov.tolist() == [0.0]
    # zero_nan only affects the leading run
    o0, o1, ov = orc.consolidate([0, 0, 0], [0, 1, 2], [nan, 2., nan], 0, orc.ADD, True)
    assert o1.tolist() == [1, 2] and ov[0] == 2. and np.isnan(ov[1])
    o0, o1, ov = orc.consolidate([0, 0], [0, 1], [nan, 2.], 0, orc.ADD, False)
    assert o1.tolist() == [0, 1]
    # all zero -> empty
    assert orc.consolidate([0, 1], [0, 1], [0., 0.], 0)[2].size == 0


# ---------------------------------------------------------------- test_xiter.cpp

def test_join2():
    """tests/test_xiter.cpp:52-98"""
    assert orc.join2([0, 2, 4, 6], list(range(8))).tolist() == [0, 2, 4, 6]
    assert orc.join2(list(range(8)), [0, 2, 4, 6]).tolist() == [0, 2, 4, 6]
    assert orc.join2([0, 2, 4, 5, 6, 7, 8, 9], [1, 2, 3, 4, 6]).tolist() == [2, 4, 6]
    assert orc.join2([], [1, 2]).tolist() == []
    assert orc.join2([1, 2], []).tolist() == []


def test_join3():
    """tests/test_xiter.cpp:102-125"""
    assert orc.join3([0, 2, 4, 6], list(range(8)), [1, 2, 3, 6]).tolist() == [2, 6]


# ------------------------------------------------------- test_multiply_sparse.cpp

def test_known_answer_row_scale_col():
    """tests/test_multiply_sparse.cpp:45-78 (the #if 0 case): {(0,0):128, (1,0):60}"""
    row = orc.Mat([0, 0, 0, 0, 1], [8, 4, 0, 3, 8], [6., 4., 2., 3., 3.], (2, 10))
    scale = orc.Vec([0, 4, 8], [2., 4., 4.], 10)
    col = orc.Mat([0, 3, 8], [0, 0, 0], [2., 3., 5.], (10, 1))
    eye = orc.Vec(list(range(10)), [1.] * 10, 10)
    for rowwise in (False, True):
        i, j, v, shape = orc.multiply(row, col, 1.0, scalei=eye, scalej=scale, scalek=eye, rowwise=rowwise)
        assert shape == (2, 1)
        assert i.tolist() == [0, 1] and j.tolist() == [0, 0] and v.tolist() == [128., 60.]


def _load_random5():
    mm, mv = [], []
    with gzip.open(os.path.join(GOLDEN, "random5_inputs.txt.gz"), "rt") as f:
        for line in f:
            tok = line.split()
            kind, seed = tok[0], int(tok[1])
            p = 2
            parts = []
            for _ in range(2):
                name, n = tok[p], int(tok[p + 1])
                p += 2
                width = 2 if name == "V" else 3
                rec = tok[p:p + n * width]
                p += n * width
                idx = [[int(rec[t * width + d]) for t in range(n)] for d in range(width - 1)]
                val = [float.fromhex(rec[t * width + width - 1]) for t in range(n)]
                parts.append((idx, val))
            (mm if kind == "MM" else mv).append((seed, parts))
    return mm, mv


def _dense(idx, val, shape):
    """VectorCooArray::to_dense (VectorCooArray.hpp:313-321): DenseAccum ADD, in insertion order."""
    d = np.zeros(shape)
    for t, v in enumerate(val):
        d[tuple(ix[t] for ix in idx)] += v
    return d


RANDOM5 = _load_random5()


def test_random_mm_property():
    """tests/test_multiply_sparse.cpp:84-136: sparse product == dense triple loop,
    seeds 1..999, scalej = eye (Join3 path).  The reference asserts 4 ULP
    (EXPECT_DOUBLE_EQ); the restatement is held to exact equality."""
    mm, _ = RANDOM5
    assert len(mm) == 999
    eye = orc.Vec(list(range(5)), [1.] * 5, 5)
    ntuples = 0
    for seed, ((ai, av), (bi, bv)) in mm:
        A = orc.Mat(ai[0], ai[1], av, (5, 5))
        B = orc.Mat(bi[0], bi[1], bv, (5, 5))
        Ad, Bd = _dense(ai, av, (5, 5)), _dense(bi, bv, (5, 5))
        want = np.zeros((5, 5))
        for i in range(5):
            for j in range(5):
                s = 0.0
                for k in range(5):
                    s += Ad[i, k] * Bd[k, j]
                want[i, j] = s
        for rowwise in (False, True):
            i, j, v, shape = orc.multiply(A, B, 1.0, scalej=eye, rowwise=rowwise)
            assert shape == (5, 5)
            got = _dense([i, j], v, (5, 5))
            assert np.array_equal(got, want), seed
            # ascending (row, col), each at most once, no explicit zeros
            keys = i.astype(np.int64) * 5 + j
            assert np.all(np.diff(keys) > 0) and np.all(v != 0)
        ntuples += len(v)
    assert ntuples > 5000


def test_random_mv_property():
    """tests/test_multiply_sparse.cpp:138-203: exact equality (`sum != Cd(i)` fails)"""
    _, mv = RANDOM5
    assert len(mv) == 999
    for seed, ((ai, av), (vi, vv)) in mv:
        A = orc.Mat(ai[0], ai[1], av, (5, 5))
        V = orc.Vec(vi[0], vv, 5)
        Ad, Vd = _dense(ai, av, (5, 5)), _dense(vi, vv, (5,))
        i, _, v, shape = orc.multiply_mv(A, V)
        assert shape == (5,)
        got = _dense([i], v, (5,))
        for r in range(5):
            s = 0.0
            for k in range(5):
                s += Ad[r, k] * Vd[k]
            assert s == got[r], seed


# ------------------------------------------------- behaviours recorded in SURVEY.md

def _rand_mat(rng, shape, nnz, zeros=False):
    i0 = rng.integers(0, shape[0], nnz)
    i1 = rng.integers(0, shape[1], nnz)
    v = rng.uniform(-1, 1, nnz)
    if zeros:
        v[rng.integers(0, nnz, max(1, nnz // 10))] = 0.0
    return orc.Mat(i0, i1, v, shape)


def _rand_vec(rng, n, density=0.7, zeros=True):
    idx = np.flatnonzero(rng.uniform(size=n) < density)
    if idx.size == 0:
        idx = np.array([0])
    v = rng.uniform(0.5, 2, idx.size)
    if zeros and idx.size > 3:
        v[1] = 0.0
    return orc.Vec(idx, v, n)


@pytest.mark.parametrize("tA", [".", "T"])
@pytest.mark.parametrize("tB", [".", "T"])
def test_rowwise_equals_innerproduct_bitwise(tA, tB):
    """The scalable checker must reproduce the restated reference bit for bit:
    transposes x scale vectors x C != 1 x duplicate policies."""
    rng = np.random.default_rng(7)
    for trial in range(12):
        m, k, n = rng.integers(1, 40, 3)
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 300)), zeros=True)
        B = _rand_mat(rng, (n, k) if tB == "T" else (k, n), int(rng.integers(1, 300)), zeros=True)
        si = _rand_vec(rng, m) if trial % 2 else None
        sj = _rand_vec(rng, k) if trial % 3 else None
        sk = _rand_vec(rng, n) if trial % 4 == 1 else None
        dup = [orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3]
        a = orc.multiply(A, B, 17.0, si, tA, sj, tB, sk, dup)
        b = orc.multiply(A, B, 17.0, si, tA, sj, tB, sk, dup, rowwise=True)
        c = orc.multiply(A, B, 17.0, si, tA, sj, tB, sk, dup, rowwise=True, nthreads=3)
        for x, y in ((a, b), (a, c)):
            assert x[3] == y[3] == (m, n)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            assert np.array_equal(x[2], y[2])


def test_rowwise_equals_innerproduct_with_nan_and_zero_nan():
    """zero_nan drops the NaNs of the LEADING run of consolidate's own sorted sequence only
    (algorithm.hpp:272-275 vs :284-292), and B's sequence is column-major (multiply_sparse.hpp:168):
    the row-wise checker must consolidate B in that order before re-sorting it.  Includes the
    two-tuple case where a row-major reading of B drops the wrong NaN."""
    A = orc.Mat([0, 0], [0, 1], [1., 1.], (1, 2))
    B = orc.Mat([0, 1], [1, 0], [np.nan, 5.], (2, 2))
    a = orc.multiply(A, B, zero_nan=True)
    b = orc.multiply(A, B, zero_nan=True, rowwise=True)
    assert a[0].tolist() == b[0].tolist() == [0, 0] and a[1].tolist() == b[1].tolist() == [0, 1]
    assert a[2][0] == b[2][0] == 5.0 and np.isnan(a[2][1]) and np.isnan(b[2][1])
    rng = np.random.default_rng(11)
    for trial in range(60):
        m, k, n = rng.integers(1, 12, 3)
        tA, tB = ".T"[trial % 2], ".T"[(trial // 2) % 2]
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 60)), zeros=True)
        B = _rand_mat(rng, (n, k) if tB == "T" else (k, n), int(rng.integers(1, 60)), zeros=True)
        for M in (A, B):
            M.val[rng.integers(0, M.val.size, max(1, M.val.size // 4))] = np.nan
        dup = [orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3]
        for zn in (False, True):
            x = orc.multiply(A, B, 1.0, None, tA, None, tB, None, dup, zero_nan=zn)
            y = orc.multiply(A, B, 1.0, None, tA, None, tB, None, dup, zero_nan=zn, rowwise=True)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            assert np.array_equal(x[2], y[2], equal_nan=True)


def test_streaming_digest_equals_the_stored_product():
    """The streaming digest mode of the checker (count, index hash, value sum, per-row count and hash; rows handed
    out dynamically; optional row mask) against the digest of the restated reference's stored tuples: transposes x
    scale vectors x C != 1 x duplicate policies x zero_nan, one and several threads."""
    rng = np.random.default_rng(23)
    for trial in range(24):
        m, k, n = rng.integers(1, 60, 3)
        tA, tB = ".T"[trial % 2], ".T"[(trial // 2) % 2]
        A = _rand_mat(rng, (k, m) if tA == "T" else (m, k), int(rng.integers(1, 500)), zeros=True)
        B = _rand_mat(rng, (n, k) if tB == "T" else (k, n), int(rng.integers(1, 500)), zeros=True)
        si = _rand_vec(rng, m) if trial % 2 else None
        sj = _rand_vec(rng, k) if trial % 3 else None
        sk = _rand_vec(rng, n) if trial % 4 == 1 else None
        dup = [orc.ADD, orc.LEAVE_ALONE, orc.REPLACE][trial % 3]
        zn = trial % 5 == 0
        if zn:
            B.val[rng.integers(0, B.val.size, max(1, B.val.size // 8))] = np.nan
        wi, wj, wv, _ = orc.multiply(A, B, 3.0, si, tA, sj, tB, sk, dup, zero_nan=zn)
        keep = ~np.isnan(wv)
        cnt, _, h = orc.digest(wi, wj, wv)
        for nt in (1, 4):
            d = orc.multiply_digest(A, B, 3.0, si, tA, sj, tB, sk, dup, zero_nan=zn, nthreads=nt, rowstats=True)
            assert (d.nnz, d.hash) == (cnt, h)
            assert np.array_equal(d.row_nnz, np.bincount(wi, minlength=m))
            rh = np.zeros(m, np.uint64)
            np.add.at(rh, wi, orc.mix64(wi, wj))
            assert np.array_equal(d.row_hash, rh)
            if keep.all():
                assert abs(d.sum - wv.sum()) <= 1e-12 * np.abs(wv).sum()
        mask = (rng.uniform(size=m) < 0.4).astype(np.uint8)
        d = orc.multiply_digest(A, B, 3.0, si, tA, sj, tB, sk, dup, zero_nan=zn, nthreads=3, row_mask=mask, rowstats=True)
        sel = mask[wi] == 1
        assert d.nnz == int(sel.sum()) and d.hash == orc.digest(wi[sel], wj[sel], wv[sel])[2]
        assert np.array_equal(d.row_nnz, np.bincount(wi[sel], minlength=m))


def test_large_inputs_sort_by_radix_to_the_same_permutation():
    """orc_sorted_permutation takes a stable LSD radix sort from 65536 tuples on (BASELINE-size operands); it must be
    the permutation of the restated std::stable_sort (algorithm.hpp:411-427), duplicates keeping insertion order."""
    rng = np.random.default_rng(3)
    n = 150000
    i0 = rng.integers(0, 700, n).astype(np.int32)
    i1 = rng.integers(0, 70000, n).astype(np.int32)
    for so in (0, 1):
        assert np.array_equal(orc.sorted_permutation(i0, i1, so), orc.sorted_permutation_merge(i0, i1, so))
    assert np.array_equal(orc.sorted_permutation(i1, None, 0), orc.sorted_permutation_merge(i1, None, 0))
    i0[5] = -3                                   # a negative index: the merge sort handles it (no radix path)
    assert np.array_equal(orc.sorted_permutation(i0, i1, 0), orc.sorted_permutation_merge(i0, i1, 0))


def test_ab_equals_btat_transposed():
    """multiply_sparse.hpp:15-18 doc example: AB == (B^T A^T)^T, bitwise."""
    rng = np.random.default_rng(3)
    A, B = _rand_mat(rng, (40, 30), 200), _rand_mat(rng, (30, 50), 250)
    i, j, v, shape = orc.multiply(A, B)
    i2, j2, v2, shape2 = orc.multiply(B, A, tA="T", tB="T")
    assert shape == (40, 50) and shape2 == (50, 40)
    o = np.lexsort((i2, j2))
    assert np.array_equal(i, j2[o]) and np.array_equal(j, i2[o]) and np.array_equal(v, v2[o])


def test_scale_semantics():
    """multiply_sparse.hpp:195,211,228,242: rows/cols absent from (or 0 in) scalei/scalek are
    skipped, k absent from scalej drops the term, value = sum*C*a_scale*b_scale."""
    A = orc.Mat([0, 1, 2], [0, 0, 0], [2., 3., 5.], (3, 1))
    B = orc.Mat([0, 0, 0], [0, 1, 2], [7., 11., 13.], (1, 3))
    si = orc.Vec([0, 2], [3., 0.], 3)           # row 1 absent, row 2 scale 0
    sk = orc.Vec([1, 2], [0.5, 0.], 3)          # col 0 absent, col 2 scale 0
    i, j, v, _ = orc.multiply(A, B, 17.0, si, ".", None, ".", sk)
    assert i.tolist() == [0] and j.tolist() == [1]
    assert v[0] == 2. * 11. * 17.0 * 3. * 0.5
    sj = orc.Vec([3], [1.], 4)
    A2 = orc.Mat([0], [0], [1.], (1, 4))
    B2 = orc.Mat([0], [0], [1.], (4, 1))
    assert orc.multiply(A2, B2, scalej=sj)[2].size == 0


def test_short_circuits_error_and_append():
    """multiply_sparse.hpp:166-184: shape set first, then the dimension error, then the
    empty short-circuits; C == 0 returns nothing."""
    A = orc.Mat([0], [1], [2.], (2, 3))
    B = orc.Mat([1], [0], [4.], (3, 2))
    E = orc.Mat([], [], [], (3, 2))
    assert orc.multiply(A, B)[2].tolist() == [8.]
    assert orc.multiply(A, B, C_=0.0)[2].size == 0
    assert orc.multiply(A, E)[2].size == 0 and orc.multiply(A, E)[3] == (2, 2)
    assert orc.multiply(A, B, scalej=orc.Vec([], [], 3))[2].size == 0
    with pytest.raises(orc.OracleError, match=r"Inner dimensions for A \(3\) and B \(2\) must match!"):
        orc.multiply(A, orc.Mat([0], [0], [1.], (2, 2)))
    # exact zero sums are dropped (multiply_sparse.hpp:238)
    A3 = orc.Mat([0, 0], [0, 1], [1., -1.], (1, 2))
    B3 = orc.Mat([0, 1], [0, 0], [1., 1.], (2, 1))
    assert orc.multiply(A3, B3)[2].size == 0
    # an operand holding only explicit zeros: empty product
    Z = orc.Mat([0], [1], [0.], (2, 3))
    assert orc.multiply(Z, B)[2].size == 0 and orc.multiply(Z, B, rowwise=True)[2].size == 0


def test_presorted_operand_is_trusted():
    """algorithm.hpp:360: an operand whose sort_order already matches is used as is."""
    rng = np.random.default_rng(11)
    A, B = _rand_mat(rng, (20, 20), 100), _rand_mat(rng, (20, 20), 100)
    a0, a1, av = orc.consolidate(A.idx0, A.idx1, A.val, 0)
    b0, b1, bv = orc.consolidate(B.idx0, B.idx1, B.val, 1)
    As, Bs = orc.Mat(a0, a1, av, (20, 20), sort0=0), orc.Mat(b0, b1, bv, (20, 20), sort0=1)
    x, y = orc.multiply(A, B), orc.multiply(As, Bs)
    assert all(np.array_equal(p, q) for p, q in zip(x[:3], y[:3]))


def test_cfg1_against_scipy():
    """BASELINE cfg1 (1k x 1k, 10 per row): the restated reference algorithm, the row-wise
    checker and an independent library product agree."""
    import scipy.sparse as sp
    from spsparse_amd import workloads as wl
    a = wl.random_rows(1000, 10, seed=1, stream_base=0)
    b = wl.random_rows(1000, 10, seed=1, stream_base=8)
    A, B = orc.Mat(*a), orc.Mat(*b)
    i, j, v, shape = orc.multiply(A, B)
    i2, j2, v2, _ = orc.multiply(A, B, rowwise=True, nthreads=4)
    assert np.array_equal(i, i2) and np.array_equal(j, j2) and np.array_equal(v, v2)
    S = (sp.coo_matrix((a[2], (a[0], a[1])), shape=a[3]).tocsr() @
         sp.coo_matrix((b[2], (b[0], b[1])), shape=b[3]).tocsr()).tocoo()
    S.sum_duplicates()
    o = np.lexsort((S.col, S.row))
    assert np.array_equal(i, S.row[o]) and np.array_equal(j, S.col[o])
    assert np.max(np.abs(v - S.data[o]) / np.abs(v)) < 1e-12
    assert 90000 < len(v) < 100000
