"""Synthetic operands for BASELINE.json's configs (host/numpy side).

Everything is a pure function of (parameters, seed) built on a counter-based
splitmix64 stream, so the numpy generators here and the device generators in
csrc/workload.hip produce bit-identical tuples; csrc/workload_common.h holds
the same arithmetic for C/HIP.  No reference code involved: the reference has
no generators (its tests draw 5x5 inputs from std::default_random_engine,
tests/test_multiply_sparse.cpp:86-95).

All generators return (idx0, idx1, val, shape) with int32 indices, float64
values, in generation order (R-MAT / random: unsorted with duplicates;
stencils: row-major sorted, unique).
"""
import numpy as np

_U = np.uint64
GOLD = _U(0x9E3779B97F4A7C15)
M1 = _U(0xBF58476D1CE4E5B9)
M2 = _U(0x94D049BB133111EB)
STREAM_MUL = _U(0xD1B54A32D192ED03)

# R-MAT quadrant thresholds on a 16-bit draw: a,b,c,d = 0.57,0.19,0.19,0.05
RMAT_TA = 37356
RMAT_TAB = 49808
RMAT_TABC = 62260


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + GOLD
        z = (z ^ (z >> _U(30))) * M1
        z = (z ^ (z >> _U(27))) * M2
    return z ^ (z >> _U(31))


def stream_key(seed, stream):
    with np.errstate(over="ignore"):
        return splitmix64(_U(seed) ^ (_U(stream) * STREAM_MUL))


def draw(seed, stream, ctr):
    """64 random bits for counter(s) ctr of (seed, stream)."""
    return splitmix64(stream_key(seed, stream) ^ splitmix64(ctr))


def unit_open(r):
    """(0,1] double from 64 random bits: never 0, so no tuple is dropped."""
    return ((r >> _U(11)) + _U(1)).astype(np.float64) * (2.0 ** -53)


def random_rows(n, per_row, seed, stream_base=0):
    """cfg1: row i gets per_row tuples (i, U{0..n-1}, U(0,1]); duplicates kept."""
    t = np.arange(n * per_row, dtype=np.uint64)
    rows = (t // _U(per_row)).astype(np.int32)
    cols = (draw(seed, stream_base + 0, t) % _U(n)).astype(np.int32)
    vals = unit_open(draw(seed, stream_base + 1, t))
    return rows, cols, vals, (n, n)


def rmat(scale, seed, edge_factor=16, first_edge=0, n_edges=None):
    """Graph500-style R-MAT, a,b,c,d = 0.57,0.19,0.19,0.05, no vertex
    scrambling; one 16-bit draw per level, four levels per 64-bit word."""
    n = 1 << scale
    total = edge_factor * n
    if n_edges is None:
        n_edges = total - first_edge
    e = np.arange(first_edge, first_edge + n_edges, dtype=np.uint64)
    row = np.zeros(n_edges, dtype=np.uint32)
    col = np.zeros(n_edges, dtype=np.uint32)
    words = (scale + 3) // 4
    for w in range(words):
        r = draw(seed, 2, e * _U(words) + _U(w))
        for q in range(4):
            level = w * 4 + q
            if level >= scale:
                break
            d = ((r >> _U(16 * q)) & _U(0xFFFF)).astype(np.uint32)
            rb = (d >= RMAT_TAB).astype(np.uint32)
            cb = (((d >= RMAT_TA) & (d < RMAT_TAB)) | (d >= RMAT_TABC)).astype(np.uint32)
            row = (row << np.uint32(1)) | rb
            col = (col << np.uint32(1)) | cb
    vals = unit_open(draw(seed, 3, e))
    return row.astype(np.int32), col.astype(np.int32), vals, (n, n)


def poisson2d(N):
    """5-point stencil on an N x N grid, Dirichlet: diag 4, off-diag -1;
    row-major grid numbering, tuples sorted row-major."""
    i = np.arange(N * N, dtype=np.int64)
    y, x = i // N, i % N
    offs = [(-N, y > 0), (-1, x > 0), (0, np.ones_like(i, dtype=bool)), (1, x < N - 1), (N, y < N - 1)]
    rows = np.concatenate([i[m] for _, m in offs])
    cols = np.concatenate([(i + o)[m] for o, m in offs])
    vals = np.concatenate([np.full(int(m.sum()), 4.0 if o == 0 else -1.0) for o, m in offs])
    order = np.lexsort((cols, rows))
    return rows[order].astype(np.int32), cols[order].astype(np.int32), vals[order], (N * N, N * N)


def laplace3d(N):
    """7-point Laplacian on an N^3 grid, Dirichlet: diag 6, off-diag -1."""
    i = np.arange(N ** 3, dtype=np.int64)
    x, y, z = i % N, (i // N) % N, i // (N * N)
    t = np.ones_like(i, dtype=bool)
    offs = [(-N * N, z > 0), (-N, y > 0), (-1, x > 0), (0, t), (1, x < N - 1), (N, y < N - 1), (N * N, z < N - 1)]
    rows = np.concatenate([i[m] for _, m in offs])
    cols = np.concatenate([(i + o)[m] for o, m in offs])
    vals = np.concatenate([np.full(int(m.sum()), 6.0 if o == 0 else -1.0) for o, m in offs])
    order = np.lexsort((cols, rows))
    return rows[order].astype(np.int32), cols[order].astype(np.int32), vals[order], (N ** 3, N ** 3)


def aggregation3d(N):
    """Piecewise-constant 2x2x2 aggregation R: (N/2)^3 x N^3, one 1.0 per
    fine cell (column); tuples sorted row-major."""
    assert N % 2 == 0
    n_c = N // 2
    f = np.arange(N ** 3, dtype=np.int64)
    x, y, z = f % N, (f // N) % N, f // (N * N)
    c = ((z // 2) * n_c + (y // 2)) * n_c + (x // 2)
    order = np.lexsort((f, c))
    return c[order].astype(np.int32), f[order].astype(np.int32), np.ones(N ** 3), (n_c ** 3, N ** 3)
