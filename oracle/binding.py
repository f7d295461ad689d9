"""ctypes binding of oracle/libspsparse_oracle.so (TEST INFRASTRUCTURE ONLY).

The C file restates the reference algorithm (citations inside); this module
only marshals numpy arrays.  Build with `make -C oracle`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspsparse_oracle.so")

LEAVE_ALONE, ADD, REPLACE = 0, 1, 2


class _Coo(C.Structure):
    _fields_ = [("i", C.POINTER(C.c_int32)), ("j", C.POINTER(C.c_int32)),
                ("v", C.POINTER(C.c_double)), ("n", C.c_size_t), ("cap", C.c_size_t),
                ("shape0", C.c_size_t), ("shape1", C.c_size_t), ("rank", C.c_int)]


class _Mat(C.Structure):
    _fields_ = [("idx0", C.c_void_p), ("idx1", C.c_void_p), ("val", C.c_void_p),
                ("nnz", C.c_size_t), ("shape0", C.c_size_t), ("shape1", C.c_size_t),
                ("sort0", C.c_int)]


class _Vec(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("val", C.c_void_p), ("nnz", C.c_size_t),
                ("shape0", C.c_size_t), ("sort0", C.c_int)]


class _DigestOut(C.Structure):
    _fields_ = [("count", C.c_uint64), ("hash", C.c_uint64), ("products", C.c_uint64), ("nnz_a", C.c_uint64),
                ("nnz_b", C.c_uint64), ("sum", C.c_double), ("row_nnz", C.c_void_p), ("row_hash", C.c_void_p)]


def build(force=False):
    src = os.path.join(_HERE, "spsparse_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_coo_init.argtypes = [C.POINTER(_Coo), C.c_int]
        L.orc_coo_free.argtypes = [C.POINTER(_Coo)]
        L.orc_sorted_permutation.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_consolidate.restype = C.c_size_t
        L.orc_consolidate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dim_beginnings.restype = C.c_size_t
        L.orc_dim_beginnings.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_join2.restype = C.c_size_t
        L.orc_join2.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_join3.restype = C.c_size_t
        L.orc_join3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        mm = [C.POINTER(_Coo), C.c_double, C.POINTER(_Vec), C.POINTER(_Mat), C.c_char, C.POINTER(_Vec),
              C.POINTER(_Mat), C.c_char, C.POINTER(_Vec), C.c_int, C.c_int]
        L.orc_multiply_mm.argtypes = mm + [C.c_char_p, C.c_size_t]
        L.orc_multiply_mm_rowwise.argtypes = mm + [C.c_int, C.c_char_p, C.c_size_t]
        L.orc_multiply_mv.argtypes = [C.POINTER(_Coo), C.c_double, C.POINTER(_Vec), C.POINTER(_Mat), C.c_char,
                                      C.POINTER(_Vec), C.POINTER(_Vec), C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.orc_multiply_mm_rowwise_digest.argtypes = [C.POINTER(_DigestOut), C.c_void_p] + mm[1:] + [C.c_int, C.c_char_p, C.c_size_t]
        L.orc_sorted_permutation_merge.argtypes = L.orc_sorted_permutation.argtypes
        L.orc_mix64.restype = C.c_uint64
        L.orc_mix64.argtypes = [C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


class OracleError(Exception):
    """The restated (*spsparse_error)(-1, ...) path (spsparse.cpp:12-28)."""


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Mat:
    """Host COO matrix operand: (idx0, idx1, val, shape, sort0)."""

    def __init__(self, idx0, idx1, val, shape, sort0=-1):
        self.idx0, self.idx1, self.val = _i32(idx0), _i32(idx1), _f64(val)
        assert self.idx0.shape == self.idx1.shape == self.val.shape
        self.shape = (int(shape[0]), int(shape[1]))
        self.sort0 = sort0

    @property
    def nnz(self):
        return self.val.size

    def _c(self):
        return _Mat(self.idx0.ctypes.data, self.idx1.ctypes.data, self.val.ctypes.data,
                    self.nnz, self.shape[0], self.shape[1], self.sort0)


class Vec:
    """Host COO vector operand (scale vectors, MV right-hand side)."""

    def __init__(self, idx, val, shape0, sort0=-1):
        self.idx, self.val = _i32(idx), _f64(val)
        self.shape0 = int(shape0)
        self.sort0 = sort0

    @property
    def nnz(self):
        return self.val.size

    def _c(self):
        return _Vec(self.idx.ctypes.data, self.val.ctypes.data, self.nnz, self.shape0, self.sort0)


def _take(coo):
    n = coo.n
    i = np.ctypeslib.as_array(coo.i, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    v = np.ctypeslib.as_array(coo.v, shape=(n,)).copy() if n else np.zeros(0, np.float64)
    if coo.rank == 2:
        j = np.ctypeslib.as_array(coo.j, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    else:
        j = None
    shape = (coo.shape0, coo.shape1) if coo.rank == 2 else (coo.shape0,)
    lib().orc_coo_free(C.byref(coo))
    return i, j, v, shape


def sorted_permutation(idx0, idx1, so0):
    idx0 = _i32(idx0)
    rank = 1 if idx1 is None else 2
    idx1 = None if idx1 is None else _i32(idx1)
    perm = np.zeros(idx0.size, dtype=np.uintp)
    lib().orc_sorted_permutation(rank, idx0.ctypes.data, None if idx1 is None else idx1.ctypes.data,
                                 idx0.size, so0, perm.ctypes.data)
    return perm


def consolidate(idx0, idx1, val, so0, duplicate_policy=ADD, zero_nan=False):
    idx0, val = _i32(idx0), _f64(val)
    rank = 1 if idx1 is None else 2
    idx1 = None if idx1 is None else _i32(idx1)
    n = val.size
    o0 = np.zeros(n, np.int32)
    o1 = np.zeros(n, np.int32)
    ov = np.zeros(n, np.float64)
    m = lib().orc_consolidate(rank, idx0.ctypes.data, None if idx1 is None else idx1.ctypes.data,
                              val.ctypes.data, n, so0, duplicate_policy, int(zero_nan),
                              o0.ctypes.data, o1.ctypes.data if rank == 2 else None, ov.ctypes.data)
    return o0[:m], (o1[:m] if rank == 2 else None), ov[:m]


def dim_beginnings(lead):
    lead = _i32(lead)
    out = np.zeros(lead.size + 1, dtype=np.uintp)
    m = lib().orc_dim_beginnings(lead.ctypes.data, lead.size, out.ctypes.data)
    return out[:m]


def join2(a, b):
    a, b = _i32(a), _i32(b)
    out = np.zeros(min(a.size, b.size), np.int32)
    m = lib().orc_join2(a.ctypes.data, a.size, b.ctypes.data, b.size, out.ctypes.data)
    return out[:m]


def join3(a, b, c):
    a, b, c = _i32(a), _i32(b), _i32(c)
    out = np.zeros(min(a.size, b.size, c.size), np.int32)
    m = lib().orc_join3(a.ctypes.data, a.size, b.ctypes.data, b.size, c.ctypes.data, c.size, out.ctypes.data)
    return out[:m]


def multiply(A, B, C_=1.0, scalei=None, tA='.', scalej=None, tB='.', scalek=None,
             duplicate_policy=ADD, zero_nan=False, rowwise=False, nthreads=1):
    """spsparse::multiply (MM).  Returns (i, j, v, shape) appended tuples.

    rowwise=False is the restated reference algorithm (inner product);
    rowwise=True the scalable row-wise checker with identical results.
    """
    coo = _Coo()
    lib().orc_coo_init(C.byref(coo), 2)
    msg = C.create_string_buffer(256)
    # keep the ctypes structs (and thus the numpy buffers) alive across the call
    keep = [x._c() if x is not None else None for x in (scalei, A, scalej, B, scalek)]
    ptr = [None if k is None else C.byref(k) for k in keep]
    args = [C.byref(coo), float(C_), ptr[0], ptr[1], tA.encode(), ptr[2], ptr[3], tB.encode(), ptr[4],
            duplicate_policy, int(zero_nan)]
    if rowwise:
        rc = lib().orc_multiply_mm_rowwise(*args, nthreads, msg, 256)
    else:
        rc = lib().orc_multiply_mm(*args, msg, 256)
    out = _take(coo)
    if rc != 0:
        raise OracleError(msg.value.decode())
    return out


class Digest:
    """Result of multiply_digest: count / hash / sum as the device's digest sink reports them, the scalar
    products, the consolidated operand sizes, and (rowstats) per-row tuple counts and index hashes."""

    def __init__(self, d, row_nnz, row_hash):
        self.nnz, self.hash, self.sum = int(d.count), int(d.hash), float(d.sum)
        self.products, self.nnz_a, self.nnz_b = int(d.products), int(d.nnz_a), int(d.nnz_b)
        self.row_nnz, self.row_hash = row_nnz, row_hash


def host_threads(cap=64):
    """Threads worth starting on this host: the CPUs this process may run on, less where a cgroup quota says so."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def multiply_digest(A, B, C_=1.0, scalei=None, tA='.', scalej=None, tB='.', scalek=None,
                    duplicate_policy=ADD, zero_nan=False, nthreads=1, row_mask=None, rowstats=False):
    """The row-wise checker as a streaming digest (no tuple is stored); row_mask: uint8 per row of op(A),
    only rows with a non-zero byte are evaluated."""
    nrow = A.shape[1] if tA == 'T' else A.shape[0]
    d = _DigestOut()
    row_nnz = row_hash = None
    if rowstats:
        row_nnz, row_hash = np.zeros(nrow, np.int64), np.zeros(nrow, np.uint64)
        d.row_nnz, d.row_hash = row_nnz.ctypes.data, row_hash.ctypes.data
    mask = None
    if row_mask is not None:
        mask = np.ascontiguousarray(row_mask, dtype=np.uint8)
        assert mask.size == nrow
    msg = C.create_string_buffer(256)
    keep = [x._c() if x is not None else None for x in (scalei, A, scalej, B, scalek)]
    ptr = [None if k is None else C.byref(k) for k in keep]
    rc = lib().orc_multiply_mm_rowwise_digest(C.byref(d), None if mask is None else mask.ctypes.data, float(C_), ptr[0], ptr[1],
                                              tA.encode(), ptr[2], ptr[3], tB.encode(), ptr[4], duplicate_policy,
                                              int(zero_nan), int(nthreads), msg, 256)
    if rc != 0:
        raise OracleError(msg.value.decode())
    return Digest(d, row_nnz, row_hash)


def sorted_permutation_merge(idx0, idx1, so0):
    """orc_sorted_permutation through the merge sort alone (the radix path of large inputs is checked against it)."""
    idx0 = _i32(idx0)
    rank = 1 if idx1 is None else 2
    idx1 = None if idx1 is None else _i32(idx1)
    perm = np.zeros(idx0.size, dtype=np.uintp)
    lib().orc_sorted_permutation_merge(rank, idx0.ctypes.data, None if idx1 is None else idx1.ctypes.data,
                                       idx0.size, so0, perm.ctypes.data)
    return perm


def multiply_mv(A, V, C_=1.0, scalei=None, tA='.', scalej=None, duplicate_policy=ADD, zero_nan=False):
    """spsparse::multiply (MV).  Returns (i, None, v, shape)."""
    coo = _Coo()
    lib().orc_coo_init(C.byref(coo), 1)
    msg = C.create_string_buffer(256)
    keep = [x._c() if x is not None else None for x in (scalei, A, scalej, V)]
    ptr = [None if k is None else C.byref(k) for k in keep]
    rc = lib().orc_multiply_mv(C.byref(coo), float(C_), ptr[0], ptr[1], tA.encode(), ptr[2], ptr[3],
                               duplicate_policy, int(zero_nan), msg, 256)
    out = _take(coo)
    if rc != 0:
        raise OracleError(msg.value.decode())
    return out


def mix64(i, j):
    """Vectorised orc_mix64 (numpy uint64, wraps mod 2^64)."""
    x = (np.asarray(i).astype(np.uint64) << np.uint64(32)) | np.asarray(j).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x * np.uint64(0x9E3779B97F4A7C15)
    return x ^ (x >> np.uint64(29))


def digest(i, j, v):
    """(count, sum, hash) -- the digest the device checksum sink reports."""
    with np.errstate(over="ignore"):
        h = int(np.sum(mix64(i, j if j is not None else np.zeros_like(i)), dtype=np.uint64))
    return int(len(v)), float(np.sum(v)), h
