"""ctypes binding of libspsparse_amd.so -- the C ABI in include/spsparse_amd.h.

Thin marshalling only: numpy arrays (host operands) or raw device pointers
(e.g. torch tensors' data_ptr()) go in, the library's result struct comes out.
There is no CPU fallback: if the shared library is missing or no MI355X is
visible this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPSAMD_LIB") or os.path.join(_HERE, "lib", "libspsparse_amd.so")    # SPSAMD_LIB: a developer variant build

LEAVE_ALONE, ADD, REPLACE = 0, 1, 2
MEM_HOST, MEM_DEVICE, MEM_PREPARED = 0, 1, 2
AS_A, AS_B = 1, 2
SINK_COO, SINK_DIGEST = 1, 2
SINK_ROWSTATS = 1
SINK_ORDERED = 2
SINK_PERMUTE = 4
SINK_EXACT_PATTERN = 8

ERRORS = {-1: "EDIM", -2: "EINVAL", -3: "EHIP", -4: "ENOMEM", -5: "ECAPACITY", -6: "ENODEVICE", -7: "EPEER"}


class SpsamdError(RuntimeError):
    """A negative return code of the C ABI (the shim's (*spsparse_error)(-1, msg))."""

    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code
        self.msg = msg


class Coo(C.Structure):
    _fields_ = [("idx0", C.c_void_p), ("idx1", C.c_void_p), ("val", C.c_void_p), ("nnz", C.c_size_t),
                ("shape0", C.c_size_t), ("shape1", C.c_size_t), ("sort0", C.c_int), ("mem", C.c_int)]


class Vec(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("val", C.c_void_p), ("nnz", C.c_size_t), ("shape0", C.c_size_t),
                ("sort0", C.c_int), ("mem", C.c_int)]


class Result(C.Structure):
    _fields_ = [("shape0", C.c_uint64), ("shape1", C.c_uint64), ("nnz", C.c_uint64), ("products", C.c_uint64),
                ("nnz_a", C.c_uint64), ("nnz_b", C.c_uint64), ("sum", C.c_double), ("hash", C.c_uint64),
                ("idx0", C.c_void_p), ("idx1", C.c_void_p), ("val", C.c_void_p),
                ("row_nnz", C.c_void_p), ("row_sum", C.c_void_p),
                ("ms_consolidate", C.c_float), ("ms_symbolic", C.c_float), ("ms_numeric", C.c_float),
                ("ms_total", C.c_float), ("ms_light", C.c_float), ("ms_mid", C.c_float), ("ms_heavy", C.c_float),
                ("ms_dense", C.c_float), ("window", C.c_uint32), ("cells_hash", C.c_uint64), ("cells_dense", C.c_uint64), ("products_dense", C.c_uint64), ("workspace_bytes", C.c_uint64),
                ("rows_light", C.c_uint64), ("rows_mid", C.c_uint64), ("rows_heavy", C.c_uint64),
                ("products_light", C.c_uint64), ("products_mid", C.c_uint64), ("products_heavy", C.c_uint64),
                ("tuples_light", C.c_uint64), ("tuples_mid", C.c_uint64), ("tuples_heavy", C.c_uint64),
                ("ms_tiles", C.c_float), ("ms_direct", C.c_float), ("products_tiles", C.c_uint64), ("products_direct", C.c_uint64),
                ("row_hash", C.c_void_p)]


class DistStats(C.Structure):
    _fields_ = [("block_nnz_a", C.c_uint64), ("panel_tuples", C.c_uint64), ("remote_tuples", C.c_uint64),
                ("sent_tuples", C.c_uint64), ("ms_exchange", C.c_float), ("pad_", C.c_float)]


# spsamd_alltoallv_fn: (user, send**, sendbytes*, recv**, recvbytes*, world, stream)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                           C.POINTER(C.c_size_t), C.c_int, C.c_void_p)

CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_size_t)

# every symbol include/spsparse_amd.h declares
SYMBOLS = ["spsamd_ctx_create", "spsamd_ctx_destroy", "spsamd_last_error", "spsamd_ctx_reserve", "spsamd_version", "spsamd_ctx_set_tuning",
           "spsamd_multiply", "spsamd_multiply_mv", "spsamd_result_fetch", "spsamd_result_scatter_dense", "spsamd_memcpy", "spsamd_consolidate", "spsamd_sorted_permutation",
           "spsamd_dim_beginnings", "spsamd_gen_rmat",
           "spsamd_gen_random_rows", "spsamd_gen_poisson2d", "spsamd_gen_laplace3d", "spsamd_gen_aggregation3d",
           "spsamd_dist_unique_id", "spsamd_dist_create", "spsamd_dist_destroy", "spsamd_dist_multiply",
           "spsamd_operand_prepare", "spsamd_operand_as_coo", "spsamd_operand_bytes", "spsamd_operand_destroy"]

_lib = None


def load():
    """dlopen the library and declare prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libspsparse_amd.so is not built: run `python -m spsparse_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    if os.environ.get("SPSAMD_NO_TORCH") != "1":
        # PyTorch wheels bundle their own HIP runtime under the same SONAME
        # (libamdhip64.so.7).  Two runtimes in one process cannot both own the
        # GPU, so when torch is installed it is imported FIRST: the loader then
        # binds this library to the runtime torch already brought in.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.spsamd_ctx_create.argtypes = [P(C.c_void_p), C.c_int, C.c_void_p]
    L.spsamd_ctx_destroy.argtypes = [C.c_void_p]
    L.spsamd_ctx_destroy.restype = None
    L.spsamd_last_error.argtypes = [C.c_void_p]
    L.spsamd_last_error.restype = C.c_char_p
    L.spsamd_ctx_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.spsamd_version.restype = C.c_char_p
    L.spsamd_ctx_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    L.spsamd_multiply.argtypes = [C.c_void_p, C.c_double, P(Vec), P(Coo), C.c_char, P(Vec), P(Coo), C.c_char, P(Vec),
                                  C.c_int, C.c_int, C.c_int, C.c_int, P(Result)]
    L.spsamd_multiply_mv.argtypes = [C.c_void_p, C.c_double, P(Vec), P(Coo), C.c_char, P(Vec), P(Vec),
                                     C.c_int, C.c_int, C.c_int, C.c_int, P(Result)]
    L.spsamd_result_fetch.argtypes = [C.c_void_p, P(Result), CHUNK_FN, C.c_void_p]
    L.spsamd_result_scatter_dense.argtypes = [C.c_void_p, P(Result), C.c_void_p, C.c_size_t, C.c_int]
    L.spsamd_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.spsamd_consolidate.argtypes = [C.c_void_p, P(Coo), C.c_int, C.c_int, C.c_int, P(Result)]
    L.spsamd_sorted_permutation.argtypes = [C.c_void_p, P(Coo), C.c_int, C.c_void_p]
    L.spsamd_dim_beginnings.argtypes = [C.c_void_p, P(Coo), C.c_int, C.c_void_p, P(C.c_size_t)]
    L.spsamd_gen_rmat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    L.spsamd_gen_random_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("spsamd_gen_poisson2d", "spsamd_gen_laplace3d", "spsamd_gen_aggregation3d"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.spsamd_dist_unique_id.argtypes = [C.c_char_p]
    L.spsamd_dist_create.argtypes = [P(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p, ALLTOALLV_FN, C.c_void_p]
    L.spsamd_dist_destroy.argtypes = [C.c_void_p]
    L.spsamd_dist_destroy.restype = None
    L.spsamd_dist_multiply.argtypes = [C.c_void_p, C.c_double, P(Vec), P(Coo), C.c_char, P(Vec), P(Coo), C.c_char, P(Vec), C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, P(Result), P(DistStats)]
    L.spsamd_operand_prepare.argtypes = [C.c_void_p, P(Coo), C.c_char, C.c_int, C.c_int, C.c_int, P(C.c_void_p)]
    L.spsamd_operand_as_coo.argtypes = [C.c_void_p, P(Coo)]
    L.spsamd_operand_bytes.argtypes = [C.c_void_p]
    L.spsamd_operand_bytes.restype = C.c_uint64
    L.spsamd_operand_destroy.argtypes = [C.c_void_p]
    L.spsamd_operand_destroy.restype = None
    _lib = L
    return L


def host_coo(idx0, idx1, val, shape, sort0=-1):
    """Coo struct over numpy arrays; returns (struct, keepalive)."""
    a0 = np.ascontiguousarray(idx0, dtype=np.int32)
    a1 = np.ascontiguousarray(idx1, dtype=np.int32)
    av = np.ascontiguousarray(val, dtype=np.float64)
    assert a0.shape == a1.shape == av.shape
    return Coo(a0.ctypes.data, a1.ctypes.data, av.ctypes.data, av.size, int(shape[0]), int(shape[1]), sort0, MEM_HOST), (a0, a1, av)


def device_coo(ptr0, ptr1, ptrv, nnz, shape, sort0=-1):
    return Coo(ptr0, ptr1, ptrv, nnz, int(shape[0]), int(shape[1]), sort0, MEM_DEVICE)


def result_operand(res):
    """A SINK_COO result as the device operand of the NEXT call on the same context (read in place:
    sorted row-major, each index once -- sort0 = 0, so it is trusted like Consolidate<> does)."""
    return Coo(res.idx0, res.idx1, res.val, int(res.nnz), int(res.shape0), int(res.shape1), 0, MEM_DEVICE)


def host_vec(idx, val, shape0, sort0=-1):
    a = np.ascontiguousarray(idx, dtype=np.int32)
    v = np.ascontiguousarray(val, dtype=np.float64)
    return Vec(a.ctypes.data, v.ctypes.data, v.size, int(shape0), sort0, MEM_HOST), (a, v)


class Context:
    """One device + stream + workspace (spsamd_ctx)."""

    def __init__(self, device=-1, stream=None):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.spsamd_ctx_create(C.byref(h), device, stream)
        if rc != 0:
            raise SpsamdError(rc, "spsamd_ctx_create failed (no MI355X visible?)")
        self.h = h

    def close(self):
        if self.h:
            self.L.spsamd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SpsamdError(rc, self.L.spsamd_last_error(self.h).decode())

    def set_tuning(self, name, value):
        """Developer knob (spsamd_ctx_set_tuning): selects between equivalent kernels, never changes a result."""
        self._check(self.L.spsamd_ctx_set_tuning(self.h, name.encode(), int(value)))

    def reserve(self, workspace_bytes=0, output_tuples=0):
        self._check(self.L.spsamd_ctx_reserve(self.h, workspace_bytes, output_tuples))

    def multiply(self, A, B, C_=1.0, scalei=None, tA='.', scalej=None, tB='.', scalek=None,
                 duplicate_policy=ADD, zero_nan=False, sink=SINK_COO, flags=0):
        """spsamd_multiply.  A, B: Coo structs; scale*: Vec structs or None."""
        res = Result()
        ptr = [None if s is None else C.byref(s) for s in (scalei, scalej, scalek)]
        rc = self.L.spsamd_multiply(self.h, float(C_), ptr[0], C.byref(A), tA.encode(), ptr[1], C.byref(B),
                                    tB.encode(), ptr[2], duplicate_policy, int(zero_nan), sink, flags, C.byref(res))
        self._check(rc)
        return res

    def multiply_mv(self, A, V, C_=1.0, scalei=None, tA='.', scalej=None, duplicate_policy=ADD, zero_nan=False,
                    sink=SINK_COO, flags=0):
        """spsamd_multiply_mv.  A: Coo struct; V, scale*: Vec structs."""
        res = Result()
        ptr = [None if s is None else C.byref(s) for s in (scalei, scalej)]
        rc = self.L.spsamd_multiply_mv(self.h, float(C_), ptr[0], C.byref(A), tA.encode(), ptr[1], C.byref(V),
                                       duplicate_policy, int(zero_nan), sink, flags, C.byref(res))
        self._check(rc)
        return res

    def consolidate(self, A, so0, duplicate_policy=ADD, zero_nan=False):
        res = Result()
        self._check(self.L.spsamd_consolidate(self.h, C.byref(A), so0, duplicate_policy, int(zero_nan), C.byref(res)))
        return res

    def sorted_permutation(self, A, so0):
        perm = np.zeros(A.nnz, dtype=np.uint64)
        self._check(self.L.spsamd_sorted_permutation(self.h, C.byref(A), so0, perm.ctypes.data))
        return perm

    def dim_beginnings(self, A, so0):
        out = np.zeros(A.nnz + 1, dtype=np.uint64)
        cnt = C.c_size_t(0)
        self._check(self.L.spsamd_dim_beginnings(self.h, C.byref(A), so0, out.ctypes.data, C.byref(cnt)))
        return out[:cnt.value]

    def fetch(self, res):
        """Host copy of a SINK_COO result through spsamd_result_fetch: (i, j, v)."""
        n = int(res.nnz)
        oi, oj, ov = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        pos = [0]

        def cb(_user, pi, pj, pv, cnt):
            o = pos[0]
            oi[o:o + cnt] = np.ctypeslib.as_array(pi, shape=(cnt,))
            if pj:                                                      # NULL for a rank-1 (MV) result
                oj[o:o + cnt] = np.ctypeslib.as_array(pj, shape=(cnt,))
            else:
                oj[o:o + cnt] = 0
            ov[o:o + cnt] = np.ctypeslib.as_array(pv, shape=(cnt,))
            pos[0] = o + cnt
            return 0

        self._check(self.L.spsamd_result_fetch(self.h, C.byref(res), CHUNK_FN(cb), None))
        assert pos[0] == n
        return oi, oj, ov

    def scatter_dense(self, res, dense_ptr, ld, duplicate_policy=ADD):
        """DenseAccum on the device: dense[i*ld + j] (+)= v for the tuples of a SINK_COO result."""
        self._check(self.L.spsamd_result_scatter_dense(self.h, C.byref(res), dense_ptr, ld, duplicate_policy))

    def to_host(self, dev_ptr, count, dtype):
        """numpy copy of `count` elements of device memory (spsamd_memcpy)."""
        out = np.empty(count, dtype=dtype)
        self._check(self.L.spsamd_memcpy(self.h, out.ctypes.data, dev_ptr, out.nbytes))
        return out

    def memcpy(self, dst_ptr, src_ptr, nbytes):
        self._check(self.L.spsamd_memcpy(self.h, dst_ptr, src_ptr, nbytes))

    # ---- device generators (outputs: caller-owned device pointers)
    def gen_rmat(self, scale, seed, first_edge, n_edges, p0, p1, pv, edge_factor=16):
        self._check(self.L.spsamd_gen_rmat(self.h, scale, edge_factor, seed, first_edge, n_edges, p0, p1, pv))

    def gen_random_rows(self, n, per_row, seed, stream_base, p0, p1, pv):
        self._check(self.L.spsamd_gen_random_rows(self.h, n, per_row, seed, stream_base, p0, p1, pv))

    def gen_poisson2d(self, N, p0, p1, pv):
        self._check(self.L.spsamd_gen_poisson2d(self.h, N, p0, p1, pv))

    def gen_laplace3d(self, N, p0, p1, pv):
        self._check(self.L.spsamd_gen_laplace3d(self.h, N, p0, p1, pv))

    def gen_aggregation3d(self, N, p0, p1, pv):
        self._check(self.L.spsamd_gen_aggregation3d(self.h, N, p0, p1, pv))


class Operand:
    """spsamd_operand: an operand prepared once (consolidated tuples, row structure, and whatever later products build
    from it), accepted by every entry point through `.coo`."""

    def __init__(self, ctx, X, transpose='.', role=AS_A | AS_B, duplicate_policy=ADD, zero_nan=False):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.L.spsamd_operand_prepare(ctx.h, C.byref(X), transpose.encode(), role, duplicate_policy, int(zero_nan), C.byref(h)))
        self.h = h
        self.coo = Coo()
        ctx._check(ctx.L.spsamd_operand_as_coo(self.h, C.byref(self.coo)))

    @property
    def bytes(self):
        return int(self.ctx.L.spsamd_operand_bytes(self.h))

    def close(self):
        if self.h:
            self.ctx.L.spsamd_operand_destroy(self.h)
            self.h = None


class Dist:
    """spsamd_dist: this rank's end of the row-block sharded multiply (one rank per GPU).

    transport=None: the built-in RCCL transport; `unique_id` (128 bytes from Dist.unique_id() on one rank,
    handed to every rank) creates its communicator.  transport=callable(send_ptrs, send_bytes, recv_ptrs,
    recv_bytes, stream) replaces it (the tests route it through gloo on host copies)."""

    @staticmethod
    def unique_id():
        L = load()
        buf = C.create_string_buffer(128)
        rc = L.spsamd_dist_unique_id(buf)
        if rc != 0:
            raise SpsamdError(rc, "spsamd_dist_unique_id failed (librccl not loadable?)")
        return buf.raw

    def __init__(self, ctx, rank, world, unique_id=None, transport=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self._cb = ALLTOALLV_FN()
        if transport is not None:
            def cb(_user, send, sendb, recv, recvb, n, stream):
                try:
                    transport([send[p] for p in range(n)], [sendb[p] for p in range(n)],
                              [recv[p] for p in range(n)], [recvb[p] for p in range(n)], stream)
                    return 0
                except Exception:                      # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLTOALLV_FN(cb)
        h = C.c_void_p()
        rc = ctx.L.spsamd_dist_create(C.byref(h), ctx.h, rank, world, unique_id, None, self._cb, None)
        ctx._check(rc)
        self.h = h

    def multiply(self, A_block, B_block, b_bounds, C_=1.0, scalei=None, tA='.', scalej=None, tB='.', scalek=None,
                 duplicate_policy=ADD, zero_nan=False, sink=SINK_DIGEST, flags=0):
        """spsamd_dist_multiply: spsparse::multiply's arguments, the two matrices block-wise (collective)."""
        res, stats = Result(), DistStats()
        bb = (C.c_uint64 * len(b_bounds))(*[int(x) for x in b_bounds])
        ptr = [None if s is None else C.byref(s) for s in (scalei, scalej, scalek)]
        rc = self.ctx.L.spsamd_dist_multiply(self.h, float(C_), ptr[0], C.byref(A_block), tA.encode(), ptr[1],
                                             None if B_block is None else C.byref(B_block), tB.encode(), ptr[2], bb,
                                             duplicate_policy, int(zero_nan), sink, flags, C.byref(res), C.byref(stats))
        self.ctx._check(rc)
        return res, stats

    def close(self):
        if self.h:
            self.ctx.L.spsamd_dist_destroy(self.h)
            self.h = None
