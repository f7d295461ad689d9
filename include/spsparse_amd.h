/*
 * spsparse_amd.h -- C ABI of the MI355X SpGEMM behind spsparse::multiply().
 *
 * This is the drop-in boundary.  The reference has no FFI today: its boundary
 * is the header-only template spsparse::multiply()
 * (slib/spsparse/multiply_sparse.hpp:138-164).  include/spsparse_amd/multiply.hpp
 * is that template re-stated on top of the entry points below; INTEGRATION.md
 * shows the lines a spsparse maintainer would add to bind them.
 *
 * Conventions
 *   - plain pointers and sizes, no C++ or torch types; never throws.
 *   - every entry point returns 0 or a negative SPSAMD_E* code; the text of the
 *     last error of a context is at spsamd_last_error().  The C++ shim forwards
 *     it to (*spsparse_error)(-1, "%s", msg) (spsparse.hpp:47,54).
 *   - indices int32, values double: the only instantiation the reference tests
 *     (tests/test_multiply_sparse.cpp:90-91).  Counts and offsets are 64 bit:
 *     nnz(C) exceeds 2^31 on BASELINE cfg2.
 *   - operands are borrowed for the duration of a call and never modified
 *     (multiply_sparse.hpp:143,146 take const&).
 *   - one context = one device + one HIP stream + one workspace; calls on
 *     different contexts may run concurrently from different host threads.
 *   - there is no CPU fallback: every compute entry point needs the GPU.
 *
 * Limits (each is an SPSAMD_EINVAL / SPSAMD_ENOMEM with a message, never a wrong result)
 *   - indices int32, fewer than 2^31 tuples per operand (like the reference's int positions,
 *     algorithm.hpp:419); one output row may hold at most 2^32-1 scalar products.
 *   - rows with more than 4096 scalar products ("heavy" rows) are cut along column windows of 8192
 *     columns (16384 above 2^21 columns), at most 2048 of them.  A product with a heavy row and more than
 *     2^25 columns in op(B) is multiplied by column blocks of op(B), 2^25 columns at a time (same result,
 *     slower: every block repeats the work on A and a COO result is assembled by one more pass).
 *   - the heavy-row path keeps dense indices of rows(op(B)) x windows entries in the workspace (10 bytes per
 *     row and window: 1.4 GB for a 2^20-square matrix, 43 GB at 2^23).  Where they would not fit the device the
 *     product also goes by column blocks, narrow enough for them to fit; SPSAMD_ENOMEM only if one window's
 *     share does not.
 */
#ifndef SPSPARSE_AMD_H
#define SPSPARSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define SPSAMD_OK            0
#define SPSAMD_EDIM        (-1)   /* inner dimensions differ (multiply_sparse.hpp:172-174) */
#define SPSAMD_EINVAL      (-2)   /* bad argument (null pointer, index out of bounds, unsorted scale vector ...) */
#define SPSAMD_EHIP        (-3)   /* HIP runtime error (message carries hipGetErrorString) */
#define SPSAMD_ENOMEM      (-4)   /* device or host allocation failed */
#define SPSAMD_ECAPACITY   (-5)   /* caller-supplied output buffer too small */
#define SPSAMD_ENODEVICE   (-6)   /* no usable gfx950 device */
#define SPSAMD_EPEER       (-7)   /* multi-GPU step: another rank of the communicator reported an error (nothing was multiplied) */

/* spsparse::DuplicatePolicy (spsparse.hpp:25-26), same enumerator order */
#define SPSAMD_LEAVE_ALONE 0
#define SPSAMD_ADD         1
#define SPSAMD_REPLACE     2

/* where the pointers of an operand live */
#define SPSAMD_MEM_HOST    0
#define SPSAMD_MEM_DEVICE  1

typedef struct spsamd_ctx spsamd_ctx;

/*
 * A COO matrix as VectorCooArray<int,double,2> stores it
 * (VectorCooArray.hpp:17,22-23,35): one index vector per dimension, one value
 * vector, shape, and sort_order[0] (-1 = unsorted / edit mode, 0 = consolidated
 * row major {0,1}, 1 = consolidated column major {1,0}).  A matching
 * sort order is trusted like Consolidate<> does (algorithm.hpp:360).
 */
typedef struct {
	const int32_t *idx0;
	const int32_t *idx1;
	const double *val;
	size_t nnz;
	size_t shape0, shape1;
	int sort0;
	int mem;             /* SPSAMD_MEM_HOST or SPSAMD_MEM_DEVICE */
} spsamd_coo;

/*
 * A sparse vector (VectorCooArray<int,double,1>): the three diagonal scale
 * operands and the right-hand side of the matrix-vector multiply.  Scale
 * vectors must be strictly ascending in idx: the reference joins them as
 * stored and silently mis-computes otherwise (multiply_sparse.hpp:83-85,
 * 223-226) -- this library rejects them (SPSAMD_EINVAL).
 */
typedef struct {
	const int32_t *idx;
	const double *val;
	size_t nnz;
	size_t shape0;
	int sort0;           /* only read for the MV right-hand side */
	int mem;
} spsamd_vec;

/* ---- sinks: the device side of the Accumulator concept (accum.hpp:12-24) ---- */

#define SPSAMD_SINK_COO       1   /* row-major sorted (i, j, v) tuples in device memory */
#define SPSAMD_SINK_DIGEST    2   /* count + sum + index hash only (ScalarAccumulator analogue, accum.hpp:158-167) */

/* sink flags */
#define SPSAMD_SINK_ROWSTATS  1   /* DIGEST: also fill row_nnz / row_sum / row_hash (length = rows of op(A)) */
#define SPSAMD_SINK_PERMUTE   4   /* COO, matrix result: emit (j, i, v) -- idx0 holds the column, idx1 the row, shape
                                   * swapped (PermuteAccum with perm {1,0}, accum.hpp:73-101; the tuples stay in
                                   * the order of C's rows, i.e. column-major for the permuted array) */
#define SPSAMD_SINK_EXACT_PATTERN 8 /* the reference's INDEX SET at arrival-order speed: a sum that could be exactly
                                   * zero in the reference's ascending-k order but not in arrival order (or the reverse) --
                                   * |sum| within the rounding bound of its cell -- is re-evaluated in ascending k and that
                                   * value decides (multiply_sparse.hpp:238) and is emitted; all other values stay within
                                   * rounding (1e-12 relative) of the reference's */
#define SPSAMD_SINK_ORDERED   2   /* every sum accumulated in ascending k like the reference's loop
                                   * (multiply_sparse.hpp:219-236): bit-identical values and zero drops on
                                   * any input, several times slower on rows with more than 64 products */

/*
 * Result of one multiply.  For SINK_COO the three arrays live in one of the
 * context's two output buffers (device memory) and stay valid until the next
 * SINK_COO multiply / consolidate or spsamd_ctx_destroy on that context;
 * tuples are in ascending (i, j), each (i, j) at most once, exact zeros dropped
 * (multiply_sparse.hpp:238).
 * Chaining: the arrays may be handed straight back as a SPSAMD_MEM_DEVICE,
 * sort0 = 0 operand of the NEXT call on the same context (T = R*A, then
 * C = T*R^T): that call reads them in place and writes its own result to the
 * other buffer, so no copy and no re-consolidation of T takes place; T stays
 * valid until the call after that.
 */
typedef struct {
	uint64_t shape0, shape1;      /* ret.set_shape(), multiply_sparse.hpp:169 */
	uint64_t nnz;                 /* tuples emitted */
	uint64_t products;            /* P = sum over A tuples of the B row length */
	uint64_t nnz_a, nnz_b;        /* consolidated operand sizes */
	double sum;                   /* DIGEST: sum of emitted values */
	uint64_t hash;                /* DIGEST: sum of mix64(i, j) mod 2^64 */
	const int32_t *idx0;          /* COO: device pointers */
	const int32_t *idx1;
	const double *val;
	const int64_t *row_nnz;       /* DIGEST|ROWSTATS: device pointers */
	const double *row_sum;
	/* timing of the device pipeline stages, milliseconds (HIP events) */
	float ms_consolidate, ms_symbolic, ms_numeric, ms_total;
	/* numeric kernels by row class (P_r = products of the output row):
	 * light P_r <= 64, mid <= 4096, heavy above */
	float ms_light, ms_mid, ms_heavy;
	float ms_dense;               /* part of ms_heavy spent in the dense-window kernel */
	uint32_t window;              /* column-window width the heavy rows were cut with (8192 / 16384; 0 = no heavy row) */
	uint64_t cells_hash, cells_dense;   /* heavy rows are cut into cells: LDS-hash cells and dense-window cells */
	uint64_t products_dense;            /* products of the dense-window cells (part of products_heavy) */
	uint64_t workspace_bytes;           /* device workspace this call carved from the context's arena */
	uint64_t rows_light, rows_mid, rows_heavy;
	uint64_t products_light, products_mid, products_heavy;
	uint64_t tuples_light, tuples_mid, tuples_heavy;      /* A tuples in the rows of each class */
	/* the heavy rows' hash-class cells by kernel: tiles (rows with <= 256 A tuples: hash / bitmap tiles), direct tiles
	 * (off by default: 0); the rest of ms_heavy - ms_dense is the windowed k_hash of the longer rows */
	float ms_tiles, ms_direct;
	uint64_t products_tiles, products_direct;
	const uint64_t *row_hash;     /* DIGEST|ROWSTATS: per row the sum of mix64(i, j) over its tuples (device pointer) */
} spsamd_result;

/* ---- context ---- */

/* device < 0: current device.  stream: a hipStream_t to run on, or NULL to
 * create a private one.  Fails with SPSAMD_ENODEVICE without a GPU. */
int spsamd_ctx_create(spsamd_ctx **out, int device, void *hip_stream);
void spsamd_ctx_destroy(spsamd_ctx *ctx);
const char *spsamd_last_error(const spsamd_ctx *ctx);
/* pre-size the workspace (bytes); optional, it grows on demand otherwise */
int spsamd_ctx_reserve(spsamd_ctx *ctx, size_t workspace_bytes, size_t output_tuples);
const char *spsamd_version(void);
/* Developer knobs (value 0 = default).  They select between equivalent kernels / cell sizes: the result of a
 * multiply is the same for every setting (tests/test_gpu_parity.py forces each in turn).
 *   window          8192 | 16384      column-window width of the heavy rows (default: 16384 when ncol > 2^21)
 *   cell_cap        64..4096          grouping target of the hash cells (2048)
 *   dense_min       64..4096          a window above this many products is a dense cell (3072 with bitmap tiles, else 2048)
 *   long_dense_min  > 0               the same for rows of more than 256 A tuples (1024 at 8192-column windows)
 *   long_cap        > 0               grouping target of those rows' hash cells (cell_cap)
 *   direct_min      > 0               a window of a tile row above this is a direct cell (off: >= dense_min)
 *   tiles_v1        1 | 2 | 3         hash tiles r01 | hash tiles v2 | bitmap-rank tiles (default: chosen per call)
 *   no_tiles, no_wmajor               1: no tiles | no window-major copy of B
 *   xcd                               0: one cell list for all XCDs | 1: static XCD parts (experiment) | 2: dense cells claimed from XCD parts (default)
 *   emit_path       1 | 2             COO order of a hash cell: LDS radix sort | bitonic network (default: by cell width)
 *   light_path      1                 binned light kernels even where every row is light
 *   light_two_pass  1                 all-light COO sink: count, scan, store (two compute passes) instead of one pass + gather
 *   trace           1                 the symbolic phase prints its choices (tile scheme, cell counts) to stderr
 *   index_budget_mb > 0               cap of the heavy rows' window indices (default: 80 % of the free device memory);
 *                                     beyond it the product goes by column blocks of op(B)
 * The environment variables of the same purpose (SPSAMD_W ...) are read once, inside spsamd_ctx_create; nothing reads
 * the environment later.  Unknown names: SPSAMD_EINVAL. */
int spsamd_ctx_set_tuning(spsamd_ctx *ctx, const char *name, long value);

/*
 * ret = C * diag(scalei) * op(A) * diag(scalej) * op(B) * diag(scalek)
 * -- spsparse::multiply, matrix x matrix (multiply_sparse.hpp:152-248).
 * Argument order and meaning follow the reference; scale pointers may be NULL;
 * only the exact character 'T' transposes (multiply_sparse.hpp:167-168).
 * Operands may be host or device resident (per-operand `mem`).  The result
 * goes to the device sink named by sink_kind; host callers then use
 * spsamd_result_fetch() to stream it out in order.
 */
int spsamd_multiply(spsamd_ctx *ctx, double C,
	const spsamd_vec *scalei,
	const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej,
	const spsamd_coo *B, char transpose_B,
	const spsamd_vec *scalek,
	int duplicate_policy, int zero_nan,
	int sink_kind, int sink_flags,
	spsamd_result *result);

/*
 * ret = C * diag(scalei) * op(A) * diag(scalej) * V
 * -- spsparse::multiply, matrix x sparse vector (multiply_sparse.hpp:281-365).
 * V is consolidated with sort order {0} (:313).  The result is rank 1:
 * result->idx0 holds the row indices, idx1 is NULL (spsamd_result_fetch then
 * passes j = NULL to the callback), shape1 is 0.
 */
int spsamd_multiply_mv(spsamd_ctx *ctx, double C,
	const spsamd_vec *scalei,
	const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej,
	const spsamd_vec *V,
	int duplicate_policy, int zero_nan,
	int sink_kind, int sink_flags,
	spsamd_result *result);

/*
 * Host delivery of the last SINK_COO result of ctx: calls cb(user, i, j, v, n)
 * with consecutive chunks (host pointers, valid during the call) in ascending
 * (i, j) order -- the shim's callback loops ret.add({i,j}, v)
 * (multiply_sparse.hpp:242).  A non-zero return of cb stops the delivery and
 * is returned.
 */
typedef int (*spsamd_chunk_fn)(void *user, const int32_t *i, const int32_t *j,
	const double *v, size_t n);
int spsamd_result_fetch(spsamd_ctx *ctx, const spsamd_result *result,
	spsamd_chunk_fn cb, void *user);

/*
 * DenseAccum (accum.hpp:110-140) on the device: apply the tuples of a SINK_COO
 * result to a row-major dense matrix in device memory,
 *     dense[i * ld + j]  (op)=  v        per duplicate_policy
 * ADD sums into the existing entry, REPLACE overwrites it.  LEAVE_ALONE does what
 * the reference's code does, on both sides of the boundary (this entry point and the
 * host mirror in spsparse_amd/multiply.hpp): `if (!std::isnan(oval)) oval = val`
 * (accum.hpp:128-130) -- the entry is overwritten unless it holds a NaN.  SURVEY
 * Appendix A.12 notes that this looks inverted against the policy's documented
 * meaning (spsparse.hpp:19-23); a drop-in keeps the behaviour callers get today.
 */
int spsamd_result_scatter_dense(spsamd_ctx *ctx, const spsamd_result *result, double *dense_device, size_t ld,
	int duplicate_policy);

/* Copy `bytes` between host and/or device memory of this context's device
 * (e.g. result->row_nnz to the host, result->idx0 into a caller's device
 * buffer), ordered after everything queued on the context's stream; returns
 * when the copy is complete. */
int spsamd_memcpy(spsamd_ctx *ctx, void *dst, const void *src, size_t bytes);

/*
 * Stand-alone consolidate (algorithm.hpp:251-319) of a COO matrix on the
 * device: stable sort by sort_order {so0, 1-so0}, zeros dropped, duplicates
 * merged by policy.  Output tuples land in the context's output buffer
 * (result->idx0/idx1/val, nnz); fetch them with spsamd_result_fetch.
 */
int spsamd_consolidate(spsamd_ctx *ctx, const spsamd_coo *A, int so0,
	int duplicate_policy, int zero_nan, spsamd_result *result);

/*
 * sorted_permutation (algorithm.hpp:411-427): the stable permutation that sorts
 * the tuples of A by sort_order {so0, 1-so0}; perm_host receives A->nnz entries.
 * Pinned by tests/test_array.cpp:67-79.
 */
int spsamd_sorted_permutation(spsamd_ctx *ctx, const spsamd_coo *A, int so0, uint64_t *perm_host);

/*
 * dim_beginnings (algorithm.hpp:74-118): offsets at which the leading sorted
 * index changes, plus the end sentinel -- only non-empty rows appear.  A must
 * carry sort0 == so0 (the reference raises "dim_beginnings() required the
 * VectorCooArray is sorted first." otherwise, algorithm.hpp:82-84).
 * beginnings_host needs room for nnz + 1 entries; *count receives the number
 * written (0 for an empty array).  Pinned by tests/test_array.cpp:146-166.
 */
int spsamd_dim_beginnings(spsamd_ctx *ctx, const spsamd_coo *A, int so0, uint64_t *beginnings_host, size_t *count);

/* ---- prepared operands ----
 * What a multiply derives from an operand before it starts -- the consolidated tuples, the row structure, the
 * (col, val)-interleaved copy and, for products with heavy rows, the column-window indices of the right operand -- kept
 * in device memory so that it is derived ONCE for an operand that takes part in many products.  The reference does the
 * same on the host: an array that carries the wanted sort_order is not consolidated again (Consolidate<>,
 * algorithm.hpp:360) and its row structure is cached inside the object (VectorCooArray.hpp:325-335).
 *   role       SPSAMD_AS_A, SPSAMD_AS_B or both: the side(s) of multiply the operand will stand on
 *   transpose  the transpose flag it will be passed with
 *   duplicate_policy, zero_nan: as for multiply; they are applied HERE, once (a multiply takes a prepared operand as it
 *              is, the way Consolidate<> takes a sorted one)
 * spsamd_operand_as_coo fills a spsamd_coo (mem = SPSAMD_MEM_PREPARED) that is accepted wherever an operand is; results
 * are identical to those of the plain operand.  A handle belongs to the context that prepared it and is immutable; the
 * structures only heavy rows need are built by the first multiply that needs them and stay (not re-entrant on one
 * handle from two threads, like the reference's lazy cache).  Used with the other transpose flag than it was prepared
 * for, the handle is read as an ordinary device operand sorted the other way (and consolidated by that call). */
#define SPSAMD_MEM_PREPARED 2
#define SPSAMD_AS_A 1
#define SPSAMD_AS_B 2
typedef struct spsamd_operand spsamd_operand;
int spsamd_operand_prepare(spsamd_ctx *ctx, const spsamd_coo *X, char transpose, int role, int duplicate_policy, int zero_nan,
	spsamd_operand **out);
int spsamd_operand_as_coo(const spsamd_operand *op, spsamd_coo *out);
uint64_t spsamd_operand_bytes(const spsamd_operand *op);     /* device memory the handle holds right now */
void spsamd_operand_destroy(spsamd_operand *op);

/* ---- multi-GPU: op(A) sharded by contiguous row blocks, one rank per GPU ----
 * The reference has no counterpart.  Output row i depends only on row i of op(A) and the op(B) rows
 * {k : op(A)(i,k) != 0} (the reference's own loop structure, multiply_sparse.hpp:192): every rank multiplies its row
 * block of op(A) with the panel of op(B) rows it needs, fetched with ONE exchange step (grouped ncclSend / ncclRecv over
 * RCCL: the all-to-allv of needed B row panels); C stays row partitioned, there is no reduction.
 */
typedef struct spsamd_dist spsamd_dist;

/* Optional transport replacing the built-in RCCL one (tests: gloo / MPI through host memory).  All-to-allv of
 * device buffers: send[p] (sendbytes[p] bytes) goes to rank p, recvbytes[p] bytes from rank p arrive in recv[p].
 * The buffers are complete when it is called and must be complete when it returns.  A step calls it several times
 * (once per kind of payload); every rank makes the same sequence of calls. */
typedef int (*spsamd_alltoallv_fn)(void *user, const void *const *send, const size_t *sendbytes,
	void *const *recv, const size_t *recvbytes, int world, void *hip_stream);

typedef struct {
	uint64_t block_nnz_a;         /* consolidated tuples of this rank's A block */
	uint64_t panel_tuples;        /* tuples of the B panel this rank multiplied with */
	uint64_t remote_tuples;       /* ... of which received from other ranks */
	uint64_t sent_tuples;         /* tuples this rank sent to other ranks */
	float ms_exchange;            /* consolidation of the blocks + both exchange rounds up to the issue of the panel transfer
	                               * (HIP events; the transfer itself overlaps the product's symbolic phase) */
	float pad_;
} spsamd_dist_stats;

/* 128 bytes identifying a new RCCL communicator (ncclGetUniqueId): call on one rank, hand to all (MPI, a file,
 * torch.distributed ...), then spsamd_dist_create(..., unique_id, ...) on every rank. */
int spsamd_dist_unique_id(char id[128]);
/* One of: unique_id (the library creates its communicator with ncclCommInitRank), nccl_comm (an ncclComm_t of
 * the caller, borrowed), or transport (+ transport_user).  ctx: this rank's context (its device and stream).
 * At most 64 ranks. */
int spsamd_dist_create(spsamd_dist **out, spsamd_ctx *ctx, int rank, int world, const char *unique_id,
	void *nccl_comm, spsamd_alltoallv_fn transport, void *transport_user);
void spsamd_dist_destroy(spsamd_dist *d);
/*
 * One step: this rank's rows of  C * diag(scalei) * op(A) * diag(scalej) * op(B) * diag(scalek)  -- the arguments of
 * spsparse::multiply (multiply_sparse.hpp:138-150) with the two matrices given block-wise:
 *   A_block  the tuples of A that belong to this rank's rows of op(A) -- raw COO with GLOBAL indices, shape = the whole
 *            matrix's, in their original relative order (which rows a rank owns is the caller's choice: any partition).
 *   B_block  the tuples of B whose op(B) ROW -- the inner index: idx0 without 'T', idx1 with it -- lies in
 *            [b_bounds[rank], b_bounds[rank+1]); or NULL where B is A with the same transpose flag and the A blocks
 *            are cut at b_bounds too (A * A: the own A block then is the own B block).
 *   b_bounds world + 1 ascending boundaries of op(B)'s distribution over the inner dimension (0 .. inner), the same on
 *            every rank.
 *   scale vectors: whole, the same on every rank.  zero_nan: the NaNs dropped are those of the leading run of the WHOLE
 *            matrix' sorted sequence, as in the reference (algorithm.hpp:272-275): the ranks agree on it first.
 * The result is this rank's rows of C in the sink of its context (digest: add the counts / hashes / sums of all ranks;
 * COO: tuples with global indices, chainable as the A_block of the next step: T = R*A, then C = T*R^T).
 * Collective: every rank of the communicator must call it, with the same shapes, bounds, flags and policies.  A rank
 * whose own operands are bad (index out of bounds, a B_block tuple outside its bounds ...) still takes part in the first
 * exchange round, which carries every rank's status: then EVERY rank returns an error -- its own, or SPSAMD_EPEER -- and
 * the communicator stays usable.  A failure after that round (out of memory, a HIP / RCCL error) cannot be agreed on any
 * more: the communicator refuses further steps (SPSAMD_EPEER); destroy it and create a new one.
 */
int spsamd_dist_multiply(spsamd_dist *d, double C,
	const spsamd_vec *scalei,
	const spsamd_coo *A_block, char transpose_A,
	const spsamd_vec *scalej,
	const spsamd_coo *B_block, char transpose_B,
	const spsamd_vec *scalek,
	const uint64_t *b_bounds,
	int duplicate_policy, int zero_nan, int sink_kind, int sink_flags,
	spsamd_result *result, spsamd_dist_stats *stats);

/* ---- synthetic operands generated on the device (bench / tests) ----
 * Bit-identical to spsparse_amd/workloads.py.  Outputs are device arrays
 * owned by the caller (capacity >= the generator's tuple count). */
int spsamd_gen_rmat(spsamd_ctx *ctx, int scale, int edge_factor, uint64_t seed,
	uint64_t first_edge, uint64_t n_edges, int32_t *idx0, int32_t *idx1, double *val);
int spsamd_gen_random_rows(spsamd_ctx *ctx, uint64_t n, uint64_t per_row, uint64_t seed,
	uint64_t stream_base, int32_t *idx0, int32_t *idx1, double *val);
/* 5-point Poisson on an N x N grid: writes 5N^2-4N tuples, row-major sorted */
int spsamd_gen_poisson2d(spsamd_ctx *ctx, uint64_t N, int32_t *idx0, int32_t *idx1, double *val);
/* 7-point Laplacian on an N^3 grid: writes 7N^3-6N^2 tuples, row-major sorted */
int spsamd_gen_laplace3d(spsamd_ctx *ctx, uint64_t N, int32_t *idx0, int32_t *idx1, double *val);
/* 2x2x2 piecewise-constant aggregation, (N/2)^3 x N^3: writes N^3 tuples */
int spsamd_gen_aggregation3d(spsamd_ctx *ctx, uint64_t N, int32_t *idx0, int32_t *idx1, double *val);

#ifdef __cplusplus
}
#endif
#endif
