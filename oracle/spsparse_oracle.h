/*
 * spsparse_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the citibeth/spsparse multiply() path.  It is
 * the checker for the HIP product path and the "port" CPU baseline in bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing under spsparse_amd/ or include/ links or calls it.
 *
 * Parity status: PINNED by the reference's own test vectors
 * (tests/test_array.cpp:67-79,135-168, tests/test_xiter.cpp:52-125,
 * tests/test_multiply_sparse.cpp:45-78 known answer, :84-136 and :138-203
 * property tests restated in tests/test_oracle_pins.py).  The reference itself
 * is NOT built here: it needs the un-vendored ibmisc + blitz headers, and
 * building it against stand-in headers is not allowed, so there is no
 * oracle/_ref.
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * restates.  Index type int32, value type double, like every reference
 * instantiation (tests/test_multiply_sparse.cpp:90-91).
 */
#ifndef SPSPARSE_ORACLE_H
#define SPSPARSE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* spsparse.hpp:25-26 -- enum class DuplicatePolicy {LEAVE_ALONE, ADD, REPLACE} */
enum { ORC_LEAVE_ALONE = 0, ORC_ADD = 1, ORC_REPLACE = 2 };

/* Growable COO output ("the Accumulator": VectorCooArray::add appends,
 * VectorCooArray.hpp:238-266).  rank 1 outputs leave j == NULL. */
typedef struct {
	int32_t *i;
	int32_t *j;
	double *v;
	size_t n, cap;
	size_t shape0, shape1;
	int rank;
} orc_coo;

void orc_coo_init(orc_coo *c, int rank);
void orc_coo_free(orc_coo *c);

/* spsparse.hpp:95-103 */
int orc_isnone(double n, int zero_nan);

/* algorithm.hpp:411-427 (+ CmpIndex :375-396).  so0 = sort_order[0]
 * (0 = {0,1} row major, 1 = {1,0} col major; rank 1: 0).  Stable. */
void orc_sorted_permutation(int rank, const int32_t *idx0, const int32_t *idx1,
	size_t n, int so0, size_t *perm);

/* algorithm.hpp:251-319.  Outputs need capacity n.  Returns the number of
 * tuples written. */
size_t orc_consolidate(int rank, const int32_t *idx0, const int32_t *idx1,
	const double *val, size_t n, int so0, int duplicate_policy, int zero_nan,
	int32_t *out0, int32_t *out1, double *outv);

/* algorithm.hpp:74-118.  lead = the leading sorted index of each tuple.
 * out needs capacity n+1; returns the number of entries (distinct + sentinel,
 * or 0 for n == 0). */
size_t orc_dim_beginnings(const int32_t *lead, size_t n, size_t *out);

/* xiter.hpp:236-278 / :149-194 with next_noincr_body.hpp:1-53: the values at
 * which all streams match.  out needs capacity min(n*). */
size_t orc_join2(const int32_t *a, size_t na, const int32_t *b, size_t nb,
	int32_t *out);
size_t orc_join3(const int32_t *a, size_t na, const int32_t *b, size_t nb,
	const int32_t *c, size_t nc, int32_t *out);

/* Operands.  sort0 = VectorCooArray::sort_order[0] (-1 unsorted/edit mode). */
typedef struct {
	const int32_t *idx0, *idx1;
	const double *val;
	size_t nnz, shape0, shape1;
	int sort0;
} orc_mat;
typedef struct {
	const int32_t *idx;
	const double *val;
	size_t nnz, shape0;
	int sort0;
} orc_vec;

/* multiply_sparse.hpp:152-248, same algorithm (inner product over sorted
 * rows x sorted columns, leap-frog joins).  Returns 0, or -1 with msg filled
 * for the inner-dimension error (:172-174).  Appends to ret; sets its shape
 * first (:169). */
int orc_multiply_mm(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, char *msg, size_t msglen);

/* Same results, row-wise (Gustavson) with the same ascending-k summation per
 * output element -- the scalable checker.  nthreads > 1 splits output rows
 * over OpenMP threads (results identical). */
int orc_multiply_mm_rowwise(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, int nthreads, char *msg, size_t msglen);

/* The same checker as a STREAMING DIGEST (nothing is stored: BASELINE cfg2's product has 1e10 tuples): count,
 * sum of the emitted values, sum of mix64(i, j) mod 2^64 -- the figures the device's digest sink reports -- the
 * number of scalar products, and optionally per output row the tuple count and the row's own index hash
 * (arrays of rows(op(A)) entries, zeroed by the caller; only evaluated rows are written).  Rows are handed to
 * the threads dynamically.  row_mask (rows(op(A)) bytes) or NULL: evaluate only the rows whose byte is non-zero
 * (a sample of a product too large to evaluate whole).  What it restates: the per-cell test of
 * tests/test_multiply_sparse.cpp:119-128 at sizes where the cells cannot be held. */
typedef struct {
	uint64_t count, hash, products, nnz_a, nnz_b;
	double sum;
	int64_t *row_nnz;       /* in: NULL or rows(op(A)) entries */
	uint64_t *row_hash;     /* in: NULL or rows(op(A)) entries */
} orc_digest_out;
int orc_multiply_mm_rowwise_digest(orc_digest_out *out, const uint8_t *row_mask, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, int nthreads, char *msg, size_t msglen);

/* orc_sorted_permutation through the merge sort alone (large inputs otherwise take a stable LSD radix sort
 * that yields the same permutation; tests compare the two). */
void orc_sorted_permutation_merge(int rank, const int32_t *idx0, const int32_t *idx1,
	size_t n, int so0, size_t *perm);

/* multiply_sparse.hpp:281-365 (matrix x sparse vector). */
int orc_multiply_mv(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_vec *V, int duplicate_policy, int zero_nan,
	char *msg, size_t msglen);

/* Order-independent digest of a COO result, the same function the device
 * checksum sink computes: count, sum of values, sum of mix64(i,j) mod 2^64. */
uint64_t orc_mix64(uint32_t i, uint32_t j);
void orc_digest(const orc_coo *c, uint64_t *count, double *sum, uint64_t *hash);

#ifdef __cplusplus
}
#endif
#endif
