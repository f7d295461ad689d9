/*
 * spsparse_oracle.c -- TEST INFRASTRUCTURE ONLY (see spsparse_oracle.h).
 *
 * CPU restatement of the spsparse multiply() path in plain C99.  Compile with
 * -ffp-contract=off: the reference (g++ -O2, x86-64 baseline) evaluates
 * `sum += a*b` as a rounded multiply followed by a rounded add.
 *
 * Parity: pinned by the reference's own test vectors, see the header.
 */
#include "spsparse_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* small helpers                                                       */

static void *xmalloc(size_t n)
{
	void *p = malloc(n ? n : 1);
	if (!p) { fprintf(stderr, "spsparse_oracle: out of memory\n"); abort(); }
	return p;
}

void orc_coo_init(orc_coo *c, int rank)
{
	memset(c, 0, sizeof(*c));
	c->rank = rank;
}

void orc_coo_free(orc_coo *c)
{
	free(c->i); free(c->j); free(c->v);
	c->i = c->j = NULL; c->v = NULL; c->n = c->cap = 0;
}

/* VectorCooArray::add (VectorCooArray.hpp:238-266): RANK+1 push_backs. */
static void coo_add(orc_coo *c, int32_t i, int32_t j, double v)
{
	if (c->n == c->cap) {
		size_t cap = c->cap ? c->cap * 2 : 1024;
		c->i = (int32_t *)realloc(c->i, cap * sizeof(int32_t));
		if (c->rank == 2) c->j = (int32_t *)realloc(c->j, cap * sizeof(int32_t));
		c->v = (double *)realloc(c->v, cap * sizeof(double));
		if (!c->i || !c->v || (c->rank == 2 && !c->j)) abort();
		c->cap = cap;
	}
	c->i[c->n] = i;
	if (c->rank == 2) c->j[c->n] = j;
	c->v[c->n] = v;
	c->n++;
}

/* spsparse.hpp:95-103 */
int orc_isnone(double n, int zero_nan)
{
	if (zero_nan) return isnan(n) || (n == 0);
	return (n == 0);
}

/* ------------------------------------------------------------------ */
/* sorted_permutation: algorithm.hpp:411-427, CmpIndex :375-396        */

typedef struct {
	const int32_t *d[2];   /* d[0] = most significant dimension */
	int rank;
} cmp_ctx;

static int cmp_less(const cmp_ctx *c, size_t i, size_t j)
{
	for (int k = 0; k < c->rank - 1; ++k) {
		if (c->d[k][i] < c->d[k][j]) return 1;
		if (c->d[k][i] > c->d[k][j]) return 0;
	}
	return c->d[c->rank - 1][i] < c->d[c->rank - 1][j];
}

/* std::stable_sort restated as a bottom-up merge sort (any stable sort gives
 * the same permutation). */
static void stable_sort_perm(const cmp_ctx *c, size_t *perm, size_t n)
{
	size_t *tmp = (size_t *)xmalloc(n * sizeof(size_t));
	size_t *src = perm, *dst = tmp;
	for (size_t w = 1; w < n; w *= 2) {
		for (size_t lo = 0; lo < n; lo += 2 * w) {
			size_t mid = lo + w < n ? lo + w : n;
			size_t hi = lo + 2 * w < n ? lo + 2 * w : n;
			size_t a = lo, b = mid, o = lo;
			while (a < mid && b < hi) {
				/* take from the right run only if strictly less: stable */
				if (cmp_less(c, src[b], src[a])) dst[o++] = src[b++];
				else dst[o++] = src[a++];
			}
			while (a < mid) dst[o++] = src[a++];
			while (b < hi) dst[o++] = src[b++];
		}
		size_t *t = src; src = dst; dst = t;
	}
	if (src != perm) memcpy(perm, src, n * sizeof(size_t));
	free(tmp);
}

/* Large inputs (the BASELINE-size checks: 1.3e8 tuples at R-MAT scale 23): the same permutation from a
 * stable LSD radix sort on the key (d[0] << 32 | d[1]) -- any stable sort of the same order gives the
 * permutation std::stable_sort gives.  Non-negative indices only (returns 0 otherwise: merge sort). */
#define ORC_RADIX_MIN ((size_t)1 << 16)

static int radix_sort_perm(const cmp_ctx *c, size_t *perm, size_t n)
{
	uint64_t *k0 = (uint64_t *)malloc(n * sizeof(uint64_t)), *k1 = (uint64_t *)malloc(n * sizeof(uint64_t));
	uint32_t *p0 = (uint32_t *)malloc(n * sizeof(uint32_t)), *p1 = (uint32_t *)malloc(n * sizeof(uint32_t));
	int ok = k0 && k1 && p0 && p1 && n < ((size_t)1 << 32);
	uint64_t all = 0;
	for (size_t i = 0; ok && i < n; ++i) {
		int32_t hi = c->rank == 2 ? c->d[0][i] : 0, lo = c->rank == 2 ? c->d[1][i] : c->d[0][i];
		if (hi < 0 || lo < 0) { ok = 0; break; }
		k0[i] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
		p0[i] = (uint32_t)i;
		all |= k0[i];
	}
	if (ok) {
		size_t *cnt = (size_t *)xmalloc(65536 * sizeof(size_t));
		for (int shift = 0; shift < 64; shift += 16) {
			if (((all >> shift) & 0xFFFF) == 0) continue;      /* a digit that is 0 everywhere moves nothing */
			memset(cnt, 0, 65536 * sizeof(size_t));
			for (size_t i = 0; i < n; ++i) cnt[(k0[i] >> shift) & 0xFFFF]++;
			size_t run = 0;
			for (size_t d = 0; d < 65536; ++d) { size_t t = cnt[d]; cnt[d] = run; run += t; }
			for (size_t i = 0; i < n; ++i) {
				size_t o = cnt[(k0[i] >> shift) & 0xFFFF]++;
				k1[o] = k0[i]; p1[o] = p0[i];
			}
			uint64_t *tk = k0; k0 = k1; k1 = tk;
			uint32_t *tp = p0; p0 = p1; p1 = tp;
		}
		free(cnt);
		for (size_t i = 0; i < n; ++i) perm[i] = p0[i];
	}
	free(k0); free(k1); free(p0); free(p1);
	return ok;
}

void orc_sorted_permutation(int rank, const int32_t *idx0, const int32_t *idx1,
	size_t n, int so0, size_t *perm)
{
	cmp_ctx c;
	c.rank = rank;
	if (rank == 1) { c.d[0] = idx0; c.d[1] = NULL; }
	else if (so0 == 0) { c.d[0] = idx0; c.d[1] = idx1; }
	else { c.d[0] = idx1; c.d[1] = idx0; }
	for (size_t i = 0; i < n; ++i) perm[i] = i;     /* algorithm.hpp:419-421 */
	if (n >= ORC_RADIX_MIN && radix_sort_perm(&c, perm, n)) return;
	stable_sort_perm(&c, perm, n);                  /* algorithm.hpp:424 */
}

/* the merge sort alone, whatever the size (tests compare the two paths) */
void orc_sorted_permutation_merge(int rank, const int32_t *idx0, const int32_t *idx1,
	size_t n, int so0, size_t *perm)
{
	cmp_ctx c;
	c.rank = rank;
	if (rank == 1) { c.d[0] = idx0; c.d[1] = NULL; }
	else if (so0 == 0) { c.d[0] = idx0; c.d[1] = idx1; }
	else { c.d[0] = idx1; c.d[1] = idx0; }
	for (size_t i = 0; i < n; ++i) perm[i] = i;
	stable_sort_perm(&c, perm, n);
}

/* ------------------------------------------------------------------ */
/* consolidate: algorithm.hpp:251-319                                  */

size_t orc_consolidate(int rank, const int32_t *idx0, const int32_t *idx1,
	const double *val, size_t n, int so0, int duplicate_policy, int zero_nan,
	int32_t *out0, int32_t *out1, double *outv)
{
	size_t nout = 0;
	if (n == 0) return 0;                           /* :263 */

	size_t *perm = (size_t *)xmalloc(n * sizeof(size_t));
	orc_sorted_permutation(rank, idx0, idx1, n, so0, perm);   /* :266 */

	size_t ii = 0;
	/* :272-275 skip the leading run of 0 (and NaN when zero_nan) */
	for (;; ++ii) {
		if (ii == n) goto finished;
		if (!orc_isnone(val[perm[ii]], zero_nan)) break;
	}
	int32_t acc0 = idx0[perm[ii]];                  /* :278-280 */
	int32_t acc1 = rank == 2 ? idx1[perm[ii]] : 0;
	double accv = val[perm[ii]];
	++ii;

	for (;; ++ii) {
		/* :284-292 -- NB isnone() without zero_nan here (Appendix A.1) */
		for (;; ++ii) {
			if (ii == n) {
				out0[nout] = acc0; if (rank == 2) out1[nout] = acc1;
				outv[nout] = accv; ++nout;
				goto finished;
			}
			if (!orc_isnone(val[perm[ii]], 0)) break;
		}
		int32_t n0 = idx0[perm[ii]];
		int32_t n1 = rank == 2 ? idx1[perm[ii]] : 0;
		if (n0 != acc0 || n1 != acc1) {             /* :296-303 */
			out0[nout] = acc0; if (rank == 2) out1[nout] = acc1;
			outv[nout] = accv; ++nout;
			acc0 = n0; acc1 = n1; accv = val[perm[ii]];
			continue;
		}
		/* :307-310 */
		if (duplicate_policy == ORC_ADD) accv += val[perm[ii]];
		else if (duplicate_policy == ORC_REPLACE) accv = val[perm[ii]];
	}
finished:
	free(perm);
	return nout;
}

/* ------------------------------------------------------------------ */
/* dim_beginnings: algorithm.hpp:74-118                                */

size_t orc_dim_beginnings(const int32_t *lead, size_t n, size_t *out)
{
	size_t nb = 0;
	if (n == 0) return 0;                           /* :89 */
	out[nb++] = 0;                                  /* :90 */
	int32_t last_row = lead[0];
	for (size_t ai = 1;; ++ai) {
		if (ai == n) { out[nb++] = ai; break; }     /* sentinel :95-99 */
		if (lead[ai] != last_row) {                 /* :100-104 */
			out[nb++] = ai;
			last_row = lead[ai];
		}
	}
	return nb;
}

/* ------------------------------------------------------------------ */
/* Join2Xiter / Join3Xiter: xiter.hpp:236-278, :149-194;               */
/* next_noincr_body.hpp:1-53                                           */

typedef struct {
	const int32_t *p;
	size_t pos, end;
} stream;

typedef struct {
	stream s[3];
	int rank;
	int64_t next_match;
	int eof;
} joinx;

static void join_next_noincr(joinx *J)
{
	stream *i1 = &J->s[0], *i2 = &J->s[1], *i3 = &J->s[2];
restart_loop:
	for (;; ++i1->pos) {                            /* body :5-15 */
		if (i1->pos == i1->end) { J->eof = 1; return; }
		if (i1->p[i1->pos] == J->next_match) break;
		if (i1->p[i1->pos] > J->next_match) { J->next_match = i1->p[i1->pos]; break; }
	}
	if (J->rank >= 2) {
		for (;; ++i2->pos) {                        /* body :20-31 */
			if (i2->pos == i2->end) { J->eof = 1; return; }
			if (i2->p[i2->pos] == J->next_match) break;
			if (i2->p[i2->pos] > J->next_match) {
				J->next_match = i2->p[i2->pos];
				++i1->pos;
				goto restart_loop;
			}
		}
	}
	if (J->rank >= 3) {
		for (;; ++i3->pos) {                        /* body :36-48 */
			if (i3->pos == i3->end) { J->eof = 1; return; }
			if (i3->p[i3->pos] == J->next_match) break;
			if (i3->p[i3->pos] > J->next_match) {
				J->next_match = i3->p[i3->pos];
				++i1->pos; ++i2->pos;
				goto restart_loop;
			}
		}
	}
}

static void join_init(joinx *J, int rank)
{
	J->rank = rank;
	J->next_match = 0;
	J->eof = (J->s[0].pos == J->s[0].end);          /* xiter.hpp:170,255 */
	if (J->eof) return;
	J->next_match = J->s[0].p[J->s[0].pos];         /* :172,257 */
	join_next_noincr(J);
}

static void join_incr(joinx *J)                     /* xiter.hpp:183-190,269-275 */
{
	for (int k = 0; k < J->rank; ++k) ++J->s[k].pos;
	join_next_noincr(J);
}

static void stream_set(stream *s, const int32_t *p, size_t b, size_t e)
{
	s->p = p; s->pos = b; s->end = e;
}

size_t orc_join2(const int32_t *a, size_t na, const int32_t *b, size_t nb,
	int32_t *out)
{
	joinx J; size_t n = 0;
	stream_set(&J.s[0], a, 0, na);
	stream_set(&J.s[1], b, 0, nb);
	for (join_init(&J, 2); !J.eof; join_incr(&J)) out[n++] = a[J.s[0].pos];
	return n;
}

size_t orc_join3(const int32_t *a, size_t na, const int32_t *b, size_t nb,
	const int32_t *c, size_t nc, int32_t *out)
{
	joinx J; size_t n = 0;
	stream_set(&J.s[0], a, 0, na);
	stream_set(&J.s[1], b, 0, nb);
	stream_set(&J.s[2], c, 0, nc);
	for (join_init(&J, 3); !J.eof; join_incr(&J)) out[n++] = a[J.s[0].pos];
	return n;
}

/* ------------------------------------------------------------------ */
/* Consolidate<> (algorithm.hpp:324-369): private sorted copy unless   */
/* the operand already carries the wanted sort_order (:360).           */

typedef struct {
	int32_t *lead, *minor;       /* owned */
	double *val;
	size_t n;
	size_t *beg;                 /* dim_beginnings, nbeg entries */
	size_t nbeg;
	int32_t *rowid;              /* leading index of each non-empty row */
	size_t nrows;
} conmat;

static void conmat_free(conmat *m)
{
	free(m->lead); free(m->minor); free(m->val); free(m->beg); free(m->rowid);
}

/* so0 = wanted sort_order[0]; afterwards lead = index(so0), minor = the other */
static void conmat_build(conmat *m, const orc_mat *A, int so0, int dup, int zero_nan)
{
	size_t n = A->nnz;
	int32_t *o0 = (int32_t *)xmalloc(n * sizeof(int32_t));
	int32_t *o1 = (int32_t *)xmalloc(n * sizeof(int32_t));
	double *ov = (double *)xmalloc(n * sizeof(double));
	size_t nout;
	if (A->sort0 == so0) {       /* algorithm.hpp:360 -- trusted as is */
		memcpy(o0, A->idx0, n * sizeof(int32_t));
		memcpy(o1, A->idx1, n * sizeof(int32_t));
		memcpy(ov, A->val, n * sizeof(double));
		nout = n;
	} else {
		nout = orc_consolidate(2, A->idx0, A->idx1, A->val, n, so0, dup, zero_nan, o0, o1, ov);
	}
	m->lead = so0 == 0 ? o0 : o1;
	m->minor = so0 == 0 ? o1 : o0;
	m->val = ov;
	m->n = nout;
	m->beg = (size_t *)xmalloc((nout + 1) * sizeof(size_t));
	m->nbeg = orc_dim_beginnings(m->lead, nout, m->beg);
	m->nrows = m->nbeg ? m->nbeg - 1 : 0;
	m->rowid = (int32_t *)xmalloc(m->nrows * sizeof(int32_t));
	for (size_t r = 0; r < m->nrows; ++r) m->rowid[r] = m->lead[m->beg[r]];
}

/* MultXiter family (multiply_sparse.hpp:41-115): cursor over the non-empty
 * leading indices, optionally Join2'ed with a scale vector as stored. */
typedef struct {
	joinx J;
	int scaled;
	const orc_vec *scale;
} multx;

static void multx_init(multx *x, const conmat *m, const orc_vec *scale)
{
	x->scaled = scale != NULL;
	x->scale = scale;
	stream_set(&x->J.s[0], m->rowid, 0, m->nrows);
	if (x->scaled) {
		stream_set(&x->J.s[1], scale->idx, 0, scale->nnz);
		join_init(&x->J, 2);                        /* ScaledMultXiter :83-86 */
	} else {
		x->J.rank = 1;                              /* SimpleMultXiter :60-62 */
		x->J.eof = (m->nrows == 0);
	}
}
static int multx_eof(const multx *x) { return x->J.eof; }
static void multx_incr(multx *x)
{
	if (x->scaled) join_incr(&x->J);
	else { ++x->J.s[0].pos; x->J.eof = (x->J.s[0].pos == x->J.s[0].end); }
}
static size_t multx_row(const multx *x) { return x->J.s[0].pos; }
static double multx_scale_val(const multx *x)
{
	return x->scaled ? x->scale->val[x->J.s[1].pos] : 1.0;   /* :67,:91 */
}

static void set_msg(char *msg, size_t msglen, const char *fmt, long a, long b)
{
	if (msg && msglen) snprintf(msg, msglen, fmt, a, b);
}

/* ------------------------------------------------------------------ */
/* multiply MM: multiply_sparse.hpp:152-248                            */

int orc_multiply_mm(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, char *msg, size_t msglen)
{
	/* :167-168  ROW_MAJOR = {0,1}, COL_MAJOR = {1,0} (spsparse.cpp:30-31) */
	int a0 = transpose_A == 'T' ? 1 : 0, a1 = 1 - a0;
	int b0 = transpose_B == 'T' ? 0 : 1, b1 = 1 - b0;
	size_t ashape[2] = { A->shape0, A->shape1 };
	size_t bshape[2] = { B->shape0, B->shape1 };
	ret->shape0 = ashape[a0];                       /* :169 */
	ret->shape1 = bshape[b0];

	if (ashape[a1] != bshape[b1]) {                 /* :172-174 */
		set_msg(msg, msglen, "Inner dimensions for A (%ld) and B (%ld) must match!",
			(long)ashape[a1], (long)bshape[b1]);
		return -1;
	}
	if (orc_isnone(C, 0)                            /* :178-184 */
		|| (scalei && scalei->nnz == 0) || A->nnz == 0
		|| (scalej && scalej->nnz == 0) || B->nnz == 0
		|| (scalek && scalek->nnz == 0))
		return 0;

	conmat Ac, Bc;                                  /* :187-188 */
	conmat_build(&Ac, A, a0, duplicate_policy, zero_nan);
	conmat_build(&Bc, B, b0, duplicate_policy, zero_nan);
	/* Appendix A.3: an operand whose stored values are all 0 makes the
	 * reference dereference an empty dim_beginnings (UB, SIGSEGV observed).
	 * The restatement returns the empty product instead. */
	if (Ac.nrows == 0 || Bc.nrows == 0) goto done;

	multx ja;
	for (multx_init(&ja, &Ac, scalei); !multx_eof(&ja); multx_incr(&ja)) {   /* :192 */
		double a_scale = multx_scale_val(&ja);
		if (orc_isnone(a_scale, 0)) continue;       /* :195 */
		size_t ra = multx_row(&ja);
		int32_t aix = Ac.rowid[ra];

		multx jb;                                   /* :208 re-created per A row */
		for (multx_init(&jb, &Bc, scalek); !multx_eof(&jb); multx_incr(&jb)) {
			double b_scale = multx_scale_val(&jb);
			if (orc_isnone(b_scale, 0)) continue;   /* :211 */
			size_t rb = multx_row(&jb);
			int32_t bix = Bc.rowid[rb];

			double sum = 0;                         /* :219 */
			joinx ab;
			if (scalej) {                           /* :221-228 */
				stream_set(&ab.s[0], Ac.minor, Ac.beg[ra], Ac.beg[ra + 1]);
				stream_set(&ab.s[1], scalej->idx, 0, scalej->nnz);
				stream_set(&ab.s[2], Bc.minor, Bc.beg[rb], Bc.beg[rb + 1]);
				for (join_init(&ab, 3); !ab.eof; join_incr(&ab))
					sum += Ac.val[ab.s[0].pos] * scalej->val[ab.s[1].pos] * Bc.val[ab.s[2].pos];
			} else {                                /* :229-236 */
				stream_set(&ab.s[0], Ac.minor, Ac.beg[ra], Ac.beg[ra + 1]);
				stream_set(&ab.s[1], Bc.minor, Bc.beg[rb], Bc.beg[rb + 1]);
				for (join_init(&ab, 2); !ab.eof; join_incr(&ab))
					sum += Ac.val[ab.s[0].pos] * Bc.val[ab.s[1].pos];
			}
			if (!orc_isnone(sum, 0))                /* :238-243 */
				coo_add(ret, aix, bix, sum * C * a_scale * b_scale);
		}
	}
done:
	conmat_free(&Ac);
	conmat_free(&Bc);
	return 0;
}

/* ------------------------------------------------------------------ */
/* Row-wise restatement: same result tuples, Gustavson loop order.     */

/* dense lookup of a sorted-unique scale vector: pos[i] = position or -1 */
static int scale_lookup(const orc_vec *s, size_t dim, int64_t **pos_out)
{
	int64_t *pos = (int64_t *)xmalloc(dim * sizeof(int64_t));
	for (size_t i = 0; i < dim; ++i) pos[i] = -1;
	for (size_t t = 0; t < s->nnz; ++t) {
		if (t > 0 && s->idx[t] <= s->idx[t - 1]) { free(pos); return -1; }
		/* an index outside the dimension can never match in the join */
		if (s->idx[t] < 0 || (size_t)s->idx[t] >= dim) continue;
		pos[s->idx[t]] = (int64_t)t;
	}
	*pos_out = pos;
	return 0;
}

static int cmp_i32(const void *a, const void *b)
{
	int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
	return (x > y) - (x < y);
}

/* Shared body of the row-wise checker.  dg == NULL: the tuples go to `ret` (rows split statically over the
 * threads, parts appended in row order).  dg != NULL: streaming digest -- nothing is stored, rows are handed
 * out dynamically (an un-permuted R-MAT has most of its products in its first rows), and only the rows with
 * row_mask[i] != 0 are evaluated when a mask is given. */
static int rowwise_core(orc_coo *ret, orc_digest_out *dg, const uint8_t *row_mask, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, int nthreads, char *msg, size_t msglen)
{
	int a0 = transpose_A == 'T' ? 1 : 0, a1 = 1 - a0;
	int b0 = transpose_B == 'T' ? 0 : 1, b1 = 1 - b0;
	size_t ashape[2] = { A->shape0, A->shape1 };
	size_t bshape[2] = { B->shape0, B->shape1 };
	if (ret) { ret->shape0 = ashape[a0]; ret->shape1 = bshape[b0]; }
	if (dg) { dg->count = 0; dg->hash = 0; dg->sum = 0; dg->products = 0; dg->nnz_a = 0; dg->nnz_b = 0; }
	if (ashape[a1] != bshape[b1]) {
		set_msg(msg, msglen, "Inner dimensions for A (%ld) and B (%ld) must match!",
			(long)ashape[a1], (long)bshape[b1]);
		return -1;
	}
	if (orc_isnone(C, 0)
		|| (scalei && scalei->nnz == 0) || A->nnz == 0
		|| (scalej && scalej->nnz == 0) || B->nnz == 0
		|| (scalek && scalek->nnz == 0))
		return 0;

	size_t nrow = ashape[a0], ninner = ashape[a1], ncol = bshape[b0];
	int64_t *pi = NULL, *pj = NULL, *pk = NULL;
	if ((scalei && scale_lookup(scalei, nrow, &pi)) ||
		(scalej && scale_lookup(scalej, ninner, &pj)) ||
		(scalek && scale_lookup(scalek, ncol, &pk))) {
		free(pi); free(pj); free(pk);
		if (msg && msglen) snprintf(msg, msglen, "row-wise checker needs sorted, unique scale vectors");
		return -2;
	}

	/* A sorted by rows of op(A).  B: consolidated exactly as the reference does it, by its
	 * own (column-major of op(B)) sort order b0 -- which tuples zero_nan drops depends on that
	 * sequence (algorithm.hpp:272-275 vs :284-292) -- and THEN put in rows-of-op(B) order by a
	 * plain stable sort that drops and merges nothing (explicit zeros left by a +x/-x merge
	 * stay, as they do in the reference's copy).  Without zero_nan the sequence does not matter
	 * (exact zeros are dropped wherever they stand and equal indices merge in insertion order under
	 * either stable sort), so B is consolidated by rows of op(B) at once. */
	conmat Ac, Bc, Bref;
	conmat_build(&Ac, A, a0, duplicate_policy, zero_nan);
	if (!zero_nan) {
		conmat_build(&Bc, B, b1, duplicate_policy, 0);
	} else {
		conmat_build(&Bref, B, b0, duplicate_policy, zero_nan);
		/* Bref: lead = index(b0), minor = index(b1); re-sort by (minor, lead) */
		size_t nb = Bref.n;
		size_t *perm = (size_t *)xmalloc((nb ? nb : 1) * sizeof(size_t));
		orc_sorted_permutation(2, Bref.minor, Bref.lead, nb, 0, perm);
		orc_mat tmp;
		int32_t *t0 = (int32_t *)xmalloc((nb ? nb : 1) * sizeof(int32_t));
		int32_t *t1 = (int32_t *)xmalloc((nb ? nb : 1) * sizeof(int32_t));
		double *tv = (double *)xmalloc((nb ? nb : 1) * sizeof(double));
		for (size_t e = 0; e < nb; ++e) { t0[e] = Bref.minor[perm[e]]; t1[e] = Bref.lead[perm[e]]; tv[e] = Bref.val[perm[e]]; }
		tmp.idx0 = t0; tmp.idx1 = t1; tmp.val = tv; tmp.nnz = nb;
		tmp.shape0 = bshape[b1]; tmp.shape1 = bshape[b0]; tmp.sort0 = 0;      /* trusted as is: nothing dropped */
		conmat_build(&Bc, &tmp, 0, duplicate_policy, 0);
		free(perm); free(t0); free(t1); free(tv);
		conmat_free(&Bref);
	}
	if (dg) { dg->nnz_a = Ac.n; dg->nnz_b = Bc.n; }

	/* row pointer of op(B) over all inner indices */
	size_t *bptr = (size_t *)xmalloc((ninner + 1) * sizeof(size_t));
	{
		size_t e = 0;
		for (size_t k = 0; k <= ninner; ++k) {
			while (e < Bc.n && (size_t)Bc.lead[e] < k) ++e;
			bptr[k] = e;
		}
	}

	if (nthreads < 1) nthreads = 1;
	orc_coo *parts = NULL;
	if (ret) {
		parts = (orc_coo *)xmalloc((size_t)nthreads * sizeof(orc_coo));
		for (int t = 0; t < nthreads; ++t) orc_coo_init(&parts[t], 2);
	}
	uint64_t g_count = 0, g_hash = 0, g_prod = 0;
	double g_sum = 0;
	size_t next_row = 0;                    /* digest mode: the shared cursor rows are handed out from */
	const size_t chunk = 16;

#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
	{
#ifdef _OPENMP
		int t = omp_get_thread_num();
		int nt = omp_get_num_threads();
#else
		int t = 0, nt = 1;
#endif
		if (t < nthreads) {
			double *acc = (double *)xmalloc(ncol * sizeof(double));
			unsigned char *flag = (unsigned char *)calloc(ncol ? ncol : 1, 1);
			int32_t *touched = (int32_t *)xmalloc(ncol * sizeof(int32_t));
			uint64_t l_count = 0, l_hash = 0, l_prod = 0;
			double l_sum = 0;
			size_t r0 = Ac.nrows * (size_t)t / (size_t)nt;
			size_t r1 = Ac.nrows * (size_t)(t + 1) / (size_t)nt;
			for (;;) {
				if (dg) {
#ifdef _OPENMP
#pragma omp atomic capture
#endif
					{ r0 = next_row; next_row += chunk; }
					if (r0 >= Ac.nrows) break;
					r1 = r0 + chunk < Ac.nrows ? r0 + chunk : Ac.nrows;
				}
				for (size_t ra = r0; ra < r1; ++ra) {
					int32_t aix = Ac.rowid[ra];
					if (row_mask && !row_mask[aix]) continue;
					double a_scale = 1.0;
					if (scalei) {
						if (pi[aix] < 0) continue;      /* absent => row skipped */
						a_scale = scalei->val[pi[aix]];
					}
					if (orc_isnone(a_scale, 0)) continue;
					size_t nt_ = 0;
					for (size_t e = Ac.beg[ra]; e < Ac.beg[ra + 1]; ++e) {   /* ascending k */
						int32_t k = Ac.minor[e];
						double a = Ac.val[e];
						if (scalej) {
							if (pj[k] < 0) continue;    /* absent => term dropped */
							a = a * scalej->val[pj[k]]; /* (a*s)*b, multiply_sparse.hpp:228 */
						}
						l_prod += (uint64_t)(bptr[k + 1] - bptr[k]);
						for (size_t f = bptr[k]; f < bptr[k + 1]; ++f) {
							int32_t j = Bc.minor[f];
							if (!flag[j]) { flag[j] = 1; touched[nt_++] = j; acc[j] = 0; }
							acc[j] += a * Bc.val[f];
						}
					}
					if (!dg) qsort(touched, nt_, sizeof(int32_t), cmp_i32);
					uint64_t r_count = 0, r_hash = 0;
					for (size_t q = 0; q < nt_; ++q) {
						int32_t j = touched[q];
						double sum = acc[j];
						flag[j] = 0;
						double b_scale = 1.0;
						if (scalek) {
							if (pk[j] < 0) continue;
							b_scale = scalek->val[pk[j]];
						}
						if (orc_isnone(b_scale, 0)) continue;
						if (orc_isnone(sum, 0)) continue;            /* multiply_sparse.hpp:238 */
						double v = sum * C * a_scale * b_scale;     /* :242 */
						if (dg) { ++r_count; r_hash += orc_mix64((uint32_t)aix, (uint32_t)j); l_sum += v; }
						else coo_add(&parts[t], aix, j, v);
					}
					if (dg) {
						l_count += r_count; l_hash += r_hash;
						if (dg->row_nnz) dg->row_nnz[aix] = (int64_t)r_count;     /* one writer per row */
						if (dg->row_hash) dg->row_hash[aix] = r_hash;
					}
				}
				if (!dg) break;
			}
			free(acc); free(flag); free(touched);
			if (dg) {
#ifdef _OPENMP
#pragma omp critical
#endif
				{ g_count += l_count; g_hash += l_hash; g_sum += l_sum; g_prod += l_prod; }
			}
		}
	}
	if (ret) {
		for (int t = 0; t < nthreads; ++t) {
			for (size_t q = 0; q < parts[t].n; ++q)
				coo_add(ret, parts[t].i[q], parts[t].j[q], parts[t].v[q]);
			orc_coo_free(&parts[t]);
		}
		free(parts);
	}
	if (dg) { dg->count = g_count; dg->hash = g_hash; dg->sum = g_sum; dg->products = g_prod; }
	free(bptr);
	free(pi); free(pj); free(pk);
	conmat_free(&Ac);
	conmat_free(&Bc);
	return 0;
}

int orc_multiply_mm_rowwise(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, int nthreads, char *msg, size_t msglen)
{
	return rowwise_core(ret, NULL, NULL, C, scalei, A, transpose_A, scalej, B, transpose_B, scalek,
		duplicate_policy, zero_nan, nthreads, msg, msglen);
}

int orc_multiply_mm_rowwise_digest(orc_digest_out *out, const uint8_t *row_mask, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_mat *B, char transpose_B, const orc_vec *scalek,
	int duplicate_policy, int zero_nan, int nthreads, char *msg, size_t msglen)
{
	return rowwise_core(NULL, out, row_mask, C, scalei, A, transpose_A, scalej, B, transpose_B, scalek,
		duplicate_policy, zero_nan, nthreads, msg, msglen);
}

/* ------------------------------------------------------------------ */
/* multiply MV: multiply_sparse.hpp:281-365                            */

int orc_multiply_mv(orc_coo *ret, double C, const orc_vec *scalei,
	const orc_mat *A, char transpose_A, const orc_vec *scalej,
	const orc_vec *V, int duplicate_policy, int zero_nan,
	char *msg, size_t msglen)
{
	int a0 = transpose_A == 'T' ? 1 : 0, a1 = 1 - a0;      /* :294 */
	size_t ashape[2] = { A->shape0, A->shape1 };
	ret->shape0 = ashape[a0];                               /* :295 */
	ret->shape1 = 0;
	if (ashape[a1] != V->shape0) {                          /* :298-300 */
		set_msg(msg, msglen, "Inner dimensions for A (%ld) and V (%ld) must match!",
			(long)ashape[a1], (long)V->shape0);
		return -1;
	}
	if (orc_isnone(C, 0)                                    /* :304-309 */
		|| (scalei && scalei->nnz == 0) || A->nnz == 0
		|| (scalej && scalej->nnz == 0) || V->nnz == 0)
		return 0;

	conmat Ac;
	conmat_build(&Ac, A, a0, duplicate_policy, zero_nan);   /* :312 */
	/* :313 Consolidate<VecT>(&V, {0}, ...) */
	int32_t *vi = (int32_t *)xmalloc(V->nnz * sizeof(int32_t));
	double *vv = (double *)xmalloc(V->nnz * sizeof(double));
	size_t vn;
	if (V->sort0 == 0) {
		memcpy(vi, V->idx, V->nnz * sizeof(int32_t));
		memcpy(vv, V->val, V->nnz * sizeof(double));
		vn = V->nnz;
	} else {
		vn = orc_consolidate(1, V->idx, NULL, V->val, V->nnz, 0, duplicate_policy, zero_nan, vi, NULL, vv);
	}
	if (Ac.nrows == 0) goto done;                           /* Appendix A.3 */

	multx ja;
	for (multx_init(&ja, &Ac, scalei); !multx_eof(&ja); multx_incr(&ja)) {   /* :319 */
		double a_scale = multx_scale_val(&ja);
		if (orc_isnone(a_scale, 0)) continue;               /* :322 */
		size_t ra = multx_row(&ja);
		int32_t aix = Ac.rowid[ra];
		double sum = 0;                                     /* :334 */
		joinx ab;
		if (scalej) {                                       /* :336-343 */
			stream_set(&ab.s[0], Ac.minor, Ac.beg[ra], Ac.beg[ra + 1]);
			stream_set(&ab.s[1], scalej->idx, 0, scalej->nnz);
			stream_set(&ab.s[2], vi, 0, vn);
			for (join_init(&ab, 3); !ab.eof; join_incr(&ab))
				sum += Ac.val[ab.s[0].pos] * scalej->val[ab.s[1].pos] * vv[ab.s[2].pos];
		} else {                                            /* :344-354 */
			stream_set(&ab.s[0], Ac.minor, Ac.beg[ra], Ac.beg[ra + 1]);
			stream_set(&ab.s[1], vi, 0, vn);
			for (join_init(&ab, 2); !ab.eof; join_incr(&ab))
				sum += Ac.val[ab.s[0].pos] * vv[ab.s[1].pos];
		}
		if (!orc_isnone(sum, 0))                            /* :356-361 */
			coo_add(ret, aix, 0, sum * C * a_scale);
	}
done:
	free(vi); free(vv);
	conmat_free(&Ac);
	return 0;
}

/* ------------------------------------------------------------------ */
/* digest shared with the device checksum sink                         */

uint64_t orc_mix64(uint32_t i, uint32_t j)
{
	uint64_t x = ((((uint64_t)i) << 32) | (uint64_t)j) * 0x9E3779B97F4A7C15ull;
	return x ^ (x >> 29);
}

void orc_digest(const orc_coo *c, uint64_t *count, double *sum, uint64_t *hash)
{
	uint64_t h = 0; double s = 0;
	for (size_t q = 0; q < c->n; ++q) {
		h += orc_mix64((uint32_t)c->i[q], c->rank == 2 ? (uint32_t)c->j[q] : 0u);
		s += c->v[q];
	}
	*count = (uint64_t)c->n; *sum = s; *hash = h;
}
