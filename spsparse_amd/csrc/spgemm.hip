// spgemm.hip -- the hot path: C = c * Di * op(A) * Dj * op(B) * Dk on gfx950.
//
// Replaces the reference's triple loop (multiply_sparse.hpp:192-246: every
// non-empty row of A x every non-empty column of B, a leap-frog merge join
// per pair, xiter.hpp / next_noincr_body.hpp) with a row-partitioned
// expand / accumulate / compress pipeline (Gustavson order):
//
//   symbolic   per A tuple the length of the B row it selects, prefix-summed;
//              per output row the product count P_r; rows are binned by P_r.
//   light      P_r <= 64: one wave handles 64/S rows; products are expanded one
//              per lane, ranked by (col, k) through LDS, summed in ascending k
//              (bit-identical to the reference's `sum += a*b` order,
//              multiply_sparse.hpp:219-236) and compacted with ballot/popcount.
//   cells      above the light bin the unit of work is a cell: one output row
//              restricted to a range of column windows (W = 8192 or 16384
//              columns).  A mid row (P_r <= 4096) is one cell.  A heavy row's
//              windows are grouped greedily into hash cells of <= 2048 products;
//              a single window with more becomes a dense cell.  B's row panels
//              are pre-indexed per window (bwin) so a cell reads exactly its B
//              segments; the cell lists are sorted window-major for L2 locality.
//   hash       persistent workgroups; LDS hash accumulator keyed by column
//              (ds_cmpswap + ds_add_f64) with a list of occupied slots, then an
//              in-LDS bitonic sort of the surviving columns for ordered emission.
//   dense      persistent workgroups; dense f64 accumulator of W columns in LDS
//              (ds_add_f64), scanned out in ascending column order.
//
// Output semantics follow multiply_sparse.hpp:238-243: exact-zero sums are
// dropped, value = sum * C * a_scale * b_scale, tuples in ascending (i, j).
// No MFMA: 2 flops per 12 bytes read.
#include "internal.h"
#include "devutil.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace spsamd {

enum { MODE_COUNT = 0, MODE_STORE = 1, MODE_DIGEST = 2 };

constexpr int NBIN = 9;          // 0 none | 1..4 light (S = 8,16,32,64) | 5..7 mid (T = 1024,4096,8192) | 8 heavy
#ifndef MID_MAX_V
#define MID_MAX_V 4096
#endif
constexpr uint32_t MID_MAX = MID_MAX_V;
#ifndef DENSE_U
#define DENSE_U 1
#endif
#ifndef HASH_U
#define HASH_U 4
#endif
constexpr int DIGEST_SLOTS = 1024;

struct EmitParams {
	double C;
	const int32_t *si_pos; const double *si_val;     // row scale (null: none)
	const int32_t *sk_pos; const double *sk_val;     // column scale (null: none)
	int emit_path;                                   // COO emission of hash cells: 0 auto, 1 never the bitmap rank, 2 bitonic network only (same result)
#ifdef SPSAMD_ABLATIONS
	int dbg;                                         // profiling builds only: ablation bits that skip work (wrong results on purpose)
#endif
	uint32_t wshift;                                 // log2 of the column-window width of the heavy path (0 before it is chosen)
	uint32_t ncolbits;                               // bits of the largest column index
	int ordered;                                     // SPSAMD_SINK_ORDERED: ascending-k sums everywhere (bit-exact)
	int pattern;                                     // SPSAMD_SINK_EXACT_PATTERN: sums that could be zero in only one summation order are re-evaluated in ascending k
};

// Ablation switches exist in profiling builds only (-DSPSAMD_ABLATIONS); the shipped library has none.
#ifdef SPSAMD_ABLATIONS
#define ABL(ep, bit) ((ep).dbg & (bit))
__device__ int g_abl;                       // the same switches for device functions that do not see EmitParams
#define ABLG(bit) (g_abl & (bit))
#else
#define ABL(ep, bit) false
#define ABLG(bit) false
#endif

struct DigestSlot { unsigned long long count; unsigned long long hash; double sum; unsigned long long pad; };

struct SinkParams {
	const uint32_t *segbase;        // per non-empty A row: first segment id        (COUNT / STORE)
	uint32_t *segcount;             // per segment: tuples reserved                 (COUNT writes)
	const int64_t *segoff;          // per segment: output offset                   (STORE reads)
	uint32_t *segactual;            // per segment: tuples written                  (STORE writes)
	int32_t *out_i; int32_t *out_j; double *out_v;
	DigestSlot *digest;             // DIGEST_SLOTS accumulators
	long long *row_nnz; double *row_sum;   // optional row statistics (DIGEST)
	uint32_t *err;                  // device error word: a kernel that meets a state the host promised cannot occur sets a bit
#ifdef SPSAMD_STAMPS
	unsigned long long *stamps;     // diagnostic builds only: per-workgroup cycle counters of k_dense's phases
#endif
};

#ifdef SPSAMD_STAMPS
#define STAMP(i) do { const unsigned long long now_ = clock64(); st_[i] += now_ - st_t; st_t = now_; } while (0)
#define STAMP_COUNT(i) (++st_[i])
#else
#define STAMP(i) do { } while (0)
#define STAMP_COUNT(i) do { } while (0)
#endif

// One B tuple as the numeric kernels read it: column and value side by side (12 bytes), so
// a short B segment sits in one or two cache lines instead of two partial lines of separate
// col[] / val[] arrays.  Same bytes per product as the SoA form (12 B).
struct __attribute__((packed, aligned(4))) BTup { int32_t col; uint32_t vlo, vhi; };
__device__ __forceinline__ double btup_val(const BTup &t) { return __hiloint2double((int)t.vhi, (int)t.vlo); }

__global__ void k_pack_b(const int32_t *bcol, const double *bval, uint32_t n, BTup *out)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double v = bval[i];
	BTup t; t.col = bcol[i]; t.vlo = (uint32_t)__double2loint(v); t.vhi = (uint32_t)__double2hiint(v);
	out[i] = t;
}

struct RowMeta {
	const uint32_t *beg;            // per non-empty A row: first tuple (+ sentinel)
	const int32_t *id;              // per non-empty A row: row index
	const int32_t *acol;            // A tuples: inner index k
	const double *aval;             // A tuples: value (already times scalej)
	const uint32_t *bptr;           // B dense row pointer
	const BTup *btup;               // B tuples, (col, val) interleaved (a dense / direct launch points it at the window-major copy)
	const BTup *btup_rm;            // ... always the row-major array (ordered re-evaluation)
	const uint32_t *elo;            // A tuples: first B tuple of the selected row (bptr[k])
	const uint32_t *elen;           // A tuples: length of the selected B row
};

__device__ __forceinline__ double row_scale(const EmitParams &p, int32_t rowid)
{
	return p.si_pos ? p.si_val[p.si_pos[rowid]] : 1.0;
}

// isnone(sum) and the scalek skip (multiply_sparse.hpp:211,238), then
// sum * C * a_scale * b_scale left to right (multiply_sparse.hpp:242).
__device__ __forceinline__ bool emit_value(const EmitParams &p, double a_scale, int32_t col, double sum, double *out)
{
	if (sum == 0) return false;
	double b_scale = 1.0;
	if (p.sk_pos) {
		int32_t q = p.sk_pos[col];
		if (q < 0) return false;
		b_scale = p.sk_val[q];
		if (b_scale == 0) return false;
	}
	*out = sum * p.C * a_scale * b_scale;
	return true;
}

__device__ __forceinline__ bool col_allowed(const EmitParams &p, int32_t col)
{
	if (!p.sk_pos) return true;
	int32_t q = p.sk_pos[col];
	return q >= 0 && p.sk_val[q] != 0;
}

// ---- SPSAMD_SINK_EXACT_PATTERN: the index set of the reference, at arrival-order speed -----------------
// Hash and dense cells add their products with LDS atomics in arrival order.  The VALUES then differ from the
// reference's ascending-k sums by rounding only (north star: 1e-12), but the test `sum == 0` that decides whether a
// tuple exists at all (multiply_sparse.hpp:238) can come out differently when terms cancel.  Two sums of the same n
// terms in different orders differ by at most 2 (n-1) u S, S = sum of the |terms|, u = 2^-53; so only a slot whose
// arrival-order sum is within that bound of zero can be zero in one order and not in the other.  Per cell the
// kernels track S over ALL its products (an upper bound of every slot's own S) and whether products of both signs
// occurred: a cell of one sign cannot cancel at all; otherwise a slot with |sum| <= 8 nseg u S_cell is re-evaluated
// in ascending k straight from the operands (ordered_sum) and that exact value decides and is emitted.
struct PatAcc { double sabs; uint32_t sor, sand; };              // per lane
struct PatCell { double sabs; uint32_t sor, sand; uint32_t pad; };   // per cell, in LDS

__device__ __forceinline__ void pat_init(PatAcc &a) { a.sabs = 0.0; a.sor = 0u; a.sand = 0xFFFFFFFFu; }
__device__ __forceinline__ void pat_note(PatAcc &a, double p)
{
	a.sabs += fabs(p);
	const uint32_t hi = (uint32_t)__double2hiint(p);
	a.sor |= hi; a.sand &= hi;
}
// every wave adds its lanes' notes to the cell's record (call before the barrier that ends the accumulation).  The wave
// reduction runs on DPP (row shifts and broadcasts: VALU only; lane 63 ends up with the total) -- with __shfl_xor it was 24
// ds_bpermute per wave and cell, in kernels whose LDS pipe is the busy one.
__device__ __forceinline__ void pat_publish(PatAcc &a, PatCell *cell)
{
	double sa = a.sabs;
	uint32_t so = a.sor, sn = a.sand;
#define PAT_DPP_STEP(ctrl, rows) do { \
		const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(sa), ctrl, rows, 0xF, true); \
		const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(sa), ctrl, rows, 0xF, true); \
		sa += __hiloint2double(hi_, lo_); \
		so |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)so, ctrl, rows, 0xF, true); \
		sn &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)sn, ctrl, rows, 0xF, false); \
	} while (0)
	PAT_DPP_STEP(0x111, 0xF);       // row_shr:1
	PAT_DPP_STEP(0x112, 0xF);       // row_shr:2
	PAT_DPP_STEP(0x114, 0xF);       // row_shr:4
	PAT_DPP_STEP(0x118, 0xF);       // row_shr:8
	PAT_DPP_STEP(0x142, 0xA);       // row_bcast:15 -> rows 1 and 3
	PAT_DPP_STEP(0x143, 0xC);       // row_bcast:31 -> rows 2 and 3
#undef PAT_DPP_STEP
	if (lane_id() == 63) { atomicAdd(&cell->sabs, sa); atomicOr(&cell->sor, so); atomicAnd(&cell->sand, sn); }
	pat_init(a);
}
__device__ __forceinline__ void pat_reset(PatCell *cell) { cell->sabs = 0.0; cell->sor = 0u; cell->sand = 0xFFFFFFFFu; }
// |sum| at or below the returned bound: re-evaluate.  -1: the cell cannot cancel (one sign, all finite).
__device__ __forceinline__ double pat_threshold(const PatCell *cell, uint32_t nseg)
{
	const double S = cell->sabs;
	if (!(S < __longlong_as_double(0x7FF0000000000000ll))) return __longlong_as_double(0x7FF0000000000000ll);   // inf / NaN terms: every sum
	if ((((cell->sor ^ cell->sand) >> 31) & 1u) == 0u) return -1.0;
	return 8.0 * (double)nseg * 0x1p-53 * S;
}
// The reference's own sum for output (row of A tuples [beg, end), column col): ascending k, `sum += a*b`
// (multiply_sparse.hpp:219-236), for the rare slots pat_threshold singles out.  Evaluated by a whole wave
// (every lane must call it, with wave-uniform arguments): the lanes look up 64 A
// tuples' B rows at a time, then the terms that exist are added in ascending position -- the reference's order -- with
// wave-uniform lane reads.  A re-evaluation by ONE lane walks the row's tuples one dependent binary search after the
// other: 17 ms for a row of 1000 tuples, and a hub row has tens of thousands.
__device__ double ordered_sum_wave(const RowMeta &m, uint32_t beg, uint32_t end, int32_t col)
{
	double sum = 0.0;
	for (uint32_t base = beg; base < end; base += 64u) {
		const uint32_t e = base + lane_id();
		double term = 0.0;
		bool has = false;
		if (e < end) {
			const int32_t k = m.acol[e];
			uint32_t lo = m.bptr[k];
			const uint32_t top = m.bptr[k + 1];
			uint32_t hi = top;
			while (lo < hi) {
				const uint32_t mid = lo + ((hi - lo) >> 1);
				if (m.btup_rm[mid].col < col) lo = mid + 1; else hi = mid;
			}
			if (lo < top && m.btup_rm[lo].col == col) { term = m.aval[e] * btup_val(m.btup_rm[lo]); has = true; }
		}
		unsigned long long hm = __ballot(has);
		while (hm) {                                                // uniform: ascending e
			const int l = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)hm) - 1);
			hm &= hm - 1ull;
			const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(term), l), __builtin_amdgcn_readlane(__double2loint(term), l));
			sum += t;
		}
	}
	return sum;
}
// x of the lanes with `need` set is replaced by the reference's sum for (the cell's row, that lane's column).  Every lane
// of the wave must call it (converged); beg / end wave-uniform.
__device__ __forceinline__ double pat_fix_wave(bool need, double x, int32_t col, const RowMeta &m, uint32_t beg, uint32_t end)
{
	unsigned long long mask = __ballot(need);
	while (mask) {                                                  // uniform
		const int l = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)mask) - 1);
		mask &= mask - 1ull;
		const int32_t c = __builtin_amdgcn_readlane(col, l);
		const double r = ordered_sum_wave(m, beg, end, c);
		if ((int)lane_id() == l) x = r;
	}
	return x;
}

// Workgroup-wide digest accumulation: one set of atomics per workgroup, spread
// over DIGEST_SLOTS accumulators so no address becomes a serial hot spot.
template <int NT>
__device__ __forceinline__ void digest_flush(DigestSlot *slots, unsigned long long cnt, unsigned long long hash, double sum,
	unsigned long long *s_u64, double *s_f64)
{
	cnt = wave_reduce_sum(cnt);
	hash = wave_reduce_sum(hash);
	sum = wave_reduce_sum(sum);
	constexpr int NW = NT / 64;
	if (lane_id() == 0) { s_u64[wave_id()] = cnt; s_u64[NW + wave_id()] = hash; s_f64[wave_id()] = sum; }
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long c = 0, h = 0; double s = 0;
		for (int w = 0; w < NW; ++w) { c += s_u64[w]; h += s_u64[NW + w]; s += s_f64[w]; }
		if (c) {
			DigestSlot *d = &slots[blockIdx.x % DIGEST_SLOTS];
			atomicAdd(&d->count, c);
			atomicAdd(&d->hash, h);
			atomicAdd(&d->sum, s);
		}
	}
	__syncthreads();
}

// ====================================================================== symbolic

__global__ void k_elem_len(const int32_t *acol, const uint32_t *bptr, uint32_t n, uint32_t *lo, uint32_t *len)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e < n) { int32_t k = acol[e]; uint32_t b = bptr[k]; lo[e] = b; len[e] = bptr[k + 1] - b; }
}

// scalej (multiply_sparse.hpp:221-228): a k absent from the vector drops the
// term -> the tuple is redirected to the empty sentinel row `ninner`;
// otherwise a' = a * s so that each product is (a*s)*b, left to right.
__global__ void k_apply_scalej(const int32_t *acol, const double *aval, uint32_t n, const int32_t *pos, const double *sval,
	int32_t ninner, int32_t *acol2, double *aval2)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n) return;
	int32_t k = acol[e];
	int32_t q = pos[k];
	if (q < 0) { acol2[e] = ninner; aval2[e] = 0.0; }
	else { acol2[e] = k; aval2[e] = aval[e] * sval[q]; }
}

struct BinCounters { unsigned long long rows[NBIN]; unsigned long long prods[NBIN]; unsigned long long tuples[NBIN]; unsigned long long too_big; };

__device__ __forceinline__ int bin_of(uint32_t P)
{
	if (P == 0) return 0;
	if (P <= 8) return 1;
	if (P <= 16) return 2;
	if (P <= 32) return 3;
	if (P <= 64) return 4;
	if (P <= 512) return 5;
	if (P <= 2048) return 6;
	if (P <= MID_MAX) return 7;
	return 8;
}

// Per non-empty A row: product count and bin.  scalei (multiply_sparse.hpp:195
// and the Join2 of ScaledMultXiter :79-86): a row absent from the vector, or
// whose scale is 0, is skipped.
constexpr int CLS_ITEMS = 16;                   // rows per thread: few workgroups -> few same-address global atomics

__global__ __launch_bounds__(256) void k_classify(const uint32_t *beg, const int32_t *id, uint32_t nrows, const int64_t *pref, const uint32_t *elen,
	const int32_t *si_pos, const double *si_val, uint32_t *rprod, uint8_t *rbin, BinCounters *bc)
{
	// bin statistics: wave-aggregated (ballot per bin present in the wave), then LDS, then one
	// global atomic per (workgroup, bin).  Same-address global atomics serialise (~5 ns each), so
	// a workgroup covers 256 * CLS_ITEMS rows.
	__shared__ unsigned int s_rows[NBIN];
	__shared__ unsigned long long s_prods[NBIN];
	__shared__ unsigned long long s_tuples[NBIN];
	if (threadIdx.x < NBIN) { s_rows[threadIdx.x] = 0; s_prods[threadIdx.x] = 0; s_tuples[threadIdx.x] = 0; }
	__syncthreads();
	for (int it = 0; it < CLS_ITEMS; ++it) {
		uint32_t r = (blockIdx.x * CLS_ITEMS + it) * 256u + threadIdx.x;
		int b = -1;
		uint32_t P = 0, La = 0;
		if (r < nrows) {
			const uint32_t b0 = beg[r], b1 = beg[r + 1];
			La = b1 - b0;
			if (La <= 8) {                          // short row: its tuples' lengths are one or two cache lines
				for (uint32_t e = b0; e < b1; ++e) P += elen[e];
			} else {
				const int64_t d = pref[b1] - pref[b0];
				if (d > 0xFFFFFFFFll) atomicAdd(&bc->too_big, 1ull);    // the per-row counters are 32 bits wide: reported by the host
				P = (uint32_t)d;
			}
			if (si_pos) {
				int32_t q = si_pos[id[r]];
				if (q < 0 || si_val[q] == 0) P = 0;
			}
			b = bin_of(P);
			rprod[r] = P;
			rbin[r] = (uint8_t)b;
		}
		uint64_t todo = __ballot(b >= 0);
		while (todo) {
			int leader = __ffsll((unsigned long long)todo) - 1;
			int bb = __shfl(b, leader, 64);
			uint64_t mine = __ballot(b == bb);
			unsigned long long p = wave_reduce_sum((unsigned long long)(b == bb ? P : 0u));
			unsigned long long t = wave_reduce_sum((unsigned long long)(b == bb ? La : 0u));
			if ((int)lane_id() == leader) {
				atomicAdd(&s_rows[bb], (unsigned int)__popcll(mine));
				atomicAdd(&s_prods[bb], p);
				atomicAdd(&s_tuples[bb], t);
			}
			todo &= ~mine;
		}
	}
	__syncthreads();
	if (threadIdx.x < NBIN && s_rows[threadIdx.x]) {
		atomicAdd(&bc->rows[threadIdx.x], (unsigned long long)s_rows[threadIdx.x]);
		atomicAdd(&bc->prods[threadIdx.x], s_prods[threadIdx.x]);
		atomicAdd(&bc->tuples[threadIdx.x], s_tuples[threadIdx.x]);
	}
}

struct BinOffsets { uint32_t off[NBIN + 1]; };

__global__ __launch_bounds__(256) void k_bin_scatter(const uint8_t *rbin, uint32_t nrows, BinOffsets bo, uint32_t *cursor, uint32_t *binrows)
{
	// a workgroup (256 * CLS_ITEMS rows) reserves one range per bin with a single global atomic;
	// inside it rows are ranked per wave with ballots (one LDS atomic per wave, bin and step)
	__shared__ unsigned int s_cnt[NBIN];
	__shared__ unsigned int s_base[NBIN];
	if (threadIdx.x < NBIN) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	unsigned int local[CLS_ITEMS];
	uint8_t bins[CLS_ITEMS];
#pragma unroll
	for (int it = 0; it < CLS_ITEMS; ++it) {
		uint32_t r = (blockIdx.x * CLS_ITEMS + it) * 256u + threadIdx.x;
		int b = r < nrows ? rbin[r] : 0;
		unsigned int loc = 0;
		uint64_t todo = __ballot(b != 0);
		while (todo) {
			int leader = __ffsll((unsigned long long)todo) - 1;
			int bb = __shfl(b, leader, 64);
			uint64_t mine = __ballot(b == bb);
			unsigned int base = 0;
			if ((int)lane_id() == leader) base = atomicAdd(&s_cnt[bb], (unsigned int)__popcll(mine));
			base = (unsigned int)__shfl((int)base, leader, 64);
			if (b == bb) loc = base + (unsigned int)__popcll(mine & lanemask_lt());
			todo &= ~mine;
		}
		local[it] = loc; bins[it] = (uint8_t)b;
	}
	__syncthreads();
	if (threadIdx.x < NBIN && threadIdx.x > 0 && s_cnt[threadIdx.x])
		s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
	__syncthreads();
#pragma unroll
	for (int it = 0; it < CLS_ITEMS; ++it) {
		uint32_t r = (blockIdx.x * CLS_ITEMS + it) * 256u + threadIdx.x;
		int b = bins[it];
		if (b) binrows[bo.off[b] + s_base[b] + local[it]] = r;
	}
}

// ====================================================================== light rows

// LDS traffic of ONE wave is ordered by the hardware; waiting for its completion makes a
// wave's stores visible to its other lanes without a workgroup barrier.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One wave handles G = 64/S rows, S product slots each.
template <int S, int MODE>
__global__ __launch_bounds__(256, 8) void k_light(const uint32_t *binrows, uint32_t nbin, RowMeta m, EmitParams ep, SinkParams sk)
{
	constexpr int G = 64 / S;
	__shared__ uint32_t s_apos[4][64];
	__shared__ uint32_t s_bpos[4][64];
	__shared__ uint64_t s_key[4][64];
	__shared__ uint64_t s_key2[4][64];
	__shared__ double s_val2[4][64];
	__shared__ unsigned long long s_u64[8];
	__shared__ double s_f64[4];

	const unsigned w = wave_id(), lane = lane_id();
	const unsigned g = lane / S, s = lane % S;
	// grid-stride loop over groups of 4*G rows: the digest of a workgroup is flushed once, not per group
	unsigned long long d_cnt = 0, d_hash = 0; double d_sum = 0.0;
	const uint32_t nvb = (nbin + 4u * G - 1u) / (4u * G);
	for (uint32_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
	const uint32_t rix = (vb * 4u + w) * G + g;
	const bool has_row = rix < nbin;
	const uint32_t r = has_row ? (binrows ? binrows[rix] : rix) : 0u;      // null list: the bin holds every row
	const uint32_t beg = has_row ? m.beg[r] : 0u;
	const uint32_t end = has_row ? m.beg[r + 1] : 0u;

	// ---- expand: slot t of the row's P products -> (A tuple, B tuple)
	const uint32_t La = end - beg;
	uint32_t off = 0;
	for (uint32_t base = 0; __any(base < La); base += S) {
		uint32_t e = beg + base + s;
		bool act = has_row && e < end;
		uint32_t lo = 0, len = 0;
		if (act) { lo = m.elo[e]; len = m.elen[e]; }
		const uint32_t inc = group_inclusive_scan_u32<S>(len, s);
		uint32_t ex = off + inc - len;
		for (uint32_t t = 0; t < len; ++t) {        // ex + t < S because P_r <= S
			s_apos[w][g * S + ex + t] = e;
			s_bpos[w][g * S + ex + t] = lo + t;
		}
		off += (uint32_t)__shfl((int)inc, (int)(g * S + S - 1), 64);
	}
	s_key2[w][lane] = ~0ull;
	wave_lds_sync();                // the LDS arrays are per wave: no workgroup barrier needed

	// ---- product + key (col, A position): ascending A position = ascending k
	const bool act = has_row && s < off;
	uint64_t key = ~0ull;
	double prod = 0;
	if (act) {
		uint32_t ap = s_apos[w][lane], bp = s_bpos[w][lane];
		const BTup t = m.btup[bp];
		prod = m.aval[ap] * btup_val(t);
		key = ((uint64_t)(uint32_t)t.col << 32) | (uint64_t)ap;
	}
	s_key[w][lane] = key;
	wave_lds_sync();
	// ---- rank inside the row's S slots (keys are unique), scatter to sorted order
	uint32_t rank = 0;
#pragma unroll 8
	for (int j = 0; j < S; ++j) rank += (s_key[w][g * S + j] < key) ? 1u : 0u;
	if (act) { s_key2[w][g * S + rank] = key; s_val2[w][g * S + rank] = prod; }
	wave_lds_sync();

	// ---- segmented sum in ascending k (sequential, like `sum += a*b`)
	const uint64_t mykey = s_key2[w][lane];
	const bool act2 = mykey != ~0ull;
	const uint32_t mycol = (uint32_t)(mykey >> 32);
	bool head = act2 && (s == 0 || (uint32_t)(s_key2[w][lane - 1] >> 32) != mycol);
	double sum = 0.0;
	if (head) sum += s_val2[w][lane];               // 0 + a*b, as `sum = 0; sum += ...` (multiply_sparse.hpp:219)
	bool more = head;
	for (int t = 1; t < S; ++t) {
		bool cont = false;
		if (more && (int)s + t < S) {
			uint64_t nk = s_key2[w][lane + t];
			cont = nk != ~0ull && (uint32_t)(nk >> 32) == mycol;
		}
		if (!__any(cont)) break;
		if (cont) sum += s_val2[w][lane + t]; else more = false;
	}

	// ---- emit
	const int32_t rowid = has_row ? m.id[r] : 0;
	double value = 0;
	bool out = head && emit_value(ep, row_scale(ep, rowid), (int32_t)mycol, sum, &value);
	uint64_t bal = __ballot(out);
	uint64_t gmask = S == 64 ? bal : ((bal >> (g * S)) & ((1ull << (S & 63)) - 1ull));
	if (MODE == MODE_COUNT) {
		if (has_row && s == 0) sk.segcount[sk.segbase[r]] = (uint32_t)__popcll(gmask);
	} else if (MODE == MODE_STORE) {
		if (has_row) {
			uint32_t seg = sk.segbase[r];
			if (out) {
				uint32_t rk = (uint32_t)__popcll(gmask & ((1ull << s) - 1ull));
				int64_t o = sk.segoff[seg] + rk;
				sk.out_i[o] = rowid; sk.out_j[o] = (int32_t)mycol; sk.out_v[o] = value;
			}
			if (s == 0) sk.segactual[seg] = (uint32_t)__popcll(gmask);
		}
	} else {
		unsigned long long cnt = out ? 1ull : 0ull;
		unsigned long long hash = out ? mix64((uint32_t)rowid, mycol) : 0ull;
		double vs = out ? value : 0.0;
		if (sk.row_nnz) {
			// one wave-group owns the row: reduce inside the S lanes, plain store
			double rs = vs;
#pragma unroll
			for (int d = S / 2; d >= 1; d >>= 1) rs += __shfl_xor(rs, d, 64);
			if (has_row && s == 0) { sk.row_nnz[rowid] = (long long)__popcll(gmask); sk.row_sum[rowid] = rs; }
		}
		d_cnt += cnt; d_hash += hash; d_sum += vs;
	}
	}
	if (MODE == MODE_DIGEST) digest_flush<256>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// ---- all rows light: the direct kernel ---------------------------------------------------------------
// When (longest row of op(A)) x (longest row of op(B)) <= 64 every output row has at most 64 products and the
// whole symbolic phase (per-tuple B row lengths, their prefix, row classes, row lists, the 12-byte B copy) is
// skipped: this kernel walks the dense row pointer of op(A) and reads everything itself.  The regular stencils
// (BASELINE cfg3 and cfg5) are the case it is for.  Per wave G = 64 / S rows, S slots each:
//   A lanes (s < La) read (k, a) and the bounds of B row k; a DPP scan numbers the products of the row;
//   a product's A lane is the running maximum of markers the A lanes drop at their first product's slot (one LDS
//   write per A TUPLE, a DPP max-scan per product) and its (start, offset, a) come over the LDS crossbar (bpermute);
//   products are put in (column, A position) order by rank -- a 32-bit key (column << log2 S | A position) where the
//   column count allows, S broadcast compares -- and summed head by head in that order: ascending k, the order of
//   the reference's `sum += a*b` (multiply_sparse.hpp:219-236), so the values are bit-identical.
template <int S, int MODE, bool K64>
__global__ __launch_bounds__(256, 8) void k_light_direct(uint32_t nrow, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, const int32_t *bcol, const double *bval, EmitParams ep, SinkParams sk, unsigned long long *prod_count)
{
	constexpr int G = 64 / S;
	constexpr int LOGS = S == 8 ? 3 : (S == 16 ? 4 : (S == 32 ? 5 : 6));
	typedef typename std::conditional<K64, uint64_t, uint32_t>::type key_t;
	constexpr key_t NOKEY = (key_t)~(key_t)0;
	__shared__ uint32_t s_mark[4][64];
	__shared__ key_t s_key[4][64];
	__shared__ key_t s_key2[4][64];
	__shared__ double s_val2[4][64];
	__shared__ unsigned long long s_u64[8];
	__shared__ double s_f64[4];

	const unsigned w = wave_id(), lane = lane_id();
	const unsigned g = lane / S, s = lane % S;
	// Element i of an array: in the narrow variant (every operand array below 4 GB, columns below 2^26) the byte offset is
	// 32-bit arithmetic on top of a scalar base -- the 64-bit address computations were 50 of the kernel's 176 vector
	// instructions
	auto at32 = [](const uint32_t *p, uint32_t i) -> uint32_t { if (K64) return p[i]; return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p) + (uint32_t)(i << 2)); };
	auto ati32 = [](const int32_t *p, uint32_t i) -> int32_t { if (K64) return p[i]; return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(p) + (uint32_t)(i << 2)); };
	auto atf64 = [](const double *p, uint32_t i) -> double { if (K64) return p[i]; return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + (uint32_t)(i << 3)); };
	unsigned long long d_cnt = 0, d_hash = 0, n_prod = 0; double d_sum = 0.0;
	const bool plain = ep.C == 1.0 && !ep.si_pos && !ep.sk_pos;
	const uint32_t nvb = (nrow + 4u * G - 1u) / (4u * G);
	// Software pipeline over the row groups of this workgroup: the chain row pointer -> A tuple -> B row bounds ->
	// B tuples is four dependent global loads; its first three links are fetched one link per round ahead
	// (branch-free: clamped indices, results masked), so that a round only waits for its B tuples.
	const uint32_t stride = gridDim.x;
	auto row_of = [&](uint32_t vb_) { return (vb_ * 4u + w) * G + g; };
	auto load_bounds = [&](uint32_t vb_, uint32_t &b_, uint32_t &e_) {
		const uint32_t r_ = row_of(vb_);
		const bool ok = vb_ < nvb && r_ < nrow;
		const uint32_t rc = ok ? r_ : 0u;
		b_ = at32(aptr, rc); e_ = at32(aptr, rc + 1);
		if (ep.si_pos) {                                             // scalei: absent or zero -> the row is skipped
			const int32_t q = ep.si_pos[rc];
			if (q < 0 || ep.si_val[q] == 0) e_ = b_;
		}
		if (!ok) e_ = b_;
	};
	auto load_tuple = [&](uint32_t b_, uint32_t e_, int32_t &k_, double &a_, bool &v_) {
		const uint32_t e = b_ + s;
		v_ = e < e_;
		const uint32_t ec = v_ ? e : (b_ < e_ ? b_ : 0u);                // any valid tuple (A has at least one)
		k_ = ati32(acol, ec); a_ = atf64(aval, ec);
	};
	auto load_brow = [&](int32_t k_, bool v_, uint32_t &lo_, uint32_t &len_) {
		const uint32_t l0 = at32(bptr, (uint32_t)k_), l1 = at32(bptr, (uint32_t)k_ + 1u);
		lo_ = l0; len_ = v_ ? l1 - l0 : 0u;
	};
	uint32_t beg1, end1, beg2, end2;                                // bounds of round +1, +2
	int32_t k1; double a1; bool v1;                                 // A tuple of round +1
	uint32_t lo0, len0; double a0;                                  // B row bounds of this round
	{
		uint32_t b0, e0; int32_t k0; bool v0;
		load_bounds(blockIdx.x, b0, e0);
		load_bounds(blockIdx.x + stride, beg1, end1);
		load_tuple(b0, e0, k0, a0, v0);
		load_brow(k0, v0, lo0, len0);
		load_tuple(beg1, end1, k1, a1, v1);
	}
	for (uint32_t vb = blockIdx.x; vb < nvb; vb += stride) {
		const uint32_t r = row_of(vb);
		const bool has_row = r < nrow;
		const uint32_t lo = lo0, len = len0; const double a = a0;
		// prefetches for the next rounds (consumed after this round's work)
		load_bounds(vb + 2 * stride, beg2, end2);
		uint32_t nlo, nlen;
		load_brow(k1, v1, nlo, nlen);
		const double na = a1;
		int32_t k2; double a2; bool v2;
		const uint32_t inc = group_inclusive_scan_u32<S>(len, s);
		const uint32_t ex = inc - len;
		const uint32_t P = (uint32_t)__shfl((int)inc, (int)(g * S + S - 1), 64);       // products of the row (<= S)
		// ---- product slot -> its A lane: markers + running maximum
		s_mark[w][lane] = 0u;
		if (len) s_mark[w][g * S + ex] = s + 1u;                     // (LDS traffic of one wave is in order)
		wave_lds_sync();
		uint32_t mk = s_mark[w][lane];
		wave_lds_sync();
		{
			int x = (int)mk, t;
			t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); if (s >= 1) x = max(x, t);
			t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); if (s >= 2) x = max(x, t);
			t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); if (s >= 4) x = max(x, t);
			if (S >= 16) { t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); if (s >= 8) x = max(x, t); }
			if (S >= 32) { t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); if (s >= 16) x = max(x, t); }
			if (S >= 64) { t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); if (s >= 32) x = max(x, t); }
			mk = (uint32_t)x;
		}
		const bool act = s < P;                                      // then mk >= 1
		const int src = (int)(g * S + (mk ? mk - 1u : 0u));
		const uint32_t slo = (uint32_t)__shfl((int)lo, src, 64), sex = (uint32_t)__shfl((int)ex, src, 64);
		const double sa = __shfl(a, src, 64);
		// ---- product, key (column, A position)
		key_t key = NOKEY;
		double prod = 0.0;
		if (act) {
			const uint32_t bp = slo + (s - sex);
			const int32_t col = ati32(bcol, bp);
			prod = sa * atf64(bval, bp);
			key = (key_t)(((key_t)(uint32_t)col << LOGS) | (key_t)(mk - 1u));
		}
		s_key[w][lane] = key;
		s_key2[w][lane] = NOKEY;
		wave_lds_sync();
		// ---- rank inside the row's S slots (keys are unique), scatter to sorted order
		uint32_t rank = 0;
#pragma unroll
		for (int j = 0; j < S; ++j) rank += (s_key[w][g * S + j] < key) ? 1u : 0u;
		if (act) { s_key2[w][g * S + rank] = key; s_val2[w][g * S + rank] = prod; }
		wave_lds_sync();
		// ---- segmented sum in ascending k (sequential, like `sum += a*b`)
		const key_t mykey = s_key2[w][lane];
		const bool act2 = mykey != NOKEY;
		const uint32_t mycol = (uint32_t)(mykey >> LOGS);
		const bool head = act2 && (s == 0 || (uint32_t)(s_key2[w][lane - 1] >> LOGS) != mycol);
		double sum = 0.0;
		if (head) sum += s_val2[w][lane];                            // 0 + a*b, as `sum = 0; sum += ...` (multiply_sparse.hpp:219)
		bool more = head;
		for (int t = 1; t < S; ++t) {
			bool cont = false;
			if (more && (int)s + t < S) {
				const key_t nk = s_key2[w][lane + t];
				cont = nk != NOKEY && (uint32_t)(nk >> LOGS) == mycol;
			}
			if (!__any(cont)) break;
			if (cont) sum += s_val2[w][lane + t]; else more = false;
		}
		wave_lds_sync();                                             // the next round overwrites the arrays
		// ---- emit
		const int32_t rowid = (int32_t)r;
		double value = 0;
		bool out;
		if (plain) { value = sum; out = head && sum != 0; }          // sum * 1 * 1 * 1 is the same bits (multiply_sparse.hpp:242)
		else out = head && emit_value(ep, row_scale(ep, rowid), (int32_t)mycol, sum, &value);
		const uint64_t bal = __ballot(out);
		const uint64_t gmask = S == 64 ? bal : ((bal >> (g * S)) & ((1ull << (S & 63)) - 1ull));
		if (s == 0) n_prod += P;
		if (MODE == MODE_COUNT) {
			if (has_row && s == 0) sk.segcount[r] = (uint32_t)__popcll(gmask);
		} else if (MODE == MODE_STORE) {
			if (has_row) {
				if (out) {
					const uint32_t rk = (uint32_t)__popcll(gmask & ((1ull << s) - 1ull));
					const int64_t o = sk.segoff[r] + rk;
					sk.out_i[o] = rowid; sk.out_j[o] = (int32_t)mycol; sk.out_v[o] = value;
				}
				if (s == 0) sk.segactual[r] = (uint32_t)__popcll(gmask);
			}
		} else {
			if (sk.row_nnz) {
				double rs = out ? value : 0.0;
#pragma unroll
				for (int d = S / 2; d >= 1; d >>= 1) rs += __shfl_xor(rs, d, 64);
				if (has_row && s == 0) { sk.row_nnz[rowid] = (long long)__popcll(gmask); sk.row_sum[rowid] = rs; }
			}
			if (out) { ++d_cnt; d_hash += mix64((uint32_t)rowid, mycol); d_sum += value; }
		}
		// rotate the pipeline: the A tuple of round +2 needs the bounds loaded at the top of THIS round
		load_tuple(beg2, end2, k2, a2, v2);
		lo0 = nlo; len0 = nlen; a0 = na;
		k1 = k2; a1 = a2; v1 = v2;
	}
	n_prod = wave_reduce_sum(n_prod);
	if (lane == 0 && n_prod) atomicAdd(prod_count, n_prod);
	if (MODE == MODE_DIGEST) digest_flush<256>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// longest row of a dense row pointer
__global__ void k_max_rowlen(const uint32_t *ptr, uint64_t nrow, uint32_t *out)
{
	uint32_t v = 0;
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (uint64_t)gridDim.x * blockDim.x) v = max(v, ptr[r + 1] - ptr[r]);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
	// (same-address atomics serialise: only a wave that would raise the maximum issues one)
	if (lane_id() == 0 && v > *(volatile uint32_t *)out) atomicMax(out, v);
}

// ====================================================================== hash cells (LDS hash accumulator)

// A cell is the unit of numeric work above the light bin: one output row
// restricted to a range [wa, wb) of column windows.
//   mid rows   (P_r <= 4096): one cell = the whole row (no window index needed)
//   heavy rows (P_r >  4096): consecutive windows are grouped greedily into
//       hash cells of <= 4096 products; a single window holding more than that
//       becomes a dense cell (k_dense).
// A cell is one output segment of the COO sink (cells of a row in window order).
struct Cell {
	uint32_t beg, end; // the row's A tuples
	int32_t rowid;     // row index of op(A)
	uint32_t seg;      // output segment id (COO sink)
	uint32_t prods;    // scalar products in the cell
	uint16_t wa, wb;   // window range
	uint32_t pad[2];
};

// Flattened product loop.  A chunk of NT A-tuples selects NT B segments
// (start, length); the scalar products of the chunk are numbered 0..total-1
// and dealt to the threads 64 consecutive products per wave, so consecutive
// lanes read consecutive B tuples of a segment (coalesced).  Finding the
// segment of product p costs no search: the segments with length > 0 are
// compacted, every such segment sets one bit (its first product) in a 64-bit
// mask per 64-product block, and lane j takes
//     q = bq[block] + popcount(mask[block] & bits(1..j))
// where bq[block] is the segment of the block's first product.
// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does
// not drain the wave's outstanding global loads (vmcnt), so prefetched operands
// stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, int PB>
struct Expand {
	uint32_t cpref[NT + 1];          // compacted segments: exclusive product prefix (+ total)
	uint32_t cstart[NT];             // first B tuple of the segment
	double caval[NT];                // the A value
	unsigned long long bmask[PB / 64];
	uint32_t bq[PB / 64];
	uint32_t scrL[2][NT / 64], scrN[2][NT / 64];    // per-wave totals, double buffered
};

// Segment (lo, len, a) of this thread's A tuple -> compacted arrays.  Returns
// the product total of the chunk and the number of non-empty segments.  One
// barrier inside; the arrays become visible at expand_batch's first barrier.
template <int NT, int PB>
__device__ __forceinline__ void expand_load(Expand<NT, PB> &L, uint32_t lo, uint32_t len, double a, uint32_t *total, uint32_t *nzc,
	uint32_t &flip, uint32_t *ex_out)
{
	constexpr int NW = NT / 64;
	const uint32_t inc = wave_inclusive_scan_u32(len);
	const uint64_t nzm = __ballot(len != 0);
	const uint32_t wrank = (uint32_t)__popcll(nzm & lanemask_lt());
	if (lane_id() == 63) L.scrL[flip][wave_id()] = inc;
	if (lane_id() == 0) L.scrN[flip][wave_id()] = (uint32_t)__popcll(nzm);
	lds_barrier();
	uint32_t baseL = 0, baseN = 0, totL = 0, totN = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		uint32_t l = L.scrL[flip][w], n = L.scrN[flip][w];
		if (w < (int)wave_id()) { baseL += l; baseN += n; }
		totL += l; totN += n;
	}
	flip ^= 1u;
	*ex_out = baseL + inc - len;                 // exclusive product prefix of this thread's segment
	if (len) {
		uint32_t rank = baseN + wrank;
		L.cpref[rank] = baseL + inc - len;
		L.cstart[rank] = lo;
		L.caval[rank] = a;
	}
	if (threadIdx.x == 0) L.cpref[totN] = totL;
	*total = totL;
	*nzc = totN;
}

// Prepare the lookup tables for products [pb, pe), pe - pb <= PB.  Two barriers.
template <int NT, int PB>
__device__ __forceinline__ void expand_batch(Expand<NT, PB> &L, uint32_t pb, uint32_t pe, uint32_t nzc)
{
	const uint32_t nblk = (pe - pb + 63) >> 6;
	for (uint32_t b = threadIdx.x; b < nblk; b += NT) L.bmask[b] = 0;
	lds_barrier();
	for (uint32_t b = threadIdx.x; b < nblk; b += NT) {
		// segment holding the block's first product: largest q with cpref[q] <= p
		uint32_t p = pb + (b << 6), lo = 0, hi = nzc - 1;
		while (hi > lo) {
			uint32_t mid = (lo + hi + 1) >> 1;
			if (L.cpref[mid] <= p) lo = mid; else hi = mid - 1;
		}
		L.bq[b] = lo;
	}
	for (uint32_t i = threadIdx.x; i < nzc; i += NT) {
		uint32_t s = L.cpref[i];
		if (s > pb && s < pe && ((s - pb) & 63u)) atomicOr(&L.bmask[(s - pb) >> 6], 1ull << ((s - pb) & 63u));
	}
	lds_barrier();
}

template <int NT, int PB>
__device__ __forceinline__ uint32_t expand_lookup(const Expand<NT, PB> &L, uint32_t p, uint32_t pb)
{
	const uint32_t b = (p - pb) >> 6, j = (p - pb) & 63u;
	return L.bq[b] + (uint32_t)__popcll(L.bmask[b] & ((2ull << j) - 1ull));
}


// ---- the same flattening in units of ITEMS of R consecutive B tuples (dense cells) ----------
// A segment of `len` tuples is ceil(len / R) items; an item never crosses a segment, so ONE lookup
// (item -> segment) serves R products: the lane then reads its R tuples as one contiguous 12 R-byte
// piece and masks the tail of the segment's last item.  Padding costs one partial item per segment
// (dense cells average 27 tuples per non-empty segment); the lookup's LDS reads and popcount
// arithmetic, which co-limit the loop with the LDS accumulate, are paid once per R products.
#ifndef DENSE_R_V
#define DENSE_R_V 4
#endif
#ifndef DENSE_DEPTH
#define DENSE_DEPTH 1
#endif
constexpr int DENSE_R = DENSE_R_V;

template <int NT, int PB>
struct ExpandR {
	uint32_t cpref[NT + 1];          // compacted segments: exclusive ITEM prefix (+ total)
	uint2 cse[NT];                   // first tuple of the segment, one past its last
	double caval[NT];                // the A value
	unsigned long long bmask[PB / 64];
	uint32_t bq[PB / 64];
	uint32_t scrL[2][NT / 64], scrN[2][NT / 64];
};

template <int NT, int PB>
__device__ __forceinline__ void expandr_load(ExpandR<NT, PB> &L, uint32_t lo, uint32_t len, double a, uint32_t *total, uint32_t *nzc,
	uint32_t &flip)
{
	constexpr int NW = NT / 64;
	const uint32_t items = (len + DENSE_R - 1) / DENSE_R;
	const uint32_t inc = wave_inclusive_scan_u32(items);
	const uint64_t nzm = __ballot(len != 0);
	const uint32_t wrank = (uint32_t)__popcll(nzm & lanemask_lt());
	if (lane_id() == 63) L.scrL[flip][wave_id()] = inc;
	if (lane_id() == 0) L.scrN[flip][wave_id()] = (uint32_t)__popcll(nzm);
	lds_barrier();
	uint32_t baseL = 0, baseN = 0, totL = 0, totN = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		uint32_t l = L.scrL[flip][w], n = L.scrN[flip][w];
		if (w < (int)wave_id()) { baseL += l; baseN += n; }
		totL += l; totN += n;
	}
	flip ^= 1u;
	if (len) {
		uint32_t rank = baseN + wrank;
		L.cpref[rank] = baseL + inc - items;
		L.cse[rank] = make_uint2(lo, lo + len);
		L.caval[rank] = a;
	}
	if (threadIdx.x == 0) L.cpref[totN] = totL;
	*total = totL;
	*nzc = totN;
}

// Lookup tables for items [pb, pe), pe - pb <= PB.  Two barriers.
template <int NT, int PB>
__device__ __forceinline__ void expandr_batch(ExpandR<NT, PB> &L, uint32_t pb, uint32_t pe, uint32_t nzc)
{
	const uint32_t nblk = (pe - pb + 63) >> 6;
	for (uint32_t b = threadIdx.x; b < nblk; b += NT) L.bmask[b] = 0;
	lds_barrier();
	for (uint32_t b = threadIdx.x; b < nblk; b += NT) {
		uint32_t p = pb + (b << 6), lo = 0, hi = nzc - 1;
		while (hi > lo) {
			uint32_t mid = (lo + hi + 1) >> 1;
			if (L.cpref[mid] <= p) lo = mid; else hi = mid - 1;
		}
		L.bq[b] = lo;
	}
	for (uint32_t i = threadIdx.x; i < nzc; i += NT) {
		uint32_t s = L.cpref[i];
		if (s > pb && s < pe && ((s - pb) & 63u)) atomicOr(&L.bmask[(s - pb) >> 6], 1ull << ((s - pb) & 63u));
	}
	lds_barrier();
}

// R consecutive B tuples as the loop reads them: 12 R bytes at a 4-byte aligned address.
struct __attribute__((packed, aligned(4))) BPiece { uint32_t w[3 * DENSE_R]; };


// XCD-aware walk of a cell list.  Workgroups are dispatched round-robin over the 8 XCDs
// (blockIdx % 8 names the group of blocks that share an XCD and its L2).  The list, which is
// in window-major order, is cut into 8 contiguous parts of equal cost (xb[0..8]); XCD group
// x walks part x, so each L2 holds the B column-window slice of ITS part only instead of all
// eight L2s fetching the same slice.  Speed only: any placement gives the same result.
struct CellWalk { uint32_t first, end, stride; };
__device__ __forceinline__ CellWalk cell_walk(const uint32_t *xb, uint32_t ncell)
{
	CellWalk w;
	if (xb && (gridDim.x & 7u) == 0) {
		uint32_t x = blockIdx.x & 7u;
		w.first = xb[x] + (blockIdx.x >> 3);
		w.end = xb[x + 1];
		w.stride = gridDim.x >> 3;
	} else {
		w.first = blockIdx.x; w.end = ncell; w.stride = gridDim.x;
	}
	return w;
}

__global__ void k_cell_cost(const Cell *cells, uint32_t n, uint32_t fixed, uint32_t *cost)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cost[i] = cells[i].prods + fixed;
}

__global__ void k_xcd_bounds(const int64_t *pref, uint32_t n, uint32_t *xb)
{
	// xb[x] = first cell whose cost prefix reaches x/8 of the total
	uint32_t x = threadIdx.x;
	if (x > 8) return;
	int64_t target = pref[n] / 8 * x;
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (pref[mid] < target) lo = mid + 1; else hi = mid; }
	xb[x] = x == 8 ? n : lo;
}

// Cells for the mid rows (P_r <= 4096): the whole row, no window index.
__global__ void k_row_cells(const uint32_t *binrows, uint32_t n, const uint32_t *rbeg, const int32_t *rid, const uint32_t *rprod,
	const uint32_t *segbase, Cell *cells)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t r = binrows[i];
	Cell c;
	c.beg = rbeg[r]; c.end = rbeg[r + 1]; c.rowid = rid[r]; c.seg = segbase ? segbase[r] : 0; c.prods = rprod[r];
	c.wa = c.wb = 0; c.pad[0] = c.pad[1] = 0;
	cells[i] = c;
}

// One lane's LDS fetch-add, spelled as the instruction: the compiler's atomic optimiser otherwise wraps
// the (already wave-aggregated) add into another mbcnt / readfirstlane / multiply sequence.
__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t *p, uint32_t v)
{
	uint32_t r;
	const uint32_t a = (uint32_t)(uintptr_t)p;       // LDS byte offset = low half of the flat address of a __shared__ object
	asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory");
	return r;
}

// ---- LDS hash accumulator shared by k_hash and k_hash_tiles -----------------------------
// T slots (power of two or 3072), at most T/2 products per cell.  The table is cleaned as it
// is emitted (list of occupied slots), so a cell costs work proportional to its products.

template <int T>
__device__ __forceinline__ uint32_t hash_slot(int32_t col)
{
	constexpr int LOGT = T == 1024 ? 10 : (T == 2048 ? 11 : (T == 4096 ? 12 : 13));   // power-of-two T only
	if constexpr ((T & (T - 1)) == 0) return ((uint32_t)col * 0x9E3779B1u) >> (32 - LOGT);
	else return (uint32_t)(((uint64_t)((uint32_t)col * 0x9E3779B1u) * (uint64_t)T) >> 32);   // multiply-shift into [0, T)
}

// Products [p0, p1) of the prepared batch starting at pb -> table.  U products per thread and
// step: all B loads of a step are issued before the first insertion.
template <int T, int NT, int PB, int MODE, bool PAT>
__device__ __forceinline__ void hash_products(const Expand<NT, PB> &X, uint32_t p0, uint32_t p1, uint32_t pb, const RowMeta &m,
	int32_t *h_key, double *h_val, uint16_t *occ, uint32_t *s_nocc, PatAcc &pat)
{
	constexpr int U = HASH_U;
	const unsigned tid = threadIdx.x;
	for (uint32_t pbase = p0; pbase < p1; pbase += NT * U) {
		int32_t col[U]; double pv[U]; bool ok[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			uint32_t p = pbase + u * NT + tid;
			ok[u] = p < p1;
			p = ok[u] ? p : p1 - 1;
			uint32_t q = expand_lookup(X, p, pb);
			uint32_t bp = X.cstart[q] + (p - X.cpref[q]);
			BTup t;
			if (ABLG(0x2000)) { t.col = (int32_t)((p * 2654435761u) >> 12); t.vlo = 0; t.vhi = 0x3FF00000u; }     // no B read
			else t = m.btup[bp];
			col[u] = t.col;
			pv[u] = (MODE != MODE_COUNT) ? X.caval[q] * btup_val(t) : 0.0;
		}
		if (ABLG(0x1000)) { bool any = false; for (int u = 0; u < U; ++u) any |= (pv[u] == 1.2345e-300); if (any) h_val[0] = 1.0; continue; }   // no insertion
		uint32_t slot_of[U]; uint64_t newmask[U]; uint32_t nnew = 0;
#pragma unroll
		for (int u = 0; u < U; ++u) {
			bool isnew = false;
			uint32_t h = 0;
			if (ok[u]) {
				h = hash_slot<T>(col[u]);
				for (;;) {
					int32_t old = atomicCAS(&h_key[h], -1, col[u]);
					if (old == -1) { isnew = true; break; }
					if (old == col[u]) break;
					if constexpr ((T & (T - 1)) == 0) h = (h + 1) & (T - 1);
					else h = h + 1 == (uint32_t)T ? 0u : h + 1;
				}
				if (MODE != MODE_COUNT) { atomicAdd(&h_val[h], pv[u]); if (PAT) pat_note(pat, pv[u]); }
			}
			slot_of[u] = h;
			newmask[u] = __ballot(isnew);
			nnew += (uint32_t)__popcll(newmask[u]);
		}
		// append the newly occupied slots of the whole step: one LDS atomic per wave and step
		if (nnew && !ABLG(0x4000)) {                                        // uniform
			uint32_t base = 0;
			if (lane_id() == 0) base = lds_add_rtn_u32(s_nocc, nnew);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if ((newmask[u] >> lane_id()) & 1ull) occ[base + __popcll(newmask[u] & lanemask_lt())] = (uint16_t)slot_of[u];
				base += (uint32_t)__popcll(newmask[u]);
			}
		}
	}
}

// SPSAMD_SINK_ORDERED: the segments [q0, q1) of the prepared chunk one after the other, in
// ascending k.  Inside one segment the columns are unique (B is consolidated), so the threads
// update distinct slots with plain read-modify-writes; a barrier separates the segments.  Every
// sum is then accumulated exactly like the reference's `sum += a*b` loop (multiply_sparse.hpp:
// 219-236): bit-identical values and the same exact-zero drops, at the price of one barrier per
// A tuple.
template <int T, int NT, int PB, int MODE>
__device__ __forceinline__ void hash_products_ordered(const Expand<NT, PB> &X, uint32_t q0, uint32_t q1, const RowMeta &m,
	int32_t *h_key, double *h_val, uint16_t *occ, uint32_t *s_nocc)
{
	const unsigned tid = threadIdx.x;
	for (uint32_t q = q0; q < q1; ++q) {
		const uint32_t len = X.cpref[q + 1] - X.cpref[q], start = X.cstart[q];
		const double a = X.caval[q];
		for (uint32_t base = 0; base < len; base += NT) {
			const uint32_t t = base + tid;
			bool isnew = false;
			uint32_t h = 0;
			if (t < len) {
				const BTup bt = m.btup[start + t];
				h = hash_slot<T>(bt.col);
				for (;;) {
					int32_t old = atomicCAS(&h_key[h], -1, bt.col);
					if (old == -1) { isnew = true; break; }
					if (old == bt.col) break;
					if constexpr ((T & (T - 1)) == 0) h = (h + 1) & (T - 1);
					else h = h + 1 == (uint32_t)T ? 0u : h + 1;
				}
				if (MODE != MODE_COUNT) h_val[h] = h_val[h] + a * btup_val(bt);
			}
			uint64_t nm = __ballot(isnew);
			if (nm) {
				uint32_t b = 0;
				if (lane_id() == 0) b = atomicAdd(s_nocc, (uint32_t)__popcll(nm));
				b = (uint32_t)__shfl((int)b, 0, 64);
				if (isnew) occ[b + __popcll(nm & lanemask_lt())] = (uint16_t)h;
			}
		}
		lds_barrier();
	}
}

// Stable LSD radix sort (4-bit digits) of n packed 32-bit keys in LDS on the bits
// [lowbit, lowbit + nbits).  a holds the keys, b is scratch of the same size; returns the array
// that holds the sorted keys.  Element i = r*NT + tid belongs to (round r, wave, lane); a key's
// position is (keys with a smaller digit) + (same digit in an earlier round / wave) + (same digit
// in a lower lane): one 16-bit counter per (digit, round, wave), filled by wave ballots, scanned
// once per pass.  Replaces a bitonic network of 66 barrier-separated stages for 2048 keys.
template <int NT, int EMAX>
__device__ __forceinline__ uint32_t *lds_radix_sort(uint32_t *a, uint32_t *b, uint32_t n, uint32_t lowbit, uint32_t nbits,
	uint16_t *cnt, uint32_t *scr32)
{
	constexpr int NW = NT / 64;
	constexpr int NC = 16 * EMAX * NW;
	constexpr int PER = (NC + NT - 1) / NT;
	const unsigned tid = threadIdx.x, wv = wave_id();
	const uint32_t rounds = (n + NT - 1) / NT;
	if (n <= 1) return a;                                               // uniform
	for (uint32_t shift = lowbit; shift < lowbit + nbits; shift += 4) {
		for (int q = tid; q < NC; q += NT) cnt[q] = 0;
		__syncthreads();
		uint32_t key[EMAX], where[EMAX];
#pragma unroll
		for (int r = 0; r < EMAX; ++r) {
			key[r] = 0; where[r] = 0xFFFFFFFFu;
			if ((uint32_t)r < rounds) {                                    // uniform
				const uint32_t i = r * NT + tid;
				const bool ok = i < n;
				const uint32_t k = ok ? a[i] : 0u;
				const uint32_t d = (k >> shift) & 15u;
				uint64_t m = __ballot(ok);
#pragma unroll
				for (int bit = 0; bit < 4; ++bit) {
					const bool set = (d >> bit) & 1u;
					const uint64_t bm = __ballot(set);
					m &= set ? bm : ~bm;
				}
				const uint32_t before = (uint32_t)__popcll(m & lanemask_lt());
				const uint32_t slot = (d * EMAX + r) * NW + wv;
				if (ok && before == 0) cnt[slot] = (uint16_t)__popcll(m);
				key[r] = k;
				if (ok) where[r] = (slot << 8) | before;                   // before < 64
			}
		}
		__syncthreads();
		{
			uint32_t loc[PER], sum = 0;
#pragma unroll
			for (int q = 0; q < PER; ++q) { const int e = tid * PER + q; loc[q] = e < NC ? cnt[e] : 0u; sum += loc[q]; }
			uint32_t ex = block_exclusive_scan<uint32_t, NT>(sum, scr32, (uint32_t *)nullptr);
#pragma unroll
			for (int q = 0; q < PER; ++q) { const int e = tid * PER + q; if (e < NC) cnt[e] = (uint16_t)ex; ex += loc[q]; }
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < EMAX; ++r)
			if (where[r] != 0xFFFFFFFFu) b[cnt[where[r] >> 8] + (where[r] & 63u)] = key[r];
		__syncthreads();
		uint32_t *t = a; a = b; b = t;
	}
	return a;
}

struct DigestAcc { unsigned long long cnt, hash; double sum; };

// Emit the occupied slots of the finished cell into the sink and clean them.
template <int T, int NT, int MODE, bool PAT>
__device__ __forceinline__ void hash_emit(uint32_t nocc, int32_t rowid, uint32_t seg, const EmitParams &ep, const SinkParams &sk,
	int32_t *h_key, double *h_val, const uint16_t *occ, uint64_t *s_sort, uint32_t *scr32, DigestAcc &d,
	uint16_t *s_cnt, uint32_t colbase, uint32_t colbits, const RowMeta &m, uint32_t pbeg, uint32_t pend, double pthr)
{
	const unsigned tid = threadIdx.x;
	const double a_scale = row_scale(ep, rowid);
	if (MODE == MODE_COUNT) {
		// structural count (an upper bound when sums cancel to exactly 0)
		uint32_t c = 0;
		for (uint32_t i = tid; i < nocc; i += NT) { uint32_t h = occ[i]; if (col_allowed(ep, h_key[h])) ++c; h_key[h] = -1; }
		uint32_t total;
		block_exclusive_scan<uint32_t, NT>(c, scr32, &total);
		if (tid == 0) sk.segcount[seg] = total;
	} else if (MODE == MODE_DIGEST) {
		unsigned long long cnt = 0; double vs = 0;
		for (uint32_t base = 0; base < nocc; base += NT) {                 // (uniform trips: pat_fix_wave wants whole waves)
			const uint32_t i = base + tid;
			const bool valid = i < nocc;
			const uint32_t h = valid ? occ[i] : 0u;
			const int32_t col = valid ? h_key[h] : 0;
			double v;
			double x = valid ? h_val[h] : 0.0;
			if (PAT) x = pat_fix_wave(valid && !(fabs(x) > pthr), x, col, m, pbeg, pend);
			if (valid) {
				if (emit_value(ep, a_scale, col, x, &v)) { ++cnt; d.hash += mix64((uint32_t)rowid, (uint32_t)col); vs += v; }
				h_key[h] = -1; h_val[h] = 0.0;
			}
		}
		d.cnt += cnt; d.sum += vs;
		if (sk.row_nnz) {
			unsigned long long rc = wave_reduce_sum(cnt); double rs = wave_reduce_sum(vs);
			if (lane_id() == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); }
		}
	} else {
		// surviving columns -> sorted -> emitted in order, cleaning the table.  Where the cell's column
		// range and the position in the occupied list fit one 32-bit word (they do for every cell of a
		// matrix with up to 2^20 columns) the keys are radix sorted, otherwise by the bitonic network.
		constexpr uint32_t PBITS = T <= 1024 ? 9 : (T <= 4096 ? 11 : 12);          // position in occ[] (< T/2)
		constexpr int EMAX = (T / 2 + NT - 1) / NT;
		// Narrow cells (column range <= 32*T bits, e.g. 16 windows of 8192 for T = 4096): no sort at all.
		// The surviving columns set bits in a bitmap laid over s_sort; the rank of a column is the number
		// of bits below it = prefix count of its 4-word superblock (s_cnt) + popcounts of at most three
		// words + its own word below the bit, and the tuple is stored straight at segoff + rank.
		const uint32_t colrange = colbits >= 32 ? 0xFFFFFFFFu : (1u << colbits);
		constexpr int NSB = T / 8;                                         // superblocks of 4 words, one per thread in the scan
		if (NSB <= NT && colrange <= (uint32_t)T * 32u && ep.emit_path < 1) {
			static_assert(NSB <= 16 * EMAX * (NT / 64), "s_cnt holds the superblock prefixes");
			unsigned long long *bm = (unsigned long long *)s_sort;
			const uint32_t nwords = (colrange + 63u) >> 6, nsb = (nwords + 3u) >> 2;
			for (uint32_t w = tid; w < nsb * 4u; w += NT) bm[w] = 0ull;
			lds_barrier();
			uint32_t rel[EMAX], slot[EMAX];
#pragma unroll
			for (int r = 0; r < EMAX; ++r) {
				rel[r] = 0xFFFFFFFFu; slot[r] = 0;
				const uint32_t i = r * NT + tid;
				const bool valid = i < nocc;
				const uint32_t h = valid ? occ[i] : 0u;
				const int32_t col = valid ? h_key[h] : 0;
				double x = valid ? h_val[h] : 0.0;
				if (PAT) x = pat_fix_wave(valid && !(fabs(x) > pthr), x, col, m, pbeg, pend);
				if (valid) {
					double v = 0;
					const bool ok = emit_value(ep, a_scale, col, x, &v);
					h_key[h] = -1;
					h_val[h] = ok ? v : 0.0;
					if (ok) {
						rel[r] = (uint32_t)col - colbase; slot[r] = h;
						atomicOr(&bm[rel[r] >> 6], 1ull << (rel[r] & 63u));
					}
				}
			}
			lds_barrier();
			uint32_t c4 = 0;
			if (tid < nsb) c4 = (uint32_t)(__popcll(bm[4 * tid]) + __popcll(bm[4 * tid + 1]) + __popcll(bm[4 * tid + 2]) + __popcll(bm[4 * tid + 3]));
			uint32_t mcount = 0, ex = 0;
			ex = block_exclusive_scan<uint32_t, NT>(c4, scr32, &mcount);
			if (tid < nsb) s_cnt[tid] = (uint16_t)ex;
			lds_barrier();
			const int64_t o = sk.segoff[seg];
#pragma unroll
			for (int r = 0; r < EMAX; ++r) {
				if (rel[r] != 0xFFFFFFFFu) {
					const uint32_t w = rel[r] >> 6, sb = w >> 2;
					uint32_t rank = s_cnt[sb] + (uint32_t)__popcll(bm[w] & ((1ull << (rel[r] & 63u)) - 1ull));
					for (uint32_t q = sb * 4u; q < w; ++q) rank += (uint32_t)__popcll(bm[q]);
					sk.out_i[o + rank] = rowid;
					sk.out_j[o + rank] = (int32_t)(colbase + rel[r]);
					sk.out_v[o + rank] = h_val[slot[r]];
					h_val[slot[r]] = 0.0;
				}
			}
			if (tid == 0) sk.segactual[seg] = mcount;
			return;
		}
		if (colbits + PBITS <= 32 && ep.emit_path < 2) {
			uint32_t *ka = (uint32_t *)s_sort, *kb = ka + T / 2;
			uint32_t run = 0;
			for (uint32_t base = 0; base < nocc; base += NT) {
				uint32_t i = base + tid;
				bool ok = false;
				const bool valid = i < nocc;
				const uint32_t h = valid ? occ[i] : 0u;
				const int32_t col = valid ? h_key[h] : 0;
				double x = valid ? h_val[h] : 0.0;
				if (PAT) x = pat_fix_wave(valid && !(fabs(x) > pthr), x, col, m, pbeg, pend);
				if (valid) {
					double v = 0;
					ok = emit_value(ep, a_scale, col, x, &v);
					h_key[h] = -1;
					h_val[h] = ok ? v : 0.0;
				}
				uint32_t total;
				uint32_t ex = block_exclusive_scan<uint32_t, NT>(ok ? 1u : 0u, scr32, &total);
				if (ok) ka[run + ex] = (((uint32_t)col - colbase) << PBITS) | i;
				run += total;
			}
			const uint32_t mcount = run;
			uint32_t *srt = lds_radix_sort<NT, EMAX>(ka, kb, mcount, PBITS, colbits, s_cnt, scr32);
			int64_t o = sk.segoff[seg];
			for (uint32_t i = tid; i < mcount; i += NT) {
				uint32_t kq = srt[i];
				uint32_t h = occ[kq & ((1u << PBITS) - 1u)];
				sk.out_i[o + i] = rowid;
				sk.out_j[o + i] = (int32_t)(colbase + (kq >> PBITS));
				sk.out_v[o + i] = h_val[h];
				h_val[h] = 0.0;
			}
			if (tid == 0) sk.segactual[seg] = mcount;
			return;
		}
		// surviving (col, slot) pairs -> bitonic sort by column -> emit in order, cleaning the table
		uint32_t run = 0;
		for (uint32_t base = 0; base < nocc; base += NT) {
			uint32_t i = base + tid;
			bool ok = false;
			const bool valid = i < nocc;
			const uint32_t h = valid ? occ[i] : 0u;
			const int32_t col = valid ? h_key[h] : 0;
			double x = valid ? h_val[h] : 0.0;
			if (PAT) x = pat_fix_wave(valid && !(fabs(x) > pthr), x, col, m, pbeg, pend);
			if (valid) {
				double v = 0;
				ok = emit_value(ep, a_scale, col, x, &v);
				h_key[h] = -1;
				h_val[h] = ok ? v : 0.0;
			}
			uint32_t total;
			uint32_t ex = block_exclusive_scan<uint32_t, NT>(ok ? 1u : 0u, scr32, &total);
			if (ok) s_sort[run + ex] = ((uint64_t)(uint32_t)col << 16) | (uint64_t)h;
			run += total;
		}
		const uint32_t mcount = run;
		uint32_t n2 = 1;
		while (n2 < mcount) n2 <<= 1;
		for (uint32_t q = mcount + tid; q < n2; q += NT) s_sort[q] = ~0ull;
		__syncthreads();
		for (uint32_t k = 2; k <= n2 && !ABL(ep, 256); k <<= 1) {
			for (uint32_t j = k >> 1; j > 0; j >>= 1) {
				for (uint32_t i = tid; i < n2; i += NT) {
					uint32_t ixj = i ^ j;
					if (ixj > i) {
						uint64_t x = s_sort[i], y = s_sort[ixj];
						bool up = (i & k) == 0;
						if ((x > y) == up) { s_sort[i] = y; s_sort[ixj] = x; }
					}
				}
				__syncthreads();
			}
		}
		int64_t o = sk.segoff[seg];
		for (uint32_t i = tid; i < mcount; i += NT) {
			uint64_t kq = s_sort[i];
			uint32_t h = (uint32_t)(kq & 0xFFFFu);
			sk.out_i[o + i] = rowid;
			sk.out_j[o + i] = (int32_t)(kq >> 16);
			sk.out_v[o + i] = h_val[h];
			h_val[h] = 0.0;
		}
		if (tid == 0) sk.segactual[seg] = mcount;
	}
}

// Persistent workgroups walk the cell list with a grid stride (the list is in window-major
// order, so concurrently processed cells read the same column windows of B).
template <int T, int NT, int MODE, bool WINDOWED, bool PAT>
__global__ __launch_bounds__(NT) void k_hash(const Cell *cells, uint32_t ncell, const uint32_t *xb, RowMeta m,
	const uint32_t *bwin, uint32_t nwin1, EmitParams ep, SinkParams sk)
{
	__shared__ int32_t h_key[T];
	__shared__ double h_val[MODE == MODE_COUNT ? 1 : T];
	__shared__ uint16_t occ[T / 2];
	__shared__ uint64_t s_sort[MODE == MODE_STORE ? T / 2 : 1];
	__shared__ uint16_t s_cnt[MODE == MODE_STORE ? 16 * ((T / 2 + NT - 1) / NT) * (NT / 64) : 1];
	__shared__ Expand<NT, T / 2> X;
	__shared__ uint32_t scr32[NT / 64 + 1];
	__shared__ uint32_t s_nocc;
	__shared__ PatCell s_pat;
	__shared__ unsigned long long s_u64[2 * (NT / 64)];
	__shared__ double s_f64[NT / 64];

	const unsigned tid = threadIdx.x;
	for (int q = tid; q < T; q += NT) { h_key[q] = -1; if (MODE != MODE_COUNT) h_val[q] = 0.0; }
	PatAcc pat; pat_init(pat);
	if (tid == 0) pat_reset(&s_pat);
	DigestAcc dacc{0, 0, 0.0};                                      // DIGEST, whole launch
	uint32_t flip = 0;

	// Software pipeline over the cells of this workgroup: the record of cell i+2, the A tuples of
	// cell i+1 and then its B segment bounds are loaded while cell i is processed (the barriers
	// inside are LDS-only, so these loads stay in flight).
	// The prefetches are branch-free (indices clamped to valid cells / tuples, results masked
	// afterwards): a load inside a conditional is waited for at the join, which would serialise it.
	const CellWalk walk = cell_walk(xb, ncell);
	const uint32_t stride = walk.stride, cend = walk.end;
	const bool any_cell = walk.first < cend;
	const uint32_t clast = any_cell ? cend - 1 : 0;
	Cell rec1 = cells[min(walk.first, clast)];
	Cell rec2 = cells[min(walk.first + stride, clast)];
	uint32_t nlo, nlen; double na;
	{
		const uint32_t e = rec1.beg + tid;
		const bool act = e < rec1.end;
		const uint32_t ec = act ? e : rec1.beg;
		const int32_t k = m.acol[ec];
		uint32_t lo, hi;
		if (WINDOWED) { const uint32_t *bw = bwin + (uint64_t)k * nwin1; lo = bw[rec1.wa]; hi = bw[rec1.wb]; }
		else { lo = m.bptr[k]; hi = m.bptr[k + 1]; }
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u;
	}
	for (uint32_t ci = walk.first; ci < cend; ci += stride) {
		const Cell cell = rec1;
		const uint32_t beg = cell.beg, end = cell.end, wa = cell.wa, wb = cell.wb, seg = cell.seg;
		const int32_t rowid = cell.rowid;
		const uint32_t lo0 = nlo, len0 = nlen; const double a0 = na;
		// stage A / B of the pipeline
		rec1 = rec2;
		rec2 = cells[min(ci + 2 * stride, clast)];
		const uint32_t ne = rec1.beg + tid;
		const bool nact = (ci + stride < cend) && ne < rec1.end;
		const uint32_t nec = ne < rec1.end ? ne : rec1.beg;
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];
		// Never: the class bounds the cell (T/2 products fit the LDS tables).  If the host's cell lists ever broke
		// that promise the cell is skipped as a whole -- the pipeline state below stays consistent -- and the error
		// word makes the multiply fail instead of returning a wrong product.
		const bool oversize = cell.prods > (uint32_t)(T / 2);
		if (oversize && tid == 0) atomicOr(sk.err, 1u);
		lds_barrier();                                              // previous cell fully emitted, its s_nocc read
		if (tid == 0) s_nocc = 0;

		for (uint32_t chunk = beg; chunk < (oversize ? beg : end); chunk += NT) {
			uint32_t lo = lo0, len = len0; double a = a0;
			if (chunk != beg) {
				uint32_t e = chunk + tid;
				lo = 0; len = 0; a = 0;
				if (e < end) {
					int32_t k = m.acol[e];
					if (WINDOWED) { const uint32_t *bw = bwin + (uint64_t)k * nwin1; lo = bw[wa]; len = bw[wb] - lo; }
					else { lo = m.bptr[k]; len = m.bptr[k + 1] - lo; }
					a = m.aval[e];
				}
			}
			uint32_t total, nzc, ex;
			expand_load(X, lo, len, a, &total, &nzc, flip, &ex);
			if (total == 0) continue;
			if (!ABL(ep, 4)) expand_batch(X, 0, total, nzc);
			if (ABL(ep, 1)) total = 0;
			if (ep.ordered) hash_products_ordered<T, NT, T / 2, MODE>(X, 0, nzc, m, h_key, h_val, occ, &s_nocc);
			else hash_products<T, NT, T / 2, MODE, PAT>(X, 0, total, 0, m, h_key, h_val, occ, &s_nocc, pat);
			lds_barrier();
		}
		if (PAT) pat_publish(pat, &s_pat);                   // (complete at the barrier below)
		// stage C of the pipeline: B segment bounds of the next cell's first chunk
		{
			uint32_t lo, hi;
			if (WINDOWED) { const uint32_t *bw = bwin + (uint64_t)nk * nwin1; lo = bw[rec1.wa]; hi = bw[rec1.wb]; }
			else { lo = m.bptr[nk]; hi = m.bptr[nk + 1]; }
			nlo = lo; nlen = nact ? hi - lo : 0u;
		}
		lds_barrier();
		uint32_t nocc = s_nocc;
		if (ABL(ep, 2)) nocc = 0;
		uint32_t colbase = 0, colbits = ep.ncolbits;
		if (WINDOWED) { colbase = wa << ep.wshift; colbits = ep.wshift + (wb - wa > 1 ? 32 - __builtin_clz(wb - wa - 1) : 0); }
		const double pthr = PAT ? pat_threshold(&s_pat, end - beg) : -1.0;
		hash_emit<T, NT, MODE, PAT>(nocc, rowid, seg, ep, sk, h_key, h_val, occ, s_sort, scr32, dacc, s_cnt, colbase, colbits, m, beg, end, pthr);
		if (PAT) { lds_barrier(); if (tid == 0) pat_reset(&s_pat); }     // (the next cell's barrier orders the reset)
	}
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, dacc.cnt, dacc.hash, dacc.sum, s_u64, s_f64);
}

// ---- tiles: several hash cells of ONE heavy row share the segment expansion ---------------
// A heavy row with L <= 256 A tuples has its hash cells (<= 2048 products each) grouped into
// tiles of up to NT / Lp cells (Lp = L rounded up to a power of two) and <= TILE_PB products.
// Thread t of the workgroup owns (cell t / Lp, tuple t % Lp): ONE expansion serves every cell of
// the tile; the cells are then accumulated one after the other in the same LDS table.
#ifndef TILE_NT_V
#define TILE_NT_V 512
#endif
constexpr int TILE_NT = TILE_NT_V;
constexpr int TILE_T = TILE_NT * 8;     // table slots: U = 4 products per thread fill it to one half
constexpr int TILE_PB = TILE_NT * 32;   // products per tile, DIGEST / COUNT launches
constexpr int TILE_PB_STORE = TILE_NT * 24;    // ... when the tiles also serve a STORE launch: its LDS then allows two workgroups per CU
constexpr uint32_t TILE_LMAX = 256;
constexpr uint32_t TILE_MAXCELLS = 16;

struct TCell { uint16_t wa, wb; uint32_t seg; uint32_t prods; };
struct Tile { uint32_t beg, end; int32_t rowid; uint32_t first, ncells, wa0, prods, pad; };

template <int MODE>
__global__ __launch_bounds__(TILE_NT) void k_hash_tiles(const Tile *tiles, uint32_t ntile, const TCell *tcells, RowMeta m,
	const uint32_t *bwin, uint32_t nwin1, EmitParams ep, SinkParams sk)
{
	constexpr int NT = TILE_NT, T = TILE_T;
	__shared__ int32_t h_key[T];
	__shared__ double h_val[MODE == MODE_COUNT ? 1 : T];
	__shared__ uint16_t occ[T / 2];
	__shared__ uint64_t s_sort[MODE == MODE_STORE ? T / 2 : 1];
	__shared__ uint16_t s_cnt[MODE == MODE_STORE ? 16 * ((T / 2 + NT - 1) / NT) * (NT / 64) : 1];
	constexpr int PB = MODE == MODE_STORE ? TILE_PB_STORE : TILE_PB;
	__shared__ Expand<NT, PB> X;
	__shared__ uint32_t scr32[NT / 64 + 1];
	__shared__ uint32_t s_nocc;
	__shared__ PatCell s_pat;
	__shared__ uint32_t cellP[TILE_MAXCELLS + 1];
	__shared__ unsigned long long s_u64[2 * (NT / 64)];
	__shared__ double s_f64[NT / 64];

	const unsigned tid = threadIdx.x;
	for (int q = tid; q < T; q += NT) { h_key[q] = -1; if (MODE != MODE_COUNT) h_val[q] = 0.0; }
	PatAcc pat; pat_init(pat);
	if (tid == 0) pat_reset(&s_pat);
	DigestAcc dacc{0, 0, 0.0};
	uint32_t flip = 0;
#ifdef SPSAMD_STAMPS
	unsigned long long st_[12] = {}; unsigned long long st_t = clock64();
#endif

	const uint32_t stride = gridDim.x;
	const uint32_t tlast = ntile - 1;
	// three-stage branch-free prefetch: tile record -> (A tuple, cell window range) -> B segment bounds
	Tile rec1 = tiles[min(blockIdx.x, tlast)];
	Tile rec2 = tiles[min(blockIdx.x + stride, tlast)];
	uint32_t nlo, nlen, nseg_; double na;
	{
		const uint32_t L = rec1.end - rec1.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		const uint32_t c = tid >> lsh, ei = tid & ((1u << lsh) - 1u);
		const bool act = c < rec1.ncells && ei < L;
		const uint32_t ec = rec1.beg + (ei < L ? ei : 0u);
		const TCell tc = tcells[rec1.first + (c < rec1.ncells ? c : 0u)];
		const uint32_t *bw = bwin + (uint64_t)m.acol[ec] * nwin1;
		const uint32_t lo = bw[tc.wa], hi = bw[tc.wb];
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u; nseg_ = tc.seg;
	}
	for (uint32_t ti = blockIdx.x; ti < ntile; ti += stride) {
		const Tile tile = rec1;
		const uint32_t lo = nlo, len = nlen, myseg = nseg_; const double a = na;
		const uint32_t L = tile.end - tile.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		const uint32_t myc = tid >> lsh, myei = tid & ((1u << lsh) - 1u);
		// stage A / B for the next tile
		rec1 = rec2;
		rec2 = tiles[min(ti + 2 * stride, tlast)];
		const bool has_next = ti + stride < ntile;
		const uint32_t nL = rec1.end - rec1.beg;
		uint32_t nsh = 0;
		while ((1u << nsh) < nL) ++nsh;
		const uint32_t nc = tid >> nsh, nei = tid & ((1u << nsh) - 1u);
		const bool nact = has_next && nc < rec1.ncells && nei < nL;
		const uint32_t nec = rec1.beg + (nei < nL ? nei : 0u);
		const TCell ntc = tcells[rec1.first + (nc < rec1.ncells ? nc : 0u)];
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];

		STAMP_COUNT(8);
		STAMP(0);
		lds_barrier();                                              // previous tile fully emitted
		STAMP(1);
		uint32_t total, nzc, ex;
		expand_load(X, lo, len, a, &total, &nzc, flip, &ex);
		if (myei == 0 && myc < tile.ncells) cellP[myc] = ex;       // first product of each cell
		if (tid == 0) { cellP[tile.ncells] = total; s_nocc = 0; }
		// segment ids of the cells of this tile: thread (c, 0) holds cell c's
		const uint32_t seg_of_mine = myseg;
		STAMP(2);
		if (total) expand_batch(X, 0, total, nzc);
		else lds_barrier();
		STAMP(3);
		// stage C: B segment bounds of the next tile
		{
			const uint32_t *bw = bwin + (uint64_t)nk * nwin1;
			const uint32_t nlo_ = bw[ntc.wa], nhi_ = bw[ntc.wb];
			nlo = nlo_; nlen = nact ? nhi_ - nlo_ : 0u; nseg_ = ntc.seg;
		}
		for (uint32_t c = 0; c < tile.ncells; ++c) {
			STAMP_COUNT(9);
			STAMP(0);
			const uint32_t p0 = cellP[c], p1 = cellP[c + 1];
			if (ep.ordered) {
				// the cell's products [p0, p1) are whole segments (a segment belongs to one cell)
				const uint32_t q0 = expand_lookup(X, p0, 0), q1 = expand_lookup(X, p1 - 1, 0) + 1;
				hash_products_ordered<T, NT, PB, MODE>(X, q0, q1, m, h_key, h_val, occ, &s_nocc);
			} else if (ep.pattern) hash_products<T, NT, PB, MODE, true>(X, p0, p1, 0, m, h_key, h_val, occ, &s_nocc, pat);
			else hash_products<T, NT, PB, MODE, false>(X, p0, p1, 0, m, h_key, h_val, occ, &s_nocc, pat);
			if (ep.pattern) pat_publish(pat, &s_pat);
			STAMP(4);
			lds_barrier();
			STAMP(5);
			const uint32_t nocc = s_nocc;
			// the cell's output segment id lives in thread (c, 0): broadcast through LDS
			if (myc == c && myei == 0) scr32[NT / 64] = seg_of_mine;
			lds_barrier();
			const uint32_t seg = scr32[NT / 64];
			if (tid == 0) s_nocc = 0;
			uint32_t colbase = 0, colbits = 0;
			if (MODE == MODE_STORE) {
				const TCell tcc = tcells[tile.first + c];                  // uniform
				colbase = (uint32_t)tcc.wa << ep.wshift;
				colbits = ep.wshift + (tcc.wb - tcc.wa > 1 ? 32 - __builtin_clz((uint32_t)(tcc.wb - tcc.wa) - 1u) : 0);
			}
			STAMP(6);
			const double pthr = ep.pattern ? pat_threshold(&s_pat, tile.end - tile.beg) : -1.0;
			if (ep.pattern) hash_emit<T, NT, MODE, true>(nocc, tile.rowid, seg, ep, sk, h_key, h_val, occ, s_sort, scr32, dacc, s_cnt, colbase, colbits, m, tile.beg, tile.end, pthr);
			else hash_emit<T, NT, MODE, false>(nocc, tile.rowid, seg, ep, sk, h_key, h_val, occ, s_sort, scr32, dacc, s_cnt, colbase, colbits, m, tile.beg, tile.end, pthr);
			STAMP(7);
			if (ep.pattern) { lds_barrier(); if (tid == 0) pat_reset(&s_pat); }
			lds_barrier();
		}
	}
#ifdef SPSAMD_STAMPS
	if (tid == 0 && sk.stamps) for (int i = 0; i < 12; ++i) sk.stamps[(size_t)blockIdx.x * 12 + i] = st_[i];
#endif
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, dacc.cnt, dacc.hash, dacc.sum, s_u64, s_f64);
}

// ====================================================================== heavy rows: window index, cells

// Window index of B: bwin[k * (nwin+1) + w] = first tuple of B row k whose
// column is >= w * W  (bwin[k][0] = bptr[k], bwin[k][nwin] = bptr[k+1]).
__global__ void k_bwin_prefill(const uint32_t *bptr, uint64_t nrowb, uint32_t nwin1, uint32_t *bwin)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t total = nrowb * nwin1;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < total; i += stride) bwin[i] = bptr[i / nwin1 + 1];
}

__global__ void k_bwin_fill(const int32_t *brow, const int32_t *bcol, const uint32_t *bptr, uint32_t nnzb, uint32_t wshift,
	uint32_t nwin1, uint32_t *bwin)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= nnzb) return;
	int32_t k = brow[e];
	int w = (int)((uint32_t)bcol[e] >> wshift);
	int wprev = (e > bptr[k]) ? (int)((uint32_t)bcol[e - 1] >> wshift) : -1;
	for (int ww = wprev + 1; ww <= w; ++ww) bwin[(uint64_t)k * nwin1 + ww] = e;
}

// Tuples of B row k in window w as 16 bits (<= W <= 16384): half the bytes of the offset pairs for
// the histogram below, which reads one whole row of this table per A tuple of a heavy row.  Rows
// are padded to an even number of entries (nwp) so that two windows are read as one 32-bit word.
__global__ void k_bwin_counts(const uint32_t *bwin, uint64_t nrowb, uint32_t nwin, uint32_t nwp, uint16_t *cnt)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t total = nrowb * nwp, stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < total; i += stride) {
		const uint64_t k = i / nwp, w = i - k * nwp;
		uint16_t v = 0;
		if (w < nwin) { const uint32_t *bw = bwin + k * (nwin + 1) + w; v = (uint16_t)(bw[1] - bw[0]); }
		cnt[i] = v;
	}
}

// ---- window-major copy of B for the dense cells -------------------------------------------
// A dense cell is one output row x ONE column window, and the cell lists are walked window by
// window.  In the row-major array the tuples of window w are scattered over all of B (a few
// tuples per 128-byte line belong to the window), and the index lookups bwin[k][w] touch one line
// per B row.  The window-major copy puts the tuples of window w side by side, ordered by (k, col),
// with a CSR row pointer per window: wptr[w * nrowb + k] .. [+1].  The working set of the
// workgroups that are on window w is then |B_w| * 12 bytes plus a 4 * nrowb byte pointer slice,
// and a row's pass over its A tuples (ascending k) moves forward through both.
__global__ __launch_bounds__(256) void k_wm_counts(const uint32_t *bwin, uint32_t nrowb, uint32_t nwin, uint32_t nwin1, uint16_t *cnt)
{
	__shared__ uint32_t tile[64][65];
	const uint32_t k0 = blockIdx.x * 64u, w0 = blockIdx.y * 64u;
	const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
	for (uint32_t ky = ty; ky < 64; ky += 4) {
		const uint32_t k = k0 + ky;
		if (k < nrowb) {
			const uint32_t *row = bwin + (uint64_t)k * nwin1;
			if (w0 + tx < nwin1) tile[ky][tx] = row[w0 + tx];
			if (tx == 0 && w0 + 64 < nwin1) tile[ky][64] = row[w0 + 64];
		}
	}
	__syncthreads();
	for (uint32_t wy = ty; wy < 64; wy += 4) {
		const uint32_t w = w0 + wy, k = k0 + tx;
		if (w < nwin && k < nrowb) cnt[(uint64_t)w * nrowb + k] = (uint16_t)(tile[tx][wy + 1] - tile[tx][wy]);   // <= W tuples of one row in one window
	}
}

__global__ void k_wm_scatter(const int32_t *brow, const int32_t *bcol, const double *bval, uint32_t nnzb, uint32_t wshift,
	const uint32_t *bwin, uint32_t nwin1, const uint32_t *wptr, uint64_t nrowb, BTup *out)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= nnzb) return;
	const uint32_t k = (uint32_t)brow[e], c = (uint32_t)bcol[e], w = c >> wshift;
	const uint32_t dst = wptr[(uint64_t)w * nrowb + k] + (e - bwin[(uint64_t)k * nwin1 + w]);
	const double v = bval[e];
	BTup t; t.col = (int32_t)c; t.vlo = (uint32_t)__double2loint(v); t.vhi = (uint32_t)__double2hiint(v);
	out[dst] = t;
}

// Per heavy row: products per column window.  One workgroup per row; a thread owns a PAIR of
// windows (one 32-bit load per A tuple), sub-groups of threads take different tuples and every
// thread keeps 8 tuples in flight.  A row's workgroup takes at most WH_HUB tuples; rows with more
// are noted in a list and their remaining tuples are dealt in parts to a second launch that adds
// into the row's histogram with global atomics (the longest hub row -- tens of thousands of tuples
// -- would otherwise set the time of the whole kernel).
constexpr int WH_NT = 256;
constexpr int WH_MAXW = 2048;                // windows supported (ncol <= 2^25 at W = 16384)
constexpr uint32_t WH_HUB = 4096;            // tuples one workgroup takes
constexpr uint32_t WH_HUB_MAX = 65536;       // list capacity (rows beyond it are finished by their own workgroup)

__device__ __forceinline__ void win_hist_span(const RowMeta &m, const uint16_t *wcnt, uint32_t nwp, uint32_t beg, uint32_t end, uint32_t *s_cnt)
{
	const uint32_t npair = nwp >> 1;
	const uint32_t *tab = (const uint32_t *)wcnt;                       // row k: npair words
	uint32_t ppad = 1;
	while (ppad < npair && ppad < WH_NT) ppad <<= 1;
	const uint32_t nsub = ppad < WH_NT ? WH_NT / ppad : 1;
	const uint32_t sub = threadIdx.x / ppad, p0 = threadIdx.x % ppad;
	for (uint32_t p = p0; p < npair; p += ppad) {                       // one pass unless there are more than 256 pairs
		uint32_t c0 = 0, c1 = 0;
		uint32_t e = beg + sub;
		for (; e + 7 * nsub < end; e += 8 * nsub) {
			uint32_t x[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) x[u] = tab[(uint64_t)m.acol[e + u * nsub] * npair + p];
#pragma unroll
			for (int u = 0; u < 8; ++u) { c0 += x[u] & 0xFFFFu; c1 += x[u] >> 16; }
		}
		for (; e < end; e += nsub) { const uint32_t x = tab[(uint64_t)m.acol[e] * npair + p]; c0 += x & 0xFFFFu; c1 += x >> 16; }
		if (c0) atomicAdd(&s_cnt[2 * p], c0);
		if (c1) atomicAdd(&s_cnt[2 * p + 1], c1);
	}
}

__global__ __launch_bounds__(WH_NT) void k_win_hist(const uint32_t *hrows, uint32_t nheavy, RowMeta m, const uint16_t *wcnt,
	uint32_t nwin, uint32_t nwp, uint32_t *winprod, uint32_t *hubcount, uint32_t *hublist)
{
	__shared__ uint32_t s_cnt[WH_MAXW];
	__shared__ uint32_t s_listed;
	const uint32_t h = blockIdx.x, r = hrows[h];
	const uint32_t beg = m.beg[r], end = m.beg[r + 1];
	for (uint32_t w = threadIdx.x; w < nwp; w += WH_NT) s_cnt[w] = 0;
	if (threadIdx.x == 0) {
		uint32_t listed = 0;
		if (end - beg > WH_HUB) {
			const uint32_t slot = atomicAdd(hubcount, 1u);
			if (slot < WH_HUB_MAX) { hublist[slot] = h; listed = 1; }
		}
		s_listed = listed;
	}
	__syncthreads();
	win_hist_span(m, wcnt, nwp, beg, s_listed ? beg + WH_HUB : end, s_cnt);
	__syncthreads();
	for (uint32_t w = threadIdx.x; w < nwin; w += WH_NT) winprod[(uint64_t)h * nwin + w] = s_cnt[w];
}

// The tuples beyond WH_HUB of the listed rows, WH_HUB at a time: work item = (listed row, part).
__global__ __launch_bounds__(WH_NT) void k_win_hist_hub(const uint32_t *hrows, RowMeta m, const uint16_t *wcnt,
	uint32_t nwin, uint32_t nwp, uint32_t *winprod, const uint32_t *hubcount, const uint32_t *hublist)
{
	__shared__ uint32_t s_cnt[WH_MAXW];
	const uint32_t nhub = min(*hubcount, WH_HUB_MAX);
	// items are enumerated row by row; a workgroup finds its items by walking the (short) list
	uint32_t item = 0;
	for (uint32_t q = 0; q < nhub; ++q) {
		const uint32_t h = hublist[q], r = hrows[h];
		const uint32_t beg = m.beg[r] + WH_HUB, end = m.beg[r + 1];
		const uint32_t parts = (end - beg + WH_HUB - 1) / WH_HUB;
		for (uint32_t part = 0; part < parts; ++part, ++item) {
			if (item % gridDim.x != blockIdx.x) continue;                 // uniform
			for (uint32_t w = threadIdx.x; w < nwp; w += WH_NT) s_cnt[w] = 0;
			__syncthreads();
			win_hist_span(m, wcnt, nwp, beg + part * WH_HUB, min(end, beg + (part + 1) * WH_HUB), s_cnt);
			__syncthreads();
			for (uint32_t w = threadIdx.x; w < nwin; w += WH_NT) { const uint32_t v = s_cnt[w]; if (v) atomicAdd(&winprod[(uint64_t)h * nwin + w], v); }
			__syncthreads();
		}
	}
}

// Cell classes: 0..3 hash (T = 1024 / 3072 / 4096 / 8192 slots; T/2 products), 4 dense
constexpr int NCLS = 5;
constexpr int CLS_DENSE = NCLS - 1;
constexpr uint32_t CELL_CAP = 4096;      // largest hash cell (T = 8192)
constexpr uint32_t CELL_CAP_DEFAULT = 2048;      // greedy grouping target of the hash cells (measured best on R-MAT scale-20)
// A single window above DENSE_MIN products becomes a dense cell.  Measured on R-MAT scale 20 with the bitmap tiles as they
// are now (cfg2, ms): 1792 -> 80.9, 2048 -> 79.0, 2560 -> 79.0, 2816 -> 78.5, 3072 -> 78.1 .. 78.6, 3328 -> 79.0, 3584 -> 80.7,
// 4096 -> 81.4 (the tiles take a 3000-product window at 4.3 ps per product, the dense kernel -- which scans all W slots
// -- needs more products than that to get to its 2.6).
constexpr uint32_t DENSE_MIN_DEFAULT = 2048;     // ... with hash tiles (2048-product cells; scale 23: 2.25 s against 2.38 s at 3072)
constexpr uint32_t LONG_DENSE_MIN_DEFAULT = 1024; // ... for the rows too long for a tile (heavy_prepare)
constexpr uint32_t DENSE_MIN_BITMAP = 3072;      // ... with bitmap tiles (4096-product cells)
// A single window of a tile row above DIRECT_MIN products becomes a direct cell (k_direct_tiles).  OFF by default (>= the
// dense threshold): the direct cells paid while a hash / bitmap tile cell cost 8 .. 10 k cycles (-0.6 ms at 1536); against
// today's bitmap tiles they lose (cfg2: 80.4 with direct cells above 1536 products, 79.0 without; 1024 -> 86.0).
constexpr uint32_t DIRECT_MIN_DEFAULT = 4096;
__device__ __forceinline__ int hash_class(uint32_t prods) { return prods <= 512 ? 0 : (prods <= 1536 ? 1 : (prods <= 2048 ? 2 : 3)); }

struct CellBases { uint32_t *base[NCLS]; };      // per heavy row: first cell index in each class list
struct CellLists { Cell *list[NCLS]; };

// Greedy grouping of a heavy row's windows into cells.  WRITE = false counts
// the cells per class (and the row's segment count); WRITE = true emits them.
struct TileBases { uint32_t *ntc, *ntl, *tcbase, *tlbase; TCell *tcells; Tile *tiles; int enabled; uint32_t pb; int by_items; };
// Two kinds of tile: [0] hash cells (ranges of sparse windows, LDS hash table), [1] direct cells (ONE window holding
// more than direct_min products, dense window accumulator with claim-by-exchange emission: k_direct_tiles)
struct TileKinds { TileBases k[2]; uint32_t direct_min; uint32_t span_cap; uint32_t long_cap; uint32_t long_dense_min; uint32_t tile_cap;
	uint32_t alt_cap, alt_span; unsigned long long *alt_cells; };   // alt_*: (counting pass) the tile cells another cap / span would give   // long_*: cell_cap / dense_min of the rows too long for tiles    // span_cap: most windows one tile cell may cover (0: any)

template <bool WRITE>
__global__ void k_cells(const uint32_t *hrows, uint32_t nheavy, const uint32_t *rbeg, const int32_t *rid,
	const uint32_t *winprod, uint32_t nwin, uint32_t cell_cap, uint32_t dense_min,
	CellBases cnt, uint32_t *nseg, CellBases base, CellLists lists, const uint32_t *segbase, unsigned long long *clsprod,
	TileKinds tk)
{
	uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= nheavy) return;
	const uint32_t r = hrows[h];
	Cell proto;
	proto.beg = rbeg[r]; proto.end = rbeg[r + 1]; proto.rowid = rid[r]; proto.pad[0] = proto.pad[1] = 0;
	const uint32_t *wp = winprod + (uint64_t)h * nwin;
	uint32_t n[NCLS] = {};
	unsigned long long np[NCLS + 2] = {};                              // + the two tile kinds
	uint32_t ordinal = 0;
	uint32_t cur = 0, start = 0, last = 0;
	// rows with few A tuples: their hash / direct cells are grouped into tiles that share one expansion
	const uint32_t L = proto.end - proto.beg;
	const bool tileable = tk.k[0].enabled && L <= TILE_LMAX;
	const bool direct_ok = tileable && tk.k[1].enabled;
	// a hash cell of a row with many A tuples visits all of them whatever it holds: such rows get larger cells
	if (tileable && tk.tile_cap > cell_cap) cell_cap = tk.tile_cap;      // tile cells may be larger than a hash table's (bitmap tiles)
	if (!tileable && tk.long_cap > cell_cap) cell_cap = tk.long_cap;
	// ... and their windows go to the dense kernel much earlier: walking a long row costs more than scanning the window
	if (!tileable && tk.long_dense_min && tk.long_dense_min < dense_min) dense_min = tk.long_dense_min;
	uint32_t lsh = 0;
	while ((1u << lsh) < L) ++lsh;
	const uint32_t G = min((uint32_t)TILE_NT >> lsh, TILE_MAXCELLS);
	uint32_t ntc[2] = {0, 0}, ntl[2] = {0, 0};                        // tile cells / tiles emitted so far for this row
	uint32_t tcnt[2] = {0, 0}, tcost[2] = {0, 0}, tprods[2] = {0, 0}, tfirst[2] = {0, 0}, twa0[2] = {0, 0};   // the open tiles
	auto close_tile = [&](int kd) {
		if (!tcnt[kd]) return;
		if (WRITE) {
			Tile t; t.beg = proto.beg; t.end = proto.end; t.rowid = proto.rowid; t.first = tk.k[kd].tcbase[h] + tfirst[kd]; t.ncells = tcnt[kd];
			t.wa0 = twa0[kd]; t.prods = tprods[kd]; t.pad = 0;
			tk.k[kd].tiles[tk.k[kd].tlbase[h] + ntl[kd]] = t;
		}
		++ntl[kd]; tcnt[kd] = 0; tcost[kd] = 0; tprods[kd] = 0;
	};
	// cost of a cell against the tile's capacity: products for a hash tile; for a direct tile an upper bound of its
	// ITEMS (R tuples each, at most one partial item per A tuple) rounded up to whole 64-item blocks
	auto tile_cell = [&](int kd, uint32_t wa, uint32_t wb, uint32_t prods) {
		const uint32_t cost = tk.k[kd].by_items ? ((prods / DENSE_R + L + 63u) & ~63u) + 64u : prods;
		if (tcnt[kd] == G || tcost[kd] + cost > tk.k[kd].pb) close_tile(kd);
		if (!tcnt[kd]) { tfirst[kd] = ntc[kd]; twa0[kd] = wa; }
		if (WRITE) {
			TCell tc; tc.wa = (uint16_t)wa; tc.wb = (uint16_t)wb; tc.seg = segbase ? segbase[r] + ordinal : 0; tc.prods = prods;
			tk.k[kd].tcells[tk.k[kd].tcbase[h] + ntc[kd]] = tc;
		}
		++ntc[kd]; ++tcnt[kd]; tcost[kd] += cost; tprods[kd] += prods;
		np[NCLS + kd] += prods;
		++ordinal;
	};
	auto flush = [&]() {
		if (!cur) return;
		if (tileable && cur <= tk.tile_cap) {
			tile_cell(0, start, last + 1, cur);
			cur = 0;
			return;
		}
		int cls = hash_class(cur);
		if (WRITE) {
			Cell c = proto; c.seg = segbase ? segbase[r] + ordinal : 0; c.prods = cur; c.wa = (uint16_t)start; c.wb = (uint16_t)(last + 1);
			lists.list[cls][base.base[cls][h] + n[cls]] = c;
		}
		++n[cls]; np[cls] += cur; ++ordinal; cur = 0;
	};
	// the alternative tile scheme's cell count (counting pass only): same greedy grouping with its own cap and span
	uint32_t alt_cur = 0, alt_start = 0, alt_n = 0;
	const bool alt_on = !WRITE && tk.alt_cells && tileable;
	// the row's histogram is read four windows per load where the row is 16-byte aligned (one thread per
	// row: consecutive threads are a whole row apart, so narrow loads waste most of every cache line)
	const bool vec4 = (nwin & 3u) == 0;
	uint4 quad = make_uint4(0, 0, 0, 0);
	for (uint32_t w = 0; w < nwin; ++w) {
		uint32_t c;
		if (vec4) {
			if ((w & 3u) == 0) quad = *reinterpret_cast<const uint4 *>(wp + w);
			c = (w & 3u) == 0 ? quad.x : ((w & 3u) == 1 ? quad.y : ((w & 3u) == 2 ? quad.z : quad.w));
		} else c = wp[w];
		if (alt_on) {
			if (c > dense_min || (direct_ok && c > tk.direct_min)) { if (alt_cur) { ++alt_n; alt_cur = 0; } }
			else if (c > 0) {
				if (alt_cur && (alt_cur + c > tk.alt_cap || (tk.alt_span && w - alt_start >= tk.alt_span))) { ++alt_n; alt_cur = 0; }
				if (!alt_cur) alt_start = w;
				alt_cur += c;
			}
		}
		if (c > dense_min) {
			flush();
			if (WRITE) {
				Cell d = proto; d.seg = segbase ? segbase[r] + ordinal : 0; d.prods = c; d.wa = (uint16_t)w; d.wb = (uint16_t)(w + 1);
				lists.list[CLS_DENSE][base.base[CLS_DENSE][h] + n[CLS_DENSE]] = d;
			}
			++n[CLS_DENSE]; np[CLS_DENSE] += c; ++ordinal;
		} else if (direct_ok && c > tk.direct_min) {
			// one window of a tile row with enough products to pay for a cell of its own: direct cell
			flush();
			tile_cell(1, w, w + 1, c);
		} else if (c > cell_cap) {
			// too large for a group, too small for a dense window: a hash cell of its own
			flush();
			cur = c; start = last = w;
			flush();
		} else if (c > 0) {
			if (cur + c > cell_cap || (cur && tileable && tk.span_cap && w - start >= tk.span_cap)) flush();
			if (!cur) start = w;
			cur += c; last = w;
		}
	}
	flush();
	close_tile(0);
	close_tile(1);
	if (!WRITE) {
#pragma unroll
		for (int k = 0; k < NCLS; ++k) cnt.base[k][h] = n[k];
		for (int k = 0; k < NCLS + 2; ++k) if (np[k]) atomicAdd(&clsprod[k], np[k]);
		nseg[r] = ordinal;
		for (int kd = 0; kd < 2; ++kd) if (tk.k[kd].enabled) { tk.k[kd].ntc[h] = ntc[kd]; tk.k[kd].ntl[h] = ntl[kd]; }
		if (alt_on) { if (alt_cur) ++alt_n; if (alt_n) atomicAdd(tk.alt_cells, (unsigned long long)alt_n); if (ntc[0]) atomicAdd(tk.alt_cells + 1, (unsigned long long)ntc[0]); }
	}
}

__global__ void k_tile_keys(const Tile *tiles, uint32_t n, uint64_t *keys)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) keys[i] = tiles[i].wa0;             // window-major, stable
}

__global__ void k_gather_tiles(const Tile *src, const uint32_t *perm, uint32_t n, Tile *dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[perm[i]];
}

__global__ void k_cell_keys(const Cell *cells, uint32_t n, int by_size, uint64_t *keys)
{
	// window-major; inside a window the largest cells first (dense) or input order (hash: stable sort)
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t w = cells[i].wa;
	// size in units of 256 products, 16 bits (cells beyond 2^24 products rank equal): a 16-bit minor key = 2 sort passes
	const uint32_t sz = min(cells[i].prods >> 8, 0xFFFFu);
	keys[i] = by_size ? ((w << 16) | (uint64_t)(0xFFFFu - sz)) : w;
}

__global__ void k_gather_cells(const Cell *src, const uint32_t *perm, uint32_t n, Cell *dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[perm[i]];
}

// ====================================================================== dense cells (f64 window accumulator in LDS)

// Persistent workgroups walk the dense cells (one window of one row holding more products than a
// hash cell takes) with a grid stride.
//
// Product loop.  A chunk of NT A tuples selects NT segments of B (one window each).  A non-empty
// segment of `len` tuples is ceil(len / R) ITEMS of R consecutive tuples; items are numbered
// 0..total-1 over the compacted segments and dealt 64 consecutive items per wave and step.  The
// segment of an item needs no search: every segment sets the bit of its first item in a bitmap
// (one 64-bit word per 64-item block), each wave keeps the words and their popcount prefix in
// registers (word l and l + 64 in lane l), and for block b -- wave-uniform -- takes both with
// v_readlane; a lane's segment is then  prefix + (bits of the word up to its own position) - 1.
// One LDS round trip (segment bounds, item prefix, A value) per R products follows, the R tuples
// are read as one 12 R-byte piece, and the sums go to the LDS accumulator with ds_add_f64.
// Occupancy: 16 waves per CU (two 512-thread workgroups, or one of 1024) = 4 per SIMD, so the kernel
// is held to 128 VGPRs (launch bound 4): a build that needs more silently halves the occupancy.
template <int W, int NT, int MODE, bool PAT>
__global__ __launch_bounds__(NT, 4) void k_dense(const Cell *cells, uint32_t ncell, const uint32_t *xb, RowMeta m,
	const uint32_t *widx, uint64_t kstride, uint64_t wstride, uint32_t narrow, EmitParams ep, SinkParams sk)
{
	constexpr int NW = NT / 64;
	constexpr int NGRP = W / 64;             // 64-slot groups per window
	constexpr int GPW = NGRP / NW;           // groups per wave
	constexpr uint32_t WSHIFT = W == 8192 ? 13 : 14;
	constexpr int R = DENSE_R;
	constexpr int NWORD = W / 64;            // bitmap words of one batch (W items)
	constexpr int WPL = NWORD / 64;          // words per lane of the per-wave copy
	__shared__ double acc[W + 64];           // + one dump slot per lane: tuples past the end of a segment's last item land there
	__shared__ uint32_t s_cpref[NT + 1];     // compacted segments: exclusive ITEM prefix (+ total)
	__shared__ uint2 s_cse[NT];              // first tuple of the segment, one past its last
	__shared__ double s_caval[NT];           // the A value
	__shared__ unsigned long long s_bmask[NWORD];
	__shared__ uint32_t s_scrL[2][NW], s_scrN[2][NW];
	__shared__ uint32_t s_wcnt[NW + 1];
	__shared__ PatCell s_pat;
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];

	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	// EXACT_PATTERN: a clean slot holds -0.0.  No sum of products is -0.0 (x + -x = +0, and a product is never a zero: zeros
	// are dropped at consolidation), -0.0 + p = p exactly, and -0.0 is "not emitted" like +0 -- so the scan-out can tell a slot
	// no product touched from one whose terms cancelled, and re-evaluates only the latter.  (With +0 as the clean value every
	// EMPTY slot of a cell with products of both signs was re-evaluated from the operands: 4.8 s instead of 11 ms on a
	// scale-18 R-MAT with random signs.)
	const double CLEAN = (PAT && MODE != MODE_COUNT) ? -0.0 : 0.0;
	for (int q = tid; q < W + 64; q += NT) acc[q] = CLEAN;
	for (int q = tid; q < NWORD; q += NT) s_bmask[q] = 0ull;
	const bool plain = ep.C == 1.0 && !ep.si_pos && !ep.sk_pos;    // uniform: the emitted value is the sum itself
	const unsigned long long laneK = (unsigned long long)lane * 0x9E3779B97F4A7C15ull;
	unsigned long long d_cnt = 0, d_hash = 0; double d_sum = 0;     // DIGEST, whole launch
	PatAcc pat; pat_init(pat);
	if (tid == 0) pat_reset(&s_pat);
	uint32_t flip = 0;
#ifdef SPSAMD_STAMPS
	unsigned long long st_[12] = {}; unsigned long long st_t = clock64();
#endif

	// Cells are ordered by (window, descending products) and dealt with a grid stride, so the
	// workgroups are on the same few column windows of B at any time and every workgroup gets a
	// mix of large and small cells.  Software pipeline over the cells: the record of cell i+2, the
	// A tuples of cell i+1 and then its B segment bounds are loaded while cell i is processed
	// (branch-free prefetches: indices clamped, results masked); inside a cell the A tuples of
	// chunk c+2 and the segment bounds of chunk c+1 are in flight while chunk c is processed.
	const CellWalk walk = cell_walk(xb, ncell);
	const uint32_t stride = walk.stride, cend = walk.end;
	const bool any_cell = walk.first < cend;
	const uint32_t clast = any_cell ? cend - 1 : 0;
	Cell rec1 = cells[min(walk.first, clast)];
	Cell rec2 = cells[min(walk.first + stride, clast)];
	uint32_t nlo, nlen; double na;
	auto seg_bounds = [&](int32_t k, uint32_t w, uint32_t &lo, uint32_t &hi) {
		// segment of B row k in window w: [widx[k*kstride + w*wstride], widx[.. + 1]) -- the row-major index
		// bwin (kstride = nwin+1, wstride = 1) or the window-major row pointer wptr (kstride = 1, wstride = nrowb)
		const uint32_t *bw = widx + (uint64_t)(uint32_t)k * kstride + (uint64_t)w * wstride;
		lo = bw[0]; hi = bw[1];
	};
	{
		const uint32_t e = rec1.beg + tid;
		const bool act = e < rec1.end;
		const uint32_t ec = act ? e : rec1.beg;
		uint32_t lo, hi;
		seg_bounds(m.acol[ec], rec1.wa, lo, hi);
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u;
	}
	__syncthreads();
	for (uint32_t ci = walk.first; ci < cend; ci += stride) {
		const Cell cell = rec1;
		const uint32_t w = cell.wa;
		const uint32_t beg = cell.beg, end = cell.end;
		const int32_t rowid = cell.rowid;
		const double a_scale = row_scale(ep, rowid);
		const uint32_t wbase = w << WSHIFT;
		uint32_t lo = nlo, len = nlen; double a = na;               // chunk 0, prefetched
		rec1 = rec2;
		rec2 = cells[min(ci + 2 * stride, clast)];
		const uint32_t ne = rec1.beg + tid;
		const bool nact = (ci + stride < cend) && ne < rec1.end;
		const uint32_t nec = ne < rec1.end ? ne : rec1.beg;
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];
		// in-cell prefetch, stage A: the A tuple of chunk 1
		int32_t kA = 0; double aA = 0.0;
		if (beg + NT < end) {                                       // uniform
			const uint32_t e1 = beg + NT + tid;
			const uint32_t e1c = e1 < end ? e1 : beg;
			kA = m.acol[e1c]; aA = m.aval[e1c];
		}

		STAMP_COUNT(8);
		uint32_t pnseg = 0;                                         // non-empty segments of the cell (EXACT_PATTERN)
		for (uint32_t chunk = beg; chunk < end; chunk += NT) {
			STAMP_COUNT(9);
			// stage B for chunk c+1 (its k arrived during chunk c-1), stage A for chunk c+2
			uint32_t lo2 = 0, hi2 = 0; double a2 = 0.0;
			if (chunk + NT < end) {                                 // uniform
				seg_bounds(kA, w, lo2, hi2);
				a2 = aA;
				if (chunk + NT + tid >= end) hi2 = lo2;
				if (chunk + 2 * NT < end) {
					const uint32_t e2 = chunk + 2 * NT + tid;
					const uint32_t e2c = e2 < end ? e2 : beg;
					kA = m.acol[e2c]; aA = m.aval[e2c];
				}
			}
			// ---- compact the non-empty segments, item prefix, first-item bits
			const uint32_t items = (len + R - 1) / R;
			const uint32_t inc = wave_inclusive_scan_u32(items);
			const uint64_t nzm = __ballot(len != 0);
			const uint32_t wrank = (uint32_t)__popcll(nzm & lanemask_lt());
			if (lane == 63) s_scrL[flip][wv] = inc;
			if (lane == 0) s_scrN[flip][wv] = (uint32_t)__popcll(nzm);
			STAMP(0);
			lds_barrier();                                          // B1: also orders the previous chunk's / cell's LDS traffic
			STAMP(1);
			uint32_t baseL = 0, baseN = 0, total = 0, nzc = 0;
#pragma unroll
			for (int q = 0; q < NW; ++q) {
				const uint32_t l = s_scrL[flip][q], n = s_scrN[flip][q];
				if (q < (int)wv) { baseL += l; baseN += n; }
				total += l; nzc += n;
			}
			flip ^= 1u;
			total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);     // uniform by construction: keep the loop control scalar
			nzc = (uint32_t)__builtin_amdgcn_readfirstlane((int)nzc);
			const uint32_t myfirst = baseL + inc - items;           // first item of this thread's segment
			if (len) {
				const uint32_t rank = baseN + wrank;
				s_cpref[rank] = myfirst;
				s_cse[rank] = make_uint2(lo, lo + len);
				s_caval[rank] = a;
				if (myfirst < (uint32_t)W) atomicOr(&s_bmask[myfirst >> 6], 1ull << (myfirst & 63u));
			}
			if (ABL(ep, 8)) total = 0;
			if (total == 0) { lo = lo2; len = hi2 - lo2; a = a2; continue; }     // uniform (nothing was marked)
			if (ep.ordered && MODE != MODE_COUNT) {
				// ascending-k accumulation, one segment (unique columns) at a time: see hash_products_ordered
				lds_barrier();
				for (uint32_t q = 0; q < nzc; ++q) {
					const uint2 se = s_cse[q];
					const double aq = s_caval[q];
					for (uint32_t t = se.x + tid; t < se.y; t += NT) {
						const BTup bt = m.btup[t];
						const uint32_t slot = (uint32_t)bt.col - wbase;
						acc[slot] = acc[slot] + aq * btup_val(bt);
					}
					lds_barrier();
				}
				for (int q = tid; q < NWORD; q += NT) s_bmask[q] = 0ull;
				lo = lo2; len = hi2 - lo2; a = a2;
				continue;
			}
			uint32_t Q0 = 0;                                        // segments that start before the batch
			for (uint32_t pb = 0; pb < total; pb += W) {
				const uint32_t pe = min(total, pb + (uint32_t)W);
				if (pb) {
					// a later batch of a very large chunk: re-mark the first-item bits of its own range
					lds_barrier();
					for (int q = tid; q < NWORD; q += NT) s_bmask[q] = 0ull;
					lds_barrier();
					if (len && myfirst >= pb && myfirst < pe) atomicOr(&s_bmask[(myfirst - pb) >> 6], 1ull << ((myfirst - pb) & 63u));
				}
				STAMP(2);
				lds_barrier();                                      // B2: compacted segments and bits visible
				STAMP(3);
				// ---- per-wave copy of the bitmap and its popcount prefix
				unsigned long long mw[WPL]; uint32_t pre[WPL];
				uint32_t run = Q0;
#pragma unroll
				for (int x = 0; x < WPL; ++x) {
					mw[x] = s_bmask[x * 64 + lane];
					const uint32_t cnt = (uint32_t)__popcll(mw[x]);
					const uint32_t incl = wave_inclusive_scan_u32(cnt);
					pre[x] = run + incl - cnt;
					run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				}
				const uint32_t nsteps = (pe - pb + NT - 1) / NT;            // uniform
				// item -> (first tuple, valid tuples, A value).  Block b is wave-uniform: its bitmap word and prefix
				// come out of the registers with v_readlane; the lane's segment is prefix + (first-item bits at
				// positions <= lane) - 1, the bits below the lane counted by mbcnt on the word shifted right by one.
				auto lookup = [&](uint32_t step, uint32_t &obp, uint32_t &onv, double &oav) {
					const uint32_t b = step * NW + wv;
					const uint32_t t = pb + (b << 6) + lane;
					const bool ok = t < pe;
					uint32_t mlo = 0, mhi = 0, pr = 0;
#pragma unroll
					for (int x = 0; x < WPL; ++x) {
						if ((b >> 6) == (uint32_t)x) {                       // uniform
							mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mw[x], (int)(b & 63u));
							mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mw[x] >> 32), (int)(b & 63u));
							pr = (uint32_t)__builtin_amdgcn_readlane((int)pre[x], (int)(b & 63u));
						}
					}
					if (ABL(ep, 128)) { obp = (t * R) & 0xFFFFFu; onv = ok ? R : 0u; oav = 1.0; return; }      // no segment lookup
					const uint32_t s1lo = (mlo >> 1) | (mhi << 31), s1hi = mhi >> 1;            // scalar
					const uint32_t qs = pr + (mlo & 1u) - 1u;                                   // scalar
					uint32_t q = qs + __builtin_amdgcn_mbcnt_hi(s1hi, __builtin_amdgcn_mbcnt_lo(s1lo, 0u));
					q = min(q, nzc - 1u);                                    // lanes past the end of the last block
					const uint2 se = s_cse[q];
					obp = se.x + (t - s_cpref[q]) * R;
					onv = ok ? min((uint32_t)R, se.y - obp) : 0u;
					oav = s_caval[q];
				};
				const char *bbase = reinterpret_cast<const char *>(m.btup);
				auto fetch = [&](uint32_t bp_) -> BPiece {
					if (ABL(ep, 64)) bp_ &= 0xFFFFu;
					// 12 * bp as a 32-bit offset from a scalar base where B is small enough (always, short of 3.5e8 tuples)
					if (narrow) return *reinterpret_cast<const BPiece *>(bbase + (uint32_t)((bp_ << 3) + (bp_ << 2)));
					return *reinterpret_cast<const BPiece *>(bbase + (uint64_t)bp_ * 12u);
				};
				auto accumulate = [&](const BPiece &piece, uint32_t nv_, double av_) {
#pragma unroll
					for (int u = 0; u < R; ++u) {
						// a tuple past the segment's end goes to the lane's dump slot: straight-line code, no exec juggling
						const uint32_t slot = (uint32_t)u < nv_ ? (ABL(ep, 64) ? (piece.w[3 * u] & (W - 1)) : piece.w[3 * u] - wbase) : (uint32_t)W + lane;
						if (ABL(ep, 32)) { if (piece.w[3 * u + 2] == 0x7FF12345u) acc[slot] = av_; }          // no LDS accumulate
						else if (MODE == MODE_COUNT) acc[slot] = 1.0;        // structural: touched
						else {
							const double pv = av_ * __hiloint2double((int)piece.w[3 * u + 2], (int)piece.w[3 * u + 1]);
							atomicAdd(&acc[slot], pv);
							if (PAT && (uint32_t)u < nv_) pat_note(pat, pv);
						}
					}
				};
				STAMP(4);
#if DENSE_DEPTH == 2
				// two pieces in flight: the loads of step s+1 are issued before the products of step s are
				// accumulated, the lookup of step s+2 runs under them
				uint32_t bp0, nv0, bp1 = 0, nv1 = 0; double av0, av1 = 0.0;
				lookup(0, bp0, nv0, av0);
				BPiece p0 = fetch(bp0), p1 = p0;
				if (nsteps > 1) lookup(1, bp1, nv1, av1);
				for (uint32_t step = 0; step < nsteps; ++step) {
					STAMP_COUNT(10);
					if (step + 1 < nsteps) p1 = fetch(bp1);                  // uniform
					uint32_t bp2 = 0, nv2 = 0; double av2 = 0.0;
					if (step + 2 < nsteps) lookup(step + 2, bp2, nv2, av2);  // uniform
					accumulate(p0, nv0, av0);
					p0 = p1; nv0 = nv1; av0 = av1;
					bp1 = bp2; nv1 = nv2; av1 = av2;
				}
#else
				uint32_t bp, nv; double av;
				lookup(0, bp, nv, av);
				for (uint32_t step = 0; step < nsteps; ++step) {
					STAMP_COUNT(10);
					const BPiece piece = fetch(bp);
					uint32_t nbp = bp, nnv = 0; double nav = 0.0;
					if (step + 1 < nsteps) lookup(step + 1, nbp, nnv, nav);     // uniform branch
					accumulate(piece, nv, av);
					bp = nbp; nv = nnv; av = nav;
				}
#endif
				Q0 = run;
				STAMP(5);
			}
			if (PAT) { pat_publish(pat, &s_pat); pnseg += nzc; }
			lds_barrier();                                          // B3: segment tables and bitmap are free again
			STAMP(6);
			for (int q = tid; q < NWORD; q += NT) s_bmask[q] = 0ull;
			lo = lo2; len = hi2 - lo2; a = a2;
		}
		// stage C of the cell pipeline: B segment bounds of the next cell's first chunk (in flight during the scan-out)
		{
			uint32_t l2, h2;
			seg_bounds(nk, rec1.wa, l2, h2);
			nlo = l2; nlen = nact ? h2 - l2 : 0u;
		}
		if (ABL(ep, 16)) continue;
		STAMP(0);
		// ---- scan-out: wave wv owns groups [wv*GPW, (wv+1)*GPW) -> ascending columns.  (The last barrier of the
		// chunk loop, B3, has every accumulate of this cell behind it; a cell with no product at all skips it and
		// scans zeros, which is still ordered by the next cell's B1.)
		// Kept lean, it runs once per cell over all W slots: with C = 1 and no scale vectors the emitted value IS the
		// sum (sum * 1 * 1 * 1, multiply_sparse.hpp:242, is the same bits), and the index hash of column J0 + lane is
		// mix64's product evaluated as X0 + lane * K with the group's X0 kept in scalar registers.
		const double pthr = PAT ? pat_threshold(&s_pat, pnseg) : -1.0;
		const uint32_t pbeg = beg, pend = end;
		double v[GPW];
		uint64_t nzmask[GPW];
		uint32_t wcount = 0;
		constexpr unsigned long long MIXK = 0x9E3779B97F4A7C15ull;
		unsigned long long X0 = ((((unsigned long long)(uint32_t)rowid) << 32) | (unsigned long long)(wbase + wv * GPW * 64u)) * MIXK;   // uniform
		// (pinned to scalar registers: left to itself the compiler re-derives the product per group with v_mad_u64_u32)
		X0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(X0 >> 32)) << 32) |
			(unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)X0);
		double r_sum = 0;
#pragma unroll
		for (int gi = 0; gi < GPW; ++gi) {
			int grp = wv * GPW + gi;
			double x = acc[grp * 64 + lane];
			acc[grp * 64 + lane] = CLEAN;
			bool ok;
			if (MODE != MODE_COUNT && PAT) {
				const bool touched = __double_as_longlong(x) != (long long)0x8000000000000000ull;
				x = pat_fix_wave(touched && !(fabs(x) > pthr), x, (int32_t)(wbase + grp * 64 + lane), m, pbeg, pend);
			}
			if (MODE == MODE_COUNT) ok = (x != 0) && col_allowed(ep, (int32_t)(wbase + grp * 64 + lane));
			else if (plain) ok = x != 0;
			else ok = emit_value(ep, a_scale, (int32_t)(wbase + grp * 64 + lane), x, &x);
			v[gi] = x;
			nzmask[gi] = __ballot(ok);
			wcount += (uint32_t)__popcll(nzmask[gi]);
			if (MODE == MODE_DIGEST) {
				unsigned long long h = X0 + laneK;
				h ^= h >> 29;
				d_hash += ok ? h : 0ull;
				if (plain && !PAT) r_sum += x;                          // (a slot that is not emitted holds +-0)
				else r_sum += ok ? x : 0.0;
				X0 += 64ull * MIXK;
			}
		}
		if (MODE == MODE_DIGEST) {
			if (lane == 0) d_cnt += wcount;
			d_sum += r_sum;
			if (sk.row_nnz) {
				double rs = wave_reduce_sum(r_sum);
				if (lane == 0 && wcount) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], (unsigned long long)wcount); atomicAdd(&sk.row_sum[rowid], rs); }
			}
		} else {
			if (lane == 0) s_wcnt[wv] = wcount;
			__syncthreads();
			uint32_t wbefore = 0, wtotal = 0;
#pragma unroll
			for (int q = 0; q < NW; ++q) { uint32_t t = s_wcnt[q]; if (q < (int)wv) wbefore += t; wtotal += t; }
			if (MODE == MODE_COUNT) {
				if (tid == 0) sk.segcount[cell.seg] = wtotal;
			} else {
				int64_t o = sk.segoff[cell.seg] + wbefore;
#pragma unroll
				for (int gi = 0; gi < GPW; ++gi) {
					uint64_t mk = nzmask[gi];
					if ((mk >> lane) & 1ull) {
						int64_t oo = o + __popcll(mk & lanemask_lt());
						sk.out_i[oo] = rowid;
						sk.out_j[oo] = (int32_t)(wbase + (wv * GPW + gi) * 64 + lane);
						sk.out_v[oo] = v[gi];
					}
					o += __popcll(mk);
				}
				if (tid == 0) sk.segactual[cell.seg] = wtotal;
			}
			__syncthreads();                                        // s_wcnt is reused by the next cell
		}
		STAMP(7);
		if (PAT) { lds_barrier(); if (tid == 0) pat_reset(&s_pat); }     // every thread has read the cell's record
	}
#ifdef SPSAMD_STAMPS
	if (tid == 0 && sk.stamps) for (int i = 0; i < 12; ++i) sk.stamps[(size_t)blockIdx.x * 12 + i] = st_[i];
#endif
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// ====================================================================== tiles, second generation
//
// A tile is up to 16 cells of ONE heavy row with few A tuples (L <= 256) that share one expansion
// of the row's A tuples: thread t owns (cell t / Lp, tuple t % Lp), Lp = L rounded up to a power
// of two.  The expansion works in ITEMS of R consecutive B tuples like k_dense: a non-empty
// segment is ceil(len / R) items, the items of the tile are numbered cell by cell with every
// cell's first item at a multiple of 64, every segment sets the bit of its first item, and each
// wave keeps the bitmap and its popcount prefix in registers: an item's segment is found with
// v_readlane + mbcnt and ONE LDS round trip.
template <int NT, int NWORD>
struct TileX {
	uint16_t cpref[NT + 2];          // compacted segments: first item (a tile has at most 64 NWORD <= 16384 items)
	uint2 cse[NT];                   // first tuple of the segment, one past its last
	double caval[NT];                // the A value
	unsigned long long bmask[NWORD]; // first-item bits
	uint32_t scrL[2][NT / 64], scrN[2][NT / 64];
	uint32_t cellI[TILE_MAXCELLS + 1];       // first item of every cell (+ end), multiples of 64
	uint32_t cellseg[TILE_MAXCELLS];         // output segment id of every cell
	uint32_t cellw[TILE_MAXCELLS];           // wa | wb << 16 of every cell
};

// Contains two barriers (B1 after the per-wave totals, B2 after the tables are written); the first one also
// separates the previous tile's last LDS traffic from this tile's.
template <int NT, int NWORD>
__device__ __forceinline__ void tile_expand(TileX<NT, NWORD> &X, uint32_t lsh, uint32_t ncells, uint32_t lo, uint32_t len, double a,
	uint32_t myseg, uint32_t myw, uint32_t &flip, uint32_t *total_out, uint32_t *nzc_out)
{
	constexpr int NW = NT / 64;
	constexpr int R = DENSE_R;
	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	const uint32_t myc = tid >> lsh, myei = tid & ((1u << lsh) - 1u);
	const uint32_t items = (len + R - 1) / R;
	uint32_t incl;                                                  // inclusive item prefix inside the wave, cell starts aligned
	{
		// cells are runs of Lp = 2^lsh consecutive threads: whole waves (Lp >= 64) or 64 / Lp cells per wave
		const uint32_t x = wave_inclusive_scan_u32(items);
		if (lsh < 6) {
			// several cells in this wave: the start of each is rounded up to 64 items, serially over the cells of
			// the wave with wave-uniform lane reads (cells are numbered from thread 0 and a tile has at most 16)
			const uint32_t cells_here = min(64u >> lsh, TILE_MAXCELLS);
			uint32_t carry = 0, out = 0;                                // carry: aligned total before the current cell
			for (uint32_t cc = 0; cc < cells_here; ++cc) {
				const uint32_t first_lane = cc << lsh, last_lane = first_lane + (1u << lsh) - 1u;
				const uint32_t before = first_lane ? (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(first_lane - 1u)) : 0u;
				const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)last_lane);
				if ((lane >> lsh) == cc) out = carry + (x - before);
				carry = (carry + (upto - before) + 63u) & ~63u;
			}
			incl = out;
			if (lane == 63) X.scrL[flip][wv] = carry;                   // aligned items of the whole wave
		} else {
			incl = x;
			if (lane == 63) X.scrL[flip][wv] = x;                       // a cell spans 2^(lsh-6) whole waves: aligned below
		}
	}
	const uint64_t nzm = __ballot(len != 0);
	const uint32_t wrank = (uint32_t)__popcll(nzm & lanemask_lt());
	if (lane == 0) X.scrN[flip][wv] = (uint32_t)__popcll(nzm);
	lds_barrier();                                                  // B1
	uint32_t baseL = 0, baseN = 0, total = 0, nzc = 0;
	{
		const uint32_t wpc = lsh > 6 ? (1u << (lsh - 6)) : 1u;          // waves per cell
#pragma unroll
		for (int q = 0; q < NW; ++q) {
			const uint32_t l = X.scrL[flip][q], n = X.scrN[flip][q];
			if ((q & (wpc - 1u)) == 0) total = (total + 63u) & ~63u;        // a cell begins with this wave
			if (q == (int)wv) baseL = total;
			if (q < (int)wv) baseN += n;
			total += l; nzc += n;
		}
		total = (total + 63u) & ~63u;
	}
	flip ^= 1u;
	const uint32_t myfirst = baseL + incl - items;
	if (len) {
		const uint32_t rank = baseN + wrank;
		X.cpref[rank] = (uint16_t)myfirst;
		X.cse[rank] = make_uint2(lo, lo + len);
		X.caval[rank] = a;
		atomicOr(&X.bmask[myfirst >> 6], 1ull << (myfirst & 63u));
	}
	if (myei == 0 && myc < ncells) { X.cellI[myc] = myfirst; X.cellseg[myc] = myseg; X.cellw[myc] = myw; }    // a cell's first thread: its items start here
	if (tid == 0) X.cellI[ncells] = total;
	lds_barrier();                                                  // B2
	*total_out = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
	*nzc_out = (uint32_t)__builtin_amdgcn_readfirstlane((int)nzc);
}

// The per-wave register copy of the item bitmap and its popcount prefix: word x * 64 + l in lane l.
template <int WPL>
struct TileTab { unsigned long long mw[WPL]; uint32_t pre[WPL]; };

template <int NT, int NWORD>
__device__ __forceinline__ void tile_tables(const TileX<NT, NWORD> &X, TileTab<NWORD / 64> &tab)
{
	uint32_t run = 0;
#pragma unroll
	for (int x = 0; x < NWORD / 64; ++x) {
		tab.mw[x] = X.bmask[x * 64 + lane_id()];
		const uint32_t cnt = (uint32_t)__popcll(tab.mw[x]);
		const uint32_t inc2 = wave_inclusive_scan_u32(cnt);
		tab.pre[x] = run + inc2 - cnt;
		run += (uint32_t)__builtin_amdgcn_readlane((int)inc2, 63);
	}
}

// Item (b << 6) + lane of block b (wave-uniform) -> first tuple, number of valid tuples, A value.  i1 = end of the
// cell's item range.  An aligned cell start leaves positions at the END of the previous cell's last block that hold no
// item: they resolve to that cell's last segment with an offset past its end -- no valid tuple.
template <int NT, int NWORD>
__device__ __forceinline__ void tile_lookup(const TileX<NT, NWORD> &X, const TileTab<NWORD / 64> &tab, uint32_t nzc, uint32_t b, uint32_t i1,
	uint32_t &obp, uint32_t &onv, double &oav)
{
	constexpr int R = DENSE_R;
	const uint32_t t = (b << 6) + lane_id();
	uint32_t mlo = 0, mhi = 0, pr = 0;
#pragma unroll
	for (int x = 0; x < NWORD / 64; ++x) {
		if ((b >> 6) == (uint32_t)x) {                                  // uniform
			mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tab.mw[x], (int)(b & 63u));
			mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tab.mw[x] >> 32), (int)(b & 63u));
			pr = (uint32_t)__builtin_amdgcn_readlane((int)tab.pre[x], (int)(b & 63u));
		}
	}
	const uint32_t s1lo = (mlo >> 1) | (mhi << 31), s1hi = mhi >> 1;
	const uint32_t qs = pr + (mlo & 1u) - 1u;
	uint32_t q = qs + __builtin_amdgcn_mbcnt_hi(s1hi, __builtin_amdgcn_mbcnt_lo(s1lo, 0u));
	q = min(q, nzc - 1u);
	const uint2 se = X.cse[q];
	obp = se.x + (t - (uint32_t)X.cpref[q]) * R;
	onv = (t < i1 && obp < se.y) ? min((uint32_t)R, se.y - obp) : 0u;
	oav = X.caval[q];
}

__device__ __forceinline__ BPiece fetch_piece(const char *bbase, uint32_t bp, uint32_t narrow)
{
	// 12 * bp as a 32-bit offset from a scalar base where B is small enough (always, short of 3.5e8 tuples)
	if (narrow) return *reinterpret_cast<const BPiece *>(bbase + (uint32_t)((bp << 3) + (bp << 2)));
	return *reinterpret_cast<const BPiece *>(bbase + (uint64_t)bp * 12u);
}

// ---- hash tiles: cells are ranges [wa, wb) of sparse column windows, accumulated in the LDS hash table ----
// Insertion: the R first probes of a lane are in flight together (ds_cmpswap with return), the rare collisions are
// then walked one by one; the values follow with ds_add_f64; the newly occupied slots of a step are appended to the
// occupied list with one LDS fetch-add per wave.  The first block of the NEXT cell is looked up and its B tuples
// requested before the current cell is emitted, so that latency is hidden behind the emission.
constexpr int TILE2_NT = 512;
constexpr int TILE2_T = 4096;
constexpr int TILE2_ITEMS = 8192;        // items per tile: bitmap of 128 words, two per lane

template <int MODE, bool PAT>
__global__ __launch_bounds__(TILE2_NT, 4) void k_hash_tiles2(const Tile *tiles, uint32_t ntile, const TCell *tcells, RowMeta m,
	const uint32_t *bwin, uint32_t nwin1, uint32_t narrow, EmitParams ep, SinkParams sk)
{
	constexpr int NT = TILE2_NT, T = TILE2_T, NW = NT / 64, R = DENSE_R;
	constexpr int NWORD = TILE2_ITEMS / 64;
	constexpr int MAXST = 3;                 // 64-item blocks of one cell per wave (T / 2 products: at most T/2/R + L items, plus alignment)
	__shared__ int32_t h_key[T + 64];        // + one dump slot per lane: the first probes are issued unconditionally
	__shared__ double h_val[MODE == MODE_COUNT ? 1 : T];
	__shared__ uint16_t occ[T / 2];
	__shared__ uint64_t s_sort[MODE == MODE_STORE ? T / 2 : 1];
	__shared__ uint16_t s_cnt[MODE == MODE_STORE ? 16 * ((T / 2 + NT - 1) / NT) * (NT / 64) : 1];
	__shared__ TileX<NT, NWORD> X;
	__shared__ uint32_t scr32[NW + 1];
	__shared__ uint32_t s_nocc;
	__shared__ PatCell s_pat;
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];

	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	for (int q = tid; q < T; q += NT) { h_key[q] = -1; if (MODE != MODE_COUNT) h_val[q] = 0.0; }
	if (tid < 64) h_key[T + tid] = -1;
	for (int q = tid; q < NWORD; q += NT) X.bmask[q] = 0ull;
	if (tid == 0) s_nocc = 0;
	PatAcc pat; pat_init(pat);
	if (tid == 0) pat_reset(&s_pat);
	DigestAcc dacc{0, 0, 0.0};
	uint32_t flip = 0;
	const char *bbase = reinterpret_cast<const char *>(m.btup);
#ifdef SPSAMD_STAMPS
	unsigned long long st_[12] = {}; unsigned long long st_t = clock64();
#endif

	const uint32_t stride = gridDim.x;
	const uint32_t tlast = ntile - 1;
	// three-stage branch-free prefetch: tile record -> (A tuple, cell window range) -> B segment bounds
	Tile rec1 = tiles[min(blockIdx.x, tlast)];
	Tile rec2 = tiles[min(blockIdx.x + stride, tlast)];
	uint32_t nlo, nlen, nseg_, nw_; double na;
	{
		const uint32_t L = rec1.end - rec1.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		const uint32_t c = tid >> lsh, ei = tid & ((1u << lsh) - 1u);
		const bool act = c < rec1.ncells && ei < L;
		const uint32_t ec = rec1.beg + (ei < L ? ei : 0u);
		const TCell tc = tcells[rec1.first + (c < rec1.ncells ? c : 0u)];
		const uint32_t *bw = bwin + (uint64_t)(uint32_t)m.acol[ec] * nwin1;
		const uint32_t lo = bw[tc.wa], hi = bw[tc.wb];
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u; nseg_ = tc.seg; nw_ = (uint32_t)tc.wa | ((uint32_t)tc.wb << 16);
	}
	__syncthreads();
	for (uint32_t ti = blockIdx.x; ti < ntile; ti += stride) {
		const Tile tile = rec1;
		const uint32_t lo = nlo, len = nlen, myseg = nseg_, myw = nw_; const double a = na;
		const uint32_t L = tile.end - tile.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		lsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)lsh);
		const int32_t rowid = tile.rowid;
		// stage A / B for the next tile
		rec1 = rec2;
		rec2 = tiles[min(ti + 2 * stride, tlast)];
		const bool has_next = ti + stride < ntile;
		const uint32_t nL = rec1.end - rec1.beg;
		uint32_t nsh = 0;
		while ((1u << nsh) < nL) ++nsh;
		const uint32_t nc = tid >> nsh, nei = tid & ((1u << nsh) - 1u);
		const bool nact = has_next && nc < rec1.ncells && nei < nL;
		const uint32_t nec = rec1.beg + (nei < nL ? nei : 0u);
		const TCell ntc = tcells[rec1.first + (nc < rec1.ncells ? nc : 0u)];
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];

		uint32_t total, nzc;
		STAMP_COUNT(8);
		STAMP(0);
		tile_expand(X, lsh, tile.ncells, lo, len, a, myseg, myw, flip, &total, &nzc);
		STAMP(1);
		// stage C: B segment bounds of the next tile
		{
			const uint32_t *bw = bwin + (uint64_t)(uint32_t)nk * nwin1;
			const uint32_t nlo_ = bw[ntc.wa], nhi_ = bw[ntc.wb];
			nlo = nlo_; nlen = nact ? nhi_ - nlo_ : 0u; nseg_ = ntc.seg; nw_ = (uint32_t)ntc.wa | ((uint32_t)ntc.wb << 16);
		}
		if (total == 0 || nzc == 0) {                               // uniform; cannot happen for real tiles
			if (MODE != MODE_DIGEST) for (uint32_t c = tid; c < tile.ncells; c += NT) { if (MODE == MODE_COUNT) sk.segcount[X.cellseg[c]] = 0; else sk.segactual[X.cellseg[c]] = 0; }
			continue;
		}
		TileTab<NWORD / 64> tab;
		tile_tables(X, tab);

		// first block of cell 0, prefetched like every later cell's
		uint32_t pbp, pnv; double pav;
		tile_lookup(X, tab, nzc, (X.cellI[0] >> 6) + wv, X.cellI[1], pbp, pnv, pav);
		BPiece ppiece = fetch_piece(bbase, pbp, narrow);
		STAMP(2);
		for (uint32_t c = 0; c < tile.ncells; ++c) {
			STAMP_COUNT(9);
			const uint32_t i0 = X.cellI[c], i1 = X.cellI[c + 1];
			const uint32_t seg = X.cellseg[c];
			const uint32_t nblk = (i1 - i0) >> 6;
			if (nblk > (uint32_t)(MAXST * NW) && tid == 0) atomicOr(sk.err, 2u);     // never: k_cells bounds a cell's items
#pragma unroll
			for (int st = 0; st < MAXST; ++st) {
				const uint32_t bl = (uint32_t)st * NW + wv;
				if (st > 0 && bl >= nblk) break;                            // wave-uniform (step 0 always runs: its piece is prefetched)
				uint32_t nv; double av; BPiece piece;
				if (st == 0) { nv = bl < nblk ? pnv : 0u; av = pav; piece = ppiece; }
				else {
					uint32_t bp;
					tile_lookup(X, tab, nzc, (i0 >> 6) + bl, i1, bp, nv, av);
					piece = fetch_piece(bbase, bp, narrow);
				}
				// ---- R first probes in flight, then the collisions
				uint32_t h[R]; int32_t old[R]; bool isnew[R];
#pragma unroll
				for (int u = 0; u < R; ++u) {
					// (a tuple past the segment's end probes the lane's dump slot: no branch, so the R atomics overlap)
					h[u] = (uint32_t)u < nv ? hash_slot<T>((int32_t)piece.w[3 * u]) : (uint32_t)T + lane;
					old[u] = atomicCAS(&h_key[h[u]], -1, (int32_t)piece.w[3 * u]);
				}
				uint64_t newmask[R]; uint32_t nnew = 0;
#pragma unroll
				for (int u = 0; u < R; ++u) {
					const int32_t col = (int32_t)piece.w[3 * u];
					isnew[u] = false;
					if ((uint32_t)u < nv) {
						int32_t o = old[u];
						while (o != -1 && o != col) {
							h[u] = (h[u] + 1) & (T - 1);
							o = atomicCAS(&h_key[h[u]], -1, col);
						}
						isnew[u] = o == -1;
						if (MODE != MODE_COUNT) {
							const double pv = av * __hiloint2double((int)piece.w[3 * u + 2], (int)piece.w[3 * u + 1]);
							atomicAdd(&h_val[h[u]], pv);
							if (PAT) pat_note(pat, pv);
						}
					}
					newmask[u] = __ballot(isnew[u]);
					nnew += (uint32_t)__popcll(newmask[u]);
				}
				if (nnew) {                                                 // uniform: one LDS fetch-add per wave and step
					uint32_t base = 0;
					if (lane == 0) base = lds_add_rtn_u32(&s_nocc, nnew);
					base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
					for (int u = 0; u < R; ++u) {
						if (isnew[u]) occ[base + __popcll(newmask[u] & lanemask_lt())] = (uint16_t)h[u];
						base += (uint32_t)__popcll(newmask[u]);
					}
				}
			}
			if (PAT) pat_publish(pat, &s_pat);
			STAMP(3);
			lds_barrier();                                          // the cell's products are in the table
			STAMP(4);
			const uint32_t nocc = s_nocc;
			// the next cell's first block: lookup and B request issued now, consumed after the emission
			if (c + 1 < tile.ncells) {                              // uniform
				tile_lookup(X, tab, nzc, (i1 >> 6) + wv, X.cellI[c + 2], pbp, pnv, pav);
				ppiece = fetch_piece(bbase, pbp, narrow);
			}
			uint32_t colbase = 0, colbits = 0;
			if (MODE == MODE_STORE) {
				const uint32_t wab = X.cellw[c];
				const uint32_t wa = wab & 0xFFFFu, wb = wab >> 16;
				colbase = wa << ep.wshift;
				colbits = ep.wshift + (wb - wa > 1 ? 32 - __builtin_clz(wb - wa - 1u) : 0);
			}
			STAMP(5);
			const double pthr = PAT ? pat_threshold(&s_pat, tile.end - tile.beg) : -1.0;
			hash_emit<T, NT, MODE, PAT>(nocc, rowid, seg, ep, sk, h_key, h_val, occ, s_sort, scr32, dacc, s_cnt, colbase, colbits, m, tile.beg, tile.end, pthr);
			if (PAT) { lds_barrier(); if (tid == 0) pat_reset(&s_pat); }
			if (tid == 0) s_nocc = 0;
			STAMP(6);
			lds_barrier();                                          // table clean, counter reset: next cell may insert
			STAMP(7);
		}
		for (int q = tid; q < NWORD; q += NT) X.bmask[q] = 0ull;     // (every wave is past its last lookup)
	}
#ifdef SPSAMD_STAMPS
	if (tid == 0 && sk.stamps) for (int i = 0; i < 12; ++i) sk.stamps[(size_t)blockIdx.x * 12 + i] = st_[i];
#endif
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, dacc.cnt, dacc.hash, dacc.sum, s_u64, s_f64);
}

// ---- bitmap tiles: cells are ranges [wa, wb) of at most BM_WORDS * 64 columns, accumulated by RANK ---------
// The cell's products are looked up and read once and stay in registers (<= 3 items of R tuples per lane):
//   1. every product sets the bit of its column in a bitmap over the cell's column range (ds_or, no return value);
//   2. the bitmap's words are prefix-summed (popcounts): the RANK of a column = set bits below it;
//   3. every product adds its value to acc[rank] (ds_add_f64) and notes its column in colof[rank];
//   4. acc[0 .. distinct) IS the cell's output in ascending column order: emitted and zeroed, bitmap words cleared.
// No probing, no compare-and-swap chains, no list of occupied slots, no sort for the COO order; the structural count of
// a cell (COUNT launch) is just the popcount total.  LDS: bitmap 16 KB + prefix 4 KB + acc 32 KB + columns 16 KB: with no
// table of keys a cell may hold 4096 products instead of a hash table's 2048 -- and per-cell bookkeeping is what sets the
// tiles' time (42.9 -> 41.8 ms on cfg2 from the cell size alone).
constexpr int BM_NT = 512;
#ifndef BM_WORDS_V
#define BM_WORDS_V 2048
#endif
constexpr int BM_WORDS = BM_WORDS_V;      // bitmap words: 2048 = 131072 columns = 16 windows of 8192 (8 of 16384)
#ifndef BM_MAXOUT_V
#define BM_MAXOUT_V 4096
#endif
constexpr int BM_MAXOUT = BM_MAXOUT_V;   // distinct columns of a cell (<= its products)
constexpr int BM_ITEMS = 8192;           // items per tile

template <int MODE, bool PAT>
__global__ __launch_bounds__(BM_NT, 4) void k_bm_tiles(const Tile *tiles, uint32_t ntile, const TCell *tcells, RowMeta m,
	const uint32_t *bwin, uint32_t nwin1, uint32_t narrow, EmitParams ep, SinkParams sk)
{
	constexpr int NT = BM_NT, NW = NT / 64, R = DENSE_R;
	constexpr int NWORD = BM_ITEMS / 64;
	constexpr int MAXST = (BM_MAXOUT / R + (int)TILE_LMAX + 64 + NT - 1) / NT;      // 64-item blocks of one cell per wave
	constexpr int WPT = BM_WORDS / NT;       // bitmap words per thread in the scan (4)
	__shared__ __attribute__((aligned(16))) unsigned long long bm[BM_WORDS];
	__shared__ __attribute__((aligned(8))) uint16_t bpre[BM_WORDS];
	__shared__ double acc[BM_MAXOUT];
	__shared__ uint32_t colof[BM_MAXOUT];                       // column (relative to the cell's first) of every rank
	__shared__ TileX<NT, NWORD> X;
	__shared__ uint32_t s_wtot[2][NW];
	__shared__ uint32_t s_wbase[NW];
	__shared__ PatCell s_pat;                                   // EXACT_PATTERN: sum of |products| and sign mix of the current cell
	__shared__ uint32_t s_scan[NW + 1];
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];

	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	for (int q = tid; q < BM_WORDS; q += NT) bm[q] = 0ull;
	for (int q = tid; q < BM_MAXOUT; q += NT) acc[q] = 0.0;
	for (int q = tid; q < NWORD; q += NT) X.bmask[q] = 0ull;
	const bool plain = ep.C == 1.0 && !ep.si_pos && !ep.sk_pos;
	unsigned long long d_cnt = 0, d_hash = 0; double d_sum = 0;
	PatAcc pat; pat_init(pat);
	if (PAT && tid == 0) pat_reset(&s_pat);
	uint32_t flip = 0, sflip = 0;
	const char *bbase = reinterpret_cast<const char *>(m.btup);
#ifdef SPSAMD_STAMPS
	unsigned long long st_[12] = {}; unsigned long long st_t = clock64();
#endif

	const uint32_t stride = gridDim.x;
	const uint32_t tlast = ntile - 1;
	Tile rec1 = tiles[min(blockIdx.x, tlast)];
	Tile rec2 = tiles[min(blockIdx.x + stride, tlast)];
	uint32_t nlo, nlen, nseg_, nw_; double na;
	{
		const uint32_t L = rec1.end - rec1.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		const uint32_t c = tid >> lsh, ei = tid & ((1u << lsh) - 1u);
		const bool act = c < rec1.ncells && ei < L;
		const uint32_t ec = rec1.beg + (ei < L ? ei : 0u);
		const TCell tc = tcells[rec1.first + (c < rec1.ncells ? c : 0u)];
		const uint32_t *bw = bwin + (uint64_t)(uint32_t)m.acol[ec] * nwin1;
		const uint32_t lo = bw[tc.wa], hi = bw[tc.wb];
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u; nseg_ = tc.seg; nw_ = (uint32_t)tc.wa | ((uint32_t)tc.wb << 16);
	}
	__syncthreads();
	for (uint32_t ti = blockIdx.x; ti < ntile; ti += stride) {
		const Tile tile = rec1;
		const uint32_t lo = nlo, len = nlen, myseg = nseg_, myw = nw_; const double a = na;
		const uint32_t L = tile.end - tile.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		lsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)lsh);
		const int32_t rowid = tile.rowid;
		const double a_scale = row_scale(ep, rowid);
		rec1 = rec2;
		rec2 = tiles[min(ti + 2 * stride, tlast)];
		const bool has_next = ti + stride < ntile;
		const uint32_t nL = rec1.end - rec1.beg;
		uint32_t nsh = 0;
		while ((1u << nsh) < nL) ++nsh;
		const uint32_t nc = tid >> nsh, nei = tid & ((1u << nsh) - 1u);
		const bool nact = has_next && nc < rec1.ncells && nei < nL;
		const uint32_t nec = rec1.beg + (nei < nL ? nei : 0u);
		const TCell ntc = tcells[rec1.first + (nc < rec1.ncells ? nc : 0u)];
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];

		uint32_t total, nzc;
		STAMP_COUNT(10);
		STAMP(0);
		tile_expand(X, lsh, tile.ncells, lo, len, a, myseg, myw, flip, &total, &nzc);
		STAMP(1);
		{
			const uint32_t *bw = bwin + (uint64_t)(uint32_t)nk * nwin1;
			const uint32_t nlo_ = bw[ntc.wa], nhi_ = bw[ntc.wb];
			nlo = nlo_; nlen = nact ? nhi_ - nlo_ : 0u; nseg_ = ntc.seg; nw_ = (uint32_t)ntc.wa | ((uint32_t)ntc.wb << 16);
		}
		if (total == 0 || nzc == 0) {                               // uniform; cannot happen for real tiles
			if (MODE != MODE_DIGEST) for (uint32_t c = tid; c < tile.ncells; c += NT) { if (MODE == MODE_COUNT) sk.segcount[X.cellseg[c]] = 0; else sk.segactual[X.cellseg[c]] = 0; }
			continue;
		}
		TileTab<NWORD / 64> tab;
		tile_tables(X, tab);

		// first block of cell 0, prefetched like every later cell's
		uint32_t pbp, pnv; double pav;
		tile_lookup(X, tab, nzc, (X.cellI[0] >> 6) + wv, X.cellI[1], pbp, pnv, pav);
		BPiece ppiece = fetch_piece(bbase, pbp, narrow);
		// ... and the second one (a cell of more than NW blocks: the usual case), so that no step of a typical cell waits
		// for memory inside the cell
#ifndef BM_D2
#define BM_D2 1
#endif
		constexpr bool D2 = BM_D2 && MODE != MODE_STORE;                     // (the COO variant has no registers left for it)
		uint32_t qnv = 0; double qav = 0; BPiece qpiece;
		if (D2 && ((X.cellI[1] - X.cellI[0]) >> 6) > (uint32_t)NW + wv) {     // wave-uniform
			uint32_t qbp;
			tile_lookup(X, tab, nzc, (X.cellI[0] >> 6) + NW + wv, X.cellI[1], qbp, qnv, qav);
			qpiece = fetch_piece(bbase, qbp, narrow);
		}
		STAMP(0);
		for (uint32_t c = 0; c < tile.ncells; ++c) {
			STAMP_COUNT(11);
			const uint32_t i0 = X.cellI[c], i1 = X.cellI[c + 1];
			const uint32_t seg = X.cellseg[c];
			const uint32_t nblk = (i1 - i0) >> 6;
			const uint32_t wab = X.cellw[c];
			const uint32_t colbase = (wab & 0xFFFFu) << ep.wshift;
			const uint32_t nwords = ((wab >> 16) - (wab & 0xFFFFu)) << (ep.wshift - 6);      // bitmap words of the cell's column range
			if ((nblk > (uint32_t)(MAXST * NW) || nwords > (uint32_t)BM_WORDS) && tid == 0) atomicOr(sk.err, 2u);   // never: k_cells bounds both
			// ---- 1. products into registers, column bits into the bitmap
			uint32_t krel[MAXST][R]; double kval[MODE == MODE_COUNT ? 1 : MAXST][MODE == MODE_COUNT ? 1 : R];
#pragma unroll
			for (int st = 0; st < MAXST; ++st) {
#pragma unroll
				for (int u = 0; u < R; ++u) krel[st][u] = 0xFFFFFFFFu;
				const uint32_t bl = (uint32_t)st * NW + wv;
				if (st > 0 && bl >= nblk) continue;                         // wave-uniform (step 0's piece is prefetched)
				uint32_t nv; double av; BPiece piece;
				if (st == 0) { nv = bl < nblk ? pnv : 0u; av = pav; piece = ppiece; }
				else if (D2 && st == 1) { nv = qnv; av = qav; piece = qpiece; }
				else {
					uint32_t bp;
					tile_lookup(X, tab, nzc, (i0 >> 6) + bl, i1, bp, nv, av);
					piece = fetch_piece(bbase, bp, narrow);
				}
#pragma unroll
				for (int u = 0; u < R; ++u) {
					if ((uint32_t)u < nv) {
						const uint32_t rel = piece.w[3 * u] - colbase;
						krel[st][u] = rel;
						if constexpr (MODE != MODE_COUNT) {
							kval[st][u] = av * __hiloint2double((int)piece.w[3 * u + 2], (int)piece.w[3 * u + 1]);
							if (PAT) pat_note(pat, kval[st][u]);
						}
						atomicOr(reinterpret_cast<uint32_t *>(bm) + (rel >> 5), 1u << (rel & 31u));      // (32-bit halves: half the bank traffic of a 64-bit or)
					}
				}
			}
			if (PAT && MODE != MODE_COUNT) pat_publish(pat, &s_pat);     // (complete at the barrier)
			STAMP(2);
			lds_barrier();                                          // the bitmap is complete
			STAMP(3);
			// EXACT_PATTERN: every thread takes the cell's bound now; the record is reset after the next barrier (all have
			// read it) and long before the next cell's waves add to it
			const double pthr = (PAT && MODE != MODE_COUNT) ? pat_threshold(&s_pat, tile.end - tile.beg) : -1.0;
			// ---- 2. rank prefix of the bitmap words: thread t owns words [WPT t, WPT t + WPT); bpre holds the prefix INSIDE
			// the wave's 64 WPT words, the waves' bases go to s_wbase after the barrier (every wave computes and writes the
			// same eight values and reads back its own writes: no barrier needed for them)
			uint32_t wcnt[WPT], mine = 0;
			{
				unsigned long long wd[WPT];
#pragma unroll
				for (int x = 0; x < WPT; ++x) wd[x] = bm[tid * WPT + x];      // (words beyond the cell's range are clean: zero)
				// the next cell's first block(s): lookup and B request issued now, consumed after this cell is done
				if (c + 1 < tile.ncells) {                              // uniform
					const uint32_t i2 = X.cellI[c + 2];
					tile_lookup(X, tab, nzc, (i1 >> 6) + wv, i2, pbp, pnv, pav);
					ppiece = fetch_piece(bbase, pbp, narrow);
					if (D2 && ((i2 - i1) >> 6) > (uint32_t)NW + wv) {        // wave-uniform
						uint32_t qbp;
						tile_lookup(X, tab, nzc, (i1 >> 6) + NW + wv, i2, qbp, qnv, qav);
						qpiece = fetch_piece(bbase, qbp, narrow);
					}
				}
#pragma unroll
				for (int x = 0; x < WPT; ++x) { wcnt[x] = (uint32_t)__popcll(wd[x]); mine += wcnt[x]; }
			}
			const uint32_t inc = wave_inclusive_scan_u32(mine);
			if (lane == 63) s_wtot[sflip][wv] = inc;
			{
				// the thread's WPT prefixes in one 64-bit store (entries past the cell's words are never read)
				static_assert(WPT == 4, "packed prefix store");
				const uint32_t r0 = inc - mine, r1 = r0 + wcnt[0], r2 = r1 + wcnt[1], r3 = r2 + wcnt[2];
				reinterpret_cast<uint2 *>(bpre)[tid] = make_uint2(r0 | (r1 << 16), r2 | (r3 << 16));
			}
			STAMP(4);
			lds_barrier();
			STAMP(5);
			if (PAT && MODE != MODE_COUNT && tid == 0) pat_reset(&s_pat);
			uint32_t distinct;
			{
				const uint32_t t = lane < (unsigned)NW ? s_wtot[sflip][lane] : 0u;
				const uint32_t ti = wave_inclusive_scan_u32(t);
				if (lane < (unsigned)NW) s_wbase[lane] = ti - t;
				distinct = (uint32_t)__builtin_amdgcn_readlane((int)ti, NW - 1);
			}
			sflip ^= 1u;
			if (MODE == MODE_COUNT && !ep.sk_pos) {
				// structural count: the distinct columns (scalek absent: every column is allowed); clean up and go on
				if (tid == 0) sk.segcount[seg] = distinct;
#pragma unroll
				for (int x = 0; x < WPT; ++x) { const uint32_t w = tid * WPT + x; if (w < nwords && wcnt[x]) bm[w] = 0ull; }
				lds_barrier();
				continue;
			}
			STAMP(4);
			// ---- 3. accumulate by rank (the structural count needs no values: it walks the words directly)
			if constexpr (MODE != MODE_COUNT) {
#pragma unroll
				for (int st = 0; st < MAXST; ++st) {
					if ((uint32_t)st * NW + wv >= nblk) continue;           // wave-uniform: no product in this step
					// all the lookups of a step in flight together (an empty slot reads word 2047 & ... of the bitmap: harmless)
					uint32_t pre[R], wb[R]; unsigned long long wd[R];
#pragma unroll
					for (int u = 0; u < R; ++u) {
						const uint32_t w = (krel[st][u] >> 6) & (uint32_t)(BM_WORDS - 1);
						pre[u] = bpre[w]; wb[u] = s_wbase[w / (64 * WPT)]; wd[u] = bm[w];
					}
#pragma unroll
					for (int u = 0; u < R; ++u) {
						const uint32_t rel = krel[st][u];
						if (rel != 0xFFFFFFFFu) {
							const uint32_t rank = wb[u] + pre[u] + (uint32_t)__popcll(wd[u] & ((1ull << (rel & 63u)) - 1ull));
							atomicAdd(&acc[rank], kval[st][u]);
							colof[rank] = rel;
						}
					}
				}
				STAMP(6);
				lds_barrier();                                      // acc[0 .. distinct) holds the cell's sums in column order
				STAMP(7);
			}
			// ---- 4. emit in order: thread i takes rank i (perfectly balanced -- walking the set bits of the words instead
			// leaves the barrier waiting for the thread with the fullest word: 94 vs 42 ms), cleans the accumulator entry
			// and the bitmap word of its column
			if constexpr (MODE != MODE_COUNT) {
				// the bitmap is cleaned by the threads that own its words (two 16-byte stores where any bit was set) rather
				// than word by word from the emission loop
				if (mine) { uint4 *z = reinterpret_cast<uint4 *>(&bm[tid * WPT]); z[0] = make_uint4(0, 0, 0, 0); z[1] = make_uint4(0, 0, 0, 0); }
			}
			if constexpr (MODE == MODE_DIGEST) {
				unsigned long long cnt = 0; double vs = 0;
				auto note = [&](uint32_t rel, double v) {
					const int32_t col = (int32_t)(colbase + rel);
					const bool ok = plain ? v != 0 : emit_value(ep, a_scale, col, v, &v);
					if (ok) { ++cnt; d_hash += mix64((uint32_t)rowid, (uint32_t)col); vs += v; }
				};
				if constexpr (!PAT) {
					unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
					for (uint32_t i = tid; i < distinct; i += 2 * NT) {     // two ranks per trip: their LDS reads overlap
						const uint32_t j = i + NT;
						const bool two = j < distinct;
						const uint32_t jj = two ? j : i;                        // (i again: the second exchange then reads the 0 the first left)
						const uint32_t rel0 = colof[i], rel1 = colof[jj];
						// read and clean in one LDS operation each
						const double v0 = __longlong_as_double((long long)atomicExch(&acc64[i], 0ull));
						const double v1 = __longlong_as_double((long long)atomicExch(&acc64[jj], 0ull));
						note(rel0, v0);
						if (two) note(rel1, v1);
					}
				} else {
					for (uint32_t base = 0; base < distinct; base += NT) {  // uniform trips: pat_fix_wave wants whole waves
						const uint32_t i = base + tid;
						const bool valid = i < distinct;
						const uint32_t rel = valid ? colof[i] : 0u;
						double v = valid ? acc[i] : 0.0;
						if (valid) acc[i] = 0.0;
						v = pat_fix_wave(valid && !(fabs(v) > pthr), v, (int32_t)(colbase + rel), m, tile.beg, tile.end);
						if (valid) note(rel, v);
					}
				}
				d_cnt += cnt; d_sum += vs;
				if (sk.row_nnz) {
					const unsigned long long rc = wave_reduce_sum(cnt); const double rs = wave_reduce_sum(vs);
					if (lane == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); }
				}
			} else if constexpr (MODE == MODE_COUNT) {
				// scalek present: count the allowed columns (every thread walks its own words: no values were accumulated)
				uint32_t cnt = 0;
#pragma unroll
				for (int x = 0; x < WPT; ++x) {
					if (!wcnt[x]) continue;
					const uint32_t w = tid * WPT + x;
					unsigned long long word = bm[w];
					bm[w] = 0ull;
					while (word) {
						const uint32_t bit = (uint32_t)__builtin_ctzll(word);
						word &= word - 1ull;
						if (col_allowed(ep, (int32_t)(colbase + (w << 6) + bit))) ++cnt;
					}
				}
				uint32_t tot;
				block_exclusive_scan<uint32_t, NT>(cnt, s_scan, &tot);
				if (tid == 0) sk.segcount[seg] = tot;
			} else {
				// COO: rank order IS column order.  Usually every column of the cell yields a tuple and its place is its
				// rank; only where a sum cancelled to exactly 0 (or scalek drops a column) the survivors are compacted by scans
				uint32_t nbad = 0;
				for (uint32_t base = 0; base < distinct; base += NT) {      // (uniform trips)
					const uint32_t i = base + tid;
					const bool valid = i < distinct;
					const uint32_t rel = valid ? colof[i] : 0u;
					double v = valid ? acc[i] : 1.0;
					if (PAT) {
						const bool need = valid && !(fabs(v) > pthr);
						v = pat_fix_wave(need, v, (int32_t)(colbase + rel), m, tile.beg, tile.end);
						if (need) acc[i] = v;                               // (kept: the store loop below reads it)
					}
					const bool ok = !valid || (plain ? v != 0 : emit_value(ep, a_scale, (int32_t)(colbase + rel), v, &v));
					if (!ok) ++nbad;
				}
				const int any_bad = __syncthreads_or((int)nbad);
				const int64_t o = sk.segoff[seg];
				uint32_t run = 0;
				for (uint32_t ibase = 0; ibase < distinct; ibase += NT) {
					const uint32_t i = ibase + tid;
					bool ok = false; double v = 0; uint32_t rel = 0;
					if (i < distinct) {
						rel = colof[i];
						v = acc[i];
						acc[i] = 0.0;
						ok = plain ? v != 0 : emit_value(ep, a_scale, (int32_t)(colbase + rel), v, &v);
					}
					uint32_t at = i;
					if (any_bad) {                                          // uniform
						uint32_t tot;
						const uint32_t ex = block_exclusive_scan<uint32_t, NT>(ok ? 1u : 0u, s_scan, &tot);
						at = run + ex;
						run += tot;
					}
					if (ok) { sk.out_i[o + at] = rowid; sk.out_j[o + at] = (int32_t)(colbase + rel); sk.out_v[o + at] = v; }
				}
				if (tid == 0) sk.segactual[seg] = any_bad ? run : distinct;
			}
			STAMP(8);
			lds_barrier();                                          // clean: the next cell may set bits
			STAMP(9);
		}
		for (int q = tid; q < NWORD; q += NT) X.bmask[q] = 0ull;     // (every wave is past its last lookup)
	}
#ifdef SPSAMD_STAMPS
	if (tid == 0 && sk.stamps) for (int i = 0; i < 12; ++i) sk.stamps[(size_t)blockIdx.x * 12 + i] = st_[i];
#endif
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// ---- direct tiles: ONE window per cell, dense accumulator, claim-by-exchange emission ----------------------
// A direct cell holds more than direct_min products in one window of a tile row -- too few to pay for k_dense's scan
// of all W accumulator slots.  It
//   accumulates  into the dense window accumulator, slot = column - window base, with ds_add_f64, and
//   emits        by CLAIM: every product thread exchanges its slot with 0; the one thread that gets a non-zero
//                sum back owns the output tuple.  No probing, no list of occupied slots, no scan of the window,
//                and the accumulator is clean again -- two barriers per cell.
// COO order: the claimed columns set bits in a window bitmap and a tuple's place is its rank (prefix popcount).
// (Measured on R-MAT scale-20: pays only for cells above ~1000 products -- a cell is a latency chain of two
// barriers whatever its size, and small cells leave most lanes idle -- hence the default threshold.)
template <int W, int NT, int MODE>
__global__ __launch_bounds__(NT, 4) void k_direct_tiles(const Tile *tiles, uint32_t ntile, const TCell *tcells, RowMeta m,
	const uint32_t *wptr, uint64_t nrowb, uint32_t narrow, EmitParams ep, SinkParams sk)
{
	constexpr int NW = NT / 64;
	constexpr uint32_t WSHIFT = W == 8192 ? 13 : 14;
	constexpr int R = DENSE_R;
	constexpr int NWORD = W / 64;            // item bitmap words of one tile (<= W items)
	constexpr int MAXST = 3;                 // 64-item blocks of one cell per wave
	__shared__ double acc[W + 64];
	__shared__ TileX<NT, NWORD> X;
	__shared__ unsigned long long s_cbm[MODE == MODE_STORE ? NWORD : 1];     // claimed columns of the current cell (COO order)
	__shared__ uint32_t s_cpre[MODE == MODE_STORE ? NWORD + 1 : 1];
	__shared__ uint32_t s_count;
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];

	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	for (int q = tid; q < W + 64; q += NT) acc[q] = 0.0;
	for (int q = tid; q < NWORD; q += NT) { X.bmask[q] = 0ull; if (MODE == MODE_STORE) s_cbm[q] = 0ull; }
	if (tid == 0) s_count = 0;
	const bool plain = ep.C == 1.0 && !ep.si_pos && !ep.sk_pos;
	unsigned long long d_cnt = 0, d_hash = 0; double d_sum = 0;
	uint32_t flip = 0;
	const char *bbase = reinterpret_cast<const char *>(m.btup);

	const uint32_t stride = gridDim.x;
	const uint32_t tlast = ntile - 1;
	Tile rec1 = tiles[min(blockIdx.x, tlast)];
	Tile rec2 = tiles[min(blockIdx.x + stride, tlast)];
	uint32_t nlo, nlen, nseg_, nw_; double na;
	{
		const uint32_t L = rec1.end - rec1.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		const uint32_t c = tid >> lsh, ei = tid & ((1u << lsh) - 1u);
		const bool act = c < rec1.ncells && ei < L;
		const uint32_t ec = rec1.beg + (ei < L ? ei : 0u);
		const TCell tc = tcells[rec1.first + (c < rec1.ncells ? c : 0u)];
		const uint32_t *bw = wptr + (uint64_t)tc.wa * nrowb + (uint32_t)m.acol[ec];
		const uint32_t lo = bw[0], hi = bw[1];
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u; nseg_ = tc.seg; nw_ = tc.wa;
	}
	__syncthreads();
	for (uint32_t ti = blockIdx.x; ti < ntile; ti += stride) {
		const Tile tile = rec1;
		const uint32_t lo = nlo, len = nlen, myseg = nseg_, myw = nw_; const double a = na;
		const uint32_t L = tile.end - tile.beg;
		uint32_t lsh = 0;
		while ((1u << lsh) < L) ++lsh;
		lsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)lsh);
		const int32_t rowid = tile.rowid;
		const double a_scale = row_scale(ep, rowid);
		rec1 = rec2;
		rec2 = tiles[min(ti + 2 * stride, tlast)];
		const bool has_next = ti + stride < ntile;
		const uint32_t nL = rec1.end - rec1.beg;
		uint32_t nsh = 0;
		while ((1u << nsh) < nL) ++nsh;
		const uint32_t nc = tid >> nsh, nei = tid & ((1u << nsh) - 1u);
		const bool nact = has_next && nc < rec1.ncells && nei < nL;
		const uint32_t nec = rec1.beg + (nei < nL ? nei : 0u);
		const TCell ntc = tcells[rec1.first + (nc < rec1.ncells ? nc : 0u)];
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];

		uint32_t total, nzc;
		tile_expand(X, lsh, tile.ncells, lo, len, a, myseg, myw, flip, &total, &nzc);
		{
			const uint32_t *bw = wptr + (uint64_t)ntc.wa * nrowb + (uint32_t)nk;
			const uint32_t nlo_ = bw[0], nhi_ = bw[1];
			nlo = nlo_; nlen = nact ? nhi_ - nlo_ : 0u; nseg_ = ntc.seg; nw_ = ntc.wa;
		}
		if (total == 0 || nzc == 0) {                               // uniform; cannot happen for real tiles
			if (MODE != MODE_DIGEST) for (uint32_t c = tid; c < tile.ncells; c += NT) { if (MODE == MODE_COUNT) sk.segcount[X.cellseg[c]] = 0; else sk.segactual[X.cellseg[c]] = 0; }
			continue;
		}
		TileTab<NWORD / 64> tab;
		tile_tables(X, tab);
		// first block of cell 0, prefetched like every later cell's (requested while the previous cell is claimed)
		uint32_t pbp, pnv; double pav;
		tile_lookup(X, tab, nzc, (X.cellI[0] >> 6) + wv, X.cellI[1], pbp, pnv, pav);
		BPiece ppiece = fetch_piece(bbase, pbp, narrow);

		for (uint32_t c = 0; c < tile.ncells; ++c) {
			const uint32_t i0 = X.cellI[c], i1 = X.cellI[c + 1];      // multiples of 64; i1 = next cell's (aligned) start
			const uint32_t wbase = X.cellw[c] << WSHIFT;
			const uint32_t seg = X.cellseg[c];
			const uint32_t nblk = (i1 - i0) >> 6;
			if (nblk > (uint32_t)(MAXST * NW) && tid == 0) atomicOr(sk.err, 2u);     // never: k_cells bounds a direct cell's items
			// ---- accumulate: block b0 + st * NW + wv per wave and step; the slots are kept for the claim
			uint32_t ks[MAXST][R];
			double kv[MODE == MODE_STORE ? MAXST : 1][MODE == MODE_STORE ? R : 1];
#pragma unroll
			for (int st = 0; st < MAXST; ++st) {
#pragma unroll
				for (int u = 0; u < R; ++u) ks[st][u] = (uint32_t)W + lane;
				const uint32_t bl = (uint32_t)st * NW + wv;
				if (bl < nblk) {                                            // wave-uniform
					uint32_t nv; double av; BPiece piece;
					if (st == 0) { nv = pnv; av = pav; piece = ppiece; }
					else {
						uint32_t bp;
						tile_lookup(X, tab, nzc, (i0 >> 6) + bl, i1, bp, nv, av);
						piece = fetch_piece(bbase, bp, narrow);
					}
#pragma unroll
					for (int u = 0; u < R; ++u) {
						const uint32_t slot = (uint32_t)u < nv ? piece.w[3 * u] - wbase : (uint32_t)W + lane;
						ks[st][u] = slot;
						if (MODE == MODE_COUNT) acc[slot] = 1.0;
						else atomicAdd(&acc[slot], av * __hiloint2double((int)piece.w[3 * u + 2], (int)piece.w[3 * u + 1]));
					}
				}
			}
			lds_barrier();                                          // every product of the cell is in the accumulator
			if (c + 1 < tile.ncells) {                              // uniform: the next cell's first block, in flight during the claim
				tile_lookup(X, tab, nzc, (i1 >> 6) + wv, X.cellI[c + 2], pbp, pnv, pav);
				ppiece = fetch_piece(bbase, pbp, narrow);
			}
			// ---- claim: exchange the slot with 0; a non-zero answer makes this thread the tuple's owner.  The R exchanges
			// of a step are in flight together (a dump slot is exchanged like any other: its answer is not looked at)
			unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
			uint32_t mycount = 0; double mysum = 0.0;
#pragma unroll
			for (int st = 0; st < MAXST; ++st) {
				if ((uint32_t)st * NW + wv < nblk) {                        // wave-uniform
					unsigned long long olds[R];
#pragma unroll
					for (int u = 0; u < R; ++u) olds[u] = atomicExch(&acc64[ks[st][u]], 0ull);
#pragma unroll
					for (int u = 0; u < R; ++u) {
						const uint32_t slot = ks[st][u];
						bool own = false;
						double v = 0.0;
						if (slot < (uint32_t)W) {
							const unsigned long long old = olds[u];
							v = __longlong_as_double((long long)old);
							const int32_t col = (int32_t)(wbase + slot);
							if (MODE == MODE_COUNT) own = (v != 0) && col_allowed(ep, col);
							else if (plain) own = v != 0;
							else own = emit_value(ep, a_scale, col, v, &v);
						}
						if constexpr (MODE == MODE_DIGEST) {
							if (own) { ++mycount; d_hash += mix64((uint32_t)rowid, wbase + slot); mysum += v; }
						} else if constexpr (MODE == MODE_COUNT) {
							if (own) ++mycount;
						} else {
							kv[st][u] = v;
							if (own) { ++mycount; atomicOr(&s_cbm[slot >> 6], 1ull << (slot & 63u)); }
							else ks[st][u] = 0xFFFFFFFFu;
						}
					}
				}
			}
			if constexpr (MODE == MODE_DIGEST) {
				d_cnt += mycount; d_sum += mysum;
				if (sk.row_nnz) {
					const unsigned long long rc = wave_reduce_sum((unsigned long long)mycount); const double rs = wave_reduce_sum(mysum);
					if (lane == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); }
				}
				lds_barrier();                                      // claims done before the next cell accumulates
			} else if constexpr (MODE == MODE_COUNT) {
				const uint32_t wc = (uint32_t)wave_reduce_sum((unsigned long long)mycount);
				if (lane == 0 && wc) atomicAdd(&s_count, wc);
				lds_barrier();
				if (tid == 0) { sk.segcount[seg] = s_count; s_count = 0; }
				lds_barrier();
			} else {
				lds_barrier();                                      // claimed-column bitmap complete
				if (wv == 0) {
					uint32_t run = 0;
#pragma unroll
					for (int x = 0; x < NWORD / 64; ++x) {
						const uint32_t cnt = (uint32_t)__popcll(s_cbm[x * 64 + lane]);
						const uint32_t inc2 = wave_inclusive_scan_u32(cnt);
						s_cpre[x * 64 + lane] = run + inc2 - cnt;
						run += (uint32_t)__builtin_amdgcn_readlane((int)inc2, 63);
					}
					if (lane == 0) s_cpre[NWORD] = run;
				}
				lds_barrier();
				const int64_t o = sk.segoff[seg];
#pragma unroll
				for (int st = 0; st < MAXST; ++st) {
#pragma unroll
					for (int u = 0; u < R; ++u) {
						const uint32_t slot = ks[st][u];
						if (slot < (uint32_t)W) {
							const uint32_t wd = slot >> 6;
							const uint32_t rank = s_cpre[wd] + (uint32_t)__popcll(s_cbm[wd] & ((1ull << (slot & 63u)) - 1ull));
							sk.out_i[o + rank] = rowid;
							sk.out_j[o + rank] = (int32_t)(wbase + slot);
							sk.out_v[o + rank] = kv[st][u];
						}
					}
				}
				if (tid == 0) sk.segactual[seg] = s_cpre[NWORD];
				lds_barrier();
				for (int q = tid; q < NWORD; q += NT) s_cbm[q] = 0ull;
				lds_barrier();
			}
		}
		for (int q = tid; q < NWORD; q += NT) X.bmask[q] = 0ull;     // (every wave is past its last lookup: the cell loop ends with a barrier)
	}
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// ====================================================================== holes (cancellation in STORE)

__global__ void k_seg_holes(const uint32_t *segcount, const uint32_t *segactual, uint32_t nseg, unsigned long long *holes, const uint32_t *err)
{
	uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s == 0) holes[1] = *err;                                // rides along with the read-back of the hole count
	unsigned long long d = 0;
	if (s < nseg) d = (unsigned long long)(segcount[s] - segactual[s]);
	d = wave_reduce_sum(d);
	if (lane_id() == 0 && d) atomicAdd(holes, d);
}

__global__ void k_seg_gather(const int64_t *oldoff, const int64_t *newoff, const uint32_t *segactual, uint32_t nseg,
	const int32_t *si, const int32_t *sj, const double *sv, int32_t *di, int32_t *dj, double *dv)
{
	uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	int64_t a = oldoff[s], b = newoff[s];
	uint32_t n = segactual[s];
	for (uint32_t t = 0; t < n; ++t) { di[b + t] = si[a + t]; dj[b + t] = sj[a + t]; dv[b + t] = sv[a + t]; }
}

__global__ void k_digest_reduce(const DigestSlot *slots, DigestSlot *out, const uint32_t *err)
{
	// one wave; deterministic order of the slot sums
	unsigned long long c = 0, h = 0; double s = 0;
	for (int q = threadIdx.x; q < DIGEST_SLOTS; q += 64) { c += slots[q].count; h += slots[q].hash; s += slots[q].sum; }
	c = wave_reduce_sum(c); h = wave_reduce_sum(h); s = wave_reduce_sum(s);
	if (threadIdx.x == 0) { out->count = c; out->hash = h; out->sum = s; out->pad = *err; }
}

// ====================================================================== host driver

static unsigned grid_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

struct Bins {
	uint32_t count[NBIN];
	uint32_t off[NBIN + 1];
	uint32_t *rows;
};

template <int MODE>
static void launch_light(spsamd_ctx *c, const Bins &b, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	hipStream_t st = c->stream;
	const unsigned cap = (unsigned)c->num_cu * 8u * 4u;           // 8 resident workgroups per CU, 4 rounds of them
	auto grid_for = [cap](size_t n, unsigned per) { return std::min<unsigned>((unsigned)((n + per - 1) / per), cap); };
	if (b.count[1]) { k_light<8, MODE><<<dim3(grid_for(b.count[1], 32)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[1] : nullptr, b.count[1], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[2]) { k_light<16, MODE><<<dim3(grid_for(b.count[2], 16)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[2] : nullptr, b.count[2], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[3]) { k_light<32, MODE><<<dim3(grid_for(b.count[3], 8)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[3] : nullptr, b.count[3], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[4]) { k_light<64, MODE><<<dim3(grid_for(b.count[4], 4)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[4] : nullptr, b.count[4], m, ep, sk); SPS_LAUNCH_CHECK(); }
}

template <int T, int NT, int MODE, bool WINDOWED>
static void launch_hash(spsamd_ctx *c, const Cell *cells, uint32_t ncell, const uint32_t *xb, const RowMeta &m, const uint32_t *bwin,
	uint32_t nwin1, const EmitParams &ep, const SinkParams &sk)
{
	if (!ncell) return;
	static int per_cu = 0;                     // resident workgroups per CU of this instantiation
	if (!per_cu) {
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hash<T, NT, MODE, WINDOWED, false>, NT, 0) != hipSuccess || nb < 1) nb = 1;
		per_cu = nb;
	}
	unsigned grid = std::min<unsigned>(ncell, (unsigned)(c->num_cu * per_cu));
	if (grid >= 64) grid &= ~7u;               // multiple of 8: the XCD-aware walk
	if (ep.pattern) k_hash<T, NT, MODE, WINDOWED, true><<<dim3(grid), dim3(NT), 0, c->stream>>>(cells, ncell, xb, m, bwin, nwin1, ep, sk);
	else k_hash<T, NT, MODE, WINDOWED, false><<<dim3(grid), dim3(NT), 0, c->stream>>>(cells, ncell, xb, m, bwin, nwin1, ep, sk);
	SPS_LAUNCH_CHECK();
}

struct MidCells { Cell *cells[3] = {nullptr, nullptr, nullptr}; };

template <int MODE>
static void launch_mid(spsamd_ctx *c, const Bins &b, const MidCells &mc, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	launch_hash<1024, 256, MODE, false>(c, mc.cells[0], b.count[5], nullptr, m, nullptr, 0, ep, sk);
	launch_hash<4096, 512, MODE, false>(c, mc.cells[1], b.count[6], nullptr, m, nullptr, 0, ep, sk);
	launch_hash<8192, 512, MODE, false>(c, mc.cells[2], b.count[7], nullptr, m, nullptr, 0, ep, sk);   // 115 KB of LDS: one workgroup per CU, so make it 8 waves
}

// Thrown by heavy_prepare: op(B) has more column windows than the heavy-row path indexes, or its window indices would not
// fit the device -- spgemm() then multiplies by column blocks of `width` columns (spgemm_column_blocks).
struct TooWide { uint64_t width; };
constexpr uint64_t COLBLK = (uint64_t)2048 << 14;                   // 2048 windows of 16384 columns: 2^25

struct Heavy {
	uint32_t n = 0;                  // heavy rows
	uint32_t *rows = nullptr;
	uint32_t *bwin = nullptr;
	uint32_t nwin = 0, nwin1 = 0;
	uint32_t *winprod = nullptr;
	uint32_t ncell[NCLS] = {};
	Cell *cells[NCLS] = {};
	CellBases cnt{}, base{};
	uint32_t *xb[NCLS] = {};         // XCD part boundaries per class
	int W = 8192;
	uint32_t cell_cap = CELL_CAP_DEFAULT, dense_min = DENSE_MIN_DEFAULT;
	TileBases tb{};                  // hash tiles
	uint32_t ntile = 0, ntcell = 0;
	TileBases tb2{};                 // direct tiles (k_direct_tiles)
	uint32_t ntile2 = 0, ntcell2 = 0;
	uint32_t direct_min = 0;
	int tiles2 = 0;
	uint32_t span_cap = 0;
	uint32_t alt_cap = 0, alt_span = 0;
	unsigned long long *alt_cells = nullptr;
	uint32_t long_cap = 0, long_dense_min = 0;
	bool coo = false;                // the tiles also serve a STORE launch
	unsigned long long clsprod[NCLS + 2] = {};
	uint32_t *wptr = nullptr;        // window-major copy of B (dense cells): row pointer per window ...
	BTup *btw = nullptr;             // ... and tuples
	uint64_t nrowb = 0;
	uint32_t nnzb = 0;
};

static TileKinds tile_kinds(const Heavy &hv)
{
	TileKinds tk{{hv.tb, hv.tb2}, hv.direct_min, hv.span_cap, hv.long_cap, hv.long_dense_min,
		hv.tiles2 == 0 ? (uint32_t)BM_MAXOUT : (uint32_t)(TILE_T / 2), hv.alt_cap, hv.alt_span, hv.alt_cells};
	return tk;
}

template <int MODE>
static void launch_heavy_hash(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	SPS_HIP(hipEventRecord(c->ev2[0], c->stream));
	if (hv.ntile && hv.tiles2 == 0) {
		const uint32_t narrow = ((uint64_t)hv.nnzb + DENSE_R) * 12u < (uint64_t(1) << 32) ? 1u : 0u;
		const unsigned grid = std::min<unsigned>(hv.ntile, (unsigned)c->num_cu * 2u);
#ifdef SPSAMD_STAMPS
		SinkParams sk2 = sk;
		sk2.stamps = c->arena.get<unsigned long long>((size_t)grid * 12);
		fill_zero(c, sk2.stamps, (size_t)grid * 12 * sizeof(unsigned long long));
		k_bm_tiles<MODE, false><<<dim3(grid), dim3(BM_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk2);
		{
			std::vector<unsigned long long> h((size_t)grid * 12);
			SPS_HIP(hipMemcpyAsync(h.data(), sk2.stamps, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipStreamSynchronize(c->stream));
			double sum[12] = {};
			for (unsigned g = 0; g < grid; ++g) for (int i = 0; i < 12; ++i) sum[i] += (double)h[(size_t)g * 12 + i];
			static const char *nm[12] = {"pre", "expand", "bits", "B", "scan", "B", "rank-add", "B", "emit", "B", "tiles", "cells"};
			fprintf(stderr, "k_bm_tiles stamps (mean cycles per workgroup, grid %u):", grid);
			for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.4g", nm[i], sum[i] / grid);
			fprintf(stderr, "\n");
		}
#else
		if (ep.pattern) k_bm_tiles<MODE, true><<<dim3(grid), dim3(BM_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk);
		else k_bm_tiles<MODE, false><<<dim3(grid), dim3(BM_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk);
#endif
		SPS_LAUNCH_CHECK();
	} else if (hv.ntile && hv.tiles2 == 2) {
		const uint32_t narrow = ((uint64_t)hv.nnzb + DENSE_R) * 12u < (uint64_t(1) << 32) ? 1u : 0u;
		static int per_cu2 = 0;
		if (!per_cu2) {
			int nb = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hash_tiles2<MODE, false>, TILE2_NT, 0) != hipSuccess || nb < 1) nb = 1;
			per_cu2 = nb;
		}
		const unsigned grid = std::min<unsigned>(hv.ntile, (unsigned)(c->num_cu * per_cu2));
#ifdef SPSAMD_STAMPS
		SinkParams sk2 = sk;
		sk2.stamps = c->arena.get<unsigned long long>((size_t)grid * 12);
		fill_zero(c, sk2.stamps, (size_t)grid * 12 * sizeof(unsigned long long));
		k_hash_tiles2<MODE, false><<<dim3(grid), dim3(TILE2_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk2);
		{
			std::vector<unsigned long long> h((size_t)grid * 12);
			SPS_HIP(hipMemcpyAsync(h.data(), sk2.stamps, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipStreamSynchronize(c->stream));
			double sum[12] = {};
			for (unsigned g = 0; g < grid; ++g) for (int i = 0; i < 12; ++i) sum[i] += (double)h[(size_t)g * 12 + i];
			static const char *nm[12] = {"pre", "expand", "tables+pf", "insert", "Bwait", "pf-next", "emit", "Bwait2", "tiles", "cells", "-", "-"};
			fprintf(stderr, "k_hash_tiles2 stamps (mean cycles per workgroup, grid %u):", grid);
			for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.4g", nm[i], sum[i] / grid);
			fprintf(stderr, "\n");
		}
#else
		if (ep.pattern) k_hash_tiles2<MODE, true><<<dim3(grid), dim3(TILE2_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk);
		else k_hash_tiles2<MODE, false><<<dim3(grid), dim3(TILE2_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, narrow, ep, sk);
#endif
		SPS_LAUNCH_CHECK();
	} else
	if (hv.ntile) {
		static int per_cu = 0;
		if (!per_cu) {
			int nb = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hash_tiles<MODE>, TILE_NT, 0) != hipSuccess || nb < 1) nb = 1;
			per_cu = nb;
		}
		unsigned grid = std::min<unsigned>(hv.ntile, (unsigned)(c->num_cu * per_cu));
#ifdef SPSAMD_STAMPS
		SinkParams sk2 = sk;
		sk2.stamps = c->arena.get<unsigned long long>((size_t)grid * 12);
		fill_zero(c, sk2.stamps, (size_t)grid * 12 * sizeof(unsigned long long));
		k_hash_tiles<MODE><<<dim3(grid), dim3(TILE_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, ep, sk2);
		{
			std::vector<unsigned long long> h((size_t)grid * 12);
			SPS_HIP(hipMemcpyAsync(h.data(), sk2.stamps, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipStreamSynchronize(c->stream));
			double sum[12] = {};
			for (unsigned g = 0; g < grid; ++g) for (int i = 0; i < 12; ++i) sum[i] += (double)h[(size_t)g * 12 + i];
			static const char *nm[12] = {"pre", "Bwait", "expand_load", "expand_batch", "products", "Bwait2", "segbcast", "emit", "tiles", "cells", "-", "-"};
			fprintf(stderr, "k_hash_tiles stamps (mean cycles per workgroup, grid %u):", grid);
			for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.4g", nm[i], sum[i] / grid);
			fprintf(stderr, "\n");
		}
#else
		k_hash_tiles<MODE><<<dim3(grid), dim3(TILE_NT), 0, c->stream>>>(hv.tb.tiles, hv.ntile, hv.tb.tcells, m, hv.bwin, hv.nwin1, ep, sk);
#endif
		SPS_LAUNCH_CHECK();
	}
	SPS_HIP(hipEventRecord(c->ev2[1], c->stream));
	if (hv.ntile2) {
		RowMeta m2 = m;
		m2.btup = hv.btw;
		const uint32_t narrow = ((uint64_t)hv.nnzb + DENSE_R) * 12u < (uint64_t(1) << 32) ? 1u : 0u;
		if (hv.W == 8192) {
			const unsigned grid = std::min<unsigned>(hv.ntile2, (unsigned)c->num_cu * 2u);
			k_direct_tiles<8192, 512, MODE><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.tb2.tiles, hv.ntile2, hv.tb2.tcells, m2, hv.wptr, hv.nrowb, narrow, ep, sk);
		} else {
			const unsigned grid = std::min<unsigned>(hv.ntile2, (unsigned)c->num_cu);
			k_direct_tiles<16384, 1024, MODE><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.tb2.tiles, hv.ntile2, hv.tb2.tcells, m2, hv.wptr, hv.nrowb, narrow, ep, sk);
		}
		SPS_LAUNCH_CHECK();
	}
	SPS_HIP(hipEventRecord(c->ev2[2], c->stream));
	launch_hash<1024, 256, MODE, true>(c, hv.cells[0], hv.ncell[0], hv.xb[0], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<3072, 512, MODE, true>(c, hv.cells[1], hv.ncell[1], hv.xb[1], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<4096, 512, MODE, true>(c, hv.cells[2], hv.ncell[2], hv.xb[2], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<8192, 512, MODE, true>(c, hv.cells[3], hv.ncell[3], hv.xb[3], m, hv.bwin, hv.nwin1, ep, sk);
}

template <int MODE>
static void launch_heavy_dense(spsamd_ctx *c, const Heavy &hv, const RowMeta &m0, const EmitParams &ep, const SinkParams &sk)
{
	if (!hv.ncell[CLS_DENSE]) return;
	RowMeta m = m0;
	const uint32_t *widx = hv.bwin;
	uint64_t kstride = hv.nwin1, wstride = 1;
	if (hv.wptr) { widx = hv.wptr; kstride = 1; wstride = hv.nrowb; m.btup = hv.btw; }
	const uint32_t narrow = ((uint64_t)hv.nnzb + DENSE_R) * 12u < (uint64_t(1) << 32) ? 1u : 0u;    // 32-bit byte offsets into B suffice
	if (hv.W == 8192) {
		unsigned grid = std::min<unsigned>(hv.ncell[CLS_DENSE], (unsigned)c->num_cu * 2u);
		if (grid >= 64) grid &= ~7u;
#ifdef SPSAMD_STAMPS
		SinkParams sk2 = sk;
		sk2.stamps = c->arena.get<unsigned long long>((size_t)grid * 12);
		fill_zero(c, sk2.stamps, (size_t)grid * 12 * sizeof(unsigned long long));
		k_dense<8192, 512, MODE, false><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk2);
		{
			std::vector<unsigned long long> h((size_t)grid * 12);
			SPS_HIP(hipMemcpyAsync(h.data(), sk2.stamps, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipStreamSynchronize(c->stream));
			double sum[12] = {}; double mx = 0;
			for (unsigned g = 0; g < grid; ++g) { double t = 0; for (int i = 0; i < 12; ++i) { sum[i] += (double)h[(size_t)g * 12 + i]; if (i < 8) t += (double)h[(size_t)g * 12 + i]; } mx = std::max(mx, t); }
			static const char *nm[12] = {"pre-B1", "B1wait", "compact", "B2wait", "tables", "steps", "B3wait", "scanout", "cells", "chunks", "steps#", "-"};
			fprintf(stderr, "k_dense stamps (mean cycles per workgroup; max total %.3g):", mx);
			for (int i = 0; i < 11; ++i) fprintf(stderr, " %s %.4g", nm[i], sum[i] / grid);
			fprintf(stderr, "\n");
		}
		return;
#endif
		if (ep.pattern) k_dense<8192, 512, MODE, true><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk);
		else k_dense<8192, 512, MODE, false><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk);
	} else {
		unsigned grid = std::min<unsigned>(hv.ncell[CLS_DENSE], (unsigned)c->num_cu);
		if (grid >= 64) grid &= ~7u;
		if (ep.pattern) k_dense<16384, 1024, MODE, true><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk);
		else k_dense<16384, 1024, MODE, false><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk);
	}
	SPS_LAUNCH_CHECK();
}

// Window-major copy of B (see k_wm_counts): built once the cell grouping has shown that dense cells exist.
static void heavy_window_major(spsamd_ctx *c, Heavy &hv, const ConMat &B, uint32_t wshift)
{
	hipStream_t st = c->stream;
	const uint64_t nrowb = hv.nrowb, total = nrowb * hv.nwin;
	uint16_t *cnt = c->arena.get<uint16_t>(total);
	hv.wptr = c->arena.get<uint32_t>(total + 1);
	k_wm_counts<<<dim3((unsigned)((nrowb + 63) / 64), (hv.nwin + 63) / 64), dim3(256), 0, st>>>(hv.bwin, (uint32_t)nrowb, hv.nwin, hv.nwin1, cnt);
	SPS_LAUNCH_CHECK();
	scan_exclusive_u16_u32(c, cnt, hv.wptr, total);
	hv.btw = c->arena.get<BTup>((size_t)B.nnz + DENSE_R);
	k_wm_scatter<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.row, B.col, B.val, B.nnz, wshift, hv.bwin, hv.nwin1, hv.wptr, nrowb, hv.btw);
	SPS_LAUNCH_CHECK();
}

static float elapsed(hipEvent_t a, hipEvent_t b)
{
	float ms = 0;
	SPS_HIP(hipEventElapsedTime(&ms, a, b));
	return ms;
}

// Heavy rows: window index of B, per-row window histogram, counting pass of the cell grouping.
static void heavy_prepare(spsamd_ctx *c, Heavy &hv, const Bins &bins, const RowMeta &m, const ConMat &B, const uint32_t *bptr,
	uint32_t extra, uint32_t *nseg, bool ordered, bool pattern)
{
	hipStream_t st = c->stream;
	hv.W = B.ncol > (uint64_t(1) << 21) ? 16384 : 8192;
	if (c->tune.window == 8192 || c->tune.window == 16384) hv.W = c->tune.window;
	const uint32_t wshift = hv.W == 8192 ? 13 : 14;
	hv.nwin = (uint32_t)((B.ncol + hv.W - 1) >> wshift);
	if (hv.nwin > (uint32_t)WH_MAXW) throw TooWide{COLBLK};         // spgemm() then multiplies by column blocks of B
	hv.nwin1 = hv.nwin + 1;
	const uint64_t nrowb = B.nrow + extra;
	{
		// The window indices (bwin, its 16-bit counts, the window-major pointer with its counts, the heavy rows' histograms)
		// grow with rows(B) x windows: 12 bytes per B row and window.  Where they would not fit what the device has left
		// (or the cap a test sets), the product goes by column blocks narrow enough for them to fit.
		const uint64_t per_window = nrowb * 12u + (uint64_t)hv.n * 4u;
		uint64_t budget;
		const uint64_t room = c->arena.slabs.empty() ? 0 : c->arena.slabs.back().cap - c->arena.slabs.back().used;
		if (c->tune.index_budget_mb > 0) budget = (uint64_t)c->tune.index_budget_mb << 20;
		else if (per_window * (hv.nwin + 1ull) <= room) budget = room;  // (the steady state: the workspace of an earlier call holds them)
		else {
			size_t freeb = 0, totalb = 0;
			SPS_HIP(hipMemGetInfo(&freeb, &totalb));
			budget = (uint64_t)((double)(freeb + room) * 0.8);
		}
		if (per_window * (hv.nwin + 1ull) > budget && hv.nwin > 1) {
			uint64_t fit = budget / per_window;                         // windows per block that fit
			if (fit < 2) throw Error{SPSAMD_ENOMEM, "the window index of one column window of op(B) does not fit the device"};
			uint64_t w2 = 1;
			while (w2 * 2 <= fit - 1) w2 *= 2;
			throw TooWide{w2 << wshift};
		}
	}
	hv.nrowb = nrowb;
	hv.nnzb = B.nnz;
	hv.bwin = c->arena.get<uint32_t>(nrowb * hv.nwin1);
	k_bwin_prefill<<<dim3(4096), dim3(256), 0, st>>>(bptr, nrowb, hv.nwin1, hv.bwin);
	SPS_LAUNCH_CHECK();
	k_bwin_fill<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.row, B.col, bptr, B.nnz, wshift, hv.nwin1, hv.bwin);
	SPS_LAUNCH_CHECK();
	hv.rows = bins.rows + bins.off[8];
	hv.winprod = c->arena.get<uint32_t>((uint64_t)hv.n * hv.nwin);
	const uint32_t nwp = (hv.nwin + 1u) & ~1u;
	uint16_t *wcnt = c->arena.get<uint16_t>(nrowb * nwp);
	k_bwin_counts<<<dim3(4096), dim3(256), 0, st>>>(hv.bwin, nrowb, hv.nwin, nwp, wcnt);
	SPS_LAUNCH_CHECK();
	uint32_t *hubcount = c->arena.get<uint32_t>(1);
	uint32_t *hublist = c->arena.get<uint32_t>(WH_HUB_MAX);
	fill_zero(c, hubcount, sizeof(uint32_t));
	k_win_hist<<<dim3(hv.n), dim3(WH_NT), 0, st>>>(hv.rows, hv.n, m, wcnt, hv.nwin, nwp, hv.winprod, hubcount, hublist);
	SPS_LAUNCH_CHECK();
	k_win_hist_hub<<<dim3((unsigned)c->num_cu * 4u), dim3(WH_NT), 0, st>>>(hv.rows, m, wcnt, hv.nwin, nwp, hv.winprod, hubcount, hublist);
	SPS_LAUNCH_CHECK();
	for (int k = 0; k < NCLS; ++k) {
		hv.cnt.base[k] = c->arena.get<uint32_t>(hv.n);
		hv.base.base[k] = c->arena.get<uint32_t>((size_t)hv.n + 1);
	}
	if (c->tune.cell_cap >= 64 && c->tune.cell_cap <= (int)CELL_CAP) hv.cell_cap = (uint32_t)c->tune.cell_cap;
	if (c->tune.dense_min >= 64 && c->tune.dense_min <= (int)CELL_CAP) hv.dense_min = (uint32_t)c->tune.dense_min;
	// (a window between dense_min and cell_cap products becomes a dense cell; smaller ones are grouped up to cell_cap)
	unsigned long long *clsprod = c->arena.get<unsigned long long>(NCLS + 2);
	hv.tb.enabled = !c->tune.no_tiles;
	// Tile kernel of the hash-class cells: 0 bitmap rank (k_bm_tiles) | 1 first generation | 2 hash tiles v2.
	// ORDERED runs on the first generation (the variant that exists), EXACT_PATTERN on the bitmap tiles or the hash tiles v2;
	// otherwise the choice is made per call below, by counting the cells either scheme would cut.
	const bool free_choice = !ordered && c->tune.tiles_v1 == 0;
	const bool user_dense_min = c->tune.dense_min >= 64 && c->tune.dense_min <= (int)CELL_CAP;
	auto set_scheme = [&](int scheme) {
		hv.tiles2 = scheme;
		if (!user_dense_min) hv.dense_min = scheme == 0 && hv.tb.enabled ? DENSE_MIN_BITMAP : DENSE_MIN_DEFAULT;
		hv.tb.by_items = scheme != 1 ? 1 : 0;
		hv.tb.pb = scheme != 1 ? (uint32_t)TILE2_ITEMS : (hv.coo ? (uint32_t)TILE_PB_STORE : (uint32_t)TILE_PB);
		hv.span_cap = scheme == 0 ? (uint32_t)BM_WORDS >> (wshift - 6) : 0u;
	};
	set_scheme(ordered ? 1 : (c->tune.tiles_v1 == 1 ? (pattern ? 2 : 1) : (c->tune.tiles_v1 == 2 ? 2 : 0)));
	// Rows too long for a tile: a hash-class cell of theirs reads B in row-major pieces of 4.4 tuples on average -- 6x the
	// algorithmic bytes from HBM at line granularity (FETCH_SIZE of k_hash<3072>: 14.8 GB for 2.5 GB) -- while the dense kernel
	// reads the window-major copy.  Their windows go to k_dense from LONG_DENSE_MIN products on.  R-MAT A*A, ms per step:
	//   threshold   scale 19   scale 20 (cfg2)   scale 21
	//     2048        26.9        78.1             251 (1536: 247.5)
	//     1024         -          76.9             250.6
	//      512        27.0        76.8             260.9
	//      128        27.2        76.5             272.3
	// (at scale 21 the same rows spread over twice the windows: a dense cell's walk over the row's A tuples and its scan of
	// all W slots buy half the products).  1024 keeps most of scale 20's gain and costs scale 21 about 1 %.
	// (8192-column windows only: the 16384-column dense kernel runs one workgroup per CU and wants full windows -- scale 23:
	// 2.49 s with 128, 2.40 with 512, 2.26 with the general threshold)
	hv.long_dense_min = c->tune.long_dense_min > 0 ? (uint32_t)c->tune.long_dense_min : (hv.W == 8192 ? LONG_DENSE_MIN_DEFAULT : 0u);
	hv.long_cap = c->tune.long_cap > 0 ? (uint32_t)std::min<int>(c->tune.long_cap, (int)CELL_CAP) : 0u;
	hv.tb2.by_items = 1;
	// direct cells need the window-major copy of B and are not used for ordered (ascending-k) sums
	hv.direct_min = c->tune.direct_min > 0 ? (uint32_t)c->tune.direct_min : DIRECT_MIN_DEFAULT;
	hv.tb2.enabled = hv.tb.enabled && !c->tune.no_wmajor && !ordered && !pattern && hv.direct_min < hv.dense_min;
	hv.tb2.pb = (uint32_t)hv.W;      // items of a direct tile: one bit each in a W-bit bitmap
	for (TileBases *t : {&hv.tb, &hv.tb2}) {
		t->ntc = c->arena.get<uint32_t>(hv.n); t->ntl = c->arena.get<uint32_t>(hv.n);
		t->tcbase = c->arena.get<uint32_t>((size_t)hv.n + 1); t->tlbase = c->arena.get<uint32_t>((size_t)hv.n + 1);
	}
	auto count_pass = [&]() {
		fill_zero(c, clsprod, (NCLS + 2) * sizeof(unsigned long long));
		for (TileBases *t : {&hv.tb, &hv.tb2}) { fill_zero(c, t->ntc, hv.n * sizeof(uint32_t)); fill_zero(c, t->ntl, hv.n * sizeof(uint32_t)); }
		k_cells<false><<<dim3(grid_for(hv.n, 128)), dim3(128), 0, st>>>(hv.rows, hv.n, m.beg, m.id, hv.winprod, hv.nwin, hv.cell_cap, hv.dense_min, hv.cnt, nseg, hv.base, CellLists{}, nullptr, clsprod, tile_kinds(hv));
		SPS_LAUNCH_CHECK();
	};
	if (free_choice && hv.tb.enabled) {
		// Bitmap tiles hold 4096 products per cell but at most 16 windows of columns; hash tiles 2048 products over any
		// range.  Per-cell bookkeeping is most of a tile kernel's time, so the scheme that cuts clearly fewer cells wins;
		// break-even measured near 1.4 bitmap cells per hash cell (R-MAT A*A: scale 20, 3.22 M against 3.85 M cells: bitmap,
		// 34.8 vs 39+ ms; scale 21, 11.4 M against 12.2 M: bitmap, 236 vs 247 ms; scale 22, 42.6 M against 29.8 M: 727 vs
		// 731 ms; scale 23 -- sparse rows spread over 512 windows, 150 M against 65 M -- hash, 0.95 vs 1.35 s).  The counting
		// pass of the bitmap scheme also counts the cells the hash scheme would cut; only where that one wins is it repeated.
		set_scheme(0);
		hv.alt_cap = (uint32_t)(TILE_T / 2); hv.alt_span = 0;
		hv.alt_cells = c->arena.get<unsigned long long>(2);          // [0] cells of the hash scheme, [1] of the bitmap scheme
		fill_zero(c, hv.alt_cells, 2 * sizeof(unsigned long long));
		count_pass();
		WordList wl; wl.add64(hv.alt_cells); wl.add64(hv.alt_cells + 1);
		uint32_t hw[4];
		read_back_words(c, wl, hw);
		const unsigned long long cells_hash = (unsigned long long)hw[0] | ((unsigned long long)hw[1] << 32);
		const unsigned long long cells_bm = (unsigned long long)hw[2] | ((unsigned long long)hw[3] << 32);
		hv.alt_cells = nullptr;
		if (getenv("SPSAMD_TRACE")) fprintf(stderr, "tile cells: bitmap %llu hash %llu\n", cells_bm, cells_hash);
		if (cells_bm * 100u > cells_hash * 140u) { set_scheme(2); count_pass(); }
	} else count_pass();
	// every per-row counter of the grouping scanned in one batch, every total read back in one round trip
	ScanBatch sb;
	for (int k = 0; k < NCLS; ++k) sb.add(hv.cnt.base[k], hv.base.base[k]);
	sb.add(hv.tb.ntc, hv.tb.tcbase); sb.add(hv.tb.ntl, hv.tb.tlbase);
	sb.add(hv.tb2.ntc, hv.tb2.tcbase); sb.add(hv.tb2.ntl, hv.tb2.tlbase);
	static_assert(NCLS + 4 <= SCAN_BATCH_MAX && NCLS + 4 + 2 * (NCLS + 2) <= WORD_LIST_MAX, "batch sizes");
	scan_exclusive_u32_batch(c, sb, hv.n);
	WordList wl;
	for (int k = 0; k < NCLS; ++k) wl.add(hv.base.base[k] + hv.n);
	wl.add(hv.tb.tcbase + hv.n); wl.add(hv.tb.tlbase + hv.n); wl.add(hv.tb2.tcbase + hv.n); wl.add(hv.tb2.tlbase + hv.n);
	for (int k = 0; k < NCLS + 2; ++k) wl.add64(clsprod + k);
	uint32_t hw[WORD_LIST_MAX];
	read_back_words(c, wl, hw);
	for (int k = 0; k < NCLS; ++k) hv.ncell[k] = hw[k];
	hv.ntcell = hw[NCLS]; hv.ntile = hw[NCLS + 1]; hv.ntcell2 = hw[NCLS + 2]; hv.ntile2 = hw[NCLS + 3];
	for (int k = 0; k < NCLS + 2; ++k) hv.clsprod[k] = (unsigned long long)hw[NCLS + 4 + 2 * k] | ((unsigned long long)hw[NCLS + 5 + 2 * k] << 32);
	if ((hv.ncell[CLS_DENSE] || hv.ntile2) && !c->tune.no_wmajor) heavy_window_major(c, hv, B, wshift);
}

// Emit the cells (needs segbase for the COO sink) and order the dense ones by descending products.
static void heavy_cells(spsamd_ctx *c, Heavy &hv, const RowMeta &m, const uint32_t *segbase)
{
	hipStream_t st = c->stream;
	int wbits = 1;                                   // bits of a window index: the cell lists are sorted on as few digits as needed
	while ((1u << wbits) < hv.nwin) ++wbits;
	CellLists lists;
	for (int k = 0; k < NCLS; ++k) { hv.cells[k] = c->arena.get<Cell>(hv.ncell[k] ? hv.ncell[k] : 1); lists.list[k] = hv.cells[k]; }
	hv.tb.tcells = c->arena.get<TCell>(hv.ntcell ? hv.ntcell : 1);
	hv.tb.tiles = c->arena.get<Tile>(hv.ntile ? hv.ntile : 1);
	hv.tb2.tcells = c->arena.get<TCell>(hv.ntcell2 ? hv.ntcell2 : 1);
	hv.tb2.tiles = c->arena.get<Tile>(hv.ntile2 ? hv.ntile2 : 1);
	k_cells<true><<<dim3(grid_for(hv.n, 128)), dim3(128), 0, st>>>(hv.rows, hv.n, m.beg, m.id, hv.winprod, hv.nwin, hv.cell_cap, hv.dense_min, hv.cnt, nullptr, hv.base, lists, segbase, nullptr, tile_kinds(hv));
	SPS_LAUNCH_CHECK();
	for (int kd = 0; kd < 2; ++kd) {
		TileBases &t = kd ? hv.tb2 : hv.tb;
		const uint32_t nd = kd ? hv.ntile2 : hv.ntile;
		if (nd < 2) continue;
		uint64_t *k0 = c->arena.get<uint64_t>(nd), *k1 = c->arena.get<uint64_t>(nd);
		uint32_t *p0 = c->arena.get<uint32_t>(nd), *p1 = c->arena.get<uint32_t>(nd);
		k_tile_keys<<<dim3(grid_for(nd)), dim3(256), 0, st>>>(t.tiles, nd, k0);
		SPS_LAUNCH_CHECK();
		int where = radix_sort_pairs(c, k0, p0, k1, p1, nd, wbits);
		Tile *sorted = c->arena.get<Tile>(nd);
		k_gather_tiles<<<dim3(grid_for(nd)), dim3(256), 0, st>>>(t.tiles, where ? p1 : p0, nd, sorted);
		SPS_LAUNCH_CHECK();
		t.tiles = sorted;
	}
	for (int k = 0; k < NCLS; ++k) {
		uint32_t nd = hv.ncell[k];
		if (nd < 2) continue;
		uint64_t *k0 = c->arena.get<uint64_t>(nd), *k1 = c->arena.get<uint64_t>(nd);
		uint32_t *p0 = c->arena.get<uint32_t>(nd), *p1 = c->arena.get<uint32_t>(nd);
		k_cell_keys<<<dim3(grid_for(nd)), dim3(256), 0, st>>>(hv.cells[k], nd, k == CLS_DENSE, k0);
		SPS_LAUNCH_CHECK();
		int where = radix_sort_pairs(c, k0, p0, k1, p1, nd, k == CLS_DENSE ? 16 + wbits : wbits);
		Cell *sorted = c->arena.get<Cell>(nd);
		k_gather_cells<<<dim3(grid_for(nd)), dim3(256), 0, st>>>(hv.cells[k], where ? p1 : p0, nd, sorted);
		SPS_LAUNCH_CHECK();
		hv.cells[k] = sorted;
	}
	// Measured on R-MAT scale-20: giving each XCD its own part of the list is SLOWER (dense 80 vs
	// 57 ms, hash 73 vs 62 ms) than letting all XCDs walk the same windows together, so the
	// partition stays an experiment behind SPSAMD_XCD=1.
	const bool xcd_aware = c->tune.xcd != 0;
	for (int k = 0; k < NCLS && xcd_aware; ++k) {
		uint32_t nd = hv.ncell[k];
		if (nd < 4096) continue;
		uint32_t *cost = c->arena.get<uint32_t>(nd);
		int64_t *pref = c->arena.get<int64_t>((size_t)nd + 1);
		k_cell_cost<<<dim3(grid_for(nd)), dim3(256), 0, st>>>(hv.cells[k], nd, k == CLS_DENSE ? 6000u : 2000u, cost);
		SPS_LAUNCH_CHECK();
		scan_exclusive_u32_i64(c, cost, pref, nd);
		hv.xb[k] = c->arena.get<uint32_t>(9);
		k_xcd_bounds<<<dim3(1), dim3(64), 0, st>>>(pref, nd, hv.xb[k]);
		SPS_LAUNCH_CHECK();
	}
}

// ---- products with a heavy row and more than 2^25 columns: by column blocks of B -----------------------------
// The heavy-row kernels index B by column windows and take 2048 of them (2^25 columns at 16384 per window), and their
// indices hold 12 bytes per row of B and window.  A wider op(B) with a heavy row -- or one whose indices would not fit the
// device -- is multiplied block by block: B restricted to `colblk` columns at a time (column indices rebased,
// scalek shifted) goes through the ordinary path into the COO sink, and the blocks' outputs -- each row-major -- are
// interleaved row by row (a row's tuples of block 0, then of block 1, ...: ascending columns).  The digest sink is fed
// from the blocks' tuples with their absolute columns.  A fallback for an uncommon shape, not a fast path: every block
// repeats the work on A, and the COO result is assembled by one extra pass over the output.

__global__ void k_col_flag(const int32_t *col, uint32_t n, uint32_t c0, uint32_t c1, uint8_t *flag)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) { const uint32_t cc = (uint32_t)col[t]; flag[t] = (cc >= c0 && cc < c1) ? 1 : 0; }
}

__global__ void k_col_compact(const int32_t *row, const int32_t *col, const double *val, const uint8_t *flag, const uint32_t *off, uint32_t n,
	uint32_t c0, int32_t *orow, int32_t *ocol, double *oval)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n && flag[t]) { const uint32_t o = off[t]; orow[o] = row[t]; ocol[o] = (int32_t)((uint32_t)col[t] - c0); oval[o] = val[t]; }
}

__global__ void k_block_digest(const int32_t *i, const int32_t *j, const double *v, uint64_t n, uint32_t c0, DigestSlot *slots,
	long long *row_nnz, double *row_sum)
{
	__shared__ unsigned long long s_u64[2 * 4];
	__shared__ double s_f64[4];
	unsigned long long cnt = 0, hash = 0; double sum = 0;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t jj = (uint32_t)j[t] + c0;
		++cnt; hash += mix64((uint32_t)i[t], jj); sum += v[t];
		if (row_nnz) { atomicAdd((unsigned long long *)&row_nnz[i[t]], 1ull); atomicAdd(&row_sum[i[t]], v[t]); }
	}
	digest_flush<256>(slots, cnt, hash, sum, s_u64, s_f64);
}

__global__ void k_add_col(int32_t *j, uint64_t n, uint32_t c0)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) j[t] = (int32_t)((uint32_t)j[t] + c0);
}

__global__ void k_row_add_counts(const uint32_t *rp, uint64_t nrow, uint32_t *cnt)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < nrow) cnt[r] += rp[r + 1] - rp[r];
}

__global__ void k_block_place(const int32_t *i, const int32_t *j, const double *v, uint64_t n, const uint32_t *rp, const int64_t *rowoff,
	const uint32_t *cur, int32_t *oi, int32_t *oj, double *ov)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const int32_t r = i[t];
	const int64_t d = rowoff[r] + (int64_t)cur[r] + (int64_t)(t - rp[r]);
	oi[d] = r; oj[d] = j[t]; ov[d] = v[t];
}

template <int S, int MODE>
static void launch_light_direct(spsamd_ctx *c, uint32_t nrow, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, const ConMat &B, bool k64, const EmitParams &ep, const SinkParams &sk, unsigned long long *pc)
{
	const unsigned per = 4u * (64u / S);
	const unsigned grid = std::min<unsigned>((nrow + per - 1) / per, (unsigned)c->num_cu * 8u * 4u);
	if (k64) k_light_direct<S, MODE, true><<<dim3(grid), dim3(256), 0, c->stream>>>(nrow, aptr, acol, aval, bptr, B.col, B.val, ep, sk, pc);
	else k_light_direct<S, MODE, false><<<dim3(grid), dim3(256), 0, c->stream>>>(nrow, aptr, acol, aval, bptr, B.col, B.val, ep, sk, pc);
	SPS_LAUNCH_CHECK();
}

template <int MODE>
static void launch_light_direct_s(spsamd_ctx *c, uint32_t maxp, uint32_t nrow, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, const ConMat &B, bool k64, const EmitParams &ep, const SinkParams &sk, unsigned long long *pc)
{
	if (maxp <= 8) launch_light_direct<8, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else if (maxp <= 16) launch_light_direct<16, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else if (maxp <= 32) launch_light_direct<32, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else launch_light_direct<64, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
}

// Every output row has at most `maxp` <= 64 products: one kernel, no symbolic phase (see k_light_direct).
static void spgemm_all_light(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, uint32_t maxp)
{
	hipStream_t st = c->stream;
	const ConMat &A = a.A, &B = a.B;
	const uint32_t nrow = (uint32_t)A.nrow;
	EmitParams ep{a.C, a.si.present ? a.si.pos : nullptr, a.si.val, a.sk.present ? a.sk.pos : nullptr, a.sk.val, c->tune.emit_path,
#ifdef SPSAMD_ABLATIONS
		c->tune.dbg,
#endif
		0u, B.ncol > 1 ? (uint32_t)(64 - __builtin_clzll((unsigned long long)(B.ncol - 1))) : 1u, 0, 0};
	// wide variant: (column << log2 S | A position) does not fit 32 bits, or an operand array reaches 4 GB
	const bool k64 = ep.ncolbits + 6u > 32u || A.nnz >= (1u << 29) || B.nnz >= (1u << 29) || A.nrow + 2 >= (uint64_t(1) << 30) || B.nrow + 3 >= (uint64_t(1) << 30);
	SinkParams sk{};
	sk.err = c->arena.get<uint32_t>(1);
	fill_zero(c, sk.err, sizeof(uint32_t));
	unsigned long long *pc = c->arena.get<unsigned long long>(1);
	fill_zero(c, pc, sizeof(unsigned long long));
	SPS_HIP(hipEventRecord(c->ev[2], st));
	SPS_HIP(hipEventRecord(c->ev[3], st));
	if (a.sink_kind != SPSAMD_SINK_COO) {
		DigestSlot *slots = c->arena.get<DigestSlot>(DIGEST_SLOTS + 1);
		fill_zero(c, slots, (DIGEST_SLOTS + 1) * sizeof(DigestSlot));
		sk.digest = slots;
		if (a.sink_flags & SPSAMD_SINK_ROWSTATS) {
			c->rowstat_n.ensure(A.nrow * sizeof(long long));
			c->rowstat_s.ensure(A.nrow * sizeof(double));
			fill_zero(c, c->rowstat_n.p, A.nrow * sizeof(long long));
			fill_zero(c, c->rowstat_s.p, A.nrow * sizeof(double));
			sk.row_nnz = (long long *)c->rowstat_n.p;
			sk.row_sum = (double *)c->rowstat_s.p;
			res->row_nnz = (const int64_t *)sk.row_nnz;
			res->row_sum = sk.row_sum;
		}
		launch_light_direct_s<MODE_DIGEST>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
		k_digest_reduce<<<dim3(1), dim3(64), 0, st>>>(slots, slots + DIGEST_SLOTS, sk.err);
		SPS_LAUNCH_CHECK();
		SPS_HIP(hipEventRecord(c->ev[4], st));
		DigestSlot d = read_back(c, slots + DIGEST_SLOTS);
		res->nnz = d.count; res->hash = d.hash; res->sum = d.sum;
	} else {
		// COO: one segment per row of op(A) (empty rows included): count, scan, store
		const size_t nsegs = nrow;
		uint32_t *segcount = c->arena.get<uint32_t>(nsegs + 1);
		uint32_t *segactual = c->arena.get<uint32_t>(nsegs + 1);
		int64_t *segoff = c->arena.get<int64_t>(nsegs + 1);
		sk.segcount = segcount; sk.segoff = segoff; sk.segactual = segactual;
		launch_light_direct_s<MODE_COUNT>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
		scan_exclusive_u32_i64(c, segcount, segoff, nsegs);
		const int64_t total = read_back(c, segoff + nsegs);
		OutSet &os = c->out[c->cur_out];
		os.i.ensure((size_t)total * sizeof(int32_t));
		os.j.ensure((size_t)total * sizeof(int32_t));
		os.v.ensure((size_t)total * sizeof(double));
		sk.out_i = (int32_t *)os.i.p; sk.out_j = (int32_t *)os.j.p; sk.out_v = (double *)os.v.p;
		fill_zero(c, pc, sizeof(unsigned long long));
		launch_light_direct_s<MODE_STORE>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
		SPS_HIP(hipEventRecord(c->ev[4], st));
		// the counting launch evaluated the same sums (ascending k, deterministic): the counts are exact, no holes
		res->nnz = (uint64_t)total;
		res->idx0 = sk.out_i; res->idx1 = sk.out_j; res->val = sk.out_v;
	}
	res->products = read_back(c, pc);
	res->rows_light = nrow; res->products_light = res->products; res->tuples_light = A.nnz;
	SPS_HIP(hipEventRecord(c->ev[7], st));
	SPS_HIP(hipEventSynchronize(c->ev[7]));
	res->ms_symbolic = elapsed(c->ev[1], c->ev[2]);
	res->ms_numeric = elapsed(c->ev[2], c->ev[7]);
	res->ms_light = elapsed(c->ev[3], c->ev[4]);
}

static void spgemm_once(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res)
{
	hipStream_t st = c->stream;
	const ConMat &A = a.A, &B = a.B;
	res->nnz_a = A.nnz; res->nnz_b = B.nnz;
	if (A.nnz == 0 || B.nnz == 0) return;           // empty product (also SURVEY Appendix A.3)

	SPS_HIP(hipEventRecord(c->ev[1], st));
	const uint32_t extra = a.sj.present ? 1u : 0u;
	uint32_t *bptr = dense_rowptr(c, B, extra);
	const int32_t *acol = A.col;
	const double *aval = A.val;
	if (a.sj.present) {
		int32_t *acol2 = c->arena.get<int32_t>(A.nnz);
		double *aval2 = c->arena.get<double>(A.nnz);
		k_apply_scalej<<<dim3(grid_for(A.nnz)), dim3(256), 0, st>>>(A.col, A.val, A.nnz, a.sj.pos, a.sj.val, (int32_t)B.nrow, acol2, aval2);
		SPS_LAUNCH_CHECK();
		acol = acol2; aval = aval2;
	}
	// ---- all rows light?  (longest A row) x (longest B row) <= 64: the direct kernel, no symbolic phase
	if (!c->tune.light_path) {
		const bool same = A.row == B.row && A.col == B.col && A.nnz == B.nnz && A.nrow == B.nrow;
		const uint32_t *aptr = same ? bptr : dense_rowptr(c, A, 0);
		uint32_t *mx = c->arena.get<uint32_t>(2);
		fill_zero(c, mx, 2 * sizeof(uint32_t));
		k_max_rowlen<<<dim3(std::min(grid_for(A.nrow), 1024u)), dim3(256), 0, st>>>(aptr, A.nrow, mx);
		SPS_LAUNCH_CHECK();
		if (!same) { k_max_rowlen<<<dim3(std::min(grid_for(B.nrow), 1024u)), dim3(256), 0, st>>>(bptr, B.nrow, mx + 1); SPS_LAUNCH_CHECK(); }
		struct { uint32_t a, b; } hm = read_back(c, (const decltype(hm) *)mx);
		if (same) hm.b = hm.a;
		if ((uint64_t)hm.a * hm.b <= 64 && A.nrow < (uint64_t(1) << 32)) {
			spgemm_all_light(c, a, res, aptr, acol, aval, bptr, (uint32_t)((uint64_t)hm.a * hm.b));
			return;
		}
	}
	// ---- row structure of A (dim_beginnings)
	RowList rl;
	dim_beginnings(c, A, &rl);

	// ---- symbolic: products per A tuple, per row, bins
	uint32_t *elen = c->arena.get<uint32_t>(A.nnz);
	uint32_t *elo = c->arena.get<uint32_t>(A.nnz);
	int64_t *pref = c->arena.get<int64_t>((size_t)A.nnz + 1);
	k_elem_len<<<dim3(grid_for(A.nnz)), dim3(256), 0, st>>>(acol, bptr, A.nnz, elo, elen);
	SPS_LAUNCH_CHECK();
	scan_exclusive_u32_i64(c, elen, pref, A.nnz);
	uint32_t *rprod = c->arena.get<uint32_t>(rl.nrows);
	uint8_t *rbin = c->arena.get<uint8_t>(rl.nrows);
	BinCounters *bc = c->arena.get<BinCounters>(1);
	fill_zero(c, bc, sizeof(BinCounters));
	k_classify<<<dim3(grid_for(rl.nrows, 256 * CLS_ITEMS)), dim3(256), 0, st>>>(rl.beg, rl.id, rl.nrows, pref, elen,
		a.si.present ? a.si.pos : nullptr, a.si.val, rprod, rbin, bc);
	SPS_LAUNCH_CHECK();
	BinCounters hbc = read_back(c, bc);
	if (hbc.too_big) throw Error{SPSAMD_EINVAL, "an output row with more than 2^32-1 scalar products is not supported"};

	Bins bins;
	BinOffsets bo;
	uint32_t run = 0;
	for (int b = 0; b < NBIN; ++b) {
		bins.count[b] = b == 0 ? 0 : (uint32_t)hbc.rows[b];
		bins.off[b] = bo.off[b] = run;
		run += bins.count[b];
	}
	bins.off[NBIN] = bo.off[NBIN] = run;
	int whole_bin = 0;                               // a light bin that holds every row needs no list
	for (int b = 1; b <= 4; ++b) if (bins.count[b] == rl.nrows) whole_bin = b;
	bins.rows = nullptr;
	if (!whole_bin) {
		bins.rows = c->arena.get<uint32_t>(run ? run : 1);
		uint32_t *cursor = c->arena.get<uint32_t>(NBIN);
		fill_zero(c, cursor, NBIN * sizeof(uint32_t));
		k_bin_scatter<<<dim3(grid_for(rl.nrows, 256 * CLS_ITEMS)), dim3(256), 0, st>>>(rbin, rl.nrows, bo, cursor, bins.rows);
		SPS_LAUNCH_CHECK();
	}

	uint64_t P = 0;
	for (int b = 1; b < NBIN; ++b) P += hbc.prods[b];
	res->products = P;
	res->rows_light = hbc.rows[1] + hbc.rows[2] + hbc.rows[3] + hbc.rows[4];
	res->rows_mid = hbc.rows[5] + hbc.rows[6] + hbc.rows[7];
	res->rows_heavy = hbc.rows[8];
	res->products_light = hbc.prods[1] + hbc.prods[2] + hbc.prods[3] + hbc.prods[4];
	res->products_mid = hbc.prods[5] + hbc.prods[6] + hbc.prods[7];
	res->products_heavy = hbc.prods[8];
	res->tuples_light = hbc.tuples[1] + hbc.tuples[2] + hbc.tuples[3] + hbc.tuples[4];
	res->tuples_mid = hbc.tuples[5] + hbc.tuples[6] + hbc.tuples[7];
	res->tuples_heavy = hbc.tuples[8];

	BTup *btup = c->arena.get<BTup>((size_t)B.nnz + DENSE_R);         // a dense-cell item reads R tuples: slack after the last one
	k_pack_b<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.col, B.val, B.nnz, btup);
	SPS_LAUNCH_CHECK();
	RowMeta m{rl.beg, rl.id, acol, aval, bptr, btup, btup, elo, elen};
#ifdef SPSAMD_ABLATIONS
	SPS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_abl), &c->tune.dbg, sizeof(int), 0, hipMemcpyHostToDevice, st));
#endif
	EmitParams ep{a.C, a.si.present ? a.si.pos : nullptr, a.si.val, a.sk.present ? a.sk.pos : nullptr, a.sk.val,
		c->tune.emit_path,
#ifdef SPSAMD_ABLATIONS
		c->tune.dbg,
#endif
		0u,
		B.ncol > 1 ? (uint32_t)(64 - __builtin_clzll((unsigned long long)(B.ncol - 1))) : 1u,
		(a.sink_flags & SPSAMD_SINK_ORDERED) ? 1 : 0,
		((a.sink_flags & SPSAMD_SINK_EXACT_PATTERN) && !(a.sink_flags & SPSAMD_SINK_ORDERED)) ? 1 : 0};

	// ---- segments (one per light/mid row, one per cell of a heavy row) and the heavy rows' cells
	const bool coo = a.sink_kind == SPSAMD_SINK_COO;
	uint32_t *nseg = c->arena.get<uint32_t>(rl.nrows);
	fill_u32(c, nseg, 1u, rl.nrows);
	Heavy hv;
	hv.n = bins.count[8];
	hv.coo = coo;
	if (hv.n) heavy_prepare(c, hv, bins, m, B, bptr, extra, nseg, (a.sink_flags & SPSAMD_SINK_ORDERED) != 0, (a.sink_flags & SPSAMD_SINK_EXACT_PATTERN) != 0);
	ep.wshift = hv.W == 8192 ? 13u : 14u;
	uint32_t *segbase = nullptr;
	int64_t nsegs = 0;
	if (coo) {
		int64_t *segbase64 = c->arena.get<int64_t>((size_t)rl.nrows + 1);
		scan_exclusive_u32_i64(c, nseg, segbase64, rl.nrows);
		nsegs = read_back(c, segbase64 + rl.nrows);
		if (nsegs >= (int64_t(1) << 32)) throw Error{SPSAMD_EINVAL, "too many output segments"};
		segbase = c->arena.get<uint32_t>((size_t)rl.nrows + 1);
		scan_exclusive_u32_u32(c, nseg, segbase, rl.nrows);
	}
	if (hv.n) heavy_cells(c, hv, m, segbase);
	MidCells mc;
	for (int k = 0; k < 3; ++k) {
		uint32_t nb = bins.count[5 + k];
		if (!nb) continue;
		mc.cells[k] = c->arena.get<Cell>(nb);
		k_row_cells<<<dim3(grid_for(nb)), dim3(256), 0, st>>>(bins.rows + bins.off[5 + k], nb, rl.beg, rl.id, rprod, segbase, mc.cells[k]);
		SPS_LAUNCH_CHECK();
	}
	res->cells_hash = (uint64_t)hv.ncell[0] + hv.ncell[1] + hv.ncell[2] + hv.ncell[3] + hv.ntcell + hv.ntcell2;
	res->cells_dense = hv.ncell[CLS_DENSE];
	res->window = hv.n ? (uint32_t)hv.W : 0u;
	res->products_dense = hv.clsprod[CLS_DENSE];
	res->products_tiles = hv.clsprod[NCLS];
	res->products_direct = hv.clsprod[NCLS + 1];
	SPS_HIP(hipEventRecord(c->ev[2], st));

	// ---- numeric
	SinkParams sk{};
	sk.segbase = segbase;
	sk.err = c->arena.get<uint32_t>(1);
	fill_zero(c, sk.err, sizeof(uint32_t));
	float ms_light = 0, ms_mid = 0, ms_heavy = 0, ms_dense = 0;
	if (!coo) {
		DigestSlot *slots = c->arena.get<DigestSlot>(DIGEST_SLOTS + 1);
		fill_zero(c, slots, (DIGEST_SLOTS + 1) * sizeof(DigestSlot));
		sk.digest = slots;
		if (a.sink_flags & SPSAMD_SINK_ROWSTATS) {
			c->rowstat_n.ensure(A.nrow * sizeof(long long));
			c->rowstat_s.ensure(A.nrow * sizeof(double));
			fill_zero(c, c->rowstat_n.p, A.nrow * sizeof(long long));
			fill_zero(c, c->rowstat_s.p, A.nrow * sizeof(double));
			sk.row_nnz = (long long *)c->rowstat_n.p;
			sk.row_sum = (double *)c->rowstat_s.p;
			res->row_nnz = (const int64_t *)sk.row_nnz;
			res->row_sum = sk.row_sum;
		}
		SPS_HIP(hipEventRecord(c->ev[3], st));
		launch_light<MODE_DIGEST>(c, bins, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[4], st));
		launch_mid<MODE_DIGEST>(c, bins, mc, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[5], st));
		launch_heavy_hash<MODE_DIGEST>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[6], st));
		launch_heavy_dense<MODE_DIGEST>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[8], st));
		k_digest_reduce<<<dim3(1), dim3(64), 0, st>>>(slots, slots + DIGEST_SLOTS, sk.err);
		SPS_LAUNCH_CHECK();
		DigestSlot d = read_back(c, slots + DIGEST_SLOTS);
		if (d.pad) throw Error{SPSAMD_EINVAL, "internal error: a numeric kernel met a cell larger than its class allows"};
		res->nnz = d.count; res->hash = d.hash; res->sum = d.sum;
		ms_light = elapsed(c->ev[3], c->ev[4]); ms_mid = elapsed(c->ev[4], c->ev[5]);
		ms_heavy = elapsed(c->ev[5], c->ev[8]); ms_dense = elapsed(c->ev[6], c->ev[8]);
	} else {
		uint32_t *segcount = c->arena.get<uint32_t>((size_t)nsegs + 1);
		uint32_t *segactual = c->arena.get<uint32_t>((size_t)nsegs + 1);
		int64_t *segoff = c->arena.get<int64_t>((size_t)nsegs + 1);
		fill_zero(c, segcount, ((size_t)nsegs + 1) * sizeof(uint32_t));
		fill_zero(c, segactual, ((size_t)nsegs + 1) * sizeof(uint32_t));
		sk.segcount = segcount; sk.segoff = segoff; sk.segactual = segactual;

		SPS_HIP(hipEventRecord(c->ev[3], st));
		launch_light<MODE_COUNT>(c, bins, m, ep, sk);
		launch_mid<MODE_COUNT>(c, bins, mc, m, ep, sk);
		launch_heavy_hash<MODE_COUNT>(c, hv, m, ep, sk);
		launch_heavy_dense<MODE_COUNT>(c, hv, m, ep, sk);
		scan_exclusive_u32_i64(c, segcount, segoff, (size_t)nsegs);
		int64_t reserved = read_back(c, segoff + nsegs);
		OutSet &os = c->out[c->cur_out];              // chosen by multiply_body: never the set an operand lives in
		os.i.ensure((size_t)reserved * sizeof(int32_t));
		os.j.ensure((size_t)reserved * sizeof(int32_t));
		os.v.ensure((size_t)reserved * sizeof(double));
		sk.out_i = (int32_t *)os.i.p; sk.out_j = (int32_t *)os.j.p; sk.out_v = (double *)os.v.p;
		SPS_HIP(hipEventRecord(c->ev[4], st));
		launch_light<MODE_STORE>(c, bins, m, ep, sk);
		launch_mid<MODE_STORE>(c, bins, mc, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[5], st));
		launch_heavy_hash<MODE_STORE>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[6], st));
		launch_heavy_dense<MODE_STORE>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[8], st));
		struct HolesErr { unsigned long long holes, err; };
		unsigned long long *holes = c->arena.get<unsigned long long>(2);
		fill_zero(c, holes, 2 * sizeof(unsigned long long));
		k_seg_holes<<<dim3(grid_for((size_t)nsegs)), dim3(256), 0, st>>>(segcount, segactual, (uint32_t)nsegs, holes, sk.err);
		SPS_LAUNCH_CHECK();
		const HolesErr he = read_back(c, (const HolesErr *)holes);
		if (he.err) throw Error{SPSAMD_EINVAL, "internal error: a numeric kernel met a cell larger than its class allows"};
		unsigned long long nholes = he.holes;
		uint64_t nnz = (uint64_t)reserved - nholes;
		if (nholes) {
			// sums that cancelled to exactly 0 left gaps: close them (rare path)
			int64_t *newoff = c->arena.get<int64_t>((size_t)nsegs + 1);
			scan_exclusive_u32_i64(c, segactual, newoff, (size_t)nsegs);
			int32_t *ti = c->arena.get<int32_t>(nnz ? nnz : 1), *tj = c->arena.get<int32_t>(nnz ? nnz : 1);
			double *tv = c->arena.get<double>(nnz ? nnz : 1);
			k_seg_gather<<<dim3(grid_for((size_t)nsegs)), dim3(256), 0, st>>>(segoff, newoff, segactual, (uint32_t)nsegs,
				sk.out_i, sk.out_j, sk.out_v, ti, tj, tv);
			SPS_LAUNCH_CHECK();
			SPS_HIP(hipMemcpyAsync(sk.out_i, ti, nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
			SPS_HIP(hipMemcpyAsync(sk.out_j, tj, nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
			SPS_HIP(hipMemcpyAsync(sk.out_v, tv, nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
			SPS_HIP(hipStreamSynchronize(st));
		}
		res->nnz = nnz;
		res->idx0 = sk.out_i; res->idx1 = sk.out_j; res->val = sk.out_v;
		ms_light = elapsed(c->ev[4], c->ev[5]);
		ms_heavy = elapsed(c->ev[5], c->ev[8]); ms_dense = elapsed(c->ev[6], c->ev[8]);
	}
	SPS_HIP(hipEventRecord(c->ev[7], st));
	SPS_HIP(hipEventSynchronize(c->ev[7]));
	res->ms_symbolic = elapsed(c->ev[1], c->ev[2]);
	res->ms_numeric = elapsed(c->ev[2], c->ev[7]);
	res->ms_light = ms_light; res->ms_mid = ms_mid; res->ms_heavy = ms_heavy; res->ms_dense = ms_dense;
	if (hv.n) { res->ms_tiles = elapsed(c->ev2[0], c->ev2[1]); res->ms_direct = elapsed(c->ev2[1], c->ev2[2]); }    // (COO: of the STORE launches)
}

static void spgemm_column_blocks(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res, uint64_t colblk)
{
	hipStream_t st = c->stream;
	const ConMat &A = a.A, &B = a.B;
	const bool coo = a.sink_kind == SPSAMD_SINK_COO;
	const uint32_t nblk = (uint32_t)((B.ncol + colblk - 1) / colblk);
	struct BlockOut { int32_t *i = nullptr, *j = nullptr; double *v = nullptr; uint64_t n = 0; };
	struct Blocks {                                                 // the blocks' COO outputs until they are interleaved
		std::vector<BlockOut> b;
		~Blocks() { for (auto &x : b) { (void)hipFree(x.i); (void)hipFree(x.j); (void)hipFree(x.v); } }
	} blocks;
	DigestSlot *slots = nullptr;
	long long *row_nnz = nullptr; double *row_sum = nullptr;
	if (!coo) {
		slots = c->arena.get<DigestSlot>(DIGEST_SLOTS + 1);
		fill_zero(c, slots, (DIGEST_SLOTS + 1) * sizeof(DigestSlot));
		if (a.sink_flags & SPSAMD_SINK_ROWSTATS) {
			c->rowstat_n.ensure(A.nrow * sizeof(long long));
			c->rowstat_s.ensure(A.nrow * sizeof(double));
			fill_zero(c, c->rowstat_n.p, A.nrow * sizeof(long long));
			fill_zero(c, c->rowstat_s.p, A.nrow * sizeof(double));
			row_nnz = (long long *)c->rowstat_n.p; row_sum = (double *)c->rowstat_s.p;
		}
	}
	spsamd_result acc{};
	for (uint32_t s = 0; s < nblk; ++s) {
		const uint64_t c0 = (uint64_t)s * colblk, c1 = std::min<uint64_t>(B.ncol, c0 + colblk);
		const Arena::Mark mk = c->arena.mark();
		// B restricted to columns [c0, c1), rebased; the tuples keep their (row, column) order
		uint8_t *flag = c->arena.get<uint8_t>(B.nnz);
		uint32_t *off = c->arena.get<uint32_t>((size_t)B.nnz + 1);
		k_col_flag<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.col, B.nnz, (uint32_t)c0, (uint32_t)c1, flag);
		SPS_LAUNCH_CHECK();
		scan_exclusive_u8_u32(c, flag, off, B.nnz);
		const uint32_t nb = read_back(c, off + B.nnz);
		if (nb) {
			MultiplyArgs as = a;
			as.B.row = c->arena.get<int32_t>(nb); as.B.col = c->arena.get<int32_t>(nb); as.B.val = c->arena.get<double>(nb);
			as.B.nnz = nb; as.B.nrow = B.nrow; as.B.ncol = c1 - c0;
			k_col_compact<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.row, B.col, B.val, flag, off, B.nnz, (uint32_t)c0, as.B.row, as.B.col, as.B.val);
			SPS_LAUNCH_CHECK();
			if (as.sk.present) { as.sk.pos += c0; as.sk.dim = c1 - c0; }
			as.sink_kind = SPSAMD_SINK_COO;
			as.sink_flags = a.sink_flags & (SPSAMD_SINK_ORDERED | SPSAMD_SINK_EXACT_PATTERN);
			spsamd_result rs{};
			try { spgemm_once(c, as, &rs); }
			catch (const TooWide &) { throw Error{SPSAMD_ENOMEM, "the window indices of a column block of op(B) do not fit the device"}; }
			acc.products += rs.products; acc.products_light += rs.products_light; acc.products_mid += rs.products_mid;
			acc.products_heavy += rs.products_heavy; acc.products_dense += rs.products_dense; acc.products_tiles += rs.products_tiles;
			acc.cells_hash += rs.cells_hash; acc.cells_dense += rs.cells_dense; acc.window = std::max(acc.window, rs.window);
			acc.rows_light = std::max(acc.rows_light, rs.rows_light); acc.rows_mid = std::max(acc.rows_mid, rs.rows_mid);
			acc.rows_heavy = std::max(acc.rows_heavy, rs.rows_heavy);
			acc.ms_symbolic += rs.ms_symbolic; acc.ms_numeric += rs.ms_numeric; acc.ms_light += rs.ms_light; acc.ms_mid += rs.ms_mid;
			acc.ms_heavy += rs.ms_heavy; acc.ms_dense += rs.ms_dense; acc.ms_tiles += rs.ms_tiles;
			if (rs.nnz) {
				if (!coo) {
					k_block_digest<<<dim3(std::min<unsigned>(grid_for((size_t)rs.nnz), 4096u)), dim3(256), 0, st>>>(rs.idx0, rs.idx1, rs.val, rs.nnz, (uint32_t)c0, slots, row_nnz, row_sum);
					SPS_LAUNCH_CHECK();
				} else {
					BlockOut bo;
					bo.n = rs.nnz;
					if (hipMalloc((void **)&bo.i, rs.nnz * sizeof(int32_t)) != hipSuccess || hipMalloc((void **)&bo.j, rs.nnz * sizeof(int32_t)) != hipSuccess ||
						hipMalloc((void **)&bo.v, rs.nnz * sizeof(double)) != hipSuccess) {
						(void)hipGetLastError();
						(void)hipFree(bo.i); (void)hipFree(bo.j); (void)hipFree(bo.v);
						throw Error{SPSAMD_ENOMEM, "hipMalloc of a column block's output failed"};
					}
					blocks.b.push_back(bo);
					SPS_HIP(hipMemcpyAsync(bo.i, rs.idx0, rs.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
					SPS_HIP(hipMemcpyAsync(bo.j, rs.idx1, rs.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
					SPS_HIP(hipMemcpyAsync(bo.v, rs.val, rs.nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
					k_add_col<<<dim3(grid_for((size_t)rs.nnz)), dim3(256), 0, st>>>(bo.j, rs.nnz, (uint32_t)c0);
					SPS_LAUNCH_CHECK();
				}
			}
		}
		SPS_HIP(hipStreamSynchronize(st));                          // (the block's workspace is handed back)
		c->arena.rewind(mk);
	}
	*res = acc;
	res->nnz_a = A.nnz; res->nnz_b = B.nnz;
	if (!coo) {
		uint32_t *noerr = c->arena.get<uint32_t>(1);
		fill_zero(c, noerr, sizeof(uint32_t));
		k_digest_reduce<<<dim3(1), dim3(64), 0, st>>>(slots, slots + DIGEST_SLOTS, noerr);
		SPS_LAUNCH_CHECK();
		const DigestSlot d = read_back(c, slots + DIGEST_SLOTS);
		res->nnz = d.count; res->hash = d.hash; res->sum = d.sum;
		if (row_nnz) { res->row_nnz = (const int64_t *)row_nnz; res->row_sum = row_sum; }
		return;
	}
	// ---- interleave the blocks row by row
	uint64_t total = 0;
	for (auto &x : blocks.b) total += x.n;
	OutSet &os = c->out[c->cur_out];
	os.i.ensure((size_t)total * sizeof(int32_t)); os.j.ensure((size_t)total * sizeof(int32_t)); os.v.ensure((size_t)total * sizeof(double));
	int32_t *oi = (int32_t *)os.i.p, *oj = (int32_t *)os.j.p; double *ov = (double *)os.v.p;
	uint32_t *cnt = c->arena.get<uint32_t>(A.nrow ? A.nrow : 1), *cur = c->arena.get<uint32_t>(A.nrow ? A.nrow : 1);
	int64_t *rowoff = c->arena.get<int64_t>((size_t)A.nrow + 1);
	fill_zero(c, cnt, A.nrow * sizeof(uint32_t));
	fill_zero(c, cur, A.nrow * sizeof(uint32_t));
	auto block_rowptr = [&](const BlockOut &x) {
		ConMat t; t.row = x.i; t.nnz = (uint32_t)x.n; t.nrow = A.nrow;
		return dense_rowptr(c, t, 0);
	};
	for (auto &x : blocks.b) {
		if (x.n >= (uint64_t(1) << 32)) throw Error{SPSAMD_EINVAL, "a column block of the product has 2^32 tuples or more"};
		const Arena::Mark mk = c->arena.mark();
		const uint32_t *rp = block_rowptr(x);
		k_row_add_counts<<<dim3(grid_for(A.nrow)), dim3(256), 0, st>>>(rp, A.nrow, cnt);
		SPS_LAUNCH_CHECK();
		SPS_HIP(hipStreamSynchronize(st));
		c->arena.rewind(mk);
	}
	scan_exclusive_u32_i64(c, cnt, rowoff, A.nrow);
	for (auto &x : blocks.b) {
		const Arena::Mark mk = c->arena.mark();
		const uint32_t *rp = block_rowptr(x);
		k_block_place<<<dim3(grid_for((size_t)x.n)), dim3(256), 0, st>>>(x.i, x.j, x.v, x.n, rp, rowoff, cur, oi, oj, ov);
		SPS_LAUNCH_CHECK();
		k_row_add_counts<<<dim3(grid_for(A.nrow)), dim3(256), 0, st>>>(rp, A.nrow, cur);
		SPS_LAUNCH_CHECK();
		SPS_HIP(hipStreamSynchronize(st));
		c->arena.rewind(mk);
	}
	res->nnz = total;
	res->idx0 = oi; res->idx1 = oj; res->val = ov;
}

void spgemm(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res)
{
	const Arena::Mark mk = c->arena.mark();
	uint64_t colblk = 0;
	try {
		spgemm_once(c, a, res);
		return;
	} catch (const TooWide &w) {
		colblk = w.width;                                           // a heavy row and too many columns for one pass: block by block
	}
	SPS_HIP(hipStreamSynchronize(c->stream));
	c->arena.rewind(mk);
	const spsamd_result keep = *res;
	*res = spsamd_result{};
	res->shape0 = keep.shape0; res->shape1 = keep.shape1;
	spgemm_column_blocks(c, a, res, colblk);
	res->shape0 = keep.shape0; res->shape1 = keep.shape1;
}

} // namespace spsamd
