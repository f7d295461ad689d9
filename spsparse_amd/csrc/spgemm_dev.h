// spgemm_dev.h -- device-side definitions shared by the kernel translation units of the multiply path
// (k_light.hip, k_hash.hip, k_dense.hip, k_tiles.hip, symbolic_heavy.hip, spgemm.hip): sink / emit parameters, the B
// tuple, cells and tiles, the segment expansions, the EXACT_PATTERN bookkeeping, the digest flush.
#pragma once
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
static __device__ int g_abl;                // the same switches for device functions that do not see EmitParams (one copy per translation unit: set_ablation_word)
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
	unsigned long long *row_hash;          // ... and the row's own index hash, sum of mix64(i, j) over its tuples
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

// (BTup, one B tuple as the numeric kernels read it -- column and value side by side, 12 bytes -- is defined in internal.h)
__device__ __forceinline__ double btup_val(const BTup &t) { return __hiloint2double((int)t.vhi, (int)t.vlo); }

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
__device__ inline double ordered_sum_wave(const RowMeta &m, uint32_t beg, uint32_t end, int32_t col)
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
// A cell's record, read ahead of its turn, must STAY in flight: read through a uniform address the compiler moves it to
// scalar registers at once (global_load, s_waitcnt vmcnt(0), v_readfirstlane -- a memory latency per cell for every wave, and
// every other load in flight drained with it).  So the address gets a per-lane zero the compiler cannot see through
// (cell_pend_zero), the words wait in vector registers (CellPend), and they are made scalar when the record's turn comes.
struct CellPend { uint32_t w[6]; };
__device__ __forceinline__ uint32_t cell_pend_zero() { uint32_t z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ CellPend cell_pend_load(const Cell *cells, uint32_t idx, uint32_t zlane)
{
	const uint32_t *p = reinterpret_cast<const uint32_t *>(cells + idx) + zlane;
	return CellPend{{p[0], p[1], p[2], p[3], p[4], p[5]}};
}
__device__ __forceinline__ Cell cell_from_pend(const CellPend &r)
{
	Cell c;
	c.beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w[0]); c.end = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w[1]);
	c.rowid = __builtin_amdgcn_readfirstlane((int)r.w[2]); c.seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w[3]);
	c.prods = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w[4]);
	const uint32_t wab = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.w[5]);
	c.wa = (uint16_t)(wab & 0xFFFFu); c.wb = (uint16_t)(wab >> 16); c.pad[0] = c.pad[1] = 0;
	return c;
}

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

// One lane's LDS fetch-add, spelled as the instruction: the compiler's atomic optimiser otherwise wraps
// the (already wave-aggregated) add into another mbcnt / readfirstlane / multiply sequence.
__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t *p, uint32_t v)
{
	uint32_t r;
	const uint32_t a = (uint32_t)(uintptr_t)p;       // LDS byte offset = low half of the flat address of a __shared__ object
	asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory");
	return r;
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
		atomicOr(reinterpret_cast<uint32_t *>(X.bmask) + (myfirst >> 5), 1u << (myfirst & 31u));
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

// tile kernels' sizes (k_tiles.hip), which the symbolic phase also needs
constexpr int TILE2_NT = 512;
constexpr int TILE2_T = 4096;
constexpr int TILE2_ITEMS = 8192;        // items per tile: bitmap of 128 words, two per lane
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

} // namespace spsamd
