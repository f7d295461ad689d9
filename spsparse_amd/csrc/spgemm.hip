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
//   mid        64 < P_r <= 4096: one workgroup per row, LDS hash accumulator
//              keyed by column (ds_cmpswap + ds_add_f64), then an in-LDS
//              bitonic sort of the surviving columns for ordered emission.
//   heavy      P_r > 4096: persistent workgroups pull rows from a ticket; the
//              column space is cut into windows of W columns whose dense f64
//              accumulator lives in LDS; B's row panels are pre-indexed per
//              window so each (row, window) reads exactly its B segments.
//
// Output semantics follow multiply_sparse.hpp:238-243: exact-zero sums are
// dropped, value = sum * C * a_scale * b_scale, tuples in ascending (i, j).
// No MFMA: 2 flops per 12 bytes read.
#include "internal.h"
#include "devutil.h"

#include <algorithm>

namespace spsamd {

enum { MODE_COUNT = 0, MODE_STORE = 1, MODE_DIGEST = 2 };

constexpr int NBIN = 9;          // 0 none | 1..4 light (S = 8,16,32,64) | 5..7 mid (T = 1024,4096,8192) | 8 heavy
constexpr uint32_t MID_MAX = 4096;
constexpr int DIGEST_SLOTS = 1024;

struct EmitParams {
	double C;
	const int32_t *si_pos; const double *si_val;     // row scale (null: none)
	const int32_t *sk_pos; const double *sk_val;     // column scale (null: none)
};

struct DigestSlot { unsigned long long count; unsigned long long hash; double sum; unsigned long long pad; };

struct SinkParams {
	const uint32_t *segbase;        // per non-empty A row: first segment id        (COUNT / STORE)
	uint32_t *segcount;             // per segment: tuples reserved                 (COUNT writes)
	const int64_t *segoff;          // per segment: output offset                   (STORE reads)
	uint32_t *segactual;            // per segment: tuples written                  (STORE writes)
	int32_t *out_i; int32_t *out_j; double *out_v;
	DigestSlot *digest;             // DIGEST_SLOTS accumulators
	long long *row_nnz; double *row_sum;   // optional row statistics (DIGEST)
};

struct RowMeta {
	const uint32_t *beg;            // per non-empty A row: first tuple (+ sentinel)
	const int32_t *id;              // per non-empty A row: row index
	const int32_t *acol;            // A tuples: inner index k
	const double *aval;             // A tuples: value (already times scalej)
	const uint32_t *bptr;           // B dense row pointer
	const int32_t *bcol;
	const double *bval;
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

__global__ void k_elem_len(const int32_t *acol, const uint32_t *bptr, uint32_t n, uint32_t *len)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e < n) { int32_t k = acol[e]; len[e] = bptr[k + 1] - bptr[k]; }
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

struct BinCounters { unsigned long long rows[NBIN]; unsigned long long prods[NBIN]; unsigned long long tuples[NBIN]; };

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
__global__ __launch_bounds__(256) void k_classify(const uint32_t *beg, const int32_t *id, uint32_t nrows, const int64_t *pref,
	const int32_t *si_pos, const double *si_val, uint32_t *rprod, uint8_t *rbin, BinCounters *bc)
{
	// bin statistics are reduced in LDS first: one global atomic per (workgroup, bin)
	__shared__ unsigned int s_rows[NBIN];
	__shared__ unsigned long long s_prods[NBIN];
	__shared__ unsigned long long s_tuples[NBIN];
	if (threadIdx.x < NBIN) { s_rows[threadIdx.x] = 0; s_prods[threadIdx.x] = 0; s_tuples[threadIdx.x] = 0; }
	__syncthreads();
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < nrows) {
		uint32_t P = (uint32_t)(pref[beg[r + 1]] - pref[beg[r]]);
		if (si_pos) {
			int32_t q = si_pos[id[r]];
			if (q < 0 || si_val[q] == 0) P = 0;
		}
		int b = bin_of(P);
		rprod[r] = P;
		rbin[r] = (uint8_t)b;
		atomicAdd(&s_rows[b], 1u);
		atomicAdd(&s_prods[b], (unsigned long long)P);
		atomicAdd(&s_tuples[b], (unsigned long long)(beg[r + 1] - beg[r]));
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
	// a workgroup reserves one range per bin with a single global atomic
	__shared__ unsigned int s_cnt[NBIN];
	__shared__ unsigned int s_base[NBIN];
	if (threadIdx.x < NBIN) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	int b = r < nrows ? rbin[r] : 0;
	unsigned int local = 0;
	if (b) local = atomicAdd(&s_cnt[b], 1u);
	__syncthreads();
	if (threadIdx.x < NBIN && threadIdx.x > 0 && s_cnt[threadIdx.x])
		s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
	__syncthreads();
	if (b) binrows[bo.off[b] + s_base[b] + local] = r;
}

// ====================================================================== light rows

// One wave handles G = 64/S rows, S product slots each.
template <int S, int MODE>
__global__ __launch_bounds__(256) void k_light(const uint32_t *binrows, uint32_t nbin, RowMeta m, EmitParams ep, SinkParams sk)
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
	const uint32_t rix = (blockIdx.x * 4u + w) * G + g;
	const bool has_row = rix < nbin;
	const uint32_t r = has_row ? binrows[rix] : 0u;
	const uint32_t beg = has_row ? m.beg[r] : 0u;
	const uint32_t end = has_row ? m.beg[r + 1] : 0u;

	// ---- expand: slot t of the row's P products -> (A tuple, B tuple)
	uint32_t La = end - beg, maxLa = La;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) maxLa = max(maxLa, (uint32_t)__shfl_xor((int)maxLa, d, 64));
	uint32_t off = 0;
	for (uint32_t base = 0; base < maxLa; base += S) {
		uint32_t e = beg + base + s;
		bool act = has_row && e < end;
		uint32_t lo = 0, len = 0;
		if (act) { int32_t k = m.acol[e]; lo = m.bptr[k]; len = m.bptr[k + 1] - lo; }
		uint32_t inc = len;
#pragma unroll
		for (int d = 1; d < S; d <<= 1) {
			uint32_t o = (uint32_t)__shfl_up((int)inc, d, S);
			if ((int)s >= d) inc += o;
		}
		uint32_t ex = off + inc - len;
		for (uint32_t t = 0; t < len; ++t) {        // ex + t < S because P_r <= S
			s_apos[w][g * S + ex + t] = e;
			s_bpos[w][g * S + ex + t] = lo + t;
		}
		off += (uint32_t)__shfl((int)inc, (int)(g * S + S - 1), 64);
	}
	s_key2[w][lane] = ~0ull;
	__syncthreads();

	// ---- product + key (col, A position): ascending A position = ascending k
	const bool act = has_row && s < off;
	uint64_t key = ~0ull;
	double prod = 0;
	if (act) {
		uint32_t ap = s_apos[w][lane], bp = s_bpos[w][lane];
		prod = m.aval[ap] * m.bval[bp];
		key = ((uint64_t)(uint32_t)m.bcol[bp] << 32) | (uint64_t)ap;
	}
	s_key[w][lane] = key;
	__syncthreads();
	// ---- rank inside the row's S slots (keys are unique), scatter to sorted order
	uint32_t rank = 0;
#pragma unroll 8
	for (int j = 0; j < S; ++j) rank += (s_key[w][g * S + j] < key) ? 1u : 0u;
	if (act) { s_key2[w][g * S + rank] = key; s_val2[w][g * S + rank] = prod; }
	__syncthreads();

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
		digest_flush<256>(sk.digest, cnt, hash, vs, s_u64, s_f64);
	}
}

// ====================================================================== mid rows (LDS hash)

template <int NT>
__device__ __forceinline__ uint32_t find_entry(const uint32_t *epref, uint32_t p)
{
	// largest q in [0, NT) with epref[q] <= p   (epref ascending, epref[0] = 0)
	uint32_t lo = 0, hi = NT;
	while (hi - lo > 1) {
		uint32_t mid = (lo + hi) >> 1;
		if (epref[mid] <= p) lo = mid; else hi = mid;
	}
	return lo;
}

// One workgroup per row; T hash slots (T/2 = the bin's product cap).
template <int T, int MODE>
__global__ __launch_bounds__(256) void k_mid(const uint32_t *binrows, uint32_t nbin, RowMeta m, EmitParams ep, SinkParams sk)
{
	constexpr int NT = 256;
	constexpr int LOGT = T == 1024 ? 10 : (T == 4096 ? 12 : 13);
	__shared__ int32_t h_key[T];
	__shared__ double h_val[MODE == MODE_COUNT ? 1 : T];
	__shared__ uint64_t s_sort[MODE == MODE_STORE ? T / 2 : 1];
	__shared__ uint32_t epref[NT + 1];
	__shared__ uint32_t estart[NT];
	__shared__ double eaval[NT];
	__shared__ uint32_t scr32[NT / 64 + 1];
	__shared__ unsigned long long s_u64[8];
	__shared__ double s_f64[4];

	const uint32_t r = binrows[blockIdx.x];
	const uint32_t beg = m.beg[r], end = m.beg[r + 1];
	const int32_t rowid = m.id[r];
	const unsigned tid = threadIdx.x;

	for (int q = tid; q < T; q += NT) { h_key[q] = -1; if (MODE != MODE_COUNT) h_val[q] = 0.0; }
	__syncthreads();

	for (uint32_t chunk = beg; chunk < end; chunk += NT) {
		uint32_t e = chunk + tid;
		uint32_t lo = 0, len = 0; double a = 0;
		if (e < end) { int32_t k = m.acol[e]; lo = m.bptr[k]; len = m.bptr[k + 1] - lo; a = m.aval[e]; }
		uint32_t total;
		uint32_t ex = block_exclusive_scan<uint32_t, NT>(len, scr32, &total);
		epref[tid] = ex; estart[tid] = lo; eaval[tid] = a;
		if (tid == 0) epref[NT] = total;
		__syncthreads();
		for (uint32_t p = tid; p < total; p += NT) {
			uint32_t q = find_entry<NT>(epref, p);
			uint32_t bp = estart[q] + (p - epref[q]);
			int32_t col = m.bcol[bp];
			uint32_t h = ((uint32_t)col * 0x9E3779B1u) >> (32 - LOGT);
			for (;;) {
				int32_t old = atomicCAS(&h_key[h], -1, col);
				if (old == -1 || old == col) break;
				h = (h + 1) & (T - 1);
			}
			if (MODE != MODE_COUNT) atomicAdd(&h_val[h], eaval[q] * m.bval[bp]);
		}
		__syncthreads();
	}

	const double a_scale = row_scale(ep, rowid);
	if (MODE == MODE_COUNT) {
		// structural count (an upper bound when sums cancel to exactly 0)
		uint32_t c = 0;
		for (int q = tid; q < T; q += NT) { int32_t col = h_key[q]; if (col >= 0 && col_allowed(ep, col)) ++c; }
		uint32_t total;
		block_exclusive_scan<uint32_t, NT>(c, scr32, &total);
		if (tid == 0) sk.segcount[sk.segbase[r]] = total;
	} else if (MODE == MODE_DIGEST) {
		unsigned long long cnt = 0, hash = 0; double vs = 0;
		for (int q = tid; q < T; q += NT) {
			int32_t col = h_key[q];
			double v;
			if (col >= 0 && emit_value(ep, a_scale, col, h_val[q], &v)) { ++cnt; hash += mix64((uint32_t)rowid, (uint32_t)col); vs += v; }
		}
		if (sk.row_nnz) {
			unsigned long long rc = wave_reduce_sum(cnt); double rs = wave_reduce_sum(vs);
			if (lane_id() == 0) { s_u64[wave_id()] = rc; s_f64[wave_id()] = rs; }
			__syncthreads();
			if (tid == 0) {
				unsigned long long c = 0; double s = 0;
				for (int w = 0; w < NT / 64; ++w) { c += s_u64[w]; s += s_f64[w]; }
				sk.row_nnz[rowid] = (long long)c; sk.row_sum[rowid] = s;
			}
			__syncthreads();
		}
		digest_flush<NT>(sk.digest, cnt, hash, vs, s_u64, s_f64);
	} else {
		// compact the surviving (col, slot) pairs, bitonic-sort them by column, emit in order
		uint32_t run = 0;
		for (int base = 0; base < T; base += NT) {
			int q = base + tid;
			int32_t col = h_key[q];
			double v = 0;
			bool ok = col >= 0 && emit_value(ep, a_scale, col, h_val[q], &v);
			if (ok) h_val[q] = v;
			uint32_t total;
			uint32_t ex = block_exclusive_scan<uint32_t, NT>(ok ? 1u : 0u, scr32, &total);
			if (ok) s_sort[run + ex] = ((uint64_t)(uint32_t)col << 16) | (uint64_t)q;
			run += total;
		}
		const uint32_t mcount = run;
		uint32_t n2 = 1;
		while (n2 < mcount) n2 <<= 1;
		for (uint32_t q = mcount + tid; q < n2; q += NT) s_sort[q] = ~0ull;
		__syncthreads();
		for (uint32_t k = 2; k <= n2; k <<= 1) {
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
		uint32_t seg = sk.segbase[r];
		int64_t o = sk.segoff[seg];
		for (uint32_t i = tid; i < mcount; i += NT) {
			uint64_t kq = s_sort[i];
			sk.out_i[o + i] = rowid;
			sk.out_j[o + i] = (int32_t)(kq >> 16);
			sk.out_v[o + i] = h_val[kq & 0xFFFFu];
		}
		if (tid == 0) sk.segactual[seg] = mcount;
	}
}

// ====================================================================== heavy rows (dense column windows in LDS)

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

// One wave per heavy row: first / last window its products can touch.
__global__ __launch_bounds__(64) void k_heavy_span(const uint32_t *hrows, uint32_t nheavy, RowMeta m, uint32_t wshift,
	uint32_t *hwlo, uint32_t *hnwin)
{
	uint32_t h = blockIdx.x;
	uint32_t r = hrows[h];
	uint32_t beg = m.beg[r], end = m.beg[r + 1];
	uint32_t cmin = 0xFFFFFFFFu, cmax = 0;
	for (uint32_t e = beg + threadIdx.x; e < end; e += 64) {
		int32_t k = m.acol[e];
		uint32_t lo = m.bptr[k], hi = m.bptr[k + 1];
		if (hi > lo) { cmin = min(cmin, (uint32_t)m.bcol[lo]); cmax = max(cmax, (uint32_t)m.bcol[hi - 1]); }
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, d, 64));
		cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d, 64));
	}
	if (threadIdx.x == 0) {
		uint32_t wlo = cmin >> wshift, whi = cmax >> wshift;
		hwlo[h] = wlo;
		hnwin[h] = whi - wlo + 1;
	}
}

__global__ void k_heavy_keys(const uint32_t *hrows, const uint32_t *rprod, uint32_t nheavy, uint64_t *keys)
{
	uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h < nheavy) keys[h] = (uint64_t)(0xFFFFFFFFu - rprod[hrows[h]]);     // descending P
}

__global__ void k_gather_u32(const uint32_t *src, const uint32_t *perm, uint32_t n, uint32_t *dst)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[perm[i]];
}

// nseg per non-empty row: 1, or the window count of a heavy row
__global__ void k_nseg_heavy(const uint32_t *hrows, const uint32_t *hnwin, uint32_t nheavy, uint32_t *nseg)
{
	uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h < nheavy) nseg[hrows[h]] = hnwin[h];
}

template <int W, int NT, int MODE>
__global__ __launch_bounds__(NT) void k_heavy(const uint32_t *hrows, const uint32_t *hwlo, const uint32_t *hnwin, uint32_t nheavy,
	uint32_t *ticket, RowMeta m, const uint32_t *bwin, uint32_t nwin1, EmitParams ep, SinkParams sk)
{
	constexpr int NW = NT / 64;
	constexpr int NGRP = W / 64;             // 64-slot groups per window
	constexpr int GPW = NGRP / NW;           // groups per wave
	constexpr uint32_t WSHIFT = W == 8192 ? 13 : 14;
	__shared__ double acc[W];
	__shared__ uint8_t dirty[NGRP];
	__shared__ uint32_t epref[NT + 1];
	__shared__ uint32_t estart[NT];
	__shared__ double eaval[NT];
	__shared__ uint32_t scr32[NW + 1];
	__shared__ uint32_t s_wcnt[NW + 1];
	__shared__ uint32_t s_ticket;
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];

	const unsigned tid = threadIdx.x, lane = lane_id(), wv = wave_id();
	for (int q = tid; q < W; q += NT) acc[q] = 0.0;
	for (int q = tid; q < NGRP; q += NT) dirty[q] = 0;
	unsigned long long d_cnt = 0, d_hash = 0; double d_sum = 0;     // DIGEST, whole launch

	for (;;) {
		__syncthreads();
		if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
		__syncthreads();
		const uint32_t h = s_ticket;
		if (h >= nheavy) break;                                     // every workgroup reaches this
		const uint32_t r = hrows[h];
		const uint32_t beg = m.beg[r], end = m.beg[r + 1];
		const int32_t rowid = m.id[r];
		const double a_scale = row_scale(ep, rowid);
		const uint32_t wlo = hwlo[h], nw = hnwin[h];
		const uint32_t seg0 = (MODE == MODE_DIGEST) ? 0u : sk.segbase[r];
		unsigned long long r_cnt = 0; double r_sum = 0;            // row statistics

		for (uint32_t wi = 0; wi < nw; ++wi) {
			const uint32_t w = wlo + wi;
			const uint32_t wbase = w << WSHIFT;
			bool any = false;
			for (uint32_t chunk = beg; chunk < end; chunk += NT) {
				uint32_t e = chunk + tid;
				uint32_t lo = 0, len = 0; double a = 0;
				if (e < end) {
					int32_t k = m.acol[e];
					const uint32_t *bw = bwin + (uint64_t)k * nwin1 + w;
					lo = bw[0]; len = bw[1] - lo;
					a = m.aval[e];
				}
				uint32_t total;
				uint32_t ex = block_exclusive_scan<uint32_t, NT>(len, scr32, &total);
				if (total == 0) continue;                           // uniform
				any = true;
				epref[tid] = ex; estart[tid] = lo; eaval[tid] = a;
				if (tid == 0) epref[NT] = total;
				__syncthreads();
				for (uint32_t p = tid; p < total; p += NT) {
					uint32_t q = find_entry<NT>(epref, p);
					uint32_t bp = estart[q] + (p - epref[q]);
					uint32_t slot = (uint32_t)m.bcol[bp] - wbase;
					if (MODE == MODE_COUNT) acc[slot] = 1.0;        // structural: touched
					else atomicAdd(&acc[slot], eaval[q] * m.bval[bp]);
					dirty[slot >> 6] = 1;
				}
				__syncthreads();
			}
			if (!any) {
				if (MODE == MODE_COUNT && tid == 0) sk.segcount[seg0 + wi] = 0;
				if (MODE == MODE_STORE && tid == 0) sk.segactual[seg0 + wi] = 0;
				continue;
			}
			// ---- scan-out: wave wv owns groups [wv*GPW, (wv+1)*GPW) -> ascending columns
			double v[GPW];
			uint64_t nzmask[GPW];
			uint32_t wcount = 0;
#pragma unroll
			for (int gi = 0; gi < GPW; ++gi) {
				int grp = wv * GPW + gi;
				v[gi] = 0; nzmask[gi] = 0;
				if (dirty[grp]) {                                   // wave-uniform
					double x = acc[grp * 64 + lane];
					acc[grp * 64 + lane] = 0.0;
					int32_t col = (int32_t)(wbase + grp * 64 + lane);
					bool ok;
					if (MODE == MODE_COUNT) ok = (x != 0) && col_allowed(ep, col);
					else ok = emit_value(ep, a_scale, col, x, &x);
					v[gi] = x;
					nzmask[gi] = __ballot(ok);
					wcount += (uint32_t)__popcll(nzmask[gi]);
				}
			}
			if (MODE == MODE_DIGEST) {
				for (int gi = 0; gi < GPW; ++gi) {
					if ((nzmask[gi] >> lane) & 1ull) {
						int32_t col = (int32_t)(wbase + (wv * GPW + gi) * 64 + lane);
						++d_cnt; d_hash += mix64((uint32_t)rowid, (uint32_t)col); d_sum += v[gi];
						++r_cnt; r_sum += v[gi];
					}
				}
				__syncthreads();                                    // all reads of dirty[] done
				for (int q = tid; q < NGRP; q += NT) dirty[q] = 0;
			} else {
				if (lane == 0) s_wcnt[wv] = wcount;
				__syncthreads();                                    // also: all reads of dirty[] done
				uint32_t wbefore = 0, wtotal = 0;
#pragma unroll
				for (int q = 0; q < NW; ++q) { uint32_t t = s_wcnt[q]; if (q < (int)wv) wbefore += t; wtotal += t; }
				for (int q = tid; q < NGRP; q += NT) dirty[q] = 0;
				if (MODE == MODE_COUNT) {
					if (tid == 0) sk.segcount[seg0 + wi] = wtotal;
				} else {
					int64_t o = sk.segoff[seg0 + wi] + wbefore;
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
					if (tid == 0) sk.segactual[seg0 + wi] = wtotal;
				}
			}
		}
		if (MODE == MODE_DIGEST && sk.row_nnz) {
			unsigned long long rc = wave_reduce_sum(r_cnt); double rs = wave_reduce_sum(r_sum);
			__syncthreads();
			if (lane == 0) { s_u64[wv] = rc; s_f64[wv] = rs; }
			__syncthreads();
			if (tid == 0) {
				unsigned long long c = 0; double s = 0;
				for (int q = 0; q < NW; ++q) { c += s_u64[q]; s += s_f64[q]; }
				sk.row_nnz[rowid] = (long long)c; sk.row_sum[rowid] = s;
			}
		}
	}
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

// ====================================================================== holes (cancellation in STORE)

__global__ void k_seg_holes(const uint32_t *segcount, const uint32_t *segactual, uint32_t nseg, unsigned long long *holes)
{
	uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
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

__global__ void k_digest_reduce(const DigestSlot *slots, DigestSlot *out)
{
	// one wave; deterministic order of the slot sums
	unsigned long long c = 0, h = 0; double s = 0;
	for (int q = threadIdx.x; q < DIGEST_SLOTS; q += 64) { c += slots[q].count; h += slots[q].hash; s += slots[q].sum; }
	c = wave_reduce_sum(c); h = wave_reduce_sum(h); s = wave_reduce_sum(s);
	if (threadIdx.x == 0) { out->count = c; out->hash = h; out->sum = s; }
}

// ====================================================================== host driver

static unsigned grid_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

struct Bins {
	uint32_t count[NBIN];
	uint32_t off[NBIN + 1];
	uint32_t *rows;
};

template <int MODE>
static void launch_light_mid(spsamd_ctx *c, const Bins &b, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	hipStream_t st = c->stream;
	if (b.count[1]) { k_light<8, MODE><<<dim3(grid_for(b.count[1], 32)), dim3(256), 0, st>>>(b.rows + b.off[1], b.count[1], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[2]) { k_light<16, MODE><<<dim3(grid_for(b.count[2], 16)), dim3(256), 0, st>>>(b.rows + b.off[2], b.count[2], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[3]) { k_light<32, MODE><<<dim3(grid_for(b.count[3], 8)), dim3(256), 0, st>>>(b.rows + b.off[3], b.count[3], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[4]) { k_light<64, MODE><<<dim3(grid_for(b.count[4], 4)), dim3(256), 0, st>>>(b.rows + b.off[4], b.count[4], m, ep, sk); SPS_LAUNCH_CHECK(); }
}

template <int MODE>
static void launch_mid(spsamd_ctx *c, const Bins &b, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	hipStream_t st = c->stream;
	if (b.count[5]) { k_mid<1024, MODE><<<dim3(b.count[5]), dim3(256), 0, st>>>(b.rows + b.off[5], b.count[5], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[6]) { k_mid<4096, MODE><<<dim3(b.count[6]), dim3(256), 0, st>>>(b.rows + b.off[6], b.count[6], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[7]) { k_mid<8192, MODE><<<dim3(b.count[7]), dim3(256), 0, st>>>(b.rows + b.off[7], b.count[7], m, ep, sk); SPS_LAUNCH_CHECK(); }
}

struct Heavy {
	uint32_t n = 0;
	uint32_t *rows = nullptr, *wlo = nullptr, *nwin = nullptr;
	uint32_t *bwin = nullptr;
	uint32_t nwin1 = 0;
	uint32_t *ticket = nullptr;
	int W = 8192;
};

template <int MODE>
static void launch_heavy(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	if (!hv.n) return;
	fill_zero(c, hv.ticket, sizeof(uint32_t));
	if (hv.W == 8192) {
		unsigned grid = std::min<unsigned>(hv.n, (unsigned)c->num_cu * 2u);
		k_heavy<8192, 512, MODE><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.rows, hv.wlo, hv.nwin, hv.n, hv.ticket, m, hv.bwin, hv.nwin1, ep, sk);
	} else {
		unsigned grid = std::min<unsigned>(hv.n, (unsigned)c->num_cu);
		k_heavy<16384, 1024, MODE><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.rows, hv.wlo, hv.nwin, hv.n, hv.ticket, m, hv.bwin, hv.nwin1, ep, sk);
	}
	SPS_LAUNCH_CHECK();
}

static float elapsed(hipEvent_t a, hipEvent_t b)
{
	float ms = 0;
	SPS_HIP(hipEventElapsedTime(&ms, a, b));
	return ms;
}

void spgemm(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res)
{
	hipStream_t st = c->stream;
	const ConMat &A = a.A, &B = a.B;
	res->nnz_a = A.nnz; res->nnz_b = B.nnz;
	if (A.nnz == 0 || B.nnz == 0) return;           // empty product (also SURVEY Appendix A.3)

	SPS_HIP(hipEventRecord(c->ev[1], st));
	// ---- row structure of A (dim_beginnings) and dense row pointer of B (+ sentinel row for scalej)
	RowList rl;
	dim_beginnings(c, A, &rl);
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

	// ---- symbolic: products per A tuple, per row, bins
	uint32_t *elen = c->arena.get<uint32_t>(A.nnz);
	int64_t *pref = c->arena.get<int64_t>((size_t)A.nnz + 1);
	k_elem_len<<<dim3(grid_for(A.nnz)), dim3(256), 0, st>>>(acol, bptr, A.nnz, elen);
	SPS_LAUNCH_CHECK();
	scan_exclusive_u32_i64(c, elen, pref, A.nnz);
	uint32_t *rprod = c->arena.get<uint32_t>(rl.nrows);
	uint8_t *rbin = c->arena.get<uint8_t>(rl.nrows);
	BinCounters *bc = c->arena.get<BinCounters>(1);
	fill_zero(c, bc, sizeof(BinCounters));
	k_classify<<<dim3(grid_for(rl.nrows)), dim3(256), 0, st>>>(rl.beg, rl.id, rl.nrows, pref,
		a.si.present ? a.si.pos : nullptr, a.si.val, rprod, rbin, bc);
	SPS_LAUNCH_CHECK();
	BinCounters hbc = read_back(c, bc);

	Bins bins;
	BinOffsets bo;
	uint32_t run = 0;
	for (int b = 0; b < NBIN; ++b) {
		bins.count[b] = b == 0 ? 0 : (uint32_t)hbc.rows[b];
		bins.off[b] = bo.off[b] = run;
		run += bins.count[b];
	}
	bins.off[NBIN] = bo.off[NBIN] = run;
	bins.rows = c->arena.get<uint32_t>(run ? run : 1);
	uint32_t *cursor = c->arena.get<uint32_t>(NBIN);
	fill_zero(c, cursor, NBIN * sizeof(uint32_t));
	k_bin_scatter<<<dim3(grid_for(rl.nrows)), dim3(256), 0, st>>>(rbin, rl.nrows, bo, cursor, bins.rows);
	SPS_LAUNCH_CHECK();

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

	RowMeta m{rl.beg, rl.id, acol, aval, bptr, B.col, B.val};
	EmitParams ep{a.C, a.si.present ? a.si.pos : nullptr, a.si.val, a.sk.present ? a.sk.pos : nullptr, a.sk.val};

	// ---- heavy rows: window index of B, window span per row, heaviest first
	Heavy hv;
	hv.n = bins.count[8];
	if (hv.n) {
		hv.W = B.ncol > (uint64_t(1) << 21) ? 16384 : 8192;
		const uint32_t wshift = hv.W == 8192 ? 13 : 14;
		const uint32_t nwin = (uint32_t)((B.ncol + hv.W - 1) >> wshift);
		hv.nwin1 = nwin + 1;
		const uint64_t nrowb = B.nrow + extra;
		hv.bwin = c->arena.get<uint32_t>(nrowb * hv.nwin1);
		k_bwin_prefill<<<dim3(4096), dim3(256), 0, st>>>(bptr, nrowb, hv.nwin1, hv.bwin);
		SPS_LAUNCH_CHECK();
		k_bwin_fill<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.row, B.col, bptr, B.nnz, wshift, hv.nwin1, hv.bwin);
		SPS_LAUNCH_CHECK();
		// sort the heavy rows by descending product count so the longest rows start first
		uint32_t *hraw = bins.rows + bins.off[8];
		uint64_t *k0 = c->arena.get<uint64_t>(hv.n), *k1 = c->arena.get<uint64_t>(hv.n);
		uint32_t *p0 = c->arena.get<uint32_t>(hv.n), *p1 = c->arena.get<uint32_t>(hv.n);
		k_heavy_keys<<<dim3(grid_for(hv.n)), dim3(256), 0, st>>>(hraw, rprod, hv.n, k0);
		SPS_LAUNCH_CHECK();
		int where = radix_sort_pairs(c, k0, p0, k1, p1, hv.n, 32);
		hv.rows = c->arena.get<uint32_t>(hv.n);
		k_gather_u32<<<dim3(grid_for(hv.n)), dim3(256), 0, st>>>(hraw, where ? p1 : p0, hv.n, hv.rows);
		SPS_LAUNCH_CHECK();
		hv.wlo = c->arena.get<uint32_t>(hv.n);
		hv.nwin = c->arena.get<uint32_t>(hv.n);
		k_heavy_span<<<dim3(hv.n), dim3(64), 0, st>>>(hv.rows, hv.n, m, wshift, hv.wlo, hv.nwin);
		SPS_LAUNCH_CHECK();
		hv.ticket = c->arena.get<uint32_t>(1);
	}
	SPS_HIP(hipEventRecord(c->ev[2], st));

	// ---- numeric
	SinkParams sk{};
	float ms_light = 0, ms_mid = 0, ms_heavy = 0;
	if (a.sink_kind == SPSAMD_SINK_DIGEST) {
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
		launch_light_mid<MODE_DIGEST>(c, bins, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[4], st));
		launch_mid<MODE_DIGEST>(c, bins, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[5], st));
		launch_heavy<MODE_DIGEST>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[6], st));
		k_digest_reduce<<<dim3(1), dim3(64), 0, st>>>(slots, slots + DIGEST_SLOTS);
		SPS_LAUNCH_CHECK();
		DigestSlot d = read_back(c, slots + DIGEST_SLOTS);
		res->nnz = d.count; res->hash = d.hash; res->sum = d.sum;
		ms_light = elapsed(c->ev[3], c->ev[4]); ms_mid = elapsed(c->ev[4], c->ev[5]); ms_heavy = elapsed(c->ev[5], c->ev[6]);
	} else {
		// segments: one per row, one per (heavy row, window); reserve, scan, store
		uint32_t *nseg = c->arena.get<uint32_t>(rl.nrows);
		fill_u32(c, nseg, 1u, rl.nrows);
		if (hv.n) {
			k_nseg_heavy<<<dim3(grid_for(hv.n)), dim3(256), 0, st>>>(hv.rows, hv.nwin, hv.n, nseg);
			SPS_LAUNCH_CHECK();
		}
		int64_t *segbase64 = c->arena.get<int64_t>((size_t)rl.nrows + 1);
		scan_exclusive_u32_i64(c, nseg, segbase64, rl.nrows);
		int64_t nsegs = read_back(c, segbase64 + rl.nrows);
		if (nsegs >= (int64_t(1) << 32)) throw Error{SPSAMD_EINVAL, "too many output segments"};
		uint32_t *segbase = c->arena.get<uint32_t>((size_t)rl.nrows + 1);
		scan_exclusive_u32_u32(c, nseg, segbase, rl.nrows);
		uint32_t *segcount = c->arena.get<uint32_t>((size_t)nsegs);
		uint32_t *segactual = c->arena.get<uint32_t>((size_t)nsegs);
		int64_t *segoff = c->arena.get<int64_t>((size_t)nsegs + 1);
		fill_zero(c, segcount, (size_t)nsegs * sizeof(uint32_t));
		fill_zero(c, segactual, (size_t)nsegs * sizeof(uint32_t));
		sk.segbase = segbase; sk.segcount = segcount; sk.segoff = segoff; sk.segactual = segactual;

		SPS_HIP(hipEventRecord(c->ev[3], st));
		launch_light_mid<MODE_COUNT>(c, bins, m, ep, sk);
		launch_mid<MODE_COUNT>(c, bins, m, ep, sk);
		launch_heavy<MODE_COUNT>(c, hv, m, ep, sk);
		scan_exclusive_u32_i64(c, segcount, segoff, (size_t)nsegs);
		int64_t reserved = read_back(c, segoff + nsegs);
		c->out_i.ensure((size_t)reserved * sizeof(int32_t));
		c->out_j.ensure((size_t)reserved * sizeof(int32_t));
		c->out_v.ensure((size_t)reserved * sizeof(double));
		sk.out_i = (int32_t *)c->out_i.p; sk.out_j = (int32_t *)c->out_j.p; sk.out_v = (double *)c->out_v.p;
		SPS_HIP(hipEventRecord(c->ev[4], st));
		launch_light_mid<MODE_STORE>(c, bins, m, ep, sk);
		launch_mid<MODE_STORE>(c, bins, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[5], st));
		launch_heavy<MODE_STORE>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[6], st));
		unsigned long long *holes = c->arena.get<unsigned long long>(1);
		fill_zero(c, holes, sizeof(unsigned long long));
		k_seg_holes<<<dim3(grid_for((size_t)nsegs)), dim3(256), 0, st>>>(segcount, segactual, (uint32_t)nsegs, holes);
		SPS_LAUNCH_CHECK();
		unsigned long long nholes = read_back(c, holes);
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
		ms_light = elapsed(c->ev[4], c->ev[5]); ms_heavy = elapsed(c->ev[5], c->ev[6]);
	}
	SPS_HIP(hipEventRecord(c->ev[7], st));
	SPS_HIP(hipEventSynchronize(c->ev[7]));
	res->ms_symbolic = elapsed(c->ev[1], c->ev[2]);
	res->ms_numeric = elapsed(c->ev[2], c->ev[7]);
	res->ms_light = ms_light; res->ms_mid = ms_mid; res->ms_heavy = ms_heavy;
}

} // namespace spsamd
