// k_light.hip -- rows with at most 64 scalar products: the binned wave-level kernels (k_light<S>) and the direct
// kernel for products in which EVERY row is light (k_light_direct<S>: the stencil configs).
#include "spgemm_host.h"

namespace spsamd {

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
			unsigned long long rh = hash;
#pragma unroll
			for (int d = S / 2; d >= 1; d >>= 1) { rs += __shfl_xor(rs, d, 64); rh += __shfl_xor(rh, d, 64); }
			if (has_row && s == 0) { sk.row_nnz[rowid] = (long long)__popcll(gmask); sk.row_sum[rowid] = rs; sk.row_hash[rowid] = rh; }
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
			if (sk.segoff) {                                             // offsets known: the second of two passes
				if (has_row) {
					if (out) {
						const uint32_t rk = (uint32_t)__popcll(gmask & ((1ull << s) - 1ull));
						const int64_t o = sk.segoff[r] + rk;
						sk.out_i[o] = rowid; sk.out_j[o] = (int32_t)mycol; sk.out_v[o] = value;
					}
					if (s == 0) sk.segactual[r] = (uint32_t)__popcll(gmask);
				}
			} else {
				// none yet (one compute pass: spgemm_all_light): the tuples of this wave's G rows go, packed in row order, to the
				// wave round's own 64 slots of a sparse buffer -- round q holds rows q G .. q G + G - 1, so the rounds' runs in
				// order ARE the row-major result -- and are gathered once every round's count is known
				const uint32_t q = vb * 4u + w;
				if (out) {
					const int64_t o = (int64_t)q * 64 + (int64_t)__popcll(bal & lanemask_lt());
					sk.out_i[o] = rowid; sk.out_j[o] = (int32_t)mycol; sk.out_v[o] = value;
				}
				if (lane == 0) sk.segactual[q] = (uint32_t)__popcll(bal);
			}
		} else {
			if (sk.row_nnz) {
				double rs = out ? value : 0.0;
				unsigned long long rh = out ? mix64((uint32_t)rowid, mycol) : 0ull;
#pragma unroll
				for (int d = S / 2; d >= 1; d >>= 1) { rs += __shfl_xor(rs, d, 64); rh += __shfl_xor(rh, d, 64); }
				if (has_row && s == 0) { sk.row_nnz[rowid] = (long long)__popcll(gmask); sk.row_sum[rowid] = rs; sk.row_hash[rowid] = rh; }
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

template <int MODE>
void launch_light(spsamd_ctx *c, const Bins &b, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	hipStream_t st = c->stream;
	const unsigned cap = (unsigned)c->num_cu * 8u * 4u;           // 8 resident workgroups per CU, 4 rounds of them
	auto grid_for = [cap](size_t n, unsigned per) { return std::min<unsigned>((unsigned)((n + per - 1) / per), cap); };
	if (b.count[1]) { k_light<8, MODE><<<dim3(grid_for(b.count[1], 32)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[1] : nullptr, b.count[1], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[2]) { k_light<16, MODE><<<dim3(grid_for(b.count[2], 16)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[2] : nullptr, b.count[2], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[3]) { k_light<32, MODE><<<dim3(grid_for(b.count[3], 8)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[3] : nullptr, b.count[3], m, ep, sk); SPS_LAUNCH_CHECK(); }
	if (b.count[4]) { k_light<64, MODE><<<dim3(grid_for(b.count[4], 4)), dim3(256), 0, st>>>(b.rows ? b.rows + b.off[4] : nullptr, b.count[4], m, ep, sk); SPS_LAUNCH_CHECK(); }
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

// The wave rounds' tuples from their 64-slot places in the sparse buffer to their final, packed places: thread = (round, slot).
__global__ __launch_bounds__(256) void k_light_gather(uint32_t nrow, uint32_t logs, const uint32_t *cnt, const int64_t *off,
	const int32_t *si, const int32_t *sj, const double *sv, int32_t *oi, int32_t *oj, double *ov)
{
	const uint64_t total = (uint64_t)nrow << logs, stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
		const uint32_t r = (uint32_t)(t >> logs), s = (uint32_t)t & ((1u << logs) - 1u);
		if (s < cnt[r]) {
			const int64_t o = off[r] + s;
			oi[o] = si[t]; oj[o] = sj[t]; ov[o] = sv[t];
		}
	}
}

void launch_light_gather(spsamd_ctx *c, uint32_t nrow, uint32_t S, const uint32_t *cnt, const int64_t *off,
	const int32_t *si, const int32_t *sj, const double *sv, int32_t *oi, int32_t *oj, double *ov)
{
	const uint32_t logs = S == 8 ? 3 : (S == 16 ? 4 : (S == 32 ? 5 : 6));
	const uint64_t total = (uint64_t)nrow << logs;
	const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)c->num_cu * 64u);
	k_light_gather<<<dim3(grid ? grid : 1), dim3(256), 0, c->stream>>>(nrow, logs, cnt, off, si, sj, sv, oi, oj, ov);
	SPS_LAUNCH_CHECK();
}

template <int MODE>
void launch_light_direct_s(spsamd_ctx *c, uint32_t maxp, uint32_t nrow, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, const ConMat &B, bool k64, const EmitParams &ep, const SinkParams &sk, unsigned long long *pc)
{
	if (maxp <= 8) launch_light_direct<8, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else if (maxp <= 16) launch_light_direct<16, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else if (maxp <= 32) launch_light_direct<32, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
	else launch_light_direct<64, MODE>(c, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
}

// Every output row has at most `maxp` <= 64 products: one kernel, no symbolic phase (see k_light_direct).
template void launch_light<MODE_COUNT>(spsamd_ctx *, const Bins &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_light<MODE_STORE>(spsamd_ctx *, const Bins &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_light<MODE_DIGEST>(spsamd_ctx *, const Bins &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_light_direct_s<MODE_COUNT>(spsamd_ctx *, uint32_t, uint32_t, const uint32_t *, const int32_t *, const double *, const uint32_t *, const ConMat &, bool, const EmitParams &, const SinkParams &, unsigned long long *);
template void launch_light_direct_s<MODE_STORE>(spsamd_ctx *, uint32_t, uint32_t, const uint32_t *, const int32_t *, const double *, const uint32_t *, const ConMat &, bool, const EmitParams &, const SinkParams &, unsigned long long *);
template void launch_light_direct_s<MODE_DIGEST>(spsamd_ctx *, uint32_t, uint32_t, const uint32_t *, const int32_t *, const double *, const uint32_t *, const ConMat &, bool, const EmitParams &, const SinkParams &, unsigned long long *);

} // namespace spsamd
