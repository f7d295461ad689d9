// spgemm_hash.h -- the LDS hash accumulator shared by k_hash, k_hash_tiles (k_hash.hip) and k_hash_tiles2 (k_tiles.hip):
// insertion (hash_products), ordered insertion, emission in column order (hash_emit) and its LDS radix sort.
#pragma once
#include "spgemm_dev.h"

namespace spsamd {

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
		unsigned long long cnt = 0, rh = 0; double vs = 0;
		for (uint32_t base = 0; base < nocc; base += NT) {                 // (uniform trips: pat_fix_wave wants whole waves)
			const uint32_t i = base + tid;
			const bool valid = i < nocc;
			const uint32_t h = valid ? occ[i] : 0u;
			const int32_t col = valid ? h_key[h] : 0;
			double v;
			double x = valid ? h_val[h] : 0.0;
			if (PAT) x = pat_fix_wave(valid && !(fabs(x) > pthr), x, col, m, pbeg, pend);
			if (valid) {
				if (emit_value(ep, a_scale, col, x, &v)) { ++cnt; rh += mix64((uint32_t)rowid, (uint32_t)col); vs += v; }
				h_key[h] = -1; h_val[h] = 0.0;
			}
		}
		d.cnt += cnt; d.sum += vs; d.hash += rh;
		if (sk.row_nnz) {
			unsigned long long rc = wave_reduce_sum(cnt); double rs = wave_reduce_sum(vs);
			rh = wave_reduce_sum(rh);
			if (lane_id() == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); atomicAdd(&sk.row_hash[rowid], rh); }
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

} // namespace spsamd
