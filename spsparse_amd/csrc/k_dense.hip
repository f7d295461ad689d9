// k_dense.hip -- dense cells: one output row x ONE column window, f64 accumulator of W columns in LDS.
#include "spgemm_host.h"

namespace spsamd {

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
// The COUNT launch (first pass of the COO sink) needs no sums: its "accumulator" is one BYTE per column of the window (8 KB
// instead of 64 KB, a scan-out of W / 16 16-byte words instead of W slots), set with plain byte stores -- every writer
// stores the same 1 (29.0 -> 28.0 ms on cfg2; the launch is bound by the B reads, not by the accumulator).
template <int W, int NT, int MODE, bool PAT>
__global__ __launch_bounds__(NT, 4) void k_dense(const Cell *cells, uint32_t ncell, const uint32_t *xb, RowMeta m,
	const uint32_t *widx, uint64_t kstride, uint64_t wstride, uint32_t narrow, EmitParams ep, SinkParams sk, uint32_t *claim_ctr)
{
	constexpr int NW = NT / 64;
	constexpr int NGRP = W / 64;             // 64-slot groups per window
	constexpr int GPW = NGRP / NW;           // groups per wave
	constexpr uint32_t WSHIFT = W == 8192 ? 13 : 14;
	constexpr int R = DENSE_R;
	constexpr int NWORD = W / 64;            // bitmap words of one batch (W items)
	constexpr int WPL = NWORD / 64;          // words per lane of the per-wave copy
	__shared__ double acc[MODE == MODE_COUNT ? 1 : W + 64];      // + one dump slot per lane: tuples past the end of a segment's last item land there
	__shared__ __attribute__((aligned(16))) uint8_t ctouch[MODE == MODE_COUNT ? W : 16];     // COUNT: the touched columns
	__shared__ uint32_t s_cpref[NT + 1];     // compacted segments: exclusive ITEM prefix (+ total)
	__shared__ uint2 s_cse[NT];              // first tuple of the segment, one past its last
	__shared__ double s_caval[NT];           // the A value
	__shared__ unsigned long long s_bmask[NWORD];
	__shared__ uint32_t s_scrL[2][NW], s_scrN[2][NW];
	__shared__ uint32_t s_wcnt[NW + 1];
	__shared__ PatCell s_pat;
	__shared__ unsigned long long s_u64[2 * NW];
	__shared__ double s_f64[NW];
	__shared__ uint32_t s_claim[4];

	const unsigned tid = threadIdx.x, lane = lane_id();
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
	// EXACT_PATTERN: a clean slot holds -0.0.  No sum of products is -0.0 (x + -x = +0, and a product is never a zero: zeros
	// are dropped at consolidation), -0.0 + p = p exactly, and -0.0 is "not emitted" like +0 -- so the scan-out can tell a slot
	// no product touched from one whose terms cancelled, and re-evaluates only the latter.  (With +0 as the clean value every
	// EMPTY slot of a cell with products of both signs was re-evaluated from the operands: 4.8 s instead of 11 ms on a
	// scale-18 R-MAT with random signs.)
	const double CLEAN = (PAT && MODE != MODE_COUNT) ? -0.0 : 0.0;
	if (MODE == MODE_COUNT) { for (int q = tid; q < W / 16; q += NT) reinterpret_cast<uint4 *>(ctouch)[q] = make_uint4(0, 0, 0, 0); }
	else for (int q = tid; q < W + 64; q += NT) acc[q] = CLEAN;
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
	// Two ways to deal the list.  STATIC: a grid stride (over the whole list, or over the XCD group's part of it).
	// CLAIMED (claim_ctr): the list is cut into eight parts of equal estimated cost, one per XCD group (blockIdx % 8 names
	// the workgroups that share an XCD and its L2: each L2 then holds the B slice of ITS windows only); a workgroup claims
	// its next cell from its part's counter, and from the other parts' once its own is used up -- the static form of the
	// same partition lost more to the parts' unequal run times than the L2 hits returned.  Claims are made three cells
	// ahead (the atomic's answer arrives under the scan-out) and handed to the other waves through LDS.
	constexpr uint32_t NONE = 0xFFFFFFFFu;
	const bool claimed = claim_ctr != nullptr && xb != nullptr && !ABL(ep, 16);       // (the no-scan-out ablation skips the end of the cell, where claims are published)
	const CellWalk walk = cell_walk(claimed ? nullptr : xb, ncell);
	const uint32_t stride = walk.stride, cend = walk.end;
	const uint32_t clast = ncell ? ncell - 1 : 0;
	uint32_t part = blockIdx.x & 7u, tried = 0;                     // (thread 0's claiming state)
	auto claim_resolve = [&](uint32_t got) -> uint32_t {            // thread 0: a claim's answer -> a cell, or NONE
		while (true) {
			if (got < xb[part + 1]) return got;
			if (++tried >= 8u) return NONE;
			part = (part + 1u) & 7u;
			got = atomicAdd(&claim_ctr[part * 32u], 1u);
		}
	};
	uint32_t i0, i1, i2;                                            // this cell, the next, the one after (uniform)
	if (claimed) {
		if (tid == 0) {
			for (int q = 0; q < 3; ++q) s_claim[q] = tried >= 8u ? NONE : claim_resolve(atomicAdd(&claim_ctr[part * 32u], 1u));
			s_claim[3] = NONE;
		}
		__syncthreads();
		i0 = s_claim[0]; i1 = s_claim[1]; i2 = NONE;                 // (the cell after the next is read from s_claim[2 + (iter & 1)] behind the cell's first barrier)
	} else {
		i0 = walk.first < cend ? walk.first : NONE;
		i1 = i0 != NONE && i0 + stride < cend ? i0 + stride : NONE;
		i2 = i1 != NONE && i1 + stride < cend ? i1 + stride : NONE;
	}
	i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0); i1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i1); i2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i2);
	// (the records read ahead stay in flight in vector registers: CellPend, spgemm_dev.h)
	const uint32_t zlane = cell_pend_zero();
	Cell rec1 = cells[min(i0, clast)];
	CellPend pend2 = cell_pend_load(cells, min(i1, clast), zlane);
	uint32_t nlo, nlen; double na;
	auto seg_bounds = [&](int32_t k, uint32_t w, uint32_t &lo, uint32_t &hi) {
		// segment of B row k in window w: [widx[k*kstride + w*wstride], widx[.. + 1]) -- the row-major index
		// bwin (kstride = nwin+1, wstride = 1) or the window-major row pointer wptr (kstride = 1, wstride = nrowb)
		const uint32_t *bw = widx + (uint64_t)(uint32_t)k * kstride + (uint64_t)w * wstride;
		lo = bw[0]; hi = bw[1];
	};
	{
		const uint32_t e = rec1.beg + tid;
		const bool act = i0 != NONE && e < rec1.end;
		const uint32_t ec = e < rec1.end ? e : rec1.beg;
		uint32_t lo, hi;
		seg_bounds(m.acol[ec], rec1.wa, lo, hi);
		na = m.aval[ec];
		nlo = lo; nlen = act ? hi - lo : 0u;
	}
	int32_t nkA; double naA;                                        // the A tuple of the next cell's chunk 1 (a tuple past the row's end re-reads its first)
	{
		const uint32_t e1 = rec1.beg + NT + tid;
		const uint32_t e1c = e1 < rec1.end ? e1 : rec1.beg;
		nkA = m.acol[e1c]; naA = m.aval[e1c];
	}
	__syncthreads();
	uint32_t iter = 0;
	for (; i0 != NONE; ++iter, i0 = i1, i1 = i2, i2 = (claimed || i2 == NONE || i2 + stride >= cend) ? (claimed ? i2 : NONE) : i2 + stride) {
		const Cell cell = rec1;
		const uint32_t w = cell.wa;
		const uint32_t beg = cell.beg, end = cell.end;
		const int32_t rowid = cell.rowid;
		const double a_scale = row_scale(ep, rowid);
		const uint32_t wbase = w << WSHIFT;
		uint32_t lo = nlo, len = nlen; double a = na;               // chunk 0, prefetched
		rec1 = cell_from_pend(pend2);
		if (!claimed) pend2 = cell_pend_load(cells, min(i2, clast), zlane);                         // (claimed: once this cell's first barrier has published i2's successor ... see below)
		const uint32_t ne = rec1.beg + tid;
		const bool nact = i1 != NONE && ne < rec1.end;
		const uint32_t nec = ne < rec1.end ? ne : rec1.beg;
		const int32_t nk = m.acol[nec];
		na = m.aval[nec];
		// in-cell prefetch, stage A: the A tuple of chunk 1 was requested a cell ago, with the cell's first chunk (its k is
		// needed at the top of chunk 0 for the segment bounds of chunk 1: requested here, the wait for it would come right
		// behind this cell's other prefetches and drain them all); the next cell's is requested now.  Branch-free like every
		// prefetch here: a load inside a conditional -- even a uniform one -- is waited for at the join, i.e. at once
		int32_t kA = nkA; double aA = naA;
		{
			const uint32_t e1 = rec1.beg + NT + tid;
			const uint32_t e1c = e1 < rec1.end ? e1 : rec1.beg;
			nkA = m.acol[e1c]; naA = m.aval[e1c];
		}

		STAMP_COUNT(8);
		uint32_t pnseg = 0;                                         // non-empty segments of the cell (EXACT_PATTERN)
		for (uint32_t chunk = beg; chunk < end; chunk += NT) {
			STAMP_COUNT(9);
			// stage B for chunk c+1 (its k arrived during chunk c-1), stage A for chunk c+2
			uint32_t lo2, hi2; double a2;
			{
				seg_bounds(kA, w, lo2, hi2);
				a2 = aA;
				if (chunk + NT + tid >= end) hi2 = lo2;             // (also: no next chunk at all)
				const uint32_t e2 = chunk + 2 * NT + tid;
				const uint32_t e2c = e2 < end ? e2 : beg;
				kA = m.acol[e2c]; aA = m.aval[e2c];
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
			if (claimed && chunk == beg) {                          // uniform: the cell after the next, claimed during the previous cell
				i2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_claim[2 + (iter & 1u)]);
				pend2 = cell_pend_load(cells, min(i2, clast), zlane);
			}
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
				if (myfirst < (uint32_t)W) atomicOr(reinterpret_cast<uint32_t *>(s_bmask) + (myfirst >> 5), 1u << (myfirst & 31u));     // (32-bit halves: half the bank traffic)
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
						if (ABL(ep, 32)) { if (piece.w[3 * u + 2] == 0x7FF12345u) acc[MODE == MODE_COUNT ? 0 : slot] = av_; }          // no LDS accumulate
						else if (MODE == MODE_COUNT) { if ((uint32_t)u < nv_) ctouch[slot] = (uint8_t)1; }      // structural: touched
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
				// accumulated, the lookup of step s+2 runs under them.  (Not the default: no faster -- nor is the ping-pong form,
				// two named register sets and the loop unrolled by two so that no piece is copied, every load and lookup outside
				// conditionals so that the accumulate waits with vmcnt(3): 28.9 against 28.0 ms -- the L1 fill path is the bound, not the latency.)
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
		uint32_t claim_pending = NONE;
		if (claimed && tid == 0 && tried < 8u) claim_pending = atomicAdd(&claim_ctr[part * 32u], 1u);     // (answered under the scan-out)
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
		unsigned long long r_hash = 0;
		if constexpr (MODE == MODE_COUNT) {
			// the touched columns of the window (those scalek allows): sixteen flag bytes per thread, cleaned on the way
			static_assert(W / 16 == NT, "sixteen flag bytes per thread");
			uint32_t cnt = 0;
			{
				uint4 *cw = reinterpret_cast<uint4 *>(ctouch);
				const uint4 x = cw[tid];
				const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
				if (x.x | x.y | x.z | x.w) {
					cw[tid] = make_uint4(0, 0, 0, 0);
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						if (!ep.sk_pos) cnt += (uint32_t)__popc(xs[q]);
						else {
#pragma unroll
							for (int b = 0; b < 4; ++b)
								if (((xs[q] >> (8 * b)) & 1u) && col_allowed(ep, (int32_t)(wbase + tid * 16u + (uint32_t)(4 * q + b)))) ++cnt;
						}
					}
				}
			}
			wcount = (uint32_t)wave_reduce_sum((unsigned long long)cnt);
		} else {
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
			if (plain) ok = x != 0;
			else ok = emit_value(ep, a_scale, (int32_t)(wbase + grp * 64 + lane), x, &x);
			v[gi] = x;
			nzmask[gi] = __ballot(ok);
			wcount += (uint32_t)__popcll(nzmask[gi]);
			if (MODE == MODE_DIGEST) {
				unsigned long long h = X0 + laneK;
				h ^= h >> 29;
				r_hash += ok ? h : 0ull;
				if (plain && !PAT) r_sum += x;                          // (a slot that is not emitted holds +-0)
				else r_sum += ok ? x : 0.0;
				X0 += 64ull * MIXK;
			}
		}
		}
		if (MODE == MODE_DIGEST) {
			if (lane == 0) d_cnt += wcount;
			d_sum += r_sum;
			d_hash += r_hash;
			if (sk.row_nnz) {
				double rs = wave_reduce_sum(r_sum);
				r_hash = wave_reduce_sum(r_hash);
				if (lane == 0 && wcount) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], (unsigned long long)wcount); atomicAdd(&sk.row_sum[rowid], rs); atomicAdd(&sk.row_hash[rowid], r_hash); }
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
		if (claimed && tid == 0) s_claim[2 + ((iter + 1u) & 1u)] = tried < 8u ? claim_resolve(claim_pending) : NONE;
	}
#ifdef SPSAMD_STAMPS
	if (tid == 0 && sk.stamps) for (int i = 0; i < 12; ++i) sk.stamps[(size_t)blockIdx.x * 12 + i] = st_[i];
#endif
	if (MODE == MODE_DIGEST) digest_flush<NT>(sk.digest, d_cnt, d_hash, d_sum, s_u64, s_f64);
}

__global__ void k_claim_init(const uint32_t *xb, uint32_t *ctr)
{
	if (threadIdx.x < 8) ctr[threadIdx.x * 32u] = xb[threadIdx.x];
}

template <int MODE>
void launch_heavy_dense(spsamd_ctx *c, const Heavy &hv, const RowMeta &m0, const EmitParams &ep, const SinkParams &sk)
{
	if (!hv.ncell[CLS_DENSE]) return;
	uint32_t *claim = nullptr;
	if (c->tune.xcd == 2 && hv.xb[CLS_DENSE]) {                     // the list in eight parts, cells claimed from per-part counters
		claim = c->arena.get<uint32_t>(8 * 32);
		k_claim_init<<<dim3(1), dim3(64), 0, c->stream>>>(hv.xb[CLS_DENSE], claim);
		SPS_LAUNCH_CHECK();
	}
	RowMeta m = m0;
	const uint32_t *widx = hv.bwin;
	uint64_t kstride = hv.nwin1, wstride = 1;
	if (hv.wptr) { widx = hv.wptr; kstride = 1; wstride = hv.nrowb; m.btup = hv.btw; }
	const uint32_t narrow = ((uint64_t)hv.nnzb + DENSE_R) * 12u < (uint64_t(1) << 32) ? 1u : 0u;    // 32-bit byte offsets into B suffice
	if (hv.W == 8192) {
		unsigned grid = std::min<unsigned>(hv.ncell[CLS_DENSE], (unsigned)c->num_cu * (MODE == MODE_COUNT ? 3u : 2u));
		// (COUNT's 20 KB of LDS allow more than two workgroups per CU.  One list for all XCDs: 28.0 / 31.2 / 33.2 ms at two / three / four -- more
		// cells in flight, fewer L2 hits; with the claimed XCD parts: 23.5 / 20.8 / 20.9)
		if (grid >= 64) grid &= ~7u;
#ifdef SPSAMD_STAMPS
		SinkParams sk2 = sk;
		sk2.stamps = c->arena.get<unsigned long long>((size_t)grid * 12);
		fill_zero(c, sk2.stamps, (size_t)grid * 12 * sizeof(unsigned long long));
		k_dense<8192, 512, MODE, false><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk2, claim);
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
		if (ep.pattern) k_dense<8192, 512, MODE, true><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk, claim);
		else k_dense<8192, 512, MODE, false><<<dim3(grid), dim3(512), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk, claim);
	} else {
		unsigned grid = std::min<unsigned>(hv.ncell[CLS_DENSE], (unsigned)c->num_cu);
		if (grid >= 64) grid &= ~7u;
		if (ep.pattern) k_dense<16384, 1024, MODE, true><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk, claim);
		else k_dense<16384, 1024, MODE, false><<<dim3(grid), dim3(1024), 0, c->stream>>>(hv.cells[CLS_DENSE], hv.ncell[CLS_DENSE], hv.xb[CLS_DENSE], m, widx, kstride, wstride, narrow, ep, sk, claim);
	}
	SPS_LAUNCH_CHECK();
}

// Window-major copy of B (see k_wm_counts): built once the cell grouping has shown that dense cells exist.
template void launch_heavy_dense<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_heavy_dense<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_heavy_dense<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);

} // namespace spsamd
