// k_tiles.hip -- tiles of the second generation: hash tiles v2, bitmap-rank tiles, direct tiles (spgemm_dev.h has the
// shared expansion: TileX, tile_expand, tile_lookup).
#include "spgemm_host.h"
#include "spgemm_hash.h"

namespace spsamd {

// ---- hash tiles: cells are ranges [wa, wb) of sparse column windows, accumulated in the LDS hash table ----
// Insertion: the R first probes of a lane are in flight together (ds_cmpswap with return), the rare collisions are
// then walked one by one; the values follow with ds_add_f64; the newly occupied slots of a step are appended to the
// occupied list with one LDS fetch-add per wave.  The first block of the NEXT cell is looked up and its B tuples
// requested before the current cell is emitted, so that latency is hidden behind the emission.

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
				unsigned long long cnt = 0, rh = 0; double vs = 0;
				auto note = [&](uint32_t rel, double v) {
					const int32_t col = (int32_t)(colbase + rel);
					const bool ok = plain ? v != 0 : emit_value(ep, a_scale, col, v, &v);
					if (ok) { ++cnt; rh += mix64((uint32_t)rowid, (uint32_t)col); vs += v; }
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
				d_cnt += cnt; d_sum += vs; d_hash += rh;
				if (sk.row_nnz) {
					const unsigned long long rc = wave_reduce_sum(cnt); const double rs = wave_reduce_sum(vs);
					rh = wave_reduce_sum(rh);
					if (lane == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); atomicAdd(&sk.row_hash[rowid], rh); }
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
				// rank: the tuples are stored there at once.  Only where a sum cancelled to exactly 0 (or scalek drops a
				// column) the dropped ones are marked (column -1) and the survivors moved down afterwards, every thread
				// re-reading what it stored itself
				const int64_t o = sk.segoff[seg];
				uint32_t nbad = 0, run = 0;
				int any_bad;
				if (!ep.sk_pos) {                                           // uniform: the cell's reservation is `distinct` places
					for (uint32_t base = 0; base < distinct; base += NT) {      // (uniform trips)
						const uint32_t i = base + tid;
						const bool valid = i < distinct;
						const uint32_t rel = valid ? colof[i] : 0u;
						double v = 1.0;
						if (valid) { v = acc[i]; acc[i] = 0.0; }
						if (PAT) v = pat_fix_wave(valid && !(fabs(v) > pthr), v, (int32_t)(colbase + rel), m, tile.beg, tile.end);
						const bool ok = !valid || (plain ? v != 0 : emit_value(ep, a_scale, (int32_t)(colbase + rel), v, &v));
						if (valid) { sk.out_i[o + i] = rowid; sk.out_j[o + i] = ok ? (int32_t)(colbase + rel) : -1; sk.out_v[o + i] = v; }
						if (!ok) ++nbad;
					}
					any_bad = __syncthreads_or((int)nbad);
					if (any_bad) {                                              // uniform, rare
						for (uint32_t ibase = 0; ibase < distinct; ibase += NT) {
							const uint32_t i = ibase + tid;
							int32_t col = -1; double v = 0;
							if (i < distinct) { col = sk.out_j[o + i]; v = sk.out_v[o + i]; }
							const bool ok = col >= 0;
							uint32_t tot;
							const uint32_t ex = block_exclusive_scan<uint32_t, NT>(ok ? 1u : 0u, s_scan, &tot);      // (its barriers: every place of this trip is read before any is written)
							if (ok) { sk.out_j[o + run + ex] = col; sk.out_v[o + run + ex] = v; }
							run += tot;
						}
					}
				} else {
					// scalek: the reservation holds the ALLOWED columns only -- count the dropped ones first, store compacted
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
					any_bad = __syncthreads_or((int)nbad);
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
			unsigned long long myhash = 0;
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
							if (own) { ++mycount; myhash += mix64((uint32_t)rowid, wbase + slot); mysum += v; }
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
				d_cnt += mycount; d_sum += mysum; d_hash += myhash;
				if (sk.row_nnz) {
					const unsigned long long rc = wave_reduce_sum((unsigned long long)mycount); const double rs = wave_reduce_sum(mysum);
					myhash = wave_reduce_sum(myhash);
					if (lane == 0 && rc) { atomicAdd((unsigned long long *)&sk.row_nnz[rowid], rc); atomicAdd(&sk.row_sum[rowid], rs); atomicAdd(&sk.row_hash[rowid], myhash); }
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

template <int MODE>
void launch_tiles_bm(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
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
}

template <int MODE>
void launch_tiles_hash2(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
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
}

template <int MODE>
void launch_tiles_direct(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
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

template void launch_tiles_bm<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_bm<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_bm<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_hash2<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_hash2<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_hash2<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_direct<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_direct<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_direct<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);

} // namespace spsamd
