// k_hash.hip -- the LDS hash accumulator: k_hash<T,NT> (mid rows; hash cells of heavy rows too long for a tile) and the
// first-generation hash tiles k_hash_tiles (the ORDERED mode runs on them).
#include "spgemm_host.h"
#include "spgemm_hash.h"

namespace spsamd {

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
	const uint32_t zlane = cell_pend_zero();                        // (records read ahead stay in flight in vector registers: CellPend)
	Cell rec1 = cells[min(walk.first, clast)];
	CellPend pend2 = cell_pend_load(cells, min(walk.first + stride, clast), zlane);
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
		rec1 = cell_from_pend(pend2);
		pend2 = cell_pend_load(cells, min(ci + 2 * stride, clast), zlane);
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

template <int MODE>
void launch_mid(spsamd_ctx *c, const Bins &b, const MidCells &mc, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	launch_hash<1024, 256, MODE, false>(c, mc.cells[0], b.count[5], nullptr, m, nullptr, 0, ep, sk);
	launch_hash<4096, 512, MODE, false>(c, mc.cells[1], b.count[6], nullptr, m, nullptr, 0, ep, sk);
	launch_hash<8192, 512, MODE, false>(c, mc.cells[2], b.count[7], nullptr, m, nullptr, 0, ep, sk);   // 115 KB of LDS: one workgroup per CU, so make it 8 waves
}

template <int MODE>
void launch_hash_windowed(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	launch_hash<1024, 256, MODE, true>(c, hv.cells[0], hv.ncell[0], hv.xb[0], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<3072, 512, MODE, true>(c, hv.cells[1], hv.ncell[1], hv.xb[1], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<4096, 512, MODE, true>(c, hv.cells[2], hv.ncell[2], hv.xb[2], m, hv.bwin, hv.nwin1, ep, sk);
	launch_hash<8192, 512, MODE, true>(c, hv.cells[3], hv.ncell[3], hv.xb[3], m, hv.bwin, hv.nwin1, ep, sk);
}

template <int MODE>
void launch_tiles_v1(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
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

template void launch_mid<MODE_COUNT>(spsamd_ctx *, const Bins &, const MidCells &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_mid<MODE_STORE>(spsamd_ctx *, const Bins &, const MidCells &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_mid<MODE_DIGEST>(spsamd_ctx *, const Bins &, const MidCells &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_hash_windowed<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_hash_windowed<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_hash_windowed<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_v1<MODE_COUNT>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_v1<MODE_STORE>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);
template void launch_tiles_v1<MODE_DIGEST>(spsamd_ctx *, const Heavy &, const RowMeta &, const EmitParams &, const SinkParams &);

#ifdef SPSAMD_ABLATIONS
void set_ablation_word(spsamd_ctx *c, int word)
{
	SPS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_abl), &word, sizeof(int), 0, hipMemcpyHostToDevice, c->stream));
	SPS_HIP(hipStreamSynchronize(c->stream));
}
#endif

} // namespace spsamd
