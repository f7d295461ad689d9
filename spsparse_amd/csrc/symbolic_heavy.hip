// symbolic_heavy.hip -- the heavy rows' symbolic phase: window index of B, window-major copy, per-row window histograms,
// the grouping of windows into cells and tiles, the sorted cell lists.
#include "spgemm_host.h"

namespace spsamd {

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

// ====================================================================== heavy rows: window index, cells

// Window index of B: bwin[k * (nwin+1) + w] = first tuple of B row k whose column is >= w * W (bwin[k][0] = bptr[k],
// bwin[k][nwin] = bptr[k+1]), and the tuples of row k in window w as 16 bits (<= W <= 16384: half the bytes of the offset
// pairs for the histogram below, which reads one whole row of that table per A tuple of a heavy row; rows padded to a
// multiple of eight entries, nwp, so that eight windows are read as one aligned 16-byte word).
// Built from the TUPLES' side in two balanced passes, nothing searched.  (1) Counts: the tuples are walked 64 per wave; the
// last lane of every run of equal (row, window) adds the run's length to the row's 16-bit count (a pair of counts per 32-bit
// word: one global atomic per run and wave -- the hub rows of an R-MAT hold runs of hundreds).  (2) Index: a workgroup per
// tile of consecutive rows loads the tile's count rows into LDS, a wave per row scans them from the row's pointer and
// writes the index row whole.  (Before: one thread per (row, window) and two binary searches each, 0.55 ms at scale 20 -- a
// third of what a rank of a sharded run spends outside its numeric kernels.  Tried on the way: a wave per row walking its
// tuples, 0.73 ms -- a chain of dependent loads per row; counting in an LDS table per tile of rows, 1.16 ms -- the tile of
// an R-MAT's first 256 rows holds 2.5 M tuples.)
constexpr int BT_NT = 256;
constexpr size_t BT_LDS = 36 * 1024;         // a tile's count rows (rows x nwp x 2 bytes) and row pointers: four workgroups per CU

__global__ __launch_bounds__(BT_NT) void k_wcnt_count(const int32_t *brow, const int32_t *bcol, uint32_t nnzb, uint32_t wshift, uint32_t nwp, uint32_t *wcnt32)
{
	const uint32_t lane = lane_id();
	const uint32_t nchunk = (nnzb + 63u) >> 6, nwaves = gridDim.x * (BT_NT / 64);
	for (uint32_t ch = blockIdx.x * (BT_NT / 64) + wave_id(); ch < nchunk; ch += nwaves) {      // uniform per wave
		const uint32_t e = (ch << 6) + lane;
		const bool valid = e < nnzb;
		const unsigned long long key = valid ? (unsigned long long)(uint32_t)brow[e] * nwp + ((uint32_t)bcol[e] >> wshift) : ~0ull;
		const unsigned long long kprev = __shfl_up(key, 1, 64), knext = __shfl_down(key, 1, 64);
		const bool is_start = valid && (lane == 0 || key != kprev);
		const bool is_end = valid && (lane == 63 || key != knext);
		const unsigned long long sm = __ballot(is_start);
		if (is_end) {
			const unsigned long long upto = sm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
			const uint32_t first = 63u - (uint32_t)__clzll((long long)upto);       // this run's first lane
			atomicAdd(&wcnt32[key >> 1], (lane - first + 1u) << ((uint32_t)(key & 1ull) << 4));    // (nwp is even: the key's parity is the window's)
		}
	}
}

__global__ __launch_bounds__(BT_NT) void k_bwin_scan(const uint16_t *wcnt, const uint32_t *bptr, uint64_t nrowb, uint32_t tr, uint32_t nwin, uint32_t nwp,
	uint32_t *bwin)
{
	extern __shared__ uint32_t s_dyn[];                              // [tr][nwp / 2] packed pairs of counts, then [tr] row pointers
	const uint32_t nwin1 = nwin + 1u, lane = lane_id(), wv = wave_id(), nph = nwp >> 1;
	const uint64_t k0 = (uint64_t)blockIdx.x * tr;
	const uint32_t rows = (uint32_t)std::min<uint64_t>(tr, nrowb - k0);
	uint32_t *s_ptr = s_dyn + (size_t)tr * nph;
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(wcnt + k0 * nwp);       // the tile's rows are contiguous, 16-byte multiples
		uint4 *dst = reinterpret_cast<uint4 *>(s_dyn);
		for (uint32_t i = threadIdx.x; i < (rows * nph) >> 2; i += BT_NT) dst[i] = src[i];
		for (uint32_t r = threadIdx.x; r < rows; r += BT_NT) s_ptr[r] = bptr[k0 + r];
	}
	__syncthreads();
	for (uint32_t r = wv; r < rows; r += BT_NT / 64) {              // a wave per row: lane = window
		const uint32_t *src = s_dyn + r * nph;
		uint32_t carry = s_ptr[r];
		uint32_t *bw = bwin + (k0 + r) * nwin1;
		for (uint32_t c0 = 0; c0 < nwin1; c0 += 64u) {
			const uint32_t w = c0 + lane;
			const uint32_t v = w < nwin ? (src[w >> 1] >> ((w & 1u) << 4)) & 0xFFFFu : 0u;
			const uint32_t inc = wave_inclusive_scan_u32(v);
			if (w < nwin1) bw[w] = carry + inc - v;
			carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
		}
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
//
// wptr[w][k] = (tuples of the windows before w) + (tuples of window w in the rows before k): a prefix sum DOWN THE COLUMNS
// of the row-major count table, written transposed.  Three kernels over tiles of WM_TR rows: column sums per tile; per
// column the scan of the tile sums (and the windows' bases); per tile the column prefixes inside it, through LDS, stored
// along k.  (Before: the table was transposed, 0.38 ms, the 134 M transposed counts were scanned like any array, 0.45 ms.)
constexpr int WM_TR = 256;                   // rows per tile
constexpr int WM_CS = 72;                    // columns of a tile in LDS at a time (x WM_TR x 2 bytes = 36 KB: four workgroups per CU)
constexpr int WM_NT = 256;

__global__ __launch_bounds__(WM_NT) void k_wm_tile_sums(const uint16_t *wcnt, uint64_t nrowb, uint32_t nwp, uint32_t ntile, uint32_t *tilesum)
{
	extern __shared__ uint32_t s_dyn[];                              // nwp column sums
	const uint32_t nq = nwp >> 3;                                    // 16-byte words per row
	for (uint32_t w = threadIdx.x; w < nwp; w += WM_NT) s_dyn[w] = 0;
	__syncthreads();
	const uint64_t k0 = (uint64_t)blockIdx.x * WM_TR;
	const uint32_t rows = (uint32_t)std::min<uint64_t>(WM_TR, nrowb - k0);
	const uint4 *tab = reinterpret_cast<const uint4 *>(wcnt + k0 * nwp);      // rows are 16-byte multiples
	uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t myq = 0xFFFFFFFFu;
	// the tile is rows * nq consecutive 16-byte words; a thread keeps to ONE column group (its word index mod nq is fixed
	// when the stride is a multiple of nq)
	const uint32_t stride = (WM_NT / nq) * nq;                       // (nq <= 256)
	if (threadIdx.x < stride) {
		myq = threadIdx.x % nq;
		for (uint32_t i = threadIdx.x; i < rows * nq; i += stride) {
			const uint4 x = tab[i];
			acc[0] += x.x & 0xFFFFu; acc[1] += x.x >> 16; acc[2] += x.y & 0xFFFFu; acc[3] += x.y >> 16;
			acc[4] += x.z & 0xFFFFu; acc[5] += x.z >> 16; acc[6] += x.w & 0xFFFFu; acc[7] += x.w >> 16;
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) if (acc[u]) atomicAdd(&s_dyn[myq * 8u + u], acc[u]);
	}
	__syncthreads();
	for (uint32_t w = threadIdx.x; w < nwp; w += WM_NT) tilesum[(uint64_t)w * ntile + blockIdx.x] = s_dyn[w];
}

// one workgroup per window: exclusive scan of its tile sums, in place; the window's total
__global__ __launch_bounds__(WM_NT) void k_wm_tile_scan(uint32_t *tilesum, uint32_t ntile, uint32_t *wtot)
{
	__shared__ uint32_t scratch[WM_NT / 64 + 1];
	uint32_t *a = tilesum + (uint64_t)blockIdx.x * ntile;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < ntile; base += WM_NT) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < ntile ? a[i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_exclusive_scan<uint32_t, WM_NT>(v, scratch, &tot);
		if (i < ntile) a[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0) wtot[blockIdx.x] = carry;
}

// the windows' bases (exclusive scan of their totals, one workgroup) and the pointer's end sentinel
__global__ __launch_bounds__(WM_NT) void k_wm_bases(const uint32_t *wtot, uint32_t nwin, uint32_t *wbase, uint32_t *wptr_end)
{
	__shared__ uint32_t scratch[WM_NT / 64 + 1];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nwin; base += WM_NT) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nwin ? wtot[i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_exclusive_scan<uint32_t, WM_NT>(v, scratch, &tot);
		if (i < nwin) wbase[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0) *wptr_end = carry;
}

__global__ __launch_bounds__(WM_NT) void k_wm_ptr(const uint16_t *wcnt, uint64_t nrowb, uint32_t nwin, uint32_t nwp, uint32_t ntile,
	const uint32_t *tileoff, const uint32_t *wbase, uint32_t *wptr)
{
	__shared__ uint16_t s_t[WM_TR * WM_CS];
	const uint32_t tile = blockIdx.x, lane = lane_id(), wv = wave_id();
	const uint64_t k0 = (uint64_t)tile * WM_TR;
	const uint32_t rows = (uint32_t)std::min<uint64_t>(WM_TR, nrowb - k0);
	for (uint32_t c0 = 0; c0 < nwin; c0 += WM_CS) {                 // column slabs of the tile
		const uint32_t ncs = std::min<uint32_t>(WM_CS, nwp - c0);    // (a multiple of 8)
		const uint32_t nq = ncs >> 3;
		if (c0) __syncthreads();
		for (uint32_t i = threadIdx.x; i < rows * nq; i += WM_NT) {
			const uint32_t r = i / nq, q = i - r * nq;
			const uint4 x = *reinterpret_cast<const uint4 *>(wcnt + (k0 + r) * nwp + c0 + q * 8u);
			*reinterpret_cast<uint4 *>(&s_t[r * WM_CS + q * 8u]) = x;
		}
		__syncthreads();
		const uint32_t ncol = std::min<uint32_t>(ncs, nwin - c0);
		for (uint32_t c = wv; c < ncol; c += WM_NT / 64) {          // a wave per column: lane = row
			const uint32_t w = c0 + c;
			uint32_t run = wbase[w] + tileoff[(uint64_t)w * ntile + tile];
#pragma unroll
			for (uint32_t ch = 0; ch < WM_TR / 64; ++ch) {
				const uint32_t r = ch * 64u + lane;
				const uint32_t v = r < rows ? (uint32_t)s_t[r * WM_CS + c] : 0u;
				const uint32_t inc = wave_inclusive_scan_u32(v);
				if (r < rows) wptr[(uint64_t)w * nrowb + k0 + r] = run + inc - v;
				run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
			}
		}
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

// Per heavy row: products per column window.  A thread owns a PAIR of windows (one 32-bit load per A tuple), sub-groups
// of threads take different tuples and every thread keeps 8 tuples in flight: a workgroup's time is the round trips of its
// tuples' loads, so no workgroup takes more than WH_PART tuples -- the row's own workgroup its first WH_PART, the rest go, part
// by part, into a work list for a second launch that adds into the row's histogram with global atomics.  (With 4096 tuples
// per workgroup both launches took as long as their longest row: 0.47 + 0.25 ms on a 1/8 block of cfg2, 1.6 + 0.4 ms on the whole.)
constexpr int WH_NT = 256;
constexpr int WH_MAXW = 2048;                // windows supported (ncol <= 2^25 at W = 16384)
constexpr uint32_t WH_PART = 512;            // tuples one workgroup takes

struct HubItem { uint32_t h, part; };

__device__ __forceinline__ void win_hist_span(const RowMeta &m, const uint16_t *wcnt, uint32_t nwp, uint32_t beg, uint32_t end, uint32_t *s_cnt)
{
	// a thread owns EIGHT windows (one aligned 16-byte load per A tuple); the workgroup's threads form WH_NT / nq groups that
	// take different tuples, four in flight each
	const uint32_t nq = nwp >> 3;                                       // 16-byte words per row of the table (<= 256)
	const uint4 *tab = reinterpret_cast<const uint4 *>(wcnt);
	const uint32_t nsub = WH_NT / nq;
	const uint32_t sub = threadIdx.x / nq, q = threadIdx.x - sub * nq;
	if (sub >= nsub) return;
	uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto add = [&](const uint4 &x) {
		acc[0] += x.x & 0xFFFFu; acc[1] += x.x >> 16; acc[2] += x.y & 0xFFFFu; acc[3] += x.y >> 16;
		acc[4] += x.z & 0xFFFFu; acc[5] += x.z >> 16; acc[6] += x.w & 0xFFFFu; acc[7] += x.w >> 16;
	};
	uint32_t e = beg + sub;
	for (; e + 3 * nsub < end; e += 4 * nsub) {
		uint4 x[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) x[u] = tab[(uint64_t)(uint32_t)m.acol[e + u * nsub] * nq + q];
#pragma unroll
		for (int u = 0; u < 4; ++u) add(x[u]);
	}
	for (; e < end; e += nsub) add(tab[(uint64_t)(uint32_t)m.acol[e] * nq + q]);
#pragma unroll
	for (int u = 0; u < 8; ++u) if (acc[u]) atomicAdd(&s_cnt[q * 8u + u], acc[u]);
}

__global__ __launch_bounds__(WH_NT) void k_win_hist(const uint32_t *hrows, uint32_t nheavy, RowMeta m, const uint16_t *wcnt,
	uint32_t nwin, uint32_t nwp, uint32_t *winprod, uint32_t *hubcount, HubItem *hublist, uint32_t hubcap)
{
	__shared__ uint32_t s_cnt[WH_MAXW];
	__shared__ uint32_t s_slot, s_parts;
	const uint32_t h = blockIdx.x, r = hrows[h];
	const uint32_t beg = m.beg[r], end = m.beg[r + 1];
	for (uint32_t w = threadIdx.x; w < nwp; w += WH_NT) s_cnt[w] = 0;
	if (threadIdx.x == 0) {
		uint32_t parts = 0, slot = 0;
		if (end - beg > WH_PART) {
			parts = (end - beg - 1) / WH_PART;                          // parts 1 .. of the row
			slot = atomicAdd(hubcount, parts);
			if (slot + parts > hubcap) parts = 0;                       // (cannot happen: the list holds sum(len / WH_PART) entries; then the row is this workgroup's)
		}
		s_slot = slot; s_parts = parts;
	}
	__syncthreads();
	const uint32_t parts = s_parts;
	for (uint32_t i = threadIdx.x; i < parts; i += WH_NT) hublist[s_slot + i] = HubItem{h, i + 1};
	win_hist_span(m, wcnt, nwp, beg, parts ? beg + WH_PART : end, s_cnt);
	__syncthreads();
	for (uint32_t w = threadIdx.x; w < nwin; w += WH_NT) winprod[(uint64_t)h * nwin + w] = s_cnt[w];
}

// The parts beyond the first of the long rows: work item = (row, part).
__global__ __launch_bounds__(WH_NT) void k_win_hist_hub(const uint32_t *hrows, RowMeta m, const uint16_t *wcnt,
	uint32_t nwin, uint32_t nwp, uint32_t *winprod, const uint32_t *hubcount, const HubItem *hublist, uint32_t hubcap)
{
	__shared__ uint32_t s_cnt[WH_MAXW];
	const uint32_t nitem = min(*hubcount, hubcap);
	for (uint32_t item = blockIdx.x; item < nitem; item += gridDim.x) {     // uniform
		const HubItem it = hublist[item];
		const uint32_t r = hrows[it.h];
		const uint32_t beg = m.beg[r] + it.part * WH_PART, end = min(m.beg[r + 1], beg + WH_PART);
		for (uint32_t w = threadIdx.x; w < nwp; w += WH_NT) s_cnt[w] = 0;
		__syncthreads();
		win_hist_span(m, wcnt, nwp, beg, end, s_cnt);
		__syncthreads();
		for (uint32_t w = threadIdx.x; w < nwin; w += WH_NT) { const uint32_t v = s_cnt[w]; if (v) atomicAdd(&winprod[(uint64_t)it.h * nwin + w], v); }
		__syncthreads();
	}
}

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

static void heavy_window_major(spsamd_ctx *c, Heavy &hv, const ConMat &B, uint32_t wshift, const uint16_t *wcnt, uint32_t nwp, Prepared *pb)
{
	hipStream_t st = c->stream;
	const uint64_t nrowb = hv.nrowb, total = nrowb * hv.nwin;
	const uint32_t ntile = (uint32_t)((nrowb + WM_TR - 1) / WM_TR);
	hv.wptr = pb->get<uint32_t>(total + 1);
	uint32_t *tilesum = c->arena.get<uint32_t>((uint64_t)nwp * ntile);
	uint32_t *wtot = c->arena.get<uint32_t>(hv.nwin), *wbase = c->arena.get<uint32_t>(hv.nwin);
	k_wm_tile_sums<<<dim3(ntile), dim3(WM_NT), nwp * sizeof(uint32_t), st>>>(wcnt, nrowb, nwp, ntile, tilesum);
	SPS_LAUNCH_CHECK();
	k_wm_tile_scan<<<dim3(hv.nwin), dim3(WM_NT), 0, st>>>(tilesum, ntile, wtot);
	SPS_LAUNCH_CHECK();
	k_wm_bases<<<dim3(1), dim3(WM_NT), 0, st>>>(wtot, hv.nwin, wbase, hv.wptr + total);
	SPS_LAUNCH_CHECK();
	k_wm_ptr<<<dim3(ntile), dim3(WM_NT), 0, st>>>(wcnt, nrowb, hv.nwin, nwp, ntile, tilesum, wbase, hv.wptr);
	SPS_LAUNCH_CHECK();
	hv.btw = pb->get<BTup>((size_t)B.nnz + DENSE_R);
	k_wm_scatter<<<dim3(grid_for(B.nnz)), dim3(256), 0, st>>>(B.row, B.col, B.val, B.nnz, wshift, hv.bwin, hv.nwin1, hv.wptr, nrowb, hv.btw);
	SPS_LAUNCH_CHECK();
}

// Heavy rows: window index of B, per-row window histogram, counting pass of the cell grouping.
void heavy_prepare(spsamd_ctx *c, Heavy &hv, const Bins &bins, const RowMeta &m, const ConMat &B, const uint32_t *bptr,
	uint32_t extra, uint32_t *nseg, bool ordered, bool pattern, Prepared *pb)
{
	hipStream_t st = c->stream;
	hv.W = B.ncol > (uint64_t(1) << 21) ? 16384 : 8192;
	if (c->tune.window == 8192 || c->tune.window == 16384) hv.W = c->tune.window;
	const uint32_t wshift = hv.W == 8192 ? 13 : 14;
	hv.nwin = (uint32_t)((B.ncol + hv.W - 1) >> wshift);
	if (hv.nwin > (uint32_t)WH_MAXW) throw TooWide{COLBLK};         // spgemm() then multiplies by column blocks of B
	hv.nwin1 = hv.nwin + 1;
	const uint64_t nrowb = B.nrow + extra;
	// B's record already holds the indices for this window width (a prepared operand after its first multiply with heavy rows)
	const bool have_index = pb->bwin && pb->W == hv.W && pb->nrowb == nrowb;
	const bool have_wmajor = have_index && pb->wptr;
	if (!have_index) {
		// The window indices (bwin, its 16-bit counts, the window-major pointer with its counts, the heavy rows' histograms)
		// grow with rows(B) x windows: 12 bytes per B row and window.  Where they would not fit what the device has left
		// (or the cap a test sets), the product goes by column blocks narrow enough for them to fit.
		const uint64_t per_window = nrowb * 12u + (uint64_t)hv.n * 4u;
		uint64_t budget;
		const uint64_t room = (pb->owns || c->arena.slabs.empty()) ? 0 : c->arena.slabs.back().cap - c->arena.slabs.back().used;
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
	hv.rows = bins.rows + bins.off[8];
	hv.winprod = c->arena.get<uint32_t>((uint64_t)hv.n * hv.nwin);
	const uint32_t nwp = (hv.nwin + 7u) & ~7u;
	if (!have_index) {
		pb->reserve(nrowb * hv.nwin1 * 4 + nrowb * nwp * 2 + (nrowb * hv.nwin + 1) * 4 + ((size_t)B.nnz + DENSE_R) * sizeof(BTup) + 4096);   // (a handle: one slab for all four)
		pb->bwin = pb->get<uint32_t>(nrowb * hv.nwin1);
		pb->wcnt = pb->get<uint16_t>(nrowb * nwp);
		pb->W = hv.W; pb->nwin = hv.nwin; pb->nwp = nwp; pb->nrowb = nrowb;
		pb->wptr = nullptr; pb->btw = nullptr;
		fill_zero(c, pb->wcnt, nrowb * nwp * sizeof(uint16_t));
		k_wcnt_count<<<dim3((unsigned)c->num_cu * 16u), dim3(BT_NT), 0, st>>>(B.row, B.col, B.nnz, wshift, nwp, reinterpret_cast<uint32_t *>(pb->wcnt));
		SPS_LAUNCH_CHECK();
		// tile rows: as many as the LDS budget holds (128 at 136 windows, 32 at 512, 8 at 2048)
		uint32_t tr = 256;
		while (tr > 4 && (size_t)tr * (nwp * 2 + 4) > BT_LDS) tr >>= 1;
		static bool lds_attr = false;                               // (more than the default 64 KB of dynamic LDS)
		if (!lds_attr) { SPS_HIP(hipFuncSetAttribute((const void *)k_bwin_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BT_LDS)); lds_attr = true; }
		k_bwin_scan<<<dim3((unsigned)((nrowb + tr - 1) / tr)), dim3(BT_NT), (size_t)tr * (nwp * 2 + 4), st>>>(pb->wcnt, bptr, nrowb, tr, hv.nwin, nwp, pb->bwin);
		SPS_LAUNCH_CHECK();
	}
	hv.bwin = pb->bwin;
	uint16_t *wcnt = pb->wcnt;
	// The window-major copy of B needs only the index just built: it goes to the context's side stream now, beside the
	// histograms, the cell grouping, their host round trips and the numeric kernels that do not read it (a product with heavy
	// rows almost always has dense cells; where it has none the copy was built for nothing).  The main stream waits for it just
	// before the dense cells' kernel (spsamd_ctx::join_side) -- or, whatever happens, before the call ends (spgemm_once).
	if (!c->tune.no_wmajor && have_wmajor) { hv.wptr = pb->wptr; hv.btw = pb->btw; }
	else if (!c->tune.no_wmajor) {
		SPS_HIP(hipEventRecord(c->ev_side[0], st));
		SPS_HIP(hipStreamWaitEvent(c->side, c->ev_side[0], 0));
		c->wm_pending = true;                                       // from here on the main stream must wait for the side stream before this call ends, whatever happens
		c->stream = c->side;                                        // (the helpers launch on c->stream)
		try { heavy_window_major(c, hv, B, wshift, wcnt, nwp, pb); } catch (...) { c->stream = st; (void)hipEventRecord(c->ev_side[1], c->side); throw; }
		c->stream = st;
		SPS_HIP(hipEventRecord(c->ev_side[1], c->side));
		pb->wptr = hv.wptr; pb->btw = hv.btw;
	}
	uint32_t *hubcount = c->arena.get<uint32_t>(1);
	const uint32_t hubcap = (uint32_t)(hv.tuples / WH_PART) + 1u;     // sum over the heavy rows of (tuples - 1) / WH_PART fits
	HubItem *hublist = c->arena.get<HubItem>(hubcap);
	fill_zero(c, hubcount, sizeof(uint32_t));
	k_win_hist<<<dim3(hv.n), dim3(WH_NT), 0, st>>>(hv.rows, hv.n, m, wcnt, hv.nwin, nwp, hv.winprod, hubcount, hublist, hubcap);
	SPS_LAUNCH_CHECK();
	k_win_hist_hub<<<dim3((unsigned)c->num_cu * 8u), dim3(WH_NT), 0, st>>>(hv.rows, m, wcnt, hv.nwin, nwp, hv.winprod, hubcount, hublist, hubcap);
	SPS_LAUNCH_CHECK();
	for (int k = 0; k < NCLS; ++k) {
		hv.cnt.base[k] = c->arena.get<uint32_t>(hv.n);
		hv.base.base[k] = c->arena.get<uint32_t>((size_t)hv.n + 1);
	}
	if (c->tune.cell_cap >= 64 && c->tune.cell_cap <= (int)CELL_CAP) hv.cell_cap = (uint32_t)c->tune.cell_cap;
	if (c->tune.dense_min >= 64 && c->tune.dense_min <= (int)CELL_CAP) hv.dense_min = (uint32_t)c->tune.dense_min;
	// (a window between dense_min and cell_cap products becomes a dense cell; smaller ones are grouped up to cell_cap)
	// everything the counting pass accumulates into sits in ONE block, zeroed by one memset (they were eleven)
	const size_t hn4 = ((size_t)hv.n + 3) & ~(size_t)3;
	uint32_t *zblock = c->arena.get<uint32_t>(4 * hn4 + 2 * (NCLS + 2) + 4);
	const size_t zbytes = (4 * hn4 + 2 * (NCLS + 2) + 4) * sizeof(uint32_t);
	unsigned long long *clsprod = reinterpret_cast<unsigned long long *>(zblock + 4 * hn4);
	unsigned long long *alt_block = clsprod + NCLS + 2;
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
	{
		int q = 0;
		for (TileBases *t : {&hv.tb, &hv.tb2}) {
			t->ntc = zblock + hn4 * q++; t->ntl = zblock + hn4 * q++;
			t->tcbase = c->arena.get<uint32_t>((size_t)hv.n + 1); t->tlbase = c->arena.get<uint32_t>((size_t)hv.n + 1);
		}
	}
	auto count_pass = [&]() {
		fill_zero(c, zblock, zbytes);
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
		hv.alt_cells = alt_block;                                    // [0] cells of the hash scheme, [1] of the bitmap scheme (zeroed with the rest)
		count_pass();
		WordList wl; wl.add64(hv.alt_cells); wl.add64(hv.alt_cells + 1);
		uint32_t hw[4];
		read_back_words(c, wl, hw);
		const unsigned long long cells_hash = (unsigned long long)hw[0] | ((unsigned long long)hw[1] << 32);
		const unsigned long long cells_bm = (unsigned long long)hw[2] | ((unsigned long long)hw[3] << 32);
		hv.alt_cells = nullptr;
		if (c->tune.trace) fprintf(stderr, "tile cells: bitmap %llu hash %llu\n", cells_bm, cells_hash);
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
}

// Emit the cells (needs segbase for the COO sink) and order the dense ones by descending products.
void heavy_cells(spsamd_ctx *c, Heavy &hv, const RowMeta &m, const uint32_t *segbase)
{
	hipStream_t st = c->stream;
	CellLists lists;
	for (int k = 0; k < NCLS; ++k) { hv.cells[k] = c->arena.get<Cell>(hv.ncell[k] ? hv.ncell[k] : 1); lists.list[k] = hv.cells[k]; }
	hv.tb.tcells = c->arena.get<TCell>(hv.ntcell ? hv.ntcell : 1);
	hv.tb.tiles = c->arena.get<Tile>(hv.ntile ? hv.ntile : 1);
	hv.tb2.tcells = c->arena.get<TCell>(hv.ntcell2 ? hv.ntcell2 : 1);
	hv.tb2.tiles = c->arena.get<Tile>(hv.ntile2 ? hv.ntile2 : 1);
	k_cells<true><<<dim3(grid_for(hv.n, 128)), dim3(128), 0, st>>>(hv.rows, hv.n, m.beg, m.id, hv.winprod, hv.nwin, hv.cell_cap, hv.dense_min, hv.cnt, nullptr, hv.base, lists, segbase, nullptr, tile_kinds(hv));
	SPS_LAUNCH_CHECK();
	SPS_HIP(hipEventRecord(c->ev_side2[0], st));
	if (c->tune.trace && hv.ntile) {
		// what the tiles' index probes touch: per (tile, A tuple) the row's window entries from the first cell's wa to the last cell's wb
		std::vector<Tile> ht(hv.ntile); std::vector<TCell> hc(hv.ntcell);
		SPS_HIP(hipMemcpyAsync(ht.data(), hv.tb.tiles, ht.size() * sizeof(Tile), hipMemcpyDeviceToHost, st));
		SPS_HIP(hipMemcpyAsync(hc.data(), hv.tb.tcells, hc.size() * sizeof(TCell), hipMemcpyDeviceToHost, st));
		SPS_HIP(hipStreamSynchronize(st));
		double sumL = 0, probes = 0, lines = 0, prods = 0, spanw = 0;
		for (const Tile &t : ht) {
			const double L = t.end - t.beg;
			const uint32_t wa = hc[t.first].wa, wb = hc[t.first + t.ncells - 1].wb;
			sumL += L; probes += L * t.ncells; prods += t.prods; spanw += wb - wa;
			lines += L * ((double)((wb - wa + 1) * 4 + 127) / 128.0 + 0.5);
		}
		fprintf(stderr, "tiles %u cells %u: mean L %.1f, cells per tile %.2f, windows per tile %.1f, probe threads %.3g, index lines (128 B) %.3g = %.1f GB, products %.3g\n",
			hv.ntile, hv.ntcell, sumL / hv.ntile, (double)hv.ntcell / hv.ntile, spanw / hv.ntile, probes, lines, lines * 128 / 1e9, prods);
	}
}

// The lists are put in window-major order by some fifty small launches -- a quarter of a millisecond of launch latency
// whatever the lists' sizes, host and device alike -- that nothing but the heavy rows' kernels waits for.  The driver
// (spgemm_once) calls this AFTER it has enqueued the light and mid rows' kernels, and the launches go to the context's third
// stream: the device sorts beside those kernels while the host is still enqueueing; the main stream joins before the first
// kernel that walks a list.
void heavy_sort_lists(spsamd_ctx *c, Heavy &hv)
{
	hipStream_t st = c->stream;
	int wbits = 1;                                   // bits of a window index: the cell lists are sorted on as few digits as needed
	while ((1u << wbits) < hv.nwin) ++wbits;
	SPS_HIP(hipStreamWaitEvent(c->side2, c->ev_side2[0], 0));       // (recorded after the cells were emitted)
	c->sort_pending = true;
	struct OnSide2 {
		spsamd_ctx *c; hipStream_t main;
		~OnSide2() { c->stream = main; (void)hipEventRecord(c->ev_side2[1], c->side2); }
	} on_side2{c, st};
	c->stream = c->side2;
	st = c->side2;
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
	// XCD parts of the (window-major) lists: eight contiguous parts of equal estimated cost, part x for the workgroups with
	// blockIdx % 8 == x, which share an XCD and its L2.  Walked STATICALLY (part x only by group x) this is slower than one
	// list for all (round 1: dense 80 vs 57 ms, hash 73 vs 62; round 3: dense 47 vs 33 although the L2 misses halve -- the
	// parts' run times differ); the dense cells' list is therefore CLAIMED (k_dense: per-part counters, other parts' cells
	// once the own part is used up: 33.1 -> 30.0 ms, L2 hit rate 51 -> 85 %).  The tiles gain nothing (a tile's B rows span
	// 16 windows of the row-major copy: 36 -> 41 % hits) and the hash cells' list is short: both keep one list.
	// SPSAMD_XCD / "xcd": 0 one list, 1 static parts for every class (the experiment), 2 (default) claimed parts for the dense cells.
	const int xmode = c->tune.xcd;
	for (int k = 0; k < NCLS && xmode != 0; ++k) {
		if (xmode == 2 && k != CLS_DENSE) continue;
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


} // namespace spsamd
