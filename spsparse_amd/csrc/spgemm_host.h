// spgemm_host.h -- host-side structures of the multiply path and the launchers its translation units export
#pragma once
#include "spgemm_dev.h"

namespace spsamd {

static inline unsigned grid_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

struct Bins {
	uint32_t count[NBIN];
	uint32_t off[NBIN + 1];
	uint32_t *rows;
};

struct MidCells { Cell *cells[3] = {nullptr, nullptr, nullptr}; };

// Thrown by heavy_prepare: op(B) has more column windows than the heavy-row path indexes, or its window indices would not
// fit the device -- spgemm() then multiplies by column blocks of `width` columns (spgemm_column_blocks).
struct TooWide { uint64_t width; };
constexpr uint64_t COLBLK = (uint64_t)2048 << 14;                   // 2048 windows of 16384 columns: 2^25

struct Heavy {
	uint32_t n = 0;                  // heavy rows
	uint64_t tuples = 0;             // ... and their A tuples
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

static inline TileKinds tile_kinds(const Heavy &hv)
{
	TileKinds tk{{hv.tb, hv.tb2}, hv.direct_min, hv.span_cap, hv.long_cap, hv.long_dense_min,
		hv.tiles2 == 0 ? (uint32_t)BM_MAXOUT : (uint32_t)(TILE_T / 2), hv.alt_cap, hv.alt_span, hv.alt_cells};
	return tk;
}

static inline float elapsed(hipEvent_t a, hipEvent_t b)
{
	float ms = 0;
	SPS_HIP(hipEventElapsedTime(&ms, a, b));
	return ms;
}

// ---- launchers (explicitly instantiated for MODE_COUNT / MODE_STORE / MODE_DIGEST in the file named)
template <int MODE> void launch_light(spsamd_ctx *c, const Bins &b, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);             // k_light.hip
template <int MODE> void launch_light_direct_s(spsamd_ctx *c, uint32_t maxp, uint32_t nrow, const uint32_t *aptr, const int32_t *acol, const double *aval,
	const uint32_t *bptr, const ConMat &B, bool k64, const EmitParams &ep, const SinkParams &sk, unsigned long long *pc);                         // k_light.hip
void launch_light_gather(spsamd_ctx *c, uint32_t nrow, uint32_t S, const uint32_t *cnt, const int64_t *off,
	const int32_t *si, const int32_t *sj, const double *sv, int32_t *oi, int32_t *oj, double *ov);                                                    // k_light.hip
template <int MODE> void launch_mid(spsamd_ctx *c, const Bins &b, const MidCells &mc, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);  // k_hash.hip
template <int MODE> void launch_hash_windowed(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);  // k_hash.hip
template <int MODE> void launch_tiles_v1(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);       // k_hash.hip
template <int MODE> void launch_tiles_bm(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);       // k_tiles.hip
template <int MODE> void launch_tiles_hash2(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);    // k_tiles.hip
template <int MODE> void launch_tiles_direct(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk);   // k_tiles.hip
template <int MODE> void launch_heavy_dense(spsamd_ctx *c, const Heavy &hv, const RowMeta &m0, const EmitParams &ep, const SinkParams &sk);   // k_dense.hip

// ---- the heavy rows' symbolic phase (symbolic_heavy.hip)
void heavy_prepare(spsamd_ctx *c, Heavy &hv, const Bins &bins, const RowMeta &m, const ConMat &B, const uint32_t *bptr,
	uint32_t extra, uint32_t *nseg, bool ordered, bool pattern, Prepared *pb);
void heavy_cells(spsamd_ctx *c, Heavy &hv, const RowMeta &m, const uint32_t *segbase);
void heavy_sort_lists(spsamd_ctx *c, Heavy &hv);

#ifdef SPSAMD_ABLATIONS
void set_ablation_word(spsamd_ctx *c, int word);          // k_hash.hip (the only unit that reads it through ABLG)
#endif

#define SPSAMD_INSTANTIATE_MODES(decl_macro) decl_macro(MODE_COUNT) decl_macro(MODE_STORE) decl_macro(MODE_DIGEST)

} // namespace spsamd
