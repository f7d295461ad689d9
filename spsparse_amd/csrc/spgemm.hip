// spgemm.hip -- the hot path: C = c * Di * op(A) * Dj * op(B) * Dk on gfx950.  This file is the DRIVER (row classes,
// segments, sinks, column-block fallback, the small symbolic kernels); the numeric kernels live in k_light.hip, k_hash.hip,
// k_dense.hip and k_tiles.hip, the heavy rows' symbolic phase in symbolic_heavy.hip (source map: DESIGN.md section 4).
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
#include "spgemm_host.h"

namespace spsamd {

__global__ void k_pack_b(const int32_t *bcol, const double *bval, uint32_t n, BTup *out)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double v = bval[i];
	BTup t; t.col = bcol[i]; t.vlo = (uint32_t)__double2loint(v); t.vhi = (uint32_t)__double2hiint(v);
	out[i] = t;
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
constexpr int CLS_ITEMS = 4;                    // rows per thread: fewer workgroups -> fewer same-address global atomics, but the rows of a thread
                                                // are a chain of dependent loads (16: 78 us for the 68 K rows of a 1/8 block of cfg2 -- 17 workgroups)

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

// longest row of a dense row pointer
__global__ void k_max_rowlen(const uint32_t *ptr, uint64_t nrow, uint32_t *out)
{
	uint32_t v = 0;
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (uint64_t)gridDim.x * blockDim.x) v = max(v, ptr[r + 1] - ptr[r]);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
	// (same-address atomics serialise: one per workgroup, and only where it would raise the maximum)
	__shared__ uint32_t s_v;
	if (threadIdx.x == 0) s_v = 0;
	__syncthreads();
	if (lane_id() == 0 && v) atomicMax(&s_v, v);
	__syncthreads();
	if (threadIdx.x == 0 && s_v > *(volatile uint32_t *)out) atomicMax(out, s_v);
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

// The heavy rows' hash-class cells: tiles (whichever scheme the call picked), direct tiles, then the un-tiled cells.
template <int MODE>
static void launch_heavy_hash(spsamd_ctx *c, const Heavy &hv, const RowMeta &m, const EmitParams &ep, const SinkParams &sk)
{
	c->join_side(hv.ntile2 != 0, true);                             // the sorted lists (and, for direct cells, the window-major copy)
	SPS_HIP(hipEventRecord(c->ev2[0], c->stream));
	if (hv.ntile && hv.tiles2 == 0) launch_tiles_bm<MODE>(c, hv, m, ep, sk);
	else if (hv.ntile && hv.tiles2 == 2) launch_tiles_hash2<MODE>(c, hv, m, ep, sk);
	else if (hv.ntile) launch_tiles_v1<MODE>(c, hv, m, ep, sk);
	SPS_HIP(hipEventRecord(c->ev2[1], c->stream));
	if (hv.ntile2) launch_tiles_direct<MODE>(c, hv, m, ep, sk);
	SPS_HIP(hipEventRecord(c->ev2[2], c->stream));
	launch_hash_windowed<MODE>(c, hv, m, ep, sk);
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
	long long *row_nnz, double *row_sum, unsigned long long *row_hash)
{
	__shared__ unsigned long long s_u64[2 * 4];
	__shared__ double s_f64[4];
	unsigned long long cnt = 0, hash = 0; double sum = 0;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t jj = (uint32_t)j[t] + c0;
		const unsigned long long h = mix64((uint32_t)i[t], jj);
		++cnt; hash += h; sum += v[t];
		if (row_nnz) { atomicAdd((unsigned long long *)&row_nnz[i[t]], 1ull); atomicAdd(&row_sum[i[t]], v[t]); atomicAdd(&row_hash[i[t]], h); }
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

// DIGEST | ROWSTATS: the context's three per-row arrays (tuple count, value sum, index hash), zeroed
static void rowstats_begin(spsamd_ctx *c, uint64_t nrow, SinkParams &sk, spsamd_result *res)
{
	c->rowstat_n.ensure(nrow * sizeof(long long));
	c->rowstat_s.ensure(nrow * sizeof(double));
	c->rowstat_h.ensure(nrow * sizeof(unsigned long long));
	fill_zero(c, c->rowstat_n.p, nrow * sizeof(long long));
	fill_zero(c, c->rowstat_s.p, nrow * sizeof(double));
	fill_zero(c, c->rowstat_h.p, nrow * sizeof(unsigned long long));
	sk.row_nnz = (long long *)c->rowstat_n.p;
	sk.row_sum = (double *)c->rowstat_s.p;
	sk.row_hash = (unsigned long long *)c->rowstat_h.p;
	res->row_nnz = (const int64_t *)sk.row_nnz;
	res->row_sum = sk.row_sum;
	res->row_hash = (const uint64_t *)sk.row_hash;
}

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
			rowstats_begin(c, A.nrow, sk, res);
		}
		launch_light_direct_s<MODE_DIGEST>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
		k_digest_reduce<<<dim3(1), dim3(64), 0, st>>>(slots, slots + DIGEST_SLOTS, sk.err);
		SPS_LAUNCH_CHECK();
		SPS_HIP(hipEventRecord(c->ev[4], st));
		DigestSlot d = read_back(c, slots + DIGEST_SLOTS);
		res->nnz = d.count; res->hash = d.hash; res->sum = d.sum;
	} else {
		// COO: one segment per row of op(A) (empty rows included) -- or, in one pass, per wave round of the kernel
		const uint32_t S = maxp <= 8 ? 8u : (maxp <= 16 ? 16u : (maxp <= 32 ? 32u : 64u));
		const uint32_t G = 64u / S;
		const size_t nround = ((size_t)nrow + 4u * G - 1u) / (4u * G) * 4u;      // wave rounds: G rows each, four waves per workgroup round
		const size_t nsegs = std::max<size_t>(nrow, nround);
		uint32_t *segcount = c->arena.get<uint32_t>(nsegs + 1);
		uint32_t *segactual = c->arena.get<uint32_t>(nsegs + 1);
		int64_t *segoff = c->arena.get<int64_t>(nsegs + 1);
		sk.segcount = segcount; sk.segactual = segactual;
		OutSet &os = c->out[c->cur_out];
		// ONE compute pass where the memory is there: every wave round writes the tuples of its G rows, packed, to its own 64
		// slots of a sparse buffer and says how many; scan; a gather packs the rounds.  The other way -- count, scan, store
		// -- evaluates every product twice: Poisson 4096^2 8.1 ms against 3.8 for the digest.
		const uint64_t slots = (uint64_t)nround * 64u;
		bool one_pass = !c->tune.light_two_pass && slots < (uint64_t(1) << 36);
		if (one_pass) {
			const uint64_t room = c->arena.slabs.empty() ? 0 : c->arena.slabs.back().cap - c->arena.slabs.back().used;
			if (slots * 16 + 4096 > room) {                             // (not the steady state: would the workspace have to grow beyond reason?)
				size_t freeb = 0, totalb = 0;
				SPS_HIP(hipMemGetInfo(&freeb, &totalb));
				if (slots * 16 > (freeb + room) / 3) one_pass = false;
			}
		}
		int64_t total;
		if (one_pass) {
			int32_t *si = c->arena.get<int32_t>(slots), *sj = c->arena.get<int32_t>(slots);
			double *sv = c->arena.get<double>(slots);
			sk.segoff = nullptr; sk.out_i = si; sk.out_j = sj; sk.out_v = sv;
			launch_light_direct_s<MODE_STORE>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
			scan_exclusive_u32_i64(c, segactual, segoff, nround);
			total = read_back(c, segoff + nround);
			os.i.ensure((size_t)total * sizeof(int32_t));
			os.j.ensure((size_t)total * sizeof(int32_t));
			os.v.ensure((size_t)total * sizeof(double));
			sk.out_i = (int32_t *)os.i.p; sk.out_j = (int32_t *)os.j.p; sk.out_v = (double *)os.v.p;
			launch_light_gather(c, (uint32_t)nround, 64u, segactual, segoff, si, sj, sv, sk.out_i, sk.out_j, sk.out_v);
		} else {
			sk.segoff = segoff;
			launch_light_direct_s<MODE_COUNT>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
			scan_exclusive_u32_i64(c, segcount, segoff, nrow);
			total = read_back(c, segoff + nrow);
			os.i.ensure((size_t)total * sizeof(int32_t));
			os.j.ensure((size_t)total * sizeof(int32_t));
			os.v.ensure((size_t)total * sizeof(double));
			sk.out_i = (int32_t *)os.i.p; sk.out_j = (int32_t *)os.j.p; sk.out_v = (double *)os.v.p;
			fill_zero(c, pc, sizeof(unsigned long long));
			launch_light_direct_s<MODE_STORE>(c, maxp, nrow, aptr, acol, aval, bptr, B, k64, ep, sk, pc);
		}
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

// ---- derived structures of an operand: taken from its Prepared record where they exist, built (and kept there) otherwise
static uint32_t *ensure_rowptr(spsamd_ctx *c, Prepared *p)
{
	// one empty sentinel row after the last: it receives the A tuples whose k is absent from scalej
	if (!p->rowptr) p->rowptr = dense_rowptr(c, p->m, 1u, p->get<uint32_t>(p->m.nrow + 2));
	return p->rowptr;
}

static void ensure_maxlen(spsamd_ctx *c, Prepared *pa, Prepared *pb)
{
	hipStream_t st = c->stream;
	Prepared *need[2]; int n = 0;
	if (!pa->have_maxlen) need[n++] = pa;
	if (pb != pa && !pb->have_maxlen) need[n++] = pb;
	if (!n) return;
	uint32_t *mx = c->arena.get<uint32_t>(2);
	fill_zero(c, mx, 2 * sizeof(uint32_t));
	for (int q = 0; q < n; ++q) {
		k_max_rowlen<<<dim3(std::min(grid_for(need[q]->m.nrow, 1024), 1024u)), dim3(256), 0, st>>>(need[q]->rowptr, need[q]->m.nrow, mx + q);
		SPS_LAUNCH_CHECK();
	}
	struct { uint32_t a, b; } hm = read_back(c, (const decltype(hm) *)mx);
	need[0]->maxlen = hm.a; need[0]->have_maxlen = true;
	if (n > 1) { need[1]->maxlen = hm.b; need[1]->have_maxlen = true; }
}

void prepared_row_structure(spsamd_ctx *c, Prepared *p)
{
	ensure_rowptr(c, p);
	ensure_maxlen(c, p, p);
}

static const RowList &ensure_rowlist(spsamd_ctx *c, Prepared *p)
{
	if (p->have_rl) return p->rl;
	if (!p->owns) dim_beginnings(c, p->m, &p->rl);
	else {
		// a handle keeps its own copy (the arena's is gone after this call)
		RowList tmp;
		dim_beginnings(c, p->m, &tmp);
		p->rl.nrows = tmp.nrows;
		p->rl.beg = p->get<uint32_t>((size_t)tmp.nrows + 1);
		p->rl.id = p->get<int32_t>(tmp.nrows ? tmp.nrows : 1);
		SPS_HIP(hipMemcpyAsync(p->rl.beg, tmp.beg, ((size_t)tmp.nrows + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
		SPS_HIP(hipMemcpyAsync(p->rl.id, tmp.id, (size_t)tmp.nrows * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
	}
	p->have_rl = true;
	return p->rl;
}

static void wait_b_tuples(spsamd_ctx *c, MultiplyArgs &a)
{
	if (a.b_ready) { SPS_HIP(hipStreamWaitEvent(c->stream, a.b_ready, 0)); a.b_ready = nullptr; }
}

static BTup *ensure_btup(spsamd_ctx *c, MultiplyArgs &a, Prepared *p)
{
	if (p->btup) return p->btup;
	wait_b_tuples(c, a);
	p->btup = p->get<BTup>((size_t)p->m.nnz + DENSE_R);             // a dense-cell item reads R tuples: slack after the last one
	k_pack_b<<<dim3(grid_for(p->m.nnz)), dim3(256), 0, c->stream>>>(p->m.col, p->m.val, p->m.nnz, p->btup);
	SPS_LAUNCH_CHECK();
	return p->btup;
}

static void spgemm_once(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res)
{
	hipStream_t st = c->stream;
	const ConMat &A = a.A, &B = a.B;
	res->nnz_a = A.nnz; res->nnz_b = B.nnz;
	if (A.nnz == 0 || B.nnz == 0) return;           // empty product (also SURVEY Appendix A.3)

	SPS_HIP(hipEventRecord(c->ev[1], st));
	// views for operands that come without a record of their own (pieces in this call's arena)
	Prepared viewA, viewB;
	viewA.ctx = viewB.ctx = c;
	viewA.m = A; viewB.m = B;
	const bool same = (a.pa && a.pa == a.pb) || (A.row == B.row && A.col == B.col && A.nnz == B.nnz && A.nrow == B.nrow);
	Prepared *pa = a.pa ? a.pa : &viewA;
	Prepared *pb = a.pb ? a.pb : (same ? pa : &viewB);
	if (same && a.pb && !a.pa) pa = pb;
	const uint32_t extra = 1u;                       // (the sentinel row: ensure_rowptr)
	uint32_t *bptr = ensure_rowptr(c, pb);
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
		const uint32_t *aptr = pa == pb ? bptr : ensure_rowptr(c, pa);
		ensure_maxlen(c, pa, pb);
		if ((uint64_t)pa->maxlen * pb->maxlen <= 64 && A.nrow < (uint64_t(1) << 32)) {
			wait_b_tuples(c, a);
			spgemm_all_light(c, a, res, aptr, acol, aval, bptr, (uint32_t)((uint64_t)pa->maxlen * pb->maxlen));
			return;
		}
	}
	// ---- row structure of A (dim_beginnings)
	const RowList rl = ensure_rowlist(c, pa);

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

	BTup *btup = ensure_btup(c, a, pb);
	RowMeta m{rl.beg, rl.id, acol, aval, bptr, btup, btup, elo, elen};
#ifdef SPSAMD_ABLATIONS
	set_ablation_word(c, c->tune.dbg);
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
	// work this call puts on the context's other streams is waited for by the main stream before the first kernel that needs
	// it -- and before the call ends, whatever happens (its buffers are this call's workspace)
	struct SideGuard { spsamd_ctx *c; ~SideGuard() { c->join_side(true, true); } } side_guard{c};
	Heavy hv;
	hv.n = bins.count[8];
	hv.tuples = hbc.tuples[8];
	hv.coo = coo;
	if (hv.n) heavy_prepare(c, hv, bins, m, B, bptr, extra, nseg, (a.sink_flags & SPSAMD_SINK_ORDERED) != 0, (a.sink_flags & SPSAMD_SINK_EXACT_PATTERN) != 0, pb);
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
			rowstats_begin(c, A.nrow, sk, res);
		}
		SPS_HIP(hipEventRecord(c->ev[3], st));
		launch_light<MODE_DIGEST>(c, bins, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[4], st));
		launch_mid<MODE_DIGEST>(c, bins, mc, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[5], st));
		if (hv.n) heavy_sort_lists(c, hv);
		launch_heavy_hash<MODE_DIGEST>(c, hv, m, ep, sk);
		SPS_HIP(hipEventRecord(c->ev[6], st));
		c->join_side(true, true);
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
		if (hv.n) heavy_sort_lists(c, hv);
		launch_heavy_hash<MODE_COUNT>(c, hv, m, ep, sk);
		c->join_side(true, true);
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
	wait_b_tuples(c, a);
	c->own[c->cur_out].sort0 = -1;                                  // (the blocks go through that output set, whatever the sink)
	struct BlockOut { int32_t *i = nullptr, *j = nullptr; double *v = nullptr; uint64_t n = 0; };
	struct Blocks {                                                 // the blocks' COO outputs until they are interleaved
		std::vector<BlockOut> b;
		~Blocks() { for (auto &x : b) { (void)hipFree(x.i); (void)hipFree(x.j); (void)hipFree(x.v); } }
	} blocks;
	DigestSlot *slots = nullptr;
	SinkParams rsk{};
	spsamd_result rres{};
	if (!coo) {
		slots = c->arena.get<DigestSlot>(DIGEST_SLOTS + 1);
		fill_zero(c, slots, (DIGEST_SLOTS + 1) * sizeof(DigestSlot));
		if (a.sink_flags & SPSAMD_SINK_ROWSTATS) rowstats_begin(c, A.nrow, rsk, &rres);
	}
	long long *const row_nnz = rsk.row_nnz; double *const row_sum = rsk.row_sum; unsigned long long *const row_hash = rsk.row_hash;
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
			as.pb = nullptr;                                            // (B restricted to the block: nothing of B's record applies)
			as.b_ready = nullptr;
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
					k_block_digest<<<dim3(std::min<unsigned>(grid_for((size_t)rs.nnz), 4096u)), dim3(256), 0, st>>>(rs.idx0, rs.idx1, rs.val, rs.nnz, (uint32_t)c0, slots, row_nnz, row_sum, row_hash);
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
		if (row_nnz) { res->row_nnz = (const int64_t *)row_nnz; res->row_sum = row_sum; res->row_hash = (const uint64_t *)row_hash; }
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
