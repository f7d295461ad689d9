// prims.hip -- workspace arena, device-wide exclusive scan, stable LSD radix sort.
//
// These replace, on the device, what the reference gets from the STL on the
// host: std::stable_sort on a permutation (algorithm.hpp:411-427) and the
// running offsets of its linear passes.  Hand-written for gfx950 (wave64).
#include "internal.h"
#include "devutil.h"

namespace spsamd {

// ------------------------------------------------------------------ arena

static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void *Arena::alloc(size_t bytes)
{
	bytes = round_up(bytes ? bytes : 1, 256);
	call_used += bytes;
	if (!slabs.empty()) {
		Slab &s = slabs.back();
		if (s.used + bytes <= s.cap) {
			void *p = s.p + s.used;
			s.used += bytes;
			return p;
		}
	}
	size_t cap = bytes > (size_t(256) << 20) ? bytes : (size_t(256) << 20);
	if (cap < high_water / 2) cap = high_water / 2;
	Slab s{nullptr, cap, 0};
	hipError_t e = hipMalloc((void **)&s.p, cap);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		throw Error{SPSAMD_ENOMEM, "workspace hipMalloc of " + std::to_string(cap) + " bytes failed: " + hipGetErrorString(e)};
	}
	s.used = bytes;
	slabs.push_back(s);
	return s.p;
}

void Arena::rewind(const Mark &m)
{
	if (call_used > high_water) high_water = call_used;             // the peak counts, not what is live afterwards
	for (size_t q = m.nslabs; q < slabs.size(); ++q) slabs[q].used = 0;
	if (m.nslabs && m.nslabs <= slabs.size() && m.nslabs == slabs.size()) slabs[m.nslabs - 1].used = m.used;
	call_used = m.call_used;
}

void Arena::reset()
{
	if (call_used > high_water) high_water = call_used;
	call_used = 0;
	if (slabs.size() > 1 || (slabs.size() == 1 && slabs[0].cap < high_water)) {
		// re-make as one slab that holds the largest call seen so far
		size_t want = round_up(high_water + high_water / 8, size_t(1) << 20);
		release();
		reserve(want);
	}
	for (auto &s : slabs) s.used = 0;
}

void Arena::reserve(size_t bytes)
{
	if (slabs.size() == 1 && slabs[0].cap >= bytes) return;
	release();
	Slab s{nullptr, bytes, 0};
	hipError_t e = hipMalloc((void **)&s.p, bytes);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		throw Error{SPSAMD_ENOMEM, "workspace hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e)};
	}
	slabs.push_back(s);
}

void Arena::release()
{
	for (auto &s : slabs) (void)hipFree(s.p);
	slabs.clear();
}

void DevBuf::ensure(size_t bytes)
{
	if (bytes <= cap) return;
	release();
	size_t want = round_up(bytes + bytes / 16, 4096);
	hipError_t e = hipMalloc(&p, want);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		p = nullptr; cap = 0;
		throw Error{SPSAMD_ENOMEM, "output hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e)};
	}
	cap = want;
}

void DevBuf::release()
{
	if (p) (void)hipFree(p);
	p = nullptr; cap = 0;
}

void Prepared::reserve(size_t bytes)
{
	if (!owns || bytes <= slab_left) return;
	bytes = round_up(bytes, size_t(2) << 20);
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		throw Error{SPSAMD_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes for a prepared operand failed: " + hipGetErrorString(e)};
	}
	owned.push_back(p);
	owned_bytes += bytes;
	slab = (char *)p; slab_left = bytes;
}

void *Prepared::alloc(size_t bytes)
{
	if (!owns) return ctx->arena.alloc(bytes);
	bytes = round_up(bytes ? bytes : 1, 256);
	if (bytes > slab_left) reserve(bytes);
	void *p = slab;
	slab += bytes; slab_left -= bytes;
	return p;
}

void Prepared::release()
{
	for (void *p : owned) (void)hipFree(p);
	owned.clear();
	owned_bytes = 0;
	slab = nullptr; slab_left = 0;
}

} // namespace spsamd

void spsamd_ctx::join_side(bool wm, bool sort)
{
	if (wm && wm_pending) { wm_pending = false; (void)hipStreamWaitEvent(stream, ev_side[1], 0); }
	if (sort && sort_pending) { sort_pending = false; (void)hipStreamWaitEvent(stream, ev_side2[1], 0); }
}

void *spsamd_ctx::host_staging(size_t bytes)
{
	if (bytes > pinned_cap) {
		if (pinned) (void)hipHostFree(pinned);
		pinned = nullptr; pinned_cap = 0;
		size_t want = bytes < 4096 ? 4096 : bytes;
		hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocDefault);
		if (e != hipSuccess) {
			(void)hipGetLastError();
			throw spsamd::Error{SPSAMD_ENOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e)};
		}
		pinned_cap = want;
	}
	return pinned;
}

namespace spsamd {

// ------------------------------------------------------------------ fill

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) p[i] = v;
}

void fill_u32(spsamd_ctx *c, uint32_t *p, uint32_t v, size_t n)
{
	if (!n) return;
	size_t blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	k_fill_u32<<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(p, v, n);
	SPS_LAUNCH_CHECK();
}

void fill_zero(spsamd_ctx *c, void *p, size_t bytes)
{
	if (bytes) SPS_HIP(hipMemsetAsync(p, 0, bytes, c->stream));
}

// ------------------------------------------------------------------ scan

constexpr int SCAN_NT = 256;
constexpr int SCAN_ITEMS = 4;     // consecutive elements per thread (8 and 16 measured slower: the strided reads coalesce worse; 2 the same)
constexpr int SCAN_TILE = SCAN_NT * SCAN_ITEMS;

template <class TIn, class TOut>
__global__ __launch_bounds__(SCAN_NT) void k_scan_tile_sums(const TIn *in, size_t n, TOut *sums)
{
	__shared__ TOut scratch[SCAN_NT / 64 + 1];
	size_t base = (size_t)blockIdx.x * SCAN_TILE;
	TOut s = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + (size_t)q * SCAN_NT + threadIdx.x;
		if (i < n) s += (TOut)in[i];
	}
	s = wave_reduce_sum(s);
	if (lane_id() == 0) scratch[wave_id()] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		TOut t = 0;
		for (int w = 0; w < SCAN_NT / 64; ++w) t += scratch[w];
		sums[blockIdx.x] = t;
	}
}

// out[i] = tile_off[tile] + exclusive prefix inside the tile; the thread that
// owns element n-1 also writes out[n] = grand total.
template <class TIn, class TOut>
__global__ __launch_bounds__(SCAN_NT) void k_scan_tiles(const TIn *in, size_t n, const TOut *tile_off, TOut *out)
{
	__shared__ TOut scratch[SCAN_NT / 64 + 1];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
	TOut v[SCAN_ITEMS];
	TOut s = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + q;
		v[q] = i < n ? (TOut)in[i] : (TOut)0;
		s += v[q];
	}
	TOut ex = block_exclusive_scan<TOut, SCAN_NT>(s, scratch, (TOut *)nullptr);
	TOut run = ex + (tile_off ? tile_off[blockIdx.x] : (TOut)0);
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + q;
		if (i < n) {
			out[i] = run;
			run += v[q];
			if (i == n - 1) out[n] = run;
		}
	}
}

template <class TIn, class TOut>
static void scan_exclusive(spsamd_ctx *c, const TIn *in, TOut *out, size_t n)
{
	if (n == 0) {
		fill_zero(c, out, sizeof(TOut));
		return;
	}
	size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	TOut *offs = nullptr;
	if (ntiles > 1) {
		TOut *sums = c->arena.get<TOut>(ntiles);
		offs = c->arena.get<TOut>(ntiles + 1);
		k_scan_tile_sums<TIn, TOut><<<dim3((unsigned)ntiles), dim3(SCAN_NT), 0, c->stream>>>(in, n, sums);
		SPS_LAUNCH_CHECK();
		scan_exclusive<TOut, TOut>(c, sums, offs, ntiles);
	}
	k_scan_tiles<TIn, TOut><<<dim3((unsigned)ntiles), dim3(SCAN_NT), 0, c->stream>>>(in, n, offs, out);
	SPS_LAUNCH_CHECK();
}

void scan_exclusive_u32_i64(spsamd_ctx *c, const uint32_t *in, int64_t *out, size_t n) { scan_exclusive<uint32_t, int64_t>(c, in, out, n); }
void scan_exclusive_u32_u32(spsamd_ctx *c, const uint32_t *in, uint32_t *out, size_t n) { scan_exclusive<uint32_t, uint32_t>(c, in, out, n); }
void scan_exclusive_u8_u32(spsamd_ctx *c, const uint8_t *in, uint32_t *out, size_t n) { scan_exclusive<uint8_t, uint32_t>(c, in, out, n); }
void scan_exclusive_u16_u32(spsamd_ctx *c, const uint16_t *in, uint32_t *out, size_t n) { scan_exclusive<uint16_t, uint32_t>(c, in, out, n); }

// ---- batched: arrays b.in[y] -> b.out[y], y = blockIdx.y ----

__global__ __launch_bounds__(SCAN_NT) void k_scan_tile_sums_batch(ScanBatch b, size_t n, uint32_t *sums, uint32_t ntiles)
{
	__shared__ uint32_t scratch[SCAN_NT / 64 + 1];
	const uint32_t *in = b.in[blockIdx.y];
	size_t base = (size_t)blockIdx.x * SCAN_TILE;
	uint32_t s = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + (size_t)q * SCAN_NT + threadIdx.x;
		if (i < n) s += in[i];
	}
	s = wave_reduce_sum(s);
	if (lane_id() == 0) scratch[wave_id()] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int w = 0; w < SCAN_NT / 64; ++w) t += scratch[w];
		sums[(size_t)blockIdx.y * ntiles + blockIdx.x] = t;
	}
}

// one workgroup per array: exclusive scan of its ntiles tile sums, in place
__global__ __launch_bounds__(SCAN_NT) void k_scan_sums_batch(uint32_t *sums, uint32_t ntiles)
{
	__shared__ uint32_t scratch[SCAN_NT / 64 + 1];
	uint32_t *a = sums + (size_t)blockIdx.x * ntiles;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < ntiles; base += SCAN_NT) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < ntiles ? a[i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_exclusive_scan<uint32_t, SCAN_NT>(v, scratch, &tot);
		if (i < ntiles) a[i] = carry + ex;
		carry += tot;
		__syncthreads();
	}
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_tiles_batch(ScanBatch b, size_t n, const uint32_t *tile_off, uint32_t ntiles)
{
	__shared__ uint32_t scratch[SCAN_NT / 64 + 1];
	const uint32_t *in = b.in[blockIdx.y];
	uint32_t *out = b.out[blockIdx.y];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS];
	uint32_t s = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + q;
		v[q] = i < n ? in[i] : 0u;
		s += v[q];
	}
	uint32_t ex = block_exclusive_scan<uint32_t, SCAN_NT>(s, scratch, (uint32_t *)nullptr);
	uint32_t run = ex + tile_off[(size_t)blockIdx.y * ntiles + blockIdx.x];
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; ++q) {
		size_t i = base + q;
		if (i < n) {
			out[i] = run;
			run += v[q];
			if (i == n - 1) out[n] = run;
		}
	}
}

void scan_exclusive_u32_batch(spsamd_ctx *c, const ScanBatch &b, size_t n)
{
	if (b.count == 0) return;
	if (n == 0) {
		for (int y = 0; y < b.count; ++y) fill_zero(c, b.out[y], sizeof(uint32_t));
		return;
	}
	const size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	uint32_t *sums = c->arena.get<uint32_t>(ntiles * (size_t)b.count);
	k_scan_tile_sums_batch<<<dim3((unsigned)ntiles, (unsigned)b.count), dim3(SCAN_NT), 0, c->stream>>>(b, n, sums, (uint32_t)ntiles);
	SPS_LAUNCH_CHECK();
	k_scan_sums_batch<<<dim3((unsigned)b.count), dim3(SCAN_NT), 0, c->stream>>>(sums, (uint32_t)ntiles);
	SPS_LAUNCH_CHECK();
	k_scan_tiles_batch<<<dim3((unsigned)ntiles, (unsigned)b.count), dim3(SCAN_NT), 0, c->stream>>>(b, n, sums, (uint32_t)ntiles);
	SPS_LAUNCH_CHECK();
}

__global__ void k_collect_words(WordList w, uint32_t *dst)
{
	const int i = (int)threadIdx.x;
	if (i < w.count) dst[i] = *w.p[i];
}

void read_back_words(spsamd_ctx *c, const WordList &w, uint32_t *host)
{
	if (w.count == 0) return;
	uint32_t *dev = c->arena.get<uint32_t>(WORD_LIST_MAX);
	k_collect_words<<<dim3(1), dim3(64), 0, c->stream>>>(w, dev);
	SPS_LAUNCH_CHECK();
	uint32_t *h = (uint32_t *)c->host_staging(sizeof(uint32_t) * WORD_LIST_MAX);
	SPS_HIP(hipMemcpyAsync(h, dev, sizeof(uint32_t) * (size_t)w.count, hipMemcpyDeviceToHost, c->stream));
	SPS_HIP(hipStreamSynchronize(c->stream));
	for (int i = 0; i < w.count; ++i) host[i] = h[i];
}

// ------------------------------------------------------------------ radix sort

constexpr int RS_NT = 256;                 // 4 waves
constexpr int RS_NW = RS_NT / 64;
constexpr int RS_ITER = 16;                // 64-item chunks per wave
constexpr int RS_TILE = RS_NT * RS_ITER;   // 4096 items per workgroup

template <int BITS>
__global__ __launch_bounds__(RS_NT) void k_rs_hist(const uint64_t *keys, size_t n, int shift, uint32_t *ghist, unsigned nblocks)
{
	constexpr int NB = 1 << BITS;
	__shared__ uint32_t hist[NB];
	for (int d = threadIdx.x; d < NB; d += RS_NT) hist[d] = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
	for (int q = 0; q < RS_ITER; ++q) {
		size_t i = base + (size_t)q * RS_NT + threadIdx.x;
		if (i < n) atomicAdd(&hist[(unsigned)(keys[i] >> shift) & (unsigned)(NB - 1)], 1u);
	}
	__syncthreads();
	for (int d = threadIdx.x; d < NB; d += RS_NT) ghist[(size_t)d * nblocks + blockIdx.x] = hist[d];
}

// Stable scatter: wave w of a workgroup owns the contiguous items
// [tile + w*1024, tile + (w+1)*1024) and walks them 64 at a time, so the order
// (workgroup, wave, chunk, lane) is the input order.  The tile is first sorted by digit INSIDE the
// workgroup (LDS), then written out: a digit's items leave as one contiguous run instead of one
// 12-byte record per lane (the direct scatter ran at 1.2 TB/s of record traffic).
template <int BITS>
__global__ __launch_bounds__(RS_NT) void k_rs_scatter(const uint64_t *keys, const uint32_t *pay, size_t n, int shift,
	const uint32_t *gbase, unsigned nblocks, uint64_t *keys_out, uint32_t *pay_out, int iota_payload)
{
	constexpr int NB = 1 << BITS;
	constexpr int DPT = NB / RS_NT;                 // digits per thread (consecutive)
	__shared__ uint32_t wcnt[RS_NW][NB];
	__shared__ uint32_t gdelta[NB];                // global position of a digit's run minus its position in the tile
	__shared__ uint32_t scr[RS_NW + 1];
	__shared__ uint64_t s_key[RS_TILE];
	__shared__ uint32_t s_pay[RS_TILE];
	const unsigned w = wave_id(), lane = lane_id();
	for (int q = threadIdx.x; q < RS_NW * NB; q += RS_NT) (&wcnt[0][0])[q] = 0;
	__syncthreads();
	const size_t tile = (size_t)blockIdx.x * RS_TILE;
	const size_t wbase = tile + (size_t)w * (RS_ITER * 64);
	uint64_t key[RS_ITER];
#pragma unroll
	for (int q = 0; q < RS_ITER; ++q) {
		size_t i = wbase + (size_t)q * 64 + lane;
		key[q] = i < n ? keys[i] : 0;
		if (i < n) atomicAdd(&wcnt[w][(unsigned)(key[q] >> shift) & (unsigned)(NB - 1)], 1u);
	}
	__syncthreads();
	{
		// thread t: digits [t DPT, (t+1) DPT): their runs inside the tile (exclusive prefix over the digits) and per-wave starts
		uint32_t tot[DPT], mine = 0;
#pragma unroll
		for (int x = 0; x < DPT; ++x) {
			const unsigned d = threadIdx.x * DPT + x;
			tot[x] = 0;
#pragma unroll
			for (int ww = 0; ww < RS_NW; ++ww) tot[x] += wcnt[ww][d];
			mine += tot[x];
		}
		uint32_t lbase = block_exclusive_scan<uint32_t, RS_NT>(mine, scr, (uint32_t *)nullptr);
#pragma unroll
		for (int x = 0; x < DPT; ++x) {
			const unsigned d = threadIdx.x * DPT + x;
			gdelta[d] = gbase[(size_t)d * nblocks + blockIdx.x] - lbase;
			uint32_t run = lbase;
#pragma unroll
			for (int ww = 0; ww < RS_NW; ++ww) {
				uint32_t t = wcnt[ww][d];
				wcnt[ww][d] = run;
				run += t;
			}
			lbase += tot[x];
		}
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < RS_ITER; ++q) {
		size_t i = wbase + (size_t)q * 64 + lane;
		bool valid = i < n;
		unsigned digit = (unsigned)(key[q] >> shift) & (unsigned)(NB - 1);
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < BITS; ++b) {
			bool bit = (digit >> b) & 1u;
			uint64_t m = __ballot(valid && bit);
			peers &= bit ? m : ~m;
		}
		// peers: valid lanes of this chunk with my digit (meaningless on invalid lanes)
		unsigned rank = __popcll(peers & lanemask_lt());
		uint32_t old = 0;
		if (valid && rank == 0) old = atomicAdd(&wcnt[w][digit], (uint32_t)__popcll(peers));
		int leader = valid ? (__ffsll((unsigned long long)peers) - 1) : (int)lane;
		old = __shfl(old, leader, 64);
		if (valid) {
			s_key[old + rank] = key[q];
			s_pay[old + rank] = iota_payload ? (uint32_t)i : pay[i];
		}
	}
	__syncthreads();
	const uint32_t nvalid = (uint32_t)(n - tile < (size_t)RS_TILE ? n - tile : (size_t)RS_TILE);
	for (uint32_t p = threadIdx.x; p < nvalid; p += RS_NT) {
		const uint64_t k = s_key[p];
		const uint32_t dst = gdelta[(unsigned)(k >> shift) & (unsigned)(NB - 1)] + p;
		keys_out[dst] = k;
		pay_out[dst] = s_pay[p];
	}
}

template <int BITS>
static int radix_sort_pairs_bits(spsamd_ctx *c, uint64_t *keys0, uint32_t *pay0, uint64_t *keys1, uint32_t *pay1, size_t n, int key_bits, int low_bit)
{
	// payload of the first pass is the identity permutation (iota_payload)
	constexpr size_t NB = size_t(1) << BITS;
	int passes = (key_bits - low_bit + BITS - 1) / BITS;
	unsigned nblocks = (unsigned)((n + RS_TILE - 1) / RS_TILE);
	uint32_t *ghist = c->arena.get<uint32_t>(NB * nblocks);
	uint32_t *gbase = c->arena.get<uint32_t>(NB * nblocks + 1);
	uint64_t *ksrc = keys0, *kdst = keys1;
	uint32_t *psrc = pay0, *pdst = pay1;
	int where = 0;
	if (passes == 0) passes = 1;   // all keys equal: one pass on a zero digit keeps the input order
	for (int p = 0; p < passes; ++p) {
		int shift = low_bit + BITS * p;
		k_rs_hist<BITS><<<dim3(nblocks), dim3(RS_NT), 0, c->stream>>>(ksrc, n, shift, ghist, nblocks);
		SPS_LAUNCH_CHECK();
		scan_exclusive_u32_u32(c, ghist, gbase, NB * nblocks);
		k_rs_scatter<BITS><<<dim3(nblocks), dim3(RS_NT), 0, c->stream>>>(ksrc, psrc, n, shift, gbase, nblocks, kdst, pdst, p == 0);
		SPS_LAUNCH_CHECK();
		uint64_t *tk = ksrc; ksrc = kdst; kdst = tk;
		uint32_t *tp = psrc; psrc = pdst; pdst = tp;
		where ^= 1;
	}
	return where;
}

int radix_sort_pairs(spsamd_ctx *c, uint64_t *keys0, uint32_t *pay0, uint64_t *keys1, uint32_t *pay1, size_t n, int key_bits, int low_bit)
{
	if (n == 0) return 0;
	if (low_bit < 0 || low_bit > key_bits) low_bit = 0;
	// (10-bit digits save a pass on 40-bit keys but each pass is 35 % slower on MI355X -- the counters' LDS leaves two
	// workgroups per CU instead of three: 1.51 against 1.40 ms for 1.7e7 pairs -- so 8 bits it stays)
	return radix_sort_pairs_bits<8>(c, keys0, pay0, keys1, pay1, n, key_bits, low_bit);
}

} // namespace spsamd
