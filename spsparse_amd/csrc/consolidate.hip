// consolidate.hip -- device restatement of the operand canonicalisation that
// spsparse::multiply runs before its loops:
//   Consolidate<>          algorithm.hpp:353-369
//   consolidate()          algorithm.hpp:251-319  (stable sort, zero drop, duplicate merge)
//   sorted_permutation()   algorithm.hpp:411-427
//   dim_beginnings()       algorithm.hpp:74-118
// Data layout in HBM: SoA like VectorCooArray (one int32 array per dimension,
// one f64 array), sorted by (row of op(X), col of op(X)).
#include "internal.h"
#include "devutil.h"

namespace spsamd {

static int bits_for(uint64_t dim)
{
	// bits needed to hold indices 0 .. dim-1
	int b = 0;
	while (b < 63 && (uint64_t(1) << b) < dim) ++b;
	return b;
}

// flags: bit0 = an index is out of [0, shape); bit1 = not (strictly sorted, no zero values); bit2 = not strictly in
// (minor, major) order; bit3 = holds a value consolidate() may drop; bit4 = the major index descends somewhere
__global__ void k_inspect(const int32_t *major, const int32_t *minor, const double *val, size_t n,
	uint64_t nrow, uint64_t ncol, int zero_nan, uint32_t *flags)
{
	// grid-stride: a wave raises the shared flag word ONCE, after all its elements (one read of that word per element, then per
	// wave of 64 elements, was most of this kernel's time on large operands)
	uint32_t f = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		int32_t r = major[i], c = minor[i];
		if (r < 0 || (uint64_t)r >= nrow || c < 0 || (uint64_t)c >= ncol) f |= 1u;
		double v = val[i];
		if (v == 0 || (zero_nan && v != v)) f |= 2u | 8u;         // 8: a value consolidate() may drop
		if (i > 0) {
			int32_t pr = major[i - 1], pc = minor[i - 1];
			if (!(pr < r || (pr == r && pc < c))) f |= 2u;
			if (pr > r) f |= 16u;                                       // the MAJOR index itself descends: not even row-grouped
			if (!(pc < c || (pc == c && pr < r))) f |= 4u;          // not STRICTLY in (minor, major) order either
		}
	}
	// one update of the shared flag word per WORKGROUP (per wave, the reads of that one word were most of this kernel's
	// time: 0.1 ms whatever the operand's size)
	__shared__ uint32_t s_f;
	if (threadIdx.x == 0) s_f = 0;
	__syncthreads();
	uint32_t wf = 0;
#pragma unroll
	for (uint32_t b = 1u; b <= 16u; b <<= 1) if (__ballot(f & b)) wf |= b;
	if (wf && lane_id() == 0) atomicOr(&s_f, wf);
	__syncthreads();
	if (threadIdx.x == 0 && s_f && (*(volatile uint32_t *)flags & s_f) != s_f) atomicOr(flags, s_f);
}

__global__ void k_build_keys(const int32_t *major, const int32_t *minor, size_t n, int minor_bits, uint64_t *keys)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) keys[i] = ((uint64_t)(uint32_t)major[i] << minor_bits) | (uint64_t)(uint32_t)minor[i];
}

// Gather the values into sorted order and flag the tuples consolidate() keeps:
// zeros are skipped (algorithm.hpp:284-292).  With zero_nan the reference also skips the NaNs
// of the LEADING run of its own sorted sequence (algorithm.hpp:272-275; later NaNs survive, :291).
// That sequence is ordered by the reference's sort order for this operand, which for B is the
// other dimension than the one this library sorts by (multiply_sparse.hpp:168,188): the first
// kept tuple is therefore found as the minimum of (reference key, insertion position) over the
// tuples that are neither 0 nor NaN -- two atomicMin passes -- and every NaN below it is dropped.
__device__ __forceinline__ uint64_t ref_key(uint64_t key, int minor_bits, int major_bits, int swap)
{
	if (!swap) return key;
	const uint64_t minor = key & ((uint64_t(1) << minor_bits) - 1), major = key >> minor_bits;
	return (minor << major_bits) | major;
}

__global__ void k_gather_flag(const double *val, const uint32_t *perm, const uint64_t *keys, size_t n, int zero_nan,
	int minor_bits, int major_bits, int swap, double *sval, uint8_t *keep, unsigned long long *first_key)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double v = val[perm[i]];
	sval[i] = v;
	keep[i] = (v != 0) ? 1 : 0;
	if (zero_nan && v != 0 && v == v) atomicMin(first_key, (unsigned long long)ref_key(keys[i], minor_bits, major_bits, swap));
}

__global__ void k_first_pos(const double *sval, const uint32_t *perm, const uint64_t *keys, size_t n,
	int minor_bits, int major_bits, int swap, const unsigned long long *first_key, uint32_t *first_pos)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double v = sval[i];
	if (v != 0 && v == v && ref_key(keys[i], minor_bits, major_bits, swap) == *first_key) atomicMin(first_pos, perm[i]);
}

__global__ void k_apply_first(const double *sval, const uint32_t *perm, const uint64_t *keys, uint8_t *keep, size_t n,
	int minor_bits, int major_bits, int swap, const unsigned long long *first_key, const uint32_t *first_pos)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double v = sval[i];
	if (v == v) return;                                  // only NaNs are in question (zeros are dropped anyway)
	const uint64_t k = ref_key(keys[i], minor_bits, major_bits, swap), fk = *first_key;
	if (k < fk || (k == fk && perm[i] < *first_pos)) keep[i] = 0;
}

__global__ void k_compact(const uint64_t *keys, const double *sval, const uint8_t *keep, const uint32_t *pos, size_t n,
	uint64_t *kk, double *kv)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && keep[i]) {
		uint32_t p = pos[i];
		kk[p] = keys[i];
		kv[p] = sval[i];
	}
}

__global__ void k_heads(const uint64_t *kk, const uint32_t *nk_ptr, size_t n, uint8_t *head)
{
	size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	uint32_t nk = *nk_ptr;
	head[j] = (j < nk && (j == 0 || kk[j] != kk[j - 1])) ? 1 : 0;
}

// One thread per distinct index: walks its run of equal keys in sorted (=
// insertion, the sort is stable) order and applies the DuplicatePolicy
// exactly as algorithm.hpp:306-310 does, left to right.
__global__ void k_merge(const uint64_t *kk, const double *kv, const uint8_t *head, const uint32_t *hpos,
	const uint32_t *nk_ptr, size_t n, int minor_bits, int policy,
	int32_t *row, int32_t *col, double *val)
{
	size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n || !head[j]) return;
	uint32_t nk = *nk_ptr;
	uint64_t key = kk[j];
	double acc = kv[j];
	for (size_t t = j + 1; t < nk && kk[t] == key; ++t) {
		if (policy == SPSAMD_ADD) acc += kv[t];
		else if (policy == SPSAMD_REPLACE) acc = kv[t];
	}
	uint32_t o = hpos[j];
	row[o] = (int32_t)(key >> minor_bits);
	col[o] = (int32_t)(key & ((uint64_t(1) << minor_bits) - 1));
	val[o] = acc;
}

// An operand with nothing to drop and nothing to merge: the sorted tuples are the consolidated ones.
__global__ void k_gather_sorted(const uint64_t *keys, const uint32_t *perm, const double *val, size_t n, int minor_bits,
	int32_t *row, int32_t *col, double *oval)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = keys[i];
	row[i] = (int32_t)(key >> minor_bits);
	col[i] = (int32_t)(key & ((uint64_t(1) << minor_bits) - 1));
	oval[i] = val[perm[i]];
}

static unsigned grid_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

template <class T>
static const T *to_device(spsamd_ctx *c, const T *p, size_t n, int mem)
{
	if (mem != SPSAMD_MEM_HOST || n == 0) return p;
	T *d = c->arena.get<T>(n);
	SPS_HIP(hipMemcpyAsync(d, p, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
	return d;
}

__global__ void k_min_key(unsigned long long *first_key, const unsigned long long *global_key)
{
	if (*global_key < *first_key) *first_key = *global_key;
}

void consolidate_operand(spsamd_ctx *c, const spsamd_coo *X, int lead, int ref_lead, int duplicate_policy, int zero_nan, ConMat *out,
	Prepared **prep, const unsigned long long *global_first_key)
{
	if (prep) *prep = nullptr;
	if (X->mem == SPSAMD_MEM_PREPARED) {
		// a prepared operand (spsamd_operand_prepare): consolidated once, by the lead it was prepared for.  Used the other way
		// round ('T' now, '.' then) its tuples are an ordinary device operand sorted by the other dimension.
		Prepared *p = (Prepared *)const_cast<int32_t *>(X->idx0);
		if (!p || p->ctx != c) throw Error{SPSAMD_EINVAL, "a prepared operand belongs to the context that prepared it"};
		if (p->lead == lead) { *out = p->m; if (prep) *prep = p; return; }
		spsamd_coo Y;
		Y.idx0 = p->lead == 0 ? p->m.row : p->m.col; Y.idx1 = p->lead == 0 ? p->m.col : p->m.row; Y.val = p->m.val;
		Y.nnz = p->m.nnz; Y.shape0 = X->shape0; Y.shape1 = X->shape1; Y.sort0 = p->lead; Y.mem = SPSAMD_MEM_DEVICE;
		consolidate_operand(c, &Y, lead, ref_lead, duplicate_policy, zero_nan, out, nullptr, global_first_key);
		return;
	}
	size_t n = X->nnz;
	if (n >= (size_t(1) << 31))
		throw Error{SPSAMD_EINVAL, "operand has 2^31 or more tuples (the reference's int positions cap it too, algorithm.hpp:419)"};
	uint64_t shape[2] = {X->shape0, X->shape1};
	out->nrow = shape[lead];
	out->ncol = shape[1 - lead];
	out->nnz = 0;
	out->row = out->col = nullptr;
	out->val = nullptr;
	if (n == 0) return;
	if (!X->idx0 || !X->idx1 || !X->val) throw Error{SPSAMD_EINVAL, "operand with nnz > 0 has a null array"};
	if (shape[0] > (uint64_t(1) << 31) || shape[1] > (uint64_t(1) << 31))
		throw Error{SPSAMD_EINVAL, "shape exceeds the int32 index range"};

	bool own_result = false;                                            // a result of this context, chained back in as it stands
	if (X->mem == SPSAMD_MEM_DEVICE && X->sort0 == lead)
		for (const auto &o : c->own)
			if (o.sort0 == X->sort0 && o.d0 == X->idx0 && o.d1 == X->idx1 && o.v == X->val && o.nnz == n && o.shape0 == X->shape0 && o.shape1 == X->shape1) own_result = true;
	if ((X->mem == SPSAMD_MEM_DEVICE_VERIFIED || own_result) && X->sort0 == lead) {       // the distributed step's own block / panel
		out->row = const_cast<int32_t *>(lead == 0 ? X->idx0 : X->idx1);
		out->col = const_cast<int32_t *>(lead == 0 ? X->idx1 : X->idx0);
		out->val = const_cast<double *>(X->val);
		out->nnz = (uint32_t)n;
		return;
	}
	const int32_t *d0 = to_device(c, X->idx0, n, X->mem);
	const int32_t *d1 = to_device(c, X->idx1, n, X->mem);
	const double *dv = to_device(c, X->val, n, X->mem);
	const int32_t *major = lead == 0 ? d0 : d1;
	const int32_t *minor = lead == 0 ? d1 : d0;

	uint32_t *flags = c->arena.get<uint32_t>(2);
	fill_zero(c, flags, 2 * sizeof(uint32_t));
	k_inspect<<<dim3(std::min(grid_for(n, 1024), 2048u)), dim3(256), 0, c->stream>>>(major, minor, dv, n, out->nrow, out->ncol, zero_nan, flags);
	SPS_LAUNCH_CHECK();
	uint32_t f = read_back(c, flags);
	if (f & 1u) throw Error{SPSAMD_EINVAL, "Sparse index out of bounds (VectorCooArray::add would reject it, VectorCooArray.hpp:246-262)"};

	// A claimed sort order is trusted like Consolidate<> trusts it (algorithm.hpp:360) as far as the ORDER INSIDE A ROW and
	// duplicates go; rows that are not even ascending would leave the dense row pointer (built from the tuples' side) with
	// entries nobody wrote -- out-of-bounds reads in the numeric kernels, where the reference would "only" mis-compute.
	if (X->sort0 == lead && (f & 16u))
		throw Error{SPSAMD_EINVAL, "operand claims sort_order but its leading index is not ascending (set_sorted() on unsorted tuples?)"};
	// Consolidate<>: a matching sort_order is trusted (algorithm.hpp:360); so is
	// an operand the inspection found strictly sorted with no zero values.
	// (Under zero_nan an operand whose reference order is the other dimension and that holds a 0 / NaN
	// still goes through the filter below: which NaNs go depends on the reference's sequence.)
	if ((X->sort0 == lead && !(zero_nan && ref_lead != lead && (f & 2u))) || !(f & 2u)) {
		out->row = const_cast<int32_t *>(major);
		out->col = const_cast<int32_t *>(minor);
		out->val = const_cast<double *>(dv);
		out->nnz = (uint32_t)n;
		return;
	}

	int mb = bits_for(out->ncol), Mb = bits_for(out->nrow);
	uint64_t *keys0 = c->arena.get<uint64_t>(n), *keys1 = c->arena.get<uint64_t>(n);
	uint32_t *pay0 = c->arena.get<uint32_t>(n), *pay1 = c->arena.get<uint32_t>(n);
	k_build_keys<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(major, minor, n, mb, keys0);
	SPS_LAUNCH_CHECK();
	// An operand that is STRICTLY in (minor, major) order -- a matrix kept sorted by rows and used with 'T', cfg5's R -- needs
	// the stable passes over the MAJOR digits only: an LSD sort that skips the low digits leaves ties in input order, which is
	// the minor order already (Galerkin 256^3, R^T: 3 passes instead of 6).  Strictly: no two tuples share their indices, so
	// with no value to drop either the sorted tuples ARE the consolidated operand (no flag / compact / merge passes).
	const int low_bit = (f & 4u) ? 0 : mb;
	int where = radix_sort_pairs(c, keys0, pay0, keys1, pay1, n, mb + Mb, low_bit);
	uint64_t *ks = where ? keys1 : keys0;
	uint32_t *ps = where ? pay1 : pay0;
	uint64_t *kk = where ? keys0 : keys1;      // the other key buffer is free now
	if (!(f & 4u) && !(f & 8u)) {
		out->row = c->arena.get<int32_t>(n);
		out->col = c->arena.get<int32_t>(n);
		out->val = c->arena.get<double>(n);
		k_gather_sorted<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(ks, ps, dv, n, mb, out->row, out->col, out->val);
		SPS_LAUNCH_CHECK();
		out->nnz = (uint32_t)n;
		return;
	}

	double *sval = c->arena.get<double>(n);
	uint8_t *keep = c->arena.get<uint8_t>(n);
	unsigned long long *first_key = c->arena.get<unsigned long long>(1);
	uint32_t *first_pos = flags + 1;
	const int swap = ref_lead != lead ? 1 : 0;
	if (zero_nan) {
		SPS_HIP(hipMemsetAsync(first_key, 0xFF, sizeof(unsigned long long), c->stream));
		fill_u32(c, first_pos, 0xFFFFFFFFu, 1);
	}
	k_gather_flag<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(dv, ps, ks, n, zero_nan, mb, Mb, swap, sval, keep, first_key);
	SPS_LAUNCH_CHECK();
	if (zero_nan) {
		// (distributed step: the first kept tuple of the WHOLE matrix may sit in another rank's block)
		if (global_first_key) { k_min_key<<<dim3(1), dim3(1), 0, c->stream>>>(first_key, global_first_key); SPS_LAUNCH_CHECK(); }
		k_first_pos<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(sval, ps, ks, n, mb, Mb, swap, first_key, first_pos);
		SPS_LAUNCH_CHECK();
		k_apply_first<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(sval, ps, ks, keep, n, mb, Mb, swap, first_key, first_pos);
		SPS_LAUNCH_CHECK();
	}
	uint32_t *pos = c->arena.get<uint32_t>(n + 1);
	scan_exclusive_u8_u32(c, keep, pos, n);
	double *kv = c->arena.get<double>(n);
	k_compact<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(ks, sval, keep, pos, n, kk, kv);
	SPS_LAUNCH_CHECK();
	uint8_t *head = keep;                      // reuse
	k_heads<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(kk, pos + n, n, head);
	SPS_LAUNCH_CHECK();
	uint32_t *hpos = c->arena.get<uint32_t>(n + 1);
	scan_exclusive_u8_u32(c, head, hpos, n);
	out->row = c->arena.get<int32_t>(n);
	out->col = c->arena.get<int32_t>(n);
	out->val = c->arena.get<double>(n);
	k_merge<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(kk, kv, head, hpos, pos + n, n, mb, duplicate_policy,
		out->row, out->col, out->val);
	SPS_LAUNCH_CHECK();
	out->nnz = read_back(c, hpos + n);
}

// The smallest reference-order key among the tuples of X (device arrays) that consolidate() keeps whatever zero_nan says --
// neither 0 nor NaN -- as k_gather_flag finds it for one operand.  The distributed step takes the minimum of this over all
// ranks' blocks: the reference's leading run (algorithm.hpp:272-275) is a property of the whole matrix.
__global__ void k_first_key_raw(const int32_t *major, const int32_t *minor, const double *val, size_t n, int minor_bits, int major_bits, int swap,
	unsigned long long *first_key)
{
	unsigned long long best = ~0ull;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const double v = val[i];
		if (v != 0 && v == v) {
			const uint64_t key = ((uint64_t)(uint32_t)major[i] << minor_bits) | (uint64_t)(uint32_t)minor[i];
			const unsigned long long k = ref_key(key, minor_bits, major_bits, swap);
			best = k < best ? k : best;
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(best, d, 64); best = o < best ? o : best; }
	if (lane_id() == 0 && best != ~0ull) atomicMin(first_key, best);
}

void first_kept_key_raw(spsamd_ctx *c, const spsamd_coo *Xdev, int lead, int ref_lead, unsigned long long *out_dev)
{
	SPS_HIP(hipMemsetAsync(out_dev, 0xFF, sizeof(unsigned long long), c->stream));
	const size_t n = Xdev->nnz;
	if (!n) return;
	const uint64_t shape[2] = {Xdev->shape0, Xdev->shape1};
	const int mb = bits_for(shape[1 - lead]), Mb = bits_for(shape[lead]);
	const int32_t *major = lead == 0 ? Xdev->idx0 : Xdev->idx1, *minor = lead == 0 ? Xdev->idx1 : Xdev->idx0;
	k_first_key_raw<<<dim3(std::min(grid_for(n), 2048u)), dim3(256), 0, c->stream>>>(major, minor, Xdev->val, n, mb, Mb, ref_lead != lead ? 1 : 0, out_dev);
	SPS_LAUNCH_CHECK();
}

uint32_t *sorted_permutation(spsamd_ctx *c, const spsamd_coo *X, int lead)
{
	size_t n = X->nnz;
	if (n >= (size_t(1) << 31)) throw Error{SPSAMD_EINVAL, "operand has 2^31 or more tuples"};
	if (n == 0) return nullptr;
	if (!X->idx0 || !X->idx1) throw Error{SPSAMD_EINVAL, "operand with nnz > 0 has a null array"};
	uint64_t shape[2] = {X->shape0, X->shape1};
	const int32_t *d0 = to_device(c, X->idx0, n, X->mem);
	const int32_t *d1 = to_device(c, X->idx1, n, X->mem);
	const int32_t *major = lead == 0 ? d0 : d1;
	const int32_t *minor = lead == 0 ? d1 : d0;
	int mb = bits_for(shape[1 - lead]), Mb = bits_for(shape[lead]);
	uint64_t *keys0 = c->arena.get<uint64_t>(n), *keys1 = c->arena.get<uint64_t>(n);
	uint32_t *pay0 = c->arena.get<uint32_t>(n), *pay1 = c->arena.get<uint32_t>(n);
	k_build_keys<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(major, minor, n, mb, keys0);
	SPS_LAUNCH_CHECK();
	int where = radix_sort_pairs(c, keys0, pay0, keys1, pay1, n, mb + Mb);
	return where ? pay1 : pay0;
}

// ------------------------------------------------------------------ dim_beginnings

__global__ void k_row_flags(const int32_t *row, size_t n, uint8_t *flag)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || row[i] != row[i - 1]) ? 1 : 0;
}

__global__ void k_row_starts(const int32_t *row, const uint8_t *flag, const uint32_t *pos, size_t n, uint32_t *beg, int32_t *id)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (flag[i]) {
		beg[pos[i]] = (uint32_t)i;
		id[pos[i]] = row[i];
	}
	if (i == n - 1) beg[pos[n]] = (uint32_t)n;      // sentinel (algorithm.hpp:95-99)
}

void dim_beginnings(spsamd_ctx *c, const ConMat &m, RowList *out)
{
	size_t n = m.nnz;
	out->nrows = 0;
	out->beg = c->arena.get<uint32_t>(n + 1);
	out->id = c->arena.get<int32_t>(n ? n : 1);
	if (n == 0) return;
	uint8_t *flag = c->arena.get<uint8_t>(n);
	uint32_t *pos = c->arena.get<uint32_t>(n + 1);
	k_row_flags<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(m.row, n, flag);
	SPS_LAUNCH_CHECK();
	scan_exclusive_u8_u32(c, flag, pos, n);
	k_row_starts<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(m.row, flag, pos, n, out->beg, out->id);
	SPS_LAUNCH_CHECK();
	out->nrows = read_back(c, pos + n);
}

// ------------------------------------------------------------------ dense row pointer

__global__ void k_rowptr(const int32_t *row, uint32_t n, uint64_t nptr, uint32_t *ptr)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nptr) return;
	// first tuple whose row index is >= r
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if ((uint64_t)(uint32_t)row[mid] < r) lo = mid + 1; else hi = mid;
	}
	ptr[r] = lo;
}

// The same from the tuples' side: tuple t (and the end, t = n) writes the entries of the rows between its predecessor's row
// and its own -- one streaming pass over the row indices instead of a binary search per row (cfg3, 8.4e7 tuples in 1.7e7
// rows: 0.41 -> 0.1 ms).  A short gap of empty rows is filled by its thread; a long one -- the rows before and after a row
// BLOCK of a sharded product are empty: hundreds of thousands of entries -- by the whole wave (left to one thread such a
// gap cost 10 ms per call).
__global__ void k_rowptr_scatter(const int32_t *row, uint32_t n, uint64_t nptr, uint32_t *ptr)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t first = 0, last = 0;                                    // [first, last) = rows r with ptr[r] = t
	if (t > 0 && t < n) {                                           // (the rows up to the first tuple's and after the last one's: k_rowptr_ends)
		first = (uint64_t)(uint32_t)row[t - 1] + 1;                 // entries below were written by earlier tuples
		last = std::min<uint64_t>((uint64_t)(uint32_t)row[t] + 1, nptr);
	}
	const uint64_t gap = last > first ? last - first : 0;
	if (gap <= 8) for (uint64_t r = first; r < last; ++r) ptr[r] = (uint32_t)t;
	uint64_t big = __ballot(gap > 8);
	while (big) {                                                   // uniform
		const int l = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)big) - 1);
		big &= big - 1ull;
		const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(first >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)first, l);
		const uint64_t e = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(last >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)last, l);
		const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, l);
		for (uint64_t r = f + lane_id(); r < e; r += 64) ptr[r] = v;
	}
}

// ... and the two ends, one thread per pointer entry: rows up to the first tuple's row point at 0, rows after the last
// tuple's at n (the two long gaps of a row block).
__global__ void k_rowptr_ends(const int32_t *row, uint32_t n, uint64_t nptr, uint32_t *ptr)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nptr) return;
	if (r <= (uint64_t)(uint32_t)row[0]) ptr[r] = 0u;
	else if (r > (uint64_t)(uint32_t)row[n - 1]) ptr[r] = n;
}

uint32_t *dense_rowptr(spsamd_ctx *c, const ConMat &m, uint32_t extra, uint32_t *into)
{
	uint64_t nptr = m.nrow + 1 + extra;
	uint32_t *ptr = into ? into : c->arena.get<uint32_t>(nptr);
	if (!m.nnz) { fill_zero(c, ptr, nptr * sizeof(uint32_t)); return ptr; }
	if (m.nnz && nptr <= 8ull * m.nnz + 1024) {
		k_rowptr_ends<<<dim3(grid_for(nptr)), dim3(256), 0, c->stream>>>(m.row, m.nnz, nptr, ptr);
		SPS_LAUNCH_CHECK();
		k_rowptr_scatter<<<dim3(grid_for((size_t)m.nnz + 1)), dim3(256), 0, c->stream>>>(m.row, m.nnz, nptr, ptr);
	}
	else k_rowptr<<<dim3(grid_for(nptr)), dim3(256), 0, c->stream>>>(m.row, m.nnz, nptr, ptr);     // (mostly empty rows: a search per pointer entry)
	SPS_LAUNCH_CHECK();
	return ptr;
}

// ------------------------------------------------------------------ scale vectors

__global__ void k_scale_scatter(const int32_t *idx, size_t n, uint64_t dim, int32_t *pos, uint32_t *bad)
{
	size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	int32_t i = idx[t];
	if (t > 0 && idx[t - 1] >= i) atomicOr(bad, 1u);
	// an index outside the dimension can never match in the reference's join
	if (i >= 0 && (uint64_t)i < dim) pos[i] = (int32_t)t;
}

void upload_scale(spsamd_ctx *c, const spsamd_vec *s, uint64_t dim, const char *name, ScaleDev *out)
{
	out->present = false;
	if (!s) return;
	out->present = true;
	out->dim = dim;
	size_t n = s->nnz;
	const int32_t *di = to_device(c, s->idx, n, s->mem);
	out->val = to_device(c, s->val, n, s->mem);
	out->pos = c->arena.get<int32_t>(dim ? dim : 1);
	fill_u32(c, (uint32_t *)out->pos, 0xFFFFFFFFu, dim);
	if (n == 0) return;
	uint32_t *bad = c->arena.get<uint32_t>(1);
	fill_zero(c, bad, sizeof(uint32_t));
	k_scale_scatter<<<dim3(grid_for(n)), dim3(256), 0, c->stream>>>(di, n, dim, out->pos, bad);
	SPS_LAUNCH_CHECK();
	if (read_back(c, bad))
		throw Error{SPSAMD_EINVAL, std::string("scale vector ") + name + " is not strictly ascending in its index "
			"(the reference joins scale vectors as stored and would silently mis-compute, multiply_sparse.hpp:83-85)"};
}

} // namespace spsamd
