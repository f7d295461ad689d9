// capi.hip -- the extern "C" boundary declared in include/spsparse_amd.h.
// Argument handling mirrors spsparse::multiply (multiply_sparse.hpp:152-248):
// shape first (:169), inner-dimension check (:172-174), short-circuits
// (:178-184), consolidation of both operands (:187-188), then the product.
#include "internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <vector>

using namespace spsamd;

#define API_GUARD(ctx, ...)                                                         \
	try { __VA_ARGS__ }                                                             \
	catch (const spsamd::Error &e) { (ctx)->last_error = e.msg; return e.code; }    \
	catch (const std::bad_alloc &) { (ctx)->last_error = "host allocation failed"; return SPSAMD_ENOMEM; } \
	catch (const std::exception &e) { (ctx)->last_error = e.what(); return SPSAMD_EINVAL; }

extern "C" const char *spsamd_version(void) { return "spsparse_amd 0.2 (gfx950)"; }

extern "C" int spsamd_ctx_create(spsamd_ctx **out, int device, void *hip_stream)
{
	if (!out) return SPSAMD_EINVAL;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return SPSAMD_ENODEVICE; }
	if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return SPSAMD_ENODEVICE; }
	if (device >= ndev) return SPSAMD_ENODEVICE;
	if (hipSetDevice(device) != hipSuccess) return SPSAMD_ENODEVICE;
	spsamd_ctx *c = new (std::nothrow) spsamd_ctx();
	if (!c) return SPSAMD_ENOMEM;
	c->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
	if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
	else {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SPSAMD_EHIP; }
		c->own_stream = true;
	}
	// (from here on a failure hands the context to spsamd_ctx_destroy, which releases whatever exists already)
	for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	for (auto &e : c->ev2) if (hipEventCreate(&e) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	for (auto &e : c->ev_side) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	if (hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	for (auto &e : c->ev_side2) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { spsamd_ctx_destroy(c); return SPSAMD_EHIP; }
	// developer knobs: the environment is consulted here and nowhere else
	static const char *const knobs[] = {"window", "cell_cap", "dense_min", "no_tiles", "xcd", "emit_path", "light_path", "no_wmajor", "direct_min", "tiles_v1", "long_cap", "long_dense_min", "index_budget_mb", "trace", "light_two_pass"};
	static const char *const envs[] = {"SPSAMD_W", "SPSAMD_CELL_CAP", "SPSAMD_DENSE_MIN", "SPSAMD_NO_TILES", "SPSAMD_XCD", "SPSAMD_EMIT_PATH", "SPSAMD_LIGHT_PATH", "SPSAMD_NO_WMAJOR", "SPSAMD_DIRECT_MIN", "SPSAMD_TILES_V1", "SPSAMD_LONG_CAP", "SPSAMD_LONG_DENSE_MIN", "SPSAMD_INDEX_BUDGET_MB", "SPSAMD_TRACE", "SPSAMD_LIGHT_TWO_PASS"};
	static_assert(sizeof knobs / sizeof knobs[0] == sizeof envs / sizeof envs[0], "one environment variable per knob");
	for (size_t k = 0; k < sizeof knobs / sizeof knobs[0]; ++k)
		if (const char *e = getenv(envs[k])) (void)spsamd_ctx_set_tuning(c, knobs[k], atol(e));
#ifdef SPSAMD_ABLATIONS
	if (const char *e = getenv("SPSAMD_DBG")) c->tune.dbg = atoi(e);
#endif
	*out = c;
	return SPSAMD_OK;
}

extern "C" int spsamd_ctx_set_tuning(spsamd_ctx *c, const char *name, long value)
{
	if (!c || !name) return SPSAMD_EINVAL;
	struct { const char *n; int *p; } tab[] = {
		{"window", &c->tune.window}, {"cell_cap", &c->tune.cell_cap}, {"dense_min", &c->tune.dense_min},
		{"no_tiles", &c->tune.no_tiles}, {"xcd", &c->tune.xcd}, {"emit_path", &c->tune.emit_path},
		{"light_path", &c->tune.light_path}, {"no_wmajor", &c->tune.no_wmajor}, {"direct_min", &c->tune.direct_min}, {"tiles_v1", &c->tune.tiles_v1}, {"long_cap", &c->tune.long_cap}, {"long_dense_min", &c->tune.long_dense_min}, {"index_budget_mb", &c->tune.index_budget_mb}, {"trace", &c->tune.trace}, {"light_two_pass", &c->tune.light_two_pass},
	};
	for (auto &t : tab) if (!std::strcmp(t.n, name)) { *t.p = (int)value; return SPSAMD_OK; }
	c->last_error = std::string("unknown tuning knob: ") + name;
	return SPSAMD_EINVAL;
}

extern "C" void spsamd_ctx_destroy(spsamd_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->side) (void)hipStreamSynchronize(c->side);
	if (c->side2) (void)hipStreamSynchronize(c->side2);
	c->arena.release();
	c->out[0].release(); c->out[1].release();
	c->rowstat_n.release(); c->rowstat_s.release(); c->rowstat_h.release();
	if (c->pinned) (void)hipHostFree(c->pinned);
	for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : c->ev2) if (e) (void)hipEventDestroy(e);
	for (auto &e : c->ev_side) if (e) (void)hipEventDestroy(e);
	for (auto &e : c->ev_side2) if (e) (void)hipEventDestroy(e);
	if (c->side) (void)hipStreamDestroy(c->side);
	if (c->side2) (void)hipStreamDestroy(c->side2);
	if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" const char *spsamd_last_error(const spsamd_ctx *c) { return c ? c->last_error.c_str() : "null context"; }

extern "C" int spsamd_ctx_reserve(spsamd_ctx *c, size_t workspace_bytes, size_t output_tuples)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		SPS_HIP(hipSetDevice(c->device));
		if (workspace_bytes) c->arena.reserve(workspace_bytes);
		if (output_tuples) {
			OutSet &o = c->out[c->cur_out];
			o.i.ensure(output_tuples * sizeof(int32_t));
			o.j.ensure(output_tuples * sizeof(int32_t));
			o.v.ensure(output_tuples * sizeof(double));
		}
		return SPSAMD_OK;
	)
}

void spsamd::pick_output_set(spsamd_ctx *c, const spsamd_coo *const *operands, int n)
{
	auto aliased = [&](int s) {
		for (int k = 0; k < n; ++k) {
			const spsamd_coo *X = operands[k];
			if (X && X->mem == SPSAMD_MEM_DEVICE && (c->out[s].holds(X->idx0) || c->out[s].holds(X->idx1) || c->out[s].holds(X->val))) return true;
		}
		return false;
	};
	if (!aliased(c->cur_out)) return;
	if (aliased(c->cur_out ^ 1))
		throw Error{SPSAMD_EINVAL, "both result buffers of this context are operands of the call: copy one of them out first (spsamd_memcpy)"};
	c->cur_out ^= 1;
}

static bool same_operand(const spsamd_coo *a, const spsamd_coo *b)
{
	return a->idx0 == b->idx0 && a->idx1 == b->idx1 && a->val == b->val && a->nnz == b->nnz &&
		a->shape0 == b->shape0 && a->shape1 == b->shape1 && a->sort0 == b->sort0 && a->mem == b->mem;
}

// Shared body of the MM and MV entry points.  `what` names the right operand in
// the inner-dimension message ("B" or "V", multiply_sparse.hpp:173,299).
int spsamd::multiply_body(spsamd_ctx *c, double C,
	const spsamd_vec *scalei, const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej, const spsamd_coo *B, char transpose_B,
	const spsamd_vec *scalek, int duplicate_policy, int zero_nan,
	int sink_kind, int sink_flags, spsamd_result *res, const char *what, bool arena_ready, const OperandParts *parts)
{
	if (duplicate_policy < 0 || duplicate_policy > 2) throw Error{SPSAMD_EINVAL, "bad duplicate_policy"};
	if (sink_kind != SPSAMD_SINK_COO && sink_kind != SPSAMD_SINK_DIGEST) throw Error{SPSAMD_EINVAL, "bad sink_kind"};
	std::memset(res, 0, sizeof(*res));
	// multiply_sparse.hpp:167-169: op(A) rows = A.shape[a0]; op(B) is read by ROWS here
	// (inner index first), its columns are B.shape[bj]
	const int a0 = transpose_A == 'T' ? 1 : 0, a1 = 1 - a0;
	const int bk = transpose_B == 'T' ? 1 : 0, bj = 1 - bk;
	const size_t ashape[2] = {A->shape0, A->shape1}, bshape[2] = {B->shape0, B->shape1};
	const bool permute = (sink_flags & SPSAMD_SINK_PERMUTE) && sink_kind == SPSAMD_SINK_COO;
	res->shape0 = permute ? bshape[bj] : ashape[a0];
	res->shape1 = permute ? ashape[a0] : bshape[bj];
	if (ashape[a1] != bshape[bk]) {                                  // :172-174
		char buf[160];
		std::snprintf(buf, sizeof buf, "Inner dimensions for A (%ld) and %s (%ld) must match!", (long)ashape[a1], what, (long)bshape[bk]);
		throw Error{SPSAMD_EDIM, buf};
	}
	if (C == 0 || (scalei && scalei->nnz == 0) || A->nnz == 0 || (scalej && scalej->nnz == 0) ||
		B->nnz == 0 || (scalek && scalek->nnz == 0))                 // :178-184
		return SPSAMD_OK;

	SPS_HIP(hipSetDevice(c->device));
	if (!arena_ready) c->arena.reset();
	hipStream_t st = c->stream;
	SPS_HIP(hipEventRecord(c->ev[0], st));
	// (also for the digest sink: a product by column blocks, spgemm_column_blocks, goes through the COO buffers)
	{ const spsamd_coo *ops[2] = {A, B}; pick_output_set(c, ops, 2); }
	MultiplyArgs a;
	a.C = C; a.sink_kind = sink_kind; a.sink_flags = sink_flags;
	if (parts) { a.pa = parts->pa; a.pb = parts->pb; a.b_ready = parts->b_ready; }
	Prepared *hp = nullptr;
	consolidate_operand(c, A, a0, a0, duplicate_policy, zero_nan, &a.A, &hp);      // :187
	if (hp) a.pa = hp;
	// A*A: one consolidation serves both -- except under zero_nan, where the NaNs dropped from B are those
	// of the leading run of the reference's column-major sequence (:168), not of A's row-major one
	// (a prepared operand is taken as it was prepared, like any operand that carries the wanted sort order)
	if (a0 == bk && same_operand(A, B) && (!zero_nan || hp)) { a.B = a.A; if (a.pa) a.pb = a.pa; }
	else { consolidate_operand(c, B, bk, bj, duplicate_policy, zero_nan, &a.B, &hp); if (hp) a.pb = hp; }  // :188
	if (sink_kind == SPSAMD_SINK_COO) c->own[c->cur_out].sort0 = -1;           // that set is about to be overwritten
	upload_scale(c, scalei, ashape[a0], "scalei", &a.si);
	upload_scale(c, scalej, ashape[a1], "scalej", &a.sj);
	upload_scale(c, scalek, bshape[bj], "scalek", &a.sk);
	spgemm(c, a, res);
	if (sink_kind == SPSAMD_SINK_COO && res->idx0 && res->idx1) {
		// row-major sorted, every (i, j) once, no zero: consolidated by sort order {0, 1} (read permuted: by {1, 0})
		auto &o = c->own[c->cur_out];
		o.d0 = permute ? res->idx1 : res->idx0; o.d1 = permute ? res->idx0 : res->idx1; o.v = res->val; o.nnz = res->nnz;
		o.shape0 = res->shape0; o.shape1 = res->shape1; o.sort0 = permute ? 1 : 0;
	}
	if (permute) std::swap(res->idx0, res->idx1);                             // PermuteAccum {1,0}: same tuples, indices swapped
	SPS_HIP(hipEventRecord(c->ev[7], st));
	SPS_HIP(hipEventSynchronize(c->ev[7]));
	if (res->nnz_a && res->nnz_b) {
		SPS_HIP(hipEventElapsedTime(&res->ms_consolidate, c->ev[0], c->ev[1]));
	}
	SPS_HIP(hipEventElapsedTime(&res->ms_total, c->ev[0], c->ev[7]));
	res->workspace_bytes = c->arena.call_used;
	return SPSAMD_OK;
}

extern "C" int spsamd_multiply(spsamd_ctx *c, double C,
	const spsamd_vec *scalei, const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej, const spsamd_coo *B, char transpose_B,
	const spsamd_vec *scalek, int duplicate_policy, int zero_nan,
	int sink_kind, int sink_flags, spsamd_result *res)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!A || !B || !res) throw Error{SPSAMD_EINVAL, "null operand or result"};
		return multiply_body(c, C, scalei, A, transpose_A, scalej, B, transpose_B, scalek, duplicate_policy, zero_nan,
			sink_kind, sink_flags, res, "B", false);
	)
}

// Matrix x sparse vector (multiply_sparse.hpp:281-365): V is the one-column
// matrix (idx, 0) of shape (V.shape0, 1); consolidating it row-major is
// Consolidate<VecT>(&V, {0}) (:313).  The result's idx1 is all zero.
extern "C" int spsamd_multiply_mv(spsamd_ctx *c, double C,
	const spsamd_vec *scalei, const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej, const spsamd_vec *V,
	int duplicate_policy, int zero_nan, int sink_kind, int sink_flags, spsamd_result *res)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!A || !V || !res) throw Error{SPSAMD_EINVAL, "null operand or result"};
		SPS_HIP(hipSetDevice(c->device));
		c->arena.reset();
		spsamd_coo Vm;
		Vm.nnz = V->nnz; Vm.shape0 = V->shape0; Vm.shape1 = 1;
		Vm.sort0 = V->sort0 == 0 ? 0 : -1;
		Vm.mem = SPSAMD_MEM_DEVICE;
		Vm.idx0 = nullptr; Vm.idx1 = nullptr; Vm.val = nullptr;
		if (V->nnz) {
			if (!V->idx || !V->val) throw Error{SPSAMD_EINVAL, "vector with nnz > 0 has a null array"};
			int32_t *vi = c->arena.get<int32_t>(V->nnz), *vz = c->arena.get<int32_t>(V->nnz);
			double *vv = c->arena.get<double>(V->nnz);
			hipMemcpyKind kind = V->mem == SPSAMD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
			SPS_HIP(hipMemcpyAsync(vi, V->idx, V->nnz * sizeof(int32_t), kind, c->stream));
			SPS_HIP(hipMemcpyAsync(vv, V->val, V->nnz * sizeof(double), kind, c->stream));
			fill_zero(c, vz, V->nnz * sizeof(int32_t));
			Vm.idx0 = vi; Vm.idx1 = vz; Vm.val = vv;
		}
		int rc = multiply_body(c, C, scalei, A, transpose_A, scalej, &Vm, '.', nullptr, duplicate_policy, zero_nan,
			sink_kind, sink_flags & ~SPSAMD_SINK_PERMUTE, res, "V", true);     // a rank-1 result has nothing to permute
		res->shape1 = 0;                              // rank-1 result: ret.set_shape({rows}) (:295)
		res->idx1 = nullptr;                          // ... with one index array: nothing to copy for a second one
		return rc;
	)
}

// ---- prepared operands -------------------------------------------------------------------------------------
// The reference keeps what it derives from an operand: an array that carries the wanted sort_order is not consolidated
// again (Consolidate<>, algorithm.hpp:360) and its row structure is computed once and cached in the object
// (VectorCooArray::dim_beginnings, VectorCooArray.hpp:325-335).  spsamd_operand is that object on the device.

struct spsamd_operand {
	spsamd::Prepared p;
	uint64_t shape0 = 0, shape1 = 0;
};

extern "C" int spsamd_operand_prepare(spsamd_ctx *c, const spsamd_coo *X, char transpose, int role, int duplicate_policy, int zero_nan,
	spsamd_operand **out)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!X || !out) throw Error{SPSAMD_EINVAL, "null operand or result"};
		*out = nullptr;
		if (role != SPSAMD_AS_A && role != SPSAMD_AS_B && role != (SPSAMD_AS_A | SPSAMD_AS_B)) throw Error{SPSAMD_EINVAL, "bad role"};
		if (duplicate_policy < 0 || duplicate_policy > 2) throw Error{SPSAMD_EINVAL, "bad duplicate_policy"};
		if (X->mem == SPSAMD_MEM_PREPARED) throw Error{SPSAMD_EINVAL, "the operand is a prepared one already"};
		if (zero_nan && role == (SPSAMD_AS_A | SPSAMD_AS_B))
			throw Error{SPSAMD_EINVAL, "under zero_nan the two sides of a product drop different NaNs (multiply_sparse.hpp:167-168): prepare one operand per side"};
		SPS_HIP(hipSetDevice(c->device));
		c->arena.reset();
		// the leading dimension multiply consolidates this side by, and the one the REFERENCE's own sequence leads with
		// (they differ for B: multiply_sparse.hpp:168)
		const int lead = transpose == 'T' ? 1 : 0;
		const int ref_lead = (role & SPSAMD_AS_A) ? lead : 1 - lead;
		ConMat m;
		consolidate_operand(c, X, lead, ref_lead, duplicate_policy, zero_nan, &m);
		spsamd_operand *h = new spsamd_operand();
		struct Guard { spsamd_operand *h; ~Guard() { if (h) { h->p.release(); delete h; } } } guard{h};
		h->shape0 = X->shape0; h->shape1 = X->shape1;
		Prepared &p = h->p;
		p.ctx = c; p.owns = true; p.lead = lead;
		p.m = m;
		const size_t n = m.nnz;
		// All of the handle's memory is taken NOW, in one piece: tuples, packed tuples, row list, row pointer -- and, for a right
		// operand with long rows, the column-window indices the first product with a heavy row will build.  (Measured on cfg2:
		// taken later and separately, after the context's workspace exists, the same arrays make the kernels that gather from
		// them at random -- the hash cells of the long rows -- 2.3x slower; presumably the fragments the driver then finds are
		// smaller.)  Not for operands whose indices would be out of proportion: a stencil matrix has no heavy rows and
		// millions of columns.
		size_t want = (n + 64) * (16 + 12 + 8) + (m.nrow + 66) * 4 + 65536;
		uint32_t maxlen = 0;
		if (n) {
			Prepared view;                                              // (the consolidated tuples still sit in the workspace)
			view.ctx = c; view.m = m;
			prepared_row_structure(c, &view);
			maxlen = view.maxlen;
		}
		if ((role & SPSAMD_AS_B) && maxlen > 64) {
			const uint64_t W = m.ncol > (uint64_t(1) << 21) ? 16384 : 8192, nwin = (m.ncol + W - 1) / W, nwp = (nwin + 7) & ~7ull, nrowb = m.nrow + 1;
			const uint64_t idx = nrowb * (nwin + 1) * 4 + nrowb * nwp * 2 + (nrowb * nwin + 1) * 4 + (n + 8) * 12 + 8192;
			size_t freeb = 0, totalb = 0;
			if (idx <= 64 * (uint64_t)n * 16 && hipMemGetInfo(&freeb, &totalb) == hipSuccess && idx + want < freeb / 2) want += idx;
		}
		p.reserve(want);
		p.m.row = p.get<int32_t>(n ? n : 1); p.m.col = p.get<int32_t>(n ? n : 1); p.m.val = p.get<double>(n ? n : 1);
		if (n) {
			SPS_HIP(hipMemcpyAsync(p.m.row, m.row, n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
			SPS_HIP(hipMemcpyAsync(p.m.col, m.col, n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
			SPS_HIP(hipMemcpyAsync(p.m.val, m.val, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
		}
		prepared_row_structure(c, &p);                                  // the row pointer and the longest row (every product asks for them)
		SPS_HIP(hipStreamSynchronize(c->stream));
		guard.h = nullptr;
		*out = h;
		return SPSAMD_OK;
	)
}

extern "C" int spsamd_operand_as_coo(const spsamd_operand *h, spsamd_coo *out)
{
	if (!h || !out) return SPSAMD_EINVAL;
	out->idx0 = (const int32_t *)h; out->idx1 = nullptr; out->val = nullptr;
	out->nnz = h->p.m.nnz; out->shape0 = h->shape0; out->shape1 = h->shape1;
	out->sort0 = h->p.lead; out->mem = SPSAMD_MEM_PREPARED;
	return SPSAMD_OK;
}

extern "C" uint64_t spsamd_operand_bytes(const spsamd_operand *h) { return h ? h->p.owned_bytes : 0; }

extern "C" void spsamd_operand_destroy(spsamd_operand *h)
{
	if (!h) return;
	if (h->p.ctx) { (void)hipSetDevice(h->p.ctx->device); (void)hipStreamSynchronize(h->p.ctx->stream); if (h->p.ctx->side) (void)hipStreamSynchronize(h->p.ctx->side); }
	h->p.release();
	delete h;
}

// The stand-alone algorithms read an operand's arrays themselves: a prepared operand is handed to them as the device
// COO it holds (sorted by its lead).
static const spsamd_coo *plain_operand(spsamd_ctx *c, const spsamd_coo *X, spsamd_coo *tmp)
{
	if (!X || X->mem != SPSAMD_MEM_PREPARED) return X;
	const spsamd_operand *h = (const spsamd_operand *)X->idx0;
	if (!h || h->p.ctx != c) throw Error{SPSAMD_EINVAL, "a prepared operand belongs to the context that prepared it"};
	const Prepared &p = h->p;
	tmp->idx0 = p.lead == 0 ? p.m.row : p.m.col; tmp->idx1 = p.lead == 0 ? p.m.col : p.m.row; tmp->val = p.m.val;
	tmp->nnz = p.m.nnz; tmp->shape0 = h->shape0; tmp->shape1 = h->shape1; tmp->sort0 = p.lead; tmp->mem = SPSAMD_MEM_DEVICE;
	return tmp;
}

extern "C" int spsamd_result_fetch(spsamd_ctx *c, const spsamd_result *res, spsamd_chunk_fn cb, void *user)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!res || !cb) throw Error{SPSAMD_EINVAL, "null result or callback"};
		if (res->nnz == 0) return SPSAMD_OK;
		if (!res->idx0 || !res->val) throw Error{SPSAMD_EINVAL, "result has no COO tuples (DIGEST sink?)"};
		SPS_HIP(hipSetDevice(c->device));
		// chunks of 2^20 tuples through two pinned staging buffers: the copy of chunk k+1 runs while the
		// callback consumes chunk k (the callback, one ret.add() per tuple upstream, is the slow side)
		const size_t chunk = size_t(1) << 20;
		char *h = (char *)c->host_staging(2 * chunk * 16);
		hipEvent_t ev[2] = {c->ev[8], c->ev[9]};
		auto issue = [&](uint64_t o, int b) {
			size_t n = (size_t)std::min<uint64_t>(chunk, res->nnz - o);
			char *hb = h + (size_t)b * chunk * 16;
			SPS_HIP(hipMemcpyAsync(hb, res->idx0 + o, n * 4, hipMemcpyDeviceToHost, c->stream));
			if (res->idx1) SPS_HIP(hipMemcpyAsync(hb + chunk * 4, res->idx1 + o, n * 4, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipMemcpyAsync(hb + chunk * 8, res->val + o, n * 8, hipMemcpyDeviceToHost, c->stream));
			SPS_HIP(hipEventRecord(ev[b], c->stream));
		};
		issue(0, 0);
		int b = 0;
		for (uint64_t o = 0; o < res->nnz; o += chunk, b ^= 1) {
			size_t n = (size_t)std::min<uint64_t>(chunk, res->nnz - o);
			if (o + chunk < res->nnz) issue(o + chunk, b ^ 1);
			SPS_HIP(hipEventSynchronize(ev[b]));
			char *hb = h + (size_t)b * chunk * 16;
			int rc = cb(user, (int32_t *)hb, res->idx1 ? (int32_t *)(hb + chunk * 4) : nullptr, (double *)(hb + chunk * 8), n);
			if (rc) { SPS_HIP(hipStreamSynchronize(c->stream)); return rc; }
		}
		return SPSAMD_OK;
	)
}

__global__ void k_scatter_dense(const int32_t *i, const int32_t *j, const double *v, uint64_t n, double *dense, size_t ld, int policy)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; t < n; t += stride) {
		double *p = dense + (size_t)i[t] * ld + (size_t)j[t];      // (i, j) is unique inside one result
		if (policy == SPSAMD_ADD) *p += v[t];
		else if (policy == SPSAMD_REPLACE) *p = v[t];
		else if (!(*p != *p)) *p = v[t];                          // LEAVE_ALONE exactly as accum.hpp:128-130 spells it: `if (!std::isnan(oval)) oval = val`
	}
}

extern "C" int spsamd_result_scatter_dense(spsamd_ctx *c, const spsamd_result *res, double *dense, size_t ld, int policy)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!res || (res->nnz && (!dense || !res->idx0 || !res->idx1 || !res->val))) throw Error{SPSAMD_EINVAL, "null result, matrix or no COO tuples"};
		if (policy < 0 || policy > 2) throw Error{SPSAMD_EINVAL, "bad duplicate_policy"};
		if (ld < res->shape1) throw Error{SPSAMD_EINVAL, "leading dimension smaller than the result's column count"};
		if (res->nnz == 0) return SPSAMD_OK;
		SPS_HIP(hipSetDevice(c->device));
		unsigned grid = (unsigned)std::min<uint64_t>((res->nnz + 255) / 256, 8192);
		k_scatter_dense<<<dim3(grid), dim3(256), 0, c->stream>>>(res->idx0, res->idx1, res->val, res->nnz, dense, ld, policy);
		SPS_LAUNCH_CHECK();
		SPS_HIP(hipStreamSynchronize(c->stream));
		return SPSAMD_OK;
	)
}

extern "C" int spsamd_memcpy(spsamd_ctx *c, void *dst, const void *src, size_t bytes)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (bytes && (!dst || !src)) throw Error{SPSAMD_EINVAL, "null pointer"};
		SPS_HIP(hipSetDevice(c->device));
		SPS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
		SPS_HIP(hipStreamSynchronize(c->stream));
		return SPSAMD_OK;
	)
}

extern "C" int spsamd_consolidate(spsamd_ctx *c, const spsamd_coo *A, int so0, int duplicate_policy, int zero_nan,
	spsamd_result *res)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!A || !res || (so0 != 0 && so0 != 1)) throw Error{SPSAMD_EINVAL, "bad argument"};
		if (duplicate_policy < 0 || duplicate_policy > 2) throw Error{SPSAMD_EINVAL, "bad duplicate_policy"};
		spsamd_coo plain;
		A = plain_operand(c, A, &plain);
		std::memset(res, 0, sizeof(*res));
		res->shape0 = A->shape0; res->shape1 = A->shape1;
		if (A->nnz == 0) return SPSAMD_OK;
		SPS_HIP(hipSetDevice(c->device));
		c->arena.reset();
		spsamd_coo X = *A;
		X.sort0 = -1;                         // the stand-alone algorithm always runs (algorithm.hpp:251)
		ConMat m;
		consolidate_operand(c, &X, so0, so0, duplicate_policy, zero_nan, &m);
		size_t n = m.nnz;
		const spsamd_coo *ops[1] = {A};
		pick_output_set(c, ops, 1);
		OutSet &o = c->out[c->cur_out];
		c->own[c->cur_out].sort0 = -1;
		o.i.ensure(n * 4 + 4); o.j.ensure(n * 4 + 4); o.v.ensure(n * 8 + 8);
		// m.row is the leading (sorted) dimension: put dimensions back in place
		int32_t *d0 = (int32_t *)o.i.p, *d1 = (int32_t *)o.j.p;
		SPS_HIP(hipMemcpyAsync(so0 == 0 ? d0 : d1, m.row, n * 4, hipMemcpyDeviceToDevice, c->stream));
		SPS_HIP(hipMemcpyAsync(so0 == 0 ? d1 : d0, m.col, n * 4, hipMemcpyDeviceToDevice, c->stream));
		SPS_HIP(hipMemcpyAsync(o.v.p, m.val, n * 8, hipMemcpyDeviceToDevice, c->stream));
		SPS_HIP(hipStreamSynchronize(c->stream));
		res->nnz = n; res->nnz_a = n;
		res->idx0 = d0; res->idx1 = d1; res->val = (double *)o.v.p;
		{ auto &w = c->own[c->cur_out]; w.d0 = d0; w.d1 = d1; w.v = res->val; w.nnz = n; w.shape0 = A->shape0; w.shape1 = A->shape1; w.sort0 = so0; }
		return SPSAMD_OK;
	)
}

extern "C" int spsamd_sorted_permutation(spsamd_ctx *c, const spsamd_coo *A, int so0, uint64_t *perm_host)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!A || (so0 != 0 && so0 != 1) || (A->nnz && !perm_host)) throw Error{SPSAMD_EINVAL, "bad argument"};
		if (A->nnz == 0) return SPSAMD_OK;
		spsamd_coo plain;
		A = plain_operand(c, A, &plain);
		SPS_HIP(hipSetDevice(c->device));
		c->arena.reset();
		uint32_t *perm = sorted_permutation(c, A, so0);
		std::vector<uint32_t> h(A->nnz);
		SPS_HIP(hipMemcpyAsync(h.data(), perm, A->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		SPS_HIP(hipStreamSynchronize(c->stream));
		for (size_t i = 0; i < A->nnz; ++i) perm_host[i] = h[i];
		return SPSAMD_OK;
	)
}

extern "C" int spsamd_dim_beginnings(spsamd_ctx *c, const spsamd_coo *A, int so0, uint64_t *beg_host, size_t *count)
{
	if (!c) return SPSAMD_EINVAL;
	API_GUARD(c,
		if (!A || !count || (so0 != 0 && so0 != 1)) throw Error{SPSAMD_EINVAL, "bad argument"};
		spsamd_coo plain;
		A = plain_operand(c, A, &plain);
		if (A->sort0 != so0) throw Error{SPSAMD_EINVAL, "dim_beginnings() required the VectorCooArray is sorted first."};
		*count = 0;
		if (A->nnz == 0) return SPSAMD_OK;                         // algorithm.hpp:89
		if (!beg_host) throw Error{SPSAMD_EINVAL, "null output"};
		SPS_HIP(hipSetDevice(c->device));
		c->arena.reset();
		ConMat m;
		consolidate_operand(c, A, so0, so0, SPSAMD_ADD, 0, &m);         // trusted as is (sort0 == so0): upload only
		RowList rl;
		dim_beginnings(c, m, &rl);
		std::vector<uint32_t> h((size_t)rl.nrows + 1);
		SPS_HIP(hipMemcpyAsync(h.data(), rl.beg, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		SPS_HIP(hipStreamSynchronize(c->stream));
		for (size_t i = 0; i < h.size(); ++i) beg_host[i] = h[i];
		*count = h.size();
		return SPSAMD_OK;
	)
}
