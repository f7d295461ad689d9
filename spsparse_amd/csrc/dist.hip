// dist.hip -- the multi-GPU step behind the C ABI (include/spsparse_amd.h, spsamd_dist_*).
//
// The reference is single threaded and has no counterpart.  What makes the path shardable is its own loop
// structure: output row i depends only on row i of op(A) and the B rows {k : A(i,k) != 0}
// (multiply_sparse.hpp:192), so A is cut into contiguous row blocks, one per GPU, B is distributed by
// contiguous row blocks over the inner dimension, and ONE exchange step brings every rank the B rows its
// block needs.  No reduction: C stays row partitioned.
//
// One step on every rank (all on the device, the context's stream):
//   1. consolidate the own A block (and the own B block, unless B is A);
//   2. need mask: one byte per inner index k that occurs in the A block;  exchange A: every owner learns which of
//      ITS rows each peer needs;
//   3. per peer: masked row lengths of the own B rows, their prefix, and (K7) the pack kernel:
//      (col, val) of the needed rows, 12 bytes per tuple, no row array -- an owner all of whose tuples are
//      needed sends its block as it stands;  exchange B: the row lengths;  exchange C: the tuples, received
//      straight into the panel at the owner's offset, so the panel arrives row-major sorted;
//   4. spsamd_multiply of the A block with the panel (both consolidated, trusted as they are).
// Transport: grouped ncclSend / ncclRecv (RCCL over xGMI) on the context's stream -- librccl is loaded at run
// time, only here -- or a caller-supplied all-to-allv (tests drive the same code over gloo on one GPU).
#include "internal.h"
#include "devutil.h"

#include <dlfcn.h>
#include <cstring>
#include <algorithm>
#include <vector>

using namespace spsamd;

namespace {

// ---- the handful of RCCL entry points used, resolved with dlsym (no link-time dependency) ----------------
struct UidBytes { char internal[128]; };

struct RcclApi {
	void *lib = nullptr;
	int (*GetUniqueId)(UidBytes *) = nullptr;
	int (*CommInitRank)(void **, int, UidBytes, int) = nullptr;
	int (*CommDestroy)(void *) = nullptr;
	int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};

RcclApi *rccl()
{
	static RcclApi api;
	static bool tried = false;
	if (tried) return api.lib ? &api : nullptr;
	tried = true;
	// a copy a host framework already loaded wins (one RCCL per process), then the ROCm installation's
	for (const char *name : {"librccl.so", "librccl.so.1"}) {
		api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
		if (api.lib) break;
	}
	if (!api.lib) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
		api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (api.lib) break;
	}
	if (!api.lib) return nullptr;
	bool ok = true;
	auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p) ok = false; return p; };
	api.GetUniqueId = (int (*)(UidBytes *))sym("ncclGetUniqueId");
	api.CommInitRank = (int (*)(void **, int, UidBytes, int))sym("ncclCommInitRank");
	api.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
	api.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
	api.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
	api.GroupStart = (int (*)())sym("ncclGroupStart");
	api.GroupEnd = (int (*)())sym("ncclGroupEnd");
	api.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
	if (!ok) { api.lib = nullptr; return nullptr; }
	return &api;
}

constexpr int NCCL_UINT8 = 1;      // ncclUint8 (ncclDataType_t, stable across NCCL / RCCL 2.x)

#define SPS_NCCL(call)                                                             \
	do {                                                                           \
		int e_ = (call);                                                           \
		if (e_ != 0) throw Error{SPSAMD_EHIP, std::string(#call) + ": " + (rccl() ? rccl()->GetErrorString(e_) : "RCCL error")}; \
	} while (0)

unsigned grid_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// ---- kernels ---------------------------------------------------------------------------------------------

__global__ void k_mark_need(const int32_t *acol, uint32_t n, uint8_t *need)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e < n) need[acol[e]] = 1;                                    // same value from every writer
}

// masked row lengths of the own rows for one peer
__global__ void k_masked_rowlen(const uint32_t *ptr, uint64_t my_lo, uint32_t my_n, const uint8_t *mask, uint32_t *rowlen)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < my_n) rowlen[j] = mask[j] ? ptr[my_lo + j + 1] - ptr[my_lo + j] : 0u;
}

// K7: the needed rows' (col, val), row after row -- dst = masked prefix of the row + position inside the row
__global__ void k_pack_rows(const int32_t *brow, const int32_t *bcol, const double *bval, uint32_t first, uint32_t count,
	const uint32_t *ptr, uint64_t my_lo, const uint8_t *mask, const uint32_t *off, int32_t *pcol, double *pval)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= count) return;
	const uint32_t e = first + t;
	const uint32_t r = (uint32_t)brow[e];
	const uint32_t j = (uint32_t)(r - my_lo);
	if (!mask[j]) return;
	const uint32_t dst = off[j] + (e - ptr[r]);
	pcol[dst] = bcol[e];
	pval[dst] = bval[e];
}

__global__ void k_pick_u32(const uint32_t *src, const uint64_t *at, int n, uint32_t *dst)
{
	for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[at[i]];
}

// panel row index of every tuple from the panel's row pointer (the row array is never sent)
__global__ void k_expand_rows(const uint32_t *ptr, uint32_t nrow, uint32_t nnz, int32_t *row)
{
	uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= nnz) return;
	uint32_t lo = 0, hi = nrow;                                      // last row r with ptr[r] <= e
	while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (ptr[mid] <= e) lo = mid; else hi = mid; }
	row[e] = (int32_t)lo;
}

} // namespace

struct spsamd_dist {
	spsamd_ctx *ctx = nullptr;
	int rank = 0, world = 1;
	void *comm = nullptr;            // ncclComm_t (built-in transport)
	bool own_comm = false;
	spsamd_alltoallv_fn xfn = nullptr;
	void *xuser = nullptr;
	hipEvent_t ev[2] = {nullptr, nullptr};
};

// all-to-allv of device buffers through the chosen transport, on the context's stream
static void exchange(spsamd_dist *d, const std::vector<const void *> &send, const std::vector<size_t> &sendb,
	const std::vector<void *> &recv, const std::vector<size_t> &recvb)
{
	spsamd_ctx *c = d->ctx;
	if (d->xfn) {
		SPS_HIP(hipStreamSynchronize(c->stream));                    // the callback may touch the buffers from the host
		int rc = d->xfn(d->xuser, send.data(), sendb.data(), recv.data(), recvb.data(), d->world, (void *)c->stream);
		if (rc) throw Error{SPSAMD_EHIP, "the caller's all-to-allv transport failed"};
		return;
	}
	RcclApi *r = rccl();
	if (!r) throw Error{SPSAMD_EHIP, "librccl could not be loaded"};
	SPS_NCCL(r->GroupStart());
	for (int p = 0; p < d->world; ++p) {
		if (sendb[p]) SPS_NCCL(r->Send(send[p], sendb[p], NCCL_UINT8, p, d->comm, c->stream));
		if (recvb[p]) SPS_NCCL(r->Recv(recv[p], recvb[p], NCCL_UINT8, p, d->comm, c->stream));
	}
	SPS_NCCL(r->GroupEnd());
}

#define DIST_GUARD(ctx, ...)                                                        \
	try { __VA_ARGS__ }                                                             \
	catch (const spsamd::Error &e) { (ctx)->last_error = e.msg; return e.code; }    \
	catch (const std::bad_alloc &) { (ctx)->last_error = "host allocation failed"; return SPSAMD_ENOMEM; } \
	catch (const std::exception &e) { (ctx)->last_error = e.what(); return SPSAMD_EINVAL; }

extern "C" int spsamd_dist_unique_id(char id[128])
{
	RcclApi *r = rccl();
	if (!r || !id) return SPSAMD_EHIP;
	UidBytes u;
	if (r->GetUniqueId(&u) != 0) return SPSAMD_EHIP;
	std::memcpy(id, u.internal, 128);
	return SPSAMD_OK;
}

extern "C" int spsamd_dist_create(spsamd_dist **out, spsamd_ctx *ctx, int rank, int world, const char *unique_id,
	void *nccl_comm, spsamd_alltoallv_fn transport, void *transport_user)
{
	if (!out || !ctx || world < 1 || rank < 0 || rank >= world) return SPSAMD_EINVAL;
	*out = nullptr;
	DIST_GUARD(ctx,
		spsamd_dist *d = new spsamd_dist();
		d->ctx = ctx; d->rank = rank; d->world = world;
		SPS_HIP(hipSetDevice(ctx->device));
		for (auto &e : d->ev) SPS_HIP(hipEventCreate(&e));
		if (transport) { d->xfn = transport; d->xuser = transport_user; }
		else if (nccl_comm) d->comm = nccl_comm;
		else {
			RcclApi *r = rccl();
			if (!r) { delete d; throw Error{SPSAMD_EHIP, "librccl could not be loaded"}; }
			if (!unique_id) { delete d; throw Error{SPSAMD_EINVAL, "spsamd_dist_create needs a unique id, a communicator or a transport"}; }
			SPS_HIP(hipSetDevice(ctx->device));
			UidBytes u;
			std::memcpy(u.internal, unique_id, 128);
			int e = r->CommInitRank(&d->comm, world, u, rank);
			if (e != 0) { std::string msg = std::string("ncclCommInitRank: ") + r->GetErrorString(e); delete d; throw Error{SPSAMD_EHIP, msg}; }
			d->own_comm = true;
		}
		*out = d;
		return SPSAMD_OK;
	)
}

extern "C" void spsamd_dist_destroy(spsamd_dist *d)
{
	if (!d) return;
	if (d->own_comm && d->comm && rccl()) {
		(void)hipStreamSynchronize(d->ctx->stream);
		(void)rccl()->CommDestroy(d->comm);
	}
	for (auto &e : d->ev) if (e) (void)hipEventDestroy(e);
	delete d;
}

extern "C" int spsamd_dist_multiply(spsamd_dist *d, double C, const spsamd_coo *A_block, const spsamd_coo *B_block,
	const uint64_t *b_bounds, int duplicate_policy, int zero_nan, int sink_kind, int sink_flags,
	spsamd_result *res, spsamd_dist_stats *stats)
{
	if (!d || !d->ctx) return SPSAMD_EINVAL;
	spsamd_ctx *c = d->ctx;
	DIST_GUARD(c,
		if (!A_block || !b_bounds || !res) throw Error{SPSAMD_EINVAL, "null block, bounds or result"};
		const int W = d->world, me = d->rank;
		const uint64_t n_inner = A_block->shape1;
		const spsamd_coo *Bsrc = B_block ? B_block : A_block;
		if (Bsrc->shape0 != n_inner) {
			char buf[160];
			std::snprintf(buf, sizeof buf, "Inner dimensions for A (%ld) and B (%ld) must match!", (long)n_inner, (long)Bsrc->shape0);
			throw Error{SPSAMD_EDIM, buf};
		}
		if (b_bounds[0] != 0 || b_bounds[W] != n_inner) throw Error{SPSAMD_EINVAL, "b_bounds must run from 0 to the inner dimension"};
		for (int p = 0; p < W; ++p) if (b_bounds[p] > b_bounds[p + 1]) throw Error{SPSAMD_EINVAL, "b_bounds must be ascending"};
		if (n_inner >= (uint64_t(1) << 32)) throw Error{SPSAMD_EINVAL, "inner dimension exceeds 32 bits"};
		SPS_HIP(hipSetDevice(c->device));
		hipStream_t st = c->stream;
		c->arena.reset();
		hipEvent_t e0 = d->ev[0], e1 = d->ev[1];
		SPS_HIP(hipEventRecord(e0, st));

		// ---- 1. consolidate the own blocks
		ConMat Ac;
		consolidate_operand(c, A_block, 0, 0, duplicate_policy, zero_nan, &Ac);
		ConMat Bc = Ac;
		if (B_block) consolidate_operand(c, B_block, 0, 1, duplicate_policy, zero_nan, &Bc);
		const uint64_t my_lo = b_bounds[me];
		const uint32_t my_n = (uint32_t)(b_bounds[me + 1] - my_lo);

		// ---- 2. need mask and its exchange
		uint8_t *need = c->arena.get<uint8_t>(n_inner ? n_inner : 1);
		fill_zero(c, need, n_inner);
		if (Ac.nnz) { k_mark_need<<<dim3(grid_for(Ac.nnz)), dim3(256), 0, st>>>(Ac.col, Ac.nnz, need); SPS_LAUNCH_CHECK(); }
		uint8_t *their = c->arena.get<uint8_t>((size_t)my_n * W + 1);   // [peer][my row]: does the peer need it
		std::vector<const void *> sp(W); std::vector<void *> rp(W); std::vector<size_t> sb(W), rb(W);
		for (int p = 0; p < W; ++p) {
			sp[p] = need + b_bounds[p]; sb[p] = (size_t)(b_bounds[p + 1] - b_bounds[p]);
			rp[p] = their + (size_t)my_n * p; rb[p] = my_n;
		}
		exchange(d, sp, sb, rp, rb);

		// ---- 3. row lengths per peer, packed tuples
		uint32_t *ptr = nullptr;                                      // dense row pointer of the own B block (global row index)
		const uint32_t first = 0, nmine = Bc.nnz;                       // the whole block is "my rows"
		ptr = dense_rowptr(c, Bc, 0);
		uint32_t *rowlen = c->arena.get<uint32_t>((size_t)my_n * W + 1);
		uint32_t *off = c->arena.get<uint32_t>(((size_t)my_n + 1) * W);
		for (int p = 0; p < W; ++p) {
			if (my_n) { k_masked_rowlen<<<dim3(grid_for(my_n)), dim3(256), 0, st>>>(ptr, my_lo, my_n, their + (size_t)my_n * p, rowlen + (size_t)my_n * p); SPS_LAUNCH_CHECK(); }
			scan_exclusive_u32_u32(c, rowlen + (size_t)my_n * p, off + ((size_t)my_n + 1) * p, my_n);
		}
		std::vector<uint32_t> send_tuples(W);
		{
			// the W totals sit one prefix array apart: one strided copy, one synchronisation
			uint32_t *h = (uint32_t *)c->host_staging((size_t)W * sizeof(uint32_t));
			SPS_HIP(hipMemcpy2DAsync(h, sizeof(uint32_t), off + my_n, ((size_t)my_n + 1) * sizeof(uint32_t), sizeof(uint32_t), (size_t)W,
				hipMemcpyDeviceToHost, st));
			SPS_HIP(hipStreamSynchronize(st));
			for (int p = 0; p < W; ++p) send_tuples[p] = h[p];
		}
		// exchange B: the masked row lengths -> the panel's row lengths over the whole inner dimension
		uint32_t *plen = c->arena.get<uint32_t>(n_inner + 1);
		for (int p = 0; p < W; ++p) {
			sp[p] = rowlen + (size_t)my_n * p; sb[p] = (size_t)my_n * 4;
			rp[p] = plen + b_bounds[p]; rb[p] = (size_t)(b_bounds[p + 1] - b_bounds[p]) * 4;
		}
		exchange(d, sp, sb, rp, rb);
		uint32_t *pptr = c->arena.get<uint32_t>(n_inner + 1);
		scan_exclusive_u32_u32(c, plen, pptr, n_inner);
		std::vector<uint32_t> recv_at(W + 1);
		{
			uint32_t *picked = c->arena.get<uint32_t>((size_t)W + 1);
			uint64_t *at = c->arena.get<uint64_t>((size_t)W + 1);
			SPS_HIP(hipMemcpyAsync(at, b_bounds, ((size_t)W + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
			k_pick_u32<<<dim3(1), dim3(64), 0, st>>>(pptr, at, W + 1, picked);
			SPS_LAUNCH_CHECK();
			uint32_t *h = (uint32_t *)c->host_staging(((size_t)W + 1) * sizeof(uint32_t));
			SPS_HIP(hipMemcpyAsync(h, picked, ((size_t)W + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
			SPS_HIP(hipStreamSynchronize(st));
			for (int p = 0; p <= W; ++p) recv_at[p] = h[p];
		}
		const uint32_t pn = recv_at[W];                                  // tuples of the panel
		int32_t *prow = c->arena.get<int32_t>(pn ? pn : 1), *pcol = c->arena.get<int32_t>(pn ? pn : 1);
		double *pval = c->arena.get<double>(pn ? pn : 1);
		// pack + exchange C (columns, then values: two grouped rounds of 4 and 8 bytes per tuple)
		std::vector<const void *> scol(W), sval(W);
		for (int p = 0; p < W; ++p) {
			if (send_tuples[p] == nmine) { scol[p] = Bc.col + first; sval[p] = Bc.val + first; continue; }   // the whole block as it stands
			int32_t *qc = c->arena.get<int32_t>(send_tuples[p] ? send_tuples[p] : 1);
			double *qv = c->arena.get<double>(send_tuples[p] ? send_tuples[p] : 1);
			if (send_tuples[p]) {
				k_pack_rows<<<dim3(grid_for(nmine)), dim3(256), 0, st>>>(Bc.row, Bc.col, Bc.val, first, nmine, ptr, my_lo,
					their + (size_t)my_n * p, off + ((size_t)my_n + 1) * p, qc, qv);
				SPS_LAUNCH_CHECK();
			}
			scol[p] = qc; sval[p] = qv;
		}
		for (int p = 0; p < W; ++p) {
			sp[p] = scol[p]; sb[p] = (size_t)send_tuples[p] * 4;
			rp[p] = pcol + recv_at[p]; rb[p] = (size_t)(recv_at[p + 1] - recv_at[p]) * 4;
		}
		exchange(d, sp, sb, rp, rb);
		for (int p = 0; p < W; ++p) {
			sp[p] = sval[p]; sb[p] = (size_t)send_tuples[p] * 8;
			rp[p] = pval + recv_at[p]; rb[p] = (size_t)(recv_at[p + 1] - recv_at[p]) * 8;
		}
		exchange(d, sp, sb, rp, rb);
		if (pn) { k_expand_rows<<<dim3(grid_for(pn)), dim3(256), 0, st>>>(pptr, (uint32_t)n_inner, pn, prow); SPS_LAUNCH_CHECK(); }
		SPS_HIP(hipEventRecord(e1, st));

		// ---- 4. the block product: both operands consolidated, trusted as they are (sort0 = 0)
		spsamd_coo Ad, Bd;
		Ad.idx0 = Ac.row; Ad.idx1 = Ac.col; Ad.val = Ac.val; Ad.nnz = Ac.nnz; Ad.shape0 = A_block->shape0; Ad.shape1 = A_block->shape1;
		Ad.sort0 = 0; Ad.mem = SPSAMD_MEM_DEVICE_VERIFIED;
		Bd.idx0 = prow; Bd.idx1 = pcol; Bd.val = pval; Bd.nnz = pn; Bd.shape0 = Bsrc->shape0; Bd.shape1 = Bsrc->shape1;
		Bd.sort0 = 0; Bd.mem = SPSAMD_MEM_DEVICE_VERIFIED;
		if (stats) {
			std::memset(stats, 0, sizeof(*stats));
			stats->panel_tuples = pn;
			stats->remote_tuples = pn - (recv_at[me + 1] - recv_at[me]);
			for (int p = 0; p < W; ++p) if (p != me) stats->sent_tuples += send_tuples[p];
			stats->block_nnz_a = Ac.nnz;
		}
		int rc = multiply_body(c, C, nullptr, &Ad, '.', nullptr, &Bd, '.', nullptr, duplicate_policy, 0, sink_kind, sink_flags, res, "B", true);
		if (stats) {
			SPS_HIP(hipEventSynchronize(e1));
			SPS_HIP(hipEventElapsedTime(&stats->ms_exchange, e0, e1));
		}
		return rc;
	)
}
