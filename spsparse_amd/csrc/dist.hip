// dist.hip -- the multi-GPU step behind the C ABI (include/spsparse_amd.h, spsamd_dist_*).
//
// The reference is single threaded and has no counterpart.  What makes the path shardable is its own loop
// structure: output row i depends only on row i of op(A) and the op(B) rows {k : op(A)(i,k) != 0}
// (multiply_sparse.hpp:192), so op(A) is cut into contiguous row blocks, one per GPU, op(B) is distributed by
// contiguous blocks of ITS rows -- the inner index -- and ONE exchange step brings every rank the op(B) rows its
// block needs.  No reduction: C stays row partitioned.  The entry point takes the reference's own arguments
// (multiply_sparse.hpp:138-150): transpose flags, the three scale vectors, DuplicatePolicy, zero_nan.
//
// One step on every rank (device work on the context's stream, RCCL traffic on the step's own stream):
//   0. (zero_nan only) the smallest reference-order key of a kept tuple, per operand: 16 bytes to every peer.  The NaNs
//      the reference drops are those of the leading run of the WHOLE matrix' sorted sequence (algorithm.hpp:272-275),
//      not of one block's.
//   1. consolidate the own block of A by rows of op(A) and the own block of B by rows of op(B) (one consolidation where B
//      is A); check that the B block lies inside this rank's bounds; need mask: one byte per inner index k that occurs in
//      the A block.
//      ROUND 1, sizes fixed by b_bounds: to every owner q its slice of the mask, the lengths of all own op(B) rows and an
//      8-byte header carrying this rank's status -- so that a rank whose operands are bad takes part in the round and
//      every rank returns an error instead of one returning and the others waiting for it for ever.
//   2. from what arrived: per peer the masked lengths of the own rows and their prefix (where to pack), the panel's row
//      pointer over the whole inner dimension (the mask is this rank's own, the lengths the owners').  ONE read-back of
//      the send / receive totals -- the step's only host synchronisation besides those of the product itself.
//   3. K7, the pack kernel: (col, val) of the rows a peer needs, row after row, 12 bytes per tuple, no row
//      array -- an owner all of whose tuples are needed sends its block as it stands.
//      ROUND 2: columns and values, received straight into the panel at the owner's offset: the panel arrives row-major
//      sorted.  While it is in flight the product's symbolic phase already runs on the panel's row pointer; the first
//      kernel that reads the panel's tuples waits for the round's event.
//   4. the block product (multiply_body), both operands consolidated and trusted as they are.
// Transport: grouped ncclSend / ncclRecv (RCCL over xGMI) -- librccl is loaded at run time, only here -- or a
// caller-supplied all-to-allv (tests drive the same code over gloo on one GPU).
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
constexpr int MAX_WORLD = 64;      // ranks of one communicator (per-peer tables are passed to kernels by value)

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

// lengths of the own op(B) rows (what every peer learns in round 1), and whether every tuple of the block lies in one of them
__global__ void k_own_rowlen(const uint32_t *ptr, uint64_t my_lo, uint32_t my_n, uint32_t *rowlen)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < my_n) rowlen[j] = ptr[my_lo + j + 1] - ptr[my_lo + j];
}

__global__ void k_rows_in_bounds(const int32_t *brow, uint32_t n, uint64_t lo, uint64_t hi, uint32_t *flag)
{
	uint32_t bad = 0;
	for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
		const uint64_t r = (uint64_t)(uint32_t)brow[e];
		if (r < lo || r >= hi) bad = 1;
	}
	if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1u);
}

struct Header { uint32_t status, pad; };

// this rank's header, once per peer: the host's status word, or EINVAL where the device found the B block out of bounds
__global__ void k_fill_headers(Header *hdr, int world, uint32_t host_status, const uint32_t *oob)
{
	const int p = threadIdx.x;
	if (p < world) { hdr[p].status = host_status ? host_status : (*oob ? (uint32_t)(-SPSAMD_EINVAL) : 0u); hdr[p].pad = 0; }
}

// masked lengths of the own rows, one row of the table per peer (blockIdx.y)
__global__ void k_masked_rowlen(const uint32_t *own_len, uint32_t my_n, const uint8_t *their, uint32_t *mlen)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const size_t p = blockIdx.y;
	if (j < my_n) mlen[p * my_n + j] = their[p * my_n + j] ? own_len[j] : 0u;
}

// lengths of the panel's rows over the whole inner dimension (+ the sentinel row): own mask x the owners' lengths
__global__ void k_panel_rowlen(const uint8_t *need, const uint32_t *len_all, uint64_t n_inner, uint32_t *plen)
{
	uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_inner) plen[k] = need[k] ? len_all[k] : 0u;
	else if (k == n_inner) plen[k] = 0u;
}

// everything the host needs after round 1, in one place: [0] OR of the peers' status words, [1 .. W] tuples to send to each
// peer, [W+1 .. 2W+1] the panel offset at which each owner's rows start (+ the panel's size)
__global__ void k_collect_totals(const Header *hdr_in, const uint32_t *off, uint32_t my_n, const uint32_t *pptr, const uint64_t *bounds, int world,
	uint32_t *out)
{
	const int t = threadIdx.x;
	if (t == 0) { uint32_t s = 0; for (int p = 0; p < world; ++p) s = s ? s : hdr_in[p].status; out[0] = s; }
	if (t < world) out[1 + t] = off[(size_t)t * (my_n + 1) + my_n];
	if (t <= world) out[1 + world + t] = pptr[bounds[t]];
}

// K7: the needed rows' (col, val), row after row -- dst = masked prefix of the row + position inside the row.  One launch
// for all peers (blockIdx.y); a peer that gets the whole block as it stands, or nothing, has no destination.
struct PackDst { int32_t *col[MAX_WORLD]; double *val[MAX_WORLD]; };

__global__ void k_pack_rows(const int32_t *brow, const int32_t *bcol, const double *bval, uint32_t count,
	const uint32_t *ptr, uint64_t my_lo, uint32_t my_n, const uint8_t *their, const uint32_t *off, PackDst dst)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	const size_t p = blockIdx.y;
	if (e >= count || !dst.col[p]) return;
	const uint64_t r = (uint64_t)(uint32_t)brow[e];
	if (r < my_lo || r >= my_lo + my_n) return;                      // (a block out of bounds was reported in round 1: never index with it)
	const uint32_t j = (uint32_t)(r - my_lo);
	if (!their[p * my_n + j]) return;
	const uint32_t d = off[p * (my_n + 1) + j] + (e - ptr[r]);
	dst.col[p][d] = bcol[e];
	dst.val[p][d] = bval[e];
}

// row index of every panel tuple, from the pointer's side (the row array is never sent).  A wave takes 64 consecutive rows
// and fills them one after the other, all lanes on the row at hand: coalesced stores whatever the rows' lengths.  Rows of
// more than ROWS_LONG tuples go to a list that the second kernel fills a workgroup per row (the first 64 rows of an
// un-permuted R-MAT hold a million tuples: left to their one wave they set the kernel's time, 0.1 ms).
constexpr uint32_t ROWS_LONG = 2048;
struct LongRow { uint32_t k, b, e, pad; };

__global__ void k_rows_from_ptr(const uint32_t *ptr, uint64_t nrow, int32_t *row, LongRow *longlist, uint32_t *longcount)
{
	const uint64_t k0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull;
	if (k0 >= nrow) return;                                          // (uniform)
	const uint32_t start = ptr[min(k0 + lane_id(), nrow)];
	const uint32_t stop = ptr[min(k0 + lane_id() + 1, nrow)];
	unsigned long long todo = __ballot(stop > start);               // the non-empty rows of the 64
	while (todo) {                                                   // uniform
		const int l = __builtin_amdgcn_readfirstlane(__ffsll(todo) - 1);
		todo &= todo - 1ull;
		const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)start, l), e = (uint32_t)__builtin_amdgcn_readlane((int)stop, l);
		const int32_t k = (int32_t)(k0 + (uint64_t)l);
		if (e - b > ROWS_LONG) {
			if (lane_id() == 0) longlist[atomicAdd(longcount, 1u)] = LongRow{(uint32_t)k, b, e, 0u};
			continue;
		}
		for (uint32_t t = b + lane_id(); t < e; t += 64u) row[t] = k;
	}
}

__global__ void k_rows_long(const LongRow *longlist, const uint32_t *longcount, int32_t *row)
{
	const uint32_t n = *longcount;
	for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
		const LongRow r = longlist[item];
		for (uint32_t t = r.b + threadIdx.x; t < r.e; t += blockDim.x) row[t] = (int32_t)r.k;
	}
}

__global__ void k_min_keys(const unsigned long long *in, int world, unsigned long long *out)
{
	// in: [peer][2] -> out[2]
	const int t = threadIdx.x;
	if (t < 2) { unsigned long long m = ~0ull; for (int p = 0; p < world; ++p) m = in[2 * p + t] < m ? in[2 * p + t] : m; out[t] = m; }
}

} // namespace

struct spsamd_dist {
	spsamd_ctx *ctx = nullptr;
	int rank = 0, world = 1;
	void *comm = nullptr;            // ncclComm_t (built-in transport)
	bool own_comm = false;
	bool broken = false;             // a step failed between its rounds: the peers' state is unknown
	spsamd_alltoallv_fn xfn = nullptr;
	void *xuser = nullptr;
	hipStream_t xs = nullptr;        // every RCCL call of this communicator is issued on this stream
	hipEvent_t ev[2] = {nullptr, nullptr};          // timing
	hipEvent_t ev_in = nullptr, ev_out = nullptr;   // context stream -> exchange stream (buffers ready), and back (round complete)
};

// One round: `nseg` all-to-allv's of device buffers (segment s: send[s][p] (sendb[s][p] bytes) goes to rank p, recvb[s][p]
// bytes from rank p arrive in recv[s][p]).  The send buffers are complete on the context's stream when this is called; the
// received data is complete on it when this returns (stream order, not host order) -- unless `defer`: then the caller makes
// its stream wait for d->ev_out itself, later.
struct Segment {
	std::vector<const void *> send; std::vector<void *> recv; std::vector<size_t> sendb, recvb;
	explicit Segment(int w) : send(w, nullptr), recv(w, nullptr), sendb(w, 0), recvb(w, 0) {}
};

static bool exchange(spsamd_dist *d, const Segment *segs, int nseg, bool defer)
{
	spsamd_ctx *c = d->ctx;
	if (d->xfn) {
		SPS_HIP(hipStreamSynchronize(c->stream));                    // the callback may touch the buffers from the host
		for (int s = 0; s < nseg; ++s) {
			int rc = d->xfn(d->xuser, segs[s].send.data(), segs[s].sendb.data(), segs[s].recv.data(), segs[s].recvb.data(), d->world, (void *)c->stream);
			if (rc) throw Error{SPSAMD_EHIP, "the caller's all-to-allv transport failed"};
		}
		return false;                                                 // complete: nothing to wait for
	}
	RcclApi *r = rccl();
	if (!r) throw Error{SPSAMD_EHIP, "librccl could not be loaded"};
	SPS_HIP(hipEventRecord(d->ev_in, c->stream));
	SPS_HIP(hipStreamWaitEvent(d->xs, d->ev_in, 0));
	SPS_NCCL(r->GroupStart());
	for (int s = 0; s < nseg; ++s)
		for (int p = 0; p < d->world; ++p) {
			if (segs[s].sendb[p]) SPS_NCCL(r->Send(segs[s].send[p], segs[s].sendb[p], NCCL_UINT8, p, d->comm, d->xs));
			if (segs[s].recvb[p]) SPS_NCCL(r->Recv(segs[s].recv[p], segs[s].recvb[p], NCCL_UINT8, p, d->comm, d->xs));
		}
	SPS_NCCL(r->GroupEnd());
	SPS_HIP(hipEventRecord(d->ev_out, d->xs));
	if (!defer) SPS_HIP(hipStreamWaitEvent(c->stream, d->ev_out, 0));
	return defer;
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

extern "C" void spsamd_dist_destroy(spsamd_dist *d)
{
	if (!d) return;
	if (d->ctx) (void)hipSetDevice(d->ctx->device);
	if (d->xs) (void)hipStreamSynchronize(d->xs);
	if (d->own_comm && d->comm && rccl()) {
		(void)hipStreamSynchronize(d->ctx->stream);
		(void)rccl()->CommDestroy(d->comm);
	}
	for (auto &e : d->ev) if (e) (void)hipEventDestroy(e);
	if (d->ev_in) (void)hipEventDestroy(d->ev_in);
	if (d->ev_out) (void)hipEventDestroy(d->ev_out);
	if (d->xs) (void)hipStreamDestroy(d->xs);
	delete d;
}

extern "C" int spsamd_dist_create(spsamd_dist **out, spsamd_ctx *ctx, int rank, int world, const char *unique_id,
	void *nccl_comm, spsamd_alltoallv_fn transport, void *transport_user)
{
	if (!out || !ctx || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return SPSAMD_EINVAL;
	*out = nullptr;
	spsamd_dist *d = nullptr;
	try {
		d = new spsamd_dist();
		d->ctx = ctx; d->rank = rank; d->world = world;
		SPS_HIP(hipSetDevice(ctx->device));
		for (auto &e : d->ev) SPS_HIP(hipEventCreate(&e));
		SPS_HIP(hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming));
		SPS_HIP(hipEventCreateWithFlags(&d->ev_out, hipEventDisableTiming));
		SPS_HIP(hipStreamCreateWithFlags(&d->xs, hipStreamNonBlocking));
		if (transport) { d->xfn = transport; d->xuser = transport_user; }
		else if (nccl_comm) d->comm = nccl_comm;
		else {
			RcclApi *r = rccl();
			if (!r) throw Error{SPSAMD_EHIP, "librccl could not be loaded"};
			if (!unique_id) throw Error{SPSAMD_EINVAL, "spsamd_dist_create needs a unique id, a communicator or a transport"};
			UidBytes u;
			std::memcpy(u.internal, unique_id, 128);
			int e = r->CommInitRank(&d->comm, world, u, rank);
			if (e != 0) throw Error{SPSAMD_EHIP, std::string("ncclCommInitRank: ") + r->GetErrorString(e)};
			d->own_comm = true;
		}
		*out = d;
		return SPSAMD_OK;
	}
	catch (const spsamd::Error &e) { ctx->last_error = e.msg; spsamd_dist_destroy(d); return e.code; }      // (events, stream and the object go with it)
	catch (const std::bad_alloc &) { ctx->last_error = "host allocation failed"; spsamd_dist_destroy(d); return SPSAMD_ENOMEM; }
	catch (const std::exception &e) { ctx->last_error = e.what(); spsamd_dist_destroy(d); return SPSAMD_EINVAL; }
}

namespace {

// a host operand brought to the device (arena), a device operand as it is
const spsamd_coo *on_device(spsamd_ctx *c, const spsamd_coo *X, spsamd_coo *tmp)
{
	if (X->mem != SPSAMD_MEM_HOST || X->nnz == 0) return X;
	if (!X->idx0 || !X->idx1 || !X->val) throw Error{SPSAMD_EINVAL, "operand with nnz > 0 has a null array"};
	*tmp = *X;
	int32_t *d0 = c->arena.get<int32_t>(X->nnz), *d1 = c->arena.get<int32_t>(X->nnz);
	double *dv = c->arena.get<double>(X->nnz);
	SPS_HIP(hipMemcpyAsync(d0, X->idx0, X->nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	SPS_HIP(hipMemcpyAsync(d1, X->idx1, X->nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	SPS_HIP(hipMemcpyAsync(dv, X->val, X->nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
	tmp->idx0 = d0; tmp->idx1 = d1; tmp->val = dv; tmp->mem = SPSAMD_MEM_DEVICE;
	return tmp;
}

struct Panel {
	ConMat m;                        // the op(B) rows this rank needs: rows global, row-major
	uint32_t *ptr = nullptr;         // dense row pointer over the inner dimension (+ sentinel row)
	bool pending = false;            // its tuples are still in flight: wait for d->ev_out before reading them
	uint64_t remote = 0, sent = 0;
};

// Steps 0 - 3: both blocks consolidated, the panel fetched.  Every rank of the communicator runs the same rounds whatever
// happens to it locally; the error every rank agrees on is thrown after round 1.
void fetch_panel(spsamd_dist *d, const spsamd_coo *A_block, char transpose_A, const spsamd_coo *B_block, char transpose_B,
	const uint64_t *b_bounds, int duplicate_policy, int zero_nan, ConMat *Aout, Panel *P)
{
	spsamd_ctx *c = d->ctx;
	const int W = d->world, me = d->rank;
	hipStream_t st = c->stream;
	Error local{0, ""};                                              // what this rank found wrong with its own arguments
	auto fail = [&](int code, std::string msg) { if (!local.code) local = Error{code, std::move(msg)}; };

	// ---- arguments.  Shapes and bounds must be the same on every rank; what a rank can check alone it reports through round 1.
	const int a0 = transpose_A == 'T' ? 1 : 0, a1 = 1 - a0;
	const int bk = transpose_B == 'T' ? 1 : 0, bj = 1 - bk;
	const spsamd_coo *Bsrc = B_block ? B_block : A_block;
	const uint64_t ashape[2] = {A_block->shape0, A_block->shape1}, bshape[2] = {Bsrc->shape0, Bsrc->shape1};
	const uint64_t n_inner = ashape[a1];
	if (bshape[bk] != n_inner) {
		char buf[160];
		std::snprintf(buf, sizeof buf, "Inner dimensions for A (%ld) and B (%ld) must match!", (long)n_inner, (long)bshape[bk]);
		throw Error{SPSAMD_EDIM, buf};                               // (multiply_sparse.hpp:172-174; every rank sees the same shapes: no round needed)
	}
	if (!B_block && a0 != bk) throw Error{SPSAMD_EINVAL, "B_block may be NULL only where op(B)'s rows are op(A)'s rows of the same array (equal transpose flags)"};
	if (b_bounds[0] != 0 || b_bounds[W] != n_inner) throw Error{SPSAMD_EINVAL, "b_bounds must run from 0 to the inner dimension"};
	for (int p = 0; p < W; ++p) if (b_bounds[p] > b_bounds[p + 1]) throw Error{SPSAMD_EINVAL, "b_bounds must be ascending"};
	if (n_inner >= (uint64_t(1) << 31)) throw Error{SPSAMD_EINVAL, "inner dimension exceeds the int32 index range"};
	if (duplicate_policy < 0 || duplicate_policy > 2) throw Error{SPSAMD_EINVAL, "bad duplicate_policy"};
	const uint64_t my_lo = b_bounds[me];
	const uint32_t my_n = (uint32_t)(b_bounds[me + 1] - my_lo);

	d->broken = true;                                                // until the rounds are through (or an agreed error ends the step): a failure in between leaves the peers waiting
	// ---- round buffers first: a rank that fails below still takes part in the rounds
	uint8_t *need = c->arena.get<uint8_t>(n_inner + 1);
	uint8_t *their = c->arena.get<uint8_t>((size_t)my_n * W + 1);   // [peer][own row]: does the peer need it
	uint32_t *own_len = c->arena.get<uint32_t>((size_t)my_n + 1);
	uint32_t *len_all = c->arena.get<uint32_t>(n_inner + 1);
	Header *hdr_out = c->arena.get<Header>(W), *hdr_in = c->arena.get<Header>(W);
	uint32_t *oob = c->arena.get<uint32_t>(1);
	unsigned long long *keys_out = c->arena.get<unsigned long long>(2), *keys_in = c->arena.get<unsigned long long>(2 * (size_t)W), *keys = c->arena.get<unsigned long long>(2);
	fill_zero(c, need, n_inner + 1);
	fill_zero(c, own_len, ((size_t)my_n + 1) * sizeof(uint32_t));
	fill_zero(c, oob, sizeof(uint32_t));

	// ---- 0. zero_nan: the first kept tuple of the whole matrix, per operand
	spsamd_coo tmpA, tmpB;
	const spsamd_coo *Ad = A_block, *Bd = B_block;
	try {
		Ad = on_device(c, A_block, &tmpA);
		if (B_block) Bd = on_device(c, B_block, &tmpB);
	} catch (const Error &e) { fail(e.code, e.msg); }
	if (zero_nan) {
		SPS_HIP(hipMemsetAsync(keys_out, 0xFF, 2 * sizeof(unsigned long long), st));
		if (!local.code) {
			try {
				if (Ad->mem == SPSAMD_MEM_PREPARED || (Bd && Bd->mem == SPSAMD_MEM_PREPARED)) throw Error{SPSAMD_EINVAL, "a prepared block cannot be consolidated under zero_nan"};
				first_kept_key_raw(c, Ad, a0, a0, keys_out);
				first_kept_key_raw(c, Bd ? Bd : Ad, bk, bj, keys_out + 1);
			} catch (const Error &e) { fail(e.code, e.msg); }
		}
		Segment s(W);
		for (int p = 0; p < W; ++p) { s.send[p] = keys_out; s.sendb[p] = 16; s.recv[p] = keys_in + 2 * (size_t)p; s.recvb[p] = 16; }
		exchange(d, &s, 1, false);
		k_min_keys<<<dim3(1), dim3(64), 0, st>>>(keys_in, W, keys);
		SPS_LAUNCH_CHECK();
	}

	// ---- 1. consolidate the own blocks; bounds check; need mask; own row lengths
	ConMat Ac, Bc;
	uint32_t *ptr = nullptr;                                        // dense row pointer of the own B block (global row index)
	if (!local.code) {
		try {
			consolidate_operand(c, Ad, a0, a0, duplicate_policy, zero_nan, &Ac, nullptr, zero_nan ? keys : nullptr);
			// (B is A: one consolidation serves both -- except under zero_nan, where the NaNs dropped from B are those of the
			// reference's column-major sequence, multiply_sparse.hpp:168)
			if (!B_block && !zero_nan) Bc = Ac;
			else consolidate_operand(c, Bd ? Bd : Ad, bk, bj, duplicate_policy, zero_nan, &Bc, nullptr, zero_nan ? keys + 1 : nullptr);
			if (Bc.nnz) {
				k_rows_in_bounds<<<dim3(std::min(grid_for(Bc.nnz), 2048u)), dim3(256), 0, st>>>(Bc.row, Bc.nnz, my_lo, my_lo + my_n, oob);
				SPS_LAUNCH_CHECK();
			}
			if (Ac.nnz) { k_mark_need<<<dim3(grid_for(Ac.nnz)), dim3(256), 0, st>>>(Ac.col, Ac.nnz, need); SPS_LAUNCH_CHECK(); }
			ptr = dense_rowptr(c, Bc, 0);
			if (my_n) { k_own_rowlen<<<dim3(grid_for(my_n)), dim3(256), 0, st>>>(ptr, my_lo, my_n, own_len); SPS_LAUNCH_CHECK(); }
		} catch (const Error &e) { fail(e.code, e.msg); }
	}
	k_fill_headers<<<dim3(1), dim3(64), 0, st>>>(hdr_out, W, (uint32_t)(local.code ? -local.code : 0), oob);
	SPS_LAUNCH_CHECK();

	// ---- ROUND 1: header, mask slice, own row lengths
	{
		Segment s[3] = {Segment(W), Segment(W), Segment(W)};
		for (int p = 0; p < W; ++p) {
			const size_t np = (size_t)(b_bounds[p + 1] - b_bounds[p]);
			s[0].send[p] = hdr_out + p; s[0].sendb[p] = sizeof(Header); s[0].recv[p] = hdr_in + p; s[0].recvb[p] = sizeof(Header);
			s[1].send[p] = need + b_bounds[p]; s[1].sendb[p] = np; s[1].recv[p] = their + (size_t)my_n * p; s[1].recvb[p] = my_n;
			s[2].send[p] = own_len; s[2].sendb[p] = (size_t)my_n * 4; s[2].recv[p] = len_all + b_bounds[p]; s[2].recvb[p] = np * 4;
		}
		exchange(d, s, 3, false);
	}

	// ---- 2. what to send, where the panel's rows go: one read-back
	uint32_t *mlen = c->arena.get<uint32_t>((size_t)my_n * W + 1);
	uint32_t *off = c->arena.get<uint32_t>(((size_t)my_n + 1) * W);
	if (my_n) { k_masked_rowlen<<<dim3(grid_for(my_n), W), dim3(256), 0, st>>>(own_len, my_n, their, mlen); SPS_LAUNCH_CHECK(); }
	for (int p0 = 0; p0 < W; p0 += SCAN_BATCH_MAX) {
		ScanBatch sb;
		for (int p = p0; p < std::min(W, p0 + SCAN_BATCH_MAX); ++p) sb.add(mlen + (size_t)my_n * p, off + ((size_t)my_n + 1) * p);
		scan_exclusive_u32_batch(c, sb, my_n);
	}
	uint32_t *plen = c->arena.get<uint32_t>(n_inner + 2);
	uint32_t *pptr = c->arena.get<uint32_t>(n_inner + 2);
	k_panel_rowlen<<<dim3(grid_for(n_inner + 1)), dim3(256), 0, st>>>(need, len_all, n_inner, plen);
	SPS_LAUNCH_CHECK();
	scan_exclusive_u32_u32(c, plen, pptr, n_inner + 1);              // pptr[n_inner + 1] = the panel's size: the sentinel row is empty
	uint32_t *totals = c->arena.get<uint32_t>(2 * (size_t)W + 2);
	uint64_t *bdev = c->arena.get<uint64_t>((size_t)W + 1);
	SPS_HIP(hipMemcpyAsync(bdev, b_bounds, ((size_t)W + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
	k_collect_totals<<<dim3(1), dim3(128), 0, st>>>(hdr_in, off, my_n, pptr, bdev, W, totals);
	SPS_LAUNCH_CHECK();
	std::vector<uint32_t> h(2 * (size_t)W + 2);
	{
		uint32_t *hp = (uint32_t *)c->host_staging(h.size() * sizeof(uint32_t));
		SPS_HIP(hipMemcpyAsync(hp, totals, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		SPS_HIP(hipStreamSynchronize(st));
		std::copy(hp, hp + h.size(), h.begin());
	}
	if (h[0] || local.code) {
		// some rank reported an error: every rank stops here, before round 2, with the communicator intact
		d->broken = false;
		if (local.code) throw local;
		const uint32_t oobh = read_back(c, oob);
		if (oobh) throw Error{SPSAMD_EINVAL, B_block ? "B_block holds a tuple whose op(B) row lies outside this rank's b_bounds"
			: "with B_block == NULL the A block is the B block: it holds a tuple whose row lies outside this rank's b_bounds"};
		throw Error{SPSAMD_EPEER, "another rank of the communicator rejected its operands (status " + std::to_string(-(int)h[0]) + "): nothing was multiplied"};
	}
	const uint32_t *send_tuples = h.data() + 1, *recv_at = h.data() + 1 + W;
	const uint32_t pn = recv_at[W], nmine = Bc.nnz;

	// ---- 3. pack, ROUND 2
	int32_t *prow = c->arena.get<int32_t>(pn ? pn : 1), *pcol = c->arena.get<int32_t>(pn ? pn : 1);
	double *pval = c->arena.get<double>(pn ? pn : 1);
	Segment s[2] = {Segment(W), Segment(W)};
	PackDst dst{};
	bool any_pack = false;
	for (int p = 0; p < W; ++p) {
		const uint32_t t = send_tuples[p];
		if (t == nmine) { s[0].send[p] = Bc.col; s[1].send[p] = Bc.val; }                      // the whole block as it stands
		else if (t) {
			dst.col[p] = c->arena.get<int32_t>(t); dst.val[p] = c->arena.get<double>(t);
			s[0].send[p] = dst.col[p]; s[1].send[p] = dst.val[p];
			any_pack = true;
		}
		s[0].sendb[p] = (size_t)t * 4; s[1].sendb[p] = (size_t)t * 8;
		const size_t rt = recv_at[p + 1] - recv_at[p];
		s[0].recv[p] = pcol + recv_at[p]; s[0].recvb[p] = rt * 4;
		s[1].recv[p] = pval + recv_at[p]; s[1].recvb[p] = rt * 8;
		if (p != me) P->sent += t;
	}
	if (any_pack) {
		k_pack_rows<<<dim3(grid_for(nmine), W), dim3(256), 0, st>>>(Bc.row, Bc.col, Bc.val, nmine, ptr, my_lo, my_n, their, off, dst);
		SPS_LAUNCH_CHECK();
	}
	P->pending = exchange(d, s, 2, true);
	d->broken = false;
	// (on the context's stream, beside the transfer: the panel's row array, which only the heavy-row indices read)
	if (pn) {
		LongRow *longlist = c->arena.get<LongRow>((size_t)pn / ROWS_LONG + 1);
		uint32_t *longcount = c->arena.get<uint32_t>(1);
		fill_zero(c, longcount, sizeof(uint32_t));
		k_rows_from_ptr<<<dim3(grid_for(n_inner)), dim3(256), 0, st>>>(pptr, n_inner, prow, longlist, longcount);
		SPS_LAUNCH_CHECK();
		k_rows_long<<<dim3(512), dim3(256), 0, st>>>(longlist, longcount, prow);
		SPS_LAUNCH_CHECK();
	}

	*Aout = Ac;
	P->m.row = prow; P->m.col = pcol; P->m.val = pval; P->m.nnz = pn;
	P->m.nrow = n_inner; P->m.ncol = bshape[bj];
	P->ptr = pptr;
	P->remote = pn - (recv_at[me + 1] - recv_at[me]);
}

} // namespace

extern "C" int spsamd_dist_multiply(spsamd_dist *d, double C,
	const spsamd_vec *scalei, const spsamd_coo *A_block, char transpose_A,
	const spsamd_vec *scalej, const spsamd_coo *B_block, char transpose_B,
	const spsamd_vec *scalek, const uint64_t *b_bounds,
	int duplicate_policy, int zero_nan, int sink_kind, int sink_flags,
	spsamd_result *res, spsamd_dist_stats *stats)
{
	if (!d || !d->ctx) return SPSAMD_EINVAL;
	spsamd_ctx *c = d->ctx;
	DIST_GUARD(c,
		if (!A_block || !b_bounds || !res) throw Error{SPSAMD_EINVAL, "null block, bounds or result"};
		if (d->broken) throw Error{SPSAMD_EPEER, "an earlier step of this communicator failed between its exchange rounds: its ranks are out of step, create a new one"};
		if (sink_kind != SPSAMD_SINK_COO && sink_kind != SPSAMD_SINK_DIGEST) throw Error{SPSAMD_EINVAL, "bad sink_kind"};
		SPS_HIP(hipSetDevice(c->device));
		hipStream_t st = c->stream;
		c->arena.reset();
		SPS_HIP(hipEventRecord(d->ev[0], st));
		// a chained result (T = R*A on this rank, then C = T*R^T) is read in place: the new result goes to the other buffer set
		{ const spsamd_coo *ops[2] = {A_block, B_block}; pick_output_set(c, ops, 2); }

		ConMat Ac;
		Panel P;
		fetch_panel(d, A_block, transpose_A, B_block, transpose_B, b_bounds, duplicate_policy, zero_nan, &Ac, &P);
		SPS_HIP(hipEventRecord(d->ev[1], st));

		// ---- 4. the block product: both operands consolidated, trusted as they are; the panel's row pointer exists already
		const int a0 = transpose_A == 'T' ? 1 : 0, bk = transpose_B == 'T' ? 1 : 0;
		const spsamd_coo *Bsrc = B_block ? B_block : A_block;
		const uint64_t ashape[2] = {A_block->shape0, A_block->shape1}, bshape[2] = {Bsrc->shape0, Bsrc->shape1};
		spsamd_coo Ad, Bd;
		Ad.idx0 = Ac.row; Ad.idx1 = Ac.col; Ad.val = Ac.val; Ad.nnz = Ac.nnz; Ad.shape0 = ashape[a0]; Ad.shape1 = ashape[1 - a0];
		Ad.sort0 = 0; Ad.mem = SPSAMD_MEM_DEVICE_VERIFIED;
		Bd.idx0 = P.m.row; Bd.idx1 = P.m.col; Bd.val = P.m.val; Bd.nnz = P.m.nnz; Bd.shape0 = bshape[bk]; Bd.shape1 = bshape[1 - bk];
		Bd.sort0 = 0; Bd.mem = SPSAMD_MEM_DEVICE_VERIFIED;
		Prepared view;
		view.ctx = c; view.m = P.m; view.lead = 0; view.rowptr = P.ptr;
		OperandParts parts;
		parts.pb = &view;
		parts.b_ready = P.pending ? d->ev_out : nullptr;
		if (stats) {
			std::memset(stats, 0, sizeof(*stats));
			stats->panel_tuples = P.m.nnz; stats->remote_tuples = P.remote; stats->sent_tuples = P.sent; stats->block_nnz_a = Ac.nnz;
		}
		int rc;
		try {
			rc = multiply_body(c, C, scalei, &Ad, '.', scalej, &Bd, '.', scalek, duplicate_policy, 0, sink_kind, sink_flags, res, "B", true, &parts);
		} catch (...) {
			// the panel may still be arriving into this call's workspace: let it land before anything reuses that memory
			if (P.pending) (void)hipStreamWaitEvent(st, d->ev_out, 0);
			throw;
		}
		if (P.pending) SPS_HIP(hipStreamWaitEvent(st, d->ev_out, 0));   // (a product that never looked at B's tuples: empty A block, C == 0 ...)
		if (stats) {
			SPS_HIP(hipEventSynchronize(d->ev[1]));
			SPS_HIP(hipEventElapsedTime(&stats->ms_exchange, d->ev[0], d->ev[1]));
		}
		return rc;
	)
}
