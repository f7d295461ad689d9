// internal.h -- shared declarations of the spsparse_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/spsparse_amd.h"

namespace spsamd {

// ---------------------------------------------------------------- errors

struct Error {
	int code;
	std::string msg;
};

#define SPS_HIP(call)                                                              \
	do {                                                                           \
		hipError_t e_ = (call);                                                    \
		if (e_ != hipSuccess)                                                      \
			throw ::spsamd::Error{SPSAMD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)}; \
	} while (0)

#define SPS_LAUNCH_CHECK() SPS_HIP(hipGetLastError())

// ---------------------------------------------------------------- workspace

// Bump allocator over device slabs.  A multiply carves everything it needs
// from here; if the first slab was too small extra slabs are chained and the
// arena is re-made as one slab of the high-water size at the next reset, so a
// steady-state call does no hipMalloc.
struct Arena {
	struct Slab { char *p; size_t cap; size_t used; };
	std::vector<Slab> slabs;
	size_t high_water = 0;
	size_t call_used = 0;

	void *alloc(size_t bytes);
	// Stack discipline inside a call: everything allocated after mark() is given back by rewind().
	struct Mark { size_t nslabs, used, call_used; };
	Mark mark() const { return Mark{slabs.size(), slabs.empty() ? 0 : slabs.back().used, call_used}; }
	void rewind(const Mark &m);
	void reset();          // start of a call
	void release();        // free everything
	void reserve(size_t bytes);
	template <class T> T *get(size_t n) { return (T *)alloc(n * sizeof(T)); }
};

// Grow-only device buffer (the context's output buffer, host staging).
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	void ensure(size_t bytes);
	void release();
	bool holds(const void *q) const { return p && q && (const char *)q >= (const char *)p && (const char *)q < (const char *)p + cap; }
};

// One set of SINK_COO / consolidate output arrays.  A context has two: a result handed back as a
// device operand of the next call (T = R*A, then C = T*R^T: accum.hpp:73-101's use case) is read
// in place while the new result goes to the other set.
struct OutSet {
	DevBuf i, j, v;
	bool holds(const void *q) const { return i.holds(q) || j.holds(q) || v.holds(q); }
	void release() { i.release(); j.release(); v.release(); }
};

} // namespace spsamd

namespace spsamd {
// Developer knobs of one context.  Read from the environment ONCE, at spsamd_ctx_create
// (SPSAMD_W, SPSAMD_CELL_CAP, SPSAMD_DENSE_MIN, SPSAMD_NO_TILES, SPSAMD_XCD, SPSAMD_EMIT_PATH),
// or set through spsamd_ctx_set_tuning; results are identical for every setting.
struct Tuning {
	int window = 0;              // 0: chosen from the column count; 8192 / 16384
	int cell_cap = 0;            // 0: default grouping target of the hash cells
	int dense_min = 0;           // 0: default threshold above which a window becomes a dense cell
	int no_tiles = 0;
	int xcd = 2;                     // 0: one list for all XCDs | 1: every cell list in eight static parts (experiment, slower) | 2: the dense cells' list in eight parts, claimed
	int emit_path = 0;           // 0 auto | 1 no bitmap rank | 2 bitonic only
	int light_path = 0;          // 0 auto | 1 generic k_light only
	int light_two_pass = 0;      // 1: the all-light COO sink counts, scans and stores (two compute passes) instead of one pass + gather
	int no_wmajor = 0;           // 1: dense cells read the row-major B through bwin (no window-major copy)
	int tiles_v1 = 0;            // tile kernel of the hash-class cells: 0 auto, 1 first generation, 2 hash tiles v2, 3 bitmap rank
	int long_cap = 0;            // 0: cell_cap; grouping target of the hash cells of rows too long for tiles (<= 4096)
	int long_dense_min = 0;      // 0: default (1024 with 8192-column windows, else dense_min); dense threshold of the rows too long for tiles
	int direct_min = 0;          // 0: default; products above which one window of a tile row becomes a direct cell (>= dense_min: never)
	int trace = 0;               // 1: the symbolic phase prints its choices to stderr
	int index_budget_mb = 0;     // 0: 80 % of the free device memory; cap (MB) of the heavy rows' window indices, beyond which the product goes by column blocks
#ifdef SPSAMD_ABLATIONS
	int dbg = 0;
#endif
};
}

struct spsamd_ctx {
	int device = 0;
	spsamd::Tuning tune;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	spsamd::Arena arena;
	spsamd::OutSet out[2];                   // SINK_COO results (see OutSet)
	int cur_out = 0;
	// what each output set holds right now, where this library wrote it and knows it to be consolidated by `sort0` with
	// valid indices: handed back as an operand (T = R*A, then C = T*R^T) it is taken as it is, without the inspection pass
	struct OwnResult { const int32_t *d0 = nullptr, *d1 = nullptr; const double *v = nullptr; uint64_t nnz = 0, shape0 = 0, shape1 = 0; int sort0 = -1; } own[2];
	spsamd::DevBuf rowstat_n, rowstat_s, rowstat_h;     // DIGEST row statistics
	void *pinned = nullptr;                  // host staging for small readbacks / fetch
	size_t pinned_cap = 0;
	std::string last_error;
	hipEvent_t ev[10] = {};
	hipEvent_t ev2[3] = {};                  // around the tile launches of the heavy rows
	hipStream_t side = nullptr;              // second stream: the window-major copy of B is built on it beside the rest of the symbolic phase
	hipEvent_t ev_side[2] = {};              // [0] main -> side (inputs ready), [1] side -> main (copy built: waited for just before the dense cells)
	hipStream_t side2 = nullptr;             // third stream: the cell lists are sorted on it beside the light and mid rows' kernels
	hipEvent_t ev_side2[2] = {};             // [0] main -> side2 (cells emitted), [1] side2 -> main (lists sorted: waited for before the heavy rows' kernels)
	bool wm_pending = false, sort_pending = false;   // work of this call still running on side / side2 that the main stream has not waited for yet
	void join_side(bool wm, bool sort);      // make the main stream wait for it (no-op where nothing is pending)
	int num_cu = 256;
	void *host_staging(size_t bytes);
};

namespace spsamd {

// ---------------------------------------------------------------- primitives (prims.hip)

// out[i] = sum_{j<i} in[j], out[n] = total (out has n+1 entries).
void scan_exclusive_u32_i64(spsamd_ctx *c, const uint32_t *in, int64_t *out, size_t n);
void scan_exclusive_u32_u32(spsamd_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
void scan_exclusive_u8_u32(spsamd_ctx *c, const uint8_t *in, uint32_t *out, size_t n);
void scan_exclusive_u16_u32(spsamd_ctx *c, const uint16_t *in, uint32_t *out, size_t n);

// The same for up to SCAN_BATCH_MAX arrays of one length in three launches (blockIdx.y = array): the symbolic phase
// scans a dozen per-row counters of the heavy rows, and a launch is worth more than the work at that size.
constexpr int SCAN_BATCH_MAX = 12;
struct ScanBatch { const uint32_t *in[SCAN_BATCH_MAX]; uint32_t *out[SCAN_BATCH_MAX]; int count = 0;
	void add(const uint32_t *i, uint32_t *o) { in[count] = i; out[count] = o; ++count; } };
void scan_exclusive_u32_batch(spsamd_ctx *c, const ScanBatch &b, size_t n);

// One device-to-host round trip for a list of 32-bit words scattered over device memory (host[i] = *p[i]).
constexpr int WORD_LIST_MAX = 40;
struct WordList { const uint32_t *p[WORD_LIST_MAX]; int count = 0;
	int add(const void *q) { p[count] = (const uint32_t *)q; return count++; }
	int add64(const void *q) { const int at = add(q); add((const uint32_t *)q + 1); return at; } };
void read_back_words(spsamd_ctx *c, const WordList &w, uint32_t *host);

// Stable LSD radix sort of (key, payload) pairs on key bits [low_bit, key_bits) (a caller whose input is already in the
// order of the low bits skips their passes).
// Returns which of the two buffer pairs holds the result (0: keys0/pay0, 1: keys1/pay1).
int radix_sort_pairs(spsamd_ctx *c, uint64_t *keys0, uint32_t *pay0, uint64_t *keys1, uint32_t *pay1,
	size_t n, int key_bits, int low_bit = 0);

void fill_u32(spsamd_ctx *c, uint32_t *p, uint32_t v, size_t n);
void fill_zero(spsamd_ctx *c, void *p, size_t bytes);

template <class T>
T read_back(spsamd_ctx *c, const T *dev)
{
	T *h = (T *)c->host_staging(sizeof(T));
	SPS_HIP(hipMemcpyAsync(h, dev, sizeof(T), hipMemcpyDeviceToHost, c->stream));
	SPS_HIP(hipStreamSynchronize(c->stream));
	return *h;
}

// Internal value of spsamd_coo::mem (never part of the public ABI): device arrays this library produced itself and
// knows to be consolidated with valid indices -- consolidate_operand takes them as they are, without the inspection pass.
#define SPSAMD_MEM_DEVICE_VERIFIED 3

// ---------------------------------------------------------------- consolidated operand (consolidate.hip)

struct Prepared;

// op(X) in row-major consolidated form on the device.
struct ConMat {
	int32_t *row = nullptr;     // leading (row of op(X)) index per tuple
	int32_t *col = nullptr;     // minor index per tuple
	double *val = nullptr;
	uint32_t nnz = 0;
	uint64_t nrow = 0, ncol = 0;
};

// (spsamd_coo::mem == SPSAMD_MEM_PREPARED: idx0 carries the spsamd_operand handle, whose first member is its Prepared record)

// Upload (if host) + consolidate `X` by sort order {lead, 1-lead} into `out`
// (arena memory).  A prepared operand whose lead matches yields its handle in *prep (otherwise null).  Mirrors Consolidate<> (algorithm.hpp:353-369): an operand
// whose sort0 == lead is used as is.
// `ref_lead` is the leading dimension of the order the REFERENCE consolidates this operand in
// (it differs from `lead` for B: multiply_sparse.hpp:168); it decides which NaNs zero_nan drops.
void consolidate_operand(spsamd_ctx *c, const spsamd_coo *X, int lead, int ref_lead, int duplicate_policy,
	int zero_nan, ConMat *out, Prepared **prep = nullptr, const unsigned long long *global_first_key = nullptr);
// (global_first_key: the distributed step's -- device word holding the smallest reference-order key of a kept tuple over ALL
// ranks' blocks; under zero_nan the NaNs below it are the leading run the reference drops, algorithm.hpp:272-275)

// Distributed step under zero_nan: smallest reference-order key of a tuple of X (DEVICE arrays) that is neither 0 nor NaN
// -> *out_dev (all ones: none).  Same key as consolidate_operand's own first-kept search.
void first_kept_key_raw(spsamd_ctx *c, const spsamd_coo *Xdev, int lead, int ref_lead, unsigned long long *out_dev);

// Row boundaries of a consolidated operand: dim_beginnings (algorithm.hpp:74-118):
// beg[r] for each non-empty row + sentinel, and the row ids.
struct RowList {
	uint32_t *beg = nullptr;    // nrows + 1 entries
	int32_t *id = nullptr;      // nrows entries
	uint32_t nrows = 0;
};
void dim_beginnings(spsamd_ctx *c, const ConMat &m, RowList *out);

// Stable permutation sorting X by {lead, 1-lead} (device array of X->nnz uint32, arena memory).
uint32_t *sorted_permutation(spsamd_ctx *c, const spsamd_coo *X, int lead);

// Dense row pointer over all `nrow + extra` rows (extra trailing empty rows); `into`: the array to fill (nrow + 1 + extra
// entries) or null for arena memory.
uint32_t *dense_rowptr(spsamd_ctx *c, const ConMat &m, uint32_t extra, uint32_t *into = nullptr);

// One B tuple as the numeric kernels read it: column and value side by side (12 bytes), so
// a short B segment sits in one or two cache lines instead of two partial lines of separate
// col[] / val[] arrays.  Same bytes per product as the SoA form (12 B).
struct __attribute__((packed, aligned(4))) BTup { int32_t col; uint32_t vlo, vhi; };

// What a multiply derives from a consolidated operand before it can start, kept so that it is derived once: the reference
// keeps an operand's row structure across calls as well (the lazy dim_beginnings cache, VectorCooArray.hpp:325-335) and
// skips the consolidation of an operand that carries the wanted sort order (algorithm.hpp:360).  Either a VIEW for the
// duration of one call (pieces in the context's arena; the distributed step hands its panel's row pointer in this way) or a
// prepared-operand HANDLE of the C ABI (spsamd_operand_prepare: pieces in device memory of their own, built on first use).
struct Prepared {
	spsamd_ctx *ctx = nullptr;
	ConMat m;                         // op(X), consolidated, row-major
	int lead = 0;                     // the stored dimension that is m.row
	bool owns = false;                // handle: the pieces live in device memory of the handle's own until release()
	std::vector<void *> owned;        // its slabs (few and large: a handle's arrays are gathered from at random like the workspace's,
	uint64_t owned_bytes = 0;         // and many small allocations cost the numeric kernels 5 % in address translation)
	char *slab = nullptr;             // the slab being carved
	size_t slab_left = 0;
	void reserve(size_t bytes);       // make the next `bytes` of alloc() calls come out of one slab
	// both roles
	uint32_t *rowptr = nullptr;       // dense row pointer over nrow + 1 rows (the last one an empty sentinel): nrow + 2 entries
	uint32_t maxlen = 0;              // longest row
	bool have_maxlen = false;
	// left operand
	RowList rl;                       // dim_beginnings (algorithm.hpp:74-118)
	bool have_rl = false;
	// right operand
	BTup *btup = nullptr;             // (col, val) interleaved, nnz + 4 entries
	int W = 0;                        // column-window width of the heavy-row indices below (0: not built)
	uint32_t nwin = 0, nwp = 0;
	uint64_t nrowb = 0;
	uint32_t *bwin = nullptr;         // [nrowb][nwin + 1] first tuple of row k with column >= w * W
	uint16_t *wcnt = nullptr;         // [nrowb][nwp]      tuples of row k in window w
	uint32_t *wptr = nullptr;         // [nwin][nrowb] + 1 window-major copy: CSR pointer per window ...
	BTup *btw = nullptr;              // ... and its tuples
	void *alloc(size_t bytes);        // arena memory of the current call (view) or device memory of the handle's own
	template <class T> T *get(size_t n) { return (T *)alloc(n * sizeof(T)); }
	void release();
};

// ---------------------------------------------------------------- multiply (spgemm.hip and the k_*.hip kernel files)

struct ScaleDev {
	bool present = false;
	int32_t *pos = nullptr;     // dense: position in the vector or -1
	const double *val = nullptr; // device copy of the values
	uint64_t dim = 0;
};

void upload_scale(spsamd_ctx *c, const spsamd_vec *s, uint64_t dim, const char *name, ScaleDev *out);

struct MultiplyArgs {
	double C;
	ScaleDev si, sj, sk;
	ConMat A, B;
	int sink_kind, sink_flags;
	Prepared *pa = nullptr, *pb = nullptr;   // derived structures of A / B that already exist (or are kept once built): may be null
	hipEvent_t b_ready = nullptr;            // the TUPLES of B arrive on another stream (the distributed step's panel): wait for this
	                                         // event before the first kernel that reads them; its row pointer (pb->rowptr) is valid at once
};
void spgemm(spsamd_ctx *c, MultiplyArgs &a, spsamd_result *res);
void prepared_row_structure(spsamd_ctx *c, Prepared *p);      // its dense row pointer and longest row, now (spgemm.hip)

// Shared body of the MM and MV entry points (capi.hip); `arena_ready`: the caller has reset the workspace
// and may hold operands in it (the distributed step does); `parts`: records of derived structures the caller already
// has for A / B (the distributed step: its panel's row pointer, and the event that says the panel's tuples have arrived).
struct OperandParts { Prepared *pa = nullptr, *pb = nullptr; hipEvent_t b_ready = nullptr; };
int multiply_body(spsamd_ctx *c, double C,
	const spsamd_vec *scalei, const spsamd_coo *A, char transpose_A,
	const spsamd_vec *scalej, const spsamd_coo *B, char transpose_B,
	const spsamd_vec *scalek, int duplicate_policy, int zero_nan,
	int sink_kind, int sink_flags, spsamd_result *res, const char *what, bool arena_ready, const OperandParts *parts = nullptr);

// Select the output set the next result is written to: the current one unless a device operand lives in it.
void pick_output_set(spsamd_ctx *c, const spsamd_coo *const *operands, int n);

} // namespace spsamd
