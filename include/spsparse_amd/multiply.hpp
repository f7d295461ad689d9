// spsparse_amd/multiply.hpp -- header-only host shim with the call shape of
// spsparse::multiply() (slib/spsparse/multiply_sparse.hpp:138-164) on top of
// the C ABI in spsparse_amd.h.  Migration is
//     s/spsparse::multiply/spsparse_amd::multiply/
// Operands may be spsparse::VectorCooArray (the real one) or the mirror type
// below: the template only touches the members the reference touches
// (shape, size(), index(d,i), val(i), sort_order -- VectorCooArray.hpp:17,45-53,
// 85-86,35).  Results stream into any Accumulator with val_type / set_shape /
// add (accum.hpp:12-24), in ascending (i,j), zeros dropped
// (multiply_sparse.hpp:238-243).  Errors go through a printf-style hook that
// defaults to throwing, like spsparse_error (spsparse.hpp:47,54; spsparse.cpp:12-28).
// Requires C++17 (inline variable for the hook).  No CPU fallback: without a
// gfx950 device the call raises through the hook.
#pragma once

#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <memory>
#include <sstream>
#include <type_traits>
#include <vector>

#include "../spsparse_amd.h"

namespace spsparse_amd {

// ---- spsparse.hpp:25-54 -------------------------------------------------
enum class DuplicatePolicy { LEAVE_ALONE, ADD, REPLACE };

class Exception : public std::exception {
public:
	virtual ~Exception() {}
	virtual const char *what() const noexcept { return "spsparse_amd::Exception()"; }
};

typedef void (*error_ptr)(int retcode, char const *str, ...);

inline void default_error(int retcode, const char *format, ...)
{
	(void)retcode;
	va_list arglist;
	va_start(arglist, format);
	std::vfprintf(stderr, format, arglist);
	va_end(arglist);
	std::fprintf(stderr, "\n");
	throw Exception();
}

// User-replaceable, like spsparse::spsparse_error.
inline error_ptr spsparse_error = &default_error;

// spsparse.cpp:30-31
// SPSAMD_SINK_* flags passed to every multiply of this process.  The drop-in template defaults to
// SPSAMD_SINK_EXACT_PATTERN: the INDEX SET is the reference's on any input (a sum that cancels to exactly 0 in the
// reference's ascending-k order is dropped, multiply_sparse.hpp:238, and only such sums), values are bit-identical on
// rows of <= 64 products and within rounding (1e-12 relative to the sum of |terms|) elsewhere; about 1.2x the time of
// flags = 0.  Set to SPSAMD_SINK_ORDERED for ascending-k sums everywhere (bit-identical values on any input, several
// times slower on heavy rows), or to 0 for the fastest path (arrival-order sums, pattern exact unless terms cancel).
inline int multiply_flags = SPSAMD_SINK_EXACT_PATTERN;

inline const std::array<int, 2> ROW_MAJOR = {0, 1};
inline const std::array<int, 2> COL_MAJOR = {1, 0};

// ---- one device context per host thread (created on first use) ----------
class Context {
	spsamd_ctx *c_ = nullptr;
public:
	explicit Context(int device = -1, void *hip_stream = nullptr)
	{
		int rc = spsamd_ctx_create(&c_, device, hip_stream);
		if (rc != 0) (*spsparse_error)(-1, "spsparse_amd: cannot create a device context (code %d): no MI355X visible?", rc);
	}
	~Context() { if (c_) spsamd_ctx_destroy(c_); }
	Context(const Context &) = delete;
	Context &operator=(const Context &) = delete;
	spsamd_ctx *get() const { return c_; }
};

inline Context &default_context()
{
	static thread_local std::unique_ptr<Context> ctx;
	if (!ctx) ctx.reset(new Context());
	return *ctx;
}

// ---- cursor over a COO container: the surface of the reference's CooIterator (array.hpp:70-115) -- index(),
// index(k), val(), offset(), set_index(), random-access stepping, comparison by position
template <class ArrayT, class IndexRefT, class ValRefT>
class CooCursor {
	ArrayT *arr_;
	int at_;
public:
	static const int rank = ArrayT::rank;
	typedef typename ArrayT::indices_type indices_type;
	typedef indices_type value_type;
	typedef typename ArrayT::index_type index_type;
	typedef typename ArrayT::val_type val_type;
	CooCursor(ArrayT *arr, int at) : arr_(arr), at_(at) {}
	indices_type operator[](int n) const { return arr_->index(at_ + n); }
	indices_type index() const { return arr_->index(at_); }
	indices_type operator*() const { return arr_->index(at_); }
	CooCursor &operator+=(int n) { at_ += n; return *this; }
	CooCursor &operator-=(int n) { at_ -= n; return *this; }
	CooCursor &operator++() { ++at_; return *this; }
	CooCursor &operator--() { --at_; return *this; }
	CooCursor operator+(int n) const { return CooCursor(arr_, at_ + n); }
	bool operator==(CooCursor const &o) const { return at_ == o.at_; }
	bool operator!=(CooCursor const &o) const { return at_ != o.at_; }
	int offset() const { return at_; }
	IndexRefT index(int k) const { return arr_->index(k, (size_t)at_); }
	void set_index(indices_type const &idx) const { arr_->set_index(at_, idx); }
	ValRefT val() const { return arr_->val((size_t)at_); }
};

// ---- host container with VectorCooArray's surface (VectorCooArray.hpp:8-158)
template <class IndexT, class ValT, int RANK>
class VectorCooArray {
public:
	static const int rank = RANK;
	typedef IndexT index_type;
	typedef ValT val_type;
	typedef std::array<index_type, RANK> indices_type;

	std::array<size_t, RANK> shape;
	bool edit_mode;
	std::array<int, RANK> sort_order;

	VectorCooArray() : edit_mode(true), sort_order() { sort_order[0] = -1; for (int k = 0; k < RANK; ++k) shape[k] = 0; }
	explicit VectorCooArray(std::array<size_t, RANK> const &_shape) : shape(_shape), edit_mode(true), sort_order() { sort_order[0] = -1; }

	void set_shape(std::array<size_t, RANK> const &_shape) { shape = _shape; }
	std::unique_ptr<VectorCooArray> new_blank() const { return std::unique_ptr<VectorCooArray>(new VectorCooArray(shape)); }

	IndexT &index(int dim, size_t ix) { return index_vecs[dim][ix]; }
	IndexT const &index(int dim, size_t ix) const { return index_vecs[dim][ix]; }
	ValT &val(size_t ix) { return val_vec[ix]; }
	ValT const &val(size_t ix) const { return val_vec[ix]; }
	std::array<IndexT, RANK> index(int ix) const
	{
		std::array<IndexT, RANK> r;
		for (int k = 0; k < RANK; ++k) r[k] = index(k, (size_t)ix);
		return r;
	}
	// VectorCooArray.hpp:60-67
	std::vector<IndexT> index_vec(int ix) const
	{
		std::vector<IndexT> r;
		for (int k = 0; k < RANK; ++k) r.push_back(index(k, (size_t)ix));
		return r;
	}
	void set_index(int ix, std::array<IndexT, RANK> const &idx) { for (int k = 0; k < RANK; ++k) index(k, (size_t)ix) = idx[k]; }
	size_t size() const { return val_vec.size(); }

	// VectorCooArray.hpp:88-104: iterator = cursor at a tuple position; begin(ix) / end(ix) offset from the ends
	typedef CooCursor<VectorCooArray, IndexT &, ValT &> iterator;
	typedef CooCursor<const VectorCooArray, IndexT const &, ValT const &> const_iterator;
	iterator begin(int ix = 0) { return iterator(this, ix); }
	iterator end(int ix = 0) { return iterator(this, (int)size() + ix); }
	const_iterator cbegin(int ix = 0) const { return const_iterator(this, ix); }
	const_iterator cend(int ix = 0) const { return const_iterator(this, (int)size() + ix); }
	const_iterator begin(int ix = 0) const { return const_iterator(this, ix); }
	const_iterator end(int ix = 0) const { return const_iterator(this, (int)size() - ix); }      // (sic: VectorCooArray.hpp:104 subtracts)

	void clear()
	{
		for (int k = 0; k < RANK; ++k) index_vecs[k].clear();
		val_vec.clear();
		edit_mode = true;
		sort_order[0] = -1;
	}
	void reserve(size_t n) { for (int k = 0; k < RANK; ++k) index_vecs[k].reserve(n); val_vec.reserve(n); }
	void edit() { edit_mode = true; sort_order[0] = -1; }
	void set_sorted(std::array<int, RANK> _sort_order) { sort_order = _sort_order; edit_mode = false; }

	// VectorCooArray.hpp:238-266
	void add(std::array<IndexT, RANK> const index, ValT const val)
	{
		if (!edit_mode) (*spsparse_error)(-1, "Must be in edit mode to use VectorCooArray::add()");
		for (int i = 0; i < RANK; ++i) {
			if (index[i] < 0 || (size_t)index[i] >= shape[i]) {
				std::ostringstream buf;
				buf << "Sparse index out of bounds: index=(";
				for (int j = 0; j < RANK; ++j) buf << index[j] << " ";
				buf << ") vs. shape=(";
				for (int j = 0; j < RANK; ++j) buf << shape[j] << " ";
				buf << ")";
				(*spsparse_error)(-1, "%s", buf.str().c_str());
			}
		}
		for (int i = 0; i < RANK; ++i) index_vecs[i].push_back(index[i]);
		val_vec.push_back(val);
	}

	// Bulk form of add() for tuples the library produced: same edit-mode rule, same bounds rule (checked
	// on the whole chunk), one append per column instead of one push_back per tuple and column.  multiply()
	// uses it when the sink is this container; any other Accumulator still receives one add() per tuple.
	void add_tuples(const IndexT *const *index, const ValT *val, size_t n)
	{
		if (!edit_mode) (*spsparse_error)(-1, "Must be in edit mode to use VectorCooArray::add()");
		for (int k = 0; k < RANK; ++k)
			for (size_t q = 0; q < n; ++q)
				if (index[k][q] < 0 || (size_t)index[k][q] >= shape[k]) { add_one_of(index, val, q); return; }   // raises like add()
		for (int k = 0; k < RANK; ++k) index_vecs[k].insert(index_vecs[k].end(), index[k], index[k] + n);
		val_vec.insert(val_vec.end(), val, val + n);
	}

	// VectorCooArray.hpp:143-147: the index columns are permuted in place; like the reference's
	// OverwriteAccum it leaves shape and sort_order as they were.
	void transpose(std::array<int, RANK> const &perm)
	{
		std::array<std::vector<IndexT>, RANK> old(index_vecs);
		for (int new_k = 0; new_k < RANK; ++new_k) index_vecs[new_k] = old[perm[new_k]];
	}

	// In-place device consolidate (VectorCooArray.hpp:299-311), rank 2 only.
	void consolidate(std::array<int, RANK> const &_sort_order, DuplicatePolicy duplicate_policy = DuplicatePolicy::ADD,
		bool handle_nan = false);

protected:
	std::array<std::vector<IndexT>, RANK> index_vecs;
	std::vector<ValT> val_vec;

	void add_one_of(const IndexT *const *index, const ValT *val, size_t q)
	{
		std::array<IndexT, RANK> ix;
		for (int k = 0; k < RANK; ++k) ix[k] = index[k][q];
		add(ix, val[q]);
	}
};

// ---- the accumulators of accum.hpp that make sense around multiply(), host side -------------
// (the device analogues are sink flags / entry points of the C ABI: SPSAMD_SINK_PERMUTE,
//  spsamd_result_scatter_dense, SPSAMD_SINK_DIGEST)

// accum.hpp:73-101.  Unlike the reference's it forwards set_shape (permuted), so it can be the sink of multiply().
template <int IN_RANK, class AccumulatorT>
class PermuteAccum {
	AccumulatorT &sub;
	std::vector<int> perm;
public:
	static const int rank = IN_RANK;
	typedef typename AccumulatorT::val_type val_type;
	PermuteAccum(AccumulatorT &_sub, std::vector<int> const &_perm) : sub(_sub), perm(_perm) {}
	void set_shape(std::array<size_t, IN_RANK> const &shape)
	{
		std::array<size_t, AccumulatorT::rank> out;
		for (int i = 0; i < AccumulatorT::rank; ++i) out[i] = shape[perm[i]];
		sub.set_shape(out);
	}
	void add(std::array<int, IN_RANK> const &index, val_type const &val)
	{
		std::array<int, AccumulatorT::rank> out;
		for (int i = 0; i < AccumulatorT::rank; ++i) out[i] = index[perm[i]];
		sub.add(out, val);
	}
};

// accum.hpp:110-140 on a caller-owned row-major dense matrix (the reference writes a blitz::Array).
template <class IndexT, class ValT>
struct DenseAccum {
	static const int rank = 2;
	typedef ValT val_type;
	ValT *dense; size_t ld;
	DuplicatePolicy duplicate_policy;
	DenseAccum(ValT *_dense, size_t _ld, DuplicatePolicy _p = DuplicatePolicy::ADD) : dense(_dense), ld(_ld), duplicate_policy(_p) {}
	void set_shape(std::array<size_t, 2> const &) {}
	void add(std::array<IndexT, 2> const &index, ValT const &val)
	{
		ValT &oval = dense[(size_t)index[0] * ld + (size_t)index[1]];
		switch (duplicate_policy) {
			case DuplicatePolicy::LEAVE_ALONE: if (!std::isnan(oval)) oval = val; break;     // accum.hpp:128-130, as written there
			case DuplicatePolicy::ADD: oval += val; break;
			case DuplicatePolicy::REPLACE: oval = val; break;
		}
	}
};

// accum.hpp:158-167
template <class IndexT, class ValT, int RANK>
struct ScalarAccumulator {
	static const int rank = RANK;
	typedef ValT val_type;
	ValT val;
	ScalarAccumulator() : val(0) {}
	void set_shape(std::array<size_t, RANK> const &) {}
	void add(std::array<IndexT, RANK> const &, ValT const &v) { val += v; }
};

// algorithm.hpp:46-57: ret.dim[i] == A.dim[perm[i]]
template <class VectorCooArrayT, class AccumulatorT>
void transpose(AccumulatorT &ret, VectorCooArrayT const &A, std::array<int, VectorCooArrayT::rank> const &perm)
{
	std::array<typename VectorCooArrayT::index_type, VectorCooArrayT::rank> idx;
	for (size_t i = 0; i < A.size(); ++i) {
		for (int new_k = 0; new_k < VectorCooArrayT::rank; ++new_k) idx[new_k] = A.index(perm[new_k], i);
		ret.add(idx, A.val(i));
	}
}

template <class IndexT, class ValT>
using VectorCooMatrix = VectorCooArray<IndexT, ValT, 2>;
template <class IndexT, class ValT>
using VectorCooVector = VectorCooArray<IndexT, ValT, 1>;

namespace detail {

template <class MatT>
spsamd_coo as_coo(MatT const &A)
{
	static_assert(sizeof(typename MatT::index_type) == 4 && std::is_integral<typename MatT::index_type>::value,
		"spsparse_amd: index_type must be a 32-bit integer");
	static_assert(std::is_same<typename MatT::val_type, double>::value, "spsparse_amd: val_type must be double");
	static_assert(MatT::rank == 2, "matrix operand must have rank 2");
	spsamd_coo c;
	size_t n = A.size();
	c.idx0 = n ? reinterpret_cast<const int32_t *>(&A.index(0, 0)) : nullptr;
	c.idx1 = n ? reinterpret_cast<const int32_t *>(&A.index(1, 0)) : nullptr;
	c.val = n ? &A.val(0) : nullptr;
	c.nnz = n;
	c.shape0 = A.shape[0];
	c.shape1 = A.shape[1];
	// the reference compares the whole sort_order (algorithm.hpp:360); for rank 2 the first entry decides
	c.sort0 = (A.sort_order[0] == 0 && A.sort_order[1] == 1) ? 0 : ((A.sort_order[0] == 1 && A.sort_order[1] == 0) ? 1 : -1);
	c.mem = SPSAMD_MEM_HOST;
	return c;
}

template <class VecT>
spsamd_vec as_vec(VecT const &V)
{
	static_assert(VecT::rank == 1, "scale operand must have rank 1");
	spsamd_vec v;
	size_t n = V.size();
	v.idx = n ? reinterpret_cast<const int32_t *>(&V.index(0, 0)) : nullptr;
	v.val = n ? &V.val(0) : nullptr;
	v.nnz = n;
	v.shape0 = V.shape[0];
	v.sort0 = V.sort_order[0];
	v.mem = SPSAMD_MEM_HOST;
	return v;
}

template <class AccumulatorT>
int add_chunk(void *user, const int32_t *i, const int32_t *j, const double *v, size_t n)
{
	AccumulatorT &ret = *static_cast<AccumulatorT *>(user);
	if constexpr (std::is_same<AccumulatorT, VectorCooArray<int32_t, double, 2>>::value) {
		const int32_t *cols[2] = {i, j};
		ret.add_tuples(cols, v, n);                                  // same tuples, same order, appended in bulk
	} else {
		for (size_t q = 0; q < n; ++q) ret.add({i[q], j[q]}, v[q]);  // multiply_sparse.hpp:242
	}
	return 0;
}

} // namespace detail

// ---- multiply, matrix x matrix: multiply_sparse.hpp:152-248 ---------------
template <class ScaleIT, class MatAT, class ScaleJT, class MatBT, class ScaleKT, class AccumulatorT>
void multiply(
	AccumulatorT &ret,
	double C,                // Multiply everything by this
	ScaleIT const *scalei,
	MatAT const &A,
	char transpose_A,        // 'T' for transpose, '.' otherwise
	ScaleJT const *scalej,
	MatBT const &B,
	char transpose_B,        // 'T' for transpose, '.' otherwise
	ScaleKT const *scalek,
	DuplicatePolicy duplicate_policy = DuplicatePolicy::ADD,
	bool zero_nan = false)
{
	// Set dimensions of output, even if we store nothing in it (multiply_sparse.hpp:166-169)
	std::array<int, 2> const &a_sort_order(transpose_A == 'T' ? COL_MAJOR : ROW_MAJOR);
	std::array<int, 2> const &b_sort_order(transpose_B == 'T' ? ROW_MAJOR : COL_MAJOR);
	ret.set_shape({A.shape[a_sort_order[0]], B.shape[b_sort_order[0]]});

	// Check inner dimensions (:172-174)
	if (A.shape[a_sort_order[1]] != B.shape[b_sort_order[1]]) {
		(*spsparse_error)(-1, "Inner dimensions for A (%ld) and B (%ld) must match!",
			(long)A.shape[a_sort_order[1]], (long)B.shape[b_sort_order[1]]);
		return;              // a user handler may return
	}

	// Short-circuit return on empty output (:178-184)
	if ((C == 0)
		|| (scalei && scalei->size() == 0)
		|| (A.size() == 0)
		|| (scalej && scalej->size() == 0)
		|| (B.size() == 0)
		|| (scalek && scalek->size() == 0))
	{ return; }

	spsamd_coo a = detail::as_coo(A), b = detail::as_coo(B);
	spsamd_vec si, sj, sk;
	if (scalei) si = detail::as_vec(*scalei);
	if (scalej) sj = detail::as_vec(*scalej);
	if (scalek) sk = detail::as_vec(*scalek);

	spsamd_ctx *ctx = default_context().get();
	if (!ctx) return;
	spsamd_result res;
	int rc = spsamd_multiply(ctx, C, scalei ? &si : nullptr, &a, transpose_A, scalej ? &sj : nullptr, &b, transpose_B,
		scalek ? &sk : nullptr, (int)duplicate_policy, zero_nan ? 1 : 0, SPSAMD_SINK_COO, multiply_flags, &res);
	if (rc != 0) { (*spsparse_error)(-1, "%s", spsamd_last_error(ctx)); return; }
	rc = spsamd_result_fetch(ctx, &res, &detail::add_chunk<AccumulatorT>, &ret);
	if (rc != 0) (*spsparse_error)(-1, "%s", spsamd_last_error(ctx));
}

template <class AccumulatorT>
int add_chunk_v(void *user, const int32_t *i, const int32_t *, const double *v, size_t n);

// ---- multiply, matrix x sparse vector: multiply_sparse.hpp:281-365 --------
template <class ScaleIT, class MatAT, class ScaleJT, class VecT, class AccumulatorT>
void multiply(
	AccumulatorT &ret,
	double C,                // Multiply everything by this
	ScaleIT const *scalei,
	MatAT const &A,
	char transpose_A,        // 'T' for transpose, '.' otherwise
	ScaleJT const *scalej,
	VecT const &V,
	DuplicatePolicy duplicate_policy = DuplicatePolicy::ADD,
	bool zero_nan = false)
{
	static_assert(VecT::rank == 1, "the right operand of the matrix-vector multiply has rank 1");
	std::array<int, 2> const &a_sort_order(transpose_A == 'T' ? COL_MAJOR : ROW_MAJOR);
	ret.set_shape({A.shape[a_sort_order[0]]});                       // :295
	if (A.shape[a_sort_order[1]] != V.shape[0]) {                    // :298-300
		(*spsparse_error)(-1, "Inner dimensions for A (%ld) and V (%ld) must match!",
			(long)A.shape[a_sort_order[1]], (long)V.shape[0]);
		return;
	}
	if ((C == 0) || (scalei && scalei->size() == 0) || (A.size() == 0)
		|| (scalej && scalej->size() == 0) || (V.size() == 0))       // :304-309
	{ return; }

	spsamd_coo a = detail::as_coo(A);
	spsamd_vec v = detail::as_vec(V), si, sj;
	if (scalei) si = detail::as_vec(*scalei);
	if (scalej) sj = detail::as_vec(*scalej);
	spsamd_ctx *ctx = default_context().get();
	if (!ctx) return;
	spsamd_result res;
	int rc = spsamd_multiply_mv(ctx, C, scalei ? &si : nullptr, &a, transpose_A, scalej ? &sj : nullptr, &v,
		(int)duplicate_policy, zero_nan ? 1 : 0, SPSAMD_SINK_COO, multiply_flags, &res);
	if (rc != 0) { (*spsparse_error)(-1, "%s", spsamd_last_error(ctx)); return; }
	rc = spsamd_result_fetch(ctx, &res, &add_chunk_v<AccumulatorT>, &ret);
	if (rc != 0) (*spsparse_error)(-1, "%s", spsamd_last_error(ctx));
}

template <class AccumulatorT>
int add_chunk_v(void *user, const int32_t *i, const int32_t *, const double *v, size_t n)
{
	AccumulatorT &ret = *static_cast<AccumulatorT *>(user);
	for (size_t q = 0; q < n; ++q) ret.add({i[q]}, v[q]);            // multiply_sparse.hpp:360
	return 0;
}

// ---- VectorCooArray::consolidate on the device ---------------------------
template <class IndexT, class ValT, int RANK>
void VectorCooArray<IndexT, ValT, RANK>::consolidate(std::array<int, RANK> const &_sort_order,
	DuplicatePolicy duplicate_policy, bool handle_nan)
{
	static_assert(RANK == 2, "device consolidate is implemented for matrices");
	if (this->sort_order == _sort_order && !edit_mode) return;       // VectorCooArray.hpp:306
	VectorCooArray ret(shape);
	if (size() > 0) {
		spsamd_coo a = detail::as_coo(*this);
		a.sort0 = -1;
		spsamd_ctx *ctx = default_context().get();
		spsamd_result res;
		int rc = spsamd_consolidate(ctx, &a, _sort_order[0], (int)duplicate_policy, handle_nan ? 1 : 0, &res);
		if (rc != 0) { (*spsparse_error)(-1, "%s", spsamd_last_error(ctx)); return; }
		ret.reserve((size_t)res.nnz);
		rc = spsamd_result_fetch(ctx, &res, &detail::add_chunk<VectorCooArray>, &ret);
		if (rc != 0) { (*spsparse_error)(-1, "%s", spsamd_last_error(ctx)); return; }
	}
	ret.set_sorted(_sort_order);
	*this = std::move(ret);
}

} // namespace spsparse_amd
