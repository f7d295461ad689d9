// devutil.h -- wave64 / workgroup helpers shared by the kernels (gfx950: wavefront = 64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spsamd {

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Inclusive scan across the 64 lanes of a wave (generic: LDS-crossbar shuffles).
template <class T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		T o = __shfl_up(v, d, 64);
		if ((int)lane_id() >= d) v += o;
	}
	return v;
}

// uint32 inclusive scan with DPP row shifts and row broadcasts (gfx9 family): six
// VALU adds instead of six dependent ds_bpermute round trips.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);      // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);      // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);      // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);      // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);      // row_bcast:15 -> rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);      // row_bcast:31 -> rows 2 and 3
	return (uint32_t)x;
}

// Inclusive scan inside aligned groups of S lanes (S = 8, 16, 32, 64), DPP only.
// s = lane index inside the group.
template <int S>
__device__ __forceinline__ uint32_t group_inclusive_scan_u32(uint32_t v, unsigned s)
{
	int x = (int)v, t;
	t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); if (s >= 1) x += t;      // row_shr:1 (rows of 16 lanes)
	t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); if (s >= 2) x += t;
	t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); if (s >= 4) x += t;
	if (S >= 16) { t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); if (s >= 8) x += t; }
	if (S >= 32) { t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); if (s >= 16) x += t; }   // row_bcast:15
	if (S >= 64) { t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); if (s >= 32) x += t; }   // row_bcast:31
	return (uint32_t)x;
}

template <class T>
__device__ __forceinline__ T wave_reduce_sum(T v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

// Exclusive scan across a workgroup of NT threads (NT multiple of 64, <= 1024).
// scratch: NT/64 + 1 entries of T in LDS.  Returns the exclusive prefix of v;
// *total (if non-null) receives the workgroup sum.  Contains three barriers.
template <class T, int NT>
__device__ __forceinline__ T block_exclusive_scan(T v, T *scratch, T *total)
{
	constexpr int NW = NT / 64;
	T inc = wave_inclusive_scan(v);
	if (lane_id() == 63) scratch[wave_id()] = inc;
	__syncthreads();
	if (threadIdx.x == 0) {
		T run = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w) { T t = scratch[w]; scratch[w] = run; run += t; }
		scratch[NW] = run;
	}
	__syncthreads();
	T base = scratch[wave_id()];
	T tot = scratch[NW];
	__syncthreads();            // scratch may be reused right away
	if (total) *total = tot;
	return base + inc - v;
}

// Index hash of the digest sink: one 64-bit multiply and a fold of (i, j).
// Same arithmetic as orc_mix64 in the test oracle.
__host__ __device__ __forceinline__ uint64_t mix64(uint32_t i, uint32_t j)
{
	uint64_t x = (((uint64_t)i << 32) | (uint64_t)j) * 0x9E3779B97F4A7C15ull;
	return x ^ (x >> 29);
}

} // namespace spsamd
