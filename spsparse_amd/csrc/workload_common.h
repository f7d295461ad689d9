// workload_common.h -- the counter-based random stream of the synthetic
// operands; same arithmetic as spsparse_amd/workloads.py (bit-identical).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SPS_HD __host__ __device__ __forceinline__
#else
#define SPS_HD static inline
#endif

namespace spsamd {

SPS_HD uint64_t wl_splitmix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

SPS_HD uint64_t wl_stream_key(uint64_t seed, uint64_t stream)
{
	return wl_splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull));
}

SPS_HD uint64_t wl_draw(uint64_t key, uint64_t ctr) { return wl_splitmix64(key ^ wl_splitmix64(ctr)); }

// (0,1]: never 0, so no generated tuple is dropped by consolidate
SPS_HD double wl_unit_open(uint64_t r) { return (double)((r >> 11) + 1ull) * 0x1.0p-53; }

// R-MAT quadrant thresholds on a 16-bit draw: a,b,c,d = 0.57,0.19,0.19,0.05
constexpr uint32_t RMAT_TA = 37356, RMAT_TAB = 49808, RMAT_TABC = 62260;

} // namespace spsamd
