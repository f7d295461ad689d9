// Microbenchmark (dev tool): rate of 12-byte segment gathers (the product loops' B reads) on gfx950
// as a function of footprint (L2 / Infinity Cache / HBM), segment length, loads in flight and occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) BTup { int32_t col; uint32_t vlo, vhi; };

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int U>
__global__ __launch_bounds__(512) void k(const BTup *b, uint32_t ntup, uint32_t segsh, int iters, double *out)
{
	extern __shared__ double pad[];
	const uint32_t seg = 1u << segsh;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t grp = (blockIdx.x * 512u + threadIdx.x) >> segsh;     // lanes of one segment share the start
	const uint32_t off = lane & (seg - 1u);
	const uint32_t range = ntup - seg;
	double acc = 0; uint32_t cacc = 0;
	for (int it = 0; it < iters; ++it) {
		BTup t[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			uint32_t r = mixu(grp * 0x9E3779B1u + (uint32_t)(it * U + u) * 0x85EBCA6Bu);
			uint32_t start = (uint32_t)(((uint64_t)r * range) >> 32);
			t[u] = b[start + off];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) { cacc += (uint32_t)t[u].col; acc += __hiloint2double((int)t[u].vhi, (int)t[u].vlo); }
	}
	if (acc == 1.2345 && cacc == 77) out[0] = acc + pad[0];
}

template <int U>
int run(const BTup *b, double *out, uint64_t bytes, uint32_t segsh, int wg_per_cu)
{
	uint32_t ntup = (uint32_t)(bytes / 12);
	int iters = 2048 / U;
	int grid = 256 * wg_per_cu;
	size_t lds = 160 * 1024 / wg_per_cu - 1024;            // forces the occupancy
	if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)k<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
	k<U><<<grid, 512, lds>>>(b, ntup, segsh, 64, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<U><<<grid, 512, lds>>>(b, ntup, segsh, iters, out);
	CK(hipEventRecord(z)); CK(hipEventSynchronize(z));
	float ms; CK(hipEventElapsedTime(&ms, a, z));
	double loads = (double)grid * 512 * iters * U;
	printf("footprint %7.1f MB seg %3u U %d waves/CU %2d: %7.3f ms  %.3g tuples/s  %.0f GB/s useful\n", bytes / 1e6, 1u << segsh, U, wg_per_cu * 8, ms,
		loads / (ms * 1e-3), loads * 12 / (ms * 1e-3) / 1e9);
	return 0;
}

int main()
{
	const uint64_t maxb = 4ull << 30;
	BTup *b; double *out;
	CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 8));
	CK(hipMemset(b, 1, maxb));
	const uint64_t fps[] = {2ull << 20, 16ull << 20, 28ull << 20, 193ull << 20, 1ull << 30, 4ull << 30};
	for (uint64_t fp : fps)
		for (uint32_t segsh : {2u, 4u, 6u}) {
			run<2>(b, out, fp, segsh, 2);
			run<4>(b, out, fp, segsh, 2);
			run<2>(b, out, fp, segsh, 4);
			run<8>(b, out, fp, segsh, 4);
		}
	return 0;
}
