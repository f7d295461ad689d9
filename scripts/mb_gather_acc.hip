// Microbenchmark (dev tool): the dense cells' product step in isolation -- gather a 12-byte B tuple from a
// random segment, multiply, ds_add_f64 into a 64 KB window accumulator -- with U tuples per thread and step,
// optionally with a dependent LDS lookup in front of the gather (as the segment lookup is).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) BTup { int32_t col; uint32_t vlo, vhi; };
__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int U, int LOOKUP>
__global__ __launch_bounds__(512) void k(const BTup *b, uint32_t ntup, uint32_t segsh, int iters, double *out)
{
	__shared__ double acc[8192];
	__shared__ uint32_t tab[1024];
	for (int q = threadIdx.x; q < 8192; q += 512) acc[q] = 0;
	for (int q = threadIdx.x; q < 1024; q += 512) tab[q] = mixu(q) & 1023u;
	__syncthreads();
	const uint32_t seg = 1u << segsh, lane = threadIdx.x & 63u;
	const uint32_t grp = (blockIdx.x * 512u + threadIdx.x) >> segsh, off = lane & (seg - 1u);
	const uint32_t range = ntup - seg;
	for (int it = 0; it < iters; ++it) {
		BTup t[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			uint32_t r = mixu(grp * 0x9E3779B1u + (uint32_t)(it * U + u) * 0x85EBCA6Bu);
			if (LOOKUP) { r ^= tab[r & 1023u]; r ^= tab[(r >> 10) & 1023u] << 10; }      // two dependent LDS reads
			uint32_t start = (uint32_t)(((uint64_t)r * range) >> 32);
			t[u] = b[start + off];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) atomicAdd(&acc[((uint32_t)t[u].col + (uint32_t)(it * 7 + u * 131 + threadIdx.x * 29)) & 8191u], __hiloint2double((int)t[u].vhi, (int)t[u].vlo));
	}
	__syncthreads();
	double s = 0;
	for (int q = threadIdx.x; q < 8192; q += 512) s += acc[q];
	if (s == 1.2345) out[0] = s;
}

template <int U, int LOOKUP>
int run(const BTup *b, double *out, uint64_t bytes, uint32_t segsh)
{
	uint32_t ntup = (uint32_t)(bytes / 12);
	int iters = 4096 / U, grid = 512;
	hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
	k<U, LOOKUP><<<grid, 512>>>(b, ntup, segsh, 64, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<U, LOOKUP><<<grid, 512>>>(b, ntup, segsh, iters, out);
	CK(hipEventRecord(z)); CK(hipEventSynchronize(z));
	float ms; CK(hipEventElapsedTime(&ms, a, z));
	double n = (double)grid * 512 * iters * U;
	printf("footprint %6.1f MB seg %3u U %d lookup %d: %7.3f ms  %.3g products/s (%.2f ps per product)\n", bytes / 1e6, 1u << segsh, U, LOOKUP, ms, n / (ms * 1e-3), ms * 1e-3 / n * 1e12);
	return 0;
}

int main()
{
	const uint64_t maxb = 256ull << 20;
	BTup *b; double *out;
	CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 8));
	CK(hipMemset(b, 1, maxb));
	for (uint64_t fp : {2ull << 20, 193ull << 20})
		for (uint32_t segsh : {3u, 6u}) {
			run<1, 0>(b, out, fp, segsh); run<2, 0>(b, out, fp, segsh); run<4, 0>(b, out, fp, segsh);
			run<1, 1>(b, out, fp, segsh); run<2, 1>(b, out, fp, segsh); run<4, 1>(b, out, fp, segsh);
		}
	return 0;
}
