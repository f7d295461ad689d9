// Microbenchmark (dev tool): how a wave should fetch its 64 48-byte PIECES (4 consecutive 12-byte B tuples per lane, each lane's
// piece somewhere else -- the dense cells' B reads, k_dense.hip) on gfx950.
//   own   : every lane reads its own piece with three 16-byte loads (what k_dense does)
//   quad  : the four lanes of a quad read the quad's 4 pieces = 12 chunks of 16 bytes, lane q chunks q, q + 4, q + 8
//           (an instruction touches <= 2 pieces per quad instead of 4); the exchange back to the owners is not timed here
//   tuple : every lane reads ONE tuple (12 bytes), the quad's four lanes consecutive tuples of one piece, four such loads
//   flat  : 64 pieces contiguous (fully coalesced; the ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct __attribute__((packed, aligned(4))) T3 { uint32_t a, b, c; };

template <int MODE>
__global__ __launch_bounds__(512) void k(const char *b, uint32_t ntup, int iters, uint32_t *out)
{
	extern __shared__ double pad[];
	const uint32_t gt = blockIdx.x * 512u + threadIdx.x, lane = threadIdx.x & 63u, q = lane & 3u;
	const uint32_t range = ntup - 512u;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		auto start_of = [&](uint32_t who, uint32_t salt) { return (uint32_t)(((uint64_t)mixu(who * 0x9E3779B1u + (uint32_t)(it * 4 + salt) * 0x85EBCA6Bu) * range) >> 32); };
		if (MODE == 0) {
			const char *p = b + (uint64_t)start_of(gt, 0) * 12u;
			uint4 x0 = *(const uint4 *)(p), x1 = *(const uint4 *)(p + 16), x2 = *(const uint4 *)(p + 32);
			acc += x0.x ^ x0.w ^ x1.y ^ x1.w ^ x2.x ^ x2.z;
		} else if (MODE == 1) {
			uint4 x[3];
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const uint32_t g = 4u * j + q, piece = g / 3u, chunk = g % 3u;
				const char *p = b + (uint64_t)start_of((gt & ~3u) + piece, 0) * 12u + chunk * 16u;
				x[j] = *(const uint4 *)p;
			}
			acc += x[0].x ^ x[0].w ^ x[1].y ^ x[1].w ^ x[2].x ^ x[2].z;
		} else if (MODE == 2) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const char *p = b + (uint64_t)(start_of((gt & ~3u) + (uint32_t)j, 0) + q) * 12u;      // (wave step j: piece j of the quad)
				T3 t = *(const T3 *)p;
				acc += t.a ^ t.c;
			}
		} else {
			const char *p = b + (uint64_t)start_of(gt >> 6, 0) * 12u + lane * 48u;
			uint4 x0 = *(const uint4 *)(p), x1 = *(const uint4 *)(p + 16), x2 = *(const uint4 *)(p + 32);
			acc += x0.x ^ x0.w ^ x1.y ^ x1.w ^ x2.x ^ x2.z;
		}
	}
	if (acc == 0x12345u) out[0] = acc + (uint32_t)pad[0];
}

template <int MODE>
int run(const char *name, const char *b, uint32_t *out, uint64_t bytes, int wg_per_cu)
{
	const uint32_t ntup = (uint32_t)(bytes / 12);
	const int iters = 2048, grid = 256 * wg_per_cu;
	const size_t lds = 160 * 1024 / wg_per_cu - 1024;
	if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
	k<MODE><<<grid, 512, lds>>>(b, ntup, 64, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<MODE><<<grid, 512, lds>>>(b, ntup, iters, out);
	CK(hipEventRecord(z)); CK(hipEventSynchronize(z));
	float ms; CK(hipEventElapsedTime(&ms, a, z));
	const double pieces = (double)grid * 512 * iters;
	printf("%-6s footprint %7.1f MB waves/CU %2d: %7.3f ms  %.3g pieces/s = %.3g tuples/s  %.0f GB/s\n", name, bytes / 1e6, wg_per_cu * 8, ms,
		pieces / (ms * 1e-3), 4 * pieces / (ms * 1e-3), pieces * 48 / (ms * 1e-3) / 1e9);
	return 0;
}

int main()
{
	const uint64_t maxb = 1ull << 30;
	char *b; uint32_t *out;
	CK(hipMalloc(&b, maxb + 4096)); CK(hipMalloc(&out, 8));
	CK(hipMemset(b, 1, maxb));
	for (uint64_t fp : {2ull << 20, 28ull << 20, 193ull << 20})
		for (int w : {2, 4}) {
			run<0>("own", b, out, fp, w); run<1>("quad", b, out, fp, w); run<2>("tuple", b, out, fp, w); run<3>("flat", b, out, fp, w);
		}
	return 0;
}
