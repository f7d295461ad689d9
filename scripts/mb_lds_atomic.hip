// Microbenchmark (dev tool): LDS f64 accumulate rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int W, int NT>
__global__ __launch_bounds__(NT) void k(const uint32_t *idx, double *out, int iters)
{
	__shared__ double acc[W];
	for (int i = threadIdx.x; i < W; i += NT) acc[i] = 0;
	__syncthreads();
	uint32_t x = idx[blockIdx.x * NT + threadIdx.x];
	double v = 1.0 + threadIdx.x * 1e-9;
	for (int it = 0; it < iters; ++it) {
		x = x * 1664525u + 1013904223u;
		uint32_t slot = (MODE == 3) ? ((threadIdx.x + it * 64) & (W - 1)) : ((x >> 8) & (W - 1));
		if (MODE == 0 || MODE == 3) atomicAdd(&acc[slot], v);                 // ds_add_f64 (no return)
		else if (MODE == 1) { double o = acc[slot]; acc[slot] = o + v; }      // racy RMW (rate reference only)
		else if (MODE == 2) { unsigned long long *p = (unsigned long long *)&acc[slot]; unsigned long long o = *p, a;
			do { a = o; o = atomicCAS(p, a, (unsigned long long)__double_as_longlong(__longlong_as_double(a) + v)); } while (o != a); }
		else if (MODE == 4) { float *f = (float *)acc; atomicAdd(&f[slot], (float)v); }   // ds_add_f32
		else if (MODE == 5) { unsigned int *u = (unsigned int *)acc; atomicAdd(&u[slot], 1u); } // ds_add_u32
	}
	__syncthreads();
	double s = 0;
	for (int i = threadIdx.x; i < W; i += NT) s += acc[i];
	if (s == 12345.678) out[0] = s;
}

template <int MODE, int W, int NT>
int run(const char *name, int wg_per_cu)
{
	int iters = 4096;
	int grid = 256 * wg_per_cu;
	uint32_t *idx; double *out;
	CK(hipMalloc(&idx, grid * NT * 4)); CK(hipMalloc(&out, 8));
	std::vector<uint32_t> h(grid * NT);
	for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u);
	CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	k<MODE, W, NT><<<grid, NT>>>(idx, out, 16);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<MODE, W, NT><<<grid, NT>>>(idx, out, iters);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	double ops = (double)grid * NT * iters;
	printf("%-34s W=%5d NT=%4d wg/cu=%d: %.3f ms  %.3g ops/s  (%.3f ops/clk/CU @2.4GHz)\n", name, W, NT, wg_per_cu, ms, ops / (ms * 1e-3),
		ops / (ms * 1e-3) / 256 / 2.4e9);
	CK(hipFree(idx)); CK(hipFree(out));
	return 0;
}

int main()
{
	run<0, 8192, 512>("ds_add_f64 random", 2);
	run<0, 8192, 256>("ds_add_f64 random", 2);
	run<0, 4096, 256>("ds_add_f64 random", 4);
	run<3, 8192, 512>("ds_add_f64 conflict-free", 2);
	run<1, 8192, 512>("read+add+write f64 (racy)", 2);
	run<2, 8192, 512>("CAS loop f64", 2);
	run<4, 8192, 512>("ds_add_f32 random", 2);
	run<5, 8192, 512>("ds_add_u32 random", 2);
	return 0;
}
