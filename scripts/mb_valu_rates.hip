// Microbenchmark (dev tool): issue cost of the VALU instructions the SpGEMM kernels lean on (gfx950).
// Each kernel runs ITER iterations of 8 independent instances of one instruction per wave; 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
	uint32_t a[8], b[8]; uint64_t w[8]; double d[8];
	for (int q = 0; q < 8; ++q) { a[q] = seed + threadIdx.x * 7 + q; b[q] = seed * 3 + q; w[q] = a[q]; d[q] = 1.0 + q; }
	for (int it = 0; it < iters; ++it) {
#define S(q) \
		if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[q]) : "v"(a[q]), "v"(b[q]) : "vcc"); \
		if (OP == 4) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[q])); \
		if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[q]) : "v"(d[(q + 1) & 7])); \
		if (OP == 6) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 8) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[q]) : "v"(d[(q + 1) & 7])); \
		if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 10) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(w[q])); \
		if (OP == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(d[(q + 1) & 7])); \
		if (OP == 12) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q])); \
		if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(a[q]) : "v"(b[q]));
		REP8(S)
#undef S
	}
	uint32_t r = 0;
	for (int q = 0; q < 8; ++q) r += a[q] + (uint32_t)w[q] + (uint32_t)d[q];
	if (r == 0x12345678u) out[0] = r;
}

template <int OP>
int run(const char *name)
{
	uint32_t *out; CK(hipMalloc(&out, 4));
	const int iters = 20000, grid = 256 * 4;                 // 4 workgroups of 4 waves per CU: 4 waves per SIMD
	hipEvent_t a, z; CK(hipEventCreate(&a)); CK(hipEventCreate(&z));
	k<OP><<<grid, 256>>>(out, 100, 1);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<OP><<<grid, 256>>>(out, iters, 1);
	CK(hipEventRecord(z)); CK(hipEventSynchronize(z));
	float ms; CK(hipEventElapsedTime(&ms, a, z));
	double wave_instrs_per_simd = (double)iters * 8 * 4;     // 4 waves per SIMD
	printf("%-22s %7.3f ms  -> %.2f clk per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instrs_per_simd);
	CK(hipFree(out));
	return 0;
}

int main()
{
	run<0>("v_add_u32"); run<1>("v_mul_lo_u32"); run<7>("v_mul_hi_u32"); run<2>("v_mul_u32_u24"); run<9>("v_mad_u32_u24");
	run<3>("v_mad_u64_u32"); run<4>("v_lshlrev_b64"); run<10>("v_lshl_add_u64"); run<6>("v_bcnt_u32_b32"); run<12>("v_mbcnt_lo_u32_b32");
	run<13>("v_cndmask_b32"); run<14>("v_xor_b32"); run<15>("v_alignbit_b32");
	run<5>("v_mul_f64"); run<8>("v_fma_f64"); run<11>("v_add_f64");
	return 0;
}
