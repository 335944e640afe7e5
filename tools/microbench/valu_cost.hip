// tools/microbench/valu_cost.hip -- issue cost of the vector instructions the path-tracing kernels are made of, on gfx950.
// Every wave runs N iterations of a body of 32 INDEPENDENT instances of one instruction (8 register sets, 4 rounds); the grid fills
// every SIMD with W waves (W = 1, 2, 4, 8).  Reported: SIMD cycles per wave64 instruction = elapsed cycles * 1024 SIMDs / instructions.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define BODY8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP4(X) X X X X

#define KERNEL(NAME, INSTR) \
__global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed) \
{ \
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	float b = seed * 1.0001f + 3.f, c = seed + 0.5f; unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)seed; \
	for (int i = 0; i < iters; i++) { REP4(INSTR(a0) INSTR(a1) INSTR(a2) INSTR(a3) INSTR(a4) INSTR(a5) INSTR(a6) INSTR(a7)) } \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[threadIdx.x] = a0 + b + c + (float)m; \
}

#define I_ADD(r)      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MUL(r)      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_FMA(r)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_FMAC(r)     asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_MAX(r)      asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MAX3(r)     asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_CNDVCC(r)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : );
#define I_CNDS(r)     asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "s"(m));
#define I_CMPVCC(r)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");
#define I_CMPS(r)     asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(r), "v"(b));
#define I_CVTUB(r)    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r));
#define I_CVTU32(r)   asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r));
#define I_AND(r)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_LSHL(r)     asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r));
#define I_BFE(r)      asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r));
#define I_ADDU(r)     asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_LSHLADD(r)  asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(b));
#define I_ADD3(r)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_RCP(r)      asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define I_SQRT(r)     asm volatile("v_sqrt_f32 %0, %0" : "+v"(r));
#define I_DIVSCALE(r) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r) : "v"(b) : "vcc");
#define I_DIVFMAS(r)  asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c) : );
#define I_DIVFIXUP(r) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_PKFMA(r)    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##r) : "v"(pb), "v"(pc));
#define I_MOV(r)      asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(b));
#define I_MBCNT(r)    asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r) : "v"(b));
#define I_MED3(r)     asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_SUBREV(r)   asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(r) : "v"(b));
#define I_NOP(r)      asm volatile("s_nop 0");
#define I_SAND(r)     asm volatile("s_and_b64 %0, %0, exec" : "+s"(m));
// a dependent chain: every instruction waits for the one before it
#define I_DEPADD(r)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
#define I_DEPFMA(r)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));

KERNEL(k_add, I_ADD) KERNEL(k_mul, I_MUL) KERNEL(k_fma, I_FMA) KERNEL(k_fmac, I_FMAC) KERNEL(k_max, I_MAX) KERNEL(k_max3, I_MAX3) KERNEL(k_med3, I_MED3)
KERNEL(k_cndvcc, I_CNDVCC) KERNEL(k_cnds, I_CNDS) KERNEL(k_cmpvcc, I_CMPVCC) KERNEL(k_cmps, I_CMPS)
KERNEL(k_cvtub, I_CVTUB) KERNEL(k_cvtu32, I_CVTU32) KERNEL(k_and, I_AND) KERNEL(k_lshl, I_LSHL) KERNEL(k_bfe, I_BFE) KERNEL(k_addu, I_ADDU) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_add3, I_ADD3)
KERNEL(k_rcp, I_RCP) KERNEL(k_sqrt, I_SQRT) KERNEL(k_divscale, I_DIVSCALE) KERNEL(k_divfmas, I_DIVFMAS) KERNEL(k_divfixup, I_DIVFIXUP)
KERNEL(k_mov, I_MOV) KERNEL(k_mbcnt, I_MBCNT) KERNEL(k_nop, I_NOP) KERNEL(k_sand, I_SAND) KERNEL(k_depadd, I_DEPADD) KERNEL(k_depfma, I_DEPFMA)

// the real thing: IEEE fp32 division as hipcc expands it, 8 independent quotients per round
__global__ void __launch_bounds__(256) k_ieee_div(float* out, int iters, float seed)
{
	float a[8]; for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x + k; float b = seed * 1.0001f + 3.f;
	for (int i = 0; i < iters; i++) { for (int r = 0; r < 4; r++) for (int k = 0; k < 8; k++) a[k] = a[k] / b; }
	float s = 0; for (int k = 0; k < 8; k++) s += a[k]; if (s == 1.2345f) out[threadIdx.x] = s;
}

typedef void (*KFn)(float*, int, float);
struct Entry { const char* name; KFn fn; int per_body; };

int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate / 1e6;
	printf("device %s, %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
	float* out; hipMalloc(&out, 4096);
	std::vector<Entry> es = {
		{"v_add_f32", k_add, 32}, {"v_mul_f32", k_mul, 32}, {"v_fma_f32", k_fma, 32}, {"v_fmac_f32", k_fmac, 32}, {"v_max_f32", k_max, 32}, {"v_max3_f32", k_max3, 32}, {"v_med3_f32", k_med3, 32},
		{"v_cndmask vcc", k_cndvcc, 32}, {"v_cndmask sgpr", k_cnds, 32}, {"v_cmp -> vcc", k_cmpvcc, 32}, {"v_cmp -> sgpr", k_cmps, 32},
		{"v_cvt_f32_ubyte1", k_cvtub, 32}, {"v_cvt_f32_u32", k_cvtu32, 32}, {"v_and_b32", k_and, 32}, {"v_lshlrev_b32", k_lshl, 32}, {"v_bfe_u32", k_bfe, 32}, {"v_add_u32", k_addu, 32}, {"v_lshl_add_u32", k_lshladd, 32}, {"v_add3_u32", k_add3, 32},
		{"v_rcp_f32", k_rcp, 32}, {"v_sqrt_f32", k_sqrt, 32}, {"v_div_scale_f32", k_divscale, 32}, {"v_div_fmas_f32", k_divfmas, 32}, {"v_div_fixup_f32", k_divfixup, 32},
		{"v_mov_b32", k_mov, 32}, {"v_mbcnt_lo", k_mbcnt, 32}, {"s_nop 0", k_nop, 32}, {"s_and_b64", k_sand, 32}, {"dependent v_add_f32", k_depadd, 32}, {"dependent v_fma_f32", k_depfma, 32},
		{"IEEE a / b (whole expansion)", k_ieee_div, 32},
	};
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	printf("%-30s", "SIMD cycles per wave64 instr:"); for (int w : {1, 2, 4, 8}) printf("  %d wave%s/SIMD", w, w > 1 ? "s" : " "); printf("\n");
	const int iters = 2000;
	for (const Entry& e : es)
	{
		printf("%-30s", e.name);
		for (int w : {1, 2, 4, 8})
		{
			const int grid = cus * w;                  // w workgroups of 256 per CU = w waves per SIMD
			hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
			hipDeviceSynchronize();
			hipEventRecord(e0); hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			const double instr_per_simd = (double)iters * e.per_body * w;
			printf("  %12.2f", ms * 1e-3 * 2.4e9 / instr_per_simd);   // at 2.4 GHz (the clock is not measured here: ratios between rows are what matters)
		}
		printf("\n");
	}
	return 0;
}
