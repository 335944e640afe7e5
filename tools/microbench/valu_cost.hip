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
#define I_SUB(r)      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MIN(r)      asm volatile("v_min_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MAXI(r)     asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MINI(r)     asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MAXU(r)     asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MAX3I(r)    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_MIN3I(r)    asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_OR(r)       asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_XOR(r)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_LSHR(r)     asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r));
#define I_ANDOR(r)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_LSHLOR(r)   asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r) : "v"(b));
#define I_ORSDWA(r)   asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r) : "v"(b));
#define I_ADDSDWA(r)  asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(b));
#define I_CVTSDWA(r)  asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(r));
#define I_PERM(r)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_MULLO(r)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MULHI(r)    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MUL24(r)    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_MAD24(r)    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_CMPI(r)     asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");
#define I_CMPX(r)     asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");
#define I_ADDE64(r)   asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(r) : "v"(b));
#define I_MULS(r)     asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "s"(c));
#define I_FMAS(r)     asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "s"(c), "v"(b));
#define I_CNDE32(r)   asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r) : "v"(b));
#define I_RCPI(r)     asm volatile("v_rcp_iflag_f32 %0, %0" : "+v"(r));
#define I_LDEXP(r)    asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define I_FRACT(r)    asm volatile("v_fract_f32 %0, %0" : "+v"(r));
#define I_NOP(r)      asm volatile("s_nop 0");
#define I_SAND(r)     asm volatile("s_and_b64 %0, %0, exec" : "+s"(m));
// a dependent chain: every instruction waits for the one before it
#define I_DEPADD(r)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
#define I_DEPFMA(r)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));

KERNEL(k_add, I_ADD) KERNEL(k_mul, I_MUL) KERNEL(k_fma, I_FMA) KERNEL(k_fmac, I_FMAC) KERNEL(k_max, I_MAX) KERNEL(k_max3, I_MAX3) KERNEL(k_med3, I_MED3)
KERNEL(k_cndvcc, I_CNDVCC) KERNEL(k_cnds, I_CNDS) KERNEL(k_cmpvcc, I_CMPVCC) KERNEL(k_cmps, I_CMPS)
KERNEL(k_cvtub, I_CVTUB) KERNEL(k_cvtu32, I_CVTU32) KERNEL(k_and, I_AND) KERNEL(k_lshl, I_LSHL) KERNEL(k_bfe, I_BFE) KERNEL(k_addu, I_ADDU) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_add3, I_ADD3)
KERNEL(k_rcp, I_RCP) KERNEL(k_sqrt, I_SQRT) KERNEL(k_divscale, I_DIVSCALE) KERNEL(k_divfmas, I_DIVFMAS) KERNEL(k_divfixup, I_DIVFIXUP)
KERNEL(k_sub, I_SUB) KERNEL(k_min, I_MIN) KERNEL(k_maxi, I_MAXI) KERNEL(k_mini, I_MINI) KERNEL(k_maxu, I_MAXU) KERNEL(k_max3i, I_MAX3I) KERNEL(k_min3i, I_MIN3I) KERNEL(k_or, I_OR) KERNEL(k_xor, I_XOR) KERNEL(k_lshr, I_LSHR)
KERNEL(k_andor, I_ANDOR) KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_orsdwa, I_ORSDWA) KERNEL(k_addsdwa, I_ADDSDWA) KERNEL(k_cvtsdwa, I_CVTSDWA) KERNEL(k_perm, I_PERM) KERNEL(k_mullo, I_MULLO) KERNEL(k_mulhi, I_MULHI) KERNEL(k_mul24, I_MUL24) KERNEL(k_mad24, I_MAD24)
KERNEL(k_cmpi, I_CMPI) KERNEL(k_cmpx, I_CMPX) KERNEL(k_adde64, I_ADDE64) KERNEL(k_muls, I_MULS) KERNEL(k_fmas, I_FMAS) KERNEL(k_cnde32, I_CNDE32) KERNEL(k_rcpi, I_RCPI) KERNEL(k_ldexp, I_LDEXP) KERNEL(k_fract, I_FRACT)
KERNEL(k_mov, I_MOV) KERNEL(k_mbcnt, I_MBCNT) KERNEL(k_nop, I_NOP) KERNEL(k_sand, I_SAND) KERNEL(k_depadd, I_DEPADD) KERNEL(k_depfma, I_DEPFMA)

// packed fp32 (64-bit register pairs) and fp64
#define KERNEL2(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed) \
{ \
	double p0 = seed + threadIdx.x, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3, p4 = p0 + 4, p5 = p0 + 5, p6 = p0 + 6, p7 = p0 + 7, pb = seed * 1.0001 + 3., pc = seed + 0.5; \
	for (int i = 0; i < iters; i++) { REP4(ASM(p0) ASM(p1) ASM(p2) ASM(p3) ASM(p4) ASM(p5) ASM(p6) ASM(p7)) } \
	if (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 == 1.2345) out[threadIdx.x] = (float)p0; \
}
#define J_PKFMA(r)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(pb), "v"(pc));
#define J_PKMUL(r)  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(pb));
#define J_PKADD(r)  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(pb));
#define J_FMA64(r)  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(pb), "v"(pc));
#define J_MUL64(r)  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(pb));
#define J_ADD64(r)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(r) : "v"(pb));
#define J_ADDU64(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(pb));
KERNEL2(k_pkfma, J_PKFMA) KERNEL2(k_pkmul, J_PKMUL) KERNEL2(k_pkadd, J_PKADD) KERNEL2(k_fma64, J_FMA64) KERNEL2(k_mul64, J_MUL64) KERNEL2(k_add64, J_ADD64) KERNEL2(k_addu64, J_ADDU64)

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
		{"v_sub_f32", k_sub, 32}, {"v_min_f32", k_min, 32}, {"v_max_i32", k_maxi, 32}, {"v_min_i32", k_mini, 32}, {"v_max_u32", k_maxu, 32}, {"v_max3_i32", k_max3i, 32}, {"v_min3_i32", k_min3i, 32},
		{"v_or_b32", k_or, 32}, {"v_xor_b32", k_xor, 32}, {"v_lshrrev_b32", k_lshr, 32}, {"v_and_or_b32", k_andor, 32}, {"v_lshl_or_b32", k_lshlor, 32}, {"v_or_b32_sdwa BYTE_1", k_orsdwa, 32}, {"v_add_f32_sdwa", k_addsdwa, 32},
		{"v_cvt_f32_u32_sdwa BYTE_2", k_cvtsdwa, 32}, {"v_perm_b32", k_perm, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_mul_hi_u32", k_mulhi, 32}, {"v_mul_u32_u24", k_mul24, 32}, {"v_mad_u32_u24", k_mad24, 32},
		{"v_cmp_lt_i32 -> vcc", k_cmpi, 32}, {"v_cmp_class_f32", k_cmpx, 32}, {"v_add_f32_e64 (neg mod)", k_adde64, 32}, {"v_mul_f32 (sgpr src)", k_muls, 32}, {"v_fma_f32 (sgpr src)", k_fmas, 32},
		{"v_cndmask_b32_e32 vcc", k_cnde32, 32}, {"v_rcp_iflag_f32", k_rcpi, 32}, {"v_ldexp_f32", k_ldexp, 32}, {"v_fract_f32", k_fract, 32},
		{"v_pk_fma_f32 (2 fma)", k_pkfma, 32}, {"v_pk_mul_f32 (2 mul)", k_pkmul, 32}, {"v_pk_add_f32 (2 add)", k_pkadd, 32}, {"v_fma_f64", k_fma64, 32}, {"v_mul_f64", k_mul64, 32}, {"v_add_f64", k_add64, 32}, {"v_lshl_add_u64", k_addu64, 32},
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
