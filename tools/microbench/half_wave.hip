// tools/microbench/half_wave.hip -- does a wave64 vector instruction cost less when one 32-lane half (or all but a few lanes) is masked off?  (gfx950: SIMD-32, two passes per wave64 instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#define I_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
#define I_ADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b));
#define REP4(X) X X X X
template <int kOp>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long mask)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed * 1.0001f + 3.f, c = seed + 0.5f;
	if ((mask >> (threadIdx.x & 63)) & 1ull)
		for (int i = 0; i < iters; i++)
		{
			if (kOp == 0) { REP4(I_FMA(a0) I_FMA(a1) I_FMA(a2) I_FMA(a3) I_FMA(a4) I_FMA(a5) I_FMA(a6) I_FMA(a7)) }
			else { REP4(I_ADD(a0) I_ADD(a1) I_ADD(a2) I_ADD(a3) I_ADD(a4) I_ADD(a5) I_ADD(a6) I_ADD(a7)) }
		}
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[threadIdx.x] = a0;
}
int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	float* out; hipMalloc(&out, 4096);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const unsigned long long masks[] = { ~0ull, 0x00000000ffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull, 0x000000000000ffffull, 0x0000ffff0000ffffull, 0x1ull, 0x00ff00ff00ff00ffull };
	const char* names[] = { "all 64 lanes", "lanes 0-31", "lanes 32-63", "every other lane", "lanes 0-15", "lanes 0-15 + 32-47", "lane 0 only", "8 of every 16" };
	const int iters = 2000, w = 8;
	for (int op = 0; op < 2; op++)
		for (int mi = 0; mi < 8; mi++)
		{
			auto fn = op == 0 ? k<0> : k<1>;
			hipLaunchKernelGGL(fn, dim3(prop.multiProcessorCount * w), dim3(256), 0, 0, out, 10, 1.0f, masks[mi]); hipDeviceSynchronize();
			hipEventRecord(e0); hipLaunchKernelGGL(fn, dim3(prop.multiProcessorCount * w), dim3(256), 0, 0, out, iters, 1.0f, masks[mi]); hipEventRecord(e1); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			printf("%-10s %-22s %6.2f SIMD cycles per wave64 instruction (8 waves/SIMD, 2.4 GHz nominal)\n", op == 0 ? "v_fma_f32" : "v_add_f32", names[mi], ms * 1e-3 * 2.4e9 / ((double)iters * 32 * w));
		}
	return 0;
}
