// jet-pbrt_amd/csrc/jp_libm.h -- logf / expf / powf / acosf / atanf / tanf exactly as glibc 2.35 computes them, for the device.
//
// The reference's BeckmannDistribution, FPhongSpecularReflection and the full-distribution sampling branches (microfacet.cc:11-167,
// 216-254, 326-357; bsdf.h:557-633) take these six functions from the host's libm.  They are accurate to under one ulp but not
// correctly rounded, so "evaluate precisely and round" does not reproduce them; what does is running the same algorithm in the same
// IEEE arithmetic, as jp_shading.h does for sinf / cosf / sincosf:
//   * logf, expf, powf (glibc >= 2.27, sysdeps/ieee754/flt-32/e_{log,exp,pow}f.c, from the Arm optimized routines): table + short
//     polynomial in fp64, rounded once to fp32.  x86-64 libm carries two builds, an IFUNC picks the one compiled with FMA contraction
//     on CPUs with FMA + AVX2: kFma selects which build is reproduced (every "a * b + c" of the routine one fma, or none).
//   * acosf, atanf, tanf (glibc 2.35: the fdlibm float routines e_acosf.c, s_atanf.c, s_tanf.c + k_tanf.c; the argument reduction of
//     e_rem_pio2f.c is the fp64 reduce_fast of sinf / cosf): plain fp32 arithmetic, one build, no contraction.
// Every function here is host + device code (the same arithmetic on both sides); jp_create_context probes them against the host's libm
// and the device only uses a function that reproduced the host on every probe argument (JpBuildInfo.libm_xbsdf bit mask), else the
// device library's function stays (tests/test_gpu_parity.py then applies the stated tolerance).  The tables and coefficients are the
// published ones of those routines; tests/test_libm_exact.py checks all of it against the running libm on 10^8 arguments.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define JP_LM __host__ __device__ __forceinline__
#else
#define JP_LM inline
#endif

namespace jp
{
namespace lm
{
JP_LM uint32_t asu(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
JP_LM float asf(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
JP_LM uint64_t asu64(double f) { union { double f; uint64_t u; } v; v.f = f; return v.u; }
JP_LM double asd(uint64_t u) { union { double f; uint64_t u; } v; v.u = u; return v.f; }
template <bool kFma> JP_LM double mad(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return kFma ? fma(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
#else
	if (kFma) return __builtin_fma(a, b, c);
	volatile double p = a * b; return p + c;                   // (volatile: no contraction whatever the host compiler's flags)
#endif
}
JP_LM float finf() { return asf(0x7f800000u); }
JP_LM float fnan() { return asf(0x7fc00000u); }

// __exp2f_data.tab (EXP2F_TABLE_BITS = 5): tab[i] = asuint64(2^(i/32)) - (i << 47)
JP_LM uint64_t exp2f_tab(int i)
{
	const uint64_t T[32] = {
		0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
		0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
		0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
		0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
	return T[i];
}

// ---- expf: e_expf.c ------------------------------------------------------------------------------------------------------------
template <bool kFma> JP_LM float expf_libm(float x)
{
	const double xd = (double)x;
	const uint32_t abstop = (asu(x) >> 20) & 0x7ffu;
	if (abstop >= (asu(88.0f) >> 20))
	{   // |x| >= 88 or x is nan
		if (asu(x) == 0xff800000u) return 0.0f;
		if (abstop >= (0x7f800000u >> 20)) return x + x;
		if (x > 0x1.62e42ep6f) return finf();                     // overflow
		if (x < -0x1.9fe368p6f) return 0.0f;                      // underflow
		if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;  // may underflow: the smallest subnormal
	}
	const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
	const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
	double z = InvLn2N * xd;
	double kd = z + Shift;
	const uint64_t ki = asu64(kd);
	kd -= Shift;
	const double r = z - kd;
	uint64_t t = exp2f_tab((int)(ki % 32));
	t += ki << (52 - 5);
	const double s = asd(t);
	z = mad<kFma>(C0, r, C1);
	const double r2 = r * r;
	double y = mad<kFma>(C2, r, 1.0);
	y = mad<kFma>(z, r2, y);
	y = y * s;
	return (float)y;
}

// ---- logf: e_logf.c ------------------------------------------------------------------------------------------------------------
JP_LM void logf_tab(int i, double* invc, double* logc)
{
	const double T[16][2] = {
		{ 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 }, { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 }, { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
		{ 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 }, { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
		{ 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1p+0, 0x0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 }, { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
		{ 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 }, { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 }, { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 } };
	*invc = T[i][0]; *logc = T[i][1];
}
template <bool kFma> JP_LM float logf_libm(float x)
{
	uint32_t ix = asu(x);
	if (ix == 0x3f800000u) return 0.0f;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u)
	{   // x < 0x1p-126 or inf or nan
		if (ix * 2 == 0) return -finf();
		if (ix == 0x7f800000u) return x;
		if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return fnan();
		ix = asu(x * 0x1p23f); ix -= 23u << 23;                    // subnormal: normalise
	}
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) % 16);
	const int k = (int32_t)tmp >> 23;
	const uint32_t iz = ix - (tmp & 0xff800000u);
	double invc, logc; logf_tab(i, &invc, &logc);
	const double z = (double)asf(iz);
	const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
	const double r = mad<kFma>(z, invc, -1.0);
	const double y0 = mad<kFma>((double)k, Ln2, logc);
	const double r2 = r * r;
	double y = mad<kFma>(A1, r, A2);
	y = mad<kFma>(A0, r2, y);
	y = mad<kFma>(y, r2, y0 + r);
	return (float)y;
}

// ---- powf: e_powf.c (POWF_SCALE_BITS = 0: the build without the round-to-int intrinsics) -------------------------------------------
JP_LM void powf_log2_tab(int i, double* invc, double* logc)
{
	const double T[16][2] = {
		{ 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 }, { 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
		{ 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 }, { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
		{ 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1p+0, 0x0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 },
		{ 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 }, { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 }, { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } };
	*invc = T[i][0]; *logc = T[i][1];
}
// 0: y is not an integer, 1: odd integer, 2: even integer (checkint)
JP_LM int powf_checkint(uint32_t iy)
{
	const int e = (int)(iy >> 23 & 0xff);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
JP_LM bool powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }
template <bool kFma> JP_LM float powf_libm(float x, float y)
{
	uint32_t sign_bias = 0;
	uint32_t ix = asu(x); const uint32_t iy = asu(y);
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy))
	{   // either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
		if (powf_zeroinfnan(iy))
		{
			if (2 * iy == 0) return 1.0f;                                            // (signalling NaNs: not produced by the callers)
			if (ix == 0x3f800000u) return 1.0f;
			if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
			if (2 * ix == 2u * 0x3f800000u) return 1.0f;
			if ((2 * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;     // |x| < 1 && y == inf or |x| > 1 && y == -inf
			return y * y;
		}
		if (powf_zeroinfnan(ix))
		{
			float x2 = x * x;
			if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
			return (iy & 0x80000000u) ? 1 / x2 : x2;
		}
		if (ix & 0x80000000u)
		{   // x < 0: finite
			const int yint = powf_checkint(iy);
			if (yint == 0) return fnan();
			if (yint == 1) sign_bias = 1u << (5 + 11);
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u)
		{   // subnormal x: normalise
			ix = asu(asf(ix) * 0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	// log2_inline
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) % 16);
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	double invc, logc; powf_log2_tab(i, &invc, &logc);
	const double z = (double)asf(iz);
	const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
	const double r = mad<kFma>(z, invc, -1.0);
	const double y0 = logc + (double)k;
	const double r2 = r * r;
	double yy = mad<kFma>(A0, r, A1);
	const double p = mad<kFma>(A2, r, A3);
	const double r4 = r2 * r2;
	double q = mad<kFma>(A4, r, y0);
	q = mad<kFma>(p, r2, q);
	yy = mad<kFma>(yy, r4, q);
	const double logx = yy;
	const double ylogx = (double)y * logx;                          // cannot overflow, y is single precision
	if ((asu64(ylogx) >> 47 & 0xffff) >= (asu64(126.0) >> 47))
	{   // |y * log(x)| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -finf() : finf();      // overflow
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                         // underflow
		if (ylogx < -149.0) { const float t = 0x1.4p-75f * 0x1.4p-75f; return sign_bias ? -t : t; }   // may underflow
	}
	// exp2_inline
	const double Shift = 0x1.8p+52 / 32;
	const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
	double kd = ylogx + Shift;
	const uint64_t ki = asu64(kd);
	kd -= Shift;
	const double rr = ylogx - kd;
	uint64_t t = exp2f_tab((int)(ki % 32));
	const uint64_t ski = ki + sign_bias;
	t += ski << (52 - 5);
	const double s = asd(t);
	double zz = mad<kFma>(C0, rr, C1);
	const double rr2 = rr * rr;
	double e = mad<kFma>(C2, rr, 1.0);
	e = mad<kFma>(zz, rr2, e);
	e = e * s;
	return (float)e;
}

// ---- fp32 helpers of the fdlibm routines: no contraction, correctly rounded divide / sqrt ---------------------------------------
JP_LM float fmul(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __fmul_rn(a, b);
#else
	volatile float r = a * b; return r;
#endif
}
JP_LM float fsqrt(float a)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return sqrtf(a);                                              // the correctly rounded form (hipcc default, jp_device.h)
#else
	return __builtin_sqrtf(a);
#endif
}
#define JP_M(a, b) fmul((a), (b))

// ---- acosf: e_acosf.c ------------------------------------------------------------------------------------------------------------
JP_LM float acosf_libm(float x)
{
	const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
		pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
		qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
	const int32_t hx = (int32_t)asu(x), ix = hx & 0x7fffffff;
	if (ix == 0x3f800000) { if (hx > 0) return 0.0f; return pi + JP_M(2.0f, pio2_lo); }
	if (ix > 0x3f800000) return fnan();
	if (ix < 0x3f000000)
	{   // |x| < 0.5
		if (ix <= 0x23000000) return pio2_hi + pio2_lo;
		const float z = JP_M(x, x);
		const float p = JP_M(z, pS0 + JP_M(z, pS1 + JP_M(z, pS2 + JP_M(z, pS3 + JP_M(z, pS4 + JP_M(z, pS5))))));
		const float q = one + JP_M(z, qS1 + JP_M(z, qS2 + JP_M(z, qS3 + JP_M(z, qS4))));
		const float r = p / q;
		return pio2_hi - (x - (pio2_lo - JP_M(x, r)));
	}
	if (hx < 0)
	{   // x < -0.5
		const float z = JP_M(one + x, 0.5f);
		const float p = JP_M(z, pS0 + JP_M(z, pS1 + JP_M(z, pS2 + JP_M(z, pS3 + JP_M(z, pS4 + JP_M(z, pS5))))));
		const float q = one + JP_M(z, qS1 + JP_M(z, qS2 + JP_M(z, qS3 + JP_M(z, qS4))));
		const float s = fsqrt(z);
		const float r = p / q;
		const float w = JP_M(r, s) - pio2_lo;
		return pi - JP_M(2.0f, s + w);
	}
	{   // x > 0.5
		const float z = JP_M(one - x, 0.5f);
		const float s = fsqrt(z);
		const float df = asf(asu(s) & 0xfffff000u);
		const float c = (z - JP_M(df, df)) / (s + df);
		const float p = JP_M(z, pS0 + JP_M(z, pS1 + JP_M(z, pS2 + JP_M(z, pS3 + JP_M(z, pS4 + JP_M(z, pS5))))));
		const float q = one + JP_M(z, qS1 + JP_M(z, qS2 + JP_M(z, qS3 + JP_M(z, qS4))));
		const float r = p / q;
		const float w = JP_M(r, s) + c;
		return JP_M(2.0f, df + w);
	}
}

// ---- atanf: s_atanf.c ------------------------------------------------------------------------------------------------------------
JP_LM float atanf_libm(float x)
{
	const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
	const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
	const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
		6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
	const float one = 1.0f;
	const int32_t hx = (int32_t)asu(x), ix = hx & 0x7fffffff;
	int id;
	if (ix >= 0x4c000000)
	{   // |x| >= 2^25
		if (ix > 0x7f800000) return x + x;
		if (hx > 0) return atanhi[3] + atanlo[3];
		return -atanhi[3] - atanlo[3];
	}
	if (ix < 0x3ee00000)
	{   // |x| < 0.4375
		if (ix < 0x31000000) return x;                             // |x| < 2^-29
		id = -1;
	}
	else
	{
		x = asf((uint32_t)ix);
		if (ix < 0x3f980000)
		{
			if (ix < 0x3f300000) { id = 0; x = (JP_M(2.0f, x) - one) / (2.0f + x); }
			else { id = 1; x = (x - one) / (x + one); }
		}
		else
		{
			if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + JP_M(1.5f, x)); }
			else { id = 3; x = -1.0f / x; }
		}
	}
	const float z = JP_M(x, x);
	const float w = JP_M(z, z);
	const float s1 = JP_M(z, aT[0] + JP_M(w, aT[2] + JP_M(w, aT[4] + JP_M(w, aT[6] + JP_M(w, aT[8] + JP_M(w, aT[10]))))));
	const float s2 = JP_M(w, aT[1] + JP_M(w, aT[3] + JP_M(w, aT[5] + JP_M(w, aT[7] + JP_M(w, aT[9])))));
	if (id < 0) return x - JP_M(x, s1 + s2);
	const float zz = atanhi[id] - ((JP_M(x, s1 + s2) - atanlo[id]) - x);
	return hx < 0 ? -zz : zz;
}

// ---- tanf: s_tanf.c + k_tanf.c + e_rem_pio2f.c (|x| < 120; beyond that the caller keeps the device library) -------------------------
JP_LM float kernel_tanf(float x, float y, int iy)
{
	const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
	const float T[13] = { 3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f, 1.4562094584e-03f,
		5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f };
	const int32_t hx = (int32_t)asu(x), ix = hx & 0x7fffffff;
	if (ix < 0x39000000)
	{   // |x| < 2^-13
		if ((int)x == 0)
		{
			if ((ix | (iy + 1)) == 0) return one / asf((uint32_t)ix);
			if (iy == 1) return x;
			return -one / x;
		}
	}
	if (ix >= 0x3f2ca140)
	{   // |x| >= 0.6744
		if (hx < 0) { x = -x; y = -y; }
		const float z = pio4 - x;
		const float w = pio4lo - y;
		x = z + w; y = 0.0f;
		if (asf(asu(x) & 0x7fffffffu) < 0x1p-13f) return JP_M((float)(1 - ((hx >> 30) & 2)) * (float)iy, 1.0f - JP_M(JP_M(2.0f, (float)iy), x));
	}
	float z = JP_M(x, x);
	float w = JP_M(z, z);
	float r = T[1] + JP_M(w, T[3] + JP_M(w, T[5] + JP_M(w, T[7] + JP_M(w, T[9] + JP_M(w, T[11])))));
	float v = JP_M(z, T[2] + JP_M(w, T[4] + JP_M(w, T[6] + JP_M(w, T[8] + JP_M(w, T[10] + JP_M(w, T[12]))))));
	float s = JP_M(z, x);
	r = y + JP_M(z, JP_M(s, r + v) + y);
	r += JP_M(T[0], s);
	w = x + r;
	if (ix >= 0x3f2ca140)
	{
		v = (float)iy;
		return JP_M((float)(1 - ((hx >> 30) & 2)), v - JP_M(2.0f, x - (JP_M(w, w) / (w + v) - r)));
	}
	if (iy == 1) return w;
	{   // -1 / (x + r), accurately
		z = asf(asu(w) & 0xfffff000u);
		v = r - (z - x);
		const float a = -1.0f / w;
		const float t = asf(asu(a) & 0xfffff000u);
		s = 1.0f + JP_M(t, z);
		return t + JP_M(a, s + JP_M(t, v));
	}
}
// __ieee754_rem_pio2f for |x| < 120 (glibc >= 2.28: e_rem_pio2f.c reduces in fp64 with reduce_fast of s_sincosf.h, the reduction of
// sinf / cosf): n = round(x * 2/pi) through a product prescaled by 2^24, y[0] + y[1] = x - n * pi/2 split into two floats
// (tanf is no IFUNC in libm: one build, no contraction -- checked against the running libm, tests/test_libm_exact.py)
JP_LM int rem_pio2f_fast(float x, float* y0, float* y1)
{
	const double dx = (double)x;
	const double r = dx * 0x1.45f306dc9c883p+23;
	const int n = ((int32_t)r + 0x800000) >> 24;
	const double red = mad<false>(-(double)n, 0x1.921fb54442d18p0, dx);
	*y0 = (float)red;
	*y1 = (float)(red - (double)*y0);
	return n;
}
// valid for |x| < 120; `ok` = false beyond (the Payne-Hanek branch is not transcribed: no call site of the reference gets there)
JP_LM float tanf_libm(float x, bool* ok)
{
	const int32_t ix = (int32_t)asu(x) & 0x7fffffff;
	*ok = true;
	if (ix <= 0x3f490fda) return kernel_tanf(x, 0.0f, 1);
	if ((asu(x) >> 20 & 0x7ff) >= (asu(120.0f) >> 20)) { *ok = false; return 0.0f; }
	float y0, y1;
	const int n = rem_pio2f_fast(x, &y0, &y1);
	return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}
#undef JP_M
} // namespace lm
} // namespace jp
