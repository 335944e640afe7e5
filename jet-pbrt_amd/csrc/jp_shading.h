// jet-pbrt_amd/csrc/jp_shading.h -- device restatement of the reference's appearance code: shading frame,
// BSDF closures (bsdf.h, bsdf.cc, microfacet.cc), material dispatch (material.h/.cc), light sampling (light.h,
// shape.h sampling) and the sampling warps (sampling.h).  Same operation order as the reference throughout;
// see jp_device.h for the numerics rules.
//
// Transcendentals: the reference calls glibc sinf/cosf (results within ~0.56 ulp, not bit-reproducible across
// hosts).  The device evaluates sin/cos in fp64 (sincos_d below) and rounds once to fp32 -- the correctly rounded
// value in all but ~1e-9 of cases, which coincides with glibc's result wherever glibc is correctly rounded.
#pragma once
#include "jp_device.h"
#define JP_HD_INLINE __host__ __device__ __forceinline__

namespace jp
{
// sin and cos of an fp32 argument, evaluated in fp64 and rounded once to fp32.  |x| stays below ~7 on every call
// site (angles in [-pi/4, 2 pi]), so one Cody-Waite step by pi/2 (two-part constant, exact product via fma) plus
// the Taylor polynomials to x^15 / x^16 on [-pi/4, pi/4] give an absolute error < 2e-16 -- the rounded fp32
// value is the correctly rounded sine/cosine except when the true value lies within ~1e-9 ulp of a rounding
// boundary.  ~25 fp64 operations for the pair, no table, no Payne-Hanek path.
__device__ __forceinline__ void sincos_d(float xf, double* s, double* c)
{
	const double x = (double)xf;
	const double kd = rint(x * 0.63661977236758134308);
	const int k = (int)kd;
	double r = fma(-kd, 1.57079632679489655800e+00, x);
	r = fma(-kd, 6.12323399573676603587e-17, r);
	const double r2 = r * r;
	double sp = fma(r2, -7.6471637318198164759e-13, 1.6059043836821614599e-10);      // -1/15!, 1/13!
	sp = fma(sp, r2, -2.5052108385441718775e-08);                                     // -1/11!
	sp = fma(sp, r2, 2.7557319223985890653e-06);                                      // 1/9!
	sp = fma(sp, r2, -1.9841269841269841270e-04);                                     // -1/7!
	sp = fma(sp, r2, 8.3333333333333333333e-03);                                      // 1/5!
	sp = fma(sp, r2, -1.6666666666666666667e-01);                                     // -1/3!
	sp = fma(sp * r2, r, r);
	double cp = fma(r2, 4.7794773323873852974e-14, -1.1470745597729724714e-11);      // 1/16!, -1/14!
	cp = fma(cp, r2, 2.0876756987868098979e-09);                                      // 1/12!
	cp = fma(cp, r2, -2.7557319223985890653e-07);                                     // -1/10!
	cp = fma(cp, r2, 2.4801587301587301587e-05);                                      // 1/8!
	cp = fma(cp, r2, -1.3888888888888888889e-03);                                     // -1/6!
	cp = fma(cp, r2, 4.1666666666666666667e-02);                                      // 1/4!
	cp = fma(cp, r2, -0.5);
	cp = fma(cp, r2, 1.0);
	const double ss = (k & 1) ? cp : sp, cc = (k & 1) ? sp : cp;
	*s = (k & 2) ? -ss : ss;
	*c = ((k + 1) & 2) ? -cc : cc;
}

// ---- sinf / cosf / sincosf exactly as the host's libm computes them ---------------------------------------------------
// glibc >= 2.28 evaluates the three functions with one algorithm (sysdeps/ieee754/flt-32/s_sincosf.h, from the Arm
// optimized routines): the argument is widened to fp64, reduced by n * pi/2 with n = ((int)(x * 2^24 * 2/pi) + 2^23) >> 24,
// and two fixed fp64 polynomials are evaluated; the fp64 results are rounded once to fp32.  Every step is IEEE fp64
// arithmetic, so the device can reproduce the host's fp32 results BIT FOR BIT -- which removes the last source of
// difference between the device film and the reference's (a 1-ulp different bounce direction is harmless in the Cornell
// box but flips whole paths on finely tessellated, flat-shaded meshes).  Two builds of that code exist in libm and an
// IFUNC picks one per CPU: compiled with FMA contraction (x86-64 with FMA + AVX2) or without.  kFma selects which one is
// reproduced; jp_create_context probes the host's sincosf and sets g_sincosf_mode (0: neither matches, keep sincos_d).
// Transcribed from the polynomial data flow of the compiled routines: sin = (s2 + x2 s3) * (x2 x3) + (xs + x3 s1),
// cos = (c3 + x2 c4) * (x2 x4) + ((c0 + x2 c1) + x4 c2), x3 = x2 xs, x4 = x2 x2, each "a * b + c" one fma when kFma.
__device__ __constant__ int g_sincosf_mode = 0;

template <bool kFma>
JP_HD_INLINE void sincosf_libm(float y, float* sp, float* cp)
{
	union { float f; unsigned int u; } bits; bits.f = y;
	const unsigned int top = (bits.u >> 20) & 0x7ffu;                      // abstop12
	const double x = (double)y;
	double xs, x2; int n = 0;
	if (top <= 0x3f3u)                                                     // |y| < pi/4
	{
		if (top <= 0x397u) { *sp = y; *cp = 1.0f; return; }                // |y| < 2^-12
		xs = x; x2 = x * x;
	}
	else
	{   // pi/4 <= |y| < 120 on every call site (angles in [0, 2 pi]); reduce_fast
		const double r = x * 0x1.45f306dc9c883p+23;
		n = ((int)r + 0x800000) >> 24;
		const double xr = kFma ? fma(-(double)n, 0x1.921fb54442d18p+0, x) : x - (double)n * 0x1.921fb54442d18p+0;
		const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;    // sign of sine in quadrants 0..3
		xs = xr * sgn; x2 = xr * xr;
	}
	const double t = (n & 2) ? -1.0 : 1.0;                                 // second table: cosine coefficients negated
	const double c0 = t, c1 = t * -0x1.ffffffd0c621cp-2, c2 = t * 0x1.55553e1068f19p-5, c3 = t * -0x1.6c087e89a359dp-10, c4 = t * 0x1.99343027bf8c3p-16;
	const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	const double x3 = x2 * xs, x4 = x2 * x2;
	const double x5 = x2 * x3, x6 = x2 * x4;
	double c1v, as, ac, bs, bc, S, C;
	if (kFma)
	{
		c1v = fma(x2, c1, c0); as = fma(x2, s3, s2); ac = fma(x2, c4, c3);
		bs = fma(x3, s1, xs); bc = fma(x4, c2, c1v);
		S = fma(as, x5, bs); C = fma(ac, x6, bc);
	}
	else
	{
		c1v = c0 + x2 * c1; as = s2 + x2 * s3; ac = c3 + x2 * c4;
		bs = xs + x3 * s1; bc = c1v + x4 * c2;
		S = bs + x5 * as; C = bc + x6 * ac;
	}
	const float fs = (float)S, fc = (float)C;
	if (n & 1) { *sp = fc; *cp = fs; } else { *sp = fs; *cp = fc; }
}

__device__ __forceinline__ void sincos_f(float xf, float* s, float* c)
{
	const int mode = g_sincosf_mode;                                       // uniform: one scalar load
	if (mode == 1 && fabsf(xf) < 120.f) sincosf_libm<true>(xf, s, c);
	else if (mode == 2 && fabsf(xf) < 120.f) sincosf_libm<false>(xf, s, c);
	else { double sd, cd; sincos_d(xf, &sd, &cd); *s = (float)sd; *c = (float)cd; }
}

#define JP_2PI      (2.0f * JP_PI)
#define JP_PI_OVER2 (JP_PI / 2.0f)
#define JP_PI_OVER4 (JP_PI / 4.0f)
#define JP_INV_PI   (1.0f / JP_PI)

enum { BS_REFLECTION = 1, BS_TRANSMISSION = 2, BS_SPECULAR = 4, BS_DIFFUSE = 8, BS_GLOSSY = 16 };   // bsdf.h:208-219
enum { CL_LAMBERT = 0, CL_MIRROR = 1, CL_FRESNEL_SPECULAR = 2, CL_MICROFACET = 3 };
enum { FR_CONDUCTOR = 0, FR_DIELECTRIC = 1 };

struct Frame { V3 s, t, n; };
__device__ __forceinline__ Frame frame_from_z(V3 nn)                                    // geometry.h:345-349, 371-376
{
	Frame f; f.n = normalize(nn);
	V3 tmp = (fabsf(f.n.x) > 0.99f) ? mk(0, 1, 0) : mk(1, 0, 0);
	f.t = normalize(cross(f.n, tmp));
	f.s = normalize(cross(f.t, f.n));
	return f;
}
__device__ __forceinline__ V3 to_local(const Frame& f, V3 w) { return mk(dot(f.s, w), dot(f.t, w), dot(f.n, w)); }
__device__ __forceinline__ V3 to_world(const Frame& f, V3 l) { return f.s * l.x + f.t * l.y + f.n * l.z; }

struct Closure
{
	int kind;
	V3 c0, c1;              // lambert albedo / mirror reflectance / glass Kr,Kt / microfacet R
	float eta_t;            // glass (eta_i = 1)
	float ax, ay;           // TrowbridgeReitz alphas
	int fresnel; V3 feta, fk;
	float lam_o;            // microfacet: Lambda(wo), the same value in every Evalf / Pdf of one shading event (closure_set_wo)
};
__device__ __forceinline__ bool is_delta(const Closure& c) { return c.kind == CL_MIRROR || c.kind == CL_FRESNEL_SPECULAR; }

struct BsdfSample { V3 f, wi; float pdf; int flags; };

// ---- sampling warps (sampling.h) ---------------------------------------------------------------------------------
__device__ __forceinline__ void concentric_disk(float ux, float uy, float& px, float& py)   // sampling.h:25-50
{
	ux = ux * 2.f - 1.f; uy = uy * 2.f - 1.f;
	if (ux == 0 && uy == 0) { px = 0; py = 0; }
	else
	{
		float radius, theta;
		if (fabsf(ux) > fabsf(uy)) { radius = ux; theta = JP_PI_OVER4 * (uy / ux); }
		else { radius = uy; theta = JP_PI_OVER2 - JP_PI_OVER4 * (ux / uy); }
		float st, ct; sincos_f(theta, &st, &ct);
		px = ct * radius; py = st * radius;
	}
}
__device__ __forceinline__ V3 cosine_hemisphere(float ux, float uy)                     // sampling.h:53-59
{
	float px, py; concentric_disk(ux, uy, px, py);
	float z = sqrtf(smax(0.f, 1 - px * px - py * py));
	return mk(px, py, z);
}
__device__ __forceinline__ V3 uniform_sphere(float ux, float uy)                        // sampling.h:80-87
{
	float z = 1 - 2 * ux;
	float radius = sqrtf(smax(0.f, 1.f - z * z));
	float phi = 2 * JP_PI * uy;
	float sp, cp; sincos_f(phi, &sp, &cp);
	return mk(radius * cp, radius * sp, z);
}

// ---- local-frame helpers (bsdf.h:17-60) ------------------------------------------------------------------------------
__device__ __forceinline__ bool same_hemi(V3 a, V3 b) { return a.z * b.z > 0; }
__device__ __forceinline__ float sin2t(V3 w) { return smax(0.f, 1.f - w.z * w.z); }
__device__ __forceinline__ float sint(V3 w) { return sqrtf(sin2t(w)); }
__device__ __forceinline__ float tant(V3 w) { return sint(w) / w.z; }
__device__ __forceinline__ float tan2t(V3 w) { return sin2t(w) / (w.z * w.z); }
__device__ __forceinline__ float cosphi(V3 w) { float s = sint(w); return (s == 0) ? 1 : clampf(w.x / s, -1.f, 1.f); }
__device__ __forceinline__ float sinphi(V3 w) { float s = sint(w); return (s == 0) ? 0 : clampf(w.y / s, -1.f, 1.f); }

__device__ __forceinline__ float fresnel_dielectric(float cos_i, float eta_i, float eta_t)     // bsdf.h:91-122
{
	cos_i = clampf(cos_i, -1.f, 1.f);
	bool entering = cos_i > 0.f;
	if (!entering) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_i = fabsf(cos_i); }
	float sin_i = sqrtf(smax(0.f, 1 - cos_i * cos_i));
	float sin_t = eta_i / eta_t * sin_i;
	if (sin_t >= 1) return 1;
	float cos_t = sqrtf(smax(0.f, 1 - sin_t * sin_t));
	float r_para = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
	float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
	return (r_para * r_para + r_perp * r_perp) / 2;
}

__device__ __forceinline__ V3 fresnel_conductor(float cosI, V3 etai, V3 etat, V3 k)           // bsdf.h:174-197
{
	cosI = clampf(cosI, -1.f, 1.f);
	V3 eta = cdiv(etat, etai), etak = cdiv(k, etai);
	float cos2 = cosI * cosI, sin2 = 1 - cos2;
	V3 eta2 = cmul(eta, eta), etak2 = cmul(etak, etak);
	V3 t0 = eta2 - etak2 - splat(sin2);
	V3 a2plusb2 = csqrt(cmul(t0, t0) + cmul(eta2 * 4.f, etak2));
	V3 t1 = a2plusb2 + splat(cos2);
	V3 a = csqrt((a2plusb2 + t0) * 0.5f);
	V3 t2 = a * (2.f * cosI);
	V3 Rs = cdiv(t1 - t2, t1 + t2);
	V3 t3 = a2plusb2 * cos2 + splat(sin2 * sin2);
	V3 t4 = t2 * sin2;
	V3 Rp = cdiv(cmul(Rs, t3 - t4), t3 + t4);
	return (Rp + Rs) * 0.5f;
}

__device__ __forceinline__ V3 fresnel_eval(const Closure& c, float cosI)                      // bsdf.cc:15-24
{
	if (c.fresnel == FR_CONDUCTOR) return fresnel_conductor(fabsf(cosI), splat(1.f), c.feta, c.fk);
	return splat(fresnel_dielectric(cosI, 1.5f, 1.f));                                        // material.cc:21
}

// TrowbridgeReitzDistribution: microfacet.cc:181-189 (D), 202-210 (Lambda), microfacet.h:22-30 (G1, G), microfacet.cc:359-365 (Pdf)
__device__ __forceinline__ float tr_D(const Closure& c, V3 wh)
{
	float t2 = tan2t(wh);
	if (isinf(t2)) return 0.f;
	const float cos4 = (wh.z * wh.z) * (wh.z * wh.z);
	float cp = cosphi(wh), sp = sinphi(wh);
	float e = (cp * cp / (c.ax * c.ax) + sp * sp / (c.ay * c.ay)) * t2;
	return 1 / (JP_PI * c.ax * c.ay * cos4 * (1 + e) * (1 + e));
}
__device__ __forceinline__ float tr_Lambda(const Closure& c, V3 w)
{
	float absTan = fabsf(tant(w));
	if (isinf(absTan)) return 0.f;
	float cp = cosphi(w), sp = sinphi(w);
	float alpha = sqrtf(cp * cp * c.ax * c.ax + sp * sp * c.ay * c.ay);
	float a2t2 = (alpha * absTan) * (alpha * absTan);
	return (-1 + sqrtf(1.f + a2t2)) / 2;
}
// the reference evaluates Lambda(wo) anew in every G / G1 of a shading event (two light samples, the BSDF sample, its pdf);
// it is a pure function of (closure, wo), so it is computed once per event and reused: same value, ~125 instructions less per use
__device__ __forceinline__ void closure_set_wo(Closure& c, V3 wo) { c.lam_o = c.kind == CL_MICROFACET ? tr_Lambda(c, wo) : 0.f; }
__device__ __forceinline__ float tr_G1o(const Closure& c) { return 1 / (1 + c.lam_o); }
__device__ __forceinline__ float tr_G(const Closure& c, V3 wi) { return 1 / (1 + c.lam_o + tr_Lambda(c, wi)); }
__device__ __forceinline__ float tr_Pdf(const Closure& c, V3 wo, V3 wh) { return tr_D(c, wh) * tr_G1o(c) * absdot(wo, wh) / fabsf(wo.z); }

// microfacet.cc:256-301; the double-precision spots of the reference (`> .9999`, unqualified sqrt/cos/sin, `> 1e10`) kept
__device__ __forceinline__ void tr_sample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y)
{
	if ((double)cosTheta > .9999)
	{
		float r = sqrtf(U1 / (1 - U1));                     // double sqrt of an fp32 value, rounded back: same value
		float phi = 6.28318530718f * U2;
		double sd, cd; sincos_d(phi, &sd, &cd);
		*slope_x = (float)((double)r * cd);
		*slope_y = (float)((double)r * sd);
		return;
	}
	float sinTheta = sqrtf(smax(0.f, 1.f - cosTheta * cosTheta));
	float tanTheta = sinTheta / cosTheta;
	float a = 1 / tanTheta;
	float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
	float A = 2 * U1 / G1 - 1;
	float tmp = 1.f / (A * A - 1.f);
	if ((double)tmp > 1e10) tmp = (float)1e10;
	float B = tanTheta;
	float D = sqrtf(smax(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
	float sx1 = B * tmp - D, sx2 = B * tmp + D;
	*slope_x = (A < 0 || sx2 > 1.f / tanTheta) ? sx1 : sx2;
	float S;
	if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
	else { S = -1.f; U2 = 2.f * (.5f - U2); }
	float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
	*slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
__device__ __forceinline__ V3 tr_sample_wh(const Closure& c, V3 wo, float u0, float u1)       // microfacet.cc:303-357
{
	bool flip = wo.z < 0;
	V3 wi = flip ? -wo : wo;
	V3 ws = normalize(mk(c.ax * wi.x, c.ay * wi.y, wi.z));
	float sx, sy; tr_sample11(ws.z, u0, u1, &sx, &sy);
	float cp = cosphi(ws), sp = sinphi(ws);
	float tmp = cp * sx - sp * sy;
	sy = sp * sx + cp * sy;
	sx = tmp;
	sx = c.ax * sx; sy = c.ay * sy;
	V3 wh = normalize(mk(-sx, -sy, 1.f));
	if (flip) wh = -wh;
	return wh;
}

__device__ __forceinline__ V3 eval_local(const Closure& c, V3 wo, V3 wi)
{
	if (c.kind == CL_LAMBERT)                                                                 // bsdf.h:347-355
	{
		if (!same_hemi(wo, wi)) return splat(0);
		return c.c0 * JP_INV_PI;
	}
	if (c.kind == CL_MICROFACET)                                                              // bsdf.cc:35-51
	{
		float cosO = fabsf(wo.z), cosI = fabsf(wi.z);
		V3 wh = wi + wo;
		if (cosI == 0 || cosO == 0) return splat(0);
		if (wh.x == 0 && wh.y == 0 && wh.z == 0) return splat(0);
		wh = normalize(wh);
		V3 ff = (dot(wh, mk(0, 0, 1)) < 0) ? -wh : wh;
		V3 F = fresnel_eval(c, dot(wi, ff));
		return cmul(c.c0 * tr_D(c, wh) * tr_G(c, wi), F) / (4 * cosI * cosO);
	}
	return splat(0);                                                                          // delta BSDFs
}

__device__ __forceinline__ BsdfSample sample_local(const Closure& c, V3 wo, float ux, float uy)
{
	BsdfSample s; s.f = splat(0); s.wi = mk(0, 0, 1); s.pdf = 0; s.flags = 0;                // bsdf.h:252-265
	if (c.kind == CL_LAMBERT)                                                                 // bsdf.h:362-377
	{
		s.wi = cosine_hemisphere(ux, uy);
		if (wo.z < 0) s.wi.z *= -1;
		s.f = eval_local(c, wo, s.wi);
		s.pdf = same_hemi(wo, s.wi) ? fabsf(s.wi.z) * JP_INV_PI : 0;
		s.flags = BS_REFLECTION | BS_DIFFUSE;
	}
	else if (c.kind == CL_MIRROR)                                                             // bsdf.h:415-429
	{
		s.wi = mk(-wo.x, -wo.y, wo.z);
		s.f = c.c0 / fabsf(s.wi.z);
		s.pdf = 1;
		s.flags = BS_REFLECTION | BS_SPECULAR;
	}
	else if (c.kind == CL_FRESNEL_SPECULAR)                                                   // bsdf.h:478-539
	{
		if (wo.z == 0.f) return s;
		float F = fresnel_dielectric(wo.z, 1.f, c.eta_t);
		if (ux < F)
		{
			s.wi = mk(-wo.x, -wo.y, wo.z);
			s.pdf = F;
			s.f = (c.c0 * F) / fabsf(s.wi.z);
			s.flags = BS_REFLECTION | BS_SPECULAR;
		}
		else
		{
			bool entering = wo.z > 0;
			V3 n = entering ? mk(0, 0, 1) : mk(-0.f, -0.f, -1);
			float etaI = entering ? 1.f : c.eta_t;
			float etaT = entering ? c.eta_t : 1.f;
			float eta = etaI / etaT;
			// refract bsdf.h:70-88
			float cos_i = dot(n, wo);
			float sin2_i = smax(0.f, 1 - cos_i * cos_i);
			float sin2_t = eta * eta * sin2_i;
			if (sin2_t >= 1) { s.f = splat(0); }
			else
			{
				float cos_t = sqrtf(1 - sin2_t);
				s.wi = eta * -wo + (eta * cos_i - cos_t) * n;
				V3 ft = c.c1 * (1 - F);
				ft = ft * ((etaI * etaI) / (etaT * etaT));
				s.pdf = 1 - F;
				s.f = ft / fabsf(s.wi.z);
				s.flags = BS_TRANSMISSION | BS_SPECULAR;
			}
		}
	}
	else                                                                                      // bsdf.cc:60-78
	{
		if (wo.z == 0) return s;
		V3 wh = tr_sample_wh(c, wo, ux, uy);
		float owh = dot(wo, wh);
		if (owh < 0) return s;
		V3 wi = -wo + 2 * owh * wh;                                                           // reflect bsdf.h:62-67
		if (!same_hemi(wo, wi)) return s;
		s.wi = wi;
		s.f = eval_local(c, wo, wi);
		s.pdf = tr_Pdf(c, wo, wh) / (4 * dot(wo, wh));
		s.flags = BS_REFLECTION | BS_GLOSSY;
	}
	return s;
}

// FMaterial::Scattering (material.h:34-37, 52-55, 72-75; material.cc:12-43).  `uplastic` is the draw
// FPlasticMaterial consumes (material.cc:14); the caller draws it only for JP_MAT_PLASTIC.
template <typename MatPtr>
__device__ __forceinline__ void make_closure(MatPtr mats, int type, int mat, float uplastic, Closure& c)
{
	const float4 p0 = mats[4 * mat + 0], p1 = mats[4 * mat + 1];
	c.c0 = splat(0); c.c1 = splat(0); c.eta_t = 1; c.ax = c.ay = 0; c.fresnel = FR_CONDUCTOR; c.feta = splat(0); c.fk = splat(0);
	if (type == JP_MAT_MATTE) { c.kind = CL_LAMBERT; c.c0 = xyz(p0); }
	else if (type == JP_MAT_MIRROR) { c.kind = CL_MIRROR; c.c0 = xyz(p0); }
	else if (type == JP_MAT_GLASS) { c.kind = CL_FRESNEL_SPECULAR; c.eta_t = p0.x; c.c0 = mk(p0.y, p0.z, p0.w); c.c1 = mk(p1.x, p1.y, p1.z); }
	else if (type == JP_MAT_PLASTIC)
	{
		float Qd = p1.w;
		if (uplastic < Qd) { c.kind = CL_LAMBERT; c.c0 = xyz(p0) / Qd; }
		else { c.kind = CL_MICROFACET; c.c0 = mk(p0.w, p1.x, p1.y) / (1 - Qd); c.fresnel = FR_DIELECTRIC; c.ax = c.ay = smax(0.001f, p1.z); }
	}
	else { c.kind = CL_MICROFACET; c.c0 = splat(1.f); c.fresnel = FR_CONDUCTOR; c.feta = xyz(p0); c.fk = mk(p0.w, p1.x, p1.y); c.ax = smax(0.001f, p1.z); c.ay = smax(0.001f, p1.w); }
}

// ---- lights ---------------------------------------------------------------------------------------------------------
struct LightSample { V3 pos, wi; float pdf; V3 Li; float dist; };   // dist: |pos - p| where the sampler has it anyway (flat area lights), else < 0

// FLight::Sample_Li for light `li` from surface point p with normal n (isect.normal, used by the sphere's
// inside branch only).  light.h:199-216 (area), :265-291 (environment); shape sampling shape.h:124-145, 353-363,
// 459-467, 549-644.
template <typename PrimPtr, typename LightPtr>
__device__ __forceinline__ LightSample sample_li(const SceneView& sc, PrimPtr prims, LightPtr lights, int li, V3 p, V3 n_isect, float ux, float uy)
{
	LightSample s; s.pos = mk(0, 0, 0); s.wi = mk(0, 0, 0); s.pdf = 0; s.Li = splat(0); s.dist = -1.f;
	const float4 l0 = lights[2 * li], l1 = lights[2 * li + 1];
	const V3 radiance = xyz(l0);
	if (__float_as_int(l0.w) == JP_LIGHT_ENVIRONMENT)
	{
		float theta = uy * JP_PI, phi = ux * 2 * JP_PI;
		float cosT, sinT, sinP, cosP; sincos_f(theta, &sinT, &cosT); sincos_f(phi, &sinP, &cosP);
		s.wi = mk(sinT * cosP, sinT * sinP, cosT);
		s.pos = p + s.wi * 2 * sc.world_radius;
		if (sinT != 0) s.pdf = 1 / (2 * JP_PI * JP_PI * sinT);
		s.Li = radiance;
		return s;
	}
	if (__float_as_int(l0.w) == JP_LIGHT_POINT)                   // FPointLight::Sample_Li light.h:94-123; l1.xyz = worldPosition
	{
		const V3 wp = xyz(l1);
		s.pos = wp;
		s.wi = normalize(wp - p);
		s.pdf = 1.f;
		s.Li = radiance / len2(wp - p);
		return s;
	}
	if (__float_as_int(l0.w) == JP_LIGHT_DIRECTION)               // FDirectionLight::Sample_Li light.h:155-164; l1.xyz = worldDir
	{
		s.wi = -xyz(l1);
		s.pos = p + s.wi * 2 * sc.world_radius;
		s.pdf = 1.f;
		s.Li = radiance;
		return s;
	}
	const int pi = __float_as_int(l1.x);
	const float inv_area = l1.y;                                  // 1 / FShape::Area(), precomputed at upload with the reference's expression
	const float4 g0 = prims[4 * pi + 0], g3 = prims[4 * pi + 3];
	const int type = __float_as_int(g3.w);
	V3 lp, ln; float pdf;
	if (type != JP_SHAPE_SPHERE)
	{
		const float4 g1 = prims[4 * pi + 1], g2 = prims[4 * pi + 2];
		if (type == JP_SHAPE_TRIANGLE)                            // shape.h:353-363 + sampling.h:121-125
		{
			float su0 = sqrtf(ux); float bx = 1 - su0, by = uy * su0;
			lp = bx * xyz(g0) + by * xyz(g1) + (1 - bx - by) * xyz(g2);
		}
		else if (type == JP_SHAPE_DISK)                           // FDisk::SamplePosition shape.h:256-268: g0 = (position, radius), g1 = normal
		{
			const Frame fr = frame_from_z(xyz(g1));
			float px, py; concentric_disk(ux, uy, px, py);
			lp = xyz(g0) + g0.w * (fr.s * px + fr.t * py);
		}
		else lp = xyz(g1) + (xyz(g0) - xyz(g1)) * ux + (xyz(g2) - xyz(g1)) * uy;   // shape.h:459-467
		ln = type == JP_SHAPE_DISK ? xyz(g1) : xyz(g3);
		pdf = inv_area;
		V3 wi = lp - p;                                           // FShape::SampleDirection shape.h:124-145
		float dist2 = len2(wi);
		if (dist2 == 0) pdf = 0;
		else
		{
			const float l = sqrtf(dist2);                         // Normalize(wi) = wi / sqrt(|wi|^2), with the length kept: FScene::Occluded's
			wi = wi / l;                                          // |position - target| squares the negated differences -- the same bits
			s.dist = l;
			pdf *= dist2 / absdot(ln, -wi);
			if (isinf(pdf)) pdf = 0;
		}
	}
	else
	{
		const V3 c = xyz(g0); const float r = g0.w;
		if (len2(p - c) <= r * r)                                 // shape.h:567-586
		{
			V3 dir = uniform_sphere(ux, uy);
			lp = c + r * dir; ln = normalize(dir);
			pdf = inv_area;
			V3 wi = lp - p;
			if (len2(wi) == 0) pdf = 0;
			else { wi = normalize(wi); pdf *= len2(lp - p) / absdot(n_isect, -wi); }
			if (isinf(pdf)) pdf = 0;
		}
		else                                                      // shape.h:603-643 cone sampling
		{
			float dist = len(p - c);
			float inv_dist = 1 / dist;
			float sin_max = r * inv_dist;
			float sin_max2 = sin_max * sin_max;
			float inv_sin_max = 1 / sin_max;
			float cos_max = sqrtf(smax(0.f, 1 - sin_max2));
			float cos_theta = (cos_max - 1) * ux + 1;
			float sin_theta2 = 1 - cos_theta * cos_theta;
			if (sin_max2 < 0.00068523f) { sin_theta2 = sin_max2 * ux; cos_theta = sqrtf(1 - sin_theta2); }
			float cos_alpha = sin_theta2 * inv_sin_max + cos_theta * sqrtf(smax(0.f, 1.f - sin_theta2 * inv_sin_max * inv_sin_max));
			float sin_alpha = sqrtf(smax(0.f, 1.f - cos_alpha * cos_alpha));
			float phi = uy * 2 * JP_PI;
			Frame fr = frame_from_z((c - p) * inv_dist);
			float sphi, cphi; sincos_f(phi, &sphi, &cphi);
			V3 wn = sin_alpha * cphi * (-fr.s) + sin_alpha * sphi * (-fr.t) + cos_alpha * (-fr.n);
			lp = c + r * wn; ln = wn;
			pdf = 1 / (2 * JP_PI * (1 - cos_max));
		}
	}
	s.pos = lp; s.pdf = pdf;
	if (pdf == 0 || len2(lp - p) == 0) s.Li = splat(0);           // light.h:205-214
	else
	{
		s.wi = normalize(lp - p);
		s.Li = (dot(ln, -s.wi) > 0.f) ? radiance : splat(0);      // FAreaLight::L light.h:234-238
	}
	return s;
}

} // namespace jp
