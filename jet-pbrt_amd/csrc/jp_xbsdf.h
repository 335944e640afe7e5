// jet-pbrt_amd/csrc/jp_xbsdf.h -- the rest of the reference's reflection API on the device, by value (JpBsdfDesc, include/jetpbrt_amd.h):
// FPhongSpecularReflection bsdf.h:557-633, BeckmannDistribution microfacet.cc:11-254 (visible-area and full-distribution sampling),
// TrowbridgeReitzDistribution's full-distribution branch microfacet.cc:326-350, FMicrofacetTransmission bsdf.cc:80-145 (including the
// world-space FBSDF::Pdf call on local vectors at bsdf.cc:141), FresnelNoOp bsdf.h:664-667 and general FresnelConductor /
// FresnelDielectric parameters.  No material of material.h instantiates these, so they live outside the render kernels (k_shade's
// closures are unchanged) and are reached through jp_bsdf / k_bsdf.  Same operation order as the reference.  Round 3: the
// transcendental functions the reference takes from libm (logf, expf, powf, acosf, atanf, tanf) are glibc's own algorithms in the same
// IEEE arithmetic (jp_libm.h), selected when jp_create_context finds that they reproduce the host's libm (g_libm_mode): these classes
// are then bit-exact against the reference's KATs (tests/golden/kat_bsdf.npz) like the closures the materials build; with another libm
// the device library's functions stay and the stated tolerance applies (tests/test_gpu_parity.py).
#pragma once
#include "jp_shading.h"
#include "jp_libm.h"

namespace jp
{
__device__ __forceinline__ V3 reflect(V3 wo, V3 n) { return -wo + 2 * dot(wo, n) * n; }       // bsdf.h:62-67
__device__ __forceinline__ bool refract(V3 wi, V3 n, float eta, V3* wt)                        // bsdf.h:70-88
{
	float cos_i = dot(n, wi);
	float sin2_i = smax(0.f, (float)(1 - cos_i * cos_i));
	float sin2_t = eta * eta * sin2_i;
	if (sin2_t >= 1) return false;
	float cos_t = sqrtf(1 - sin2_t);
	*wt = eta * -wi + (eta * cos_i - cos_t) * n;
	return true;
}
}
namespace jp
{
namespace xb
{
// bit 0: jp_libm.h reproduces the host's logf / expf / powf / acosf / atanf / tanf; bit 1: the host runs libm's FMA build of the first three
__device__ __constant__ int g_libm_mode = 0;
__device__ __forceinline__ float m_logf(float x) { const int m = g_libm_mode; if (m & 1) return (m & 2) ? lm::logf_libm<true>(x) : lm::logf_libm<false>(x); return logf(x); }
__device__ __forceinline__ float m_expf(float x) { const int m = g_libm_mode; if (m & 1) return (m & 2) ? lm::expf_libm<true>(x) : lm::expf_libm<false>(x); return expf(x); }
__device__ __forceinline__ float m_powf(float x, float y) { const int m = g_libm_mode; if (m & 1) return (m & 2) ? lm::powf_libm<true>(x, y) : lm::powf_libm<false>(x, y); return powf(x, y); }
__device__ __forceinline__ float m_acosf(float x) { return (g_libm_mode & 1) ? lm::acosf_libm(x) : acosf(x); }
__device__ __forceinline__ float m_atanf(float x) { return (g_libm_mode & 1) ? lm::atanf_libm(x) : atanf(x); }
__device__ __forceinline__ float m_tanf(float x) { if (g_libm_mode & 1) { bool ok; const float t = lm::tanf_libm(x, &ok); if (ok) return t; } return tanf(x); }
#define JP_XB_INV2PI (1.0f / (2.0f * JP_PI))                       // pbrt.h:40,45
struct Dist { int kind; float ax, ay; bool vis; };
__device__ __forceinline__ float cos2phi(V3 w) { return cosphi(w) * cosphi(w); }                 // bsdf.h:50-52
__device__ __forceinline__ float sin2phi(V3 w) { return sinphi(w) * sinphi(w); }
__device__ __forceinline__ Dist make_dist(const JpBsdfDesc& d) { Dist r; r.kind = d.distribution; r.ax = smax(0.001f, d.alpha_x); r.ay = smax(0.001f, d.alpha_y); r.vis = d.sample_visible != 0; return r; }   // microfacet.h:66-69, 82-85

__device__ __forceinline__ float ErfInv(float x)                                                 // microfacet.cc:11-41
{
	float w, p;
	x = clampf(x, -.99999f, .99999f);
	w = -m_logf((1 - x) * (1 + x));
	if (w < 5)
	{
		w = w - 2.5f;
		p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w;
		p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
	}
	else
	{
		w = sqrtf(w) - 3;
		p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w; p = 0.00573950773f + p * w;
		p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
	}
	return p * x;
}
__device__ __forceinline__ float Erf(float x)                                                    // microfacet.cc:43-64
{
	float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	int sign = 1;
	if (x < 0) sign = -1;
	x = fabsf(x);
	float t = 1 / (1 + p * x);
	float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * m_expf(-x * x);
	return sign * y;
}
__device__ __forceinline__ void BeckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y)   // microfacet.cc:67-144
{
	if (cosThetaI > .9999f)
	{
		float r = sqrtf(-m_logf(1.0f - U1));
		float sinPhi, cosPhi; sincos_f(2 * JP_PI * U2, &sinPhi, &cosPhi);
		*slope_x = r * cosPhi; *slope_y = r * sinPhi;
		return;
	}
	float sinThetaI = sqrtf(smax((float)0, (float)1 - cosThetaI * cosThetaI));
	float tanThetaI = sinThetaI / cosThetaI;
	float cotThetaI = 1 / tanThetaI;
	float a = -1, c = Erf(cotThetaI);
	float u = smax(U1, (float)1e-6f);
	float thetaI = m_acosf(cosThetaI);
	float poly = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
	float b = c - (1 + c) * m_powf(1 - u, poly);
	const float rsp = 1.f / sqrtf(JP_PI);
	float nrm = 1 / (1 + c + rsp * tanThetaI * m_expf(-cotThetaI * cotThetaI));
	int it = 0;
	while (++it < 10)
	{
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		float ei = ErfInv(b);
		float value = nrm * (1 + b + rsp * tanThetaI * m_expf(-ei * ei)) - u;
		float derivative = nrm * (1 - ei * tanThetaI);
		if (fabsf(value) < 1e-5f) break;
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	*slope_x = ErfInv(b);
	*slope_y = ErfInv(2.0f * smax(U2, (float)1e-6f) - 1.0f);
}
__device__ __forceinline__ V3 stretch_sample(const Dist& D, V3 wi, float U1, float U2)           // BeckmannSample microfacet.cc:146-170 / TrowbridgeReitzSample :303-324
{
	V3 ws = normalize(mk(D.ax * wi.x, D.ay * wi.y, wi.z));
	float sx, sy;
	if (D.kind == JP_DIST_BECKMANN) BeckmannSample11(ws.z, U1, U2, &sx, &sy); else tr_sample11(ws.z, U1, U2, &sx, &sy);
	float tmp = cosphi(ws) * sx - sinphi(ws) * sy;
	sy = sinphi(ws) * sx + cosphi(ws) * sy;
	sx = tmp;
	sx = D.ax * sx; sy = D.ay * sy;
	return normalize(mk(-sx, -sy, 1.f));
}
__device__ __forceinline__ float dist_D(const Dist& D, V3 wh)                                    // microfacet.cc:175-192
{
	float tan2Theta = tan2t(wh);
	if (isinf(tan2Theta)) return 0.;
	const float cos4Theta = (wh.z * wh.z) * (wh.z * wh.z);
	if (D.kind == JP_DIST_BECKMANN)
		return m_expf(-tan2Theta * (cos2phi(wh) / (D.ax * D.ax) + sin2phi(wh) / (D.ay * D.ay))) / (JP_PI * D.ax * D.ay * cos4Theta);
	float e = (cos2phi(wh) / (D.ax * D.ax) + sin2phi(wh) / (D.ay * D.ay)) * tan2Theta;
	return 1 / (JP_PI * D.ax * D.ay * cos4Theta * (1 + e) * (1 + e));
}
__device__ __forceinline__ float dist_Lambda(const Dist& D, V3 w)                                // microfacet.cc:194-214
{
	float absTanTheta = fabsf(tant(w));
	if (isinf(absTanTheta)) return 0.;
	float alpha = sqrtf(cos2phi(w) * D.ax * D.ax + sin2phi(w) * D.ay * D.ay);
	if (D.kind == JP_DIST_BECKMANN)
	{
		float a = 1 / (alpha * absTanTheta);
		if (a >= 1.6f) return 0;
		return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
	}
	float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
	return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
__device__ __forceinline__ float dist_G1(const Dist& D, V3 w) { return 1 / (1 + dist_Lambda(D, w)); }                             // microfacet.h:22-25
__device__ __forceinline__ float dist_G(const Dist& D, V3 wo, V3 wi) { return 1 / (1 + dist_Lambda(D, wo) + dist_Lambda(D, wi)); } // microfacet.h:26-28
__device__ __forceinline__ float dist_Pdf(const Dist& D, V3 wo, V3 wh)                           // microfacet.cc:359-365
{
	if (D.vis) return dist_D(D, wh) * dist_G1(D, wo) * absdot(wo, wh) / fabsf(wo.z);
	return dist_D(D, wh) * fabsf(wh.z);
}
__device__ __forceinline__ V3 spherical(float sinTheta, float cosTheta, float phi) { float sp, cp; sincos_f(phi, &sp, &cp); return mk(sinTheta * cp, sinTheta * sp, cosTheta); }   // geometry.h:203-209
__device__ __forceinline__ V3 dist_sample_wh(const Dist& D, V3 wo, float u0, float u1)           // microfacet.cc:216-254 (Beckmann), :326-357 (TrowbridgeReitz)
{
	if (D.vis)
	{
		bool flip = wo.z < 0;
		V3 wh = stretch_sample(D, flip ? -wo : wo, u0, u1);
		if (flip) wh = -wh;
		return wh;
	}
	V3 wh;
	if (D.kind == JP_DIST_BECKMANN)
	{
		float tan2Theta, phi;
		if (D.ax == D.ay)
		{
			float logSample = m_logf(1 - u0);
			tan2Theta = -D.ax * D.ax * logSample;
			phi = u1 * 2 * JP_PI;
		}
		else
		{
			float logSample = m_logf(1 - u0);
			phi = m_atanf(D.ay / D.ax * m_tanf(2 * JP_PI * u1 + 0.5f * JP_PI));
			if (u1 > 0.5f) phi += JP_PI;
			float sinPhi, cosPhi; sincos_f(phi, &sinPhi, &cosPhi);
			float ax2 = D.ax * D.ax, ay2 = D.ay * D.ay;
			tan2Theta = -logSample / (cosPhi * cosPhi / ax2 + sinPhi * sinPhi / ay2);
		}
		float cosTheta = 1 / sqrtf(1 + tan2Theta);
		float sinTheta = sqrtf(smax((float)0, 1 - cosTheta * cosTheta));
		wh = spherical(sinTheta, cosTheta, phi);
	}
	else
	{
		float cosTheta = 0, phi = (2 * JP_PI) * u1;
		if (D.ax == D.ay)
		{
			float tanTheta2 = D.ax * D.ax * u0 / (1.0f - u0);
			cosTheta = 1 / sqrtf(1 + tanTheta2);
		}
		else
		{
			phi = m_atanf(D.ay / D.ax * m_tanf(2 * JP_PI * u1 + .5f * JP_PI));
			if (u1 > .5f) phi += JP_PI;
			float sinPhi, cosPhi; sincos_f(phi, &sinPhi, &cosPhi);
			const float ax2 = D.ax * D.ax, ay2 = D.ay * D.ay;
			const float alpha2 = 1 / (cosPhi * cosPhi / ax2 + sinPhi * sinPhi / ay2);
			float tanTheta2 = alpha2 * u0 / (1 - u0);
			cosTheta = 1 / sqrtf(1 + tanTheta2);
		}
		float sinTheta = sqrtf(smax((float)0., (float)1. - cosTheta * cosTheta));
		wh = spherical(sinTheta, cosTheta, phi);
	}
	if (!same_hemi(wo, wh)) wh = -wh;
	return wh;
}
__device__ __forceinline__ V3 fresnel_of(const JpBsdfDesc& d, float cosI)                        // bsdf.cc:15-24, bsdf.h:664-667
{
	if (d.fresnel == JP_FRESNEL_NOOP) return splat(1.f);
	if (d.fresnel == JP_FRESNEL_DIELECTRIC) return splat(fresnel_dielectric(cosI, d.fr_eta_i[0], d.fr_eta_t[0]));
	return fresnel_conductor(fabsf(cosI), mk(d.fr_eta_i[0], d.fr_eta_i[1], d.fr_eta_i[2]), mk(d.fr_eta_t[0], d.fr_eta_t[1], d.fr_eta_t[2]), mk(d.fr_k[0], d.fr_k[1], d.fr_k[2]));
}

struct Out { V3 f; float pdf; };
// Evalf_Local / Pdf_Local
__device__ __forceinline__ V3 x_eval(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi);
__device__ __forceinline__ float x_pdf(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi)
{
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: return same_hemi(wo, wi) ? fabsf(wi.z) * JP_INV_PI : 0;                    // bsdf.h:357-360
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:53-58
	{
		if (!same_hemi(wo, wi)) return 0;
		Dist D = make_dist(d);
		V3 wh = normalize(wo + wi);
		return dist_Pdf(D, wo, wh) / (4 * dot(wo, wh));
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:110-124
	{
		if (same_hemi(wo, wi)) return 0;
		Dist D = make_dist(d);
		float eta = wo.z > 0 ? (d.eta_b / d.eta_a) : (d.eta_a / d.eta_b);
		V3 wh = normalize(wo + wi * eta);
		if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
		float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
		float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
		return dist_Pdf(D, wo, wh) * dwh_dwi;
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:584-590, 622-626
	{
		const V3 wr = reflect(wo, mk(0, 0, 1));
		const float cosTheta = smax((float)0, dot(wr, wi));
		return (d.exponent + 1) * m_powf(cosTheta, d.exponent) * JP_XB_INV2PI;
	}
	default: return 0;                                                                               // delta BSDFs bsdf.h:410-413, 473-476
	}
}
__device__ __forceinline__ V3 x_eval(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi)
{
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: return same_hemi(wo, wi) ? mk(d.color[0], d.color[1], d.color[2]) * JP_INV_PI : splat(0);
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:35-51
	{
		Dist D = make_dist(d);
		float cosO = fabsf(wo.z), cosI = fabsf(wi.z);
		V3 wh = wi + wo;
		if (cosI == 0 || cosO == 0) return splat(0);
		if (wh.x == 0 && wh.y == 0 && wh.z == 0) return splat(0);
		wh = normalize(wh);
		V3 ff = (dot(wh, mk(0, 0, 1)) < 0) ? -wh : wh;
		V3 F = fresnel_of(d, dot(wi, ff));
		return cmul(mk(d.color[0], d.color[1], d.color[2]) * dist_D(D, wh) * dist_G(D, wo, wi), F) / (4 * cosI * cosO);
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:85-108
	{
		if (same_hemi(wo, wi)) return splat(0);
		Dist D = make_dist(d);
		float cosO = wo.z, cosI = wi.z;
		if (cosI == 0 || cosO == 0) return splat(0);
		float eta = wo.z > 0 ? (d.eta_b / d.eta_a) : (d.eta_a / d.eta_b);
		V3 wh = normalize(wo + wi * eta);
		if (wh.z < 0) wh = -wh;
		if (dot(wo, wh) * dot(wi, wh) > 0) return splat(0);
		V3 F = splat(fresnel_dielectric(dot(wo, wh), d.eta_a, d.eta_b));
		float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
		float factor = (1 / eta);
		return cmul(splat(1) - F, mk(d.color[0], d.color[1], d.color[2])) *
			fabsf(dist_D(D, wh) * dist_G(D, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cosI * cosO * sqrtDenom * sqrtDenom));
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:571-582
	{
		if (!same_hemi(wo, wi)) return splat(0);
		const V3 wr = reflect(wo, mk(0, 0, 1));
		const float cos_alpha = dot(wr, wi);
		const V3 rho = mk(d.color[0], d.color[1], d.color[2]) * (d.exponent + 2.f) * JP_XB_INV2PI;
		return rho * m_powf(cos_alpha, d.exponent);
	}
	default: return splat(0);
	}
}
__device__ __forceinline__ BsdfSample x_sample(const JpBsdfDesc& d, const Frame& fr, V3 wo, float ux, float uy)
{
	BsdfSample s; s.f = splat(0); s.wi = mk(0, 0, 1); s.pdf = 0; s.flags = 0;
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: case JP_BSDF_MIRROR: case JP_BSDF_FRESNEL_SPECULAR:
	{   // the closures the materials build: the path's own code
		Closure c; c.c0 = mk(d.color[0], d.color[1], d.color[2]); c.c1 = mk(d.color2[0], d.color2[1], d.color2[2]); c.eta_t = d.eta_b; c.ax = c.ay = 0; c.fresnel = FR_CONDUCTOR; c.feta = c.fk = splat(0); c.lam_o = 0;
		c.kind = d.kind == JP_BSDF_LAMBERT ? CL_LAMBERT : (d.kind == JP_BSDF_MIRROR ? CL_MIRROR : CL_FRESNEL_SPECULAR);
		return sample_local(c, wo, ux, uy);                                   // FFresnelSpecular on the device assumes etaI = 1 (material.h:72-75), checked by jp_bsdf
	}
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:60-78
	{
		if (wo.z == 0) return s;
		Dist D = make_dist(d);
		V3 wh = dist_sample_wh(D, wo, ux, uy);
		if (dot(wo, wh) < 0) return s;
		V3 wi = reflect(wo, wh);
		if (!same_hemi(wo, wi)) return s;
		s.wi = wi;
		s.f = x_eval(d, fr, wo, wi);
		s.pdf = dist_Pdf(D, wo, wh) / (4 * dot(wo, wh));
		s.flags = BS_REFLECTION | BS_GLOSSY;
		return s;
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:126-145
	{
		if (wo.z == 0) return s;
		Dist D = make_dist(d);
		V3 wh = dist_sample_wh(D, wo, ux, uy);
		if (dot(wo, wh) < 0) return s;
		V3 wi;
		float eta = wo.z > 0 ? (d.eta_a / d.eta_b) : (d.eta_b / d.eta_a);
		if (!refract(wo, wh, eta, &wi)) return s;
		s.wi = wi;
		// bsdf.cc:141 calls FBSDF::Pdf -- the WORLD-space entry (bsdf.h:290-293) -- on the local vectors: they go through ToLocal once more
		s.pdf = x_pdf(d, fr, to_local(fr, wo), to_local(fr, wi));
		s.f = x_eval(d, fr, wo, wi);
		s.flags = BS_TRANSMISSION | BS_GLOSSY;
		return s;
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:592-611
	{
		const float phi = 2 * JP_PI * ux;
		const float cos_theta = m_powf(uy, (float)1 / (d.exponent + 1));
		const float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
		float sphi, cphi; sincos_f(phi, &sphi, &cphi);
		V3 wl = mk(cphi * sin_theta, sphi * sin_theta, cos_theta);
		const V3 wr = reflect(wo, mk(0, 0, 1));
		Frame lobe = frame_from_z(wr);
		s.wi = to_world(lobe, wl);
		if (wo.z < 0) s.wi.z *= -1;
		s.f = x_eval(d, fr, wo, s.wi);
		s.pdf = x_pdf(d, fr, wo, s.wi);
		s.flags = BS_REFLECTION | BS_GLOSSY;
		return s;
	}
	}
	return s;
}
} // namespace xb
} // namespace jp
