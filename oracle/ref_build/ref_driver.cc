// oracle/ref_build/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never part of the shipped product).
//
// A C-ABI driver over the UNMODIFIED reference sources under /root/reference/src (compiled where
// they lie by oracle/ref_build/Makefile; nothing is copied).  It exists to
//   * pin the CPU restatement in oracle/pt_oracle.cc (whole-film bit-exact comparisons), and
//   * generate the golden vectors committed under tests/golden/ (tests/golden/make_golden.py).
// Everything here goes through the reference's PUBLIC interfaces:
//   scene building  : FScene::Create{Camera,Light,Material,Shape,Primitive,TriangleMesh,Primitives,
//                     AreaLight(s)}  (scene.h:66-124, scene.cc:49-97), exactly as main.cc:13-111 does;
//   rendering       : FIntegrator::DoRender (integrator.cc:82-111) per 20-row band with a cloned
//                     sampler, bands run by the reference's FParallelSystem (parallel.cc) -- this is
//                     FIntegrator::Render (integrator.cc:35-80) minus its Win32 timer, which cannot
//                     be built here (see prelude.h);
//   sampler hook    : a counter-based FSampler subclass injected through the virtual interface
//                     sampler.h:64-105 (tier T1 of SURVEY.md section 8c).
#include "pbrt.h"
#include "light.h"
#include "integrator.h"
#include "parallel.h"
#include "microfacet.h"

#include <atomic>
#include <map>
#include <cstring>

#include "jp_counter_rng.h"   // include/jp_counter_rng.h: the shared counter-based stream (own code)
#include "jetpbrt_amd.h"      // JpBsdfDesc: the by-value description ref_bsdf_direct builds the reference's BSDF objects from

using namespace pbrt;

namespace {

// ---- counter-based sampler injected through the reference's FSampler interface -----------------
class FCounterSampler : public FSampler
{
public:
	FCounterSampler(int spp, uint32_t seed) : FSampler(spp), seed(seed), key(0), dim(0) { current_sample_index = 0; }
	std::unique_ptr<FSampler> Clone() override { return std::make_unique<FCounterSampler>(samples_per_pixel, seed); }
	Float GetFloat() override { return jp_rng_float(key, dim++); }
	FFloat2 GetFloat2() override
	{
		Float a = jp_rng_float(key, dim++);   // explicit order: first draw -> .x
		Float b = jp_rng_float(key, dim++);
		return FFloat2(a, b);
	}
	FCameraSample GetCameraSample(const FPoint2& posfilm) override
	{
		key = jp_rng_key(seed, (uint32_t)(int)posfilm.x, (uint32_t)(int)posfilm.y, (uint32_t)current_sample_index);
		dim = 0;
		FCameraSample cs;
		Float a = jp_rng_float(key, dim++);
		Float b = jp_rng_float(key, dim++);
		cs.posfilm = posfilm + FFloat2(a, b);
		return cs;
	}
private:
	uint32_t seed, key, dim;
};

// sampler returning scripted values (for per-function known-answer vectors)
class FScriptSampler : public FSampler
{
public:
	FScriptSampler(const float* v, int n) : FSampler(1), vals(v, v + n), pos(0) { current_sample_index = 0; }
	std::unique_ptr<FSampler> Clone() override { return std::make_unique<FScriptSampler>(vals.data(), (int)vals.size()); }
	Float next() { Float r = pos < vals.size() ? vals[pos] : 0.5f; pos++; return r; }
	Float GetFloat() override { return next(); }
	FFloat2 GetFloat2() override { Float a = next(); Float b = next(); return FFloat2(a, b); }
	FCameraSample GetCameraSample(const FPoint2& p) override { FCameraSample cs; Float a = next(); Float b = next(); cs.posfilm = p + FFloat2(a, b); return cs; }
	std::vector<float> vals; size_t pos;
};

// ---- ray counters: forwarding BVH root placed in the public FScene::shadow_bvh -----------------
struct RefCounters { std::atomic<unsigned long long> closest{0}, closest_hit{0}, shadow{0}, shadow_hit{0}; };

class FCountingRoot : public FBVH_NodeBase
{
public:
	FCountingRoot(FBVH_NodeBase* inner, RefCounters* c) : inner(inner), c(c) {}
	bool Intersect(const FRay& ray, FIntersection& oisect) const override
	{
		const bool isShadow = !std::isinf(ray.MaxT());
		bool hit = inner->Intersect(ray, oisect);
		if (isShadow) { c->shadow++; if (hit) c->shadow_hit++; }
		else { c->closest++; if (hit) c->closest_hit++; }
		return hit;
	}
	FBVH_NodeBase* inner; RefCounters* c;
};

// exposes the protected DoRender()/Li() of the configured integrator (integrator.h:32-39)
class FExposedPathIntegrator : public FPathIntegratorIteration
{
public:
	using FPathIntegratorIteration::FPathIntegratorIteration;
	void Band(const FScene* s, FSampler* smp, FFilmView* v) const { DoRender(s, smp, v); }
	FColor LiPublic(const FRay& r, const FScene* s, FSampler* smp) const { return Li(r, s, smp); }
};

class FExposedRecursiveIntegrator : public FPathIntegratorRecursive        // integrator.h:88-106
{
public:
	using FPathIntegratorRecursive::FPathIntegratorRecursive;
	void Band(const FScene* s, FSampler* smp, FFilmView* v) const { DoRender(s, smp, v); }
};

class FExposedWhittedIntegrator : public FWhittedIntegrator                 // integrator.h:62-85
{
public:
	using FWhittedIntegrator::FWhittedIntegrator;
	void Band(const FScene* s, FSampler* smp, FFilmView* v) const { DoRender(s, smp, v); }
};
class FExposedDebugIntegrator : public FDebugIntegrator                     // integrator.h:44-58
{
public:
	void Band(const FScene* s, FSampler* smp, FFilmView* v) const { DoRender(s, smp, v); }
};

class FBandTask : public FTask
{
public:
	FBandTask(const FExposedPathIntegrator* i, const FScene* s, std::shared_ptr<FSampler> smp, const FFilmView& v)
		: integ(i), scene(s), sampler(smp), view(v) {}
	void Execute() override { integ->Band(scene, sampler.get(), &view); }
	const FExposedPathIntegrator* integ; const FScene* scene; std::shared_ptr<FSampler> sampler; FFilmView view;
};

struct RefScene
{
	std::shared_ptr<FScene> scene;
	std::vector<std::shared_ptr<FMaterial>> mats;
	std::map<const FPrimitive*, int> primIndex;   // creation order (the BVH build re-sorts shadow_primitives)
	RefCounters counters;
	std::shared_ptr<FCountingRoot> countingRoot;
	bool preprocessed = false;
};

FColor C3(const float* v) { return FColor(v[0], v[1], v[2]); }
FVector3 V3(const float* v) { return FVector3(v[0], v[1], v[2]); }

void attach(RefScene* rs, const std::shared_ptr<FShape>& shape, int mat, const float* radiance)
{
	std::shared_ptr<FMaterial> m = mat >= 0 ? rs->mats[mat] : nullptr;
	if (radiance) rs->scene->CreateAreaLight(1, C3(radiance), shape, m);           // main.cc:36,86 / scene.cc:91-97
	else rs->scene->CreatePrimitive(shape.get(), m.get(), nullptr);                // main.cc:91
}

} // namespace

extern "C" {

void* ref_scene_new(const char* name) { RefScene* rs = new RefScene; rs->scene = std::make_shared<FScene>(name); return rs; }
void  ref_scene_free(void* h) { delete (RefScene*)h; }

void ref_scene_camera(void* h, const float* lookfrom, const float* front, const float* up, float vfov, float resx, float resy)
{
	RefScene* rs = (RefScene*)h;
	rs->scene->CreateCamera<FCamera>(V3(lookfrom), V3(front), V3(up), vfov, FVector2(resx, resy));   // main.cc:22
}

int ref_scene_envlight(void* h, const float* rgb)
{
	RefScene* rs = (RefScene*)h;
	rs->scene->CreateLight<FEnvironmentLight>(FPoint3(0, 0, 0), 1, C3(rgb));                         // main.cc:25
	return rs->scene->LightNum() - 1;
}

int ref_scene_pointlight(void* h, const float* pos, const float* intensity)
{ RefScene* rs = (RefScene*)h; rs->scene->CreateLight<FPointLight>(V3(pos), 1, C3(intensity)); return rs->scene->LightNum() - 1; }   // main.cc:38 (commented-out line)
int ref_scene_dirlight(void* h, const float* dir, const float* irradiance)
{ RefScene* rs = (RefScene*)h; rs->scene->CreateLight<FDirectionLight>(FPoint3(0, 0, 0), 1, C3(irradiance), V3(dir)); return rs->scene->LightNum() - 1; }

int ref_mat_matte(void* h, const float* rgb)
{ RefScene* rs = (RefScene*)h; rs->mats.push_back(rs->scene->CreateMaterial<FMatteMaterial>(C3(rgb))); return (int)rs->mats.size() - 1; }
int ref_mat_mirror(void* h, const float* rgb)
{ RefScene* rs = (RefScene*)h; rs->mats.push_back(rs->scene->CreateMaterial<FMirrorMaterial>(C3(rgb))); return (int)rs->mats.size() - 1; }
int ref_mat_glass(void* h, float eta, const float* kr, const float* kt)
{ RefScene* rs = (RefScene*)h; rs->mats.push_back(rs->scene->CreateMaterial<FGlassMaterial>(eta, C3(kr), C3(kt))); return (int)rs->mats.size() - 1; }
int ref_mat_plastic(void* h, const float* kd, const float* ks, float rough, int remap)
{ RefScene* rs = (RefScene*)h; rs->mats.push_back(rs->scene->CreateMaterial<FPlasticMaterial>(C3(kd), C3(ks), rough, remap != 0)); return (int)rs->mats.size() - 1; }
int ref_mat_metal(void* h, const float* eta, const float* k, float ur, float vr, int remap)
{ RefScene* rs = (RefScene*)h; rs->mats.push_back(rs->scene->CreateMaterial<FMetalMaterial>(C3(eta), C3(k), ur, vr, remap != 0)); return (int)rs->mats.size() - 1; }

// OBJ mesh through the reference's own ingest (shape.cc:23-68 + external/obj_loader.h); returns #triangles
int ref_scene_mesh(void* h, const char* path, int flip_normal, int flip_handedness, const float* offset, float scale, int mat, const float* radiance)
{
	RefScene* rs = (RefScene*)h;
	std::vector<std::shared_ptr<FShape>> mesh = rs->scene->CreateTriangleMesh(path, flip_normal != 0, flip_handedness != 0, V3(offset), scale);
	std::shared_ptr<FMaterial> m = mat >= 0 ? rs->mats[mat] : nullptr;
	if (radiance) rs->scene->CreateAreaLights(1, C3(radiance), mesh, m);            // main.cc:36
	else rs->scene->CreatePrimitives(mesh, m);                                      // main.cc:42
	return (int)mesh.size();
}

// axis: 0 = FromXY(a0,a1,b0,b1,c), 1 = FromXZ, 2 = FromYZ   (shape.cc:76-95)
void ref_scene_rect(void* h, int axis, float a0, float a1, float b0, float b1, float c, int flip, int mat, const float* radiance)
{
	RefScene* rs = (RefScene*)h;
	FRectangle r = axis == 0 ? FRectangle::FromXY(a0, a1, b0, b1, c, flip != 0)
	             : axis == 1 ? FRectangle::FromXZ(a0, a1, b0, b1, c, flip != 0)
	                         : FRectangle::FromYZ(a0, a1, b0, b1, c, flip != 0);
	std::shared_ptr<FShape> shape = rs->scene->CreateShape<FRectangle>(r);          // main.cc:85,90
	attach(rs, shape, mat, radiance);
}

void ref_scene_sphere(void* h, const float* center, float radius, int mat, const float* radiance)
{
	RefScene* rs = (RefScene*)h;
	std::shared_ptr<FShape> shape = rs->scene->CreateShape<FSphere>(V3(center), radius);   // main.cc:57
	attach(rs, shape, mat, radiance);
}

void ref_scene_disk(void* h, const float* pos, const float* normal, float radius, int mat, const float* radiance)
{
	RefScene* rs = (RefScene*)h;
	std::shared_ptr<FShape> shape = rs->scene->CreateShape<FDisk>(V3(pos), V3(normal), radius);   // shape.h:192
	attach(rs, shape, mat, radiance);
}

void ref_scene_preprocess(void* h)
{
	RefScene* rs = (RefScene*)h;
	for (size_t i = 0; i < rs->scene->shadow_primitives.size(); i++) rs->primIndex[rs->scene->shadow_primitives[i]] = (int)i;
	rs->scene->Preprocess();                                                        // scene.cc:11-23
	rs->countingRoot = std::make_shared<FCountingRoot>(rs->scene->shadow_bvh, &rs->counters);
	rs->scene->shadow_bvh = rs->countingRoot.get();
	rs->preprocessed = true;
}

int ref_num_primitives(void* h) { return (int)((RefScene*)h)->scene->shadow_primitives.size(); }
int ref_num_lights(void* h) { return ((RefScene*)h)->scene->LightNum(); }

// sampler_mode 0: stock FRandomSampler (mt19937_64, seed 1234 per band); 1: counter sampler with `seed`.
// nthreads < 1: the serial whole-frame path of Render() (integrator.cc:45-50).
int ref_render(void* h, int W, int H, int spp, int maxdepth, int sampler_mode, unsigned seed, int nthreads, float* film_out)
{
	RefScene* rs = (RefScene*)h;
	if (!rs->preprocessed) return -1;
	FFilm film(W, H);
	FExposedPathIntegrator integ(maxdepth);
	std::shared_ptr<FSampler> sampler;
	if (sampler_mode == 0) sampler = std::make_shared<FRandomSampler>(spp);
	else sampler = std::make_shared<FCounterSampler>(spp, seed);

	if (nthreads < 1)
	{
		FFilmView view(&film, 0, 0, W, H);
		integ.Band(rs->scene.get(), sampler.get(), &view);
	}
	else
	{
		const int lines_per_task = 20;                                               // integrator.cc:53
		FParallelSystem parallel;
		std::vector<std::shared_ptr<FBandTask>> tasks;
		for (int y = 0; y < H; y += lines_per_task)
		{
			int endy = y + lines_per_task; if (endy > H) endy = H;
			std::shared_ptr<FSampler> dup = sampler->Clone();                        // integrator.cc:66
			tasks.push_back(std::make_shared<FBandTask>(&integ, rs->scene.get(), dup, FFilmView(&film, 0, y, W, endy)));
			parallel.AddTask(tasks.back().get());
		}
		parallel.Start(nthreads);
		parallel.WaitForFinish();
	}
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		const FColor& c = film(x, y);                                                // film.h:51
		float* o = film_out + 3 * ((size_t)y * W + x);
		o[0] = c.r; o[1] = c.g; o[2] = c.b;
	}
	return 0;
}

// FPathIntegratorRecursive over the whole frame, serial, counter sampler (for the "same estimator" check of the host alias)
int ref_render_recursive(void* h, int W, int H, int spp, int maxdepth, unsigned seed, float* film_out)
{
	RefScene* rs = (RefScene*)h;
	if (!rs->preprocessed) return -1;
	FFilm film(W, H);
	FExposedRecursiveIntegrator integ(maxdepth);
	FCounterSampler sampler(spp, seed);
	FFilmView view(&film, 0, 0, W, H);
	integ.Band(rs->scene.get(), &sampler, &view);
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		const FColor& c = film(x, y);
		float* o = film_out + 3 * ((size_t)y * W + x);
		o[0] = c.r; o[1] = c.g; o[2] = c.b;
	}
	return 0;
}

// kind 1: FWhittedIntegrator(maxdepth), kind 2: FDebugIntegrator -- whole frame, serial, counter sampler
int ref_render_other(void* h, int kind, int W, int H, int spp, int maxdepth, unsigned seed, float* film_out)
{
	RefScene* rs = (RefScene*)h;
	if (!rs->preprocessed) return -1;
	FFilm film(W, H);
	FCounterSampler sampler(spp, seed);
	FFilmView view(&film, 0, 0, W, H);
	if (kind == 1) { FExposedWhittedIntegrator integ(maxdepth); integ.Band(rs->scene.get(), &sampler, &view); }
	else { FExposedDebugIntegrator integ; integ.Band(rs->scene.get(), &sampler, &view); }
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
	{
		const FColor& c = film(x, y);
		float* o = film_out + 3 * ((size_t)y * W + x);
		o[0] = c.r; o[1] = c.g; o[2] = c.b;
	}
	return 0;
}

void ref_counters_reset(void* h) { RefScene* rs = (RefScene*)h; rs->counters.closest = 0; rs->counters.closest_hit = 0; rs->counters.shadow = 0; rs->counters.shadow_hit = 0; }
void ref_counters_get(void* h, unsigned long long* out4)
{ RefScene* rs = (RefScene*)h; out4[0] = rs->counters.closest; out4[1] = rs->counters.closest_hit; out4[2] = rs->counters.shadow; out4[3] = rs->counters.shadow_hit; }

// ---- known-answer hooks ------------------------------------------------------------------------
// closest-hit records through FScene::Intersect (scene.cc:25-33)
void ref_trace(void* h, int n, const float* o, const float* d, const float* tmin, const float* tmax,
               int* hit, float* t, int* prim, float* nrm, float* pos)
{
	RefScene* rs = (RefScene*)h;
	for (int i = 0; i < n; i++)
	{
		FRay ray(V3(o + 3 * i), V3(d + 3 * i), tmin[i], tmax[i]);
		FIntersection isect;
		bool b = rs->scene->Intersect(ray, isect);
		hit[i] = b ? 1 : 0;
		t[i] = ray.MaxT();
		prim[i] = b ? rs->primIndex[isect.primitive] : -1;
		for (int k = 0; k < 3; k++) { nrm[3 * i + k] = b ? isect.normal[k] : 0.f; pos[3 * i + k] = b ? isect.position[k] : 0.f; }
	}
}

// camera rays through FCamera::GenerateRay (camera.h:52-58); film positions in pixels
void ref_camera_rays(void* h, int n, const float* pxy, float* o, float* d)
{
	RefScene* rs = (RefScene*)h;
	for (int i = 0; i < n; i++)
	{
		FCameraSample cs; cs.posfilm = FPoint2(pxy[2 * i], pxy[2 * i + 1]);
		FRay r = rs->scene->Camera()->GenerateRay(cs);
		for (int k = 0; k < 3; k++) { o[3 * i + k] = r.origin[k]; d[3 * i + k] = r.dir[k]; }
	}
}

// BSDF closure of material `mat` at a surface with normal `n`: f = Evalf(wo,wi); sample = Sample(wo,u2).
// `uscat` feeds the draw FPlasticMaterial::Scattering consumes (material.cc:14).
// out: feval[3], sf[3], swi[3], spdf, sflags, isdelta
void ref_bsdf(void* h, int count, int mat, const float* n, const float* wo, const float* wi, const float* u2, const float* uscat,
              float* feval, float* sf, float* swi, float* spdf, int* sflags, int* isdelta)
{
	RefScene* rs = (RefScene*)h;
	for (int i = 0; i < count; i++)
	{
		FIntersection isect(FPoint3(0, 0, 0), V3(n + 3 * i), V3(wo + 3 * i));
		FScriptSampler smp(uscat + i, 1);
		std::unique_ptr<FBSDF> b = rs->mats[mat]->Scattering(isect, &smp);
		FColor f = b->Evalf(V3(wo + 3 * i), V3(wi + 3 * i));
		FBSDFSample s = b->Sample(V3(wo + 3 * i), FFloat2(u2[2 * i], u2[2 * i + 1]));
		feval[3 * i] = f.r; feval[3 * i + 1] = f.g; feval[3 * i + 2] = f.b;
		sf[3 * i] = s.f.r; sf[3 * i + 1] = s.f.g; sf[3 * i + 2] = s.f.b;
		swi[3 * i] = s.wi.x; swi[3 * i + 1] = s.wi.y; swi[3 * i + 2] = s.wi.z;
		spdf[i] = s.pdf; sflags[i] = s.ebsdf; isdelta[i] = b->IsDelta() ? 1 : 0;
	}
}

// FFilm::SaveAsImage (film.cc:11-188) of the unmodified reference on a given fp32 film: type 0 PPM, 1 BMP, 2 HDR; writes <filename>.<ext>
int ref_film_save(const float* rgb, int W, int H, const char* filename, int type)
{
	FFilm film(W, H);
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const float* p = rgb + 3 * ((size_t)y * W + x); film(x, y) = FColor(p[0], p[1], p[2]); }
	return film.SaveAsImage(filename, type == 0 ? EImageType::PPM : (type == 2 ? EImageType::HDR : EImageType::BMP)) ? 1 : 0;
}
// gamma_encoding (film.h:24) of the reference for n values
void ref_gamma_encode(const float* x, int n, unsigned char* out) { for (int i = 0; i < n; i++) out[i] = gamma_encoding(x[i]); }

// The reference's BSDF classes constructed directly (the ones no material instantiates included) from a JpBsdfDesc, and FBSDF::Evalf /
// Pdf / Sample (bsdf.h:284-302) called on them: frame = FFrame(normal)
static std::unique_ptr<FBSDF> make_bsdf(const JpBsdfDesc& d, const FFrame& frame)
{
	auto col = [](const float* p) { return FColor(p[0], p[1], p[2]); };
	auto dist = [&]() -> MicrofacetDistribution* {
		if (d.distribution == JP_DIST_BECKMANN) return new BeckmannDistribution(d.alpha_x, d.alpha_y, d.sample_visible != 0);
		return new TrowbridgeReitzDistribution(d.alpha_x, d.alpha_y, d.sample_visible != 0);
	};
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: return std::make_unique<FLambertionReflection>(frame, col(d.color));
	case JP_BSDF_MIRROR: return std::make_unique<FSpecularReflection>(frame, col(d.color));
	case JP_BSDF_FRESNEL_SPECULAR: return std::make_unique<FFresnelSpecular>(frame, d.eta_a, d.eta_b, col(d.color), col(d.color2));
	case JP_BSDF_MICROFACET_REFLECTION:
	{
		Fresnel* fr = d.fresnel == JP_FRESNEL_NOOP ? (Fresnel*)new FresnelNoOp() : d.fresnel == JP_FRESNEL_DIELECTRIC ? (Fresnel*)new FresnelDielectric(d.fr_eta_i[0], d.fr_eta_t[0])
		              : (Fresnel*)new FresnelConductor(col(d.fr_eta_i), col(d.fr_eta_t), col(d.fr_k));
		return std::make_unique<FMicrofacetReflection>(frame, col(d.color), dist(), fr);
	}
	case JP_BSDF_MICROFACET_TRANSMISSION: return std::make_unique<FMicrofacetTransmission>(frame, col(d.color), dist(), d.eta_a, d.eta_b);
	case JP_BSDF_PHONG: return std::make_unique<FPhongSpecularReflection>(frame, col(d.color), d.exponent);
	}
	return nullptr;
}
void ref_bsdf_direct(const JpBsdfDesc* d, int count, const float* n, const float* wo, const float* wi, const float* u2,
                     float* feval, float* pdfeval, float* sf, float* swi, float* spdf, int* sflags)
{
	for (int i = 0; i < count; i++)
	{
		FFrame frame(V3(n + 3 * i));
		std::unique_ptr<FBSDF> b = make_bsdf(*d, frame);
		FColor f = b->Evalf(V3(wo + 3 * i), V3(wi + 3 * i));
		pdfeval[i] = b->Pdf(V3(wo + 3 * i), V3(wi + 3 * i));
		FBSDFSample s = b->Sample(V3(wo + 3 * i), FFloat2(u2[2 * i], u2[2 * i + 1]));
		feval[3 * i] = f.r; feval[3 * i + 1] = f.g; feval[3 * i + 2] = f.b;
		sf[3 * i] = s.f.r; sf[3 * i + 1] = s.f.g; sf[3 * i + 2] = s.f.b;
		swi[3 * i] = s.wi.x; swi[3 * i + 1] = s.wi.y; swi[3 * i + 2] = s.wi.z;
		spdf[i] = s.pdf; sflags[i] = s.ebsdf;
	}
}

// FLight::Sample_Li of light `li` (Lights() order) from a surface point p with normal n
void ref_light_sample(void* h, int count, int li, const float* p, const float* n, const float* u2,
                      float* pos, float* wi, float* pdf, float* Li)
{
	RefScene* rs = (RefScene*)h;
	const FLight* light = rs->scene->Lights()[li];
	for (int i = 0; i < count; i++)
	{
		FIntersection isect(V3(p + 3 * i), V3(n + 3 * i), FVector3(0, 0, 1));
		FLightSample s = light->Sample_Li(isect, FFloat2(u2[2 * i], u2[2 * i + 1]));
		for (int k = 0; k < 3; k++) { pos[3 * i + k] = s.pos[k]; wi[3 * i + k] = s.wi[k]; }
		pdf[i] = s.pdf; Li[3 * i] = s.Li.r; Li[3 * i + 1] = s.Li.g; Li[3 * i + 2] = s.Li.b;
	}
}

// Li() of single camera samples with scripted random numbers (nvals per path), for path-level KATs
void ref_li_scripted(void* h, int count, int maxdepth, const float* pxy, const float* vals, int nvals, float* out)
{
	RefScene* rs = (RefScene*)h;
	FExposedPathIntegrator integ(maxdepth);
	for (int i = 0; i < count; i++)
	{
		FScriptSampler smp(vals + (size_t)i * nvals, nvals);
		FCameraSample cs = smp.GetCameraSample(FPoint2(pxy[2 * i], pxy[2 * i + 1]));
		FRay r = rs->scene->Camera()->GenerateRay(cs);
		FColor c = integ.LiPublic(r, rs->scene.get(), &smp);
		out[3 * i] = c.r; out[3 * i + 1] = c.g; out[3 * i + 2] = c.b;
	}
}

// the stock stream, for pinning the restated mt19937_64 -> float conversion (sampler.h:16-54)
void ref_stock_stream(int n, float* out)
{
	FRNG rng;
	for (int i = 0; i < n; i++) out[i] = rng.uniform_float();
}
void ref_stock_float2(float* out2) { FRNG rng; FFloat2 v = rng.uniform_float2(); out2[0] = v.x; out2[1] = v.y; }

} // extern "C"
