// jet-pbrt_amd/host/c_api.cc -- a procedural C view of the host API (jp_host_*), so that Python harnesses
// (tests, bench.py, __graft_entry__) can build scenes through the SAME call sequence they use on the compiled
// reference (oracle/ref_build/ref_driver.cc: ref_*), i.e. the calls of main.cc:13-111.
#include "jetpbrt.h"

#include <cstring>

using namespace jetpbrt;

namespace
{
struct HostScene
{
	std::shared_ptr<FScene> scene;
	std::vector<std::shared_ptr<FMaterial>> mats;
	FlatScene flat; bool flattened = false;
	std::unique_ptr<FGpuPathIntegrator> integ; int integDepth = -1;
	std::string error;
};
FColor C3(const float* v) { return FColor(v[0], v[1], v[2]); }
FVector3 V3(const float* v) { return FVector3(v[0], v[1], v[2]); }
void attach(HostScene* hs, const std::shared_ptr<FShape>& shape, int mat, const float* radiance)
{
	std::shared_ptr<FMaterial> m = mat >= 0 ? hs->mats[mat] : nullptr;
	if (radiance) hs->scene->CreateAreaLight(1, C3(radiance), shape, m);
	else hs->scene->CreatePrimitive(shape.get(), m.get(), (const FAreaLight*)nullptr);
}
}

extern "C" {

void* jp_host_scene_new(const char* name) { HostScene* hs = new HostScene; hs->scene = std::make_shared<FScene>(name); return hs; }
void  jp_host_scene_free(void* h) { delete (HostScene*)h; }
const char* jp_host_last_error(void* h) { return ((HostScene*)h)->error.c_str(); }

void jp_host_scene_camera(void* h, const float* lookfrom, const float* front, const float* up, float vfov, float resx, float resy)
{ ((HostScene*)h)->scene->CreateCamera<FCamera>(V3(lookfrom), V3(front), V3(up), vfov, FVector2(resx, resy)); }

int jp_host_scene_envlight(void* h, const float* rgb)
{ HostScene* hs = (HostScene*)h; hs->scene->CreateLight<FEnvironmentLight>(FPoint3(0, 0, 0), 1, C3(rgb)); return hs->scene->LightNum() - 1; }

int jp_host_scene_pointlight(void* h, const float* pos, const float* intensity)
{ HostScene* hs = (HostScene*)h; hs->scene->CreateLight<FPointLight>(V3(pos), 1, C3(intensity)); return hs->scene->LightNum() - 1; }
int jp_host_scene_dirlight(void* h, const float* dir, const float* irradiance)
{ HostScene* hs = (HostScene*)h; hs->scene->CreateLight<FDirectionLight>(FPoint3(0, 0, 0), 1, C3(irradiance), V3(dir)); return hs->scene->LightNum() - 1; }

int jp_host_mat_matte(void* h, const float* rgb) { HostScene* hs = (HostScene*)h; hs->mats.push_back(hs->scene->CreateMaterial<FMatteMaterial>(C3(rgb))); return (int)hs->mats.size() - 1; }
int jp_host_mat_mirror(void* h, const float* rgb) { HostScene* hs = (HostScene*)h; hs->mats.push_back(hs->scene->CreateMaterial<FMirrorMaterial>(C3(rgb))); return (int)hs->mats.size() - 1; }
int jp_host_mat_glass(void* h, float eta, const float* kr, const float* kt) { HostScene* hs = (HostScene*)h; hs->mats.push_back(hs->scene->CreateMaterial<FGlassMaterial>(eta, C3(kr), C3(kt))); return (int)hs->mats.size() - 1; }
int jp_host_mat_plastic(void* h, const float* kd, const float* ks, float rough, int remap) { HostScene* hs = (HostScene*)h; hs->mats.push_back(hs->scene->CreateMaterial<FPlasticMaterial>(C3(kd), C3(ks), rough, remap != 0)); return (int)hs->mats.size() - 1; }
int jp_host_mat_metal(void* h, const float* eta, const float* k, float ur, float vr, int remap) { HostScene* hs = (HostScene*)h; hs->mats.push_back(hs->scene->CreateMaterial<FMetalMaterial>(C3(eta), C3(k), ur, vr, remap != 0)); return (int)hs->mats.size() - 1; }

int jp_host_scene_mesh(void* h, const char* path, int flip_normal, int flip_handedness, const float* offset, float scale, int mat, const float* radiance)
{
	HostScene* hs = (HostScene*)h;
	std::vector<std::shared_ptr<FShape>> mesh = hs->scene->CreateTriangleMesh(path, flip_normal != 0, flip_handedness != 0, V3(offset), scale);
	std::shared_ptr<FMaterial> m = mat >= 0 ? hs->mats[mat] : nullptr;
	if (radiance) hs->scene->CreateAreaLights(1, C3(radiance), mesh, m);
	else hs->scene->CreatePrimitives(mesh, m);
	return (int)mesh.size();
}

void jp_host_scene_rect(void* h, int axis, float a0, float a1, float b0, float b1, float c, int flip, int mat, const float* radiance)
{
	HostScene* hs = (HostScene*)h;
	FRectangle r = axis == 0 ? FRectangle::FromXY(a0, a1, b0, b1, c, flip != 0) : axis == 1 ? FRectangle::FromXZ(a0, a1, b0, b1, c, flip != 0) : FRectangle::FromYZ(a0, a1, b0, b1, c, flip != 0);
	attach(hs, hs->scene->CreateShape<FRectangle>(r), mat, radiance);
}

void jp_host_scene_sphere(void* h, const float* center, float radius, int mat, const float* radiance)
{ HostScene* hs = (HostScene*)h; attach(hs, hs->scene->CreateShape<FSphere>(V3(center), radius), mat, radiance); }

void jp_host_scene_disk(void* h, const float* pos, const float* normal, float radius, int mat, const float* radiance)
{ HostScene* hs = (HostScene*)h; attach(hs, hs->scene->CreateShape<FDisk>(V3(pos), V3(normal), radius), mat, radiance); }

void jp_host_scene_set_reference_tree(void* h, int on) { ((HostScene*)h)->scene->referenceTree = on != 0; ((HostScene*)h)->scene->certifiedWalk = on == 2; }   // 1: verbatim walk, 2: certified walk
void jp_host_scene_set_device_build(void* h, int on) { ((HostScene*)h)->scene->deviceBuild = on != 0; ((HostScene*)h)->scene->hostBuild = on == 0; }   // explicit either way
void jp_host_scene_preprocess(void* h) { HostScene* hs = (HostScene*)h; hs->scene->Preprocess(); hs->flattened = false; }
int  jp_host_num_primitives(void* h) { return (int)((HostScene*)h)->scene->primitives.size(); }
int  jp_host_num_lights(void* h) { return ((HostScene*)h)->scene->LightNum(); }

// the flattened SoA view (owned by the handle; valid until the next preprocess/free)
const JpScene* jp_host_flatten(void* h)
{
	HostScene* hs = (HostScene*)h;
	if (!hs->flattened) { if (!FlattenScene(*hs->scene, hs->flat, &hs->error)) return nullptr; hs->flattened = true; }
	return &hs->flat.view;
}

// FGpuPathIntegrator(maxdepth).Render(scene, FCounterSampler(spp, seed), film, numthreads) -> rgb (added onto zeros)
int jp_host_render(void* h, int W, int H, int spp, int maxdepth, unsigned seed, int device, int shard_index, int shard_count, float* film_out, JpCounters* counters)
{
	HostScene* hs = (HostScene*)h;
	if (!hs->integ || hs->integDepth != maxdepth) { hs->integ.reset(new FGpuPathIntegrator(maxdepth, device)); hs->integDepth = maxdepth; }
	hs->integ->SetShard(shard_index, shard_count > 0 ? shard_count : 1);
	FFilm film(W, H);
	FCounterSampler sampler(spp, seed);
	hs->integ->Render(hs->scene.get(), &sampler, &film, 16);
	if (hs->integ->LastStatus() != JP_OK) return hs->integ->LastStatus();
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const FColor& c = static_cast<const FFilm&>(film)(x, y); float* o = film_out + 3 * ((size_t)y * W + x); o[0] = c.r; o[1] = c.g; o[2] = c.b; }
	if (counters) *counters = hs->integ->Counters();
	return JP_OK;
}

// FWhittedIntegrator(maxdepth) (kind 1) / FDebugIntegrator (kind 2) .Render(scene, FCounterSampler(spp, seed), film, numthreads)
int jp_host_render_other(void* h, int kind, int W, int H, int spp, int maxdepth, unsigned seed, int device, float* film_out)
{
	HostScene* hs = (HostScene*)h;
	std::unique_ptr<FGpuPathIntegrator> integ;
	if (kind == JP_INTEGRATOR_WHITTED) integ.reset(new FWhittedIntegrator(maxdepth, device)); else integ.reset(new FDebugIntegrator(device));
	FFilm film(W, H);
	FCounterSampler sampler(spp, seed);
	integ->Render(hs->scene.get(), &sampler, &film, 16);
	if (integ->LastStatus() != JP_OK) return integ->LastStatus();
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const FColor& c = static_cast<const FFilm&>(film)(x, y); float* o = film_out + 3 * ((size_t)y * W + x); o[0] = c.r; o[1] = c.g; o[2] = c.b; }
	return JP_OK;
}

// The film output path with the tone map on the device: FFilm::RequestDeviceLDR(ldr_only) -> Render -> the 8-bit pixels (and,
// unless ldr_only, the fp32 film) -> optionally FFilm::SaveAsImage(filename, type).  rgb8_out: W*H*3 bytes; film_out may be null.
int jp_host_render_ldr(void* h, int W, int H, int spp, int maxdepth, unsigned seed, int device, int ldr_only, unsigned char* rgb8_out, float* film_out, const char* filename, int type)
{
	HostScene* hs = (HostScene*)h;
	if (ldr_only && ((filename && filename[0] && type == 2) || film_out)) return JP_ERR_INVALID_ARGUMENT;   // an LDR-only film has no fp32 pixels for an .hdr file or a float buffer
	if (!hs->integ || hs->integDepth != maxdepth) { hs->integ.reset(new FGpuPathIntegrator(maxdepth, device)); hs->integDepth = maxdepth; }
	hs->integ->SetShard(0, 1);
	FFilm film(W, H);
	film.RequestDeviceLDR(ldr_only != 0);
	FCounterSampler sampler(spp, seed);
	hs->integ->Render(hs->scene.get(), &sampler, &film, 16);
	if (hs->integ->LastStatus() != JP_OK) return hs->integ->LastStatus();
	if (!film.HasLDR()) return JP_ERR_DEVICE;
	if (rgb8_out) std::memcpy(rgb8_out, film.ldr8.data(), film.ldr8.size());
	if (film_out) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const FColor& c = static_cast<const FFilm&>(film)(x, y); float* o = film_out + 3 * ((size_t)y * W + x); o[0] = c.r; o[1] = c.g; o[2] = c.b; }
	if (filename && filename[0]) return film.SaveAsImage(filename, type == 0 ? EImageType::PPM : (type == 2 ? EImageType::HDR : EImageType::BMP)) ? JP_OK : JP_ERR_INVALID_ARGUMENT;
	return JP_OK;
}

// The reference's reflection classes by name (jetpbrt.h "reflection API"): builds the named class the way reference code would and
// calls Evalf / Pdf / Sample once.  which: 0 FPhongSpecularReflection(Ks, exponent), 1 FMicrofacetReflection(R, Beckmann(ax, ay, vis),
// FresnelNoOp), 2 FMicrofacetTransmission(T, TrowbridgeReitz(ax, ay, vis), etaA, etaB).  out: f[3], pdf, sample f[3], wi[3], pdf, flags
int jp_host_bsdf_class(int which, const float* color, float p0, float p1, int vis, float etaA, float etaB, const float* n, const float* wo, const float* wi, const float* u, float* out)
{
	FFrame frame(FVector3(n[0], n[1], n[2]));
	std::unique_ptr<FBSDF> b;
	const FColor c(color[0], color[1], color[2]);
	if (which == 0) b.reset(new FPhongSpecularReflection(frame, c, p0));
	else if (which == 1) b.reset(new FMicrofacetReflection(frame, c, new BeckmannDistribution(p0, p1, vis != 0), new FresnelNoOp()));
	else if (which == 2) b.reset(new FMicrofacetTransmission(frame, c, new TrowbridgeReitzDistribution(p0, p1, vis != 0), etaA, etaB));
	else return JP_ERR_INVALID_ARGUMENT;
	const FVector3 a(wo[0], wo[1], wo[2]), d(wi[0], wi[1], wi[2]);
	FColor f = b->Evalf(a, d); Float pdf = b->Pdf(a, d); FBSDFSample s = b->Sample(a, FVector2(u[0], u[1]));
	out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = pdf; out[4] = s.f.r; out[5] = s.f.g; out[6] = s.f.b; out[7] = s.wi.x; out[8] = s.wi.y; out[9] = s.wi.z; out[10] = s.pdf; out[11] = (float)s.ebsdf;
	return JP_OK;
}

// FGpuPathIntegrator::Render with one of the reference's other samplers: 0 FRandomSampler, 1 FStratifiedSampler, 2 FDebugSampler
int jp_host_render_sampler(void* h, int sampler, int W, int H, int spp, int maxdepth, int device, float* film_out)
{
	HostScene* hs = (HostScene*)h;
	FGpuPathIntegrator integ(maxdepth, device);
	FFilm film(W, H);
	std::unique_ptr<FSampler> s;
	if (sampler == 2) s.reset(new FDebugSampler(spp)); else if (sampler == 1) s.reset(new FStratifiedSampler(spp)); else s.reset(new FRandomSampler(spp));
	integ.Render(hs->scene.get(), s.get(), &film, 16);
	if (integ.LastStatus() != JP_OK) return integ.LastStatus();
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const FColor& c = static_cast<const FFilm&>(film)(x, y); float* o = film_out + 3 * ((size_t)y * W + x); o[0] = c.r; o[1] = c.g; o[2] = c.b; }
	return JP_OK;
}

// gamma_encoding (film.h:24) on the host for n values (tests: the device's bytes must equal these)
void jp_host_gamma_encode(const float* x, int n, unsigned char* out) { for (int i = 0; i < n; i++) out[i] = gamma_encoding(x[i]); }

// FFilm::SaveAsImage on an rgb buffer (type 0 PPM, 1 BMP, 2 HDR): returns 1 on success
int jp_host_save_image(const float* rgb, int W, int H, const char* filename, int type)
{
	FFilm film(W, H);
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const float* p = rgb + 3 * ((size_t)y * W + x); film(x, y) = FColor(p[0], p[1], p[2]); }
	return film.SaveAsImage(filename, type == 0 ? EImageType::PPM : (type == 2 ? EImageType::HDR : EImageType::BMP)) ? 1 : 0;
}

} // extern "C"
