// jet-pbrt_amd/host/integrator.cc -- FGpuPathIntegrator::Render: the replacement for the reference's
// FIntegrator::Render (integrator.cc:35-80).  Instead of cutting the film into 20-row FRenderTasks for a
// std::thread pool (integrator.cc:53-74, parallel.cc), it flattens the scene once and calls the HIP library
// through the C ABI of include/jetpbrt_amd.h.  The HIP library is bound with dlopen so this host library has no
// link-time GPU dependency; if it is missing Render() fails loudly -- there is no CPU fallback.
#include "jetpbrt.h"

#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace jetpbrt
{
namespace
{
struct HipApi
{
	void* lib = nullptr;
	const char* (*last_error)() = nullptr;
	int (*create_context)(int, JpContext**) = nullptr;
	int (*destroy_context)(JpContext*) = nullptr;
	int (*upload_scene)(JpContext*, const JpScene*) = nullptr;
	int (*render)(JpContext*, const JpRenderParams*, float*) = nullptr;
	int (*render_rgb8)(JpContext*, const JpRenderParams*, uint8_t*, float*) = nullptr;
	int (*bsdf)(JpContext*, const JpBsdfDesc*, int32_t, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*, int32_t*) = nullptr;
	int (*get_counters)(JpContext*, JpCounters*) = nullptr;
	int (*abi_version)() = nullptr;
	int (*set_options)(JpContext*, const JpOptions*) = nullptr;
	std::string error;
};

HipApi& Api()
{
	static HipApi api; static std::once_flag once;
	std::call_once(once, []() {
		// libjetpbrt_amd.so lives next to this library's directory: <pkg>/csrc/libjetpbrt_amd.so
		std::string dir;
		Dl_info info;
		if (dladdr((void*)&Api, &info) && info.dli_fname) { dir = info.dli_fname; size_t p = dir.find_last_of('/'); dir = p == std::string::npos ? "." : dir.substr(0, p); }
		const char* env = getenv("JETPBRT_AMD_LIB");
		std::string cands[3] = { env ? env : "", dir + "/../csrc/libjetpbrt_amd.so", "libjetpbrt_amd.so" };
		for (auto& c : cands) { if (c.empty()) continue; api.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL); if (api.lib) break; api.error = dlerror(); }
		if (!api.lib) return;
		api.last_error = (const char* (*)())dlsym(api.lib, "jp_last_error");
		api.create_context = (int (*)(int, JpContext**))dlsym(api.lib, "jp_create_context");
		api.destroy_context = (int (*)(JpContext*))dlsym(api.lib, "jp_destroy_context");
		api.upload_scene = (int (*)(JpContext*, const JpScene*))dlsym(api.lib, "jp_upload_scene");
		api.render = (int (*)(JpContext*, const JpRenderParams*, float*))dlsym(api.lib, "jp_render");
		api.get_counters = (int (*)(JpContext*, JpCounters*))dlsym(api.lib, "jp_get_counters");
		api.render_rgb8 = (int (*)(JpContext*, const JpRenderParams*, uint8_t*, float*))dlsym(api.lib, "jp_render_rgb8");
		api.bsdf = (decltype(api.bsdf))dlsym(api.lib, "jp_bsdf");
		api.abi_version = (int (*)())dlsym(api.lib, "jp_abi_version");
		api.set_options = (int (*)(JpContext*, const JpOptions*))dlsym(api.lib, "jp_set_options");
		if (!api.last_error || !api.create_context || !api.destroy_context || !api.upload_scene || !api.render || !api.get_counters || !api.render_rgb8 || !api.bsdf || !api.abi_version || !api.set_options)
		{ api.error = "libjetpbrt_amd.so lacks a required jp_* symbol"; dlclose(api.lib); api.lib = nullptr; }
		else if (api.abi_version() != JP_ABI_VERSION)            // a stale build would be handed structs of another size (JpCounters, JpBuildInfo, JpOptions)
		{ api.error = "libjetpbrt_amd.so implements ABI " + std::to_string(api.abi_version()) + ", this host library was built for ABI " + std::to_string(JP_ABI_VERSION); dlclose(api.lib); api.lib = nullptr; }
	});
	return api;
}
}

// ---- reflection API: one shading event on the device (jp_bsdf), on a context of this process created on first use ----
namespace
{
struct BsdfOut { float f[3], pdf, sf[3], swi[3], spdf; int32_t flags; bool ok; };
BsdfOut DeviceBsdf(const JpBsdfDesc& d, const FVector3& n, const FVector3& wo, const FVector3& wi, const FVector2& u)
{
	static JpContext* bctx = nullptr; static std::mutex mu;
	// the context lives on the device this process renders on (JETPBRT_DEVICE, else LOCAL_RANK, else 0 -- under HIP_VISIBLE_DEVICES
	// that is the rank's own GPU) and is destroyed at exit
	BsdfOut o; std::memset(&o, 0, sizeof(o));
	HipApi& api = Api();
	if (!api.lib) { fprintf(stderr, "FBSDF: HIP library not available (%s)\n", api.error.c_str()); return o; }
	std::lock_guard<std::mutex> lock(mu);
	if (!bctx)
	{
		int dev = 0;
		if (const char* e = getenv("JETPBRT_DEVICE")) dev = atoi(e); else if (const char* r = getenv("LOCAL_RANK")) dev = atoi(r);
		if (api.create_context(dev, &bctx) != JP_OK && (dev == 0 || api.create_context(0, &bctx) != JP_OK)) { fprintf(stderr, "FBSDF: %s\n", api.last_error()); bctx = nullptr; return o; }
		atexit([]() { if (bctx) { Api().destroy_context(bctx); bctx = nullptr; } });
	}
	const float nn[3] = { n.x, n.y, n.z }, a[3] = { wo.x, wo.y, wo.z }, b[3] = { wi.x, wi.y, wi.z }, uu[2] = { u.x, u.y };
	if (api.bsdf(bctx, &d, 1, nn, a, b, uu, o.f, &o.pdf, o.sf, o.swi, &o.spdf, &o.flags) != JP_OK) { fprintf(stderr, "FBSDF: %s\n", api.last_error()); return o; }
	o.ok = true;
	return o;
}
}
FColor FBSDF::Evalf(const FVector3& wo, const FVector3& wi) const { BsdfOut o = DeviceBsdf(desc, normal, wo, wi, FVector2(0.5f, 0.5f)); return FColor(o.f[0], o.f[1], o.f[2]); }
Float FBSDF::Pdf(const FVector3& wo, const FVector3& wi) const { return DeviceBsdf(desc, normal, wo, wi, FVector2(0.5f, 0.5f)).pdf; }
FBSDFSample FBSDF::Sample(const FVector3& wo, const FVector2& random) const
{
	BsdfOut o = DeviceBsdf(desc, normal, wo, wo, random);
	FBSDFSample s; s.f = FColor(o.sf[0], o.sf[1], o.sf[2]); s.wi = FVector3(o.swi[0], o.swi[1], o.swi[2]); s.pdf = o.spdf; s.ebsdf = o.flags;
	return s;
}

FGpuPathIntegrator::FGpuPathIntegrator(int maxDepth, int deviceId) : maxDepth(maxDepth), deviceId(deviceId) { std::memset(&counters, 0, sizeof(counters)); }

FGpuPathIntegrator::~FGpuPathIntegrator()
{
	if (ctx && Api().lib) Api().destroy_context(ctx);
}

void FGpuPathIntegrator::Render(const FScene* scene, FSampler* sampler, FFilm* film, int /*numthreads*/) const
{
	fprintf(stderr, "start rendering ...\n");                             // integrator.cc:44
	HipApi& api = Api();
	if (!api.lib) { fprintf(stderr, "FGpuPathIntegrator::Render: HIP library not available (%s); nothing rendered\n", api.error.c_str()); lastStatus = JP_ERR_NO_DEVICE; return; }
	if (!scene || !sampler || !film) { fprintf(stderr, "FGpuPathIntegrator::Render: null argument\n"); lastStatus = JP_ERR_INVALID_ARGUMENT; return; }
	if (!ctx) { lastStatus = api.create_context(deviceId, &ctx); if (lastStatus != JP_OK) { fprintf(stderr, "FGpuPathIntegrator::Render: %s\n", api.last_error()); ctx = nullptr; return; } }
	if (optionsDirty) { lastStatus = api.set_options(ctx, &options); if (lastStatus != JP_OK) { fprintf(stderr, "FGpuPathIntegrator::Render: %s\n", api.last_error()); return; } optionsDirty = false; }
	if (uploaded != scene)
	{
		FlatScene flat; std::string err;
		if (!FlattenScene(*scene, flat, &err)) { fprintf(stderr, "FGpuPathIntegrator::Render: %s\n", err.c_str()); lastStatus = JP_ERR_INVALID_ARGUMENT; return; }
		lastStatus = api.upload_scene(ctx, &flat.view);
		if (lastStatus == JP_ERR_UNSUPPORTED && flat.view.n_bvh_nodes == 0)
		{   // FScene::deviceBuild, but the device-built tree was refused (deeper than the traversal stack): build the SAH tree
			// on the host for this upload -- a different hierarchy, the same GPU path
			fprintf(stderr, "FGpuPathIntegrator::Render: %s; building the hierarchy on the host instead\n", api.last_error());
			std::vector<FBounds3> pb; pb.reserve(scene->primitives.size());
			for (auto& p : scene->primitives) pb.push_back(p->shape->tightBox);
			BuildBVH(pb, flat.bvh, 4);
			flat.view.n_bvh_nodes = (int)flat.bvh.left.size(); flat.view.bvh_bounds = flat.bvh.bounds.data(); flat.view.bvh_left = flat.bvh.left.data(); flat.view.bvh_right = flat.bvh.right.data();
			flat.view.n_bvh_prim_indices = (int)flat.bvh.prim_index.size(); flat.view.bvh_prim_index = flat.bvh.prim_index.data();
			lastStatus = api.upload_scene(ctx, &flat.view);
		}
		if (lastStatus != JP_OK) { fprintf(stderr, "FGpuPathIntegrator::Render: %s\n", api.last_error()); return; }
		uploaded = scene;
	}
	JpRenderParams rp; std::memset(&rp, 0, sizeof(rp));
	rp.width = film->Width(); rp.height = film->Height();
	rp.spp = sampler->GetSamplesPerPixel(); rp.max_depth = maxDepth;
	rp.sampler_mode = sampler->Mode(); rp.seed = sampler->Seed();
	rp.band_rows = bandRows; rp.shard_index = shardIndex; rp.shard_count = shardCount; rp.integrator = kind;
	const bool floatFilm = !(film->wantLDR && film->ldrOnly);
	std::vector<float> rgb(floatFilm ? (size_t)rp.width * rp.height * 3 : 0);
	std::vector<uint8_t> ldr;
	if (film->wantLDR)
	{   // FFilm::RequestDeviceLDR: gamma_encoding (film.h:24) runs on the GPU, the film comes back as 3 bytes per pixel
		ldr.assign((size_t)rp.width * rp.height * 3, 0);
		lastStatus = api.render_rgb8(ctx, &rp, ldr.data(), floatFilm ? rgb.data() : nullptr);
	}
	else lastStatus = api.render(ctx, &rp, rgb.data());
	if (lastStatus != JP_OK) { fprintf(stderr, "FGpuPathIntegrator::Render: %s\n", api.last_error()); return; }
	if (floatFilm)
		for (int y = 0; y < rp.height; y++) for (int x = 0; x < rp.width; x++)
		{
			const float* p = &rgb[3 * ((size_t)y * rp.width + x)];
			film->AddColor(x, y, FColor(p[0], p[1], p[2]));               // integrator.cc:108 / film.h:64-68
		}
	if (film->wantLDR) { film->ldr8.swap(ldr); film->floatValid = floatFilm; }   // (after AddColor: any later change of a pixel drops the bytes again)
	api.get_counters(ctx, &counters);
	fprintf(stderr, "finish rendering ...\n");
	fprintf(stderr, "FIntegrator::Render used %f seconds.\n", (float)(counters.render_ms / 1000.0));   // integrator.cc:77-79
}

} // namespace jetpbrt
