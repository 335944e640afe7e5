// jet-pbrt_amd/host/cli_main.cc -- the reference's command line (main.cc:113-163) on the GPU integrator:
//     jetpbrt sceneid spp [width height] [--assets DIR] [--out NAME] [--format bmp|ppm|hdr] [--device-bvh | --reference-tree | --reference-tree=certified] [--integrator path|recursive|whitted|debug]
// sceneid 0 = Cornell box, 1 = bunny scene; spp defaults to 50, the film to 1024 x 1024, the output to
// <scene name>_<spp>.bmp, as in the reference.  The scene scripts are the calls of main.cc:13-111; the meshes are
// read from DIR/cornellbox/{light,floor,shortbox,tallbox,left,right}.obj and DIR/bunny/bunny.obj (the reference
// reads scene\cornellbox\... relative to the working directory; its assets are not in its repository --
// `python -m jet_pbrt_amd.scenes DIR` writes the synthetic stand-ins).
#include "jetpbrt.h"

#include <cstdlib>
#include <cstring>

using namespace jetpbrt;

static FColor LightRadiance()                                    // main.cc:35
{
	auto V = [](Float a, Float b, Float c) { return FVector3(a, b, c); };
	FVector3 r = V(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) * 8.0f + V(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) * 15.6f + V(0.737f + 0.642f, 0.737f + 0.159f, 0.737f) * 18.4f;
	return FColor(r.x, r.y, r.z);
}

static std::shared_ptr<FScene> create_cornellbox_scene(const FVector2& filmsize, const std::string& dir)   // main.cc:13-62
{
	const FPoint3 lookfrom(278, 273, 960), lookat(278, 273, 0);
	std::shared_ptr<FScene> scene = std::make_shared<FScene>("cornell_box_scene");
	scene->CreateCamera<FCamera>(lookfrom, Normalize(lookat - lookfrom), FVector3(0, 1, 0), (Float)60.0, filmsize);
	scene->CreateLight<FEnvironmentLight>(FPoint3(0, 0, 0), 1, FColor(0.f, 0.f, 0.f));
	std::shared_ptr<FMaterial> red = scene->CreateMaterial<FMatteMaterial>(FColor(0.63f, 0.065f, 0.05f));
	std::shared_ptr<FMaterial> green = scene->CreateMaterial<FMatteMaterial>(FColor(0.14f, 0.45f, 0.091f));
	std::shared_ptr<FMaterial> white = scene->CreateMaterial<FMatteMaterial>(FColor(0.725f, 0.71f, 0.68f));
	std::shared_ptr<FMaterial> golden = scene->CreateMaterial<FMetalMaterial>(FColor(0.18f, 0.15f, 0.81f), FColor(0.11f, 0.11f, 0.11f), 0.2f, 0.2f, false);
	std::shared_ptr<FMaterial> mat_light = scene->CreateMaterial<FMatteMaterial>(FColor(0.65f, 0.65f, 0.65f));
	const std::string d = dir + "/cornellbox/";
	scene->CreateAreaLights(1, LightRadiance(), scene->CreateTriangleMesh((d + "light.obj").c_str(), true, true), mat_light);
	scene->CreatePrimitives(scene->CreateTriangleMesh((d + "floor.obj").c_str(), true, true), white);
	scene->CreatePrimitives(scene->CreateTriangleMesh((d + "shortbox.obj").c_str(), true, true), white);
	scene->CreatePrimitives(scene->CreateTriangleMesh((d + "tallbox.obj").c_str(), true, true), golden);
	scene->CreatePrimitives(scene->CreateTriangleMesh((d + "left.obj").c_str(), true, true), red);
	scene->CreatePrimitives(scene->CreateTriangleMesh((d + "right.obj").c_str(), true, true), green);
	scene->Preprocess();
	return scene;
}

static std::shared_ptr<FScene> create_bunny_scene(const FVector2& filmsize, const std::string& dir)        // main.cc:64-111
{
	const FPoint3 lookfrom(-300, 300, -300), lookat(0, 0, 0);
	std::shared_ptr<FScene> scene = std::make_shared<FScene>("bunny_scene");
	scene->CreateCamera<FCamera>(lookfrom, Normalize(lookat - lookfrom), FVector3(0, 1, 0), (Float)60.0, filmsize);
	scene->CreateLight<FEnvironmentLight>(FPoint3(0, 0, 0), 1, FColor(0.1f, 0.1f, 0.5f));
	std::shared_ptr<FMaterial> red = scene->CreateMaterial<FMatteMaterial>(FColor(0.63f, 0.065f, 0.05f));
	std::shared_ptr<FMaterial> green = scene->CreateMaterial<FMatteMaterial>(FColor(0.14f, 0.45f, 0.091f));
	scene->CreateMaterial<FMatteMaterial>(FColor(0.725f, 0.71f, 0.68f));
	std::shared_ptr<FMaterial> mat_light = scene->CreateMaterial<FMatteMaterial>(FColor(0.65f, 0.65f, 0.65f));
	std::shared_ptr<FShape> light = scene->CreateShape<FRectangle>(FRectangle::FromXZ(-100, 100, -100, 100, 350, true));
	scene->CreateAreaLight(1, LightRadiance(), light, mat_light);
	std::shared_ptr<FShape> floor = scene->CreateShape<FRectangle>(FRectangle::FromXZ(-200, 200, -200, 200, 0));
	scene->CreatePrimitive(floor.get(), green.get(), (const FAreaLight*)nullptr);
	const std::string obj = dir + "/bunny/bunny.obj";
	scene->CreatePrimitives(scene->CreateTriangleMesh(obj.c_str(), true, true, FVector3(0, 0, 0), 500.f), red);
	std::shared_ptr<FMaterial> plastic = scene->CreateMaterial<FPlasticMaterial>(FColor(0.35f, 0.12f, 0.48f), FColor(1) - FColor(0.35f, 0.12f, 0.48f), 0.1f, false);
	scene->CreatePrimitives(scene->CreateTriangleMesh(obj.c_str(), true, true, FVector3(-100, 0, -100), 500.f), plastic);
	std::shared_ptr<FMaterial> golden = scene->CreateMaterial<FMetalMaterial>(FColor(0.18f, 0.15f, 0.81f), FColor(0.11f, 0.11f, 0.11f), 0.2f, 0.2f, false);
	scene->CreatePrimitives(scene->CreateTriangleMesh(obj.c_str(), true, true, FVector3(0, 0, -100), 500.f), golden);
	std::shared_ptr<FMaterial> glass = scene->CreateMaterial<FGlassMaterial>(1.5f, FColor(0.98f), FColor(0.98f));
	scene->CreatePrimitives(scene->CreateTriangleMesh(obj.c_str(), true, true, FVector3(-100, 0, 0), 500.f), glass);
	scene->Preprocess();
	return scene;
}

int main(int argc, char* argv[])
{
	int width = 1024, height = 1024, samples_per_pixel = 50;     // main.cc:115,119
	std::string assets = "scene", out, format = "bmp", integratorName = "path";
	fprintf(stderr, "pbrt.exe  sceneid   spp\n");                 // main.cc:121
	std::vector<const char*> pos;
	for (int i = 1; i < argc; i++)
	{
		if (!strcmp(argv[i], "--assets") && i + 1 < argc) assets = argv[++i];
		else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
		else if (!strcmp(argv[i], "--format") && i + 1 < argc) format = argv[++i];
		else if (!strcmp(argv[i], "--device-bvh")) setenv("JETPBRT_DEVICE_BVH", "1", 1);   // FScene::deviceBuild for the scenes created below
		else if (!strcmp(argv[i], "--integrator") && i + 1 < argc) integratorName = argv[++i];
		else if (!strcmp(argv[i], "--reference-tree")) setenv("JETPBRT_REFERENCE_TREE", "1", 1);   // FScene::referenceTree: the reference's own BVH and traversal semantics
		else if (!strcmp(argv[i], "--reference-tree=certified")) setenv("JETPBRT_REFERENCE_TREE", "2", 1);   // ... with the certified walk (FScene::certifiedWalk)
		else pos.push_back(argv[i]);
	}
	if (pos.empty()) return 0;                                    // main.cc:122-125
	const int sceneId = atoi(pos[0]);
	if (pos.size() > 1) { int spp = atoi(pos[1]); if (spp > 0) samples_per_pixel = spp; }
	if (pos.size() > 3) { int w = atoi(pos[2]), h = atoi(pos[3]); if (w > 0 && h > 0) { width = w; height = h; } }
	FFilm film(width, height);
	std::shared_ptr<FScene> scene;
	switch (sceneId)
	{
	case 0: scene = create_cornellbox_scene(film.GetResolution(), assets); break;
	case 1: scene = create_bunny_scene(film.GetResolution(), assets); break;
	default: return 0;
	}
	fprintf(stderr, "current scene: %s\n", scene->NameStr());
	if (scene->primitives.empty()) { fprintf(stderr, "no geometry loaded from %s\n", assets.c_str()); return 2; }
	std::shared_ptr<FSampler> sampler = std::make_shared<FRandomSampler>(samples_per_pixel);
	// main.cc:154 constructs FPathIntegratorIteration(5); the commented-out alternatives of main.cc:150-153 are selectable here
	std::unique_ptr<FGpuPathIntegrator> integrator;
	if (integratorName == "whitted") integrator.reset(new FWhittedIntegrator(5));
	else if (integratorName == "debug") integrator.reset(new FDebugIntegrator());
	else if (integratorName == "recursive") integrator.reset(new FPathIntegratorRecursive(5));
	else integrator.reset(new FPathIntegratorIteration(5));
	if (format != "hdr") film.RequestDeviceLDR(true);             // BMP / PPM: gamma_encoding runs on the GPU, 3 bytes per pixel come back
	integrator->Render(scene.get(), sampler.get(), &film, 16);    // main.cc:156
	if (integrator->LastStatus() != JP_OK) return 3;              // no GPU / no library: fail loudly, nothing is written
	char fullname[512];
	snprintf(fullname, sizeof(fullname), "%s_%d", scene->NameStr(), samples_per_pixel);
	const std::string name = out.empty() ? fullname : out;
	const EImageType t = format == "ppm" ? EImageType::PPM : (format == "hdr" ? EImageType::HDR : EImageType::BMP);
	return film.SaveAsImage(name, t) ? 0 : 4;                     // main.cc:160
}
