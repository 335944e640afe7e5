// jet-pbrt_amd/host/scene.cc -- host scene description: bounds, shapes, materials, camera, FScene, OBJ ingest,
// and the flattener that produces the JpScene SoA view.  No ray is traced on the host.
#include "jetpbrt.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace jetpbrt
{
namespace
{
// libstdc++ std::min/std::max argument semantics, kept explicit (the reference's box maths is NaN-order sensitive)
inline Float smin(Float a, Float b) { return (b < a) ? b : a; }
inline Float smax(Float a, Float b) { return (a < b) ? b : a; }
inline FPoint3 Min3(const FPoint3& a, const FPoint3& b) { return FPoint3(smin(a.x, b.x), smin(a.y, b.y), smin(a.z, b.z)); }
inline FPoint3 Max3(const FPoint3& a, const FPoint3& b) { return FPoint3(smax(a.x, b.x), smax(a.y, b.y), smax(a.z, b.z)); }
}

// ---- FBounds3 (geometry.h:244-315) ---------------------------------------------------------------------------
FBounds3::FBounds3()
{
	const Float lo = std::numeric_limits<Float>::lowest(), hi = std::numeric_limits<Float>::max();
	_min = FPoint3(hi, hi, hi); _max = FPoint3(lo, lo, lo);
}
FBounds3::FBounds3(const FPoint3& p1, const FPoint3& p2) : _min(Min3(p1, p2)), _max(Max3(p1, p2)) {}
void FBounds3::Expand(const FBounds3& b) { _min = Min3(_min, b._min); _max = Max3(_max, b._max); }
FBounds3 FBounds3::Join(const FPoint3& p) const { return FBounds3(Min3(_min, p), Max3(_max, p)); }
void FBounds3::CheckThinness(Float t)
{
	if (_min.x == _max.x) { _min.x -= t; _max.x += t; }
	if (_min.y == _max.y) { _min.y -= t; _max.y += t; }
	if (_min.z == _max.z) { _min.z -= t; _max.z += t; }
}
void FBounds3::BoundingSphere(FPoint3& center, Float& radius) const      // geometry.h:307-311
{
	center = _min + (_max - _min) * (Float)0.5;                           // Lerp(u, v, t) = u + t * (v - u)
	bool inside = center.x >= _min.x && center.x <= _max.x && center.y >= _min.y && center.y <= _max.y && center.z >= _min.z && center.z <= _max.z;
	radius = inside ? (center - _max).Length() : (Float)0;
}

// ---- shapes ---------------------------------------------------------------------------------------------------
FTriangle::FTriangle(const FPoint3& a, const FPoint3& b, const FPoint3& c, bool flip_normal) : p0(a), p1(b), p2(c)
{
	normal = Normalize(Cross(p1 - p0, p2 - p0));                          // shape.h:284-286
	if (flip_normal) normal = -normal;
	FBounds3 bbox(p0, p1); bbox = bbox.Join(p2); tightBox = bbox; bbox.CheckThinness();    // shape.h:342-349
	worldBox = bbox;
}

FRectangle::FRectangle(const FPoint3& a, const FPoint3& b, const FPoint3& c, const FPoint3& d, bool flip_normal) : p0(a), p1(b), p2(c), p3(d)
{
	normal = Normalize(Cross(p1 - p0, p2 - p0));                          // shape.h:388-390
	if (flip_normal) normal = -normal;
	FBounds3 bbox = FBounds3(p0, p1).Join(p2).Join(p3); tightBox = bbox; bbox.CheckThinness();
	worldBox = bbox;
}
FRectangle FRectangle::FromXY(Float x0, Float x1, Float y0, Float y1, Float z, bool f)
{ return FRectangle(FPoint3(x0, y0, z), FPoint3(x1, y0, z), FPoint3(x1, y1, z), FPoint3(x0, y1, z), f); }   // shape.cc:76-81
FRectangle FRectangle::FromXZ(Float x0, Float x1, Float z0, Float z1, Float y, bool f)
{ return FRectangle(FPoint3(x0, y, z0), FPoint3(x0, y, z1), FPoint3(x1, y, z1), FPoint3(x1, y, z0), f); }   // shape.cc:83-88
FRectangle FRectangle::FromYZ(Float y0, Float y1, Float z0, Float z1, Float x, bool f)
{ return FRectangle(FPoint3(x, y0, z0), FPoint3(x, y1, z0), FPoint3(x, y1, z1), FPoint3(x, y0, z1), f); }   // shape.cc:90-95

FSphere::FSphere(const FVector3& c, Float r) : center(c), radius(r)
{
	FVector3 half(r, r, r);
	worldBox = FBounds3(center + half, center - half);                    // shape.h:540-544
	tightBox = worldBox;
}

FDisk::FDisk(const FPoint3& pos, const FVector3& nrm, Float r) : position(pos), normal(Normalize(nrm)), radius(r)
{
	// CalcWorldBounds shape.h:238-251: the square spanned by the frame's binormal / tangent (FFrame(normal) -> SetFromZ,
	// geometry.h:344-348, 372-377: n is normalised once more, t = Normalize(Cross(n, tmp_s)), s = Normalize(Cross(t, n)))
	const FVector3 n = Normalize(normal);
	const FVector3 tmp_s = (std::abs(n.x) > 0.99f) ? FVector3(0, 1, 0) : FVector3(1, 0, 0);
	const FVector3 t = Normalize(Cross(n, tmp_s));
	const FVector3 s = Normalize(Cross(t, n));
	const FVector3 rb = s * radius, rt = t * radius;
	FBounds3 bbox(position + rb + rt, position + rb + rt);
	bbox = bbox.Join(position + rb - rt);
	bbox = bbox.Join(position - rb - rt);
	bbox = bbox.Join(position - rb + rt);
	tightBox = bbox;
	bbox.CheckThinness();
	worldBox = bbox;
}

// ---- OBJ ingest: own reader, the reference's vertex transform (shape.cc:23-68) ------------------------------
bool LoadTriangleMesh(const char* filename, std::vector<std::shared_ptr<FTriangle>>& out, bool flip_normal, bool bFlipHandedness, const FVector3& offset, Float inScale)
{
	out.clear();
	std::ifstream file(filename);
	if (!file.is_open()) { fprintf(stderr, "load triangle mesh failed. %s\n", filename); return false; }
	std::vector<FVector3> pos;
	std::vector<int> face;
	std::string line;
	auto xform = [&](FVector3 v) {
		if (bFlipHandedness) v.z = -v.z;
		v = v * inScale;
		v = v + offset;
		return v;
	};
	while (std::getline(file, line))
	{
		const char* s = line.c_str();
		while (*s == ' ' || *s == '\t') s++;
		if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t'))
		{
			char* e = nullptr; const char* p = s + 1;
			float x = strtof(p, &e); p = e; float y = strtof(p, &e); p = e; float z = strtof(p, &e);
			pos.push_back(FVector3(x, y, z));
		}
		else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t'))
		{
			face.clear();
			const char* p = s + 1;
			for (;;)
			{
				while (*p == ' ' || *p == '\t') p++;
				if (*p == 0 || *p == '\r' || *p == '\n') break;
				char* e = nullptr; long idx = strtol(p, &e, 10);
				if (e == p) break;
				if (idx < 0) idx = (long)pos.size() + idx + 1;
				if (idx < 1 || idx > (long)pos.size()) { fprintf(stderr, "load triangle mesh failed. %s: face index out of range\n", filename); out.clear(); return false; }
				face.push_back((int)idx - 1);
				p = e; while (*p && *p != ' ' && *p != '\t') p++;             // skip /vt/vn
			}
			for (size_t k = 1; k + 1 < face.size(); k++)                      // triangles as given; polygons as a fan
			{
				FVector3 v0 = xform(pos[face[0]]), v1 = xform(pos[face[k]]), v2 = xform(pos[face[k + 1]]);
				out.push_back(std::make_shared<FTriangle>(v0, v1, v2, flip_normal));
			}
		}
	}
	return true;
}

// ---- materials --------------------------------------------------------------------------------------------------
Float RoughnessToAlpha(Float roughness)                                  // microfacet.h:85-90
{
	roughness = smax(roughness, (Float)1e-3);
	Float x = std::log(roughness);
	return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static void zero(float* o) { for (int i = 0; i < JP_MAT_PARAM_STRIDE; i++) o[i] = 0.f; }
void FMatteMaterial::Flatten(float* o) const { zero(o); o[0] = diffuseColor.r; o[1] = diffuseColor.g; o[2] = diffuseColor.b; }
void FMirrorMaterial::Flatten(float* o) const { zero(o); o[0] = specularColor.r; o[1] = specularColor.g; o[2] = specularColor.b; }
void FGlassMaterial::Flatten(float* o) const { zero(o); o[0] = eta; o[1] = Kr.r; o[2] = Kr.g; o[3] = Kr.b; o[4] = Kt.r; o[5] = Kt.g; o[6] = Kt.b; }
FPlasticMaterial::FPlasticMaterial(const FColor& kd, const FColor& ks, Float rough, bool remap) : Kd(kd), Ks(ks), roughness(rough), remapRoughness(remap)
{
	Float Ld = Kd.Luminance(), Ls = Ks.Luminance(), L = Ld + Ls;          // material.h:94-98
	Qd = Ld / L;
}
void FPlasticMaterial::Flatten(float* o) const
{
	zero(o); o[0] = Kd.r; o[1] = Kd.g; o[2] = Kd.b; o[3] = Ks.r; o[4] = Ks.g; o[5] = Ks.b;
	o[6] = remapRoughness ? RoughnessToAlpha(roughness) : roughness;       // material.cc:23-26
	o[7] = Qd;
}
void FMetalMaterial::Flatten(float* o) const
{
	zero(o); o[0] = eta.r; o[1] = eta.g; o[2] = eta.b; o[3] = k.r; o[4] = k.g; o[5] = k.b;
	o[6] = remapRoughness ? RoughnessToAlpha(uRoughness) : uRoughness;     // material.cc:33-38
	o[7] = remapRoughness ? RoughnessToAlpha(vRoughness) : vRoughness;
}

// ---- camera (camera.h:36-49) --------------------------------------------------------------------------------------
FCamera::FCamera(const FVector3& ipos, const FVector3& ifront, const FVector3& iup, Float ifov, const FVector2& ires)
	: pos(ipos), front(ifront.Normalize()), up(iup.Normalize()), resolution(ires)
{
	Float tan_fov = std::tan(((ifov * kPi) / (Float)180) / 2);
	Float aspect = resolution.x / resolution.y;
	right = up.Cross(front).Normalize() * (tan_fov * aspect);
	up = front.Cross(right).Normalize() * tan_fov;
}

// ---- lights ---------------------------------------------------------------------------------------------------------
void FEnvironmentLight::Preprocess(const FScene& scene)                  // light.cc:26-33
{
	FBounds3 bound = scene.WorldBound();
	bound.BoundingSphere(worldCenter, worldRadius);
}

void FDirectionLight::Preprocess(const FScene& scene)                    // light.cc:17-24
{
	FBounds3 bound = scene.WorldBound();
	bound.BoundingSphere(worldCenter, worldRadius);
}

// ---- scene (scene.cc) -------------------------------------------------------------------------------------------
std::vector<std::shared_ptr<FShape>> FScene::CreateTriangleMesh(const char* filename, bool flip_normal, bool bFlipHandedness, const FVector3& offset, Float inScale)
{
	std::vector<std::shared_ptr<FTriangle>> mesh;
	std::vector<std::shared_ptr<FShape>> newshapes;
	if (LoadTriangleMesh(filename, mesh, flip_normal, bFlipHandedness, offset, inScale))
		for (auto& t : mesh) { shapes.push_back(t); newshapes.push_back(t); }
	return newshapes;
}
std::vector<std::shared_ptr<FPrimitive>> FScene::CreatePrimitives(const std::vector<std::shared_ptr<FShape>>& inMesh, const std::shared_ptr<FMaterial>& inMaterial)
{
	std::vector<std::shared_ptr<FPrimitive>> r;
	for (auto& s : inMesh) r.push_back(CreatePrimitive(s.get(), inMaterial.get(), (const FAreaLight*)nullptr));
	return r;
}
std::vector<std::shared_ptr<FAreaLight>> FScene::CreateAreaLights(int samplesNum, const FColor& radiance, const std::vector<std::shared_ptr<FShape>>& inShapes, const std::shared_ptr<FMaterial>& inMaterial)
{
	std::vector<std::shared_ptr<FAreaLight>> r;                            // one light per shape (scene.cc:79-89)
	for (auto& s : inShapes) r.push_back(CreateAreaLight(samplesNum, radiance, s, inMaterial));
	return r;
}
std::shared_ptr<FAreaLight> FScene::CreateAreaLight(int samplesNum, const FColor& radiance, const std::shared_ptr<FShape>& inShape, const std::shared_ptr<FMaterial>& inMaterial)
{
	std::shared_ptr<FAreaLight> l = CreateLight<FAreaLight>(FPoint3(0, 0, 0), samplesNum, radiance, inShape.get());   // scene.cc:91-97
	CreatePrimitive(inShape.get(), inMaterial.get(), (const FAreaLight*)l.get());
	return l;
}

void FScene::Preprocess()                                                // scene.cc:11-23
{
	FBounds3 bound;
	for (auto& p : primitives) bound.Expand(p->shape->WorldBounds());     // scene.cc:35-45
	worldBound = bound;
	for (auto& l : lights) l->Preprocess(*this);
	if (const char* e = getenv("JETPBRT_REFERENCE_TREE")) { if (atoi(e) == 1 || atoi(e) == 2) referenceTree = true; if (atoi(e) == 2) certifiedWalk = true; }
	builtOnDevice = false;
	if (referenceTree)
	{
		std::vector<FBounds3> wb; wb.reserve(primitives.size());
		for (auto& p : primitives) wb.push_back(p->shape->WorldBounds());
		BuildReferenceBVH(wb, bvh);
		preprocessed = true;
		return;
	}
	// (the decision goes to builtOnDevice: deviceBuild / hostBuild stay what the caller set, so a later Preprocess() of a changed scene decides afresh)
	bool dev = deviceBuild || (!hostBuild && primitives.size() > kDeviceBuildFrom);
	if (const char* e = getenv("JETPBRT_DEVICE_BVH")) dev = atoi(e) == 1 ? true : (atoi(e) == 0 ? false : dev);
	builtOnDevice = dev;
	if (builtOnDevice) { bvh = FlatBVH(); preprocessed = true; return; }
	std::vector<FBounds3> pb; pb.reserve(primitives.size());
	// own BVH over the exact extents: a ray leaving a flat surface (min_t 0.001) or ending 0.001 short of a light
	// then misses that surface's box instead of visiting its leaf through the reference's 0.01 thinness pad
	for (auto& p : primitives) pb.push_back(p->shape->tightBox);
	int maxLeaf = 4;
	if (const char* e = getenv("JETPBRT_BVH_MAXLEAF")) { int v = atoi(e); if (v >= 1 && v <= 16) maxLeaf = v; }
	BuildBVH(pb, bvh, maxLeaf);
	preprocessed = true;
}

// ---- flattener ------------------------------------------------------------------------------------------------------
static void push3(std::vector<float>& v, const FVector3& p) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); }

bool FlattenScene(const FScene& scene, FlatScene& out, std::string* error)
{
	auto fail = [&](const char* msg) { if (error) *error = msg; return false; };
	if (!scene.preprocessed) return fail("FlattenScene: call FScene::Preprocess() first");
	if (!scene.camera) return fail("FlattenScene: scene has no camera");
	out = FlatScene();
	std::map<const FShape*, std::pair<int, int>> shapeRef;                // shape -> (kind, index); only referenced shapes are emitted
	std::map<const FMaterial*, int> matRef;
	std::map<const FLight*, int> lightRef;
	for (size_t i = 0; i < scene.materials.size(); i++)
	{
		matRef[scene.materials[i].get()] = (int)i;
		out.mat_type.push_back(scene.materials[i]->Kind());
		float p[JP_MAT_PARAM_STRIDE]; scene.materials[i]->Flatten(p);
		out.mat_params.insert(out.mat_params.end(), p, p + JP_MAT_PARAM_STRIDE);
	}
	for (size_t i = 0; i < scene.lights.size(); i++) lightRef[scene.lights[i].get()] = (int)i;

	std::map<const FShape*, int> shapePrim;                               // emitting shape -> primitive index
	for (size_t i = 0; i < scene.primitives.size(); i++)
	{
		const FPrimitive& P = *scene.primitives[i];
		if (!P.shape) return fail("FlattenScene: primitive without a shape");
		auto it = shapeRef.find(P.shape);
		if (it == shapeRef.end())
		{
			int kind = P.shape->Kind(), idx = 0;
			if (kind == JP_SHAPE_TRIANGLE) { const FTriangle* t = static_cast<const FTriangle*>(P.shape); idx = (int)out.tri_p0.size() / 3; push3(out.tri_p0, t->p0); push3(out.tri_p1, t->p1); push3(out.tri_p2, t->p2); push3(out.tri_n, t->normal); }
			else if (kind == JP_SHAPE_RECTANGLE) { const FRectangle* r = static_cast<const FRectangle*>(P.shape); idx = (int)out.rect_p0.size() / 3; push3(out.rect_p0, r->p0); push3(out.rect_p1, r->p1); push3(out.rect_p2, r->p2); push3(out.rect_p3, r->p3); push3(out.rect_n, r->normal); }
			else if (kind == JP_SHAPE_DISK) { const FDisk* k = static_cast<const FDisk*>(P.shape); idx = (int)out.disk_radius.size(); push3(out.disk_center, k->position); push3(out.disk_normal, k->normal); out.disk_radius.push_back(k->radius); }
			else { const FSphere* s = static_cast<const FSphere*>(P.shape); idx = (int)out.sph_radius.size(); push3(out.sph_center, s->center); out.sph_radius.push_back(s->radius); }
			it = shapeRef.insert(std::make_pair(P.shape, std::make_pair(kind, idx))).first;
		}
		out.prim_shape_type.push_back(it->second.first);
		out.prim_shape_index.push_back(it->second.second);
		int m = -1; if (P.material) { auto mi = matRef.find(P.material); if (mi == matRef.end()) return fail("FlattenScene: primitive material not created through this scene"); m = mi->second; }
		out.prim_material.push_back(m);
		int l = -1; if (P.arealight) { auto li = lightRef.find(P.arealight); if (li == lightRef.end()) return fail("FlattenScene: primitive light not created through this scene"); l = li->second; shapePrim[P.shape] = (int)i; }
		out.prim_light.push_back(l);
	}
	float worldRadius = 0.f;
	for (size_t i = 0; i < scene.lights.size(); i++)
	{
		const FLight* L = scene.lights[i].get();
		out.light_type.push_back(L->Kind());
		FColor rad; FVector3 vec(0, 0, 0); int prim = -1;
		if (L->Kind() == JP_LIGHT_AREA)
		{
			const FAreaLight* a = static_cast<const FAreaLight*>(L);
			rad = a->radiance;
			auto sp = shapePrim.find(a->shape);
			if (sp == shapePrim.end()) return fail("FlattenScene: area light whose shape has no primitive");
			prim = sp->second;
		}
		else if (L->Kind() == JP_LIGHT_POINT) { const FPointLight* q = static_cast<const FPointLight*>(L); rad = q->intensity; vec = q->worldPosition; }
		else if (L->Kind() == JP_LIGHT_DIRECTION) { const FDirectionLight* q = static_cast<const FDirectionLight*>(L); rad = q->irradiance; vec = q->worldDir; worldRadius = q->worldRadius; }
		else { const FEnvironmentLight* e = static_cast<const FEnvironmentLight*>(L); rad = e->radiance; worldRadius = e->worldRadius; }
		out.light_radiance.push_back(rad.r); out.light_radiance.push_back(rad.g); out.light_radiance.push_back(rad.b);
		out.light_vec.push_back(vec.x); out.light_vec.push_back(vec.y); out.light_vec.push_back(vec.z);
		out.light_prim.push_back(prim);
	}
	out.bvh = scene.bvh;

	JpScene& v = out.view; std::memset(&v, 0, sizeof(v));
	const FCamera& c = *scene.camera;
	const float cam[14] = { c.pos.x, c.pos.y, c.pos.z, c.front.x, c.front.y, c.front.z, c.right.x, c.right.y, c.right.z, c.up.x, c.up.y, c.up.z, c.resolution.x, c.resolution.y };
	std::memcpy(v.camera.pos, cam, 3 * sizeof(float)); std::memcpy(v.camera.front, cam + 3, 3 * sizeof(float));
	std::memcpy(v.camera.right, cam + 6, 3 * sizeof(float)); std::memcpy(v.camera.up, cam + 9, 3 * sizeof(float));
	v.camera.res_x = cam[12]; v.camera.res_y = cam[13];
	v.n_triangles = (int)out.tri_p0.size() / 3; v.tri_p0 = out.tri_p0.data(); v.tri_p1 = out.tri_p1.data(); v.tri_p2 = out.tri_p2.data(); v.tri_n = out.tri_n.data();
	v.n_rectangles = (int)out.rect_p0.size() / 3; v.rect_p0 = out.rect_p0.data(); v.rect_p1 = out.rect_p1.data(); v.rect_p2 = out.rect_p2.data(); v.rect_p3 = out.rect_p3.data(); v.rect_n = out.rect_n.data();
	v.n_spheres = (int)out.sph_radius.size(); v.sph_center = out.sph_center.data(); v.sph_radius = out.sph_radius.data();
	v.n_disks = (int)out.disk_radius.size(); v.disk_center = out.disk_center.data(); v.disk_normal = out.disk_normal.data(); v.disk_radius = out.disk_radius.data();
	v.n_primitives = (int)out.prim_shape_type.size(); v.prim_shape_type = out.prim_shape_type.data(); v.prim_shape_index = out.prim_shape_index.data();
	v.prim_material = out.prim_material.data(); v.prim_light = out.prim_light.data();
	v.n_materials = (int)out.mat_type.size(); v.mat_type = out.mat_type.data(); v.mat_params = out.mat_params.data();
	v.n_lights = (int)out.light_type.size(); v.light_type = out.light_type.data(); v.light_radiance = out.light_radiance.data(); v.light_prim = out.light_prim.data(); v.light_vec = out.light_vec.data();
	v.world_radius = worldRadius;
	v.bvh_reference_semantics = scene.referenceTree ? (scene.certifiedWalk ? 2 : 1) : 0;
	v.n_bvh_nodes = (int)out.bvh.left.size();                    // 0: hierarchy to be built on the device
	v.bvh_bounds = out.bvh.bounds.data(); v.bvh_left = out.bvh.left.data(); v.bvh_right = out.bvh.right.data();
	v.n_bvh_prim_indices = (int)out.bvh.prim_index.size(); v.bvh_prim_index = out.bvh.prim_index.data();
	return true;
}

} // namespace jetpbrt
