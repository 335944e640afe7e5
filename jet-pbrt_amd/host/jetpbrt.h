// jet-pbrt_amd/host/jetpbrt.h -- host-side mirror of the reference's Scene/Camera/Film/Sampler/Material/
// Integrator API vocabulary (SURVEY.md section 7 "design stance"), written from scratch for the GPU path:
// objects here only DESCRIBE a scene; nothing in this layer traces a ray.  FScene::Preprocess() builds the
// BVH, FlattenScene() turns the preprocessed scene into the SoA arrays of include/jetpbrt_amd.h, and
// FGpuPathIntegrator::Render() -- same signature as FIntegrator::Render (integrator.h:32) -- hands them to
// the HIP library through the C ABI.
//
// Names follow the reference so that code written against it (main.cc:13-111) ports by changing the
// namespace.  Unlike the reference objects (whose parameters are protected), every
// object exposes the state the flattener needs.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "jetpbrt_amd.h"

namespace jetpbrt
{
typedef float Float;                                             // pbrt.h:27
constexpr Float kPi = (Float)3.14159265358979323846;             // pbrt.h:39

struct FColor                                                    // color.h:13-69 (data + what the host needs)
{
	Float r, g, b;
	FColor() : r(0), g(0), b(0) {}
	FColor(Float v) : r(v), g(v), b(v) {}
	FColor(Float rr, Float gg, Float bb) : r(rr), g(gg), b(bb) {}
	FColor operator-(const FColor& c) const { return FColor(r - c.r, g - c.g, b - c.b); }
	Float Luminance() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }   // color.h:45-48
};

struct FVector2 { Float x, y; FVector2() : x(0), y(0) {} FVector2(Float vx, Float vy) : x(vx), y(vy) {} };
typedef FVector2 FPoint2;

struct FVector3                                                  // geometry.h:65-159
{
	Float x, y, z;
	FVector3() : x(0), y(0), z(0) {}
	FVector3(Float vx, Float vy, Float vz) : x(vx), y(vy), z(vz) {}
	FVector3 operator-() const { return FVector3(-x, -y, -z); }
	FVector3 operator+(const FVector3& v) const { return FVector3(x + v.x, y + v.y, z + v.z); }
	FVector3 operator-(const FVector3& v) const { return FVector3(x - v.x, y - v.y, z - v.z); }
	FVector3 operator*(Float s) const { return FVector3(x * s, y * s, z * s); }
	FVector3 operator/(Float s) const { return FVector3(x / s, y / s, z / s); }
	Float Length2() const { return x * x + y * y + z * z; }
	Float Length() const { return std::sqrt(Length2()); }
	FVector3 Normalize() const { return *this / Length(); }
	FVector3 Cross(const FVector3& v) const { return FVector3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }
	Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
typedef FVector3 FPoint3;
typedef FVector3 FNormal3;
inline FVector3 Normalize(const FVector3& v) { return v.Normalize(); }
inline FVector3 Cross(const FVector3& a, const FVector3& b) { return a.Cross(b); }

// axis-aligned box with the reference's semantics (geometry.h:244-315): needed bit-for-bit because the
// environment light's worldRadius (light.cc:26-33) is derived from the union of the shape bounds.
struct FBounds3
{
	FPoint3 _min, _max;
	FBounds3();
	FBounds3(const FPoint3& p1, const FPoint3& p2);
	void Expand(const FBounds3& b);
	FBounds3 Join(const FPoint3& p) const;
	void CheckThinness(Float thinness = 0.01f);
	void BoundingSphere(FPoint3& center, Float& radius) const;
};

// ---- shapes (geometry only; intersection lives on the device) -----------------------------------------------
class FShape
{
public:
	virtual ~FShape() = default;
	virtual int Kind() const = 0;                                // JP_SHAPE_*
	const FBounds3& WorldBounds() const { return worldBox; }
	FBounds3 worldBox;          // the reference's bounds (thin boxes padded by 0.01, geometry.h:299-304): world bound / env radius
	FBounds3 tightBox;          // exact extent: what our own BVH is built over (padding happens at upload, relative)
};

class FTriangle : public FShape                                  // shape.h:277-369
{
public:
	FTriangle(const FPoint3& p0, const FPoint3& p1, const FPoint3& p2, bool flip_normal = false);
	int Kind() const override { return JP_SHAPE_TRIANGLE; }
	FPoint3 p0, p1, p2; FNormal3 normal;
};

class FRectangle : public FShape                                 // shape.h:380-472, shape.cc:76-95
{
public:
	FRectangle(const FPoint3& p0, const FPoint3& p1, const FPoint3& p2, const FPoint3& p3, bool flip_normal = false);
	static FRectangle FromXY(Float x0, Float x1, Float y0, Float y1, Float z, bool flip_normal = false);
	static FRectangle FromXZ(Float x0, Float x1, Float z0, Float z1, Float y, bool flip_normal = false);
	static FRectangle FromYZ(Float y0, Float y1, Float z0, Float z1, Float x, bool flip_normal = false);
	int Kind() const override { return JP_SHAPE_RECTANGLE; }
	FPoint3 p0, p1, p2, p3; FNormal3 normal;
};

class FSphere : public FShape                                    // shape.h:476-662
{
public:
	FSphere(const FVector3& center, Float r);
	int Kind() const override { return JP_SHAPE_SPHERE; }
	FVector3 center; Float radius;
};

class FDisk : public FShape                                      // shape.h:189-275
{
public:
	FDisk(const FPoint3& pos, const FVector3& normal, Float radius);
	int Kind() const override { return JP_SHAPE_DISK; }
	FPoint3 position; FVector3 normal; Float radius;             // normal normalised by the constructor (shape.h:194)
};

// triangulated OBJ ingest with the reference's transform order (shape.cc:23-68): z flip, scale, offset
bool LoadTriangleMesh(const char* filename, std::vector<std::shared_ptr<FTriangle>>& outTriangles, bool flip_normal = false,
                      bool bFlipHandedness = false, const FVector3& offset = FVector3(0, 0, 0), Float inScale = 1.f);

// ---- materials: parameter holders with a flatten hook -------------------------------------------------------
class FMaterial
{
public:
	virtual ~FMaterial() {}
	virtual int Kind() const = 0;                                // JP_MAT_*
	virtual void Flatten(float out[JP_MAT_PARAM_STRIDE]) const = 0;
};
Float RoughnessToAlpha(Float roughness);                         // microfacet.h:85-90

class FMatteMaterial : public FMaterial                          // material.h:27-41
{ public: FMatteMaterial(const FColor& c) : diffuseColor(c) {} int Kind() const override { return JP_MAT_MATTE; } void Flatten(float*) const override; FColor diffuseColor; };
class FMirrorMaterial : public FMaterial                         // material.h:45-59
{ public: FMirrorMaterial(const FColor& c) : specularColor(c) {} int Kind() const override { return JP_MAT_MIRROR; } void Flatten(float*) const override; FColor specularColor; };
class FGlassMaterial : public FMaterial                          // material.h:63-81
{ public: FGlassMaterial(Float eta, const FColor& kr = FColor(1, 1, 1), const FColor& kt = FColor(1, 1, 1)) : eta(eta), Kr(kr), Kt(kt) {}
  int Kind() const override { return JP_MAT_GLASS; } void Flatten(float*) const override; Float eta; FColor Kr, Kt; };
class FPlasticMaterial : public FMaterial                        // material.h:85-110, material.cc:12-29
{ public: FPlasticMaterial(const FColor& Kd, const FColor& Ks, Float roughness, bool remapRoughness);
  int Kind() const override { return JP_MAT_PLASTIC; } void Flatten(float*) const override; FColor Kd, Ks; Float roughness; bool remapRoughness; Float Qd; };
class FMetalMaterial : public FMaterial                          // material.h:113-137, material.cc:31-43
{ public: FMetalMaterial(const FColor& eta, const FColor& k, Float ur, Float vr, bool remap) : eta(eta), k(k), uRoughness(ur), vRoughness(vr), remapRoughness(remap) {}
  int Kind() const override { return JP_MAT_METAL; } void Flatten(float*) const override; FColor eta, k; Float uRoughness, vRoughness; bool remapRoughness; };

// ---- lights -------------------------------------------------------------------------------------------------
class FScene;
enum eLightFlags { DeltaPosition = 1, DeltaDirection = 2, AreaLight = 4, InfiniteLight = 8 };   // light.h:16-23

class FLight
{
public:
	virtual ~FLight() {}
	FLight(int flags) : lightFlags(flags) {}
	int Flags() const { return lightFlags; }
	virtual int Kind() const = 0;                                // JP_LIGHT_*
	virtual void Preprocess(const FScene&) {}
	int lightFlags;
};
class FAreaLight : public FLight                                 // light.h:183-244
{ public: FAreaLight(const FPoint3&, int, const FColor& radiance, const FShape* shape) : FLight(eLightFlags::AreaLight), radiance(radiance), shape(shape) {}
  int Kind() const override { return JP_LIGHT_AREA; } FColor radiance; const FShape* shape; };
class FEnvironmentLight : public FLight                          // light.h:248-311, light.cc:26-33
{ public: FEnvironmentLight(const FPoint3&, int, const FColor& radiance) : FLight(eLightFlags::InfiniteLight), radiance(radiance), worldRadius(0) {}
  int Kind() const override { return JP_LIGHT_ENVIRONMENT; } void Preprocess(const FScene& scene) override; FColor radiance; FPoint3 worldCenter; Float worldRadius; };

class FPointLight : public FLight                                // light.h:81-132
{ public: FPointLight(const FPoint3& worldpos, int, const FColor& intensity) : FLight(eLightFlags::DeltaPosition), worldPosition(worldpos), intensity(intensity) {}
  int Kind() const override { return JP_LIGHT_POINT; } FPoint3 worldPosition; FColor intensity; };
class FDirectionLight : public FLight                            // light.h:136-180, light.cc:17-24
{ public: FDirectionLight(const FPoint3&, int, const FColor& irradiance, const FVector3& worlddir) : FLight(eLightFlags::DeltaDirection), irradiance(irradiance), worldDir(Normalize(worlddir)), worldRadius(0) {}
  int Kind() const override { return JP_LIGHT_DIRECTION; } void Preprocess(const FScene& scene) override; FColor irradiance; FVector3 worldDir; FPoint3 worldCenter; Float worldRadius; };

struct FPrimitive                                                // primitive.h:20-64
{
	const FShape* shape; const FMaterial* material; const FAreaLight* arealight;
	FPrimitive(const FShape* s, const FMaterial* m, const FAreaLight* l) : shape(s), material(m), arealight(l) {}
};

// ---- sampler / camera / film ----------------------------------------------------------------------------------
struct FCameraSample { FPoint2 posfilm; };

class FSampler                                                   // sampler.h:64-105 (description only: the draws happen on the device)
{
public:
	virtual ~FSampler() {}
	FSampler(int spp) : samples_per_pixel(spp) {}
	virtual std::unique_ptr<FSampler> Clone() = 0;
	virtual int GetSamplesPerPixel() { return samples_per_pixel; }
	virtual void SetSamplesPerPixel(int s) { samples_per_pixel = s; }
	virtual uint32_t Seed() const = 0;
	virtual int Mode() const { return JP_SAMPLER_COUNTER; }     // JpRenderParams::sampler_mode
protected:
	int samples_per_pixel;
};
// The counter stream of include/jp_counter_rng.h.
class FCounterSampler : public FSampler
{ public: FCounterSampler(int spp, uint32_t seed = 1234) : FSampler(spp), seed(seed) {} std::unique_ptr<FSampler> Clone() override { return std::make_unique<FCounterSampler>(samples_per_pixel, seed); }
  uint32_t Seed() const override { return seed; } uint32_t seed; };
// Drop-in for main.cc:149.  The stock sequential mt19937_64 stream cannot be reproduced in parallel, so a
// "random sampler" is served by the counter stream with the stock seed 1234 (sampler.h:26): same
// distribution, different random numbers.
class FRandomSampler : public FCounterSampler { public: FRandomSampler(int spp) : FCounterSampler(spp, 1234) {} };
// sampler.h:160-185: the reference's stratified sampler is an unfinished copy of its random sampler ("TODO") -- the same here
class FStratifiedSampler : public FCounterSampler { public: FStratifiedSampler(int spp) : FCounterSampler(spp, 1234) {} };
// sampler.h:109-127: every draw is 0.5 and the camera sample is the pixel centre (the reference's GetCameraSample computes
// posfilm + (0.5, 0.5) but lacks its return statement); served by the device's JP_SAMPLER_DEBUG mode
class FDebugSampler : public FSampler
{ public: FDebugSampler(int spp) : FSampler(spp) {} std::unique_ptr<FSampler> Clone() override { return std::make_unique<FDebugSampler>(samples_per_pixel); }
  uint32_t Seed() const override { return 0; } int Mode() const override { return JP_SAMPLER_DEBUG; } };

class FCamera                                                    // camera.h:32-70
{
public:
	virtual ~FCamera() {}
	FCamera(const FVector3& ipos, const FVector3& ifront, const FVector3& iup, Float ifov, const FVector2& iresolution);
	FVector3 pos, front, right, up; FVector2 resolution;
};

enum class EImageType { PPM, BMP, HDR };                         // film.h:15-20

inline Float Clamp01(Float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }                                   // film.h:22
inline uint8_t gamma_encoding(Float x) { return (uint8_t)(std::pow(Clamp01(x), (Float)(1 / 2.2)) * 255.0); }   // film.h:24

class FFilm                                                      // film.h:27-94
{
public:
	FFilm(int w, int h) : width(w), height(h), pixels((size_t)w * h) {}
	int Width() const { return width; } int Height() const { return height; }
	FVector2 GetResolution() const { return FVector2((Float)width, (Float)height); }
	// (writable access: the 8-bit bytes the device delivered describe the film as it was rendered, so any change drops them)
	FColor& operator()(int x, int y) { InvalidateLDR(); return pixels[(size_t)width * y + x]; }
	const FColor& operator()(int x, int y) const { return pixels[(size_t)width * y + x]; }
	void AddColor(int x, int y, const FColor& c) { FColor& p = (*this)(x, y); p.r += c.r; p.g += c.g; p.b += c.b; }
	void Clear() { InvalidateLDR(); for (auto& p : pixels) p = FColor(); }
	void InvalidateLDR() { if (!ldr8.empty()) { ldr8.clear(); floatValid = true; } }
	// the output step right after the hot path (main.cc:160): <filename>.ppm/.bmp/.hdr, gamma 1/2.2 for the 8-bit formats
	bool SaveAsImage(const std::string& filename, EImageType imgType) const;
	// Tone mapping on the GPU: ask the integrator for the film as 8-bit gamma-encoded RGB (gamma_encoding of film.h:24 applied on
	// the device after the resolve, byte-identical to the host's) next to -- or, with ldrOnly, INSTEAD of -- the fp32 pixels: a
	// BMP / PPM then costs a 3-bytes-per-pixel download instead of 12.  SaveAsImage writes ldr8 when it is present.
	void RequestDeviceLDR(bool only = true) { wantLDR = true; ldrOnly = only; }
	bool HasLDR() const { return ldr8.size() == (size_t)width * height * 3; }
	int width, height; std::vector<FColor> pixels;
	bool wantLDR = false, ldrOnly = false; std::vector<uint8_t> ldr8;   // R G B per pixel, top row first
	bool floatValid = true;                                            // false after an LDR-only render: the fp32 pixels were not downloaded
};

// ---- reflection API (bsdf.h, bsdf.cc, microfacet.h, microfacet.cc) --------------------------------------------------------
// The reference's BSDF classes by name, for code that builds or probes a BSDF directly (the render path builds its closures on the
// device from the material table).  Objects only hold their parameters; Evalf / Pdf / Sample run ON THE DEVICE through jp_bsdf
// (k_bsdf, csrc/jp_xbsdf.h) -- there is no host implementation.  World-space vectors, frame = FFrame(normal) as in bsdf.h:284-302.
struct FFrame { FVector3 n; FFrame(const FVector3& nn) : n(nn) {} };                     // geometry.h:326-378 (built on the device)
struct FBSDFSample { FColor f; FVector3 wi; Float pdf = 0; int ebsdf = 0; };             // bsdf.h:252-265
class Fresnel { public: virtual ~Fresnel() {} virtual void Fill(JpBsdfDesc& d) const = 0; };               // bsdf.h:637-643
class FresnelConductor : public Fresnel                                                   // bsdf.h:645-655
{ public: FresnelConductor(const FColor& etaI, const FColor& etaT, const FColor& k) : etaI(etaI), etaT(etaT), k(k) {}
  void Fill(JpBsdfDesc& d) const override { d.fresnel = JP_FRESNEL_CONDUCTOR; d.fr_eta_i[0] = etaI.r; d.fr_eta_i[1] = etaI.g; d.fr_eta_i[2] = etaI.b; d.fr_eta_t[0] = etaT.r; d.fr_eta_t[1] = etaT.g; d.fr_eta_t[2] = etaT.b; d.fr_k[0] = k.r; d.fr_k[1] = k.g; d.fr_k[2] = k.b; }
  FColor etaI, etaT, k; };
class FresnelDielectric : public Fresnel                                                  // bsdf.h:657-662
{ public: FresnelDielectric(Float etaI, Float etaT) : etaI(etaI), etaT(etaT) {}
  void Fill(JpBsdfDesc& d) const override { d.fresnel = JP_FRESNEL_DIELECTRIC; d.fr_eta_i[0] = d.fr_eta_i[1] = d.fr_eta_i[2] = etaI; d.fr_eta_t[0] = d.fr_eta_t[1] = d.fr_eta_t[2] = etaT; }
  Float etaI, etaT; };
class FresnelNoOp : public Fresnel { public: void Fill(JpBsdfDesc& d) const override { d.fresnel = JP_FRESNEL_NOOP; } };   // bsdf.h:664-667
class MicrofacetDistribution                                                              // microfacet.h:16-39
{ public: virtual ~MicrofacetDistribution() {} MicrofacetDistribution(int kind, Float ax, Float ay, bool vis) : kind(kind), alphax(ax), alphay(ay), sampleVisibleArea(vis) {}
  int kind; Float alphax, alphay; bool sampleVisibleArea; };
class BeckmannDistribution : public MicrofacetDistribution                               // microfacet.h:42-63
{ public: BeckmannDistribution(Float ax, Float ay, bool samplevis = true) : MicrofacetDistribution(JP_DIST_BECKMANN, ax, ay, samplevis) {}
  static Float RoughnessToAlpha(Float roughness) { roughness = roughness < (Float)1e-3 ? (Float)1e-3 : roughness; Float x = std::log(roughness); return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x; } };
class TrowbridgeReitzDistribution : public MicrofacetDistribution                         // microfacet.h:65-99
{ public: TrowbridgeReitzDistribution(Float ax, Float ay, bool samplevis = true) : MicrofacetDistribution(JP_DIST_TROWBRIDGE_REITZ, ax, ay, samplevis) {}
  static Float RoughnessToAlpha(Float roughness) { return BeckmannDistribution::RoughnessToAlpha(roughness); } };
class FBSDF                                                                               // bsdf.h:268-332
{
public:
	virtual ~FBSDF() {}
	FBSDF(const FFrame& frame, int kind) : normal(frame.n) { desc = JpBsdfDesc(); desc.kind = kind; desc.color[0] = desc.color[1] = desc.color[2] = 1; desc.eta_a = 1; desc.eta_b = 1; desc.sample_visible = 1; }
	virtual bool IsDelta() const { return desc.kind == JP_BSDF_MIRROR || desc.kind == JP_BSDF_FRESNEL_SPECULAR; }
	FColor Evalf(const FVector3& world_wo, const FVector3& world_wi) const;
	Float Pdf(const FVector3& world_wo, const FVector3& world_wi) const;
	FBSDFSample Sample(const FVector3& world_wo, const FVector2& random) const;
	JpBsdfDesc desc; FVector3 normal;
protected:
	void SetColor(const FColor& c) { desc.color[0] = c.r; desc.color[1] = c.g; desc.color[2] = c.b; }
	void SetDist(const MicrofacetDistribution* d) { desc.distribution = d->kind; desc.alpha_x = d->alphax; desc.alpha_y = d->alphay; desc.sample_visible = d->sampleVisibleArea ? 1 : 0; }
};
class FLambertionReflection : public FBSDF { public: FLambertionReflection(const FFrame& f, const FColor& albedo) : FBSDF(f, JP_BSDF_LAMBERT) { SetColor(albedo); } };   // bsdf.h:336-385
class FSpecularReflection : public FBSDF { public: FSpecularReflection(const FFrame& f, const FColor& R) : FBSDF(f, JP_BSDF_MIRROR) { SetColor(R); } };                 // bsdf.h:394-435
class FFresnelSpecular : public FBSDF                                                     // bsdf.h:455-552
{ public: FFresnelSpecular(const FFrame& f, Float eta_i, Float eta_t, const FColor& Kr, const FColor& Kt) : FBSDF(f, JP_BSDF_FRESNEL_SPECULAR) { SetColor(Kr); desc.color2[0] = Kt.r; desc.color2[1] = Kt.g; desc.color2[2] = Kt.b; desc.eta_a = eta_i; desc.eta_b = eta_t; } };
class FPhongSpecularReflection : public FBSDF                                             // bsdf.h:557-633
{ public: FPhongSpecularReflection(const FFrame& f, const FColor& Ks, Float exponent) : FBSDF(f, JP_BSDF_PHONG) { SetColor(Ks); desc.exponent = exponent; } };
class FMicrofacetReflection : public FBSDF                                                // bsdf.h:676-701, bsdf.cc:29-78 (owns both objects, as the reference does)
{ public: FMicrofacetReflection(const FFrame& f, const FColor& R, MicrofacetDistribution* distribution, Fresnel* fresnel) : FBSDF(f, JP_BSDF_MICROFACET_REFLECTION), distribution(distribution), fresnel(fresnel) { SetColor(R); SetDist(distribution); fresnel->Fill(desc); }
  ~FMicrofacetReflection() { delete distribution; delete fresnel; }
  const MicrofacetDistribution* distribution; const Fresnel* fresnel; };
class FMicrofacetTransmission : public FBSDF                                              // bsdf.h:703-734, bsdf.cc:80-145
{ public: FMicrofacetTransmission(const FFrame& f, const FColor& T, MicrofacetDistribution* distribution, Float etaA, Float etaB) : FBSDF(f, JP_BSDF_MICROFACET_TRANSMISSION), distribution(distribution) { SetColor(T); SetDist(distribution); desc.eta_a = etaA; desc.eta_b = etaB; }
  ~FMicrofacetTransmission() { delete distribution; }
  const MicrofacetDistribution* distribution; };

// ---- scene ----------------------------------------------------------------------------------------------------
struct FlatBVH { std::vector<float> bounds; std::vector<int32_t> left, right, prim_index; };

class FScene                                                     // scene.h:23-150, scene.cc
{
public:
	FScene(const char* inName) : name(inName) {}
	const char* NameStr() const { return name.c_str(); }
	void Preprocess();                                           // world bound -> light preprocess -> BVH build (scene.cc:11-23)
	FBounds3 WorldBound() const { return worldBound; }
	const FCamera* Camera() const { return camera.get(); }
	int LightNum() const { return (int)lights.size(); }

	template<typename T, typename... U> std::shared_ptr<T> CreateCamera(const U&... args) { auto c = std::make_shared<T>(args...); camera = c; return c; }
	template<typename T, typename... U> std::shared_ptr<T> CreateShape(const U&... args) { auto s = std::make_shared<T>(args...); shapes.push_back(s); return s; }
	template<typename T, typename... U> std::shared_ptr<T> CreateMaterial(const U&... args) { auto m = std::make_shared<T>(args...); materials.push_back(m); return m; }
	template<typename T, typename... U> std::shared_ptr<T> CreateLight(const U&... args) { auto l = std::make_shared<T>(args...); lights.push_back(l); return l; }
	template<typename... U> std::shared_ptr<FPrimitive> CreatePrimitive(const U&... args) { auto p = std::make_shared<FPrimitive>(args...); primitives.push_back(p); return p; }

	std::vector<std::shared_ptr<FShape>> CreateTriangleMesh(const char* filename, bool flip_normal = false, bool bFlipHandedness = false, const FVector3& offset = FVector3(0, 0, 0), Float inScale = 1.f);
	std::vector<std::shared_ptr<FPrimitive>> CreatePrimitives(const std::vector<std::shared_ptr<FShape>>& inMesh, const std::shared_ptr<FMaterial>& inMaterial);
	std::vector<std::shared_ptr<FAreaLight>> CreateAreaLights(int samplesNum, const FColor& radiance, const std::vector<std::shared_ptr<FShape>>& inShapes, const std::shared_ptr<FMaterial>& inMaterial);
	std::shared_ptr<FAreaLight> CreateAreaLight(int samplesNum, const FColor& radiance, const std::shared_ptr<FShape>& inShape, const std::shared_ptr<FMaterial>& inMaterial);

	std::string name;
	std::shared_ptr<FCamera> camera;
	std::vector<std::shared_ptr<FShape>> shapes;
	std::vector<std::shared_ptr<FMaterial>> materials;
	std::vector<std::shared_ptr<FLight>> lights;
	std::vector<std::shared_ptr<FPrimitive>> primitives;
	FBounds3 worldBound;
	FlatBVH bvh;                                                 // built by Preprocess() unless deviceBuild
	bool preprocessed = false;
	// true: Preprocess() skips the host SAH build and the flattened scene carries no hierarchy (n_bvh_nodes = 0), so
	// jp_upload_scene builds the tree on the device (round 3: PLOC clustering, jp_ploc.h -- as fast to walk as the host's binned-SAH
	// tree on the 280k-triangle scene, setup 0.2 s -> 0.06 s; rounds 1-2: an LBVH, 7-18 % slower to walk).
	// Default (round 3): scenes of more than kDeviceBuildFrom primitives build on the device, smaller ones on the host (their trees
	// become flat leaf lists / LDS-resident trees, which need the host's leaves).  hostBuild = true or env JETPBRT_DEVICE_BVH=0 keep the
	// host build for every scene, deviceBuild = true or JETPBRT_DEVICE_BVH=1 force the device build.
	bool deviceBuild = false, hostBuild = false;
	bool builtOnDevice = false;                                  // what the last Preprocess() decided (read-only for callers)
	static constexpr size_t kDeviceBuildFrom = 4096;
	// true: Preprocess() builds the reference's own tree (BuildReferenceBVH: its rand() sequence from the default seed, its
	// std::sort, median split, leaves <= 5, over the reference's WorldBounds) and the flattened scene asks the device to walk
	// it with the reference's semantics (JpScene.bvh_reference_semantics): the film then equals the reference's bit for bit on
	// tessellated meshes too, at several times the traversal cost.  Default from env JETPBRT_REFERENCE_TREE=1.
	bool referenceTree = false;
	// With referenceTree: the CERTIFIED walk (JpScene.bvh_reference_semantics = 2; DESIGN.md "Certified walk") -- an ordered walk over the leaves
	// of the reference's tree that proves, ray by ray, that FBVH_Node::Intersect returns the same hit, and repeats the few rays it cannot prove
	// the reference's way.  About twice as fast as the verbatim walk on large meshes.  Outside the proof: a ray that lies within fp32 noise of a
	// triangle's plane is decided in the reference by the signs of rounding errors wherever that triangle is; camera rays are covered by flags on the
	// leaves edge-on to the camera, secondary rays only by measurement (the 280k-triangle frame is bit-identical).  Default from env JETPBRT_REFERENCE_TREE=2.
	bool certifiedWalk = false;
};

// binned-SAH BVH over primitive bounds -> the flat node arrays of JpScene (own topology)
void BuildBVH(const std::vector<FBounds3>& primBounds, FlatBVH& out, int maxLeaf = 4);
// the reference's tree node for node (bvh.h:54-146 with glibc's rand() from seed 1), over the reference's WorldBounds
void BuildReferenceBVH(const std::vector<FBounds3>& primWorldBounds, FlatBVH& out);

// the flattener: owns the SoA storage a JpScene view points into
struct FlatScene
{
	JpScene view;
	std::vector<float> tri_p0, tri_p1, tri_p2, tri_n, rect_p0, rect_p1, rect_p2, rect_p3, rect_n, sph_center, sph_radius;
	std::vector<float> disk_center, disk_normal, disk_radius;
	std::vector<int32_t> prim_shape_type, prim_shape_index, prim_material, prim_light, mat_type, light_type, light_prim;
	std::vector<float> mat_params, light_radiance, light_vec;
	FlatBVH bvh;
};
bool FlattenScene(const FScene& scene, FlatScene& out, std::string* error = nullptr);

// ---- integrator -------------------------------------------------------------------------------------------------
class FIntegrator
{
public:
	virtual ~FIntegrator() {}
	// same contract as integrator.h:32 / integrator.cc:35-80: blocking, results ADDED onto `film` (film.h:64-68)
	virtual void Render(const FScene* scene, FSampler* sampler, FFilm* film, int numthreads = 1) const = 0;
};

// FPathIntegratorIteration (integrator.h:109-122, integrator.cc:316-403) executed by the HIP wavefront kernels.
// `numthreads` is accepted for signature compatibility and ignored.  Failures (no GPU, library missing, bad
// scene) print through PBRT-style stderr logging and leave the film untouched -- the reference's error
// convention; there is NO CPU fallback.  LastStatus() exposes the JpStatus.
class FGpuPathIntegrator : public FIntegrator
{
public:
	explicit FGpuPathIntegrator(int maxDepth, int deviceId = 0);
	~FGpuPathIntegrator();
	void Render(const FScene* scene, FSampler* sampler, FFilm* film, int numthreads = 1) const override;
	int LastStatus() const { return lastStatus; }
	const JpCounters& Counters() const { return counters; }
	// multi-GPU band sharding (JpRenderParams::shard_*); default renders the whole film
	void SetShard(int index, int count, int bandRows = 20) { shardIndex = index; shardCount = count; this->bandRows = bandRows; }
	// ABI 7: the library's switches by value (JpOptions of include/jetpbrt_amd.h: stream lanes, traversal overrides, certified-walk slack, ...);
	// zero-initialised = the defaults.  Handed to jp_set_options before the next upload / render.
	void SetOptions(const JpOptions& o) { options = o; options.struct_bytes = (int32_t)sizeof(JpOptions); optionsDirty = true; uploaded = nullptr; }
	const JpOptions& Options() const { return options; }
protected:
	JpOptions options = {};
	mutable bool optionsDirty = false;
	int maxDepth, deviceId;
	int shardIndex = 0, shardCount = 1, bandRows = 20;
	mutable JpContext* ctx = nullptr;
	mutable const FScene* uploaded = nullptr;
	mutable int lastStatus = 0;
	mutable JpCounters counters;
protected:
	int kind = JP_INTEGRATOR_PATH;                               // JpRenderParams::integrator
};
// the reference's other integrators behind the same Render() (integrator.h:44-85): device megakernel k_other, not the hot path
class FWhittedIntegrator : public FGpuPathIntegrator
{ public: explicit FWhittedIntegrator(int maxDepth, int deviceId = 0) : FGpuPathIntegrator(maxDepth, deviceId) { kind = JP_INTEGRATOR_WHITTED; } };
class FDebugIntegrator : public FGpuPathIntegrator
{ public: explicit FDebugIntegrator(int deviceId = 0) : FGpuPathIntegrator(0, deviceId) { kind = JP_INTEGRATOR_DEBUG_NORMAL; } };
typedef FGpuPathIntegrator FPathIntegratorIteration;             // main.cc:154 compiles unchanged
// FPathIntegratorRecursive (integrator.h:88-106, integrator.cc:233-307) is the same estimator written recursively: the same
// draws in the same order, emission on bounce 0 / after a specular bounce, NEE over all lights, roulette from bounce 3.  It
// differs from the iterative form only in how the throughput products are rounded (nested f * cos * Li / pdf instead of a
// running beta), i.e. in the last bits of a path's radiance -- measured against the reference's recursive integrator on the
// golden scenes: mean per-pixel L2 ~1e-8 (tests/test_gpu_parity.py).  The device serves it with the same kernels.
typedef FGpuPathIntegrator FPathIntegratorRecursive;

} // namespace jetpbrt
