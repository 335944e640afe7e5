/* include/jetpbrt_amd.h -- C ABI of the MI355X path-tracing integrator (libjetpbrt_amd.so).
 *
 * Drop-in boundary.  The reference has no FFI; the seam this library sits behind is
 *     void FIntegrator::Render(const FScene*, FSampler*, FFilm*, int numthreads) const   (integrator.h:32,
 *     integrator.cc:35-80)  with  FPathIntegratorIteration::Li  (integrator.cc:316-403)  as the integrator.
 * A host `Render()` (ours: jet-pbrt_amd/host/integrator.h, FGpuPathIntegrator::Render) flattens the
 * preprocessed FScene into the plain arrays of JpScene and calls the entry points below instead of spawning
 * the 20-row CPU tasks of integrator.cc:53-74 / parallel.cc.
 *
 * Conventions: plain C types only; every function returns JP_OK (0) or a negative JpStatus and never throws;
 * jp_last_error() returns a thread-local message for the last failure.  The caller owns every host buffer;
 * the context owns all device memory.  All floats are IEEE binary32 ("Float" = float, pbrt.h:27).
 * Arrays of points/vectors/colours are tightly packed xyz / rgb triples.
 */
#ifndef JETPBRT_AMD_H
#define JETPBRT_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JP_ABI_VERSION 7

typedef enum JpStatus {
    JP_OK = 0,
    JP_ERR_INVALID_ARGUMENT = -1,   /* null pointer, bad enum, index out of range, malformed BVH ... */
    JP_ERR_NO_DEVICE = -2,          /* no HIP device / device id out of range                         */
    JP_ERR_DEVICE = -3,             /* a HIP runtime call failed (message has the hipError string)    */
    JP_ERR_NO_SCENE = -4,           /* jp_render before jp_upload_scene                               */
    JP_ERR_UNSUPPORTED = -5         /* e.g. sampler mode the device path cannot reproduce             */
} JpStatus;

/* shape kinds: FTriangle shape.h:277-369, FRectangle shape.h:380-472, FSphere shape.h:476-662, FDisk shape.h:189-275 */
enum { JP_SHAPE_TRIANGLE = 0, JP_SHAPE_RECTANGLE = 1, JP_SHAPE_SPHERE = 2, JP_SHAPE_DISK = 3 };
/* material kinds: material.h:27-41 (matte), :45-59 (mirror), :63-81 (glass), :85-110 + material.cc:12-29
 * (plastic), material.h:113-137 + material.cc:31-43 (metal) */
enum { JP_MAT_MATTE = 0, JP_MAT_MIRROR = 1, JP_MAT_GLASS = 2, JP_MAT_PLASTIC = 3, JP_MAT_METAL = 4 };
/* light kinds: FEnvironmentLight light.h:248-311, FAreaLight light.h:183-244, FPointLight light.h:81-132,
 * FDirectionLight light.h:136-180 */
enum { JP_LIGHT_ENVIRONMENT = 0, JP_LIGHT_AREA = 1, JP_LIGHT_POINT = 2, JP_LIGHT_DIRECTION = 3 };
/* sampler: the stock sequential mt19937_64 stream (sampler.h:16-54) cannot be reproduced by a parallel
 * device; the device path implements the counter-based stream of include/jp_counter_rng.h only. */
enum { JP_SAMPLER_STOCK_MT19937 = 0, JP_SAMPLER_COUNTER = 1,
       JP_SAMPLER_DEBUG = 2 };  /* FDebugSampler sampler.h:109-127: every draw is 0.5, the camera sample is the pixel + (0.5, 0.5) (what the body of its
                                 * GetCameraSample computes; the reference's function lacks its return statement)                                  */
/* integrators (integrator.h): PATH = FPathIntegratorIteration (integrator.cc:316-403; FPathIntegratorRecursive is the same
 * estimator); WHITTED = FWhittedIntegrator (integrator.cc:115-220; branches at mirrors, one thread walks a sample's whole
 * tree); DEBUG_NORMAL = FDebugIntegrator (integrator.h:44-58).  The last two are API completeness, not the hot path. */
enum { JP_INTEGRATOR_PATH = 0, JP_INTEGRATOR_WHITTED = 1, JP_INTEGRATOR_DEBUG_NORMAL = 2 };

#define JP_MAT_PARAM_STRIDE 16
/* mat_params layout, JP_MAT_PARAM_STRIDE floats per material (the state the reference material objects hold):
 *   MATTE   [0..2] diffuseColor                                                     material.h:29-40
 *   MIRROR  [0..2] specularColor                                                    material.h:47-58
 *   GLASS   [0] eta, [1..3] Kr, [4..6] Kt                                           material.h:65-80
 *   PLASTIC [0..2] Kd, [3..5] Ks, [6] alpha (roughness after the optional remap), [7] Qd      material.h:88-109
 *   METAL   [0..2] eta, [3..5] k, [6] alpha_x, [7] alpha_y (after the optional remap)         material.h:116-136
 */

/* FCamera state after its constructor (camera.h:36-49): right/up already scaled by tan(fov/2) (and aspect). */
typedef struct JpCamera {
    float pos[3], front[3], right[3], up[3];
    float res_x, res_y;
} JpCamera;

typedef struct JpScene {
    JpCamera camera;

    /* shapes, SoA */
    int32_t n_triangles;   const float *tri_p0, *tri_p1, *tri_p2, *tri_n;              /* n = FTriangle::normal (flip applied) */
    int32_t n_rectangles;  const float *rect_p0, *rect_p1, *rect_p2, *rect_p3, *rect_n;
    int32_t n_spheres;     const float *sph_center; const float *sph_radius;

    /* primitives (FPrimitive, primitive.h:20-64) in creation order == FScene::shadow_primitives before Preprocess */
    int32_t n_primitives;
    const int32_t *prim_shape_type;    /* JP_SHAPE_*                                    */
    const int32_t *prim_shape_index;   /* index into the arrays of that shape kind      */
    const int32_t *prim_material;      /* material index, -1 = nullptr material         */
    const int32_t *prim_light;         /* index into lights (FPrimitive::arealight), -1 = none */

    int32_t n_materials;   const int32_t *mat_type; const float *mat_params;

    /* lights in FScene::Lights() order (creation order; scene.h:92-106) */
    int32_t n_lights;
    const int32_t *light_type;         /* JP_LIGHT_*                                    */
    const float   *light_radiance;     /* rgb: radiance (AREA, ENVIRONMENT), intensity (POINT), irradiance (DIRECTION) */
    const int32_t *light_prim;         /* AREA: primitive whose shape emits; otherwise -1 */
    const float   *light_vec;          /* xyz per light: POINT worldPosition, DIRECTION normalised worldDir, otherwise 0;
                                          may be NULL when the scene has neither kind */
    float world_radius;                /* FEnvironmentLight / FDirectionLight::worldRadius after Preprocess (light.cc:17-33) */

    /* bounding volume hierarchy over the primitives, built by the host (own topology; closest-hit and
     * occlusion results do not depend on it).  Node i: bounds 6 floats (min xyz, max xyz);
     * bvh_left[i] >= 0: interior, children bvh_left[i], bvh_right[i];
     * bvh_left[i] <  0: leaf, primitives bvh_prim_index[first .. first+count) with first = -bvh_left[i]-1,
     *                   count = bvh_right[i].  Node 0 is the root.
     * n_bvh_nodes == 0 (all four pointers may be NULL): no hierarchy is handed over and jp_upload_scene builds one on
     * the device from the primitive extents (LBVH; replaces the host build of FScene::Preprocess, scene.cc:11-23). */
    int32_t n_bvh_nodes;   const float *bvh_bounds; const int32_t *bvh_left, *bvh_right;
    int32_t n_bvh_prim_indices; const int32_t *bvh_prim_index;
    /* 0: the hierarchy is acceleration only -- the library is free to re-shape it (8-wide shadow tree, flat leaf list) and
     *    tests boxes conservatively, so no hit is ever dropped.
     * 1: REFERENCE SEMANTICS -- the given tree is walked node for node the way FBVH_Node::Intersect does (bvh.h:94-103: box
     *    test of geometry.cc:10-30 with its `tmax <= tmin` rejection on the unpadded bounds, left subtree then right, leaf
     *    objects in order).  With the reference's own tree (host: FScene::referenceTree) the hits are then the reference's hits
     *    even where those depend on its topology (finely tessellated meshes, DESIGN.md "Numerics").  Several times slower.
     * 2: (ABI 6) REFERENCE SEMANTICS, CERTIFIED WALK -- scenes of more than 1024 primitives: the rays take an ordered walk over the LEAVES
     *    of the given tree; each result carries a proof that the walk of (1) returns the same hit (FBounds3::Intersect on the hit's leaf
     *    box with max_t = the hit distance implies the same for every ancestor and every larger max_t), and the rays without a proof are
     *    walked again as in (1).  About twice as fast as (1).  One class of rays is outside the proof: a ray within fp32 noise of a
     *    triangle's plane is accepted or not by the signs of rounding errors, wherever along the ray that triangle lies.  Rays from the camera
     *    position are covered (leaves with a primitive edge-on to the camera are never culled by distance for them); for secondary rays there is
     *    only a generous distance-cull slack and the measurement: the 280k-triangle frame at 800x600x2048 and a 1/8 shard of 1920x1080x4096 are bit-identical
     *    to (1) (DESIGN.md "Certified walk").  Smaller scenes: as (1). */
    int32_t bvh_reference_semantics;

    /* FDisk (shape.h:189-275): position, normal (already normalised by the constructor, shape.h:194), radius */
    int32_t n_disks;       const float *disk_center, *disk_normal; const float *disk_radius;
} JpScene;

typedef struct JpRenderParams {
    int32_t width, height;       /* film size == camera resolution (main.cc:115)                          */
    int32_t spp;                 /* FSampler::samples_per_pixel                                           */
    int32_t max_depth;           /* FPathIntegratorIteration::maxDepth (main.cc:154: 5)                   */
    int32_t sampler_mode;        /* JP_SAMPLER_*                                                          */
    uint32_t seed;               /* counter sampler seed                                                  */
    /* pixel-band sharding (multi-GPU): the film is cut into bands of `band_rows` rows -- the unit of
     * FRenderTask, lines_per_task = 20, integrator.cc:53 -- and this call renders band b iff
     * b % shard_count == shard_index.  Pixels outside the shard are written as 0, so the sum over shards
     * (one RCCL reduce) is the full film.  shard_count <= 1 renders everything. */
    int32_t band_rows, shard_index, shard_count;
    int32_t integrator;          /* JP_INTEGRATOR_*: 0 = FPathIntegratorIteration / Recursive (the hot path, wavefront kernels)   */
} JpRenderParams;

typedef struct JpCounters {
    uint64_t samples;            /* camera samples traced                                                 */
    uint64_t closest_rays;       /* FScene::Intersect calls from Li (integrator.cc:327)                   */
    uint64_t closest_hits;
    uint64_t shadow_rays;        /* FScene::Occluded calls (integrator.cc:367)                            */
    uint64_t shadow_occluded;
    double   render_ms;          /* device time of the last jp_render*, HIP events on the context stream  */
    double   extend_ms, shade_ms, shadow_ms, other_ms;   /* per kernel class, when profiling is enabled   */
    uint64_t extend_launches, shade_launches, shadow_launches;
    /* ABI 5: the fused schedule (k_path: ray generation and every bounce of a queue region in one launch, DESIGN.md "One schedule").
     * When it runs, extend_ms / shade_ms / shadow_ms stay 0 and path_ms is the sum of the k_path launches. */
    double   path_ms;
    uint64_t path_launches;
    /* ABI 6: reference semantics with the certified walk (JpBuildInfo.certified_walk): rays (closest-hit and shadow) whose result the
     * ordered walk could not certify and which were walked again node for node the reference's way */
    uint64_t certified_fallback_rays;
} JpCounters;

/* what jp_upload_scene did with the hierarchy */
typedef struct JpBuildInfo {
    int32_t built_on_device;     /* 1: built by the device LBVH pass, 0: the caller's tree was used          */
    int32_t traversal_mode;      /* 0 binary tree in HBM, 1 binary tree in LDS, 2 flat leaf list, 3 binary + 8-wide */
    int32_t bvh_nodes, bvh_height;
    double  device_build_ms;     /* HIP-event time of the device build (0 when the caller's tree was used)     */
    int32_t libm_sincosf;        /* which build of the host libm's sinf/cosf/sincosf the device reproduces bit for bit:
                                    1 = glibc's FMA build, 2 = its build without contraction, 0 = none (own correctly
                                    rounded evaluation; films then differ from the host reference by rare path flips)  */
    int32_t lanes_last_render;   /* stream lanes the last jp_render* used (1-4; DESIGN.md "Stream lanes")                      */
    /* ABI 5: the last jp_render* ran the fused schedule (1) or the per-bounce launches of rounds 1-2 (0: JETPBRT_FUSED=0, scenes
     * whose tables do not fit LDS, the Whitted / debug integrators); region size and resident workgroups of that launch */
    int32_t fused_last_render, fused_region, fused_workgroups;
    int32_t q4_nodes;            /* ABI 5: nodes of the 4-wide quantised tree the closest-hit rays walk (0: none; scenes of <= 1024 primitives, device-built
                                    and reference-semantics trees, JETPBRT_Q4=0) */
    int32_t libm_xbsdf;          /* ABI 5: bit 0: the device reproduces the host libm's logf / expf / powf / acosf / atanf / tanf bit for bit (the BSDF
                                    classes behind jp_bsdf that call them are then bit-exact against the reference); bit 1: with libm's FMA build */
    int32_t certified_walk;      /* ABI 6: bvh_reference_semantics on a large scene: 1 = the rays take an ordered walk over the LEAVES of the caller's tree and
                                    carry a proof that FBVH_Node::Intersect returns the same hit; the few that cannot are walked again verbatim
                                    (DESIGN.md "Certified walk"; JETPBRT_CERTIFIED=0: every ray verbatim, as in ABI <= 5) */
    int32_t certified_nodes;     /* nodes of the 4-wide tree over the caller's leaves */
    int32_t certified_eye_leaves;/* leaves of the caller's tree holding a primitive whose plane passes through the camera position (within 5e-3 of its distance):
                                    camera rays are not culled by distance there (DESIGN.md "Certified walk") */
} JpBuildInfo;


/* ABI 7: everything a host application may want to steer, by value (round 3 read 42 JETPBRT_* environment variables at upload / render time).
 * Set with jp_set_options BEFORE the jp_upload_scene / jp_render* calls it is to affect; fields left 0 keep the library's defaults (tri-state
 * switches: 0 default, 1 on, -1 off).  The environment is read ONCE, in jp_create_context, as the initial value of this struct (JETPBRT_<FIELD> in
 * capitals; "0" means off): a test override, not an interface.  jp_get_options returns the values in force. */
typedef struct JpOptions {
    int32_t struct_bytes;        /* sizeof(JpOptions) of the caller (a longer struct of a later ABI is truncated, a shorter one zero-extended) */
    /* ---- schedule (jp_render*) ---- */
    int32_t lanes;               /* stream lanes per GPU, 1-4 (0: by frame size -- three for the benchmark frames; DESIGN.md "Stream lanes")  */
    int32_t lane_rows;           /* rows per lane group (0: 4)                                                                                 */
    int32_t blocks_per_cu;       /* workgroups (= queue regions) per CU and launch (0: 16; 5 per lane with three lanes)                        */
    int64_t max_slots;           /* cap on the path slots of one batch (0: from free device memory, <= 2^26)                                   */
    int32_t compact_regions;     /* k_raygen: a region = consecutive (pixel block, sample) chunks (default for scenes walked through HBM)      */
    int32_t fused;               /* 1: the single-launch schedule k_path (DESIGN.md "One schedule"; measured slower, kept for its 1.2 GB of queues) */
    int32_t fused_region, fused_job_spp, fused_workgroups;   /* its region size / samples per job / resident workgroups per CU (0: defaults)  */
    /* ---- traversal (jp_upload_scene) ---- */
    int32_t traversal;           /* force a traversal mode for a host-built tree: 1 + mode (1: binary tree in HBM, 2: binary tree in LDS, 3: flat leaf list, 4: + 8-wide shadow tree); 0: by scene size */
    int32_t q4;                  /* closest-hit rays of large scenes through the 4-wide quantised tree (default on)                            */
    int32_t q4_shadow;           /* ... and their shadow rays (default on; off: the 8-wide tree)                                               */
    int32_t persist;             /* lane refill in the traversal kernels of large scenes: idle lanes that trigger a refill, 8 / 16 / 32 (0: 16; -1: one ray per lane) */
    int32_t vote;                /* per-iteration node / leaf vote of the refill kernels (default on, off for the verbatim reference walk)    */
    int32_t stack_lds_words;     /* traversal-stack words per thread kept in LDS, the rest spills to HBM (0: 12)                               */
    int32_t shade_sort;          /* k_shade partitions its region by material class (default: scenes with more than one material kind)         */
    int32_t device_tree;         /* hierarchy built on the device: 1 PLOC clustering (default), 2 LBVH (Karras) topology                       */
    int32_t device_wide;         /* device build: also the 8-wide shadow tree (default on)                                                     */
    int32_t bvh_max_leaf;        /* device build: primitives per leaf (0: 2 for PLOC, 3 for LBVH)                                              */
    int32_t ploc_radius, ploc_max_rounds;   /* PLOC neighbour search radius (0: 16) and round limit (0: 512; beyond it the LBVH topology serves) */
    /* ---- reference semantics, certified walk (JpScene.bvh_reference_semantics = 2; DESIGN.md "Certified walk") ---- */
    int32_t certified;           /* -1: walk every ray verbatim even when the scene asks for the certified walk (it never turns the certified walk ON for a scene that asked for the verbatim one) */
    float   cert_slack;          /* K of the distance-cull slack tmax + K eps / (mean leaf diagonal) tmax^2 for rays not from the camera (0: 16384; < 0: none) */
    float   cert_slack_eye;      /* the same for rays from the camera position (0: 1024; < 0: none)                                            */
    float   cert_eye_tau;        /* a leaf is "edge-on to the camera" when |n.(p - eye)| <= tau |p - eye| (0: 5e-3; < 0: no flags)             */
    /* ---- host libm the device reproduces (0: probe the running libm in jp_create_context) ---- */
    int32_t libm_sincosf;        /* 1 / 2: glibc's FMA / plain build; -1: none (own correctly rounded evaluation)                              */
    int32_t libm_xbsdf;          /* bit 0 exact, bit 1 FMA build; -1: none                                                                     */
    /* ---- diagnostics (tests, tools/) ---- */
    int32_t trace_walk;          /* jp_trace on a large scene walks: 0 what the render's closest-hit rays walk, 1 the binary tree, 2 the 8-wide tree, 3 the caller's tree verbatim */
    float   box_pad;             /* every box of a host-built tree grows by this many scene units (fringe census, tools/gpu_fringe_census.py)  */
    int32_t reserved[8];
} JpOptions;

typedef struct JpContext JpContext;

const char* jp_last_error(void);
int  jp_abi_version(void);
/* which build of the host libm's sinf / cosf / sincosf the device will reproduce (JpBuildInfo.libm_sincosf): probes the
 * host's libm against the library's transcription of glibc's algorithm on 200,000 arguments.  Pure host code, no GPU needed. */
int  jp_probe_libm_sincosf(void);
/* the same for logf / expf / powf / acosf / atanf / tanf (JpBuildInfo.libm_xbsdf) */
int  jp_probe_libm_xbsdf(void);

/* one context per process per GPU (device_id = LOCAL_RANK) */
int  jp_create_context(int device_id, JpContext** out);
int  jp_destroy_context(JpContext* ctx);
/* ABI 7: options by value (see JpOptions); jp_set_options(ctx, NULL) restores the defaults of jp_create_context (environment overrides included) */
int  jp_set_options(JpContext* ctx, const JpOptions* options);
int  jp_get_options(JpContext* ctx, JpOptions* out);

/* validates every index in `scene` on the host, then copies it into device-resident SoA tables */
int  jp_upload_scene(JpContext* ctx, const JpScene* scene);

/* replaces FIntegrator::Render (integrator.cc:35-80): blocking; fills film_rgb (width*height*3 floats,
 * row-major, top row first, film.h:51-57) with Clamp01(sum_s Li_s / spp) (integrator.cc:89-108).
 * Values are SET, i.e. the result of AddColor onto the zero-initialised film of film.h:30-35. */
int  jp_render(JpContext* ctx, const JpRenderParams* params, float* film_rgb_host);
/* same, film left in device memory (film_rgb_device must hold width*height*3 floats on ctx's device);
 * asynchronous on the context stream unless `sync` != 0.  Used for the multi-GPU reduce. */
int  jp_render_device(JpContext* ctx, const JpRenderParams* params, void* film_rgb_device, int sync);
/* the film output step right after the hot path (FFilm::SaveAsImage, film.cc:11-145, main.cc:160): the same render, but the
 * film leaves the device as 8-bit gamma-encoded RGB -- gamma_encoding of film.h:24, (uint8_t)(pow(Clamp01(x), 1/2.2f) * 255.0),
 * applied on the GPU after the resolve -- so a BMP / PPM writer downloads width*height*3 BYTES instead of 12 bytes per pixel.
 * rgb8_host: width*height*3 bytes, row-major, top row first, R G B.  film_rgb_host may be NULL (8-bit image only) or receive
 * the fp32 film as jp_render does.  The encoding is byte-identical to the host libm's powf: the library tabulates, once per
 * process, the 255 fp32 thresholds at which the host's gamma_encoding steps (binary search over the float bit patterns of
 * [0, 1]) and the device counts the thresholds <= x (jp_gamma_thresholds: that table, for tests). */
int  jp_render_rgb8(JpContext* ctx, const JpRenderParams* params, uint8_t* rgb8_host, float* film_rgb_host);
int  jp_gamma_thresholds(float* out255);
/* test hook (pure host code): every fp32 value of [0, 1] -- 1,065,353,217 bit patterns, n_threads host threads -- through the host's
 * gamma_encoding and through the threshold table; returns the number of values whose bytes differ (0: the device tone map is
 * byte-identical for EVERY input; the table's binary search assumes the host's powf-based curve never steps down) */
long long jp_gamma_sweep(int n_threads);
int  jp_synchronize(JpContext* ctx);

/* per-kernel-class event timing (adds two events per launch); off by default */
int  jp_set_profiling(JpContext* ctx, int enabled);
int  jp_get_counters(JpContext* ctx, JpCounters* out);
int  jp_get_build_info(JpContext* ctx, JpBuildInfo* out);

/* A BSDF of bsdf.h / bsdf.cc / microfacet.cc described by value: every class of the reference's reflection API, including the
 * ones no material instantiates (FPhongSpecularReflection bsdf.h:557-633, BeckmannDistribution microfacet.cc:11-254,
 * FMicrofacetTransmission bsdf.cc:80-145, FresnelNoOp bsdf.h:664-667). */
enum { JP_BSDF_LAMBERT = 0, JP_BSDF_MIRROR = 1, JP_BSDF_FRESNEL_SPECULAR = 2, JP_BSDF_MICROFACET_REFLECTION = 3,
       JP_BSDF_MICROFACET_TRANSMISSION = 4, JP_BSDF_PHONG = 5 };
enum { JP_DIST_TROWBRIDGE_REITZ = 0, JP_DIST_BECKMANN = 1 };
enum { JP_FRESNEL_CONDUCTOR = 0, JP_FRESNEL_DIELECTRIC = 1, JP_FRESNEL_NOOP = 2 };
typedef struct JpBsdfDesc
{
    int32_t kind;                /* JP_BSDF_*                                                                                   */
    float   color[3];            /* albedo / R / Kr / T / Ks                                                                    */
    float   color2[3];           /* FRESNEL_SPECULAR: Kt                                                                        */
    float   eta_a, eta_b;        /* FRESNEL_SPECULAR: etaI, etaT; MICROFACET_TRANSMISSION: etaA, etaB                           */
    int32_t distribution;        /* JP_DIST_*       (microfacet kinds)                                                          */
    float   alpha_x, alpha_y;    /* as handed to the distribution's constructor (clamped to >= 0.001 there)                     */
    int32_t sample_visible;      /* sampleVisibleArea (microfacet.h:33-39)                                                      */
    int32_t fresnel;             /* JP_FRESNEL_*    (MICROFACET_REFLECTION)                                                     */
    float   fr_eta_i[3], fr_eta_t[3], fr_k[3];   /* conductor: etaI, etaT, k; dielectric: fr_eta_i[0], fr_eta_t[0]              */
    float   exponent;            /* PHONG                                                                                        */
} JpBsdfDesc;
/* test hook / utility: FBSDF::Evalf, ::Pdf and ::Sample (bsdf.h:284-302) of `desc` for n shading events on the device.  Host arrays:
 * normal / wo / wi 3n floats (world space; the frame is FFrame(normal), geometry.h:345-349), u 2n floats ->
 * f_eval 3n, pdf_eval n (for wo, wi); s_f 3n, s_wi 3n, s_pdf n, s_flags n (eBSDFType bits) for Sample(wo, u). */
int  jp_bsdf(JpContext* ctx, const JpBsdfDesc* desc, int32_t n, const float* normal, const float* wo, const float* wi, const float* u,
             float* f_eval, float* pdf_eval, float* s_f, float* s_wi, float* s_pdf, int32_t* s_flags);

/* test hook: closest-hit query for n rays (FScene::Intersect, scene.cc:25-33).  Host arrays:
 * origin/dir 3n floats, tmin/tmax n floats -> hit (0/1), t (ray.max_t after the call), prim (-1 if none),
 * normal 3n floats (FIntersection::normal).  Runs the same device traversal the render uses. */
int  jp_trace(JpContext* ctx, int32_t n, const float* origin, const float* dir, const float* tmin, const float* tmax,
              int32_t* hit, float* t, int32_t* prim, float* normal);

#ifdef __cplusplus
}
#endif
#endif
