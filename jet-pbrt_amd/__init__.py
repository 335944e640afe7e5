"""jet-pbrt_amd -- MI355X-native path-tracing integrator behind jet-pbrt's Render() seam.

Python here is plumbing only (ctypes bindings for tests / bench / smoke): the product is
  * csrc/libjetpbrt_amd.so   hand-written HIP wavefront kernels + the C ABI of include/jetpbrt_amd.h
  * host/libjetpbrt_host.so  C++ mirror of the reference's FScene/FCamera/FFilm/FSampler/FMaterial/
                             FIntegrator::Render API, BVH builder and scene flattener.
Nothing in this package imports, links or calls anything under oracle/ (test infrastructure).
The directory name has a hyphen; import it as `jet_pbrt_amd` (shim module at the repo root).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.environ.get("JETPBRT_AMD_LIB") or os.path.join(PKG_DIR, "csrc", "libjetpbrt_amd.so")   # env override: A/B of kernel builds
HOST_LIB_PATH = os.path.join(PKG_DIR, "host", "libjetpbrt_host.so")
CLI_PATH = os.path.join(PKG_DIR, "host", "jetpbrt")

JP_MAT_PARAM_STRIDE = 16
JP_SAMPLER_STOCK_MT19937, JP_SAMPLER_COUNTER, JP_SAMPLER_DEBUG = 0, 1, 2
JP_OK = 0

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class JpCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("res_x", C.c_float), ("res_y", C.c_float)]


class JpScene(C.Structure):
    _fields_ = [
        ("camera", JpCamera),
        ("n_triangles", C.c_int32), ("tri_p0", _fp), ("tri_p1", _fp), ("tri_p2", _fp), ("tri_n", _fp),
        ("n_rectangles", C.c_int32), ("rect_p0", _fp), ("rect_p1", _fp), ("rect_p2", _fp), ("rect_p3", _fp), ("rect_n", _fp),
        ("n_spheres", C.c_int32), ("sph_center", _fp), ("sph_radius", _fp),
        ("n_primitives", C.c_int32), ("prim_shape_type", _ip), ("prim_shape_index", _ip), ("prim_material", _ip), ("prim_light", _ip),
        ("n_materials", C.c_int32), ("mat_type", _ip), ("mat_params", _fp),
        ("n_lights", C.c_int32), ("light_type", _ip), ("light_radiance", _fp), ("light_prim", _ip), ("light_vec", _fp),
        ("world_radius", C.c_float),
        ("n_bvh_nodes", C.c_int32), ("bvh_bounds", _fp), ("bvh_left", _ip), ("bvh_right", _ip),
        ("n_bvh_prim_indices", C.c_int32), ("bvh_prim_index", _ip),
        ("bvh_reference_semantics", C.c_int32),
        ("n_disks", C.c_int32), ("disk_center", _fp), ("disk_normal", _fp), ("disk_radius", _fp),
    ]


class JpRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("sampler_mode", C.c_int32), ("seed", C.c_uint32),
                ("band_rows", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("integrator", C.c_int32)]


class JpCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("closest_hits", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("shadow_occluded", C.c_uint64),
                ("render_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("shadow_ms", C.c_double),
                ("other_ms", C.c_double),
                ("extend_launches", C.c_uint64), ("shade_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("path_ms", C.c_double), ("path_launches", C.c_uint64), ("certified_fallback_rays", C.c_uint64)]


class JpBuildInfo(C.Structure):
    _fields_ = [("built_on_device", C.c_int32), ("traversal_mode", C.c_int32), ("bvh_nodes", C.c_int32), ("bvh_height", C.c_int32),
                ("device_build_ms", C.c_double), ("libm_sincosf", C.c_int32), ("lanes_last_render", C.c_int32),
                ("fused_last_render", C.c_int32), ("fused_region", C.c_int32), ("fused_workgroups", C.c_int32), ("q4_nodes", C.c_int32), ("libm_xbsdf", C.c_int32), ("certified_walk", C.c_int32), ("certified_nodes", C.c_int32), ("certified_eye_leaves", C.c_int32)]


JP_INTEGRATOR_PATH, JP_INTEGRATOR_WHITTED, JP_INTEGRATOR_DEBUG_NORMAL = 0, 1, 2
JP_ABI_VERSION = 7      # include/jetpbrt_amd.h; hip_lib() refuses a library built for another ABI


class JpOptions(C.Structure):
    """ABI 7: the switches of the library by value (include/jetpbrt_amd.h: JpOptions); 0 = default, tri-state switches 1 on / -1 off"""
    _fields_ = [("struct_bytes", C.c_int32), ("lanes", C.c_int32), ("lane_rows", C.c_int32), ("blocks_per_cu", C.c_int32), ("max_slots", C.c_int64),
                ("compact_regions", C.c_int32), ("fused", C.c_int32), ("fused_region", C.c_int32), ("fused_job_spp", C.c_int32), ("fused_workgroups", C.c_int32),
                ("traversal", C.c_int32), ("q4", C.c_int32), ("q4_shadow", C.c_int32), ("persist", C.c_int32), ("vote", C.c_int32), ("stack_lds_words", C.c_int32),
                ("shade_sort", C.c_int32), ("device_tree", C.c_int32), ("device_wide", C.c_int32), ("bvh_max_leaf", C.c_int32), ("ploc_radius", C.c_int32), ("ploc_max_rounds", C.c_int32),
                ("certified", C.c_int32), ("cert_slack", C.c_float), ("cert_slack_eye", C.c_float), ("cert_eye_tau", C.c_float),
                ("libm_sincosf", C.c_int32), ("libm_xbsdf", C.c_int32), ("trace_walk", C.c_int32), ("box_pad", C.c_float), ("reserved", C.c_int32 * 8)]


class JpBsdfDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("color", C.c_float * 3), ("color2", C.c_float * 3), ("eta_a", C.c_float), ("eta_b", C.c_float),
                ("distribution", C.c_int32), ("alpha_x", C.c_float), ("alpha_y", C.c_float), ("sample_visible", C.c_int32), ("fresnel", C.c_int32),
                ("fr_eta_i", C.c_float * 3), ("fr_eta_t", C.c_float * 3), ("fr_k", C.c_float * 3), ("exponent", C.c_float)]


JP_BSDF_LAMBERT, JP_BSDF_MIRROR, JP_BSDF_FRESNEL_SPECULAR, JP_BSDF_MICROFACET_REFLECTION, JP_BSDF_MICROFACET_TRANSMISSION, JP_BSDF_PHONG = range(6)
JP_DIST_TROWBRIDGE_REITZ, JP_DIST_BECKMANN = 0, 1
JP_FRESNEL_CONDUCTOR, JP_FRESNEL_DIELECTRIC, JP_FRESNEL_NOOP = 0, 1, 2


def bsdf_desc(kind, color=(1, 1, 1), color2=(1, 1, 1), eta_a=1.0, eta_b=1.5, distribution=0, alpha=(0.2, 0.2), sample_visible=True, fresnel=0,
              fr_eta_i=(1, 1, 1), fr_eta_t=(1.5, 1.5, 1.5), fr_k=(0, 0, 0), exponent=10.0):
    d = JpBsdfDesc()
    d.kind = kind; d.color[:] = color; d.color2[:] = color2; d.eta_a = eta_a; d.eta_b = eta_b; d.distribution = distribution
    d.alpha_x, d.alpha_y = alpha; d.sample_visible = 1 if sample_visible else 0; d.fresnel = fresnel
    d.fr_eta_i[:] = fr_eta_i; d.fr_eta_t[:] = fr_eta_t; d.fr_k[:] = fr_k; d.exponent = exponent
    return d


def render_params(width, height, spp, max_depth=5, seed=1234, sampler_mode=JP_SAMPLER_COUNTER,
                  band_rows=20, shard_index=0, shard_count=1, integrator=JP_INTEGRATOR_PATH):
    return JpRenderParams(width, height, spp, max_depth, sampler_mode, seed, band_rows, shard_index, shard_count, integrator)


class JetPbrtError(RuntimeError):
    pass


_host = None
_hip = None


def host_lib():
    """libjetpbrt_host.so (C++ host mirror + flattener).  Raises if it has not been built."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise JetPbrtError("host library missing: %s (run __graft_entry__.build())" % HOST_LIB_PATH)
        L = C.CDLL(HOST_LIB_PATH)
        L.jp_host_scene_new.restype = C.c_void_p
        L.jp_host_scene_new.argtypes = [C.c_char_p]
        L.jp_host_scene_free.argtypes = [C.c_void_p]
        L.jp_host_last_error.restype = C.c_char_p
        L.jp_host_last_error.argtypes = [C.c_void_p]
        L.jp_host_scene_camera.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float]
        L.jp_host_scene_envlight.argtypes = [C.c_void_p, _fp]
        L.jp_host_scene_pointlight.argtypes = [C.c_void_p, _fp, _fp]
        L.jp_host_scene_dirlight.argtypes = [C.c_void_p, _fp, _fp]
        L.jp_host_mat_matte.argtypes = [C.c_void_p, _fp]
        L.jp_host_mat_mirror.argtypes = [C.c_void_p, _fp]
        L.jp_host_mat_glass.argtypes = [C.c_void_p, C.c_float, _fp, _fp]
        L.jp_host_mat_plastic.argtypes = [C.c_void_p, _fp, _fp, C.c_float, C.c_int]
        L.jp_host_mat_metal.argtypes = [C.c_void_p, _fp, _fp, C.c_float, C.c_float, C.c_int]
        L.jp_host_scene_mesh.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, _fp, C.c_float, C.c_int, _fp]
        L.jp_host_scene_rect.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _fp]
        L.jp_host_scene_sphere.argtypes = [C.c_void_p, _fp, C.c_float, C.c_int, _fp]
        L.jp_host_scene_disk.argtypes = [C.c_void_p, _fp, _fp, C.c_float, C.c_int, _fp]
        L.jp_host_scene_preprocess.argtypes = [C.c_void_p]
        L.jp_host_scene_set_device_build.argtypes = [C.c_void_p, C.c_int]
        L.jp_host_scene_set_reference_tree.argtypes = [C.c_void_p, C.c_int]
        L.jp_host_num_primitives.argtypes = [C.c_void_p]
        L.jp_host_num_lights.argtypes = [C.c_void_p]
        L.jp_host_flatten.restype = C.POINTER(JpScene)
        L.jp_host_flatten.argtypes = [C.c_void_p]
        L.jp_host_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.POINTER(JpCounters)]
        L.jp_host_render_other.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
        L.jp_host_save_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.jp_host_render_ldr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        L.jp_host_gamma_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.jp_host_bsdf_class.argtypes = [C.c_int, _fp, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp]
        L.jp_host_render_sampler.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _host = L
    return _host


def hip_lib():
    """libjetpbrt_amd.so (HIP kernels + C ABI).  Raises loudly if missing: there is no CPU fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise JetPbrtError("HIP library missing: %s (run __graft_entry__.build()); the product has no CPU fallback" % HIP_LIB_PATH)
        L = C.CDLL(HIP_LIB_PATH)
        if L.jp_abi_version() != JP_ABI_VERSION:                 # (a stale or foreign build would be handed structs of the wrong size)
            raise JetPbrtError("%s implements ABI %d, this binding ABI %d: rebuild with __graft_entry__.build()" % (HIP_LIB_PATH, L.jp_abi_version(), JP_ABI_VERSION))
        L.jp_last_error.restype = C.c_char_p
        L.jp_set_options.argtypes = [C.c_void_p, C.POINTER(JpOptions)]
        L.jp_get_options.argtypes = [C.c_void_p, C.POINTER(JpOptions)]
        L.jp_create_context.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.jp_destroy_context.argtypes = [C.c_void_p]
        L.jp_upload_scene.argtypes = [C.c_void_p, C.POINTER(JpScene)]
        L.jp_render.argtypes = [C.c_void_p, C.POINTER(JpRenderParams), C.c_void_p]
        L.jp_render_device.argtypes = [C.c_void_p, C.POINTER(JpRenderParams), C.c_void_p, C.c_int]
        L.jp_synchronize.argtypes = [C.c_void_p]
        L.jp_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.jp_get_counters.argtypes = [C.c_void_p, C.POINTER(JpCounters)]
        L.jp_get_build_info.argtypes = [C.c_void_p, C.POINTER(JpBuildInfo)]
        L.jp_trace.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8
        L.jp_render_rgb8.argtypes = [C.c_void_p, C.POINTER(JpRenderParams), C.c_void_p, C.c_void_p]
        L.jp_gamma_thresholds.argtypes = [C.c_void_p]
        L.jp_bsdf.argtypes = [C.c_void_p, C.POINTER(JpBsdfDesc), C.c_int32] + [C.c_void_p] * 10
        _hip = L
    return _hip


class Context:
    """Thin RAII wrapper over JpContext for harness code."""

    def __init__(self, device_id=0):
        self.lib = hip_lib()
        self.h = C.c_void_p()
        self._check(self.lib.jp_create_context(device_id, C.byref(self.h)))

    def _check(self, st):
        if st != JP_OK:
            raise JetPbrtError("jetpbrt_amd status %d: %s" % (st, self.lib.jp_last_error().decode()))

    def get_options(self):
        o = JpOptions()
        self._check(self.lib.jp_get_options(self.h, C.byref(o)))
        return o

    def set_options(self, **kw):
        """jp_set_options: the options in force with the given fields changed (e.g. lanes=1, persist=-1); no arguments: back to the initial value.
        Traversal fields apply to the next upload, schedule fields to the next render."""
        if not kw:
            self._check(self.lib.jp_set_options(self.h, None))
            return
        o = self.get_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise JetPbrtError("JpOptions has no field %r" % k)
            setattr(o, k, v)
        o.struct_bytes = C.sizeof(JpOptions)
        self._check(self.lib.jp_set_options(self.h, C.byref(o)))

    def upload(self, scene_ptr):
        self._check(self.lib.jp_upload_scene(self.h, scene_ptr))

    def render(self, params):
        import numpy as np
        film = np.zeros((params.height, params.width, 3), np.float32)
        self._check(self.lib.jp_render(self.h, C.byref(params), film.ctypes.data_as(C.c_void_p)))
        return film

    def render_rgb8(self, params, with_film=False):
        """jp_render_rgb8: the film as 8-bit gamma-encoded RGB (H, W, 3) uint8 [, and the fp32 film]"""
        import numpy as np
        rgb8 = np.zeros((params.height, params.width, 3), np.uint8)
        film = np.zeros((params.height, params.width, 3), np.float32) if with_film else None
        self._check(self.lib.jp_render_rgb8(self.h, C.byref(params), rgb8.ctypes.data_as(C.c_void_p), film.ctypes.data_as(C.c_void_p) if with_film else None))
        return (rgb8, film) if with_film else rgb8

    def render_device(self, params, device_ptr, sync=False):
        self._check(self.lib.jp_render_device(self.h, C.byref(params), C.c_void_p(device_ptr), 1 if sync else 0))

    def synchronize(self):
        self._check(self.lib.jp_synchronize(self.h))

    def set_profiling(self, on):
        self._check(self.lib.jp_set_profiling(self.h, 1 if on else 0))

    def counters(self):
        c = JpCounters()
        self._check(self.lib.jp_get_counters(self.h, C.byref(c)))
        return c

    def build_info(self):
        b = JpBuildInfo()
        self._check(self.lib.jp_get_build_info(self.h, C.byref(b)))
        return b

    def trace(self, origin, direction, tmin, tmax):
        import numpy as np
        n = origin.shape[0]
        o = np.ascontiguousarray(origin, np.float32); d = np.ascontiguousarray(direction, np.float32)
        t0 = np.ascontiguousarray(tmin, np.float32); t1 = np.ascontiguousarray(tmax, np.float32)
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.lib.jp_trace(self.h, n, p(o), p(d), p(t0), p(t1), p(hit), p(t), p(prim), p(nrm)))
        return hit, t, prim, nrm

    def bsdf(self, desc, normal, wo, wi, u):
        """jp_bsdf: FBSDF::Evalf / Pdf / Sample of a by-value BSDF on the device -> dict of arrays"""
        import numpy as np
        n = normal.shape[0]
        a = [np.ascontiguousarray(x, np.float32) for x in (normal, wo, wi, u)]
        out = dict(f=np.zeros((n, 3), np.float32), pdf=np.zeros(n, np.float32), sf=np.zeros((n, 3), np.float32), swi=np.zeros((n, 3), np.float32),
                   spdf=np.zeros(n, np.float32), sflags=np.zeros(n, np.int32))
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        self._check(self.lib.jp_bsdf(self.h, C.byref(desc), n, p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(out["f"]), p(out["pdf"]), p(out["sf"]), p(out["swi"]), p(out["spdf"]), p(out["sflags"])))
        return out

    def close(self):
        if self.h:
            self.lib.jp_destroy_context(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import scenes  # noqa: E402,F401
from . import distributed  # noqa: E402,F401
