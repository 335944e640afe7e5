"""Test-side bindings: the oracle restatement (oracle/libjp_oracle.so) and, where /root/reference exists, the
compiled reference (oracle/_ref/libjp_ref.so).  TEST INFRASTRUCTURE: the product never imports this."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
import jet_pbrt_amd as jp  # noqa: E402
from jet_pbrt_amd import scenes  # noqa: E402

ORACLE_PATH = os.path.join(REPO, "oracle", "libjp_oracle.so")
REF_PATH = os.path.join(REPO, "oracle", "_ref", "libjp_ref.so")
GOLDEN = os.path.join(REPO, "tests", "golden")
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def ptr(a):
    return a.ctypes.data_as(_vp)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        L = C.CDLL(ORACLE_PATH)
        L.jp_oracle_render.argtypes = [C.POINTER(jp.JpScene), C.POINTER(jp.JpRenderParams), C.c_int, _vp, C.POINTER(jp.JpCounters)]
        L.jp_oracle_scene_new.restype = _vp
        L.jp_oracle_scene_new.argtypes = [C.POINTER(jp.JpScene)]
        L.jp_oracle_scene_free.argtypes = [_vp]
        L.jp_oracle_trace.argtypes = [_vp, C.c_int] + [_vp] * 9
        L.jp_oracle_tree_dump.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.jp_oracle_set_watertight.argtypes = [C.c_int]
        L.jp_oracle_camera_rays.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
        L.jp_oracle_bsdf.argtypes = [_vp, C.c_int, C.c_int] + [_vp] * 11
        L.jp_oracle_light_sample.argtypes = [_vp, C.c_int, C.c_int] + [_vp] * 7
        L.jp_oracle_li_scripted.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]
        L.jp_oracle_stock_stream.argtypes = [C.c_int, _vp]
        L.jp_oracle_bsdf_direct.argtypes = [C.POINTER(jp.JpBsdfDesc), C.c_int] + [_vp] * 10
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_PATH)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_PATH)
        L.ref_scene_new.restype = _vp
        L.ref_scene_new.argtypes = [C.c_char_p]
        L.ref_scene_free.argtypes = [_vp]
        L.ref_scene_camera.argtypes = [_vp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float]
        L.ref_scene_envlight.argtypes = [_vp, _fp]
        L.ref_scene_pointlight.argtypes = [_vp, _fp, _fp]
        L.ref_scene_dirlight.argtypes = [_vp, _fp, _fp]
        L.ref_mat_matte.argtypes = [_vp, _fp]
        L.ref_mat_mirror.argtypes = [_vp, _fp]
        L.ref_mat_glass.argtypes = [_vp, C.c_float, _fp, _fp]
        L.ref_mat_plastic.argtypes = [_vp, _fp, _fp, C.c_float, C.c_int]
        L.ref_mat_metal.argtypes = [_vp, _fp, _fp, C.c_float, C.c_float, C.c_int]
        L.ref_scene_mesh.argtypes = [_vp, C.c_char_p, C.c_int, C.c_int, _fp, C.c_float, C.c_int, _fp]
        L.ref_scene_rect.argtypes = [_vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _fp]
        L.ref_scene_sphere.argtypes = [_vp, _fp, C.c_float, C.c_int, _fp]
        L.ref_scene_disk.argtypes = [_vp, _fp, _fp, C.c_float, C.c_int, _fp]
        L.ref_scene_preprocess.argtypes = [_vp]
        L.ref_num_primitives.argtypes = [_vp]
        L.ref_num_lights.argtypes = [_vp]
        L.ref_render.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, _vp]
        L.ref_render_recursive.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, _vp]
        L.ref_render_other.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, _vp]
        L.ref_counters_reset.argtypes = [_vp]
        L.ref_counters_get.argtypes = [_vp, _vp]
        L.ref_trace.argtypes = [_vp, C.c_int] + [_vp] * 9
        L.ref_camera_rays.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
        L.ref_bsdf.argtypes = [_vp, C.c_int, C.c_int] + [_vp] * 11
        L.ref_light_sample.argtypes = [_vp, C.c_int, C.c_int] + [_vp] * 7
        L.ref_li_scripted.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]
        L.ref_stock_stream.argtypes = [C.c_int, _vp]
        L.ref_stock_float2.argtypes = [_vp]
        L.ref_bsdf_direct.argtypes = [C.POINTER(jp.JpBsdfDesc), C.c_int] + [_vp] * 10
        L.ref_film_save.argtypes = [_vp, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.ref_gamma_encode.argtypes = [_vp, C.c_int, _vp]
        _ref = L
    return _ref


class RefBackend(scenes.HostBackend):
    """The same procedural scene API over the compiled, unmodified reference."""
    prefix = "ref_"

    def __init__(self, name="scene"):
        super().__init__(name, lib=ref_lib())

    def _new(self, name):
        self.h = C.c_void_p(self.L.ref_scene_new(name.encode()))

    rand_seed = 1

    def preprocess(self):
        """FScene::Preprocess (scene.cc:11-23) grows the tree from libc rand(): from the state a fresh reference process has, whatever this process did before
        (oracle_render; set rand_seed = None on the instance to leave the state alone)"""
        if self.rand_seed is not None:
            libc_srand(self.rand_seed)
        super().preprocess()

    def render(self, W, H, spp, maxdepth=5, sampler_mode=1, seed=1234, nthreads=8):
        film = np.zeros((H, W, 3), np.float32)
        st = self.L.ref_render(self.h, W, H, spp, maxdepth, sampler_mode, seed, nthreads, ptr(film))
        assert st == 0
        return film

    def render_recursive(self, W, H, spp, maxdepth=5, seed=1234):
        """the reference's FPathIntegratorRecursive (integrator.cc:233-307), counter sampler, serial"""
        film = np.zeros((H, W, 3), np.float32)
        st = self.L.ref_render_recursive(self.h, W, H, spp, maxdepth, seed, ptr(film))
        assert st == 0
        return film

    def render_other(self, kind, W, H, spp, maxdepth=5, seed=1234):
        """the reference's FWhittedIntegrator (kind 1) / FDebugIntegrator (kind 2), counter sampler, serial"""
        film = np.zeros((H, W, 3), np.float32)
        st = self.L.ref_render_other(self.h, kind, W, H, spp, maxdepth, seed, ptr(film))
        assert st == 0
        return film

    def counters(self):
        out = np.zeros(4, np.uint64)
        self.L.ref_counters_get(self.h, ptr(out))
        return out


def oracle_render(scene_ptr, params, nthreads=8, watertight=False, rand_seed=1):
    """watertight=True: the reference's arithmetic with a conservative box test (oracle/pt_oracle.cc box_hit), i.e. without
    the hits the reference's own BVH drops on finely tessellated meshes.
    rand_seed: the oracle rebuilds the reference's BVH, whose split axes come from libc rand() (bvh.h:61); every render starts from the state a fresh
    reference process has (srand(1)), so that a film on a mesh -- where the hits in the fp32 acceptance fringe depend on that tree -- does not depend on how many
    trees this process has built before -- nor on the HIP runtime: libamd_comgr calls srand() with a seed of its own and rand() whenever it loads a code object
    (tools/randtrace: 66 calls and one srand in the main thread of a GPU test run), which used to make every later un-seeded oracle tree a different one per run.  None: leave the state alone (tools that sweep it)."""
    film = np.zeros((params.height, params.width, 3), np.float32)
    cnt = jp.JpCounters()
    L = oracle_lib()
    if rand_seed is not None:
        libc_srand(rand_seed)
    L.jp_oracle_set_watertight(1 if watertight else 0)
    try:
        st = L.jp_oracle_render(scene_ptr, C.byref(params), nthreads, ptr(film), C.byref(cnt))
    finally:
        L.jp_oracle_set_watertight(0)
    assert st == 0
    return film, cnt


SCENES = {
    "cornell": lambda be, W, H: scenes.build_cornell(be, W, H, lambert_only=False),
    "cornell_lambert": lambda be, W, H: scenes.build_cornell(be, W, H, lambert_only=True),
    # the committed OBJ fixture (numpy's sin/cos may differ in the last bit between hosts; the goldens must not)
    "bunny_small": lambda be, W, H: scenes.build_bunny(be, W, H, obj_path=os.path.join(GOLDEN, "bunny_24x16.obj")),
    "misc": lambda be, W, H: scenes.build_misc(be, W, H),
    "lights": lambda be, W, H: scenes.build_lights(be, W, H),
    "disks": lambda be, W, H: scenes.build_disks(be, W, H),
}


def build_random_scene(be, W, Hh, seed, n_tris=120, tmpdir=None):
    """Random triangle soup + rectangles + spheres, every material kind, triangle / rectangle / sphere area lights and
    a coloured environment -- the same call sequence on any backend."""
    import tempfile
    rng = np.random.default_rng(seed)
    d = tmpdir or tempfile.mkdtemp(prefix="jp_rand_")
    be.camera((0, 0, 9), (0, 0, -1), (0, 1, 0), 55.0, W, Hh)
    be.envlight(tuple(float(v) for v in rng.uniform(0.0, 0.3, 3)))
    mats = [be.mat_matte(tuple(rng.uniform(0.1, 0.9, 3))), be.mat_matte(tuple(rng.uniform(0.1, 0.9, 3))),
            be.mat_metal(tuple(rng.uniform(0.1, 1.5, 3)), tuple(rng.uniform(0.1, 3.0, 3)), float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4)), bool(rng.integers(0, 2))),
            be.mat_plastic(tuple(rng.uniform(0.1, 0.6, 3)), tuple(rng.uniform(0.1, 0.4, 3)), float(rng.uniform(0.05, 0.5)), bool(rng.integers(0, 2))),
            be.mat_glass(float(rng.uniform(1.2, 1.8)), (0.95, 0.95, 0.95), (0.9, 0.95, 0.9)), be.mat_mirror((0.85, 0.85, 0.9))]
    # four soups with different materials
    for m in range(4):
        c = rng.uniform(-3, 3, (n_tris // 4, 1, 3)); v = (c + rng.normal(0, 0.7, (n_tris // 4, 3, 3))).reshape(-1, 3).astype(np.float32)
        f = np.arange(v.shape[0]).reshape(-1, 3)
        path = os.path.join(d, "soup_%d_%d.obj" % (seed, m))
        scenes.write_obj(path, v, f)
        be.mesh(path, bool(m & 1), bool(m & 2), (0, 0, -1.0 * m), 1.0, mats[m], None)
    # a mesh light (two triangles), a rectangle light and a sphere light
    lp = os.path.join(d, "light_%d.obj" % seed)
    scenes.write_obj(lp, np.array([[-1, 3.9, -1], [1, 3.9, -1], [1, 3.9, 1], [-1, 3.9, 1]], np.float32), np.array([[0, 1, 2], [0, 2, 3]]))
    be.mesh(lp, False, False, (0, 0, 0), 1.0, mats[0], tuple(rng.uniform(5, 20, 3)))
    be.rect(scenes.AXIS_YZ, -1, 1, -2, 0, -4.0, False, mats[1], tuple(rng.uniform(2, 8, 3)))
    be.sphere((2.5, 2.0, 1.0), 0.5, mats[0], tuple(rng.uniform(5, 15, 3)))
    be.sphere((-1.5, -1.0, 1.5), 0.8, mats[4], None)
    be.sphere((1.2, -1.5, 0.5), 0.6, mats[5], None)
    be.rect(scenes.AXIS_XZ, -6, 6, -6, 6, -3.0, False, mats[1], None)
    be.rect(scenes.AXIS_XY, -6, 6, -3, 5, -6.0, False, mats[3], None)
    be.preprocess()
    return be


def oracle_scene(L, scene_ptr, rand_seed=1):
    """jp_oracle_scene_new from the rand() state of a fresh reference process (see oracle_render)"""
    if rand_seed is not None:
        libc_srand(rand_seed)
    return L.jp_oracle_scene_new(scene_ptr)


def libc_srand(seed=1):
    """the reference BVH draws its split axes from libc rand() (bvh.h:61); reset it so that the compiled
    reference and the restatement build the same tree."""
    C.CDLL(None).srand(seed)


def bsdf_cases():
    """the by-value BSDFs of the KAT set (tests/golden/kat_bsdf.npz): every class of bsdf.h / microfacet.h, the ones no material builds included"""
    D = jp.bsdf_desc
    return {
        "lambert": D(jp.JP_BSDF_LAMBERT, color=(0.63, 0.065, 0.05)),
        "mirror": D(jp.JP_BSDF_MIRROR, color=(0.9, 0.85, 0.8)),
        "fresnel_specular": D(jp.JP_BSDF_FRESNEL_SPECULAR, color=(0.98, 0.97, 0.96), color2=(0.95, 0.96, 0.97), eta_a=1.0, eta_b=1.5),
        "phong_20": D(jp.JP_BSDF_PHONG, color=(0.7, 0.6, 0.5), exponent=20.0),
        "phong_3": D(jp.JP_BSDF_PHONG, color=(0.2, 0.9, 0.4), exponent=3.0),
        "tr_conductor": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(1, 1, 1), distribution=0, alpha=(0.2, 0.2), fresnel=0, fr_eta_i=(1, 1, 1), fr_eta_t=(0.18, 0.15, 0.81), fr_k=(0.11, 0.11, 0.11)),
        "tr_noop_aniso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.8, 0.7, 0.9), distribution=0, alpha=(0.15, 0.4), fresnel=2),
        "tr_full_iso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.8, 0.8, 0.8), distribution=0, alpha=(0.3, 0.3), sample_visible=False, fresnel=1, fr_eta_i=(1.0,) * 3, fr_eta_t=(1.5,) * 3),
        "tr_full_aniso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.8, 0.8, 0.8), distribution=0, alpha=(0.1, 0.35), sample_visible=False, fresnel=2),
        "beck_conductor": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(1, 1, 1), distribution=1, alpha=(0.25, 0.25), fresnel=0, fr_eta_i=(1.2, 1.1, 1.0), fr_eta_t=(0.2, 0.9, 1.1), fr_k=(3.9, 2.4, 2.2)),
        "beck_dielectric_aniso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.9, 0.9, 0.9), distribution=1, alpha=(0.12, 0.3), fresnel=1, fr_eta_i=(1.5,) * 3, fr_eta_t=(1.0,) * 3),
        "beck_full_iso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.7, 0.7, 0.7), distribution=1, alpha=(0.2, 0.2), sample_visible=False, fresnel=2),
        "beck_full_aniso": D(jp.JP_BSDF_MICROFACET_REFLECTION, color=(0.7, 0.7, 0.7), distribution=1, alpha=(0.4, 0.1), sample_visible=False, fresnel=2),
        "trans_tr": D(jp.JP_BSDF_MICROFACET_TRANSMISSION, color=(0.95, 0.9, 0.85), distribution=0, alpha=(0.2, 0.2), eta_a=1.0, eta_b=1.5),
        "trans_beck": D(jp.JP_BSDF_MICROFACET_TRANSMISSION, color=(0.9, 0.95, 1.0), distribution=1, alpha=(0.3, 0.15), eta_a=1.0, eta_b=1.33),
        "trans_tr_full": D(jp.JP_BSDF_MICROFACET_TRANSMISSION, color=(1, 1, 1), distribution=0, alpha=(0.25, 0.25), sample_visible=False, eta_a=1.5, eta_b=1.0),
    }


def bsdf_inputs(n, seed):
    """shading events: unit normals, wo / wi spread over both hemispheres (a third of the wi on the far side: transmission), grazing and
    normal incidence included, u in [0, 1)^2 with a few extreme values"""
    rng = np.random.default_rng(seed)
    def unit(v): return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    nrm = unit(rng.normal(size=(n, 3)))
    wo = unit(rng.normal(size=(n, 3))); wi = unit(rng.normal(size=(n, 3)))
    same = rng.random(n) < 0.66
    so = np.sign((wo * nrm).sum(1)); si = np.sign((wi * nrm).sum(1))
    flip = np.where(same, so * si < 0, so * si > 0)
    wi = np.where(flip[:, None], wi - 2 * (wi * nrm).sum(1, keepdims=True) * nrm, wi).astype(np.float32)
    wo[:8] = unit(nrm[:8] + 1e-3 * rng.normal(size=(8, 3)))            # normal incidence
    u = rng.random((n, 2)).astype(np.float32)
    u[:4] = [[0.0, 0.0], [0.999999, 0.5], [0.5, 0.999999], [1e-7, 0.25]]
    return nrm, unit(wo), unit(wi), u


def run_bsdf(fn, desc, nrm, wo, wi, u):
    n = nrm.shape[0]
    out = dict(f=np.zeros((n, 3), np.float32), pdf=np.zeros(n, np.float32), sf=np.zeros((n, 3), np.float32), swi=np.zeros((n, 3), np.float32),
               spdf=np.zeros(n, np.float32), sflags=np.zeros(n, np.int32))
    fn(C.byref(desc), n, ptr(nrm), ptr(wo), ptr(wi), ptr(u), ptr(out["f"]), ptr(out["pdf"]), ptr(out["sf"]), ptr(out["swi"]), ptr(out["spdf"]), ptr(out["sflags"]))
    return out
