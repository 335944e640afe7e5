"""CPU suite, part 1: the oracle restatement (oracle/pt_oracle.cc) against the golden vectors generated from the
compiled, unmodified reference (tests/golden/make_golden.py).  Everything here is BIT-EXACT: same compiler,
same libm, same operation order."""
import json
import os

import numpy as np
import pytest

W = Hh = 48
SPP = 8
SCENE_NAMES = ["cornell", "cornell_lambert", "bunny_small", "misc", "lights"]


def _scene(H, name):
    H.libc_srand(1)
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
    return hb, hb.flatten()


@pytest.mark.parametrize("name", SCENE_NAMES)
@pytest.mark.parametrize("mode,tag", [(0, "stock"), (1, "counter")])
def test_film_bit_exact(H, name, mode, tag):
    """Tier T0 (stock mt19937_64 stream, per-band reseed) and T1 (counter stream) whole-film equality."""
    hb, sp = _scene(H, name)
    gold = np.load(os.path.join(H.GOLDEN, "film_%s_%s.npy" % (name, tag)))
    H.libc_srand(1)
    film, cnt = H.oracle_render(sp, H.jp.render_params(W, Hh, SPP, 5, 1234, mode), 4)
    assert film.shape == gold.shape
    assert np.array_equal(film.view(np.uint32), gold.view(np.uint32)), "max diff %g" % np.abs(film - gold).max()
    counts = json.load(open(os.path.join(H.GOLDEN, "counts.json")))["%s_%s" % (name, tag)]
    assert [cnt.closest_rays, cnt.closest_hits, cnt.shadow_rays, cnt.shadow_occluded] == counts


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_thread_count_and_serial_path(H, name):
    """Per-band samplers make the film independent of the thread count (integrator.cc:66); the serial
    numthreads<1 path (integrator.cc:45-50) uses ONE stream and therefore differs for the stock sampler only."""
    hb, sp = _scene(H, name)
    p0 = H.jp.render_params(W, Hh, 2, 5, 1234, 0)
    a, _ = H.oracle_render(sp, p0, 1)
    b, _ = H.oracle_render(sp, p0, 7)
    assert np.array_equal(a, b)
    s, _ = H.oracle_render(sp, p0, 0)
    assert np.array_equal(s[:20], a[:20]) and not np.array_equal(s[20:], a[20:])      # first band shares the stream start
    p1 = H.jp.render_params(W, Hh, 2, 5, 1234, 1)
    c, _ = H.oracle_render(sp, p1, 3)
    d, _ = H.oracle_render(sp, p1, 0)
    assert np.array_equal(c, d)                                                         # counter stream: order-free


def test_stock_stream(H, kat):
    out = np.zeros(4096, np.float32)
    H.oracle_lib().jp_oracle_stock_stream(4096, H.ptr(out))
    assert np.array_equal(out, kat["stock_stream"])
    # g++ hands the FIRST draw to .y (sampler.h:49-52)
    assert kat["stock_float2"][1] == kat["stock_stream"][0] and kat["stock_float2"][0] == kat["stock_stream"][1]
    assert abs(float(out[0]) - 0.947231591) < 1e-7


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_camera_trace_light_li_kats(H, kat, name):
    hb, sp = _scene(H, name)
    L = H.oracle_lib()
    oh = H.oracle_scene(L, sp)
    try:
        n = kat[name + "_cam_pxy"].shape[0]
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
        pxy = np.ascontiguousarray(kat[name + "_cam_pxy"])
        L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
        assert np.array_equal(o, kat[name + "_cam_o"]) and np.array_equal(d, kat[name + "_cam_d"])
        for tag, oo, dd, tm in (("tr1", o, d, np.full(n, np.inf, np.float32)),
                                ("tr2", np.ascontiguousarray(kat[name + "_tr2_o"]), np.ascontiguousarray(kat[name + "_tr2_d"]), np.ascontiguousarray(kat[name + "_tr2_tmax"]))):
            tmin = np.full(n, 0.001, np.float32)
            hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
            L.jp_oracle_trace(oh, n, H.ptr(oo), H.ptr(dd), H.ptr(tmin), H.ptr(tm), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
            assert np.array_equal(hit, kat["%s_%s_hit" % (name, tag)])
            assert np.array_equal(t, kat["%s_%s_t" % (name, tag)])
            assert np.array_equal(prim, kat["%s_%s_prim" % (name, tag)])
            assert np.array_equal(nrm, kat["%s_%s_nrm" % (name, tag)]) and np.array_equal(pos, kat["%s_%s_pos" % (name, tag)])
            assert hit.sum() > 16
        p = np.ascontiguousarray(kat[name + "_ls_p"]); nn = np.ascontiguousarray(kat[name + "_ls_n"]); u2 = np.ascontiguousarray(kat[name + "_ls_u"])
        for li in range(int(kat[name + "_nlights"][0])):
            lp = np.zeros((n, 3), np.float32); wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); Li = np.zeros((n, 3), np.float32)
            L.jp_oracle_light_sample(oh, n, li, H.ptr(p), H.ptr(nn), H.ptr(u2), H.ptr(lp), H.ptr(wi), H.ptr(pdf), H.ptr(Li))
            assert np.array_equal(lp, kat["%s_ls%d_pos" % (name, li)])
            assert np.array_equal(pdf, kat["%s_ls%d_pdf" % (name, li)])
            assert np.array_equal(Li, kat["%s_ls%d_Li" % (name, li)])
            live = ~((Li == 0).all(1))
            assert np.array_equal(wi[live], kat["%s_ls%d_wi" % (name, li)][live])
        vals = np.ascontiguousarray(kat[name + "_li_vals"]); ppx = np.ascontiguousarray(kat[name + "_li_pxy"])
        out = np.zeros((n, 3), np.float32)
        L.jp_oracle_li_scripted(oh, n, 5, H.ptr(ppx), H.ptr(vals), vals.shape[1], H.ptr(out))
        assert np.array_equal(out.view(np.uint32), kat[name + "_li_out"].view(np.uint32))
        assert (out.sum(1) > 0).mean() > 0.3
    finally:
        L.jp_oracle_scene_free(oh)


def test_bsdf_kats(H, kat):
    """FBSDF::Evalf / Sample of every material kind (matte, metal, glass, mirror, remapped plastic) on random frames."""
    hb, sp = _scene(H, "misc")
    L = H.oracle_lib()
    oh = H.oracle_scene(L, sp)
    try:
        nn = np.ascontiguousarray(kat["bsdf_n"]); wo = np.ascontiguousarray(kat["bsdf_wo"]); wi = np.ascontiguousarray(kat["bsdf_wi"])
        u2 = np.ascontiguousarray(kat["bsdf_u2"]); us = np.ascontiguousarray(kat["bsdf_us"]); n = nn.shape[0]
        kinds = set()
        for m in range(int(kat["bsdf_nmat"][0])):
            fe = np.zeros((n, 3), np.float32); sf = np.zeros((n, 3), np.float32); swi = np.zeros((n, 3), np.float32)
            spdf = np.zeros(n, np.float32); sfl = np.zeros(n, np.int32); sd = np.zeros(n, np.int32)
            L.jp_oracle_bsdf(oh, n, m, H.ptr(nn), H.ptr(wo), H.ptr(wi), H.ptr(u2), H.ptr(us), H.ptr(fe), H.ptr(sf), H.ptr(swi), H.ptr(spdf), H.ptr(sfl), H.ptr(sd))
            for k, v in dict(feval=fe, sf=sf, swi=swi, spdf=spdf, sflags=sfl, delta=sd).items():
                g = kat["bsdf%d_%s" % (m, k)]
                assert np.array_equal(v.view(np.uint32), g.view(np.uint32)), "material %d %s" % (m, k)
            kinds.update(np.unique(sfl).tolist())
        assert {9, 17, 5, 6}.issubset(kinds)     # diffuse-reflection, glossy-reflection, specular reflection / transmission
    finally:
        L.jp_oracle_scene_free(oh)


@pytest.mark.parametrize("name", ["cornell", "misc", "lights"])
def test_recursive_integrator_is_the_same_estimator(H, name):
    """FPathIntegratorRecursive (integrator.cc:233-307) vs FPathIntegratorIteration, both run by the UNMODIFIED reference on
    the same counter stream (committed goldens): same draws, same decisions -- the films differ in the rounding of the
    throughput products only.  The host layer therefore serves FPathIntegratorRecursive with the iterative kernels."""
    it = np.load(os.path.join(H.GOLDEN, "film_%s_counter.npy" % name))
    rec = np.load(os.path.join(H.GOLDEN, "film_%s_counter_recursive.npy" % name))
    d = np.sqrt(((it - rec) ** 2).sum(-1))
    assert d.mean() < 1e-7 and d.max() < 1e-6 and not np.array_equal(it, rec)
    hb = H.SCENES[name](H.scenes.HostBackend(name), 48, 48)
    ref, _ = H.oracle_render(hb.flatten(), H.jp.render_params(48, 48, 8, 5, 1234), 4)
    assert np.sqrt(((ref - rec) ** 2).sum(-1)).mean() < 1e-6


@pytest.mark.parametrize("name", ["cornell", "cornell_lambert"])
def test_baseline_config0_cpu_plumbing_at_its_size(H, name):
    """BASELINE.json configs[0]: cornell_box 256x256, 16 spp on the CPU path through parallel.cc (20-row tasks, stock FRandomSampler
    per task, 16 threads as main.cc:156).  The oracle's film equals the compiled reference's bit for bit: SHA-256 of the raw fp32
    film, four film rows and the ray statistics from tests/golden/make_golden_config0.py."""
    import hashlib
    g = json.load(open(os.path.join(H.GOLDEN, "config0.json")))[name]
    Wc, Hc, spp = g["width"], g["height"], g["spp"]
    H.libc_srand(1)
    hb = H.SCENES[name](H.scenes.HostBackend(name), Wc, Hc)
    film, cnt = H.oracle_render(hb.flatten(), H.jp.render_params(Wc, Hc, spp, 5, 1234, 0), 16)
    for y, row in g["rows"].items():
        assert np.array_equal(film[int(y)].reshape(-1), np.array(row, np.float32)), "row %s differs" % y
    assert hashlib.sha256(np.ascontiguousarray(film).tobytes()).hexdigest() == g["sha256"]
    assert [cnt.closest_rays, cnt.closest_hits, cnt.shadow_rays, cnt.shadow_occluded] == g["counts"]
    assert 3.3 < cnt.closest_rays / (Wc * Hc * spp) < 3.6                                     # 3.45 closest-hit rays per sample


def test_reflection_api_by_value_bit_exact(H):
    """every BSDF class of bsdf.h / bsdf.cc / microfacet.cc constructed directly -- FPhongSpecularReflection,
    BeckmannDistribution (both sampling branches), FMicrofacetTransmission (with the world-space Pdf call of bsdf.cc:141 on local
    vectors), FresnelNoOp, general conductor / dielectric Fresnel terms, next to the closures the materials build -- Evalf / Pdf /
    Sample of the oracle restatement against the compiled reference's outputs (tests/golden/make_golden_bsdf.py): BIT-EXACT."""
    g = np.load(os.path.join(H.GOLDEN, "kat_bsdf.npz"))
    nrm, wo, wi, u = H.bsdf_inputs(384, 77)
    for k, v in (("in_nrm", nrm), ("in_wo", wo), ("in_wi", wi), ("in_u", u)):
        assert np.array_equal(v.view(np.uint32), g[k].view(np.uint32)), "input generator drifted: " + k
    L = H.oracle_lib()
    for name, desc in H.bsdf_cases().items():
        r = H.run_bsdf(L.jp_oracle_bsdf_direct, desc, nrm, wo, wi, u)
        for k, v in r.items():
            want = g["%s__%s" % (name, k)]
            same = np.array_equal(v.view(np.uint32), want.view(np.uint32)) if v.dtype == np.float32 else np.array_equal(v, want)
            assert same, "%s.%s: %d of %d differ, max abs %g" % (name, k, int((v != want).sum()), v.size, float(np.nanmax(np.abs(v.astype(np.float64) - want))))
