"""CPU suite, part 2 (only where /root/reference exists and oracle/_ref was built): the restatement against the
LIVE compiled reference on configurations the committed goldens do not hold."""
import numpy as np
import pytest

import harness as Hm

pytestmark = pytest.mark.skipif(not Hm.have_ref(), reason="oracle/_ref/libjp_ref.so not built (needs /root/reference)")


@pytest.mark.parametrize("name,W,Hh,spp,depth,seed", [
    ("cornell", 40, 30, 3, 5, 7), ("cornell_lambert", 33, 47, 2, 2, 99), ("bunny_small", 64, 40, 4, 5, 1234),
    ("misc", 56, 56, 6, 8, 5), ("cornell", 16, 16, 32, 0, 3), ("misc", 20, 61, 5, 1, 11), ("lights", 48, 40, 4, 5, 21),
    ("disks", 64, 64, 8, 5, 31), ("disks", 37, 53, 3, 2, 8)])
def test_live_reference_equality(H, name, W, Hh, spp, depth, seed):
    H.libc_srand(1)
    rb = H.SCENES[name](H.RefBackend(name), W, Hh)
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
    sp = hb.flatten()
    for mode in (0, 1):
        fr = rb.render(W, Hh, spp, depth, mode, seed, 4)
        H.libc_srand(1)
        fo, _ = H.oracle_render(sp, H.jp.render_params(W, Hh, spp, depth, seed, mode), 4)
        assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32)), (name, mode, float(np.abs(fr - fo).max()))
    # the serial whole-frame path of Render (numthreads < 1)
    fr = rb.render(W, Hh, 1, depth, 0, seed, 0)
    H.libc_srand(1)
    fo, _ = H.oracle_render(sp, H.jp.render_params(W, Hh, 1, depth, seed, 0), 0)
    assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32))


def test_bunny_mesh_resolution_sweep(H):
    """larger procedural meshes: reference ingest (obj_loader) vs own OBJ reader + topology-independent hits."""
    W, Hh = 48, 36
    for nlon, nlat in ((40, 30), (61, 44)):
        H.libc_srand(1)
        rb = H.scenes.build_bunny(H.RefBackend("b"), W, Hh, n_lon=nlon, n_lat=nlat)
        hb = H.scenes.build_bunny(H.scenes.HostBackend("b"), W, Hh, n_lon=nlon, n_lat=nlat)
        assert rb.num_primitives() == hb.num_primitives() == 4 * 2 * nlon * (nlat - 1) + 2
        fr = rb.render(W, Hh, 2, 5, 1, 1234, 4)
        H.libc_srand(1)
        fo, _ = H.oracle_render(hb.flatten(), H.jp.render_params(W, Hh, 2, 5, 1234, 1), 4)
        assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scenes_live_reference(H, tmp_path, seed):
    """random soups + all shape kinds as lights + all materials: the restatement stays bit-identical to the reference"""
    W, Hh, spp = 40, 32, 4
    H.libc_srand(1)
    rb = H.build_random_scene(H.RefBackend("r"), W, Hh, seed, tmpdir=str(tmp_path))
    hb = H.build_random_scene(H.scenes.HostBackend("r"), W, Hh, seed, tmpdir=str(tmp_path))
    assert rb.num_primitives() == hb.num_primitives() and rb.num_lights() == hb.num_lights() == 5
    for mode in (0, 1):
        fr = rb.render(W, Hh, spp, 5, mode, 77, 4)
        H.libc_srand(1)
        fo, _ = H.oracle_render(hb.flatten(), H.jp.render_params(W, Hh, spp, 5, 77, mode), 4)
        assert np.isfinite(fr).all()
        assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32)), float(np.abs(fr - fo).max())


@pytest.mark.parametrize("name", ["cornell", "misc", "lights", "disks", "bunny_small"])
@pytest.mark.parametrize("kind", [1, 2])
def test_other_integrators_live_reference(H, name, kind):
    """FWhittedIntegrator (integrator.cc:115-220: two branches at every mirror, draws in depth-first order) and
    FDebugIntegrator (integrator.h:44-58): the restatement is bit-identical to the compiled reference"""
    W, Hh, spp = 40, 36, 3
    H.libc_srand(1)
    rb = H.SCENES[name](H.RefBackend(name), W, Hh)
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
    for depth in (5, 2):
        fr = rb.render_other(kind, W, Hh, spp, depth, 55)
        H.libc_srand(1)
        fo, _ = H.oracle_render(hb.flatten(), H.jp.render_params(W, Hh, spp, depth, 55, integrator=kind), 0)
        assert np.isfinite(fr).all()
        assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32)), (name, kind, depth, float(np.abs(fr - fo).max()))
