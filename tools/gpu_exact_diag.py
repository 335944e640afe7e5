"""With the libm-exact sincosf: is the GPU film bit-identical to the oracle's?"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
ctx = jp.Context(0)
for name, W, Hh, spp in (("cornell", 128, 128, 64), ("cornell_lambert", 128, 128, 64), ("misc", 128, 128, 64), ("lights", 128, 128, 64), ("bunny_small", 160, 120, 32)):
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh); sp = hb.flatten()
    ctx.upload(sp)
    p = jp.render_params(W, Hh, spp, 5, 77)
    film = ctx.render(p)
    H.libc_srand(1)
    ref, cnt = H.oracle_render(sp, p, 16)
    c = ctx.counters()
    d = np.sqrt(((film - ref) ** 2).sum(-1))
    print(name, "libm mode", ctx.build_info().libm_sincosf, "bit-identical:", np.array_equal(film.view(np.uint32), ref.view(np.uint32)), "exact px %.5f" % (film == ref).all(-1).mean(), "mean L2 %.2e" % d.mean(),
          "rays gpu/oracle %d/%d shadow %d/%d" % (c.closest_rays, cnt.closest_rays, c.shadow_rays, cnt.shadow_rays), flush=True)
for seed in (1, 2, 3, 4):
    hb = H.build_random_scene(H.scenes.HostBackend("r"), 96, 80, seed, n_tris=400); sp = hb.flatten()
    ctx.upload(sp); p = jp.render_params(96, 80, 16, 5, 5 + seed)
    film = ctx.render(p); H.libc_srand(1); ref, cnt = H.oracle_render(sp, p, 16)
    print("random", seed, "bit-identical:", np.array_equal(film.view(np.uint32), ref.view(np.uint32)), "exact px %.5f" % (film == ref).all(-1).mean(), "mean L2 %.2e" % np.sqrt(((film - ref) ** 2).sum(-1)).mean(), flush=True)
