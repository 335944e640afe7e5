import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
hb = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh); sp = hb.flatten()
ctx = jp.Context(0); ctx.upload(sp)
b = 17
for spp in (1, 4, 32):
    p = jp.render_params(W, Hh, spp, shard_index=b, shard_count=30)
    film = ctx.render(p)
    ref, cnt = H.oracle_render(sp, p, len(os.sched_getaffinity(0)))
    g = film[b*20:b*20+20]; r = ref[b*20:b*20+20]
    d = np.sqrt(((g - r) ** 2).sum(-1))
    c = ctx.counters()
    print("spp", spp, "mean L2 %.3e" % d.mean(), "exact px %.4f" % (g == r).all(-1).mean(), "px with d>1e-3: %d of %d" % ((d > 1e-3).sum(), d.size), "max %.3f" % d.max(),
          "rays gpu %d oracle %d shadow gpu %d oracle %d" % (c.closest_rays, cnt.closest_rays, c.shadow_rays, cnt.shadow_rays), flush=True)
    if spp == 1:
        ys, xs = np.nonzero(d > 1e-3)
        print("  sample of differing pixels (x,y,gpu,ref):", [(int(x), int(y) + b*20, g[y, x].round(4).tolist(), r[y, x].round(4).tolist()) for y, x in list(zip(ys, xs))[:8]])
        # clustering by x
        print("  x histogram of differing px:", np.histogram(xs, bins=8, range=(0, W))[0].tolist())
