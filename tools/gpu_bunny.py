"""Bunny-shaped scene (4 x 69,938 triangles + 2 rectangles, all materials, env light) on the GPU box: parity at low spp
against the oracle, throughput at 800x600."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
t0 = time.time()
hb = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh)
sp = hb.flatten()
print("scene built in %.1fs: prims %d nodes %d" % (time.time() - t0, sp.contents.n_primitives, sp.contents.n_bvh_nodes), flush=True)
ctx = jp.Context(0); t0 = time.time(); ctx.upload(sp); print("upload %.2fs" % (time.time() - t0), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "parity":
    p = jp.render_params(W, Hh, 2)
    film = ctx.render(p)
    t0 = time.time(); ref, cnt = H.oracle_render(sp, p, 16); print("oracle %.1fs" % (time.time() - t0))
    print("L2", float(np.sqrt(((film - ref) ** 2).sum(-1)).mean()), "exact px", float((film == ref).all(-1).mean()))
ctx.render(jp.render_params(W, Hh, 8))
for spp, prof in [(64, False), (32, True)]:
    ctx.set_profiling(prof)
    ctx.render(jp.render_params(W, Hh, spp))
    c = ctx.counters()
    print("bunny spp", spp, "ms %.2f" % c.render_ms, "Msamples/s %.1f" % (W * Hh * spp / c.render_ms / 1e3), "seg/sample %.3f shadow/sample %.3f" % (c.closest_rays / c.samples, c.shadow_rays / c.samples),
          "extend %.2f shade %.2f shadow %.2f other %.2f" % (c.extend_ms, c.shade_ms, c.shadow_ms, c.other_ms), flush=True)
