"""Device-built LBVH vs host-built SAH tree on the 280k-triangle scene: setup time and render throughput."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
ctx = jp.Context(0)
films = {}
for dev in (False, True, False, True):
    hb = H.scenes.HostBackend("bunny"); hb.set_device_build(dev)
    orig = hb.preprocess
    tp = [0.0]
    def timed(orig=orig, tp=tp):
        t = time.time(); orig(); tp[0] = time.time() - t
    hb.preprocess = timed
    H.scenes.build_bunny(hb, W, Hh)
    t = time.time(); sp = hb.flatten(); tf = time.time() - t
    t = time.time(); ctx.upload(sp); tu = time.time() - t
    bi = ctx.build_info()
    ctx.render(jp.render_params(W, Hh, 4))
    ctx.set_profiling(True)
    film = ctx.render(jp.render_params(W, Hh, 32)); c = ctx.counters()
    ctx.set_profiling(False)
    films[dev] = film
    print("device_build=%d preprocess %.3fs flatten %.3fs upload %.3fs (device build %.2f ms) height %d nodes %d mode %d | 32spp %.1f ms %.1f Msamples/s extend %.1f shade %.1f shadow %.1f"
          % (dev, tp[0], tf, tu, bi.device_build_ms, bi.bvh_height, bi.bvh_nodes, bi.traversal_mode, c.render_ms, W * Hh * 32 / c.render_ms / 1e3, c.extend_ms, c.shade_ms, c.shadow_ms), flush=True)
print("film L2 host-tree vs device-tree: %.3e" % float(np.sqrt(((films[False] - films[True]) ** 2).sum(-1)).mean()))
