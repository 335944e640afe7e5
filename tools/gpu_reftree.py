"""Cost of the reference-semantics traversal (the reference's own tree, unordered, its box test) vs the default path."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
ctx = jp.Context(0)
for name, W, Hh, spp, build in (("bunny", 800, 600, 32, lambda be, w, h: H.scenes.build_bunny(be, w, h)), ("cornell_lambert", 512, 512, 64, None)):
    for ref in (False, True):
        hb = H.scenes.HostBackend(name); hb.set_reference_tree(ref)
        t = time.time(); (build or H.SCENES[name])(hb, W, Hh); sp = hb.flatten(); tb = time.time() - t
        ctx.upload(sp)
        ctx.render(jp.render_params(W, Hh, 4))
        ctx.set_profiling(True); ctx.render(jp.render_params(W, Hh, spp)); c = ctx.counters(); ctx.set_profiling(False)
        print("%s reference_tree=%d: scene+tree %.2f s, mode %d, %d nodes height %d | %d spp %.1f ms %.1f Msamples/s extend %.1f shade %.1f shadow %.1f" % (
            name, ref, tb, ctx.build_info().traversal_mode, ctx.build_info().bvh_nodes, ctx.build_info().bvh_height, spp, c.render_ms, W * Hh * spp / c.render_ms / 1e3, c.extend_ms, c.shade_ms, c.shadow_ms), flush=True)
