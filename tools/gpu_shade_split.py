"""k_shade cost split (debug variants built with -DJP_DBG_*): Cornell 512x512x64spp at max_depth 1 and 5."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_lambert"
W = Hh = 512
hb = H.SCENES[name](H.scenes.HostBackend("c2"), W, Hh)
ctx = jp.Context(0); ctx.upload(hb.flatten())
for depth in (1, 5):
    ctx.render(jp.render_params(W, Hh, 64, depth))
    ctx.set_profiling(True)
    ctx.render(jp.render_params(W, Hh, 64, depth)); c = ctx.counters()
    ctx.set_profiling(False)
    print(os.path.basename(os.environ.get("JETPBRT_AMD_LIB", "base")), name, "depth", depth, "ms %.2f extend %.2f shade %.2f shadow %.2f | closest %d shadow %d" % (c.render_ms, c.extend_ms, c.shade_ms, c.shadow_ms, c.closest_rays, c.shadow_rays), flush=True)
