"""Quick throughput probe on the GPU box: Cornell Lambert 512x512, per-class kernel times."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_lambert"
W = Hh = 512
hb = H.SCENES[name](H.scenes.HostBackend("c2"), W, Hh)
ctx = jp.Context(0); ctx.upload(hb.flatten())
ctx.render(jp.render_params(W, Hh, 64))
for spp, prof in [(256, False), (64, True)]:
    ctx.set_profiling(prof)
    ctx.render(jp.render_params(W, Hh, spp))
    c = ctx.counters()
    print(os.environ.get("JETPBRT_BLOCKS_PER_CU", "-"), name, "spp", spp, "ms %.2f" % c.render_ms, "Msamples/s %.1f" % (W * Hh * spp / c.render_ms / 1e3),
          "extend %.2f shade %.2f shadow %.2f other %.2f" % (c.extend_ms, c.shade_ms, c.shadow_ms, c.other_ms), flush=True)
