"""configs[4] (1920x1080, 4096 spp), shard 0 of 8: certified walk vs verbatim walk, pixels that differ -- and whether a wider edge-on cover (JETPBRT_CERT_EYE) or a
larger cull slack (JETPBRT_CERT_SLACK) removes them, i.e. whether they are camera rays or rays the proof does not cover."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh, spp, world = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
band = jp.distributed.balanced_band_rows(Hh, world)
p0 = jp.render_params(W, Hh, spp, band_rows=band, shard_index=0, shard_count=world)
def film(certified, env):
    for k, v in env.items(): os.environ[k] = v
    hb = H.scenes.HostBackend("b"); hb.set_reference_tree(True, certified=certified); H.scenes.build_bunny(hb, W, Hh)
    ctx = jp.Context(0)
    try:
        ctx.upload(hb.flatten()); f = ctx.render(p0); c = ctx.counters(); bi = ctx.build_info()
    finally:
        ctx.close()
        for k in env: os.environ.pop(k, None)
    return f, c, bi
ref, rc, _ = film(False, {})
variants = [dict(x.split("=") for x in v.split()) if v.strip() else {} for v in sys.argv[2:]] or [{}, {"JETPBRT_CERT_EYE": "0.05"}, {"JETPBRT_CERT_SLACK": "1024"}, {"JETPBRT_CERT_SLACK": "1024", "JETPBRT_CERT_EYE": "0.05"}]
for env in variants:
    f, c, bi = film(True, env)
    bad = np.argwhere(~(f == ref).all(-1))
    print("%-50s differing pixels %d %s | rays %d/%d vs %d/%d | edge-on leaves %d, walked again %d" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", len(bad),
          [(int(y), int(x)) for y, x in bad[:6]], c.closest_rays, c.shadow_rays, rc.closest_rays, rc.shadow_rays, bi.certified_eye_leaves, c.certified_fallback_rays), flush=True)
