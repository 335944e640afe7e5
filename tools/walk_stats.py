"""Per-ray work of the 4-wide walks on the 280k-triangle scene: default path (own PLOC tree, Walker<4>) against the certified walk over the reference tree's
leaves (Walker<6>).  Needs the diagnostic build: make -C jet-pbrt_amd/csrc walk_stats."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["JETPBRT_AMD_LIB"] = os.path.join(REPO, "jet-pbrt_amd", "csrc", "libjetpbrt_amd_walk_stats.so")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh, spp = 800, 600, 8
lib = C.CDLL(os.environ["JETPBRT_AMD_LIB"])
def run(tag, certified, env):
    for k, v in env.items(): os.environ[k] = v
    hb = H.scenes.HostBackend("b")
    if certified: hb.set_reference_tree(True, certified=True)
    H.scenes.build_bunny(hb, W, Hh)
    ctx = jp.Context(0)
    try:
        ctx.upload(hb.flatten())
        out = (C.c_ulonglong * 8)(); lib.jp_dbg_walk_stats(out)
        ctx.render(jp.render_params(W, Hh, spp)); c = ctx.counters()
        lib.jp_dbg_walk_stats(out)
        s = list(out)
        print("%-34s closest-hit rays: node steps %.2f  leaf steps %.2f  primitive tests %.2f  pushes %.2f | shadow rays: node steps %.2f  leaf steps %.2f  primitive tests %.2f  pushes %.2f"
              % (tag, s[0] / c.closest_rays, s[1] / c.closest_rays, s[2] / c.closest_rays, s[3] / c.closest_rays, s[4] / c.shadow_rays, s[5] / c.shadow_rays, s[6] / c.shadow_rays, s[7] / c.shadow_rays), flush=True)
    finally:
        ctx.close()
        for k in env: os.environ.pop(k, None)
run("default (PLOC tree, Walker<4>)", False, {})
run("certified (reference leaves)", True, {})
run("certified, cull slack 0", True, {"JETPBRT_CERT_SLACK": "0"})
run("certified, no edge-on flags", True, {"JETPBRT_CERT_EYE": "0"})
