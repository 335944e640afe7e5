"""Wave-iteration statistics of the refill kernels on the 280k-triangle scene (diagnostic build: make -C jet-pbrt_amd/csrc walk_stats):
how many lanes step in a node / leaf turn, how many are live, how much of the walk happens after the region's pool has run dry.
  python tools/turn_stats.py [WxH:SPP] ["ENV=.. ENV=.."] ..."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["JETPBRT_AMD_LIB"] = os.path.join(REPO, "jet-pbrt_amd", "csrc", "libjetpbrt_amd_walk_stats.so")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
spec = sys.argv[1] if len(sys.argv) > 1 else "800x600:64"
W, Hh = (int(x) for x in spec.split(":")[0].split("x")); spp = int(spec.split(":")[1])
lib = C.CDLL(os.environ["JETPBRT_AMD_LIB"])
for v in (sys.argv[2:] or [""]):
    kv = dict(x.split("=") for x in v.split()) if v.strip() else {}
    for k, val in kv.items(): os.environ[k] = val
    hb = H.scenes.HostBackend("b"); H.scenes.build_bunny(hb, W, Hh)
    ctx = jp.Context(0)
    try:
        ctx.upload(hb.flatten())
        ws = (C.c_ulonglong * 8)(); ts = (C.c_ulonglong * 32)()
        ctx.render(jp.render_params(W, Hh, spp)); lib.jp_dbg_walk_stats(ws); lib.jp_dbg_turn_stats(ts)
        ctx.render(jp.render_params(W, Hh, spp)); c = ctx.counters()
        lib.jp_dbg_walk_stats(ws); lib.jp_dbg_turn_stats(ts)
        w = list(ws); t = list(ts)
        print("##", v or "(default)")
        for name, o, rays, wo in (("closest-hit", 0, c.closest_rays, 0), ("shadow", 16, c.shadow_rays, 4)):
            nt, nl, nv, lt, ll, lv, rf, rl, dt, dv, dl = (t[o + i] for i in range(11))
            turns = nt + lt + rf
            print("  %-11s rays %d | per ray: node steps %.2f leaf steps %.2f prim tests %.2f | wave iterations per 64 rays: node %.1f leaf %.1f refill %.1f"
                  % (name, rays, w[wo] / rays, w[wo + 1] / rays, w[wo + 2] / rays, nt * 64 / rays, lt * 64 / rays, rf * 64 / rays))
            print("              node turns: %.1f lanes step of %.1f live | leaf turns: %.1f step of %.1f live | refills: %.1f lanes each | after the pool ran dry: %.1f %% of the turns, %.1f live lanes, %.1f stepping"
                  % (nl / max(1, nt), nv / max(1, nt), ll / max(1, lt), lv / max(1, lt), rl / max(1, rf), 100.0 * dt / max(1, nt + lt), dv / max(1, dt), dl / max(1, dt)))
    finally:
        ctx.close()
        for k in kv: os.environ.pop(k, None)
