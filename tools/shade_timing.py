"""Where a k_shade wave spends its cycles.  Needs a diagnostic build of the kernel library:
     make -C jet-pbrt_amd/csrc EXTRA=-DJP_SHADE_TIMING && JETPBRT_AMD_LIB=... python tools/shade_timing.py SCENE[:WxH[:SPP]]
   Every wave adds shader-clock cycles per section (s_memtime deltas) to a device array; printed as shares of the wave's lifetime."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
NAMES = ["table staging", "partition", "loop overhead", "wait: prefetched records + earlier stores", "decode, emission, closure, frame",
         "next-event estimation", "BSDF sample + roulette", "output room + store issue", "closing barrier", "memory wait at the end of the chunk"]
TRAV = ["primitive records to LDS, region fill", "loop overhead", "wait: prefetched entry + earlier stores", "load issue (contribution, next ray)",
        "box phase", "primitive phase", "contribution (waits for its load)", "radiance store", "-", "-"]
if os.environ.get("JP_TIMING_KERNEL") == "shadow":                # a -DJP_TRAV_TIMING build: k_shadow<2> instead of k_shade
    NAMES = TRAV
spec = sys.argv[1].split(":")
W, Hh = (int(x) for x in spec[1].split("x")) if len(spec) > 1 else (512, 512)
spp = int(spec[2]) if len(spec) > 2 else 256
hb = H.scenes.build_bunny(H.scenes.HostBackend("t"), W, Hh) if spec[0] == "bunny" else H.SCENES[spec[0]](H.scenes.HostBackend("t"), W, Hh)
lib = ctypes.CDLL(jp.HIP_LIB_PATH)
lib.jp_dbg_shade_timing.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
for lanes in ("1", "3"):
    os.environ["JETPBRT_LANES"] = lanes
    ctx = jp.Context(0); ctx.upload(hb.flatten()); p = jp.render_params(W, Hh, spp)
    ctx.render(p)
    buf = (ctypes.c_ulonglong * 16)(); assert lib.jp_dbg_shade_timing(buf) == 0
    ctx.render(p)
    assert lib.jp_dbg_shade_timing(buf) == 0
    t = np.array(buf[:10], dtype=np.float64); waves = buf[15]
    print("%s %dx%dx%d, %s lane(s): %d timed waves, %.0f cycles per wave" % (spec[0], W, Hh, spp, lanes, waves, t.sum() / max(1, waves)))
    for n_, v in zip(NAMES, t):
        print("   %-44s %5.1f %%" % (n_, 100 * v / t.sum()))
    ctx.close()
