"""Where a k_path wave spends its cycles, phase by phase.  Needs the diagnostic build of the kernel library:
     make -C jet-pbrt_amd/csrc path_timing && JETPBRT_AMD_LIB=jet-pbrt_amd/csrc/libjetpbrt_amd_path_timing.so python tools/path_timing.py SCENE[:WxH[:SPP]] ["ENV=.."]
   Every wave adds shader-clock cycles per phase (s_memtime deltas) to a device array; printed as shares of the wave's lifetime."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
NAMES = ["job take + ray generation", "extend", "barrier after extend", "partition", "shade", "barrier after shade", "shadow (+ barrier at the loop top)", "radiance write-out"]
spec = sys.argv[1].split(":")
W, Hh = (int(x) for x in spec[1].split("x")) if len(spec) > 1 else (512, 512)
spp = int(spec[2]) if len(spec) > 2 else 256
for v in sys.argv[2:] or [""]:
    kv = dict(x.split("=") for x in v.split()) if v.strip() else {}
    os.environ.update(kv)
    hb = H.scenes.build_bunny(H.scenes.HostBackend("t"), W, Hh) if spec[0] == "bunny" else H.SCENES[spec[0]](H.scenes.HostBackend("t"), W, Hh)
    lib = ctypes.CDLL(jp.HIP_LIB_PATH)
    lib.jp_dbg_path_timing.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    ctx = jp.Context(0); ctx.upload(hb.flatten()); p = jp.render_params(W, Hh, spp)
    ctx.render(p)
    buf = (ctypes.c_ulonglong * 16)(); assert lib.jp_dbg_path_timing(buf) == 0
    ctx.render(p); c = ctx.counters()
    assert lib.jp_dbg_path_timing(buf) == 0
    t = np.array(buf[:8], dtype=np.float64); waves = buf[15]
    print("%s %dx%dx%d %s: %d waves, %.3g cycles per wave, frame %.1f ms" % (spec[0], W, Hh, spp, v, waves, t.sum() / max(1, waves), c.render_ms), flush=True)
    for n_, x in zip(NAMES, t):
        print("   %-40s %5.1f %%" % (n_, 100 * x / t.sum()))
    ctx.close()
    for k in kv: os.environ.pop(k, None)
