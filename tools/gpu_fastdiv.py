"""What bit-exact division / square root cost: the kernel library built WITHOUT correctly rounded fp32 divide / sqrt
(-fno-hip-fp32-correctly-rounded-divide-sqrt: v_rcp / v_rsq based, ~2 ulp) against the product build.  Diagnostic only -- the
product is the correctly rounded build; this quantifies DESIGN.md's "40 % of k_shade's vector instructions are IEEE expansions"."""
import os, subprocess, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import harness as H
jp = H.jp
name, W, Hh, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hb = H.scenes.build_bunny(H.scenes.HostBackend("f"), W, Hh) if name == "bunny" else H.SCENES[name](H.scenes.HostBackend("f"), W, Hh)
ctx = jp.Context(0); ctx.upload(hb.flatten()); p = jp.render_params(W, Hh, spp)
ctx.render(p); t0 = time.perf_counter(); f = ctx.render(p); f = ctx.render(p); dt = (time.perf_counter() - t0) / 2
ctx.set_options(lanes=1); ctx.set_profiling(True); ctx.render(p); c = ctx.counters()
np.save(sys.argv[5], f)
print("%%s %%dx%%dx%%d: %%.1f Msamples/s | 1 lane: extend %%.2f shade %%.2f shadow %%.2f ms" %% (name, W, Hh, spp, W * Hh * spp / dt / 1e6, c.extend_ms, c.shade_ms, c.shadow_ms), flush=True)
''' % (REPO, REPO)
import numpy as np
for name, W, Hh, spp in (("cornell", 512, 512, 1024), ("bunny", 800, 600, 512)):
    films = {}
    for tag, lib in (("exact", "libjetpbrt_amd.so"), ("fastdiv", "libjetpbrt_amd_fastdiv.so")):
        out = os.path.join(REPO, "gpurun_out", "fd_%s_%s.npy" % (name, tag))
        env = dict(os.environ, JETPBRT_AMD_LIB=os.path.join(REPO, "jet-pbrt_amd", "csrc", lib))
        r = subprocess.run([sys.executable, "-c", code, name, str(W), str(Hh), str(spp), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(tag, r.stdout.strip() or r.stderr[-400:], flush=True)
        films[tag] = np.load(out)
    d = np.sqrt(((films["exact"] - films["fastdiv"]) ** 2).sum(-1))
    print("   film exact vs fastdiv: mean per-pixel L2 %.3e, max %.3e, identical px %.4f" % (d.mean(), d.max(), (films["exact"] == films["fastdiv"]).all(-1).mean()), flush=True)
