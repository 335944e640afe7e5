"""A/B of kernel variants in ONE process on the GPU box: every variant is a set of JETPBRT_* environment switches (read by
jp_upload_scene / jp_render), the scene is re-uploaded per variant, films are compared bit for bit against the first variant.
  python tools/gpu_ab.py SCENE[:WxH[:SPP]] "A=1 B=2" "A=0" ...      (SCENE: cornell | cornell_lambert | bunny | misc ...)
Prints per variant: Msamples/s with the default stream lanes, and the per-class kernel times of one single-lane frame."""
import hashlib, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp


def main():
    spec = sys.argv[1].split(":")
    name = spec[0]
    W, Hh = (int(x) for x in spec[1].split("x")) if len(spec) > 1 else (512, 512)
    spp = int(spec[2]) if len(spec) > 2 else 256
    variants = sys.argv[2:] or [""]
    if name == "bunny":
        hb = H.scenes.build_bunny(H.scenes.HostBackend("ab"), W, Hh)
    else:
        hb = H.SCENES[name](H.scenes.HostBackend("ab"), W, Hh)
    sp = hb.flatten()
    ref = None
    for v in variants:
        kv = dict(x.split("=") for x in v.split()) if v.strip() else {}
        for k, val in kv.items():
            os.environ[k] = val
        if any(k.startswith("JETPBRT_BVH_") or k.startswith("JETPBRT_SAH_") for k in kv):     # host tree parameters: build the scene again
            hb = H.scenes.build_bunny(H.scenes.HostBackend("ab"), W, Hh) if name == "bunny" else H.SCENES[name](H.scenes.HostBackend("ab"), W, Hh)
            sp = hb.flatten()
        try:
            ctx = jp.Context(0); ctx.upload(sp)
            p = jp.render_params(W, Hh, spp)
            ctx.render(p)                                       # warm: allocations, lanes, clocks
            t0 = time.perf_counter(); film = ctx.render(p); film = ctx.render(p); dt = (time.perf_counter() - t0) / 2
            lanes = ctx.build_info().lanes_last_render
            ctx.set_options(lanes=int(kv.get("JETPBRT_LANES", "1")))       # (the environment is read when the context is created; a live context takes JpOptions)
            ctx.set_profiling(True); ctx.render(p); c = ctx.counters(); ctx.set_profiling(False)
            ctx.set_options()
            same = "-" if ref is None else ("bit-identical" if np.array_equal(ref.view(np.uint32), film.view(np.uint32)) else "DIFFERENT mean L2 %.3e, identical px %.5f" % (
                float(np.sqrt(((film - ref) ** 2).sum(-1)).mean()), float((film == ref).all(-1).mean())))
            if ref is None:
                ref = film
            bi = ctx.build_info()
            sched = ("fused R=%d wgs=%d" % (bi.fused_region, bi.fused_workgroups)) if bi.fused_last_render else ("%d lanes" % lanes)
            print("%-44s %s %dx%dx%d: %7.1f Msamples/s (%s) | 1 lane: %7.1f ms  path %.2f extend %.2f shade %.2f shadow %.2f other %.2f | rays %d/%d | film %s" % (
                v or "(default)", name, W, Hh, spp, W * Hh * spp / dt / 1e6, sched, c.render_ms, c.path_ms, c.extend_ms, c.shade_ms, c.shadow_ms, c.other_ms,
                c.closest_rays, c.shadow_rays, same + " sha1 " + hashlib.sha1(film.tobytes()).hexdigest()[:12])
                + ((" | certified walk: %d nodes, %d edge-on leaves, %d rays walked again" % (bi.certified_nodes, bi.certified_eye_leaves, c.certified_fallback_rays)) if bi.certified_walk else ""), flush=True)
            ctx.close()
        finally:
            for k in kv:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
