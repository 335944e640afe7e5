"""Developer check run on the GPU box: GPU films vs the oracle restatement on the small scenes, trace parity,
and a first throughput number.  Writes a log under gpurun_out/."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp

def l2(a, b):
    return float(np.sqrt(((a - b) ** 2).sum(-1)).mean())

def main():
    ctx = jp.Context(0)
    rng = np.random.default_rng(1)
    for name in ["cornell_lambert", "cornell", "bunny_small", "misc", "lights"]:
        W, Hh, spp = 96, 96, 16
        hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
        sp = hb.flatten()
        ctx.upload(sp)
        # trace parity: camera rays + random rays
        o = np.tile(np.array([[278, 273, 400]], np.float32), (20000, 1)) if "cornell" in name or name == "misc" else np.tile(np.array([[-300, 300, -300]], np.float32), (20000, 1))
        d = rng.normal(size=(20000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        tmin = np.full(20000, 0.001, np.float32); tmax = np.full(20000, np.inf, np.float32)
        hit, t, prim, nrm = ctx.trace(o, d, tmin, tmax)
        oh = H.oracle_lib().jp_oracle_scene_new(sp)
        ohit = np.zeros(20000, np.int32); ot = np.zeros(20000, np.float32); oprim = np.zeros(20000, np.int32); onrm = np.zeros((20000, 3), np.float32); opos = np.zeros((20000, 3), np.float32)
        H.oracle_lib().jp_oracle_trace(oh, 20000, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(ohit), H.ptr(ot), H.ptr(oprim), H.ptr(onrm), H.ptr(opos))
        print(name, "trace: hit mismatch", int((hit != ohit).sum()), "t mismatch", int((t != ot).sum()), "prim mismatch", int((prim != oprim).sum()), "nrm mismatch", int((nrm != onrm).any(1).sum()), flush=True)
        params = jp.render_params(W, Hh, spp, 5, 1234)
        t0 = time.time(); film = ctx.render(params); dt = time.time() - t0
        fo, cnt = H.oracle_render(sp, params, 8)
        c = ctx.counters()
        print(name, "render %.3fs" % dt, "L2 mean", l2(film, fo), "max", float(np.abs(film - fo).max()), "exact px", float((film == fo).all(-1).mean()),
              "rays", c.closest_rays, cnt.closest_rays, "shadow", c.shadow_rays, cnt.shadow_rays, "hits", c.closest_hits, cnt.closest_hits, "occ", c.shadow_occluded, cnt.shadow_occluded, flush=True)
    # throughput: Cornell Lambert 512x512
    W = Hh = 512
    hb = H.SCENES["cornell_lambert"](H.scenes.HostBackend("c2"), W, Hh)
    ctx.upload(hb.flatten())
    for spp, prof in [(64, False), (64, False), (256, False), (64, True)]:
        ctx.set_profiling(prof)
        params = jp.render_params(W, Hh, spp, 5, 1234)
        t0 = time.time(); film = ctx.render(params); dt = time.time() - t0
        c = ctx.counters()
        print("C2 spp", spp, "wall %.3fs" % dt, "device ms %.2f" % c.render_ms, "Msamples/s %.1f" % (W * Hh * spp / c.render_ms / 1e3),
              "seg/sample %.3f shadow/sample %.3f" % (c.closest_rays / c.samples, c.shadow_rays / c.samples),
              "extend %.2f shade %.2f shadow %.2f other %.2f" % (c.extend_ms, c.shade_ms, c.shadow_ms, c.other_ms), c.extend_launches, flush=True)

if __name__ == "__main__":
    main()
