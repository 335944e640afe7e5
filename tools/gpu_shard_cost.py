"""Strong-scaling ceiling from ONE GPU: the time of one rank's shard of a frame (shard_count = N, band rows dealt round-robin) against
1/N of the whole frame's time.  What a rank pays that does not shrink with its share -- scene-independent launches (19 per batch),
partially filled batches and lanes, the full-film clear and the film download -- shows as efficiency < 1 before any 8-GPU node does.
  python tools/gpu_shard_cost.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp


def timed(ctx, p, reps=3):
    if p.spp > 1024: reps = 1
    ctx.render(p)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render(p)
    return (time.perf_counter() - t0) / reps


def main():
    ctx = jp.Context(0)
    for name, W, Hh, spp in (("cornell", 512, 512, 1024), ("bunny", 1920, 1080, 512), ("bunny", 1920, 1080, 4096)):   # the last: BASELINE.json configs[4] at its own sample count
        hb = H.scenes.build_bunny(H.scenes.HostBackend("s"), W, Hh) if name == "bunny" else H.SCENES[name](H.scenes.HostBackend("s"), W, Hh)
        ctx.upload(hb.flatten())
        t1 = timed(ctx, jp.render_params(W, Hh, spp, band_rows=jp.distributed.balanced_band_rows(Hh, 1)))
        print("%s %dx%dx%d: whole frame %.1f ms (%.0f Msamples/s)" % (name, W, Hh, spp, t1 * 1e3, W * Hh * spp / t1 / 1e6), flush=True)
        for n in ((8,) if spp > 1024 else (2, 4, 8)):
            band = jp.distributed.balanced_band_rows(Hh, n)
            ts = [timed(ctx, jp.render_params(W, Hh, spp, band_rows=band, shard_index=r, shard_count=n)) for r in (0, n - 1)]
            tn = max(ts)
            print("   shard 1/%d (%d-row bands): %.1f ms  -> strong-scaling ceiling %.2f (x%.2f of %d); lanes %d" % (n, band, tn * 1e3, t1 / (n * tn), t1 / tn, n, ctx.build_info().lanes_last_render), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
