import os, sys, time
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import harness as H
jp = H.jp
ctx = jp.Context(0)
hb = H.SCENES["cornell"](H.scenes.HostBackend("s"), 512, 512)
ctx.upload(hb.flatten())
for n, spp in ((8, 8192), (4, 4096), (8, 1024), (4, 1024)):
    band = jp.distributed.balanced_band_rows(512, n)
    p = jp.render_params(512, 512, spp, band_rows=band, shard_index=0, shard_count=n)
    for lanes in ("1", "2", "3", "4"):
        ctx.set_options(lanes=int(lanes))
        ctx.render(p); t0 = time.perf_counter(); ctx.render(p); ctx.render(p); dt = (time.perf_counter() - t0) / 2
        print("shard 1/%d spp %d lanes %s: %.1f ms  %.0f Msamples/s per GPU" % (n, spp, lanes, dt * 1e3, 512 * (512 // n) * spp / dt / 1e6), flush=True)
