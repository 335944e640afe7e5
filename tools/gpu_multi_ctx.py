"""How many concurrent contexts pay off?  K contexts of one process, one host thread, async enqueue, 1/K of the bands each."""
import os, sys, time
import torch
torch.zeros(1, device="cuda")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W = Hh = 512; spp = 1024
hb = H.SCENES[sys.argv[1] if len(sys.argv) > 1 else "cornell_lambert"](H.scenes.HostBackend("c2"), W, Hh); sp = hb.flatten()
ctxs = [jp.Context(0) for _ in range(4)]
for c in ctxs: c.upload(sp)
films = [torch.zeros((Hh, W, 3), dtype=torch.float32, device="cuda") for _ in range(4)]
for K in (1, 2, 3, 4, 2, 1):
    ps = [jp.render_params(W, Hh, spp, band_rows=8, shard_index=i, shard_count=K) for i in range(K)]
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.time()
        for i in range(K): ctxs[i].render_device(ps[i], films[i].data_ptr(), sync=False)
        for i in range(K): ctxs[i].synchronize()
        dt = time.time() - t
    print("K=%d contexts (each itself %s-lane): %.1f ms  %.1f Msamples/s" % (K, os.environ.get("JETPBRT_LANES", "2"), dt * 1e3, W * Hh * spp / dt / 1e6), flush=True)
