"""Concurrent contexts on the 280k-triangle scene: K contexts, 1/K of the bands each, one host thread."""
import os, sys, time
import torch
torch.zeros(1, device="cuda")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh, spp = 800, 600, 512
hb = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh); sp = hb.flatten()
ctxs = [jp.Context(0) for _ in range(4)]
for c in ctxs: c.upload(sp)
films = [torch.zeros((Hh, W, 3), dtype=torch.float32, device="cuda") for _ in range(4)]
for K in (1, 2, 3, 4):
    ps = [jp.render_params(W, Hh, spp, band_rows=4, shard_index=i, shard_count=K) for i in range(K)]
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        for i in range(K): ctxs[i].render_device(ps[i], films[i].data_ptr(), sync=False)
        for i in range(K): ctxs[i].synchronize()
        dt = time.time() - t
    print("bunny K=%d contexts bpc=%s: %.1f ms  %.1f Msamples/s" % (K, os.environ.get("JETPBRT_BLOCKS_PER_CU", "16"), dt * 1e3, W * Hh * spp / dt / 1e6), flush=True)
