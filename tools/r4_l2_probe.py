import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness as H
jp = H.jp
W, Hh, spp = 200, 150, 4
hb = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh); sp = hb.flatten()
ctx = jp.Context(0); ctx.upload(sp); p = jp.render_params(W, Hh, spp)
film = ctx.render(p)
for seed in (1, 2, 3, 4, 5, 6, 7, 8):
    ref, cnt = H.oracle_render(sp, p, 16, rand_seed=seed)
    l2 = float(np.sqrt(((film - ref) ** 2).sum(-1)).mean())
    print("rand seed %d: mean L2 %.3e, identical px %.4f" % (seed, l2, float((film == ref).all(-1).mean())), flush=True)
