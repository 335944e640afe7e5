"""The 4-wide quantised tree (Walker<4>) against the binary tree on the 280k-triangle scene: closest-hit records of random rays
(camera rays + rays leaving surface points in random directions) through jp_trace with both trees, then the frame A/B."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
hb = H.scenes.build_bunny(H.scenes.HostBackend("q4"), W, Hh)
sp = hb.flatten()
ctx = jp.Context(0); ctx.upload(sp)
rng = np.random.default_rng(5)
n = 400000
cam = np.array(sp.contents.camera.pos[:], np.float32)
front = np.array(sp.contents.camera.front[:], np.float32); right = np.array(sp.contents.camera.right[:], np.float32); up = np.array(sp.contents.camera.up[:], np.float32)
u = rng.random((n, 2), dtype=np.float32)
d = front + right * (u[:, :1] - 0.5) + up * (0.5 - u[:, 1:])
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.tile(cam, (n, 1)).astype(np.float32)
tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
def both(o, d, label):
    ctx.set_options()
    h4, t4, p4, n4 = ctx.trace(o, d, tmin[:len(o)], tmax[:len(o)])
    ctx.set_options(trace_walk=1)
    h2, t2, p2, n2 = ctx.trace(o, d, tmin[:len(o)], tmax[:len(o)])
    ctx.set_options()
    same = (h4 == h2) & (p4 == p2) & (t4.view(np.uint32) == t2.view(np.uint32))
    print("%s: %d rays, %d hits; identical records %d, different %d (4-wide nearer: %d, binary nearer: %d)" % (
        label, len(o), int(h4.sum()), int(same.sum()), int((~same).sum()), int(((~same) & (t4 < t2)).sum()), int(((~same) & (t2 < t4)).sum())), flush=True)
    return h4, t4, n4
h, t, nr = both(o, d.astype(np.float32), "camera rays")
# secondary rays: from the hit points, random directions in the normal's hemisphere
m = h == 1
p = (o[m] + d[m] * t[m, None]).astype(np.float32)
v = rng.normal(size=(m.sum(), 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
v *= np.sign((v * nr[m]).sum(1, keepdims=True)); v = v.astype(np.float32)
both(p, v, "secondary rays")
ctx.close()
