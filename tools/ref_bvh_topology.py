"""On finely tessellated meshes the reference's closest hit depends on its BVH topology (bvh.h:54-146: random split axis from
libc rand(), median split): far from the camera the fp32 edge functions of FTriangle::Intersect accept points ~0.01 outside a
triangle -- outside its bounding box -- so such hits are found or not depending on the leaves a ray visits, and
FBounds3::Intersect (geometry.cc:10-30) additionally drops a subtree when rounding gives `tmax <= tmin`.  On the 280k-triangle
scene two reference-style trees built with different rand() seeds return DIFFERENT closest hits for ~3.5e-4 of the camera rays
through the meshes.  CPU only (uses the oracle's restatement of that tree, which is pinned bit-exact to the compiled reference)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
W, Hh = 800, 600
hb = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh); sp = hb.flatten()
L = H.oracle_lib()
rng = np.random.default_rng(1)
n = 400000
pxy = np.stack([rng.uniform(300, 520, n), rng.uniform(330, 370, n)], 1).astype(np.float32)
res = []
for seed in (1, 2, 3):
    H.libc_srand(seed)
    oh = L.jp_oracle_scene_new(sp)
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
    tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
    L.jp_oracle_trace(oh, n, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
    L.jp_oracle_scene_free(oh)
    res.append((hit.copy(), t.copy(), prim.copy()))
for a in range(3):
    for b in range(a + 1, 3):
        print("rand() seeds %d vs %d: %d of %d camera rays get a different closest hit" % (a + 1, b + 1, int((res[a][1].view(np.uint32) != res[b][1].view(np.uint32)).sum()), n))
k = np.nonzero(res[0][1].view(np.uint32) != res[1][1].view(np.uint32))[0][:4]
for i in k:
    print("  ray %d: t = %.4f (primitive %d) with seed 1, t = %.4f (primitive %d) with seed 2" % (i, res[0][1][i], res[0][2][i], res[1][1][i], res[1][2][i]))
