"""How much of the default path's difference from the reference on the tessellated meshes is the reference's own tree dependence?
400,000 camera rays through the band of the four meshes (the rays of tools/ref_bvh_topology.py), closest hit by
  R1, R2, R3  three reference-style trees (the oracle's restatement of bvh.h:54-146) grown from libc rand() seeds 1, 2, 3 -- seed 1 is the reference process's own
  D           the device default path: PLOC tree built on the device, 4-wide quantised walk
  H           the device on the host's binned-SAH tree, 4-wide quantised walk
  S           the device on the host tree with every box grown by 0.05 scene units (JETPBRT_BOX_PAD): visits every leaf whose triangles could accept a
              fringe hit = what testing every primitive would return ("superset-complete")
and the number of rays on which two of them return a different hit distance."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
L = H.oracle_lib()
rng = np.random.default_rng(1)
n = 400000
pxy = np.stack([rng.uniform(300, 520, n), rng.uniform(330, 370, n)], 1).astype(np.float32)
hbh = H.scenes.HostBackend("bunny"); hbh.set_device_build(False); H.scenes.build_bunny(hbh, W, Hh); sph = hbh.flatten()
res = {}
o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
for seed in (1, 2, 3):
    H.libc_srand(seed)
    oh = L.jp_oracle_scene_new(sph)
    L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
    L.jp_oracle_trace(oh, n, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
    L.jp_oracle_scene_free(oh)
    res["R%d" % seed] = t.copy()
def device(tag, sp, env):
    for k, v in env.items(): os.environ[k] = v
    ctx = jp.Context(0)
    try:
        ctx.upload(sp)
        bi = ctx.build_info()
        hit, t, prim, nrm = ctx.trace(o, d, tmin, tmax)
        print("%s: built_on_device %d, 4-wide nodes %d" % (tag, bi.built_on_device, bi.q4_nodes), flush=True)
    finally:
        ctx.close()
        for k in env: os.environ.pop(k, None)
    res[tag] = t.copy()
hbd = H.scenes.build_bunny(H.scenes.HostBackend("bunny"), W, Hh); spd = hbd.flatten()
device("D", spd, {})
device("H", sph, {})
device("S", sph, {"JETPBRT_BOX_PAD": "0.05"})
device("S2", sph, {"JETPBRT_BOX_PAD": "0.2"})
names = ["R1", "R2", "R3", "D", "H", "S", "S2"]
print("rays with a different closest-hit distance, of %d:" % n)
print("      " + "".join("%7s" % b for b in names))
for a in names:
    print("%-6s" % a + "".join("%7s" % ("-" if a == b else int((res[a].view(np.uint32) != res[b].view(np.uint32)).sum())) for b in names))
for a in ("D", "H", "S"):
    nearer = int((res[a] < res["R1"]).sum()); farther = int((res[a] > res["R1"]).sum())
    print("%s vs R1: %d rays nearer than the reference's hit (a fringe hit the reference's walk did not visit), %d farther (one it did and %s did not)" % (a, nearer, farther, a))
