"""Certified walk (Walker<6>) against the verbatim reference walk (Walker<5>) ray by ray on the 280k-triangle scene with the reference's tree:
camera rays over the whole image and secondary rays leaving the first hits.  Prints how many hit records differ and in which direction
(nearer: the certificate let through a hit the reference does not find; farther: the ordered walk missed a hit the reference finds; same
distance: a tie resolved differently), for several values of the distance-cull slack."""
import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H
jp = H.jp
W, Hh = 800, 600
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
rb = H.scenes.HostBackend("bunny"); rb.set_device_build(False); rb.set_reference_tree(True, certified=True); H.scenes.build_bunny(rb, W, Hh); rsp = rb.flatten()
rng = np.random.default_rng(7)
pxy = np.stack([rng.uniform(0, W, n), rng.uniform(0, Hh, n)], 1).astype(np.float32)
L = H.oracle_lib()
o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
H.libc_srand(1)
oh = L.jp_oracle_scene_new(rsp); L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d)); L.jp_oracle_scene_free(oh)
tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)

def trace(env, o, d, tmax):
    for k, v in env.items(): os.environ[k] = v
    ctx = jp.Context(0)
    try:
        ctx.upload(rsp); bi = ctx.build_info()
        hit, t, prim, nrm = ctx.trace(o, d, tmin[:len(o)], tmax)
    finally:
        ctx.close()
        for k in env: os.environ.pop(k, None)
    return hit, t, prim, nrm, bi

def compare(tag, ref, got):
    h5, t5, p5, _, _ = ref; h6, t6, p6, _, bi = got
    dt = t5.view(np.uint32) != t6.view(np.uint32); dp = (p5 != p6) & ~dt; dh = h5 != h6
    near = int(((t6 < t5) & (h6 != 0)).sum() + ((h6 != 0) & (h5 == 0)).sum()); far = int(((t6 > t5) & (h5 != 0) & (h6 != 0)).sum() + ((h5 != 0) & (h6 == 0)).sum())
    print("%-28s certified_walk %d | of %d rays: distance differs %d (certified nearer %d, farther / missed %d), same distance other primitive %d, hit flag differs %d"
          % (tag, bi.certified_walk, len(t5), int(dt.sum()), near, far, int(dp.sum()), int(dh.sum())), flush=True)
    return np.nonzero(dt | dp | dh)[0]

ref = trace({"JETPBRT_TRACE_VERBATIM": "1"}, o, d, tmax)
bad = None
for env in ({}, {"JETPBRT_CERT_EYE": "0"}, {"JETPBRT_CERT_EYE": "0", "JETPBRT_CERT_SLACK": "0"}, {"JETPBRT_CERT_EYE": "0.05"}):
    got = trace(env, o, d, tmax)
    idx = compare("camera " + (" ".join("%s=%s" % (k[8:], v) for k, v in env.items()) or "(default)"), ref, got)
    if bad is None:
        bad = idx
        for i in idx[:12]:
            print("   ray %d px (%.2f, %.2f): verbatim hit %d t %.9g prim %d | certified hit %d t %.9g prim %d" % (i, pxy[i, 0], pxy[i, 1], ref[0][i], ref[1][i], ref[2][i], got[0][i], got[1][i], got[2][i]))
# secondary rays: from the verbatim hit points, cosine-ish random directions about the normal
h5, t5, p5, n5, _ = ref
m = h5 != 0
P = (o + d * t5[:, None])[m]; N = n5[m]; din = d[m]
N = np.where((N * din).sum(1, keepdims=True) > 0, -N, N)
r = rng.normal(size=P.shape).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
d2 = (N + r).astype(np.float32); d2 /= np.maximum(np.linalg.norm(d2, axis=1, keepdims=True), 1e-20).astype(np.float32)
o2 = P.astype(np.float32); d2 = d2.astype(np.float32); k = len(o2)
tm2 = np.full(k, np.inf, np.float32)
ref2 = trace({"JETPBRT_TRACE_VERBATIM": "1"}, o2, d2, tm2)
for env in ({}, {"JETPBRT_CERT_SLACK": "0"}):
    got = trace(env, o2, d2, tm2)
    idx = compare("secondary " + (" ".join("%s=%s" % (k_[8:], v) for k_, v in env.items()) or "(default)"), ref2, got)
    for i in idx[:6]:
        print("   ray %d: verbatim hit %d t %.9g prim %d | certified hit %d t %.9g prim %d" % (i, ref2[0][i], ref2[1][i], ref2[2][i], got[0][i], got[1][i], got[2][i]))
