"""Traversal statistics on the 280k-triangle scene (host SAH tree): binary walk vs the 4-wide collapse used by Walker<4>.
Per ray: node steps, leaf steps, primitive plane tests / full tests, bytes fetched through the vector memory path."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H

def arr(p, n, shape=None):
    a = np.ctypeslib.as_array(p, (n,)).copy()
    return a if shape is None else a.reshape(shape)

W = Hh = 64
hb = H.scenes.HostBackend("st"); hb.set_device_build(False)      # the host SAH tree is what this tool walks
H.scenes.build_bunny(hb, W, Hh)
sp = hb.flatten(); s = sp.contents
left = arr(s.bvh_left, s.n_bvh_nodes); right = arr(s.bvh_right, s.n_bvh_nodes); bnd = arr(s.bvh_bounds, 6 * s.n_bvh_nodes, (-1, 6))
pidx = arr(s.bvh_prim_index, s.n_bvh_prim_indices)
p0 = arr(s.tri_p0, 3 * s.n_triangles, (-1, 3)); p1 = arr(s.tri_p1, 3 * s.n_triangles, (-1, 3)); p2 = arr(s.tri_p2, 3 * s.n_triangles, (-1, 3)); tn = arr(s.tri_n, 3 * s.n_triangles, (-1, 3))
stype = arr(s.prim_shape_type, s.n_primitives); sidx = arr(s.prim_shape_index, s.n_primitives)
leaves = np.array([right[i] for i in range(s.n_bvh_nodes) if left[i] < 0])
print("nodes", s.n_bvh_nodes, "leaves", len(leaves), "leaf sizes", np.bincount(leaves))
area = lambda b: (b[3]-b[0])*(b[4]-b[1]) + (b[4]-b[1])*(b[5]-b[2]) + (b[5]-b[2])*(b[3]-b[0])
# 4-wide collapse (jp_upload_scene): children of binary node b
kids4 = {}
def children4(b):
    if b in kids4: return kids4[b]
    ch = [left[b], right[b]]
    while len(ch) < 4:
        best = -1; ba = -1
        for k, c in enumerate(ch):
            if left[c] >= 0 and area(bnd[c]) > ba: ba = area(bnd[c]); best = k
        if best < 0: break
        c = ch[best]; ch[best] = left[c]; ch.append(right[c])
    kids4[b] = ch
    return ch

rng = np.random.default_rng(3)
L = H.oracle_lib(); oh = L.jp_oracle_scene_new(sp)
n = 1500
pxy = (rng.random((n, 2)) * [W, Hh]).astype(np.float32)
o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
L.jp_oracle_trace(oh, n, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
dd = rng.normal(size=(n, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
dd = np.where(((dd * nrm).sum(1) < 0)[:, None], -dd, dd).astype(np.float32)
cam = [(o[i], d[i]) for i in range(n)]
sec = [(pos[i], dd[i]) for i in range(n) if hit[i] > 0]

def box(b, o, idr, tmin, tmax):
    t0 = (b[:3] - o) * idr; t1 = (b[3:] - o) * idr
    with np.errstate(invalid="ignore"):
        tn_ = max(np.fmax(np.fmin(t0, t1), -np.inf).max(), tmin); tf = min(np.fmin(np.fmax(t0, t1), np.inf).min(), tmax)
    return tn_ <= tf * 1.000002, tn_
def leaf(cur, o, d, tmax, st):
    st["leaf"] += 1
    first = -left[cur] - 1
    for k in range(right[cur]):
        p = pidx[first + k]
        st["plane"] += 1
        if stype[p] != 0: continue
        i = sidx[p]
        oa = p0[i] - o; den = float(np.dot(tn[i], d))
        if den == 0: continue
        tt = float(np.dot(tn[i], oa)) / den
        if not (tt > 0.001 and tt < tmax): continue
        st["full"] += 1
        ob = p1[i] - o; oc = p2[i] - o
        a = np.dot(np.cross(oc, ob), d); b = np.dot(np.cross(ob, oa), d); c = np.dot(np.cross(oa, oc), d)
        if (a < 0 and b < 0 and c < 0) or (a >= 0 and b >= 0 and c >= 0): tmax = tt
    return tmax
def trace(o, d, wide, st):
    with np.errstate(divide="ignore"):
        idr = (1.0 / d).astype(np.float32)
    tmax = np.inf; stack = []; cur = 0; st["rays"] += 1
    while True:
        if left[cur] >= 0:
            st["node"] += 1
            ch = children4(cur) if wide else [left[cur], right[cur]]
            hits = []
            for c in ch:
                h, tn_ = box(bnd[c], o, idr, 0.001, tmax)
                if h: hits.append((tn_, c))
            hits.sort(key=lambda x: x[0])
            if hits:
                cur = hits[0][1]
                for x in reversed(hits[1:]): stack.append(x[1])
                continue
        else:
            tmax = leaf(cur, o, d, tmax, st)
        if not stack: break
        cur = stack.pop()
for label, rays in (("camera", cam), ("secondary", sec)):
    for wide in (False, True):
        st = dict(rays=0, node=0, leaf=0, plane=0, full=0)
        for (o_, d_) in rays[:800]:
            trace(o_, d_, wide, st)
        r = st["rays"]
        nb = st["node"] / r * (56 if wide else 64); pb = st["plane"] / r * 64
        print("  %-9s %-7s per ray: node steps %.2f  leaf steps %.2f  plane tests %.2f  full tests %.2f | bytes: nodes %.0f prims %.0f (lazy: %.0f)" % (
            label, "4-wide" if wide else "binary", st["node"] / r, st["leaf"] / r, st["plane"] / r, st["full"] / r, nb, pb, st["plane"] / r * 32 + st["full"] / r * 32))
