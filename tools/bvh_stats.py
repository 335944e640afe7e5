"""Offline BVH traversal statistics (host BVH, device traversal logic restated in Python) for builder tuning."""
import os, sys, math
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H

def arr(p, n, shape=None):
    a = np.ctypeslib.as_array(p, (n,)).copy()
    return a if shape is None else a.reshape(shape)

def main(name="cornell_lambert", nrays=3000):
    W = Hh = 64
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
    sp = hb.flatten(); s = sp.contents
    left = arr(s.bvh_left, s.n_bvh_nodes); right = arr(s.bvh_right, s.n_bvh_nodes); bnd = arr(s.bvh_bounds, 6 * s.n_bvh_nodes, (-1, 6))
    pidx = arr(s.bvh_prim_index, s.n_bvh_prim_indices)
    p0 = arr(s.tri_p0, 3 * s.n_triangles, (-1, 3)); p1 = arr(s.tri_p1, 3 * s.n_triangles, (-1, 3)); p2 = arr(s.tri_p2, 3 * s.n_triangles, (-1, 3)); tn = arr(s.tri_n, 3 * s.n_triangles, (-1, 3))
    sidx = arr(s.prim_shape_index, s.n_primitives)
    leaves = [right[i] for i in range(s.n_bvh_nodes) if left[i] < 0]
    print(name, "nodes", s.n_bvh_nodes, "leaves", len(leaves), "leaf sizes", np.bincount(leaves))
    # ray set from the oracle: camera hits -> cosine-ish bounce rays and shadow rays to light points
    rng = np.random.default_rng(3)
    L = H.oracle_lib(); oh = L.jp_oracle_scene_new(sp)
    n = nrays
    pxy = (rng.random((n, 2)) * [W, Hh]).astype(np.float32)
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
    def otrace(o, d):
        tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
        L.jp_oracle_trace(oh, n, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
        return hit, pos, nrm
    hit, pos, nrm = otrace(o, d)
    # second-generation origins: bounce once more for incoherent rays
    dd = rng.normal(size=(n, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    dd = np.where(((dd * nrm).sum(1) * (-(d * nrm).sum(1)) < 0)[:, None], -dd, dd).astype(np.float32)
    hit2, pos2, nrm2 = otrace(pos, dd)
    ok = (hit > 0) & (hit2 > 0)
    org = pos2[ok]; nn = nrm2[ok]
    m = org.shape[0]
    d3 = rng.normal(size=(m, 3)).astype(np.float32); d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    ext = [(org[i], d3[i], np.inf) for i in range(m)]
    lp = np.stack([rng.uniform(213, 343, m), np.full(m, 548.7), -rng.uniform(227, 332, m)], 1).astype(np.float32)
    sd = lp - org; dist = np.linalg.norm(sd, axis=1); sd = (sd / dist[:, None]).astype(np.float32)
    shd = [(org[i], sd[i], dist[i] - 0.001) for i in range(m)]

    def box(b, o, idr, tmin, tmax):
        t0 = (b[:3] - o) * idr; t1 = (b[3:] - o) * idr
        with np.errstate(invalid="ignore"):
            tn_ = max(np.fmax(np.fmin(t0, t1), -np.inf).max(), tmin); tf = min(np.fmin(np.fmax(t0, t1), np.inf).min(), tmax)
        return tn_ <= tf * 1.000002, tn_
    def tri(i, o, d, tmin, tmax, st):
        st["tri"] += 1
        oa = p0[i] - o; den = float(np.dot(tn[i], d))
        if den == 0: return None
        t = float(np.dot(tn[i], oa)) / den
        if not (t > tmin and t < tmax): return None
        st["full"] += 1
        ob = p1[i] - o; oc = p2[i] - o
        a = np.dot(np.cross(oc, ob), d); b = np.dot(np.cross(ob, oa), d); c = np.dot(np.cross(oa, oc), d)
        if (a < 0 and b < 0 and c < 0) or (a >= 0 and b >= 0 and c >= 0): return t
        return None
    def trace(o, d, tmax, anyhit, nearskip, st):
        with np.errstate(divide="ignore"):
            idr = (1.0 / d).astype(np.float32)
        stack = []; cur = 0; st["rays"] += 1
        while True:
            alive = True
            while left[cur] >= 0:
                st["node"] += 1
                l, r = left[cur], right[cur]
                hl, ln = box(bnd[l], o, idr, 0.001, tmax); hr, rn = box(bnd[r], o, idr, 0.001, tmax)
                if hl and hr:
                    lf = True if anyhit else ln <= rn
                    cur = l if lf else r; stack.append((r if lf else l, rn if lf else ln))
                elif hl: cur = l
                elif hr: cur = r
                else:
                    got = False
                    while stack:
                        c, tnn = stack.pop()
                        if nearskip and tnn * 0.999998 > tmax: st["skip"] += 1; continue
                        cur = c; got = True; break
                    if not got: alive = False; break
            if not alive: break
            st["leaf"] += 1
            first = -left[cur] - 1
            for k in range(right[cur]):
                t = tri(sidx[pidx[first + k]], o, d, 0.001, tmax, st)
                if t is not None:
                    tmax = t
                    if anyhit: st["occ"] += 1; return
            got = False
            while stack:
                c, tnn = stack.pop()
                if nearskip and tnn * 0.999998 > tmax: st["skip"] += 1; continue
                cur = c; got = True; break
            if not got: break
    for label, rays, anyhit in (("extension", ext, False), ("shadow", shd, True)):
        st = dict(rays=0, node=0, leaf=0, tri=0, full=0, skip=0, occ=0)
        for (o_, d_, tm) in rays[:1500]:
            trace(o_, d_, tm, anyhit, not anyhit, st)
        r = st["rays"]
        print("  %-9s per ray: interior nodes %.2f  leaves %.2f  plane tests %.2f  full tests %.2f  skipped pops %.2f  occluded %.2f" % (label, st["node"] / r, st["leaf"] / r, st["tri"] / r, st["full"] / r, st["skip"] / r, st["occ"] / r))

if __name__ == "__main__":
    main(*(sys.argv[1:2]))
