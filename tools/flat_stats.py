"""Offline statistics for the tiny-scene traversal (traverse_flat, jp_device.h): how many leaf boxes / primitive tests does a
ray pay, and what does a 64-lane wave pay (max over lanes), for different groupings of the primitives into flat entries.
CPU only:  python tools/flat_stats.py [scene]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import harness as H


def arr(p, n, shape=None):
    a = np.ctypeslib.as_array(p, (n,)).copy()
    return a if shape is None else a.reshape(shape)


def cluster(boxes, max_clusters=32, max_size=4, grow=1.0001):
    """agglomerative: merge the pair with the smallest surface-area increase; free merges (identical boxes) first"""
    cl = [([i], boxes[i].copy()) for i in range(len(boxes))]
    def area(b):
        d = np.maximum(b[3:] - b[:3], 0); return d[0] * d[1] + d[1] * d[2] + d[2] * d[0] + 1e-3 * d.sum()
    while True:
        best = None
        for i in range(len(cl)):
            for j in range(i + 1, len(cl)):
                if len(cl[i][0]) + len(cl[j][0]) > max_size: continue
                u = np.concatenate([np.minimum(cl[i][1][:3], cl[j][1][:3]), np.maximum(cl[i][1][3:], cl[j][1][3:])])
                inc = area(u) * (len(cl[i][0]) + len(cl[j][0])) - area(cl[i][1]) * len(cl[i][0]) - area(cl[j][1]) * len(cl[j][0])
                if best is None or inc < best[0]: best = (inc, i, j, u)
        if best is None: break
        free = best[0] <= (grow - 1.0) * area(best[3])
        if not free and len(cl) <= max_clusters: break
        _, i, j, u = best
        cl[i] = (cl[i][0] + cl[j][0], u); del cl[j]
    return cl


def main(name="cornell_lambert", n=20000):
    W = Hh = 64
    hb = H.SCENES[name](H.scenes.HostBackend(name), W, Hh)
    sp = hb.flatten(); s = sp.contents
    left = arr(s.bvh_left, s.n_bvh_nodes); right = arr(s.bvh_right, s.n_bvh_nodes); bnd = arr(s.bvh_bounds, 6 * s.n_bvh_nodes, (-1, 6))
    pidx = arr(s.bvh_prim_index, s.n_bvh_prim_indices)
    assert s.n_triangles == s.n_primitives, "triangle scenes only"
    p0 = arr(s.tri_p0, 3 * s.n_triangles, (-1, 3)); p1 = arr(s.tri_p1, 3 * s.n_triangles, (-1, 3)); p2 = arr(s.tri_p2, 3 * s.n_triangles, (-1, 3)); tn = arr(s.tri_n, 3 * s.n_triangles, (-1, 3))
    sidx = arr(s.prim_shape_index, s.n_primitives)
    host_leaves = []
    for i in range(s.n_bvh_nodes):
        if left[i] < 0:
            first = -left[i] - 1
            host_leaves.append(([int(sidx[pidx[first + k]]) for k in range(right[i])], bnd[i].copy()))
    tb = np.concatenate([np.minimum(np.minimum(p0, p1), p2), np.maximum(np.maximum(p0, p1), p2)], 1)
    groupings = {"host leaves (<=4)": host_leaves, "one entry per primitive": [([i], tb[i]) for i in range(len(tb))],
                 "clustered, free merges only": cluster(tb, 64, 4), "clustered to <=16 x4": cluster(tb, 16, 4)}
    # rays: camera hits -> bounce -> bounce rays (incoherent) and shadow rays towards the light
    rng = np.random.default_rng(3)
    L = H.oracle_lib(); oh = L.jp_oracle_scene_new(sp)
    pxy = (rng.random((n, 2)) * [W, Hh]).astype(np.float32)
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    L.jp_oracle_camera_rays(oh, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
    def otrace(o, d):
        m = o.shape[0]
        tmin = np.full(m, 0.001, np.float32); tmax = np.full(m, np.inf, np.float32)
        hit = np.zeros(m, np.int32); t = np.zeros(m, np.float32); prim = np.zeros(m, np.int32); nrm = np.zeros((m, 3), np.float32); pos = np.zeros((m, 3), np.float32)
        L.jp_oracle_trace(oh, m, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
        return hit, pos, nrm
    hit, pos, nrm = otrace(o, d)
    def bounce(pos, nrm, din):
        dd = rng.normal(size=pos.shape).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        return np.where(((dd * nrm).sum(1) * (-(din * nrm).sum(1)) < 0)[:, None], -dd, dd).astype(np.float32)
    d1 = bounce(pos, nrm, d)
    sets = {"camera": (o, d, np.full(n, np.inf, np.float32)), "bounce 1": (pos[hit > 0], d1[hit > 0], np.full(int((hit > 0).sum()), np.inf, np.float32))}
    lp = np.stack([rng.uniform(213, 343, n), np.full(n, 548.7), -rng.uniform(227, 332, n)], 1).astype(np.float32)
    sd = lp - pos; dist = np.linalg.norm(sd, axis=1); sd = (sd / dist[:, None]).astype(np.float32)
    sets["shadow"] = (pos[hit > 0], sd[hit > 0], (dist - 0.001).astype(np.float32)[hit > 0])
    for gname, g in groupings.items():
        print("== %s: %d entries, sizes %s" % (gname, len(g), np.bincount([len(c[0]) for c in g])))
        B = np.stack([c[1] for c in g]).astype(np.float32)
        e = np.maximum(np.abs(B[:, :3]), np.abs(B[:, 3:])) * 1e-6 + 1e-6
        B = np.concatenate([B[:, :3] - e, B[:, 3:] + e], 1)
        cnt = np.array([len(c[0]) for c in g])
        for sname, (ro, rd, tmx) in sets.items():
            with np.errstate(divide="ignore", invalid="ignore"):
                idr = 1.0 / rd
                t0 = (B[None, :, :3] - ro[:, None, :]) * idr[:, None, :]; t1 = (B[None, :, 3:] - ro[:, None, :]) * idr[:, None, :]
                tn_ = np.maximum(np.fmax(np.fmin(t0, t1), -np.inf).max(2), 0.001); tf = np.minimum(np.fmin(np.fmax(t0, t1), np.inf).min(2), tmx[:, None])
            ent = tn_ <= tf * 1.000002
            leaves = ent.sum(1); prims = (ent * cnt[None, :]).sum(1)
            m = (len(leaves) // 64) * 64
            wl = leaves[:m].reshape(-1, 64); wp = prims[:m].reshape(-1, 64)
            # wave cost model of the current kernel: outer loop = max popcount, inner loop = max count among the lanes' current leaves
            srt = {}
            for T in (256, 512, 1024, 4096):
                mm = (len(prims) // T) * T
                tp = np.sort(prims[:mm].reshape(-1, T), axis=1).reshape(-1, 64)
                srt[T] = prims[:mm].mean() / tp.max(1).mean()
            print("   %-9s tile-sorted by prim count: util %s" % (sname, "  ".join("T=%d: %.2f" % (T, u) for T, u in srt.items())))
            print("   %-9s boxes entered/ray %.2f  prim tests/ray %.2f | per wave: max boxes %.2f, max prim tests %.2f (lane-sum bound), util %.2f" % (
                sname, leaves.mean(), prims.mean(), wl.max(1).mean(), wp.max(1).mean(), prims[:m].mean() / wp.max(1).mean()))


if __name__ == "__main__":
    main(*(sys.argv[1:2]))


def lane_local_order(prims, K):
    """utilisation when every lane holds K rays and takes them in descending test count (no cross-lane movement)"""
    m = (len(prims) // (64 * K)) * 64 * K
    a = np.sort(prims[:m].reshape(-1, K, 64), axis=1)          # per lane: its K rays sorted
    return prims[:m].sum() / (a.max(2).sum() * 64.0)
