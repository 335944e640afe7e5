#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the UNMODIFIED reference compiled in this container
(oracle/_ref/libjp_ref.so, built by oracle/ref_build/Makefile from /root/reference/src).  Run where
/root/reference exists:   python tests/golden/make_golden.py

Fixtures are data only -- inputs and the reference's outputs:
  film_<scene>_<stock|counter>.npy   raw fp32 films, 48x48, 8 spp, maxDepth 5 (FFilm::operator() values)
  kat.npz                            per-function known-answer vectors (camera rays, hit records, BSDF eval/sample
                                     for every material, light samples for every light, scripted Li, RNG stream)
  counts.json                        the reference's ray statistics for the same renders
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402

W = Hh = 48
SPP = 8


def main():
    assert H.have_ref(), "oracle/_ref/libjp_ref.so missing: run `make -C oracle ref` where /root/reference exists"
    L = H.ref_lib()
    counts = {}
    kat = {}
    rng = np.random.default_rng(20261004)
    for name in H.SCENES:
        H.libc_srand(1)
        rb = H.SCENES[name](H.RefBackend(name), W, Hh)
        for mode, tag in ((0, "stock"), (1, "counter")):
            L.ref_counters_reset(rb.h)
            film = rb.render(W, Hh, SPP, 5, mode, 1234, 4)
            np.save(os.path.join(HERE, "film_%s_%s.npy" % (name, tag)), film)
            counts["%s_%s" % (name, tag)] = [int(v) for v in rb.counters()]
        n = 512
        # camera rays
        pxy = (rng.random((n, 2)) * [W, Hh]).astype(np.float32)
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
        L.ref_camera_rays(rb.h, n, H.ptr(pxy), H.ptr(o), H.ptr(d))
        kat[name + "_cam_pxy"] = pxy; kat[name + "_cam_o"] = o; kat[name + "_cam_d"] = d
        # closest-hit records: the camera rays plus rays from the hit points in random directions
        tmin = np.full(n, 0.001, np.float32); tmax = np.full(n, np.inf, np.float32)
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); pos = np.zeros((n, 3), np.float32)
        L.ref_trace(rb.h, n, H.ptr(o), H.ptr(d), H.ptr(tmin), H.ptr(tmax), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
        for k, v in dict(hit=hit, t=t, prim=prim, nrm=nrm, pos=pos).items():
            kat["%s_tr1_%s" % (name, k)] = v.copy()
        o2 = np.where(hit[:, None] > 0, pos, o).astype(np.float32)
        d2 = rng.normal(size=(n, 3)).astype(np.float32); d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
        tmax2 = np.where(rng.random(n) < 0.5, np.inf, rng.random(n) * 600).astype(np.float32)
        L.ref_trace(rb.h, n, H.ptr(o2), H.ptr(d2), H.ptr(tmin), H.ptr(tmax2), H.ptr(hit), H.ptr(t), H.ptr(prim), H.ptr(nrm), H.ptr(pos))
        kat[name + "_tr2_o"] = o2; kat[name + "_tr2_d"] = d2; kat[name + "_tr2_tmax"] = tmax2
        for k, v in dict(hit=hit, t=t, prim=prim, nrm=nrm, pos=pos).items():
            kat["%s_tr2_%s" % (name, k)] = v.copy()
        # light samples for every light
        nl = rb.num_lights()
        p = (rng.random((n, 3)) * [500, 500, 500] + [20, 20, -520 if "bunny" not in name else -250]).astype(np.float32)
        if "bunny" in name:
            p = (rng.random((n, 3)) * [400, 300, 400] - [200, -5, 200]).astype(np.float32)
        nn = rng.normal(size=(n, 3)).astype(np.float32); nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
        u2 = rng.random((n, 2)).astype(np.float32)
        kat[name + "_ls_p"] = p; kat[name + "_ls_n"] = nn; kat[name + "_ls_u"] = u2
        for li in range(nl):
            lp = np.zeros((n, 3), np.float32); wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); Li = np.zeros((n, 3), np.float32)
            L.ref_light_sample(rb.h, n, li, H.ptr(p), H.ptr(nn), H.ptr(u2), H.ptr(lp), H.ptr(wi), H.ptr(pdf), H.ptr(Li))
            kat["%s_ls%d_pos" % (name, li)] = lp; kat["%s_ls%d_wi" % (name, li)] = wi; kat["%s_ls%d_pdf" % (name, li)] = pdf; kat["%s_ls%d_Li" % (name, li)] = Li
        kat[name + "_nlights"] = np.array([nl])
        # scripted Li: whole paths with given random numbers
        nv = 64
        vals = rng.random((n, nv)).astype(np.float32)
        ppx = (rng.integers(0, W, size=(n, 2))).astype(np.float32)
        out = np.zeros((n, 3), np.float32)
        L.ref_li_scripted(rb.h, n, 5, H.ptr(ppx), H.ptr(vals), nv, H.ptr(out))
        kat[name + "_li_pxy"] = ppx; kat[name + "_li_vals"] = vals; kat[name + "_li_out"] = out
        rb.close()
    # BSDF eval/sample for every material kind of the misc scene (+ cornell metal) on random frames
    H.libc_srand(1)
    rb = H.SCENES["misc"](H.RefBackend("misc"), W, Hh)
    n = 1024
    nn = rng.normal(size=(n, 3)).astype(np.float32); nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    wo = rng.normal(size=(n, 3)).astype(np.float32); wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    wi = rng.normal(size=(n, 3)).astype(np.float32); wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    # half of the pairs on the same side of the surface (so that reflection BSDFs are non-zero)
    flip = (np.sign((wo * nn).sum(1)) != np.sign((wi * nn).sum(1))) & (np.arange(n) % 2 == 0)
    wi[flip] = (wi[flip] - 2 * (wi[flip] * nn[flip]).sum(1, keepdims=True) * nn[flip]).astype(np.float32)
    u2 = rng.random((n, 2)).astype(np.float32); us = rng.random(n).astype(np.float32)
    kat["bsdf_n"] = nn; kat["bsdf_wo"] = wo; kat["bsdf_wi"] = wi; kat["bsdf_u2"] = u2; kat["bsdf_us"] = us
    nm = 8     # materials of the misc scene: red, green, white, metal, light-matte, glass, mirror, plastic(remap)
    for m in range(nm):
        fe = np.zeros((n, 3), np.float32); sf = np.zeros((n, 3), np.float32); swi = np.zeros((n, 3), np.float32)
        spdf = np.zeros(n, np.float32); sfl = np.zeros(n, np.int32); sd = np.zeros(n, np.int32)
        L.ref_bsdf(rb.h, n, m, H.ptr(nn), H.ptr(wo), H.ptr(wi), H.ptr(u2), H.ptr(us), H.ptr(fe), H.ptr(sf), H.ptr(swi), H.ptr(spdf), H.ptr(sfl), H.ptr(sd))
        for k, v in dict(feval=fe, sf=sf, swi=swi, spdf=spdf, sflags=sfl, delta=sd).items():
            kat["bsdf%d_%s" % (m, k)] = v
    kat["bsdf_nmat"] = np.array([nm])
    rb.close()
    s = np.zeros(4096, np.float32); L.ref_stock_stream(4096, H.ptr(s)); kat["stock_stream"] = s
    s2 = np.zeros(2, np.float32); L.ref_stock_float2(H.ptr(s2)); kat["stock_float2"] = s2
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **kat)
    json.dump(counts, open(os.path.join(HERE, "counts.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(kat), "KAT arrays and", len(counts), "films")


if __name__ == "__main__":
    main()
