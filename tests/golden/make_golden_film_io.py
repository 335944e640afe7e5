#!/usr/bin/env python3
"""Golden for the film output step (FFilm::SaveAsImage, film.cc:11-188): the UNMODIFIED reference
(oracle/_ref/libjp_ref.so) writes a BMP and an HDR of a deterministic fp32 film whose width needs no row padding
(width * 3 % 4 == 0 -- for other widths the reference's BMP writer shears the image, film.cc:137-141) and whose pixels are all
>= 1e-32 (below that its HDR writer emits uninitialised bytes, film.cc:158-179); plus gamma_encoding (film.h:24) of 4096 values
around every step of the 8-bit curve.  Fixture = inputs + the reference's output bytes (data only).
Run where /root/reference exists:  python tests/golden/make_golden_film_io.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402


def main():
    assert H.have_ref()
    L = H.ref_lib()
    rng = np.random.default_rng(20261005)
    W, Hh = 12, 7
    film = rng.random((Hh, W, 3)).astype(np.float32) ** 2.0
    film[0, 0] = (0.0001, 1.0, 0.5); film[1, 2] = (1.0, 1.0, 1.0); film[2, 3] = (3.5, 0.25, 1e-6)      # saturated / above 1 / tiny
    out = {"film": film}
    d = tempfile.mkdtemp(prefix="jp_film_")
    for t, ext in ((1, "bmp"), (2, "hdr")):
        base = os.path.join(d, "ref_%s" % ext)
        assert L.ref_film_save(H.ptr(np.ascontiguousarray(film)), W, Hh, base.encode(), t) == 1
        out["file_" + ext] = np.frombuffer(open(base + "." + ext, "rb").read(), np.uint8)
    # the 8-bit curve: random values, and the neighbourhood of every step (found on the reference's own function)
    x = rng.random(2048).astype(np.float32)
    steps = []
    for k in range(1, 256):
        lo, hi = np.uint32(0), np.uint32(0x3f800000)
        while hi - lo > 1:
            mid = np.uint32((int(lo) + int(hi)) // 2)
            v = np.array([mid], np.uint32).view(np.float32); o = np.zeros(1, np.uint8)
            L.ref_gamma_encode(H.ptr(v), 1, H.ptr(o))
            if o[0] >= k: hi = mid
            else: lo = mid
        steps += [int(hi) - 2, int(hi) - 1, int(hi), int(hi) + 1]
    xs = np.concatenate([x, np.array(steps, np.uint32).view(np.float32), np.array([0.0, 1.0, 1.5, -0.25, 1e-30], np.float32)]).astype(np.float32)
    enc = np.zeros(xs.size, np.uint8)
    L.ref_gamma_encode(H.ptr(xs), xs.size, H.ptr(enc))
    out["gamma_x"] = xs; out["gamma_y"] = enc
    np.savez_compressed(os.path.join(HERE, "film_io.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "bmp bytes", out["file_bmp"].size, "hdr bytes", out["file_hdr"].size)


if __name__ == "__main__":
    main()
