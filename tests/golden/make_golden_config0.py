#!/usr/bin/env python3
"""Golden for BASELINE.json configs[0] (cornell_box 256x256, 16 spp, the reference's own CPU path through parallel.cc): the
UNMODIFIED reference (oracle/_ref/libjp_ref.so) renders the frame with its stock FRandomSampler on its FParallelSystem; the
fixture keeps the SHA-256 of the raw fp32 film, four film rows and the ray statistics (data only).  Run where /root/reference
exists:  python tests/golden/make_golden_config0.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402


def main():
    assert H.have_ref()
    W = Hh = 256; spp = 16
    out = {}
    for name in ("cornell", "cornell_lambert"):
        H.libc_srand(1)
        rb = H.SCENES[name](H.RefBackend(name), W, Hh)
        H.ref_lib().ref_counters_reset(rb.h)
        film = rb.render(W, Hh, spp, 5, 0, 1234, 16)                 # sampler_mode 0: FRandomSampler(1234) per 20-row task, 16 threads (main.cc:156)
        out[name] = {"width": W, "height": Hh, "spp": spp, "sha256": hashlib.sha256(np.ascontiguousarray(film).tobytes()).hexdigest(),
                     "mean": float(film.mean(dtype=np.float64)), "counts": [int(v) for v in rb.counters()],
                     "rows": {str(y): film[y].reshape(-1).tolist() for y in (0, 77, 128, 255)}}
    json.dump(out, open(os.path.join(HERE, "config0.json"), "w"))
    print({k: (v["sha256"][:16], v["mean"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
