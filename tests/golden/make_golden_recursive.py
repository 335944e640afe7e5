#!/usr/bin/env python3
"""Golden films of the reference's FPathIntegratorRecursive (integrator.cc:233-307) from the UNMODIFIED reference
(oracle/_ref/libjp_ref.so): film_<scene>_counter_recursive.npy, 48x48, 8 spp, maxDepth 5, counter sampler seed 1234 --
the same renders as film_<scene>_counter.npy (FPathIntegratorIteration), for the "same estimator" check.
Run where /root/reference exists:   python tests/golden/make_golden_recursive.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402

assert H.have_ref()
for name in ("cornell", "misc", "lights"):
    H.libc_srand(1)
    rb = H.SCENES[name](H.RefBackend(name), 48, 48)
    rec = rb.render_recursive(48, 48, 8, 5, 1234)
    it = np.load(os.path.join(HERE, "film_%s_counter.npy" % name))
    np.save(os.path.join(HERE, "film_%s_counter_recursive.npy" % name), rec)
    d = np.sqrt(((rec - it) ** 2).sum(-1))
    print(name, "recursive vs iterative reference films: mean L2 %.3e max %.3e, bit-identical pixels %.4f" % (d.mean(), d.max(), (rec == it).all(-1).mean()))
    rb.close()
