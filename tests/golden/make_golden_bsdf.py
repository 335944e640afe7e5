#!/usr/bin/env python3
"""Golden KATs for the whole reflection API by value: the UNMODIFIED reference's BSDF classes -- including the
ones no material instantiates: FPhongSpecularReflection, BeckmannDistribution, FMicrofacetTransmission, FresnelNoOp, the
non-visible-area sampling branches -- are constructed directly (oracle/ref_build/ref_driver.cc: ref_bsdf_direct) and FBSDF::Evalf /
Pdf / Sample are called on 384 shading events each.  Fixture = the reference's outputs (inputs are regenerated from the seed by
tests/harness.py: bsdf_cases / bsdf_inputs).  Run where /root/reference exists:  python tests/golden/make_golden_bsdf.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness as H  # noqa: E402

N, SEED = 384, 77


def main():
    assert H.have_ref()
    L = H.ref_lib()
    out = {}
    nrm, wo, wi, u = H.bsdf_inputs(N, SEED)
    out["in_nrm"], out["in_wo"], out["in_wi"], out["in_u"] = nrm, wo, wi, u
    for name, desc in H.bsdf_cases().items():
        r = H.run_bsdf(L.ref_bsdf_direct, desc, nrm, wo, wi, u)
        for k, v in r.items():
            out["%s__%s" % (name, k)] = v
        print("%-22s f!=0: %3d  pdf>0: %3d  sampled: %3d  nan: %d" % (name, int((r["f"] != 0).any(1).sum()), int((r["pdf"] > 0).sum()), int((r["spdf"] > 0).sum()),
                                                                  int(sum(np.isnan(v).sum() for v in r.values() if v.dtype == np.float32))))
    np.savez_compressed(os.path.join(HERE, "kat_bsdf.npz"), **out)


if __name__ == "__main__":
    main()
