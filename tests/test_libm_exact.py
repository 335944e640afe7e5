"""csrc/jp_libm.h -- the device's logf / expf / powf / acosf / atanf / tanf -- is glibc 2.35's arithmetic restated; this CPU test
compiles the header for the host (tests/libm_check.cc, g++ -ffp-contract=off) and compares every function with the running libm
BIT FOR BIT on 2 x 10^7 arguments (raw bit patterns of both signs and every exponent, and the ranges of the reference's call
sites: microfacet.cc:11-167, 326-357, bsdf.h:557-633).  The library's own probe (jp_create_context) must reach the same verdict."""
import ctypes as C
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libm_transcription_is_bit_exact_on_this_host(tmp_path):
    exe = str(tmp_path / "libm_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(REPO, "tests", "libm_check.cc"), "-lm"])
    out = subprocess.check_output([exe, "20000000"], text=True).split("\n")
    rows = {l.split()[0]: [int(v) for v in l.split()[1:]] for l in out if l.strip()}
    assert set(rows) == {"expf", "logf", "powf", "acosf", "atanf", "tanf", "tanf8"}
    for name, (bad_fma, bad_plain, n) in rows.items():
        assert n == 20000000
        if name in ("expf", "logf", "powf"):
            assert min(bad_fma, bad_plain) == 0, (name, bad_fma, bad_plain)      # one of libm's two builds (IFUNC: FMA + AVX2 or not)
        else:
            assert bad_fma == 0, (name, bad_fma)                                   # one build, no contraction


def test_library_probe_selects_the_transcription():
    import sys
    sys.path.insert(0, REPO)
    import jet_pbrt_amd as jp
    lib = C.CDLL(jp.HIP_LIB_PATH)
    mode = lib.jp_probe_libm_xbsdf()
    assert mode & 1, "jp_libm.h does not reproduce this host's libm: the by-value BSDF classes would fall back to a tolerance"
