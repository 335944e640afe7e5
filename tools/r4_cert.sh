#!/bin/bash
# round 4: the certified walk at 8 vs 7 waves per SIMD (JP_CERT_WAVES builds), interleaved; then the GPU suite on the refactored library
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_cert.log; : > $O
C=$ROOT/jet-pbrt_amd/csrc
for i in 1 2; do for L in libjetpbrt_amd.so libjetpbrt_amd_c7.so; do
  echo "== $L" >> $O
  JETPBRT_AMD_LIB=$C/$L JETPBRT_REFERENCE_TREE=2 timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
done; done
grep -E "^==|Msamples|FAILED" $O | cut -c1-330
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest_b.log 2>&1; tail -6 gpurun_out/r4_gputest_b.log
