#!/bin/bash
# round 4: node-step diet of the 4-wide walkers (no valid bits, v_rcp reciprocal direction, integer-difference ranks) -- parity tests, then the shipped build vs the round-3 node step, interleaved
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_diet.log; : > $O
C=$ROOT/jet-pbrt_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "four_wide or certified or device_build or device_built or bunny or trace_and_whitted or fused or config3 or config4 or wavefront or options_by_value" >> $O 2>&1 || { echo FAILED tests >> $O; tail -30 $O; exit 1; }
for i in 1 2; do for L in libjetpbrt_amd_base.so libjetpbrt_amd.so; do
  echo "== $L" >> $O
  JETPBRT_AMD_LIB=$C/$L timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
done; done
for L in libjetpbrt_amd_base.so libjetpbrt_amd.so; do
  echo "== $L certified" >> $O
  JETPBRT_AMD_LIB=$C/$L JETPBRT_REFERENCE_TREE=2 timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
done
grep -E "^==|Msamples|passed|failed|FAILED" $O | cut -c1-330
