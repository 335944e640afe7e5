#!/bin/bash
# A/B of two builds of the kernel library on one frame, interleaved (A B A B): tools/r3_lib_ab.sh LIB_A LIB_B [spec] [check-test-filter for LIB_B]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r3_lib_ab.log; : > $O
A=$ROOT/jet-pbrt_amd/csrc/$1; B=$ROOT/jet-pbrt_amd/csrc/$2; SPEC=${3:-bunny:800x600:512}
if [ -n "$4" ]; then ( unset JETPBRT_REFERENCE_TREE; JETPBRT_AMD_LIB=$B timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "$4" ) >> $O 2>&1 || { echo FAILED tests >> $O; tail -30 $O; exit 1; }; fi
for i in 1 2; do
  for L in $A $B; do
    echo "== $(basename $L)" >> $O
    JETPBRT_AMD_LIB=$L timeout -k 10 200 python tools/gpu_ab.py $SPEC "" >> $O 2>&1 || { echo FAILED ab >> $O; tail -20 $O; exit 1; }
  done
done
grep -E "^==|Msamples|sha|film|passed|failed" $O
