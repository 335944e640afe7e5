#!/bin/bash
# A/B of two builds of the kernel library on the bunny frame, interleaved (A B A B): tools/r3_lib_ab.sh LIB_B [spec] [check-test-filter]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r3_lib_ab.log; : > $O
A=$ROOT/jet-pbrt_amd/csrc/libjetpbrt_amd.so; B=$ROOT/jet-pbrt_amd/csrc/$1; SPEC=${2:-bunny:800x600:512}
if [ -n "$3" ]; then JETPBRT_AMD_LIB=$B timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "$3" >> $O 2>&1 || { echo FAILED tests >> $O; tail -30 $O; exit 1; }; fi
for i in 1 2; do
  for L in $A $B; do
    echo "== $(basename $L)" >> $O
    JETPBRT_AMD_LIB=$L timeout -k 10 200 python tools/gpu_ab.py $SPEC "" >> $O 2>&1 || { echo FAILED ab >> $O; tail -20 $O; exit 1; }
  done
done
grep -E "^==|Msamples|sha|film|passed|failed" $O
