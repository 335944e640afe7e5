#!/bin/bash
# round 4: two builds of the kernel library, default walk and certified walk, interleaved: tools/r4_ab2.sh LIB_A LIB_B
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_ab2.log; : > $O
C=$ROOT/jet-pbrt_amd/csrc
for i in 1 2; do for L in $1 $2; do
  echo "== $L default" >> $O
  JETPBRT_AMD_LIB=$C/$L timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
  echo "== $L certified" >> $O
  JETPBRT_AMD_LIB=$C/$L JETPBRT_REFERENCE_TREE=2 timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
done; done
grep -E "^==|Msamples|FAILED" $O | cut -c1-300
