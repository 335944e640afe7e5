#!/bin/bash
# round 4: two rays per lane -- parity (films vs the one-ray kernels, bit for bit), occupancy variants, turn statistics
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_dual.log; : > $O
C=$ROOT/jet-pbrt_amd/csrc
echo "## small parity: one-ray kernels first, then two rays per lane" >> $O
timeout -k 10 300 python tools/gpu_ab.py bunny:200x152:16 "JETPBRT_DUAL_RAYS=0" "" "JETPBRT_PERSIST=8" >> $O 2>&1 || { echo FAILED small >> $O; tail -20 $O; exit 1; }
for L in libjetpbrt_amd.so libjetpbrt_amd_d5.so libjetpbrt_amd_d4.so; do
  echo "== $L" >> $O
  JETPBRT_AMD_LIB=$C/$L timeout -k 10 300 python tools/gpu_ab.py bunny:800x600:512 "JETPBRT_DUAL_RAYS=0" "" >> $O 2>&1 || { echo FAILED $L >> $O; tail -20 $O; exit 1; }
done
timeout -k 10 300 python tools/turn_stats.py 800x600:64 "JETPBRT_LANES=1 JETPBRT_DUAL_RAYS=0" "JETPBRT_LANES=1" >> $O 2>&1
grep -E "^##|^==|Msamples|rays|turns|FAILED" $O | cut -c1-300
