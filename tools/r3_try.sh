#!/bin/bash
# round-3 bring-up of the fused schedule: small bit-identity A/Bs first, then the two benchmark frames
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r3_try.log; : > $O
run() { echo "## $*" >> $O; timeout -k 10 ${T:-240} python tools/gpu_ab.py "$@" >> $O 2>&1 || { echo "FAILED rc=$? : $*" >> $O; return 1; }; }
run cornell:128x128:32 "JETPBRT_FUSED=0" "" "JETPBRT_REGION=2048" "JETPBRT_JOB_SPP=4" "JETPBRT_REGION=512 JETPBRT_JOB_SPP=8" &&
run cornell_lambert:128x128:32 "JETPBRT_FUSED=0" "" &&
run bunny:200x152:16 "JETPBRT_FUSED=0" "" "JETPBRT_REGION=2048" &&
run cornell:512x512:1024 "JETPBRT_FUSED=0" "" "JETPBRT_FUSED_WGS=2" &&
run bunny:800x600:512 "JETPBRT_FUSED=0" "" 
tail -30 $O
