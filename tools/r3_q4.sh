#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r3_q4.log; : > $O
timeout -k 10 300 python tools/gpu_q4_check.py >> $O 2>&1 || { echo FAILED check >> $O; tail -20 $O; exit 1; }
timeout -k 10 500 python tools/gpu_ab.py bunny:800x600:512 "JETPBRT_Q4=0" "" "JETPBRT_Q4_SHADOW=0" "" >> $O 2>&1 || { echo FAILED ab >> $O; tail -20 $O; exit 1; }
tail -20 $O
