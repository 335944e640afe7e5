#!/bin/bash
# round 4: work lists in the refill kernels -- parity, pool sizes, turn statistics
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_wl.log; : > $O
timeout -k 10 300 python tools/gpu_ab.py bunny:200x152:16 "" "JETPBRT_POOL_ITEMS=1" "JETPBRT_POOL_ITEMS=32 JETPBRT_POOL_WGS=16" "JETPBRT_PERSIST=0" >> $O 2>&1 || { echo FAILED small >> $O; tail -20 $O; exit 1; }
timeout -k 10 400 python tools/gpu_ab.py bunny:800x600:512 "" "JETPBRT_POOL_WGS=8192" "JETPBRT_POOL_WGS=2048" "JETPBRT_POOL_WGS=16384" "JETPBRT_POOL_ITEMS=4 JETPBRT_POOL_WGS=16384" "JETPBRT_POOL_ITEMS=1" "JETPBRT_POOL_ITEMS=2 JETPBRT_POOL_WGS=8192" "" >> $O 2>&1 || { echo FAILED big >> $O; tail -20 $O; exit 1; }
timeout -k 10 300 python tools/turn_stats.py 800x600:64 "JETPBRT_LANES=1" "JETPBRT_LANES=1 JETPBRT_POOL_WGS=8192" >> $O 2>&1
grep -E "^##|^==|Msamples|rays|turns|FAILED" $O | cut -c1-300
