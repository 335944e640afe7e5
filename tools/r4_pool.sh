#!/bin/bash
# round 4: pooled regions in the refill kernels -- parity, pool sizes, turn statistics
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_pool.log; : > $O
timeout -k 10 300 python tools/gpu_ab.py bunny:200x152:16 "JETPBRT_POOL_MAX=1" "" "JETPBRT_POOL_MAX=32 JETPBRT_POOL_RAYS=100000" >> $O 2>&1 || { echo FAILED small >> $O; tail -20 $O; exit 1; }
timeout -k 10 400 python tools/gpu_ab.py bunny:800x600:512 "JETPBRT_POOL_MAX=1" "" "JETPBRT_POOL_MAX=2" "JETPBRT_POOL_MAX=8" "JETPBRT_POOL_MAX=4 JETPBRT_POOL_RAYS=32768" "JETPBRT_POOL_MAX=8 JETPBRT_POOL_RAYS=32768" "JETPBRT_POOL_MAX=16 JETPBRT_POOL_RAYS=65536" "JETPBRT_POOL_MAX=4 JETPBRT_POOL_RAYS=8192" "JETPBRT_POOL_MAX=1" >> $O 2>&1 || { echo FAILED big >> $O; tail -20 $O; exit 1; }
timeout -k 10 300 python tools/turn_stats.py 800x600:64 "JETPBRT_LANES=1 JETPBRT_POOL_MAX=1" "JETPBRT_LANES=1" "JETPBRT_LANES=1 JETPBRT_POOL_MAX=8 JETPBRT_POOL_RAYS=32768" >> $O 2>&1
grep -E "^##|^==|Msamples|rays|turns|FAILED" $O | cut -c1-300
