#!/bin/bash
# rocprofv3 kernel stats + SQ counters of one tools/gpu_ab.py variant:  tools/prof_ab.sh TAG "SCENE:WxH:SPP" "ENV=.. ENV=.."
set -o pipefail
TAG=$1; SPEC=$2; VAR=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/sq2.log 2>&1
cut -c1-110 $OUT/trace/*/*_kernel_stats.csv | head -8
