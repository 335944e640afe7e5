#!/bin/bash
# diagnostic PMC passes (instruction cache, memory-instruction levels, instruction mix) of one tools/gpu_ab.py variant:
#   tools/prof_diag.sh TAG "SCENE:WxH:SPP" "ENV=.."
set -o pipefail
TAG=$1; SPEC=$2; VAR=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export JETPBRT_LANES=1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH --output-format csv -d $OUT/p4 -- python3 $ROOT/tools/gpu_ab.py "$SPEC" "$VAR" > $OUT/p4.log 2>&1
python3 $ROOT/tools/diag_table.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt
