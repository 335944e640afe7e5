#!/bin/bash
# two SQ counter passes of one gpu_ab.py variant (single lane): tools/pmc_sq_ab.sh TAG SPEC "ENV=.." ; table: gpurun_out/pmc_TAG/table.txt
set -o pipefail
TAG=${1:-sq}; SPEC=${2:-bunny:800x600:64}; VAR=${3:-JETPBRT_LANES=1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p1.log 2> $OUT/p1.err || echo p1 failed
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p2.log 2> $OUT/p2.err || echo p2 failed
cd $ROOT && python3 tools/pmc_table.py $TAG > $OUT/table.txt 2>&1
grep -A22 "k_extend_persist\|k_shadow_persist" $OUT/table.txt | grep -v "^--" | head -60
