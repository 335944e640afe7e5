#!/bin/bash
# L2 hit rate and vector-memory path busy counters of one gpu_ab.py variant: tools/pmc_cache_ab.sh TAG SPEC "ENV=.."
set -o pipefail
TAG=${1:-cache}; SPEC=${2:-bunny:800x600:64}; VAR=${3:-JETPBRT_LANES=1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p1.log 2> $OUT/p1.err || echo p1 failed
timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p2.log 2> $OUT/p2.err || echo p2 failed
timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p3.log 2> $OUT/p3.err || echo p3 failed
timeout -k 10 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $OUT/p4 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p4.log 2> $OUT/p4.err || echo p4 failed
cd $ROOT && python3 tools/pmc_any_table.py $OUT > $OUT/table.txt 2>&1
grep -A8 "persist" $OUT/table.txt | head -60; grep -l "exceeds\|failed" $OUT/*.err 2>/dev/null
