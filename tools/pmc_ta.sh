#!/bin/bash
# TA / TCP (vector L1) counter passes for one scene: is the traversal bound by the vector-memory address / data path?
#   tools/pmc_ta.sh TAG "gpu_ab spec" ["ENV=.."]
set -o pipefail
TAG=${1:-ta}
SPEC=${2:-bunny:800x600:64}
VAR=${3:-JETPBRT_LANES=1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p1.log 2> $OUT/p1.err || echo p1 failed
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p2.log 2> $OUT/p2.err || echo p2 failed
rocprofv3 --pmc TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/gpu_ab.py $SPEC "$VAR" > $OUT/p3.log 2> $OUT/p3.err || echo p3 failed
python3 $ROOT/tools/pmc_any_table.py $OUT > $OUT/table.txt 2>&1
tail -5 $OUT/p1.log
