#!/bin/bash
set -o pipefail
TAG=${1:-c3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --spp 64 --full-materials > $OUT/b1.json 2> $OUT/p1.err || echo p1 failed
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --spp 64 --full-materials > $OUT/b2.json 2> $OUT/p2.err || echo p2 failed
