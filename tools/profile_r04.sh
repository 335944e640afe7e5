#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 passes of ONE BASELINE.json config of bench.py.
#   tools/profile_r04.sh TAG CONFIG [PMC_SPP]
# pass 1: --kernel-trace --stats at the full size (2 timed frames)          -> per-kernel average durations
# pass 2/3: --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, reduced spp) -> HBM bytes per launch
# pass 4/5: two SQ counter sets (reduced spp)                                -> instruction mix, stalls, lane utilisation
# pass 6 [round 4]: SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE                    -> VALUBusy per kernel (vector-issue occupancy: the roof these kernels sit at)
# The program comes directly after `--` (no env / bash -c hop).  Summaries: tools/summarize_r04.py TAG CONFIG -> profiles/.
set -o pipefail
TAG=${1:-r04}
CFG=${2:-2}
PSPP=${3:-128}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--config $CFG --configs= --no-cpu --no-exclusive"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 $COMMON > $OUT/bench_trace.json 2> $OUT/trace.err || echo "trace failed" >> $OUT/status
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON --spp $PSPP > $OUT/bench_fetch.json 2> $OUT/fetch.err || echo "fetch failed" >> $OUT/status
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON --spp $PSPP > $OUT/bench_write.json 2> $OUT/write.err || echo "write failed" >> $OUT/status
timeout -k 10 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON --spp $PSPP > $OUT/bench_sq1.json 2> $OUT/sq1.err || echo "sq1 failed" >> $OUT/status
timeout -k 10 420 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON --spp $PSPP > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 failed" >> $OUT/status
timeout -k 10 420 rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/vb -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON --spp $PSPP > $OUT/bench_vb.json 2> $OUT/vb.err || echo "vb failed" >> $OUT/status
ls $OUT; cat $OUT/status 2>/dev/null; tail -c 600 $OUT/bench_trace.json
