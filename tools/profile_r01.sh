#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes of the default bench workload.
# Summaries are written under gpurun_out/prof_$TAG/ ; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive > $OUT/bench_trace.json 2> $OUT/trace.err || echo "trace failed" >> $OUT/status
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-exclusive --spp 128 > $OUT/bench_fetch.json 2> $OUT/fetch.err || echo "fetch failed" >> $OUT/status
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-exclusive --spp 128 > $OUT/bench_write.json 2> $OUT/write.err || echo "write failed" >> $OUT/status
find $OUT -name "*.csv" | head -30 > $OUT/files.txt
ls -laR $OUT | head -60
