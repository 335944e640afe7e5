#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of gpu_ab.py variants (single lane): tools/pmc_hbm_ab.sh TAG SPEC "ENV=.." ["ENV=.." ...]
set -o pipefail
TAG=$1; SPEC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for VAR in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$i; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_ab.py $SPEC "JETPBRT_LANES=1 $VAR" > $OUT/p1.log 2> $OUT/p1.err || echo p1 failed
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_ab.py $SPEC "JETPBRT_LANES=1 $VAR" > $OUT/p2.log 2> $OUT/p2.err || echo p2 failed
  echo "## $VAR" ; tail -1 $OUT/p1.log | cut -c1-200
  python3 $ROOT/tools/pmc_any_table.py $OUT | grep -A3 "k_shade\|k_extend\|k_shadow" | grep -v "^--"
  i=$((i+1))
done
