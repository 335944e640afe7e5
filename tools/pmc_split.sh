#!/bin/bash
# VALU instruction counts of k_shade debug variants (tools/gpu_shade_split.py) -> gpurun_out/pmc_split/<variant>/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base SKIP_NEE SKIP_SAMPLE SKIP_BOTH; do
  if [ $v = base ]; then unset JETPBRT_AMD_LIB; else export JETPBRT_AMD_LIB=$ROOT/tools/variant_$v.so; fi
  OUT=$ROOT/gpurun_out/pmc_split/$v; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $ROOT/tools/gpu_shade_split.py > $OUT/out.txt 2> $OUT/err.txt || echo "$v failed"
done
