#!/bin/bash
# which PMC counters the box offers (names only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $ROOT/gpurun_out/counters_avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|SQC_[A-Z_0-9]*" $ROOT/gpurun_out/counters_avail.txt | sort -u > $ROOT/gpurun_out/counters_sq.txt
wc -l $ROOT/gpurun_out/counters_sq.txt
