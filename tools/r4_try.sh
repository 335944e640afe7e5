#!/bin/bash
# round-4 bring-up of the postponed-leaf / triangle-record walkers: parity tests first, then A/Bs on the 280k-triangle scene
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r4_try.log; : > $O
run() { echo "## $*" >> $O; timeout -k 10 ${T:-300} python tools/gpu_ab.py "$@" >> $O 2>&1 || { echo "FAILED rc=$? : $*" >> $O; return 1; }; }
if [ -n "$K" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" >> $O 2>&1 || { echo FAILED tests >> $O; tail -40 $O; exit 1; }; fi
run "$@"
grep -E "^##|Msamples|passed|failed|FAILED" $O | tail -40
