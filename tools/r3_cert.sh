#!/bin/bash
# reference semantics on the 280k-triangle scene: verbatim walk vs certified walk, cull-slack sweep: tools/r3_cert.sh [spec] [variants...]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; O=gpurun_out/r3_cert.log; : > $O
export JETPBRT_REFERENCE_TREE=1
SPEC=${1:-bunny:800x600:32}; shift
if [ $# -eq 0 ]; then set -- "JETPBRT_CERTIFIED=0" "JETPBRT_CERTIFIED=1" "JETPBRT_CERT_SLACK=16" "JETPBRT_CERT_SLACK=256" "JETPBRT_CERT_SLACK=0"; fi
timeout -k 10 900 python tools/gpu_ab.py $SPEC "$@" >> $O 2>&1 || { echo FAILED >> $O; tail -30 $O; exit 1; }
cat $O
