#!/bin/bash
# A/B of two builds of the kernel library on the same scenes: tools/ab_libs.sh BASE.so "SPEC" ["SPEC" ...]   (film hashes must agree)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BASE=$1; shift
for spec in "$@"; do
  JETPBRT_AMD_LIB=$ROOT/$BASE python3 $ROOT/tools/gpu_ab.py "$spec" "" | sed 's/^/base /'
  python3 $ROOT/tools/gpu_ab.py "$spec" "" | sed 's/^/new  /'
done
