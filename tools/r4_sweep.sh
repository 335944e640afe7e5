#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=()
for L in 2 3 4; do for B in 4 5 6 8; do for R in 2 4 8; do V+=("JETPBRT_LANES=$L JETPBRT_BLOCKS_PER_CU=$B JETPBRT_LANE_ROWS=$R"); done; done; done
timeout -k 10 800 python tools/gpu_ab.py cornell:512x512:1024 "" "${V[@]}" "" > gpurun_out/r04o_c2_sweep.txt 2>&1
cut -c1-120 gpurun_out/r04o_c2_sweep.txt
