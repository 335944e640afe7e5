#!/bin/bash
# clocks and power of the GPU while a command runs:  tools/smi_watch.sh OUT.log CMD...   (sampler = a child killed by PID when CMD ends)
OUT=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' >> "$OUT"; echo >> "$OUT"; sleep 0.25; done ) &
SMI=$!
"$@"
RC=$?
kill $SMI 2>/dev/null
exit $RC
