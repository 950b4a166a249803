#!/bin/bash
# power / clock of the GPU while a command runs: tools/power_trace.sh <out.txt> <command...>   (rocm-smi sampled every 0.25 s)
OUT=$1; shift
( while true; do rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|Temperature \(Sensor junction\)" | tr '\n' ' ' ; echo; sleep 0.25; done ) > $OUT &
SMI=$!
"$@"
kill $SMI 2>/dev/null
