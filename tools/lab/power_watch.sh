#!/bin/bash
# samples power / clocks while a command runs:  power_watch.sh <outfile> <cmd...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|Temperature \(Sensor junction\)" | tr '\n' ' '; echo; sleep 0.2; done ) > $out 2>&1 &
w=$!
"$@"
kill $w
