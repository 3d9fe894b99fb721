#!/bin/bash
# sample power / clocks while a command runs: tools/power_watch.sh out.log -- cmd args...
out=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "GPU\[0\]" | grep -i "power\|sclk\|mclk\|use" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" 2>&1 &
W=$!
"$@"
rc=$?
kill $W 2>/dev/null
exit $rc
