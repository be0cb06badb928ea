#!/bin/bash
# sample the GPU's clocks twice a second while a command runs:  bash tools/clock_watch.sh <out> -- <command ...>
OUT=$1; shift; shift
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks 2>/dev/null | grep -a 'clock level' | sed 's/GPU\[0\]\s*: //' | tr '\n' ';')"; sleep 0.4; done ) > "$OUT" &
W=$!
"$@"
RC=$?
kill $W
exit $RC
