#!/bin/bash
# usage: tools/power_trace.sh <out file> <command...>   -- samples rocm-smi (socket power, sclk) every 50 ms while the command runs
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' ' ; echo; sleep 0.05; done ) > "$out" &
spid=$!
"$@"
rc=$?
kill $spid
exit $rc
