#!/bin/bash
# sample GPU clock / power with rocm-smi while a command runs:  tools/clock_watch.sh <logfile> <cmd...>
log=$1; shift
( while true; do
    echo "t=$(date +%s.%N)"
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)"
  done ) > $log 2>&1 &
wp=$!
"$@"
rc=$?
kill $wp
exit $rc
