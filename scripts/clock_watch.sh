#!/bin/bash
# Sample the shader clock / power of GPU 0 while a command runs (is the step power-limited?):
#   scripts/clock_watch.sh <out.log> <command ...>
out=$1; shift
( while true; do rocm-smi -d 0 --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.2; done ) > "$out" &
wp=$!
"$@"
rc=$?
kill $wp
exit $rc
