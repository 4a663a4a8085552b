#!/bin/bash
# Run GPU steps one after another on the gpurun box: each under its own timeout, output to gpurun_out/<name>.log.
# An ordinary failure (assert, exception) does not stop the sequence; a timeout or a kill (124 / 137) does - no further
# GPU step is started after one had to be killed.
#   scripts/gpu_seq.sh "name1|seconds|command ..." "name2|seconds|command ..."
mkdir -p gpurun_out
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== [$name] (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] rc=$rc in $(( $(date +%s) - start ))s"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== [$name] was killed at its limit: stopping the sequence"
        exit $rc
    fi
done
exit 0
