#!/bin/bash
# Runs GPU steps one after another on the gpurun box: tools/gpu_steps.sh "<name>|<seconds>|<command>" ...
# Every step runs under `timeout -k 10`, logs to gpurun_out/<name>.log and prints a progress line; an ordinary failure (a failing
# test) does not stop the sequence, a step that had to be KILLED at its limit does (no further GPU step after a hang).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "[gpu_steps] $(date +%T) start $name (limit ${secs}s)"
    timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "[gpu_steps] $(date +%T) end   $name rc=$rc"
    tail -n 3 "gpurun_out/$name.log" | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[gpu_steps] $name was killed at its limit: stopping here"
        exit 1
    fi
done
exit 0
