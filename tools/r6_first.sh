#!/bin/bash
# round 6, first GPU call: the rebuilt bench line (default run, timed), then the new split-slides CLI test
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
t0=$(date +%s)
python3 bench.py > gpurun_out/bench_r06_first.json 2> gpurun_out/bench_r06_first.err
echo "bench rc $? in $(( $(date +%s) - t0 )) s; line bytes $(tail -n 1 gpurun_out/bench_r06_first.json | wc -c)"
tail -n 1 gpurun_out/bench_r06_first.json
cp gpurun_out/bench_full.json gpurun_out/bench_full_r06_first.json
timeout -k 10 600 python3 -m pytest tests/test_gpu_ccl_strips.py -x -q -k "allgather_transport" 2>&1 | tail -5
