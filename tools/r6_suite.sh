#!/bin/bash
# the whole GPU suite, progress to a file (no pipe into tail: a silent pipe looks hung)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
rm -f gpurun_out/blank_report.jsonl
BLANK_REPORT=$PWD/gpurun_out/blank_report.jsonl timeout -k 10 1150 python3 -m pytest tests -m gpu -q -x --durations=15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_suite.log | cut -c1-400 | tail -60
