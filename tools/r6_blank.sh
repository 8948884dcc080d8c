#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
rm -f gpurun_out/blank_report.jsonl
BLANK_REPORT=$PWD/gpurun_out/blank_report.jsonl timeout -k 10 1000 python3 -m pytest tests/test_gpu_blank_windows.py -v 2>&1 | grep -v amdgpu.ids | tee gpurun_out/blank_tests.log | cut -c1-600
