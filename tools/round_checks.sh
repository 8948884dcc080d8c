#!/bin/bash
# before a round's final records: the classic randomised sweep (CC chain, strips, tiler, interior map, metric) and the whole GPU suite
set -uo pipefail
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:?}"
rm -f gpurun_out/fuzz_${TAG}.jsonl
for s in 61 62 63; do timeout -k 10 200 python3 tests/diagnostics/gpu_fuzz.py --seconds 120 --seed $s 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/fuzz_${TAG}.jsonl | cut -c1-300; done
timeout -k 10 200 python3 tests/diagnostics/gpu_fuzz.py --seconds 60 --seed 64 --channels 16 32 64 128 256 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/fuzz_${TAG}.jsonl | cut -c1-300
BLANK_REPORT=$PWD/gpurun_out/blank_report.jsonl timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_suite.log | cut -c1-300 | tail -25
