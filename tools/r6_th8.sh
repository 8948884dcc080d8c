#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 600 python3 -m pytest tests/test_gpu_layers.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for t in 0 1; do
  echo "== CS_CONVT_TH8=$t"
  CS_CONVT_TH8=$t python3 bench.py --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null > /tmp/l.json
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); k=d['kernels']
print('configs4', d['ms_per_step'], d['ms_per_step_pipelined'], 'convT bf16', k['convT3x3s2_mfma'])"
  CS_CONVT_TH8=$t python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null > /tmp/l.json
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); k=d['kernels']
print('bf16 slide', d['ms_per_step'], d['ms_per_step_pipelined'], 'convT', k['convT3x3s2_mfma'])"
done
