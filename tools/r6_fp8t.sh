#!/bin/bash
# A/B of the e4m3 transposed conv at 64 output channels (CS_FP8_MIN_COUT_T): configs[4] slide time and the 32-base fp8 slide
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for t in 128 64; do
  echo "== CS_FP8_MIN_COUT_T=$t"
  CS_FP8_MIN_COUT_T=$t python3 bench.py --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs4', d['ms_per_step'], d['ms_per_step_pipelined'], d.get('roofline_by_dtype'))"
  CS_FP8_MIN_COUT_T=$t python3 bench.py --precision fp8 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32-base fp8', d['ms_per_step'], d['ms_per_step_pipelined'])"
done
