#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG=${1:-r06}
# Round-end bench records (one gpurun call): the default line and the three training legs
python3 bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; echo "default: rc $?"; cp gpurun_out/bench_full.json gpurun_out/bench_full_${TAG}.json
python3 bench.py --mode train --precision fp32 --no-cpu-baseline > gpurun_out/bench_${TAG}_train_fp32.json 2>/dev/null; echo "train fp32: rc $?"
python3 bench.py --mode train --precision bf16 --no-cpu-baseline > gpurun_out/bench_${TAG}_train_bf16.json 2>/dev/null; echo "train bf16: rc $?"
python3 bench.py --mode train --width reference16 --train-batch 8 --no-cpu-baseline > gpurun_out/bench_${TAG}_train_reference16.json 2>/dev/null; echo "train ref16: rc $?"
for t in "ref16 --width reference16 --batch 8 --precision fp32" "bf16 --width baseline32 --batch 32 --precision bf16" "fp32 --width baseline32 --batch 32 --precision fp32"; do bash tools/tl_one.sh $t; done
