#!/bin/bash
# Round-end evidence on the gpurun box (one call): rocprofv3 stats + PMC passes for the default command, the bf16 datapath and configs[4].
# Summaries are made afterwards, here: tools/pmc_summary.py / tools/pmc_mfma_summary.py.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG=${1:-r06}
bash tools/gpu_profile_round.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; echo "fp32 done: $(tail -1 gpurun_out/prof_$TAG.log)"
bash tools/gpu_profile_round.sh ${TAG}_bf16 --precision bf16 > gpurun_out/prof_${TAG}_bf16.log 2>&1; echo "bf16 done"
bash tools/gpu_profile_round.sh ${TAG}_fp8c5 --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1 > gpurun_out/prof_${TAG}_fp8c5.log 2>&1; echo "fp8c5 done"
# keep only the CSVs the summaries read (the merge back is capped at 64 MiB)
find gpurun_out/prof_${TAG}* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" -delete
du -sh gpurun_out/prof_${TAG}* | tail -5
