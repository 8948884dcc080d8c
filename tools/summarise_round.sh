#!/bin/bash
# After tools/round_evidence.sh <tag> has been merged back (gpurun_out/prof_<tag>*): write the round's summaries into profiles/.
#   bash tools/summarise_round.sh r06
set -euo pipefail
cd "$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-r06}
one() {   # <suffix> <extra bench args...>
    local sfx=$1; shift
    local d=gpurun_out/prof_${TAG}${sfx}
    local stats fetch write sq
    stats=$(ls -t $(find $d/stats -name "*kernel_stats.csv") | head -1)
    fetch=$(ls -t $(find $d/fetch -name "*counter_collection.csv") | head -1)
    write=$(ls -t $(find $d/write -name "*counter_collection.csv") | head -1)
    sq=$(ls -t $(find $d/sq -name "*counter_collection.csv") | head -1)
    cp "$stats" profiles/rocprofv3_kernel_stats_${TAG}${sfx}.csv
    python3 tools/pmc_summary.py ${TAG}${sfx} "$stats" "$fetch" "$write" "$@"
    python3 tools/pmc_mfma_summary.py ${TAG}${sfx} "$sq" "$@"
}
one ""
one _bf16 --precision bf16
one _fp8c5 --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1
ls -la profiles/*${TAG}*
