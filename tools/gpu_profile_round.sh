#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (outputs under gpurun_out/prof_<tag>/); summarise afterwards with
#   python tools/pmc_summary.py <tag> <stats.csv> <fetch.csv> <write.csv>;  python tools/pmc_mfma_summary.py <tag> <sq.csv>
# All passes use --streams 1: with the default two internal streams kernels overlap and per-kernel durations stretch.
# Counter passes run on their own (no tracing flags), one --pmc set per run, the program itself after "--".
set -e
TAG=${1:-r01}
EXTRA="${@:2}"          # extra bench.py arguments (e.g. --precision bf16); the tag should then say so (r02_bf16)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary $EXTRA > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary $EXTRA > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary $EXTRA > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 0 --streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary $EXTRA > $O/sq.log 2>&1
find $O -name "*.csv" | sort
