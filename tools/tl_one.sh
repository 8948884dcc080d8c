# usage: bash tools/tl_one.sh <tag> <gpu_train_timeline.py args...>: kernel-trace timeline of the training step -> gpurun_out/tl_<tag>.txt
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"        # the repo root: the gpurun box exports it; elsewhere derived from this script's own path
tag=$1; shift
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python3 tests/diagnostics/gpu_train_timeline.py "$@" > gpurun_out/tl_$tag.log 2>&1
f=$(find gpurun_out/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f --list > gpurun_out/tl_$tag.txt 2>&1
rm -rf gpurun_out/tl_$tag
tail -1 gpurun_out/tl_$tag.log
