#!/bin/bash
# builds tests/diagnostics/wino_layer_bench against the in-tree library (run from the repo root, after `make` in csrc)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -I neurips-cellseg_amd/csrc tests/diagnostics/wino_layer_bench.hip \
    -L neurips-cellseg_amd/cellseg_amd -lcellseg_hip -Wl,-rpath,'$ORIGIN/../../neurips-cellseg_amd/cellseg_amd' \
    -o tests/diagnostics/wino_layer_bench
