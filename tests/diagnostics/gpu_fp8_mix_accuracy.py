"""Diagnostic (GPU): accuracy of the fp8 mode as a function of CS_FP8_MIN_C (the channel count from which a layer takes the e4m3
kernels; narrower layers run on the bf16 kernels) - bench.lowprec_accuracy on 8 held-out 1024^2 synthetic images, one process per
setting (the variable is read once per process).

    python tests/diagnostics/gpu_fp8_mix_accuracy.py  > profiles/fp8_mix_accuracy_r04.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import json, sys, torch
sys.path[:0] = [{root!r}, {root!r} + "/neurips-cellseg_amd"]
from bench import lowprec_accuracy
r = lowprec_accuracy(torch.device("cuda:0"), n_images=8, precisions=("bf16", "fp8"))
print(json.dumps(r))
"""


def main():
    out = {}
    for min_c in (0, 64, 128, 256):
        cp = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=900,
                            env=dict(os.environ, CS_FP8_MIN_C=str(min_c)))
        if cp.returncode != 0:
            out[str(min_c)] = {"error": cp.stderr[-800:]}
            continue
        r = json.loads(cp.stdout.strip().splitlines()[-1])
        out[str(min_c)] = {"trained_net": r["trained_net"], "bf16": r["bf16"], "fp8": r["fp8"]}
        print(min_c, r["trained_net"]["cells_found_fp32"], r["bf16"], r["fp8"], file=sys.stderr)
    print(json.dumps({"note": "fp8 mode with layers below CS_FP8_MIN_C channels on the bf16 kernels (0 = every layer e4m3, the round-2/3 behaviour)",
                      "by_min_channels": out}))


if __name__ == "__main__":
    main()
