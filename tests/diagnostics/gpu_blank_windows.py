#!/usr/bin/env python3
"""Margins of the fp32 engine on the inputs real slides are made of: blank, constant and sparse windows, where
InstanceNorm's rstd grows towards 1/sqrt(eps) = 316 and amplifies every rounding difference.  For each net x input x Winograd form
(CS_WINO 0/1/2, one child process each): max |engine - torch-fp32 oracle|, and both against an fp64 evaluation of the oracle.

    python tests/diagnostics/gpu_blank_windows.py [out.json]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "neurips-cellseg_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from blank_inputs import blank_windows, make_nets  # noqa: E402


def main():
    from oracle.unet_ref import UNetRef
    nets = make_nets(trained=True)
    windows = blank_windows()
    rows = []
    with tempfile.TemporaryDirectory() as td:
        job = os.path.join(td, "job.pt")
        torch.save({"nets": nets, "windows": windows}, job)
        got = {}
        for form in ("0", "1", "2"):
            out = os.path.join(td, f"out{form}.pt")
            subprocess.run([sys.executable, os.path.join(HERE, "gpu_engine_eval.py"), job, out], check=True, env=dict(os.environ, CS_WINO=form), timeout=900)
            got[form] = torch.load(out)
    for nname, net in nets.items():
        ch = tuple(net["channels"])
        ref = UNetRef(channels=ch, strides=(2,) * (len(ch) - 1), num_res_units=2).eval()
        ref.load_state_dict(net["state"])
        ref64 = UNetRef(channels=ch, strides=(2,) * (len(ch) - 1), num_res_units=2).eval().double()
        ref64.load_state_dict({k: v.double() for k, v in net["state"].items()})
        for wname, x in windows.items():
            with torch.no_grad():
                t32 = ref(x)
                f64 = ref64(x.double())
            row = {"net": nname, "input": wname, "torch32_vs_fp64": float((t32.double() - f64).abs().max()), "logit_scale": float(f64.abs().max())}
            for form in ("0", "1", "2"):
                e = got[form][f"{nname}/{wname}"]
                row[f"wino{form}_vs_torch32"] = float((e - t32).abs().max())
                row[f"wino{form}_vs_fp64"] = float((e.double() - f64).abs().max())
            rows.append(row)
            print(json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
