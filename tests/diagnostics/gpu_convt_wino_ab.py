"""Diagnostic (GPU): eval-mode logits of the 32-base and 64-base nets with the F(2,2) forms of the stride-2 / transposed layers switched off and on (CS_CONVT_WINO; one process per setting), on shapes with partial tiles.  Prints the max abs differences."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, torch
sys.path[:0] = [{root!r}, {root!r} + "/neurips-cellseg_amd"]
import cellseg_amd as ca
dev = torch.device("cuda:0")
outs = {{}}
for tag, channels, shape in (("a", (32, 64, 128, 256, 512), (3, 3, 272, 208)), ("b", (64, 128, 256), (5, 3, 100, 76)), ("c", (32, 64, 128, 256, 512), (2, 3, 256, 256))):
    torch.manual_seed(11)
    m = ca.UNet(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=2).to(dev).eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        outs[tag] = m(x).cpu()
torch.save(outs, sys.argv[1])
"""
res = {}
for s2, ct in ((0, 0), (0, 1)):
    path = f"/tmp/s2ab_{s2}{ct}.pt"
    subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), path], check=True, env=dict(os.environ, CS_S2_WINO=str(s2), CS_CONVT_WINO=str(ct)))
    res[(s2, ct)] = torch.load(path)
for key in ((0, 1),):
    for tag in res[(0, 0)]:
        d = (res[key][tag] - res[(0, 0)][tag]).abs().max().item()
        print("s2_wino", key[0], "convt_wino", key[1], tag, "max|dlogit|", f"{d:.3e}", "scale", f"{res[(0, 0)][tag].abs().max().item():.2f}",
              "bitwise equal" if d == 0 else "")
