#!/usr/bin/env python3
"""One engine forward over a batch of windows (for `rocprofv3 --kernel-trace`: per-layer launch durations).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/layers -- python3 tools/one_forward.py --batch 256 --width baseline32
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca

WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256)}
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--width", choices=sorted(WIDTHS), default="baseline32")
ap.add_argument("--precision", default="fp32")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(2022)
model = ca.UNet(channels=WIDTHS[args.width], strides=(2, 2, 2, 2), num_res_units=2).to(dev).eval()
model.precision = args.precision
x = torch.rand(args.batch, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(args.reps):
        y = model(x)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
