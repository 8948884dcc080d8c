#!/usr/bin/env python3
"""K training steps of cellseg_amd.TrainStep and nothing else (run under `rocprofv3 --kernel-trace` for a launch timeline;
tools/kernel_timeline.py reads the trace).  python3 tests/diagnostics/gpu_train_timeline.py [--width reference16] [--batch 8] [--precision fp32]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca

WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256)}
ap = argparse.ArgumentParser()
ap.add_argument("--width", default="reference16")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(2022)
model = ca.UNet(channels=WIDTHS[a.width], strides=(2, 2, 2, 2), num_res_units=2).to(dev)
model.precision = a.precision
model.train()
x = torch.rand(a.batch, 3, 256, 256, device=dev)
t = ca.one_hot(torch.randint(0, 3, (a.batch, 1, 256, 256), device=dev), 3)
step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
for _ in range(5):
    step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step(x, t)
torch.cuda.synchronize()
print(f"{a.width} x {a.batch} {a.precision}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step")
