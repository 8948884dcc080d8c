#!/usr/bin/env python3
"""Diagnostic: the training step three ways - autograd loop (the reference's body, launch by launch), ca.TrainStep eager (same C-ABI
calls, no autograd), ca.TrainStep captured (hipGraph replay) - at the reference's setting (16...256, batch 8) and at 32-base x 32.
    python tests/diagnostics/gpu_train_step_modes.py [only=<captured|lean|autograd>] [cfg=<ref|base>] [steps=N] [precision=fp32|fp16x3|bf16]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch  # noqa: E402

import cellseg_amd as ca  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
dev = torch.device("cuda:0")
CFG = {"ref": ((16, 32, 64, 128, 256), 8), "base": ((32, 64, 128, 256, 512), 32)}
steps = int(opts.get("steps", 30))


def run(channels, b, mode):
    torch.manual_seed(0)
    model = ca.UNet(channels=channels, strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
    model.precision = opts.get("precision", "fp32")
    x = torch.rand(b, 3, 256, 256, device=dev)
    y = ca.one_hot(torch.randint(0, 3, (b, 1, 256, 256), device=dev), 3)
    loss_fn, opt = ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4)
    if mode == "autograd":
        def step():
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(model(x), y)
            loss.backward()
            opt.step()
            return loss
    else:
        ts = ca.TrainStep(model, loss_fn, opt, capture=(mode == "captured"))
        step = lambda: ts(x, y)   # noqa: E731
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    return t / steps * 1e3, t_host / steps * 1e3


for name, (ch, b) in CFG.items():
    if opts.get("cfg", name) != name:
        continue
    for mode in ("autograd", "lean", "captured"):
        if opts.get("only", mode) != mode:
            continue
        ms, host = run(ch, b, mode)
        print(f"{name} batch {b} {mode:9s}: {ms:.3f} ms per step (host enqueue {host:.3f} ms)", flush=True)
