#!/usr/bin/env python3
"""Diagnostic: per-tensor agreement of the training gradients of the 16-bit modes with the fp32 engine's (cosine, relative error).
    python tests/diagnostics/gpu_grad_modes.py [channels=32,64,128,256,512] [batch=4] [size=128]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch  # noqa: E402

import cellseg_amd as ca  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
ch = tuple(int(v) for v in opts.get("channels", "32,64,128,256,512").split(","))
b, size = int(opts.get("batch", 4)), int(opts.get("size", 128))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ca.UNet(channels=ch, strides=(2,) * (len(ch) - 1), num_res_units=2).to(dev).train()
x = torch.rand(b, 3, size, size, device=dev)
y = ca.one_hot(torch.randint(0, 3, (b, 1, size, size), device=dev), 3)
loss_fn = ca.DiceCELoss(softmax=True)
grads = {}
for prec in ("fp32", "fp16x3", "bf16"):
    model.precision = prec
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()
    grads[prec] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
for prec in ("fp16x3", "bf16"):
    worst = []
    for k, g32 in grads["fp32"].items():
        g = grads[prec][k]
        den = g32.norm().item()
        rel = (g - g32).norm().item() / max(den, 1e-30)
        cos = torch.nn.functional.cosine_similarity(g.flatten(), g32.flatten(), dim=0).item() if den > 0 else 1.0
        worst.append((rel, cos, k, den))
    worst.sort(reverse=True)
    print(prec, "worst relative errors (rel, cosine, tensor, |g32|):")
    for w in worst[:8]:
        print("   %.3e  cos %.6f  %s  %.3e" % w)
