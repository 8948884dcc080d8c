#!/usr/bin/env python3
"""bf16 forward of N tiles (3 x 256 x 256) for a few N: ms per forward (the streaming kernels' launch-size thresholds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ca.UNet(channels=(32, 64, 128, 256, 512), strides=(2, 2, 2, 2), num_res_units=2).to(dev).eval()
m.precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
out = []
with torch.no_grad():
    for n in (8, 16, 32, 64, 128, 240, 481):
        x = torch.rand(n, 3, 256, 256, device=dev)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 20
        for _ in range(k):
            m(x)
        torch.cuda.synchronize()
        out.append(f"N={n}: {(time.perf_counter() - t0) / k * 1e3:.3f}")
print(" ".join(out))
