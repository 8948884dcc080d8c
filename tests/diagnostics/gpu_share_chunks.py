#!/usr/bin/env python3
"""Diagnostic (GPU): how one rank's share of the 8-rank window-list partition (120 windows of the 4096^2 slide) is best dealt to the
engine's internal streams - ms per share for (engine_batch, streams) settings.  TEST INFRASTRUCTURE, not a bench line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca
from bench import synthetic_wsi, make_model, WIDTHS

dev = torch.device("cuda:0")
model = make_model(WIDTHS["baseline32"]).to(dev)
img = torch.from_numpy(synthetic_wsi(4096, 2022)).to(dev)
out = {}
with torch.no_grad():
    for streams in (1, 2, 3, 4):
        for eb in (None, 20, 30, 40, 60, 120):
            def step():
                ca.swi_infer_share(model, img, (256, 256), 0.5, 3, 8, engine_batch=eb, streams=streams, exchange=lambda *a: 0)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(16):
                step()
            torch.cuda.synchronize()
            out[f"streams{streams}_batch{eb}"] = round((time.perf_counter() - t0) / 16 * 1e3, 3)
print(json.dumps(out))
