#!/usr/bin/env python3
"""configs[4] accuracy at its own width (benchlib.accuracy.lowprec_accuracy: 64...1024 net trained on synthetic cells, 512^2 windows): the change
of the instance F1 against the synthetic ground truth in the bf16 and fp8 modes, per held-out image, for several TRAINING SEEDS -> mean and
standard error of delta F1 per seed and pooled (|mean| + 2 s.e. <= 1e-3).  TEST INFRASTRUCTURE.

    python tests/diagnostics/gpu_configs4_accuracy_n.py [n_images=48] [seeds=7,8,9] [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
from benchlib.accuracy import lowprec_accuracy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "7,8,9").split(",")]
rows, pooled = [], {"bf16": [], "fp8": []}
for seed in seeds:
    r = lowprec_accuracy(torch.device("cuda:0"), n_images=n, size=2048, precisions=("bf16", "fp8"), width="deep64", roi=512, seed=seed, per_image=True)
    row = {"train_seed": seed, "held_out_images": n, "cells": r["trained_net"]["cells_found_fp32"], "f1_fp32": r["trained_net"]["f1_vs_truth_fp32"],
           "loss_first_last": [r["trained_net"]["loss_first"], r["trained_net"]["loss_last"]]}
    for p in ("bf16", "fp8"):
        pooled[p] += r[p].pop("delta_f1_per_image")
        row[p] = r[p]
    rows.append(row)
    print(json.dumps(row), flush=True)
summary = {}
for p, d in pooled.items():
    t = torch.tensor(d, dtype=torch.float64)
    summary[p] = {"n": len(d), "delta_f1_mean": round(float(t.mean()), 6), "delta_f1_se": round(float(t.std(unbiased=True) / len(d) ** 0.5), 6),
                  "abs_mean_plus_2se": round(abs(float(t.mean())) + 2 * float(t.std(unbiased=True) / len(d) ** 0.5), 6)}
print(json.dumps({"pooled": summary}), flush=True)
if len(sys.argv) > 3:
    json.dump({"rows": rows, "pooled": summary}, open(sys.argv[3], "w"), indent=1)
