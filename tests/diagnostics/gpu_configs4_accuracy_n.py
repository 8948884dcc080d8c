#!/usr/bin/env python3
"""configs[4] accuracy at its own width (bench.lowprec_accuracy, deep64 / 512^2 windows) for several held-out set sizes: how much of
delta F1 is the sample (TEST INFRASTRUCTURE; the function trains the net anew for every call, same seed)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
from bench import lowprec_accuracy
for n in (4, 8, 16):
    r = lowprec_accuracy(torch.device("cuda:0"), n_images=n, size=2048, precisions=("bf16", "fp8"), width="deep64", roi=512)
    print(json.dumps({"held_out_images": n, "cells": r["trained_net"]["cells_found_fp32"], "f1_fp32": r["trained_net"]["f1_vs_truth_fp32"],
                      "bf16": r["bf16"], "fp8": r["fp8"]}), flush=True)
