#!/usr/bin/env python3
"""Diagnostic: accuracy of the low-precision datapaths on a synthetic-TRAINED 32-base net (150 AdamW steps on seeded synthetic cells):
instance F1@0.5 against the fp32 path's instances, pixel agreement of the interior mask, and F1 against the synthetic ground truth.
    python tests/diagnostics/gpu_lowprec_accuracy.py [steps=150] [images=4]  -> JSON on stdout"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch  # noqa: E402

from bench import lowprec_accuracy  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
res = lowprec_accuracy(torch.device("cuda:0"), steps=int(opts.get("steps", 150)), n_images=int(opts.get("images", 4)),
                       precisions=tuple(opts.get("modes", "fp16x3,bf16x6,bf16,fp8").split(",")))
print(json.dumps(res, indent=1))
