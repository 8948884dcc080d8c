#!/usr/bin/env python3
"""Max |logit error| of every matrix-core arithmetic mode against the fp32 torch-CPU oracle (and an fp64 run of it).
TEST INFRASTRUCTURE (imports the oracle).   python tests/diagnostics/gpu_precision_errors.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca
from oracle.unet_ref import make_reference_unet

out = {}
for name, ch, shape in [("reference16", (16, 32, 64, 128, 256), (3, 3, 256, 256)), ("baseline32", (32, 64, 128, 256, 512), (2, 3, 256, 256))]:
    ref = make_reference_unet(2022, channels=ch, strides=(2, 2, 2, 2), num_res_units=2).eval()
    hip = ca.UNet(channels=ch, strides=(2, 2, 2, 2), num_res_units=2)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to("cuda:0").eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        want32 = ref(x)
        want64 = ref.double()(x.double()).float()
        res = {"oracle_fp32_vs_fp64": (want32 - want64).abs().max().item()}
        for prec in ("fp32", "fp16x3", "bf16x6", "bf16x3", "bf16"):
            hip.precision = prec
            got = hip(x.to("cuda:0")).cpu()
            res[prec] = {"vs_oracle_fp32": (got - want32).abs().max().item(), "vs_fp64": (got - want64).abs().max().item()}
    out[name] = res
print(json.dumps(out, indent=1))
