#!/usr/bin/env python3
"""Evaluate the engine (through the C-ABI) on stored nets and inputs; the parent process compares with the oracle.

    CS_WINO={0,1,2} python gpu_engine_eval.py job.pt out.pt

job.pt: {"nets": {name: {"channels": [...], "state": state_dict}}, "windows": {name: [N,3,H,W]}, "slides": {name: ([3,H,W], roi, overlap)}}
out.pt: {"<net>/<windows>": logits [N,3,H,W], "<net>/<slide>": {"logits": [3,H,W], "mask": uint8 [H,W]}}
The Winograd form of the fp32 3x3 layers is fixed when the library loads (CS_WINO), hence one process per form."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "neurips-cellseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import cellseg_amd as ca
    job = torch.load(sys.argv[1])
    dev = torch.device("cuda:0")
    out = {}
    with torch.no_grad():
        for nname, net in job["nets"].items():
            ch = tuple(net["channels"])
            hip = ca.UNet(channels=ch, strides=(2,) * (len(ch) - 1), num_res_units=2)
            hip.load_state_dict(net["state"])
            hip = hip.to(dev).eval()
            for wname, x in job.get("windows", {}).items():
                out[f"{nname}/{wname}"] = hip(x.to(dev)).cpu()
            for sname, (img, roi, overlap) in job.get("slides", {}).items():
                lg, _, mask = ca.swi_infer_image(hip, img.to(dev), (roi, roi), overlap, want_logits=True, want_mask=True)
                out[f"{nname}/{sname}"] = {"logits": lg.cpu(), "mask": mask.cpu()}
    torch.save(out, sys.argv[2])


if __name__ == "__main__":
    main()
