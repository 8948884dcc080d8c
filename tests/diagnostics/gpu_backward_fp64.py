#!/usr/bin/env python3
"""Diagnostic: where does the engine's backward differ from an fp64 autograd oracle, and is it the PReLU kink?

    python tests/diagnostics/gpu_backward_fp64.py [--batch 8] [--size 256] [--out profiles/backward_fp64_r02.json]

For every parameter gradient: |hip - fp64| and |torch-fp32 - fp64| (max-abs), once with the default PReLU slopes and once
with every slope forced to 1.0 (no kink: dz = da whatever the sign of z).  Also counts, on the CPU nets, how many PReLU
inputs have a different sign in fp32 and fp64 (the elements a kink can flip)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]


def run(channels, shape, slope, seed=0):
    import cellseg_amd as ca
    from oracle.unet_ref import UNetRef
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    kw = dict(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=2)
    ref = UNetRef(**kw).train()
    if slope is not None:
        for n, p in ref.named_parameters():
            if n.endswith("A.weight"):
                p.data.fill_(slope)
    ref64 = UNetRef(**kw).double().train()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    hip = ca.UNet(**kw)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev).train()
    g = torch.Generator().manual_seed(seed + 1)
    x, wgt = torch.rand(*shape, generator=g), torch.randn(*shape, generator=g)
    # sign flips at PReLU inputs (fp32 vs fp64 CPU nets)
    z32, z64 = {}, {}

    def hook(store):
        def f(mod, inp):
            store.setdefault(id(mod), []).append(inp[0].detach())
        return f
    names = {}
    for (n32, m32), (_, m64) in zip(ref.named_modules(), ref64.named_modules()):
        if isinstance(m32, torch.nn.PReLU):
            m32.register_forward_pre_hook(hook(z32))
            m64.register_forward_pre_hook(hook(z64))
            names[id(m32)] = (n32, id(m64))
    (ref64(x.double()) * wgt.double()).sum().backward()
    (ref(x) * wgt).sum().backward()
    (hip(x.to(dev)) * wgt.to(dev)).sum().backward()
    flips, elems = 0, 0
    for i32, (n, i64) in names.items():
        a, b = z32[i32][0], z64[i64][0]
        flips += int(((a >= 0) != (b >= 0)).sum())
        elems += a.numel()
    g64 = {k: p.grad for k, p in ref64.named_parameters()}
    g32 = {k: p.grad.double() for k, p in ref.named_parameters()}
    rows = []
    for k, p in hip.named_parameters():
        gh = p.grad.cpu().double()
        eh, et, m = (gh - g64[k]).abs().max().item(), (g32[k] - g64[k]).abs().max().item(), g64[k].abs().max().item()
        # how concentrated is the engine's error: elements above 10x the tensor's median error
        d = (gh - g64[k]).abs().flatten()
        med = d.median().item()
        rows.append(dict(key=k, hip_err=eh, torch32_err=et, max_abs=m, hip_rel=eh / max(m, 1e-30), torch32_rel=et / max(m, 1e-30),
                         n=d.numel(), outliers=int((d > 10 * max(med, 1e-30)).sum()) if d.numel() > 16 else 0))
    return dict(slope=slope, prelu_inputs=elems, sign_flips_fp32_vs_fp64=flips, tensors=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ch, shape = (16, 32, 64, 128, 256), (args.batch, 3, args.size, args.size)
    out = {"shape": list(shape), "channels": list(ch), "runs": [run(ch, shape, None), run(ch, shape, 1.0)]}
    for r in out["runs"]:
        scale = max(t["max_abs"] for t in r["tensors"])
        rows = sorted(r["tensors"], key=lambda t: -t["hip_err"] / max(t["max_abs"], 1e-6 * scale))
        print(f"slope={r['slope']}: PReLU inputs {r['prelu_inputs']}, fp32-vs-fp64 sign flips {r['sign_flips_fp32_vs_fp64']}")
        big = [t for t in rows if t["max_abs"] > 1e-6 * scale]
        print(f"  net scale {scale:.3e}; worst hip_rel {max(t['hip_rel'] for t in big):.2e}, worst torch32_rel {max(t['torch32_rel'] for t in big):.2e}")
        for t in rows[:8]:
            print(f"   {t['key']:<62} hip {t['hip_err']:.3e} t32 {t['torch32_err']:.3e} max {t['max_abs']:.3e} rel {t['hip_rel']:.1e}/{t['torch32_rel']:.1e} outliers {t['outliers']}/{t['n']}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
