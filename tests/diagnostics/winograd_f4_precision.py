"""Diagnostic (CPU): would Winograd F(4x4, 3x3) - 2.25 instead of 4 multiplies per output pixel - keep the logits inside the 1e-4 budget?

Evaluates the U-Net with its 3x3 stride-1 convolutions as F(2x2,3x3) (what the GPU does today) or F(4x4,3x3) (Lavin's matrices) for
all / the wide layers, weights transformed in fp64 and rounded once (as the packing kernel does), everything else in fp32, against an
fp64 direct evaluation.  Result (round 3): F(4x4) on every layer with >= 128 input channels: 2.1e-5 ... 3.8e-5 (direct fp32 5e-6 ... 8e-6).

    python tests/diagnostics/winograd_f4_precision.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.unet_ref import make_reference_unet
BT4 = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=torch.float64)
G4 = torch.tensor([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype=torch.float64)
AT4 = torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
def wino(x, w, b, m):
    dt = x.dtype
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    t = m + 2
    n, c, h, wd = x.shape
    k = w.shape[0]
    u = torch.einsum("ai,kcij,bj->abkc", G, w.double(), G).to(dt)     # weights transformed in fp64, rounded once
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)
    bt, at = BT.to(dt), AT.to(dt)
    v = torch.einsum("ai,nchwij,bj->abnchw", bt, tiles, bt)
    mm = torch.einsum("abkc,abnchw->abnkhw", u, v)
    y = torch.einsum("ia,abnkhw,jb->nkhwij", at, mm, at)
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)
    return y + b.view(1, -1, 1, 1)
class Patch:
    def __init__(self, rule): self.orig = F.conv2d; self.rule = rule; self.count = {2:0,4:0}
    def __enter__(self):
        def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            s = stride if isinstance(stride, int) else stride[0]
            if w.shape[2:] == (3, 3) and s == 1 and w.shape[1] % 16 == 0 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0:
                m = self.rule(w.shape[1], w.shape[0])
                if m:
                    self.count[m] += 1
                    return wino(x, w, b, m)
            return self.orig(x, w, b, stride, padding, dilation, groups)
        F.conv2d = conv2d; torch.nn.functional.conv2d = conv2d
        return self
    def __exit__(self, *a): F.conv2d = self.orig; torch.nn.functional.conv2d = self.orig
for width in (32, 16):
    ch = tuple(width * 2 ** i for i in range(5))
    net = make_reference_unet(2022, channels=ch).eval()
    x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref64 = net.double()(x.double()); net = net.float(); d32 = net(x)
        print(f"width {width}: max|logit| {ref64.abs().max():.3f}; fp32 direct vs fp64 {(d32.double()-ref64).abs().max():.3e}")
        for name, rule in (("all F(2x2)", lambda ci, co: 2), ("F(4x4) where cin>=128 else F(2x2)", lambda ci, co: 4 if ci >= 128 else 2),
                           ("F(4x4) where cin>=64", lambda ci, co: 4 if ci >= 64 else 2), ("all F(4x4)", lambda ci, co: 4)):
            with Patch(rule) as p: w32 = net(x)
            print(f"   {name:38s} {p.count}: vs fp64 {(w32.double()-ref64).abs().max():.3e}   vs fp32 direct {(w32-d32).abs().max():.3e}")
        net = net.double()
        with Patch(lambda ci, co: 4): w64 = net(x.double())
        print(f"   algebra check fp64 F(4x4): {(w64-ref64).abs().max():.3e}")
