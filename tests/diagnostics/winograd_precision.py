"""Diagnostic (CPU): how far do the logits move when every 3x3 stride-1 convolution of the U-Net is evaluated as
Winograd F(2x2, 3x3) in fp32 (what `conv_wino_kernel` does on the GPU) instead of as a direct fp32 convolution?

Compares, against an fp64 direct evaluation of the same network on the same input:
  * torch fp32 direct (the oracle the GPU parity tests use),
  * fp32 Winograd (transforms and the 16 point-wise GEMMs all in fp32).
Budget: north_star's 1e-4 max-abs on the logits.

    python tests/diagnostics/winograd_precision.py [--width 32] [--n 2]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.unet_ref import make_reference_unet  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv3x3(x, w, b):
    """x [N,C,H,W] (H, W even), w [K,C,3,3]; all arithmetic in x.dtype."""
    dt = x.dtype
    n, c, h, wd = x.shape
    k = w.shape[0]
    bt, g, at = BT.to(dt), G.to(dt), AT.to(dt)
    u = torch.einsum("ai,kcij,bj->abkc", g, w, g)                     # [4,4,K,C]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # [N,C,H/2,W/2,4,4]
    v = torch.einsum("ai,nchwij,bj->abnchw", bt, tiles, bt)            # [4,4,N,C,th,tw]
    m = torch.einsum("abkc,abnchw->abnkhw", u, v)
    y = torch.einsum("ia,abnkhw,jb->nkhwij", at, m, at)                # [N,K,th,tw,2,2]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)
    return y + b.view(1, -1, 1, 1)


class Patch:
    def __init__(self):
        self.orig = F.conv2d
        self.count = 0

    def __enter__(self):
        def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            s = stride if isinstance(stride, int) else stride[0]
            if w.shape[2:] == (3, 3) and s == 1 and w.shape[1] % 16 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                self.count += 1
                return winograd_conv3x3(x, w, b)
            return self.orig(x, w, b, stride, padding, dilation, groups)
        F.conv2d = conv2d
        torch.nn.functional.conv2d = conv2d
        return self

    def __exit__(self, *a):
        F.conv2d = self.orig
        torch.nn.functional.conv2d = self.orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    ch = tuple(args.width * 2 ** i for i in range(5))
    net = make_reference_unet(2022, channels=ch).eval()
    x = torch.rand(args.n, 3, args.size, args.size, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref64 = net.double()(x.double())
        net = net.float()
        d32 = net(x)
        with Patch() as p:
            w32 = net(x)
        print(f"channels {ch}: {p.count} convolutions routed through Winograd F(2x2,3x3)")
        net = net.double()
        with Patch():
            w64 = net(x.double())
    print(f"max|logit| {ref64.abs().max():.3f}")
    print(f"fp32 direct   vs fp64: {(d32.double() - ref64).abs().max():.3e}")
    print(f"fp32 winograd vs fp64: {(w32.double() - ref64).abs().max():.3e}")
    print(f"fp32 winograd vs fp32 direct: {(w32 - d32).abs().max():.3e}")
    print(f"fp64 winograd vs fp64 direct (algebra check): {(w64 - ref64).abs().max():.3e}")


if __name__ == "__main__":
    main()
