"""Pure-PyTorch CPU restatement of the U-Net the reference instantiates.

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Follows the call sites
  /root/reference/baseline/model_training_3class.py:201-208
  /root/reference/baseline/predict.py:45-52
i.e. ``monai.networks.nets.UNet(spatial_dims=2, in_channels=3, out_channels=3,
channels=(16,32,64,128,256), strides=(2,2,2,2), num_res_units=2)``.

MONAI (``monai>=0.9``, requirements.txt:1, unpinned) is a third-party
dependency that is absent from /root/reference and not installable here, so
this file restates MONAI's published topology (SURVEY.md section 8(a) row M and
Appendix E items 1-3).  PARITY UNPINNED against MONAI itself; structural
anchors checked by tests/test_oracle_unet.py: 1 626 072 parameters,
63 state-dict tensors, MONAI state-dict key names, output shape.

Topology (all assumptions live in this one file):
  Convolution  = conv -> ADN("NDA") = InstanceNorm2d(affine=False, eps=1e-5)
                 -> Dropout(0) -> PReLU(1 param, 0.25);  ``conv_only`` drops ADN
  ResidualUnit = conv-path(x) + residual(x); residual is a strided 3x3 conv if
                 stride != 1, a 1x1 conv if only channels differ, else identity
  up layer     = Sequential(Convolution(transposed k3 s2 p1 op1),
                            ResidualUnit(subunits=1, last_conv_only=is_top))
  block        = Sequential(down, SkipConnection(sub) [cat([x, sub(x)], 1)], up)
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn


def _norm(kind: str, ch: int) -> nn.Module:
    if kind == "instance":
        return nn.InstanceNorm2d(ch, eps=1e-5, affine=False, track_running_stats=False)
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    raise ValueError(kind)


def _act(kind: str) -> nn.Module:
    if kind == "prelu":
        return nn.PReLU(num_parameters=1, init=0.25)
    if kind == "relu":
        return nn.ReLU()
    raise ValueError(kind)


class ADN(nn.Sequential):
    """norm -> dropout(0) -> act, children named N / D / A (ordering "NDA")."""

    def __init__(self, ch: int, norm: str, act: str):
        super().__init__()
        self.add_module("N", _norm(norm, ch))
        self.add_module("D", nn.Dropout(0.0))
        self.add_module("A", _act(act))


class Convolution(nn.Sequential):
    def __init__(self, cin: int, cout: int, stride: int, kernel: int = 3, *,
                 transposed: bool = False, conv_only: bool = False,
                 norm: str = "instance", act: str = "prelu"):
        super().__init__()
        pad = (kernel - 1) // 2
        if transposed:
            conv = nn.ConvTranspose2d(cin, cout, kernel, stride, pad,
                                      output_padding=stride - 1, bias=True)
        else:
            conv = nn.Conv2d(cin, cout, kernel, stride, pad, bias=True)
        self.add_module("conv", conv)
        if not conv_only:
            self.add_module("adn", ADN(cout, norm, act))


class ResidualUnit(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int, subunits: int, *,
                 last_conv_only: bool = False, norm: str = "instance", act: str = "prelu"):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual: nn.Module = nn.Identity()
        sc, ss = cin, stride
        for su in range(max(1, subunits)):
            only = last_conv_only and su == subunits - 1
            self.conv.add_module(f"unit{su:d}",
                                 Convolution(sc, cout, ss, 3, conv_only=only, norm=norm, act=act))
            sc, ss = cout, 1
        if stride != 1 or cin != cout:
            if stride != 1:
                self.residual = nn.Conv2d(cin, cout, 3, stride, 1, bias=True)
            else:
                self.residual = nn.Conv2d(cin, cout, 1, 1, 0, bias=True)

    def forward(self, x):
        return self.conv(x) + self.residual(x)


class SkipConnection(nn.Module):
    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.submodule = submodule

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], dim=1)


class UNetRef(nn.Module):
    """State-dict-compatible restatement of ``monai.networks.nets.UNet`` (2-D)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 channels: Sequence[int] = (16, 32, 64, 128, 256),
                 strides: Sequence[int] = (2, 2, 2, 2), num_res_units: int = 2,
                 norm: str = "instance", act: str = "prelu"):
        super().__init__()
        assert len(strides) == len(channels) - 1 and num_res_units > 0
        self.channels, self.strides, self.nru = tuple(channels), tuple(strides), num_res_units
        self.norm, self.act = norm, act
        self.model = self._block(in_channels, out_channels, self.channels, self.strides, True)

    def _down(self, cin, cout, stride):
        return ResidualUnit(cin, cout, stride, self.nru, norm=self.norm, act=self.act)

    def _up(self, cin, cout, stride, is_top):
        conv = Convolution(cin, cout, stride, 3, transposed=True, conv_only=False,
                           norm=self.norm, act=self.act)
        ru = ResidualUnit(cout, cout, 1, 1, last_conv_only=is_top, norm=self.norm, act=self.act)
        return nn.Sequential(conv, ru)

    def _block(self, inc, outc, channels, strides, is_top):
        c, s = channels[0], strides[0]
        if len(channels) > 2:
            sub = self._block(c, c, channels[1:], strides[1:], False)
            upc = c * 2
        else:
            sub = self._down(c, channels[1], 1)      # bottom layer
            upc = c + channels[1]
        return nn.Sequential(self._down(inc, c, s), SkipConnection(sub), self._up(upc, outc, s, is_top))

    def forward(self, x):
        return self.model(x)


def make_reference_unet(seed: int = 2022, **kw) -> UNetRef:
    """Default-initialised net under a fixed seed (SURVEY.md section 8(d) synthetic weights)."""
    torch.manual_seed(seed)
    return UNetRef(**kw).eval()
