"""Structural anchors for the MONAI restatements (U-Net topology, sliding-window tiler).  CPU only.

MONAI itself is unavailable: these pin the restatement to the facts the reference's
call sites fix - parameter count, state-dict names, shapes, window counts, partition of unity."""
import numpy as np
import pytest
import torch

from oracle.sliding_window_ref import scan_starts, sliding_window_inference_ref, window_grid
from oracle.unet_ref import UNetRef, make_reference_unet


def test_unet_param_count_and_keys():
    m = make_reference_unet()
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 1_626_072
    assert len(sd) == 63
    assert sd["model.0.conv.unit0.conv.weight"].shape == (16, 3, 3, 3)
    assert sd["model.0.residual.weight"].shape == (16, 3, 3, 3)
    assert sd["model.1.submodule.1.submodule.1.submodule.1.submodule.residual.weight"].shape == (256, 128, 1, 1)
    assert sd["model.1.submodule.1.submodule.1.submodule.2.0.conv.weight"].shape == (384, 64, 3, 3)
    assert sd["model.2.0.conv.weight"].shape == (32, 3, 3, 3)
    assert sd["model.2.1.conv.unit0.conv.weight"].shape == (3, 3, 3, 3)
    assert "model.2.1.conv.unit0.adn.A.weight" not in sd            # top conv is conv-only
    assert sd["model.2.0.adn.A.weight"].shape == (1,)


def test_unet_forward_shape_and_fp64_noise():
    m = make_reference_unet()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = m(x)
        y64 = m.double()(x.double())
    assert y.shape == (2, 3, 64, 64)
    assert (y.double() - y64).abs().max() < 1e-4


@pytest.mark.parametrize("size,overlap,expect", [(4096, 0.25, 21), (4096, 0.5, 31), (1024, 0.25, 5),
                                                 (512, 0.25, 3), (256, 0.25, 1), (1000, 0.25, 5)])
def test_scan_counts(size, overlap, expect):
    s = scan_starts(size, 256, overlap)
    assert len(s) == expect and s[0] == 0 and s[-1] == size - 256
    assert all(b - a <= int(256 * (1 - overlap)) for a, b in zip(s, s[1:]))


def test_window_order_row_major():
    g = window_grid(512, 600, (256, 256), 0.25)
    assert g[:4] == [(0, 0), (0, 192), (0, 344), (192, 0)]
    assert scan_starts(1000, 256, 0.25) == [0, 192, 384, 576, 744]


@pytest.mark.parametrize("hw", [(200, 200), (256, 256), (300, 520), (600, 257)])
def test_tiler_identity_predictor_is_identity(hw):
    x = torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(3))
    y = sliding_window_inference_ref(x, (256, 256), 4, lambda t: t.clone())
    assert y.shape == x.shape and torch.allclose(y, x, atol=1e-6)


def test_tiler_small_image_pads_symmetrically():
    seen = []
    x = torch.ones(1, 1, 200, 201)
    sliding_window_inference_ref(x, (256, 256), 4, lambda t: (seen.append(t.clone()), t)[1])
    w = seen[0][0, 0]
    assert w.shape == (256, 256)
    assert w[:28].sum() == 0 and w[28:228, 27:228].sum() == 200 * 201 and w[228:].sum() == 0 and w[:, :27].sum() == 0
