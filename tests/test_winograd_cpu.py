"""CPU checks of the Winograd F(2x2, 3x3) restatement the GPU kernel (csrc/kernels_conv_wino.hip) is built on: the transform
matrices are the standard ones (Lavin & Gray), the algebra is exact in fp64, and in fp32 the form costs no accuracy against the
direct convolution - the premise for running the 3x3 stride-1 layers this way inside the 1e-4 logit budget."""
import importlib.util
import os

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("winograd_precision", os.path.join(ROOT, "tests", "diagnostics", "winograd_precision.py"))
wp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(wp)


def test_winograd_algebra_is_exact_in_fp64():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 12, 20, generator=g, dtype=torch.float64)
    w = torch.randn(32, 16, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(32, generator=g, dtype=torch.float64)
    assert (wp.winograd_conv3x3(x, w, b) - F.conv2d(x, w, b, padding=1)).abs().max().item() < 1e-12


def test_winograd_fp32_is_as_accurate_as_direct_fp32():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 32, 32, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    e_direct = (F.conv2d(x, w, b, padding=1).double() - ref).abs().max().item()
    e_wino = (wp.winograd_conv3x3(x, w, b).double() - ref).abs().max().item()
    assert e_wino <= 4 * e_direct + 1e-6, (e_wino, e_direct)


def test_weight_transform_matrix_g():
    """U = G g G^T as the packing kernel computes it (pack_jobs.h: wino_weight_transform): rows (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)."""
    gk = torch.arange(9, dtype=torch.float64).reshape(3, 3) + 1
    u = wp.G @ gk @ wp.G.T
    col = torch.stack([gk[0], 0.5 * (gk[0] + gk[1] + gk[2]), 0.5 * (gk[0] - gk[1] + gk[2]), gk[2]])
    want = torch.stack([col[:, 0], 0.5 * (col[:, 0] + col[:, 1] + col[:, 2]), 0.5 * (col[:, 0] - col[:, 1] + col[:, 2]), col[:, 2]], 1)
    assert torch.equal(u, want)
