#!/usr/bin/env python3
"""Fixtures that pin the deterministic kernels of the training augmentations (cellseg_amd/data.py) to the third-party PRIMITIVES
MONAI's transforms call for them (MONAI itself is absent; torch and numpy are here):

  RandZoomd(mode=["area","nearest"], keep_size=True)  ->  torch.nn.functional.interpolate(x[None], scale_factor=z, mode=...,
        recompute_scale_factor=True) followed by MONAI's centre-crop / symmetric edge-pad back to the input size   [recalled]
  RandHistogramShiftd(num_control_points=3)           ->  np.interp(img, ref * (max - min) + min, floating * (max - min) + min)
  RandGaussianSmoothd(sigma_x=(1,2))                  ->  separable torch conv (zero padding) with the "erf" taps of
        monai.networks.layers.gaussian_1d(sigma, truncated=4.0): 0.5 (erf(t (x + .5)) - erf(t (x - .5))), t = 0.70710678 / sigma

Reference call site: /root/reference/baseline/model_training_3class.py:119-162.  Run with the main interpreter:
    python tests/golden/make_augment_golden.py        (writes tests/golden/augment_golden.npz)
Nothing here imports the product; the arrays are inputs and the primitives' outputs only.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))


def monai_zoom(img: np.ndarray, z: float, mode: str) -> np.ndarray:
    t = torch.from_numpy(img).float()
    zoomed = F.interpolate(t[None], scale_factor=[z, z], mode=mode, recompute_scale_factor=True)[0]
    pad_vec, slice_vec = [(0, 0)] * 3, [slice(None)] * 3
    for idx, (od, zd) in enumerate(zip(t.shape, zoomed.shape)):
        diff = od - zd
        half = abs(diff) // 2
        if diff > 0:
            pad_vec[idx] = (half, diff - half)
        elif diff < 0:
            slice_vec[idx] = slice(half, half + od)
    out = np.pad(zoomed.numpy(), pad_vec, mode="edge")
    return out[tuple(slice_vec)]


def monai_hist_shift(img: np.ndarray, floating: np.ndarray) -> np.ndarray:
    ref = np.linspace(0, 1, len(floating))
    lo, hi = img.min(), img.max()
    return np.interp(img, ref * (hi - lo) + lo, floating * (hi - lo) + lo)


def monai_gaussian_1d(sigma: float, truncated: float = 4.0) -> torch.Tensor:
    s = torch.as_tensor(sigma, dtype=torch.float)
    tail = int(max(float(s) * truncated, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float)
    t = 0.70710678 / torch.abs(s)
    return (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)


def monai_gaussian_smooth(img: np.ndarray, s0: float, s1: float) -> np.ndarray:
    """sigma s0 along the first spatial axis, s1 along the second; zero padding (separable_filtering's default)."""
    t = torch.from_numpy(img).float()[:, None]                    # [C,1,H,W]: every channel on its own
    k0, k1 = monai_gaussian_1d(s0), monai_gaussian_1d(s1)
    t = F.conv2d(t, k0.view(1, 1, -1, 1), padding=(len(k0) // 2, 0))
    t = F.conv2d(t, k1.view(1, 1, 1, -1), padding=(0, len(k1) // 2))
    return t[:, 0].numpy()


def main():
    rng = np.random.default_rng(2022)
    out = {}
    img = rng.random((3, 37, 52), dtype=np.float32)
    lab = rng.integers(0, 3, (1, 37, 52)).astype(np.uint8)
    out["zoom_img"], out["zoom_lab"] = img, lab
    zs = np.array([0.8, 0.93, 1.0, 1.17, 1.31, 1.5, 0.875, 1.4142], np.float64)
    out["zoom_factors"] = zs
    for i, z in enumerate(zs):
        out[f"zoom_area_{i}"] = monai_zoom(img, float(z), "area").astype(np.float32)
        out[f"zoom_nearest_{i}"] = monai_zoom(lab.astype(np.float32), float(z), "nearest").astype(np.uint8)
    himg = (rng.random((3, 40, 33), dtype=np.float32) * 0.7 + 0.1).astype(np.float32)
    out["hist_img"] = himg
    fls = np.array([[0.0, 0.5, 1.0], [0.0, 0.13, 1.0], [0.0, 0.91, 1.0], [0.0, 0.25, 0.5, 0.6, 1.0][:3]], np.float64)
    out["hist_floating"] = fls
    for i, fl in enumerate(fls):
        out[f"hist_out_{i}"] = monai_hist_shift(himg, fl).astype(np.float32)
    gimg = rng.random((3, 45, 38), dtype=np.float32)
    out["gauss_img"] = gimg
    sig = np.array([[1.0, 0.25], [2.0, 1.5], [1.37, 0.9], [1.9, 0.26]], np.float64)
    out["gauss_sigmas"] = sig
    for i, (s0, s1) in enumerate(sig):
        out[f"gauss_out_{i}"] = monai_gaussian_smooth(gimg, float(s0), float(s1)).astype(np.float32)
        out[f"gauss_taps_{i}"] = monai_gaussian_1d(float(s0)).numpy()
    # the same primitives on SQUARE inputs: what the device chain (cs_augment_batch works on size x size crops) is fed directly in
    # tests/test_gpu_round4.py - drawn AFTER everything above, so the earlier arrays keep their values
    simg = rng.random((3, 48, 48), dtype=np.float32)
    simg[:, :16] *= 0.3
    slab = rng.integers(0, 3, (1, 48, 48)).astype(np.uint8)
    out["sq_img"], out["sq_lab"] = simg, slab
    for i, z in enumerate(zs):
        out[f"sq_zoom_area_{i}"] = monai_zoom(simg, float(z), "area").astype(np.float32)
        out[f"sq_zoom_nearest_{i}"] = monai_zoom(slab.astype(np.float32), float(z), "nearest").astype(np.uint8)
    for i, fl in enumerate(fls):
        out[f"sq_hist_out_{i}"] = monai_hist_shift(simg, fl).astype(np.float32)
    for i, (s0, s1) in enumerate(sig):
        out[f"sq_gauss_out_{i}"] = monai_gaussian_smooth(simg, float(s0), float(s1)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "augment_golden.npz"), **out)
    print("wrote augment_golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
