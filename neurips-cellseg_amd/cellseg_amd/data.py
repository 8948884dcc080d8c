"""Host-side training data pipeline for the ``model_training_3class`` CLI (a "next" row).

Mirrors the structure of /root/reference/baseline/model_training_3class.py:94-188: ``data_path/{images,labels}``
with ``<stem>_label.png`` labels, seeded 90/10 split, whole-image min-max intensity scaling, pad/crop to
``input_size``, random axis flip, random 90-degree rotation, Gaussian noise (p 0.25, std 0.1), gamma contrast
(p 0.25, gamma in (1,2)), Gaussian smoothing (p 0.25, sigma_x in (1,2)), histogram shift (p 0.25, 3 control points)
and zoom (p 0.15, 0.8-1.5, area / nearest, size kept).  They are CPU-side augmentations in DataLoader workers, outside
the accelerated path, written from MONAI's published behaviour (MONAI itself is absent).  Their deterministic kernels - zoom, histogram
shift, Gaussian smoothing - are pinned to the third-party primitives MONAI calls for them (``torch.nn.functional.interpolate``
area / nearest, ``np.interp``, a separable torch convolution with MONAI's erf taps): fixtures ``tests/golden/augment_golden.npz``,
generator ``tests/golden/make_augment_golden.py``.  The random streams differ from MONAI's by construction.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from . import preprocess

join = os.path.join


def split_files(data_path: str, seed: int, val_frac: float = 0.1):
    img_path, gt_path = join(data_path, "images"), join(data_path, "labels")
    img_names = sorted(os.listdir(img_path))
    gt_names = [n.split(".")[0] + "_label.png" for n in img_names]
    np.random.seed(seed)
    indices = np.arange(len(img_names))
    np.random.shuffle(indices)
    val_split = int(len(img_names) * val_frac)
    mk = lambda idx: [(join(img_path, img_names[i]), join(gt_path, gt_names[i])) for i in idx]   # noqa: E731
    return mk(indices[val_split:]), mk(indices[:val_split])


def _load(img_file: str, gt_file: str) -> Tuple[np.ndarray, np.ndarray]:
    img = preprocess.read_image(img_file).astype(np.uint8)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, -1)
    lab = preprocess.read_image(gt_file).astype(np.uint8)
    img = img[:, :, :3].transpose(2, 0, 1).astype(np.float32)
    lo, hi = float(img.min()), float(img.max())
    img = (img - lo) / (hi - lo) if hi > lo else img * 0.0          # ScaleIntensity: whole-image min-max
    return img, lab[None]


def gaussian_kernel_1d(sigma: float, truncated: float = 4.0) -> np.ndarray:
    """Pixel-integrated ("erf") Gaussian taps, tail = int(max(sigma * truncated, 0.5) + 0.5), not re-normalised."""
    from scipy.special import erf
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    x = np.arange(-tail, tail + 1, dtype=np.float64)
    t = 0.70710678 / abs(float(sigma))
    return np.clip(0.5 * (erf(t * (x + 0.5)) - erf(t * (x - 0.5))), 0.0, None).astype(np.float32)


def gaussian_smooth(img: np.ndarray, sigma_y: float, sigma_x: float) -> np.ndarray:
    """[C,H,W] separable smoothing with zero padding (RandGaussianSmoothd's deterministic part)."""
    from scipy.ndimage import correlate1d
    out = img.astype(np.float32)
    for axis, sg in ((1, sigma_y), (2, sigma_x)):
        if sg > 0:
            out = correlate1d(out, gaussian_kernel_1d(sg), axis=axis, mode="constant", cval=0.0)
    return out


def histogram_shift(img: np.ndarray, floating: np.ndarray) -> np.ndarray:
    """Piecewise-linear intensity remap through control points (RandHistogramShiftd's deterministic part): reference
    points are evenly spaced over [min, max]; `floating` holds their images as fractions of that range."""
    lo, hi = img.min(), img.max()
    ref = np.linspace(0, 1, len(floating))                     # MONAI scales both point sets the same way: p * (max - min) + min
    return np.interp(img, ref * (hi - lo) + lo, np.asarray(floating, np.float64) * (hi - lo) + lo).astype(np.float32)


def random_control_points(rng, n: int) -> np.ndarray:
    ref = np.linspace(0.0, 1.0, n)
    fl = ref.copy()
    for i in range(1, n - 1):
        fl[i] = rng.uniform(fl[i - 1], fl[i + 1])
    return fl


def zoom_keep_size(img: np.ndarray, lab: np.ndarray, z: float):
    """RandZoomd(mode=["area","nearest"], keep_size=True, padding_mode="edge") for one zoom factor: resample by z, then
    centre-crop (z > 1) or edge-pad (z < 1) back to the input size."""
    import torch.nn.functional as F
    h, w = img.shape[1:]
    oh, ow = int(np.floor(h * z)), int(np.floor(w * z))
    zi = F.interpolate(torch.from_numpy(np.ascontiguousarray(img))[None].float(), size=(oh, ow), mode="area")[0].numpy()
    zl = F.interpolate(torch.from_numpy(np.ascontiguousarray(lab))[None].float(), size=(oh, ow), mode="nearest")[0].numpy().astype(lab.dtype)

    def fit(a):
        for ax, (have, want) in ((1, (a.shape[1], h)), (2, (a.shape[2], w))):
            if have > want:
                o = (have - want) // 2
                a = np.take(a, np.arange(o, o + want), axis=ax)
            elif have < want:
                d = want - have
                pad = [(0, 0)] * 3
                pad[ax] = (d // 2, d - d // 2)
                a = np.pad(a, pad, mode="edge")
        return a
    return fit(zi).astype(np.float32), fit(zl)


def seed_worker(worker_id: int) -> None:
    """DataLoader ``worker_init_fn``: every worker process derives its own augmentation stream from the seed torch hands it
    (base_seed + worker_id, fresh per DataLoader iterator unless the workers persist) - forked workers would otherwise all
    start from the parent's generator state and draw identical crops / flips / noise, epoch after epoch."""
    info = torch.utils.data.get_worker_info()
    ds = info.dataset
    if hasattr(ds, "rng"):
        ds.rng = np.random.default_rng([int(getattr(ds, "seed", 0)), int(info.seed) & 0xFFFFFFFF, worker_id])


class CellDataset(Dataset):
    def __init__(self, files: List[Tuple[str, str]], input_size: int, train: bool, seed: int = 0):
        self.files, self.size, self.train, self.seed = files, input_size, train, seed
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        img, lab = _load(*self.files[i])
        if self.train:
            img, lab = self._augment(img, lab)
        return {"img": torch.from_numpy(np.ascontiguousarray(img)), "label": torch.from_numpy(np.ascontiguousarray(lab)),
                "name": os.path.basename(self.files[i][0])}

    def _augment(self, img, lab):
        return apply_params(img, lab, draw_params(self.rng, img.shape[1], img.shape[2], self.size), self.size)


def draw_params(rng, src_h: int, src_w: int, size: int) -> dict:
    """One sample's random decisions of the reference's transform chain (model_training_3class.py:119-162), in its order.  The same
    record drives the host pipeline (`apply_params`, DataLoader workers) and the device pipeline (`cellseg_amd.gpu_data`,
    csrc/kernels_augment.hip)."""
    s = size
    ph, pw = max(s - src_h, 0), max(s - src_w, 0)                      # SpatialPadd: symmetric
    p = {"src_h": src_h, "src_w": src_w, "pad_top": ph // 2, "pad_left": pw // 2, "pad_h": ph, "pad_w": pw}
    p["crop_y"] = int(rng.integers(0, src_h + ph - s + 1))
    p["crop_x"] = int(rng.integers(0, src_w + pw - s + 1))
    p["flip_axis"] = 1 + int(rng.integers(0, 2)) if rng.random() < 0.5 else 0
    p["rot_k"] = int(rng.integers(1, 4)) if rng.random() < 0.5 else 0
    if rng.random() < 0.25:                                            # RandGaussianNoised(prob 0.25, std 0.1): std ~ U(0, 0.1)
        p["noise_std"], p["noise_seed"] = float(rng.uniform(0, 0.1)), int(rng.integers(0, 2 ** 32))
    else:
        p["noise_std"], p["noise_seed"] = -1.0, 0
    p["gamma"] = float(rng.uniform(1, 2)) if rng.random() < 0.25 else 0.0
    if rng.random() < 0.25:                                            # RandGaussianSmoothd(sigma_x=(1,2)); sigma_y default (0.25,1.5)
        p["sigma_y"], p["sigma_x"] = float(rng.uniform(1.0, 2.0)), float(rng.uniform(0.25, 1.5))
    else:
        p["sigma_y"] = p["sigma_x"] = 0.0
    p["hist"] = random_control_points(rng, 3) if rng.random() < 0.25 else None        # RandHistogramShiftd(num_control_points=3)
    p["zoom"] = float(rng.uniform(0.8, 1.5)) if rng.random() < 0.15 else 0.0          # RandZoomd(0.8, 1.5)
    return p


def apply_params(img: np.ndarray, lab: np.ndarray, p: dict, size: int):
    """The host implementation of one parameter record (numpy / scipy / torch-CPU primitives)."""
    s = size
    if p["pad_h"] or p["pad_w"]:
        pad = ((0, 0), (p["pad_top"], p["pad_h"] - p["pad_top"]), (p["pad_left"], p["pad_w"] - p["pad_left"]))
        img, lab = np.pad(img, pad), np.pad(lab, pad)
    y0, x0 = p["crop_y"], p["crop_x"]
    img, lab = img[:, y0:y0 + s, x0:x0 + s], lab[:, y0:y0 + s, x0:x0 + s]
    if p["flip_axis"]:
        img, lab = np.flip(img, p["flip_axis"]), np.flip(lab, p["flip_axis"])
    if p["rot_k"]:
        img, lab = np.rot90(img, p["rot_k"], (1, 2)), np.rot90(lab, p["rot_k"], (1, 2))
    if p["noise_std"] >= 0:
        img = img + np.random.default_rng(p["noise_seed"]).normal(0.0, p["noise_std"], img.shape).astype(np.float32)
    if p["gamma"] > 0:
        lo, hi = img.min(), img.max()
        img = ((img - lo) / (hi - lo + 1e-7)) ** p["gamma"] * (hi - lo) + lo
    if p["sigma_y"] > 0 or p["sigma_x"] > 0:
        img = gaussian_smooth(img, p["sigma_y"], p["sigma_x"])
    if p["hist"] is not None:
        img = histogram_shift(img, p["hist"])
    if p["zoom"] > 0:
        img, lab = zoom_keep_size(np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(lab), p["zoom"])
    return img.astype(np.float32), lab


class SyntheticCells(Dataset):
    """Seeded blobs with the 3-class label contract (0 background / 1 interior / 2 boundary) for smoke runs."""

    def __init__(self, n: int, size: int, seed: int = 2022):
        self.n, self.size, self.seed = n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from scipy.ndimage import binary_erosion, gaussian_filter
        rng = np.random.default_rng(self.seed + i)
        g = gaussian_filter(rng.random((self.size, self.size), dtype=np.float32), 5)
        g = (g - g.min()) / (g.max() - g.min())
        fg = g > np.quantile(g, 0.6)
        inner = binary_erosion(fg, iterations=2)
        lab = np.zeros((self.size, self.size), np.uint8)
        lab[fg] = 2
        lab[inner] = 1
        img = np.stack([g, g, g]) + rng.normal(0, 0.03, (3, self.size, self.size)).astype(np.float32)
        return {"img": torch.from_numpy(np.clip(img, 0, 1).astype(np.float32)), "label": torch.from_numpy(lab[None]), "name": f"synthetic_{i}"}
