"""Host-side training data pipeline for the ``model_training_3class`` CLI (a "next" row).

Mirrors the structure of /root/reference/baseline/model_training_3class.py:94-188: ``data_path/{images,labels}``
with ``<stem>_label.png`` labels, seeded 90/10 split, whole-image min-max intensity scaling, pad/crop to
``input_size``, random axis flip, random 90-degree rotation, Gaussian noise (p 0.25, std 0.1) and gamma contrast
(p 0.25, gamma in (1,2)).  MONAI's RandGaussianSmooth / RandHistogramShift / RandZoom are not reproduced yet (they
are CPU-side augmentations in DataLoader workers, outside the accelerated path).
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


class CellDataset(Dataset):
    def __init__(self, files: List[Tuple[str, str]], input_size: int, train: bool, seed: int = 0):
        self.files, self.size, self.train = files, input_size, train
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
        s, rng = self.size, self.rng
        ph, pw = max(s - img.shape[1], 0), max(s - img.shape[2], 0)
        if ph or pw:                                                    # SpatialPad: symmetric
            pad = ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2))
            img, lab = np.pad(img, pad), np.pad(lab, pad)
        y0 = rng.integers(0, img.shape[1] - s + 1)
        x0 = rng.integers(0, img.shape[2] - s + 1)
        img, lab = img[:, y0:y0 + s, x0:x0 + s], lab[:, y0:y0 + s, x0:x0 + s]
        if rng.random() < 0.5:
            ax = 1 + int(rng.integers(0, 2))
            img, lab = np.flip(img, ax), np.flip(lab, ax)
        if rng.random() < 0.5:
            k = int(rng.integers(1, 4))
            img, lab = np.rot90(img, k, (1, 2)), np.rot90(lab, k, (1, 2))
        if rng.random() < 0.25:
            img = img + rng.normal(0.0, rng.uniform(0, 0.1), img.shape).astype(np.float32)
        if rng.random() < 0.25:
            lo, hi = img.min(), img.max()
            img = ((img - lo) / (hi - lo + 1e-7)) ** rng.uniform(1, 2) * (hi - lo) + lo
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
