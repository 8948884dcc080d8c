"""CPU restatement of the reference's input normalisation (row N).

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Follows /root/reference/baseline/predict.py:14-21 (``normalize_channel``) and
:94-108 (channel fix-up, per-channel normalisation, float64 ``/ np.max``, cast
to float32, NCHW).  ``exposure.rescale_intensity(..., out_range='uint8')`` is
restated from scikit-image 0.18.3 ``exposure/exposure.py``: clip to
[p_lo, p_hi] in float64, scale to [0, 255], truncate to uint8.
PINNED: tests/golden/normalize_*.npz come from the reference's own function
(imported from data/pre_process_3class.py:22-29, identical text to
predict.py:14-21) under /opt/conda python3.9 + scikit-image 0.18.3.
"""
from __future__ import annotations

import numpy as np


def normalize_channel_ref(img: np.ndarray, lower: float = 1, upper: float = 99) -> np.ndarray:
    nz = img[np.nonzero(img)]
    p = np.percentile(nz, [lower, upper])
    if p[1] - p[0] > 0.001:
        lo, hi = float(p[0]), float(p[1])
        x = np.clip(img, lo, hi)                       # float64 for integer inputs
        x = (x - lo) / (hi - lo)
        return np.asarray(x * 255.0 + 0.0, dtype=np.uint8)   # truncation
    return img.astype(np.uint8)                        # reference quirk: wraps for uint16


def preprocess_image_ref(img_data: np.ndarray) -> np.ndarray:
    """Any [H,W] / [H,W,C] image -> uint8 [H,W,3] exactly as predict.py:94-104."""
    if img_data.ndim == 2:
        img_data = np.repeat(np.expand_dims(img_data, axis=-1), 3, axis=-1)
    elif img_data.ndim == 3 and img_data.shape[-1] > 3:
        img_data = img_data[:, :, :3]
    pre = np.zeros(img_data.shape, dtype=np.uint8)
    for i in range(3):
        ch = img_data[:, :, i]
        if len(ch[np.nonzero(ch)]) > 0:
            pre[:, :, i] = normalize_channel_ref(ch, 1, 99)
    return pre


def to_network_input_ref(pre: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3] -> float32 [1,3,H,W] in [0,1] (predict.py:107-108; division in float64)."""
    x01 = pre / np.max(pre)
    return np.ascontiguousarray(np.expand_dims(x01, 0).transpose(0, 3, 1, 2)).astype(np.float32)
