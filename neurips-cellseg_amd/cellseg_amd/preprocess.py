"""Host-side image preparation and label writing for the ``predict`` CLI (rows N and W of SURVEY.md 8(a)).

Follows /root/reference/baseline/predict.py:14-21 (``normalize_channel``), :88-108 (reader, channel fix-up,
per-channel 1-99 percentile normalisation of the non-zero pixels, float64 ``/ max``) and :114 (int32 zlib TIFF named
``<stem>_label.tiff``).  These steps sit either side of the accelerated path and are outside the reference's own
inference timer except the TIFF write; they run on the host with numpy + Pillow (tifffile / scikit-image are not
required).  The on-device version of the percentile normalisation is a "next" row.
"""
from __future__ import annotations

import os

import numpy as np


def normalize_channel(img: np.ndarray, lower: float = 1, upper: float = 99) -> np.ndarray:
    non_zero = img[np.nonzero(img)]
    lo, hi = np.percentile(non_zero, [lower, upper])
    if hi - lo > 0.001:
        x = (np.clip(img, float(lo), float(hi)) - float(lo)) / (float(hi) - float(lo))
        return np.asarray(x * 255.0, dtype=np.uint8)          # skimage rescale_intensity(out_range='uint8'): truncation
    return img.astype(np.uint8)


def prepare_image(img_data: np.ndarray) -> np.ndarray:
    """[H,W] or [H,W,C] of any dtype -> uint8 [H,W,3]."""
    if img_data.ndim == 2:
        img_data = np.repeat(img_data[:, :, None], 3, axis=-1)
    elif img_data.ndim == 3 and img_data.shape[-1] > 3:
        img_data = img_data[:, :, :3]
    pre = np.zeros(img_data.shape, dtype=np.uint8)
    for i in range(3):
        ch = img_data[:, :, i]
        if np.count_nonzero(ch) > 0:
            pre[:, :, i] = normalize_channel(ch, 1, 99)
    return pre


def to_unit_float(pre: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3] -> float32 [H,W,3] in [0,1]; the division happens in float64 like the reference's."""
    return (pre / np.max(pre)).astype(np.float32)


def read_image(path: str) -> np.ndarray:
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        return np.array(im)


def write_label_tiff(path: str, labels: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(labels.astype(np.int32), mode="I").save(path, format="TIFF", compression="tiff_adobe_deflate")


def label_name(img_name: str) -> str:
    return img_name.split(".")[0] + "_label.tiff"          # predict.py:114


def list_images(input_path: str):
    return sorted(os.listdir(input_path))
