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


def write_label_tiff(path: str, labels: np.ndarray, rows_per_strip: int = 64, level: int = 6, threads: int = 0) -> None:
    """int32 single-page TIFF with deflate compression (`tif.imwrite(..., compression='zlib')`, predict.py:114).

    Own baseline-TIFF writer: the image is cut into strips of `rows_per_strip` rows which are deflated in parallel (zlib
    releases the GIL), because the single-threaded deflate of a 4096^2 label map (170 ms through Pillow) is 3x the whole
    GPU path.  Layout: header, strip data, then the IFD (little endian; tags 256-259, 262, 273, 277-279, 339)."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    a = np.ascontiguousarray(labels, dtype="<i4")
    if a.ndim != 2:
        raise ValueError("write_label_tiff expects an [H,W] label map")
    h, w = a.shape
    rps = max(1, min(int(rows_per_strip), h)) if h else 1
    bounds = [(y, min(y + rps, h)) for y in range(0, h, rps)] or [(0, 0)]
    compress = lambda yy: zlib.compress(a[yy[0]:yy[1]].tobytes(), level)      # noqa: E731
    n_thr = threads or min(16, len(bounds), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    if n_thr > 1:
        with ThreadPoolExecutor(n_thr) as ex:
            strips = list(ex.map(compress, bounds))
    else:
        strips = [compress(b) for b in bounds]
    n = len(strips)
    offsets, pos = [], 8
    for st in strips:
        offsets.append(pos)
        pos += len(st) + (len(st) & 1)                                # word-align every strip
    arrays_at = pos                                                    # StripOffsets, then StripByteCounts (when n > 1)
    ifd_at = arrays_at + (8 * n if n > 1 else 0)
    SHORT, LONG = 3, 4
    val = lambda v, t: struct.pack("<HH" if t == SHORT else "<I", *((v, 0) if t == SHORT else (v,)))   # noqa: E731
    entries = [(256, LONG, 1, val(w, LONG)), (257, LONG, 1, val(h, LONG)), (258, SHORT, 1, val(32, SHORT)),
               (259, SHORT, 1, val(8, SHORT)), (262, SHORT, 1, val(1, SHORT)),
               (273, LONG, n, val(offsets[0], LONG) if n == 1 else struct.pack("<I", arrays_at)),
               (277, SHORT, 1, val(1, SHORT)), (278, LONG, 1, val(rps, LONG)),
               (279, LONG, n, val(len(strips[0]), LONG) if n == 1 else struct.pack("<I", arrays_at + 4 * n)),
               (339, SHORT, 1, val(2, SHORT))]
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_at))
        for st in strips:
            f.write(st)
            if len(st) & 1:
                f.write(b"\0")
        if n > 1:
            f.write(struct.pack(f"<{n}I", *offsets))
            f.write(struct.pack(f"<{n}I", *[len(st) for st in strips]))
        f.write(struct.pack("<H", len(entries)))
        for tag, typ, cnt, v in entries:
            f.write(struct.pack("<HHI", tag, typ, cnt) + v)
        f.write(struct.pack("<I", 0))


def label_name(img_name: str) -> str:
    return img_name.split(".")[0] + "_label.tiff"          # predict.py:114


def list_images(input_path: str):
    return sorted(os.listdir(input_path))


def normalize_image_gpu(img, return_uint8: bool = False):
    """Device version of ``to_unit_float(prepare_image(img))`` for uint8 / uint16 images (bit-exact with the host path).

    img: torch tensor [H,W] or [H,W,C] (uint8, or uint16 bits held in an int16/uint16 tensor) on the GPU
    -> float32 [H,W,3] in [0,1]."""
    import ctypes as C

    import torch

    from . import _lib
    from ._lib import check, lib
    _lib.require_gpu(img)
    if img.dim() == 2:
        img = img[:, :, None]
    if img.dtype == torch.uint8:
        bps = 1
    elif img.dtype in (torch.uint16, torch.int16):
        bps = 2
    else:
        raise ValueError(f"normalize_image_gpu: dtype {img.dtype} is not supported (uint8 / uint16 images)")
    img = img.contiguous()
    h, w, c = img.shape
    out = torch.empty((h, w, 3), dtype=torch.float32, device=img.device)
    u8 = torch.empty((h, w, 3), dtype=torch.uint8, device=img.device) if return_uint8 else None
    with torch.cuda.device(img.device):
        ws = torch.empty(lib.cs_normalize_workspace_bytes(h, w), dtype=torch.uint8, device=img.device)
        check(lib.cs_normalize_image(C.c_void_p(img.data_ptr()), bps, h, w, c, C.c_void_p(out.data_ptr()),
                                     C.c_void_p(u8.data_ptr()) if u8 is not None else C.c_void_p(), C.c_void_p(ws.data_ptr()),
                                     ws.numel(), _lib.stream_ptr()), "cs_normalize_image")
    return (out, u8) if return_uint8 else out
