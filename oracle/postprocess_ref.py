"""CPU restatement of the reference's instance post-process chain (rows S, P1-P4).

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Call site followed: /root/reference/baseline/predict.py:110-113
    softmax(dim=1)[0,1] -> ``> 0.5`` -> morphology.remove_small_holes (defaults
    area_threshold=64, connectivity=1) -> morphology.remove_small_objects(., 16)
    (connectivity=1) -> measure.label (connectivity = ndim = 2 -> 8-connected)
scikit-image is third-party (requirements.txt:3, unpinned); the algorithm is
restated from scikit-image 0.18.3 (``morphology/misc.py:51-139`` and
``remove_small_holes`` below it, ``measure/_label.py:5-29,116-120``) ->
``scipy.ndimage.label``.  PINNED: tests/golden/postprocess_*.npz were produced
by real scikit-image 0.18.3 (tests/golden/make_postprocess_golden.py, run under
/opt/conda/bin/python3.9) and both implementations below match them bit for
bit, including the raster-order numbering of component ids.

Two implementations:
  * ``postprocess_scipy``  - numpy + scipy.ndimage.label
  * ``postprocess_c``      - plain C two-pass union-find (oracle/ccl_ref.c),
                             no scipy; also the bench's CPU "port" baseline.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FOUR = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
_EIGHT = np.ones((3, 3), dtype=bool)


def label_scipy(mask: np.ndarray, connectivity: int) -> np.ndarray:
    """connectivity 1 -> 4-neighbour, 2 -> 8-neighbour; int32 ids in raster order."""
    from scipy import ndimage as ndi
    out = np.empty(mask.shape, dtype=np.int32)
    ndi.label(mask, structure=_FOUR if connectivity == 1 else _EIGHT, output=out)
    return out


def remove_small_objects_ref(mask: np.ndarray, min_size: int, connectivity: int = 1) -> np.ndarray:
    """skimage.morphology.remove_small_objects for bool input (misc.py:117-137): strict ``<``."""
    if min_size == 0:
        return mask.copy()
    cc = label_scipy(mask, connectivity)
    sizes = np.bincount(cc.ravel())
    too_small = sizes < min_size
    out = mask.copy()
    out[too_small[cc]] = False
    return out


def remove_small_holes_ref(mask: np.ndarray, area_threshold: int = 64, connectivity: int = 1) -> np.ndarray:
    """skimage.morphology.remove_small_holes: small objects of the inverse, re-inverted."""
    inv = np.logical_not(mask)
    inv = remove_small_objects_ref(inv, area_threshold, connectivity)
    return np.logical_not(inv)


def postprocess_scipy(prob: np.ndarray, hole_area: int = 64, min_size: int = 16) -> np.ndarray:
    """prob [H,W] float -> int32 instance labels (predict.py:113)."""
    m = prob > 0.5                                   # strict
    m = remove_small_holes_ref(m, hole_area, 1)
    m = remove_small_objects_ref(m, min_size, 1)
    return label_scipy(m, 2)


def mask_postprocess_scipy(mask: np.ndarray, hole_area: int = 64, min_size: int = 16) -> np.ndarray:
    m = remove_small_holes_ref(mask.astype(bool), hole_area, 1)
    m = remove_small_objects_ref(m, min_size, 1)
    return label_scipy(m, 2)


def softmax_interior_ref(logits: np.ndarray) -> np.ndarray:
    """logits [3,H,W] fp32 -> class-1 probability, as torch softmax(dim=1)[0,1] (predict.py:110-111)."""
    import torch
    return torch.softmax(torch.from_numpy(np.ascontiguousarray(logits))[None], dim=1)[0, 1].numpy()


# ----------------------------------------------------------------------------
# plain-C restatement (oracle/ccl_ref.c)
# ----------------------------------------------------------------------------
_LIB = None


def build_c(force: bool = False) -> str:
    """Compile oracle/ccl_ref.c -> oracle/_build/libccl_ref.so (gcc only)."""
    src = os.path.join(_HERE, "ccl_ref.c")
    outdir = os.path.join(_HERE, "_build")
    os.makedirs(outdir, exist_ok=True)
    so = os.path.join(outdir, "libccl_ref.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c())
        lib.ccl_ref_label.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.ccl_ref_label.restype = ctypes.c_int
        lib.ccl_ref_postprocess.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p]
        lib.ccl_ref_postprocess.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def label_c(mask: np.ndarray, connectivity: int) -> np.ndarray:
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(m.shape, dtype=np.int32)
    k = _lib().ccl_ref_label(m.ctypes.data, m.shape[0], m.shape[1], 4 if connectivity == 1 else 8, out.ctypes.data)
    assert k >= 0
    return out


def mask_postprocess_c(mask: np.ndarray, hole_area: int = 64, min_size: int = 16) -> np.ndarray:
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(m.shape, dtype=np.int32)
    k = _lib().ccl_ref_postprocess(m.ctypes.data, m.shape[0], m.shape[1], hole_area, min_size, out.ctypes.data)
    assert k >= 0
    return out


def postprocess_c(prob: np.ndarray, hole_area: int = 64, min_size: int = 16) -> np.ndarray:
    return mask_postprocess_c(prob > 0.5, hole_area, min_size)
