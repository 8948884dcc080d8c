"""Instance post-process on the GPU - drop-in for the reference's
``measure.label(morphology.remove_small_objects(morphology.remove_small_holes(prob > 0.5), 16))``
(/root/reference/baseline/predict.py:113).  Bit-exact with scikit-image 0.18.3 / scipy.ndimage.label, including
the raster-order numbering of the ids.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check, lib

_WS = {}


def _workspace(h: int, w: int, device) -> torch.Tensor:
    nbytes = lib.cs_ccl_workspace_bytes(h, w)
    ws = _WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS[device] = ws
    return ws


def _as_mask(mask: torch.Tensor) -> torch.Tensor:
    _lib.require_gpu(mask)
    if mask.dim() != 2:
        raise ValueError("mask must be [H,W]")
    if mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    if mask.dtype != torch.uint8:
        raise ValueError("mask must be bool or uint8")
    return mask.contiguous()


def label(mask: torch.Tensor, connectivity: int = 2, return_num: bool = False):
    """``skimage.measure.label(mask, connectivity=...)`` for a 2-D binary mask: 1 -> 4-neighbours, 2 -> 8."""
    mask = _as_mask(mask)
    h, w = mask.shape
    out = torch.zeros((h, w), dtype=torch.int32, device=mask.device)
    num = torch.zeros((), dtype=torch.int32, device=mask.device)
    if h * w:
        with torch.cuda.device(mask.device):
            ws = _workspace(h, w, mask.device)
            check(lib.cs_ccl_label(C.c_void_p(mask.data_ptr()), h, w, 4 if connectivity == 1 else 8, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(num.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr()),
                  "cs_ccl_label")
    return (out, num) if return_num else out


def mask_to_instances(mask: torch.Tensor, hole_area: int = 64, min_size: int = 16, return_num: bool = False):
    """remove_small_holes(area_threshold=64) -> remove_small_objects(16) -> label (8-connected), int32 [H,W]."""
    mask = _as_mask(mask)
    h, w = mask.shape
    out = torch.zeros((h, w), dtype=torch.int32, device=mask.device)
    num = torch.zeros((), dtype=torch.int32, device=mask.device)
    if h * w:
        with torch.cuda.device(mask.device):
            ws = _workspace(h, w, mask.device)
            check(lib.cs_ccl_postprocess(C.c_void_p(mask.data_ptr()), h, w, int(hole_area), int(min_size),
                                         C.c_void_p(out.data_ptr()), C.c_void_p(num.data_ptr()), C.c_void_p(ws.data_ptr()),
                                         ws.numel(), _lib.stream_ptr()), "cs_ccl_postprocess")
    return (out, num) if return_num else out


def prob_to_instances(prob: torch.Tensor, hole_area: int = 64, min_size: int = 16, return_num: bool = False):
    """predict.py:113 on a class-1 probability map (strict ``> 0.5``)."""
    _lib.require_gpu(prob)
    return mask_to_instances(prob > 0.5, hole_area, min_size, return_num)


def create_interior_map(inst_map: torch.Tensor, min_size: int = 16) -> torch.Tensor:
    """Instance map [H,W] (any integer dtype, on the GPU) -> uint8 3-class map: 0 background, 1 interior, 2 boundary.

    Drop-in for ``create_interior_map`` (/root/reference/data/pre_process_3class.py:31-56): inner boundaries
    (``find_boundaries(mode='inner')``) dilated by ``disk(1)``, interior = the rest of the foreground with 4-connected
    components smaller than `min_size` removed."""
    _lib.require_gpu(inst_map)
    m = inst_map.to(torch.int32).contiguous()
    if m.dim() != 2:
        raise ValueError("create_interior_map expects an [H,W] instance map")
    h, w = m.shape
    out = torch.zeros((h, w), dtype=torch.uint8, device=m.device)
    with torch.cuda.device(m.device):
        ws = torch.empty(max(int(lib.cs_ccl_workspace_bytes(h, w)), 256), dtype=torch.uint8, device=m.device)
        check(lib.cs_interior_map(C.c_void_p(m.data_ptr()), h, w, int(min_size), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr()), "cs_interior_map")
    return out
