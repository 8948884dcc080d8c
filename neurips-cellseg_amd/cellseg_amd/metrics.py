"""F1 @ IoU harness - drop-in for the functions of /root/reference/baseline/compute_metric.py (the metric of record).

The O(H*W) parts (label maxima, the pixel-pair overlap histogram ``_label_overlap`` :39-70 and
``remove_boundary_cells`` :124-133) run on the MI355X; the small matching problem (IoU matrix :21-37, Hungarian
assignment :83-105, F1 :234-241) stays on the host with numpy / scipy exactly as the reference does it.
"""
from __future__ import annotations

import ctypes as C

from typing import Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def _i32(t: torch.Tensor) -> torch.Tensor:
    _lib.require_gpu(t)
    return t.to(torch.int32).contiguous()


def label_overlap(masks_true: torch.Tensor, masks_pred: torch.Tensor) -> np.ndarray:
    """[max(true)+1, max(pred)+1] matrix of pixel overlaps (``_label_overlap``)."""
    x, y = _i32(masks_true).reshape(-1), _i32(masks_pred).reshape(-1)
    if x.numel() != y.numel():
        raise ValueError("label maps differ in size")
    with torch.cuda.device(x.device):
        mx = torch.zeros(2, dtype=torch.int32, device=x.device)
        check(lib.cs_label_max(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), C.c_void_p(mx.data_ptr()),
                               _lib.stream_ptr()), "cs_label_max")
        nx, ny = (int(v) for v in mx.tolist())
        ov = torch.empty((nx + 1, ny + 1), dtype=torch.int32, device=x.device)
        check(lib.cs_label_overlap(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), nx, ny,
                                   C.c_void_p(ov.data_ptr()), _lib.stream_ptr()), "cs_label_overlap")
    return ov.cpu().numpy().astype(np.uint64)


def intersection_over_union(masks_true: torch.Tensor, masks_pred: torch.Tensor) -> np.ndarray:
    overlap = label_overlap(masks_true, masks_pred).astype(np.float64)
    n_pred = overlap.sum(axis=0, keepdims=True)
    n_true = overlap.sum(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = overlap / (n_pred + n_true - overlap)
    iou[np.isnan(iou)] = 0.0
    return iou


def true_positive(iou: np.ndarray, th: float) -> int:
    from scipy.optimize import linear_sum_assignment
    n_min = min(iou.shape[0], iou.shape[1])
    costs = -(iou >= th).astype(float) - iou / (2 * n_min)
    ti, pi = linear_sum_assignment(costs)
    return int((iou[ti, pi] >= th).sum())


def eval_tp_fp_fn(masks_true: torch.Tensor, masks_pred: torch.Tensor, threshold: float = 0.5):
    iou_full = intersection_over_union(masks_true, masks_pred)
    n_gt, n_seg = iou_full.shape[0] - 1, iou_full.shape[1] - 1
    if n_seg > 0:
        tp = true_positive(iou_full[1:, 1:], threshold) if n_gt > 0 else 0
        return tp, n_seg - tp, n_gt - tp
    return 0, 0, 0


def remove_boundary_cells(mask: torch.Tensor, border: int = 2) -> torch.Tensor:
    """Drop every label that touches the `border`-pixel image frame, then relabel sequentially (order preserving)."""
    m = _i32(mask)
    h, w = m.shape
    out = torch.zeros_like(m)
    with torch.cuda.device(m.device):
        mx = torch.zeros(2, dtype=torch.int32, device=m.device)
        check(lib.cs_label_max(C.c_void_p(m.data_ptr()), C.c_void_p(), m.numel(), C.c_void_p(mx.data_ptr()), _lib.stream_ptr()),
              "cs_label_max")
        n_labels = int(mx[0].item())
        ws = torch.empty(lib.cs_remove_boundary_workspace_bytes(n_labels), dtype=torch.uint8, device=m.device)
        check(lib.cs_remove_boundary_cells(C.c_void_p(m.data_ptr()), h, w, int(border), n_labels, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(), C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr()),
              "cs_remove_boundary_cells")
    return out


PATCH_PIXELS = 25_000_000      # compute_metric.py:183 - images at or above this size are scored patch by patch
PATCH_EDGE = 2000              # compute_metric.py:195


def patch_counts(gt: torch.Tensor, seg: torch.Tensor, edge: int = PATCH_EDGE, threshold: float = 0.5,
                 count_bd_cells: bool = False) -> Tuple[int, int, int, int, int]:
    """The reference's large-image walk (compute_metric.py:191-232): the maps are zero-padded to a multiple of `edge` and
    every `edge` x `edge` patch is scored on its own - boundary cells dropped (2-px frame of the PATCH), relabelled,
    matched - and (tp, fp, fn, true cells, predicted cells) are summed over the patches."""
    gt, seg = _i32(gt), _i32(seg)
    h, w = gt.shape
    border = 0 if count_bd_cells else 2
    tot = [0, 0, 0, 0, 0]
    for y in range(0, h, edge):
        for x in range(0, w, edge):
            a = torch.zeros((edge, edge), dtype=torch.int32, device=gt.device)
            b = torch.zeros_like(a)
            hh, ww = min(edge, h - y), min(edge, w - x)
            a[:hh, :ww] = gt[y:y + hh, x:x + ww]
            b[:hh, :ww] = seg[y:y + hh, x:x + ww]
            a, b = remove_boundary_cells(a, border), remove_boundary_cells(b, border)
            tp, fp, fn = eval_tp_fp_fn(a, b, threshold)
            for k, v in enumerate((tp, fp, fn, int(a.max().item()), int(b.max().item()))):
                tot[k] += v
    return tuple(tot)


def cell_counts(gt: torch.Tensor, seg: torch.Tensor, threshold: float = 0.5, count_bd_cells: bool = False,
                patch_pixels: int = PATCH_PIXELS, patch_edge: int = PATCH_EDGE) -> Tuple[int, int, int, int, int]:
    """(tp, fp, fn, true cells, predicted cells) as compute_metric.py main() counts them for one image: the whole-image
    branch (:183-190) below `patch_pixels`, the patch walk (:191-232) at or above it."""
    if gt.numel() >= patch_pixels:
        return patch_counts(gt, seg, patch_edge, threshold, count_bd_cells)
    border = 0 if count_bd_cells else 2          # border 0: relabel_sequential only
    gt, seg = remove_boundary_cells(gt, border), remove_boundary_cells(seg, border)
    tp, fp, fn = eval_tp_fp_fn(gt, seg, threshold)
    return tp, fp, fn, tp + fn, tp + fp


def f1_from_counts(tp: int, n_true: int, n_pred: int) -> Tuple[float, float, float]:
    """precision = tp / predicted cells, recall = tp / true cells, all three 0 when tp == 0 (compute_metric.py:233-241)."""
    if tp == 0:
        return 0.0, 0.0, 0.0
    precision, recall = tp / n_pred, tp / n_true
    return precision, recall, 2 * precision * recall / (precision + recall)


def f1_score(gt: torch.Tensor, seg: torch.Tensor, threshold: float = 0.5, count_bd_cells: bool = False,
             patch_pixels: int = PATCH_PIXELS, patch_edge: int = PATCH_EDGE):
    """(tp, fp, fn, precision, recall, f1) of one image."""
    tp, fp, fn, n_true, n_pred = cell_counts(gt, seg, threshold, count_bd_cells, patch_pixels, patch_edge)
    return (tp, fp, fn) + f1_from_counts(tp, n_true, n_pred)
