"""F1 @ IoU harness - drop-in for the functions of /root/reference/baseline/compute_metric.py (the metric of record).

The O(H*W) parts (label maxima, the pixel-pair overlap histogram ``_label_overlap`` :39-70 and
``remove_boundary_cells`` :124-133) run on the MI355X; the small matching problem (IoU matrix :21-37, Hungarian
assignment :83-105, F1 :234-241) stays on the host with numpy / scipy exactly as the reference does it.
"""
from __future__ import annotations

import ctypes as C

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


def f1_score(gt: torch.Tensor, seg: torch.Tensor, threshold: float = 0.5, count_bd_cells: bool = False):
    """(tp, fp, fn, precision, recall, f1) of the reference's small-image branch (compute_metric.py:183-190, :234-241)."""
    if not count_bd_cells:
        gt, seg = remove_boundary_cells(gt), remove_boundary_cells(seg)
    else:      # relabel_sequential only
        gt, seg = remove_boundary_cells(gt, 0), remove_boundary_cells(seg, 0)
    tp, fp, fn = eval_tp_fp_fn(gt, seg, threshold)
    if tp == 0:
        return tp, fp, fn, 0.0, 0.0, 0.0
    precision, recall = tp / (tp + fp), tp / (tp + fn)
    return tp, fp, fn, precision, recall, 2 * precision * recall / (precision + recall)
