"""CPU restatement of the reference's F1@IoU0.5 harness (metric of record).

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Follows /root/reference/baseline/compute_metric.py:
  _label_overlap :39-70, _intersection_over_union :21-37, _true_positive :83-105,
  eval_tp_fp_fn :107-122, remove_boundary_cells :124-133, dice :72-81,
  precision/recall/F1 :234-241 (tp == 0 -> all zero); >= 25 Mpx patch walk :191-232.
PINNED: tests/golden/f1_*.npz hold tp/fp/fn produced by importing the
reference's compute_metric.py here (tests/golden/make_f1_golden.py).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment


def label_overlap_ref(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    x = x.ravel().astype(np.int64)
    y = y.ravel().astype(np.int64)
    nx, ny = 1 + int(x.max()), 1 + int(y.max())
    return np.bincount(x * ny + y, minlength=nx * ny).reshape(nx, ny).astype(np.uint64)


def iou_ref(masks_true: np.ndarray, masks_pred: np.ndarray) -> np.ndarray:
    overlap = label_overlap_ref(masks_true, masks_pred).astype(np.float64)
    n_pred = overlap.sum(axis=0, keepdims=True)
    n_true = overlap.sum(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = overlap / (n_pred + n_true - overlap)
    iou[np.isnan(iou)] = 0.0
    return iou


def true_positive_ref(iou: np.ndarray, th: float) -> int:
    n_min = min(iou.shape[0], iou.shape[1])
    costs = -(iou >= th).astype(float) - iou / (2 * n_min)
    ti, pi = linear_sum_assignment(costs)
    return int((iou[ti, pi] >= th).sum())


def eval_tp_fp_fn_ref(masks_true: np.ndarray, masks_pred: np.ndarray, threshold: float = 0.5):
    n_gt, n_seg = int(masks_true.max()), int(masks_pred.max())
    if n_seg > 0:
        iou = iou_ref(masks_true, masks_pred)[1:, 1:]
        tp = true_positive_ref(iou, threshold)
        return tp, n_seg - tp, n_gt - tp
    return 0, 0, 0


def relabel_sequential_ref(mask: np.ndarray) -> np.ndarray:
    """skimage.segmentation.relabel_sequential(mask)[0] for non-negative ints (order preserving)."""
    vals = np.unique(mask)
    vals = vals[vals > 0]
    lut = np.zeros(int(mask.max()) + 1, dtype=mask.dtype)
    lut[vals] = np.arange(1, len(vals) + 1, dtype=mask.dtype)
    return lut[mask]


def remove_boundary_cells_ref(mask: np.ndarray) -> np.ndarray:
    mask = mask.copy()
    h, w = mask.shape
    bd = np.ones((h, w), dtype=bool)
    bd[2:h - 2, 2:w - 2] = False
    ids = np.unique(mask[bd])
    ids = ids[ids > 0]
    mask[np.isin(mask, ids)] = 0
    return relabel_sequential_ref(mask)


def f1_ref(gt: np.ndarray, seg: np.ndarray, threshold: float = 0.5, count_bd_cells: bool = False):
    """Small-image branch of compute_metric.py main() (:183-190, :234-241)."""
    if not count_bd_cells:
        gt, seg = remove_boundary_cells_ref(gt.astype(np.int32)), remove_boundary_cells_ref(seg.astype(np.int32))
    gt, seg = relabel_sequential_ref(gt), relabel_sequential_ref(seg)
    tp, fp, fn = eval_tp_fp_fn_ref(gt, seg, threshold)
    if tp == 0:
        return tp, fp, fn, 0.0, 0.0, 0.0
    precision, recall = tp / (tp + fp), tp / (tp + fn)
    return tp, fp, fn, precision, recall, 2 * precision * recall / (precision + recall)


def f1_patch_counts_ref(gt: np.ndarray, seg: np.ndarray, edge: int = 2000, threshold: float = 0.5) -> np.ndarray:
    """Large-image branch of compute_metric.py main() (:191-232): zero-pad to a multiple of `edge`, then per patch drop the
    boundary cells, relabel, match; returns summed (tp, fp, fn, true cells, predicted cells).  Pinned by
    tests/golden/f1_patch_golden.npz (reference functions driven patch by patch)."""
    h, w = gt.shape
    ph, pw = -(-h // edge) * edge, -(-w // edge) * edge
    g = np.zeros((ph, pw), np.int32); g[:h, :w] = gt
    s = np.zeros((ph, pw), np.int32); s[:h, :w] = seg
    tot = np.zeros(5, np.int64)
    for y in range(0, ph, edge):
        for x in range(0, pw, edge):
            a = remove_boundary_cells_ref(g[y:y + edge, x:x + edge])
            b = remove_boundary_cells_ref(s[y:y + edge, x:x + edge])
            tp, fp, fn = eval_tp_fp_fn_ref(a, b, threshold)
            tot += np.array([tp, fp, fn, a.max(), b.max()], np.int64)
    return tot


def dice_ref(gt: np.ndarray, seg: np.ndarray) -> float:
    g, s = np.count_nonzero(gt), np.count_nonzero(seg)
    if g == 0 and s == 0:
        return 1.0
    if g == 0 and s > 0:
        return 0.0
    return 2 * np.count_nonzero(np.logical_and(gt, seg)) / (g + s)
