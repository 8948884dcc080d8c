"""CPU restatement of the instance -> 3-class label contract ("next" row).

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Follows /root/reference/data/pre_process_3class.py:31-56 (``create_interior_map``):
  boundary = find_boundaries(inst, mode='inner')  (4-neighbourhood max != min, foreground only)
  boundary = binary_dilation(boundary, disk(1))   (cross, zero border)
  interior = remove_small_objects(~boundary & inst>0, 16);  out: 1 interior, 2 boundary
PINNED by tests/golden/label_contract_golden.npz (outputs of the imported reference function).
"""
from __future__ import annotations

import numpy as np

from .postprocess_ref import remove_small_objects_ref


def _cross_reduce(a: np.ndarray, fn) -> np.ndarray:
    p = np.pad(a, 1, mode="edge")
    out = p[1:-1, 1:-1].copy()
    for sl in (p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]):
        out = fn(out, sl)
    return out


def create_interior_map_ref(inst: np.ndarray) -> np.ndarray:
    boundary = (_cross_reduce(inst, np.maximum) != _cross_reduce(inst, np.minimum)) & (inst != 0)
    p = np.pad(boundary, 1, mode="constant")
    boundary = p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:]
    interior = remove_small_objects_ref(np.logical_and(~boundary, inst > 0), 16, 1)
    out = np.zeros(inst.shape, dtype=np.uint8)
    out[interior] = 1
    out[boundary] = 2
    return out
