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


# ----------------------------------------------------------------------------------------------------------------------
# One slide over several GPUs: strip-local labelling with a seam merge (third row, first option)
# ----------------------------------------------------------------------------------------------------------------------
class StripLabeller:
    """predict.py:113 for ONE strip of rows of a slide (one rank per strip); the labels are bit-identical to ``mask_to_instances`` on
    the whole mask.  Three device steps with an all-gather between them (``parallel.label_strips`` drives them over a process group):

        rec  = s.local(mask)                  # filters on strip + halo, local forest, edge record           int32 [edge_ints]
        ids  = s.resolve(all_recs)            # seam union-find over every strip's record, final numbering  int32 [2 * W]
        lab  = s.finish(all_ids)              # int32 [rows, W]

    mask: the thresholded slide [H, W] (uint8 / bool, on the GPU) - only rows [row_begin - halo, row_end + halo) are read, so a caller
    that holds just those rows can pass ``mask_rows=`` (those rows) instead."""

    def __init__(self, img_h: int, img_w: int, rank: int, world: int, device, hole_area: int = 64, min_size: int = 16,
                 rows: Optional[tuple] = None):
        from .inferers import rank_rows
        self.h, self.w, self.rank, self.world = int(img_h), int(img_w), int(rank), int(world)
        self.hole_area, self.min_size = int(hole_area), int(min_size)
        self.row_begin, self.row_end = rows if rows is not None else rank_rows(self.h, self.rank, self.world)
        halo = lib.cs_ccl_strip_halo(self.hole_area, self.min_size)
        self.ext_begin, self.ext_end = max(0, self.row_begin - halo), min(self.h, self.row_end + halo)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CellSegHipError(f"cellseg_amd runs on MI355X only: StripLabeller on '{self.device}'. There is no CPU path.")
        self.edge_ints = lib.cs_ccl_strip_edge_ints(self.w)
        self._ws = torch.empty(max(int(lib.cs_ccl_workspace_bytes(self.ext_end - self.ext_begin, self.w)), 256), dtype=torch.uint8,
                               device=self.device)
        self._seam = torch.empty(max(int(lib.cs_ccl_seam_workspace_bytes(self.world, self.w)), 256), dtype=torch.uint8, device=self.device)
        self.n_labels = torch.zeros((), dtype=torch.int32, device=self.device)

    @property
    def strip_h(self) -> int:
        return self.row_end - self.row_begin

    def local(self, mask: Optional[torch.Tensor] = None, mask_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask_rows is None:
            mask_rows = _as_mask(mask)[self.ext_begin:self.ext_end]
        mask_rows = _as_mask(mask_rows)
        if tuple(mask_rows.shape) != (self.ext_end - self.ext_begin, self.w):
            raise ValueError(f"StripLabeller.local: rows {tuple(mask_rows.shape)}, expected {(self.ext_end - self.ext_begin, self.w)}")
        rec = torch.empty(self.edge_ints, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.cs_ccl_strip_local(C.c_void_p(mask_rows.data_ptr()), self.ext_end - self.ext_begin, self.w, self.row_begin - self.ext_begin,
                                         self.strip_h, self.hole_area, self.min_size, C.c_void_p(rec.data_ptr()),
                                         C.c_void_p(self._ws.data_ptr()), self._ws.numel(), _lib.stream_ptr()), "cs_ccl_strip_local")
        return rec

    def resolve(self, all_recs: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(all_recs)
        all_recs = all_recs.to(torch.int32).contiguous()
        if all_recs.numel() != self.world * self.edge_ints:
            raise ValueError(f"StripLabeller.resolve: {all_recs.numel()} ints, expected {self.world} x {self.edge_ints}")
        ids = torch.empty(2 * self.w, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.cs_ccl_strip_resolve(C.c_void_p(all_recs.data_ptr()), self.world, self.rank, self.ext_end - self.ext_begin, self.w,
                                           self.strip_h, C.c_void_p(ids.data_ptr()), C.c_void_p(self.n_labels.data_ptr()),
                                           C.c_void_p(self._ws.data_ptr()), self._ws.numel(), C.c_void_p(self._seam.data_ptr()),
                                           self._seam.numel(), _lib.stream_ptr()), "cs_ccl_strip_resolve")
        return ids

    def finish(self, all_ids: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        _lib.require_gpu(all_ids)
        all_ids = all_ids.to(torch.int32).contiguous()
        if all_ids.numel() != self.world * 2 * self.w:
            raise ValueError(f"StripLabeller.finish: {all_ids.numel()} ints, expected {self.world} x {2 * self.w}")
        if out is None:
            out = torch.empty((self.strip_h, self.w), dtype=torch.int32, device=self.device)
        if tuple(out.shape) != (self.strip_h, self.w) or out.dtype != torch.int32 or not out.is_contiguous():
            raise ValueError("StripLabeller.finish: out must be a contiguous int32 [rows, W] tensor")
        with torch.cuda.device(self.device):
            check(lib.cs_ccl_strip_finish(C.c_void_p(all_ids.data_ptr()), self.world, self.rank, self.ext_end - self.ext_begin, self.w,
                                          self.strip_h, C.c_void_p(out.data_ptr()), C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                          _lib.stream_ptr()), "cs_ccl_strip_finish")
        return out


def mask_to_instances_strips(mask: torch.Tensor, world: int, hole_area: int = 64, min_size: int = 16, return_num: bool = False,
                             rows: Optional[list] = None):
    """The strip-local labelling with every strip of `world` evaluated in THIS process (the all-gathers are concatenations): what the
    ranks of ``parallel.label_strips`` compute, for tests and for timing one GPU's share.  -> int32 [H, W] (bit-identical to
    ``mask_to_instances``)."""
    mask = _as_mask(mask)
    h, w = mask.shape
    strips = [StripLabeller(h, w, r, world, mask.device, hole_area, min_size, rows=None if rows is None else rows[r]) for r in range(world)]
    recs = torch.stack([s.local(mask) for s in strips])
    ids = torch.stack([s.resolve(recs) for s in strips])
    out = torch.cat([s.finish(ids) for s in strips], 0) if h * w else torch.zeros((h, w), dtype=torch.int32, device=mask.device)
    return (out, strips[0].n_labels) if return_num else out
