"""ctypes binding of libcellseg_hip.so (the C-ABI declared in include/cellseg_hip.h).

The library is the product: there is NO fallback.  A missing or unloadable library raises at import
time, and every call site raises ``CellSegHipError`` when the C side reports a failure.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  MUST precede the CDLL below: libcellseg_hip.so has to bind to the HIP runtime PyTorch already
#               loaded (same SONAME, libamdhip64.so.7) - streams and device pointers are shared with torch.  Loading our
#               library first pulls in /opt/rocm's copy, and the process then holds two HIP runtimes ("no ROCm-capable
#               device is detected" on the first hipMalloc).

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcellseg_hip.so")


class CellSegHipError(RuntimeError):
    pass


class SwiStrip(C.Structure):
    """Mirror of ``cs_swi_strip`` (window-row partition of one image over ranks)."""
    _fields_ = [(n, C.c_int) for n in ("win_row_begin", "win_row_end", "halo_begin", "row_begin", "row_end")]


class SwiShare(C.Structure):
    """Mirror of ``cs_swi_share`` (window-list partition of one image over ranks)."""
    _fields_ = [("win_begin", C.c_int), ("win_end", C.c_int), ("slot_row_begin", C.c_int), ("slot_row_end", C.c_int), ("n_rects", C.c_int),
                ("rect", (C.c_int * 4) * 3)]

    def rects(self):
        return [tuple(self.rect[i]) for i in range(self.n_rects)]


class SwiPlan(C.Structure):
    """Mirror of ``cs_swi_plan``."""
    _fields_ = [(n, C.c_int) for n in ("img_h", "img_w", "roi_h", "roi_w", "pad_top", "pad_left", "pad_h", "pad_w",
                                       "interval_h", "interval_w", "n_h", "n_w")]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C neurips-cellseg_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, sz, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double
    sig = {
        "cs_last_error": (C.c_char_p, []),
        "cs_version": (i32, []),
        "cs_device_count": (i32, []),
        "cs_unet_create": (i32, [C.POINTER(vp), i32, i32, C.POINTER(i32), i32, i32, i32, i32]),
        "cs_unet_destroy": (i32, [vp]),
        "cs_unet_set_precision": (i32, [vp, i32]),
        "cs_unet_set_training": (i32, [vp, i32]),
        "cs_unet_set_batch_stats": (i32, [vp, i32]),
        "cs_unet_read_param": (i32, [vp, C.c_char_p, vp, i64, vp]),
        "cs_unet_load_begin": (i32, [vp]),
        "cs_unet_load_end": (i32, [vp, vp]),
        "cs_unet_set_streams": (i32, [vp, i32]),
        "cs_unet_trace": (i32, [vp, i32]),
        "cs_unet_trace_count": (i32, [vp]),
        "cs_unet_trace_entry": (i32, [vp, i32, C.c_char_p, C.POINTER(i64)]),
        "cs_unet_num_params": (i32, [vp]),
        "cs_unet_param_info": (i32, [vp, i32, C.c_char_p, C.POINTER(i64)]),
        "cs_unet_load_param": (i32, [vp, C.c_char_p, vp, i64, vp]),
        "cs_unet_workspace_bytes": (sz, [vp, i32, i32, i32]),
        "cs_unet_forward": (i32, [vp, vp, i64, i64, i64, i64, i32, i32, i32, vp, vp, sz, vp]),
        "cs_unet_grad_numel": (i64, [vp]),
        "cs_unet_grad_offset": (i64, [vp, i32]),
        "cs_unet_backward_workspace_bytes": (sz, [vp, i32, i32, i32]),
        "cs_unet_backward": (i32, [vp, vp, vp, sz, vp, vp]),
        "cs_swi_make_plan": (i32, [i32, i32, i32, i32, f64, C.POINTER(SwiPlan)]),
        "cs_swi_window_start": (i32, [i32, i32, i32, i32]),
        "cs_swi_workspace_bytes": (sz, [vp, C.POINTER(SwiPlan), i32]),
        "cs_swi_windows_workspace_bytes": (sz, [vp, C.POINTER(SwiPlan), i32]),
        "cs_swi_windows": (i32, [vp, C.POINTER(SwiPlan), vp, i64, i64, i64, i32, i32, i32, vp, vp, sz, vp]),
        "cs_swi_partition": (i32, [C.POINTER(SwiPlan), i32, i32, C.POINTER(SwiStrip)]),
        "cs_swi_partition_windows": (i32, [C.POINTER(SwiPlan), i32, i32, C.POINTER(SwiShare)]),
        "cs_swi_windows_range": (i32, [vp, C.POINTER(SwiPlan), vp, i64, i64, i64, i32, i32, i32, vp, vp, sz, vp]),
        "cs_swi_finalize_rect": (i32, [C.POINTER(SwiPlan), vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
        "cs_swi_infer": (i32, [vp, C.POINTER(SwiPlan), vp, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
        "cs_swi_finalize": (i32, [C.POINTER(SwiPlan), vp, i32, i32, i32, i32, vp, vp, vp, vp]),
        "cs_normalize_workspace_bytes": (sz, [i32, i32]),
        "cs_normalize_image": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
        "cs_label_max": (i32, [vp, vp, i64, vp, vp]),
        "cs_label_overlap": (i32, [vp, vp, i64, i32, i32, vp, vp]),
        "cs_remove_boundary_workspace_bytes": (sz, [i32]),
        "cs_remove_boundary_cells": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
        "cs_augment_workspace_bytes": (sz, [i32, i32, i32]),
        "cs_augment_batch": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]),
        "cs_label_deflate_workspace_bytes": (sz, [i32, i32, i32]),
        "cs_label_deflate_bound": (sz, [i32, i32, i32]),
        "cs_label_deflate": (i32, [vp, i32, i32, i32, vp, sz, vp, vp, sz, vp]),
        "cs_ccl_workspace_bytes": (sz, [i32, i32]),
        "cs_ccl_postprocess": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
        "cs_ccl_label": (i32, [vp, i32, i32, i32, vp, vp, vp, sz, vp]),
        "cs_ccl_strip_halo": (i32, [i32, i32]),
        "cs_ccl_strip_edge_ints": (i32, [i32]),
        "cs_ccl_seam_workspace_bytes": (sz, [i32, i32]),
        "cs_ccl_strip_local": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
        "cs_ccl_strip_resolve": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp, sz, vp]),
        "cs_ccl_strip_finish": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
        "cs_interior_map": (i32, [vp, i32, i32, i32, vp, vp, sz, vp]),
        "cs_dice_ce_workspace_bytes": (sz, [i32, i32]),
        "cs_dice_ce_forward": (i32, [vp, C.POINTER(i64), vp, C.POINTER(i64), i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float,
                                     vp, vp, sz, vp]),
        "cs_dice_ce_backward": (i32, [vp, C.POINTER(i64), vp, C.POINTER(i64), i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float,
                                      vp, vp, sz, vp, vp]),
        "cs_adamw_step": (i32, [vp, i32, i64, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]),
        "cs_unet_bind_params": (i32, [vp, C.POINTER(vp), i32, vp]),
        "cs_unet_repack": (i32, [vp, vp]),
        "cs_unet_readback": (i32, [vp, vp]),
        "cs_adamw_step_dev": (i32, [vp, i32, i64, vp, vp, vp, vp, vp, vp]),
        "cs_graph_begin": (i32, [vp]),
        "cs_graph_end": (i32, [vp, C.POINTER(vp)]),
        "cs_graph_abort": (i32, [vp]),
        "cs_graph_launch": (i32, [vp, vp]),
        "cs_graph_num_nodes": (i32, [vp]),
        "cs_graph_destroy": (i32, [vp]),
        "cs_prof_is_enabled": (i32, []),
        "cs_prof_enable": (i32, [i32]),
        "cs_prof_reset": (i32, []),
        "cs_prof_collect": (i32, [C.c_char_p, C.POINTER(f64), C.POINTER(i64), C.POINTER(f64), C.POINTER(f64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
EXPORTED = ("cs_last_error cs_version cs_device_count cs_unet_create cs_unet_destroy cs_unet_set_precision cs_unet_set_training cs_unet_set_batch_stats cs_unet_read_param cs_unet_set_streams cs_unet_trace cs_unet_trace_count cs_unet_trace_entry cs_unet_num_params "
            "cs_unet_param_info cs_unet_load_param cs_unet_load_begin cs_unet_load_end cs_unet_workspace_bytes cs_unet_forward cs_unet_grad_numel "
            "cs_unet_grad_offset cs_unet_backward_workspace_bytes cs_unet_backward cs_swi_make_plan "
            "cs_swi_window_start cs_swi_workspace_bytes cs_swi_windows_workspace_bytes cs_swi_windows cs_swi_partition cs_swi_partition_windows cs_swi_windows_range cs_swi_infer cs_swi_finalize cs_swi_finalize_rect cs_normalize_workspace_bytes "
            "cs_normalize_image cs_label_max cs_label_overlap cs_remove_boundary_workspace_bytes cs_remove_boundary_cells "
            "cs_augment_workspace_bytes cs_augment_batch cs_label_deflate_workspace_bytes cs_label_deflate_bound cs_label_deflate cs_ccl_workspace_bytes "
            "cs_ccl_postprocess cs_ccl_label cs_ccl_strip_halo cs_ccl_strip_edge_ints cs_ccl_seam_workspace_bytes cs_ccl_strip_local cs_ccl_strip_resolve cs_ccl_strip_finish cs_interior_map cs_dice_ce_workspace_bytes cs_dice_ce_forward cs_dice_ce_backward cs_adamw_step "
            "cs_unet_bind_params cs_unet_repack cs_unet_readback cs_adamw_step_dev cs_graph_begin cs_graph_end cs_graph_abort cs_graph_launch "
            "cs_graph_num_nodes cs_graph_destroy cs_prof_is_enabled cs_prof_enable cs_prof_reset cs_prof_collect").split()


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        raise CellSegHipError(f"{what or 'cellseg_hip'} failed ({rc}): {lib.cs_last_error().decode(errors='replace')}")
    return rc


def require_gpu(tensor=None) -> None:
    """Fail loudly when the hot path is asked to run without a HIP device (no CPU fallback exists)."""
    import torch
    if tensor is not None and tensor.device.type != "cuda":
        raise CellSegHipError(f"cellseg_amd runs on MI355X only: got a tensor on '{tensor.device}'. There is no CPU path.")
    if not torch.cuda.is_available():
        raise CellSegHipError("cellseg_amd needs a HIP device (torch.cuda.is_available() is False). There is no CPU path.")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def prof_collect():
    """-> {kernel_name: dict(ms, launches, flops, bytes)} accumulated since the last cs_prof_reset."""
    n = 48          # CS_PROF_MAX_KERNELS (include/cellseg_hip.h)
    names = C.create_string_buffer(48 * n)
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    la = (C.c_int64 * n)()
    k = check(lib.cs_prof_collect(names, ms, la, fl, by), "cs_prof_collect")
    out = {}
    for i in range(k):
        nm = names.raw[48 * i:48 * (i + 1)].split(b"\0", 1)[0].decode()
        if la[i]:
            out[nm] = dict(ms=ms[i], launches=la[i], flops=fl[i], bytes=by[i])
    return out
