"""``sliding_window_inference`` - drop-in for ``monai.inferers.sliding_window_inference`` as the reference uses it.

Reference call sites: /root/reference/baseline/predict.py:83-84,109 and
/root/reference/baseline/model_training_3class.py:289-293 (roi 256x256, sw_batch_size 4, MONAI defaults
overlap=0.25, mode="constant", padding_mode="constant", cval=0).

When ``predictor`` is a :class:`cellseg_amd.UNet` the whole call is one C-ABI entry (``cs_swi_infer``): window
gather fused into the first convolution, all windows batched through the HIP U-Net, overlap averaging done in
window-index order (bit-identical accumulation order to the reference's ``out[slice] += pred``).  For any other
predictor the windows are gathered with torch indexing, the predictor is called as MONAI would call it, and
the HIP finalize kernel (``cs_swi_finalize``) does the averaging.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import SwiPlan, SwiShare, SwiStrip, check, lib
from .unet import UNet

DEFAULT_ENGINE_BATCH = 256   # windows per engine call when the caller fixes the batch (strips leg, tests)
MAX_AUTO_BATCH = 512


def balanced_batch(n_windows: int, streams: int = 2, max_batch: int = MAX_AUTO_BATCH) -> int:
    """Engine batch of the automatic setting (``engine_batch=None``): the windows split into EQUAL chunks, one round per internal stream
    where that fits ``max_batch`` (961 windows, 2 streams -> 2 x 481 instead of 256 + 256 + 256 + 193 dealt 2 + 2: the streams finish
    together and every launch is twice as large - measured 35.2 -> 34.5 ms on the 4096^2 slide).  Results do not depend on the batch."""
    n, s = max(1, int(n_windows)), max(1, int(streams))
    if n < 16 * s:                                             # a handful of windows: one launch
        return n
    rounds = -(-n // (s * max_batch))
    return -(-n // (s * rounds))
DEFAULT_STREAMS = 2          # window chunks alternate between two internal streams (memory-bound kernels overlap MFMA-bound ones)


def make_plan(img_h: int, img_w: int, roi_size: Sequence[int], overlap: float = 0.25) -> SwiPlan:
    plan = SwiPlan()
    check(lib.cs_swi_make_plan(int(img_h), int(img_w), int(roi_size[0]), int(roi_size[1]), float(overlap), C.byref(plan)),
          "cs_swi_make_plan")
    return plan


def window_starts(plan: SwiPlan) -> Tuple[list, list]:
    hs = [lib.cs_swi_window_start(i, plan.interval_h, plan.pad_h, plan.roi_h) for i in range(plan.n_h)]
    ws = [lib.cs_swi_window_start(i, plan.interval_w, plan.pad_w, plan.roi_w) for i in range(plan.n_w)]
    return hs, ws


def rank_rows(img_h: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous strip of output rows owned by `rank` (multi-GPU tiled inference)."""
    base, rem = divmod(img_h, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def strip_partition(plan: SwiPlan, rank: int, world: int) -> SwiStrip:
    """Window-row partition of one image (``cs_swi_partition``): the window rows `rank` evaluates, the output rows it finalises
    and the earlier ranks' window rows (the halo) whose lower part still covers them.  Shares differ by at most one window row:
    31 rows over 8 ranks = 3,4,4,4,4,4,4,4 (larger shares last) - no window is evaluated twice (``rank_rows`` strips duplicate one per seam)."""
    st = SwiStrip()
    check(lib.cs_swi_partition(C.byref(plan), int(rank), int(world), C.byref(st)), "cs_swi_partition")
    return st


def share_partition(plan: SwiPlan, rank: int, world: int) -> SwiShare:
    """Window-LIST partition of one image (``cs_swi_partition_windows``): the row-major window list cut into contiguous shares that
    differ by at most one window (961 windows over 8 ranks = 120 / 121; whole window rows give 93...124), the image rectangles the
    rank finalises (the pixels whose LAST covering window is one of its own) and the window rows its buffer has to hold."""
    sh = SwiShare()
    check(lib.cs_swi_partition_windows(C.byref(plan), int(rank), int(world), C.byref(sh)), "cs_swi_partition_windows")
    return sh


WORKSPACE_BUDGET = 64 << 30     # bytes of HBM the tiler may take for window activations (of 288 GB)


def _held_bytes(model: UNet) -> int:
    """Workspace bytes the model already holds for the tiler (they are free for re-use by the next call)."""
    return sum(v.numel() for k, v in getattr(model, "_ws", {}).items() if k and k[0] in ("swi", "swistrip", "swiwin"))


def swi_infer_image(model: UNet, image: torch.Tensor, roi_size, overlap: float = 0.25, *, engine_batch: Optional[int] = None,
                    rows: Optional[Tuple[int, int]] = None, want_logits: bool = True, want_prob: bool = False,
                    want_mask: bool = False, streams: int = DEFAULT_STREAMS):
    """image [3,H,W] (any strides) on the GPU -> (logits [3,H,W] | None, prob1 [H,W] | None, mask uint8 [H,W] | None)."""
    _lib.require_gpu(image)
    if image.dim() != 3:
        raise ValueError("image must be [C,H,W]")
    if image.dtype != torch.float32:
        image = image.float()
    _, h, w = image.shape
    plan = make_plan(h, w, roi_size, overlap)
    eng = model.engine()
    dev = image.device
    with torch.cuda.device(dev):
        check(lib.cs_unet_set_streams(eng.handle, int(streams)), "cs_unet_set_streams")
        if not engine_batch:
            engine_batch = balanced_batch(plan.n_h * plan.n_w, streams)
        # the activation workspace scales with engine_batch x streams x window area x width: stay inside a memory budget
        # (wide nets on 512^2 windows would otherwise ask for > 100 GB at a batch of 256 windows)
        engine_batch, nbytes = _capped_batch(model, eng, plan, engine_batch, dev, lib.cs_swi_workspace_bytes)
        ws = model.workspace(("swi", h, w, plan.roi_h, plan.roi_w, plan.n_h, plan.n_w, int(engine_batch), int(streams)), nbytes, dev)
        logits = torch.empty((model.out_channels, h, w), dtype=torch.float32, device=dev) if want_logits else None
        prob = torch.empty((h, w), dtype=torch.float32, device=dev) if want_prob else None
        mask = torch.empty((h, w), dtype=torch.uint8, device=dev) if want_mask else None
        r0, r1 = rows if rows is not None else (0, h)
        sc, sh, sw = image.stride()
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()   # noqa: E731
        check(lib.cs_swi_infer(eng.handle, C.byref(plan), ptr(image), sc, sh, sw, int(engine_batch), int(r0), int(r1),
                               ptr(logits), ptr(prob), ptr(mask), ptr(ws), ws.numel(), _lib.stream_ptr()), "cs_swi_infer")
    return logits, prob, mask


def _capped_batch(model: UNet, eng, plan: SwiPlan, engine_batch: int, dev, size_fn) -> Tuple[int, int]:
    """Largest engine batch (halving from the request) whose workspace stays inside the memory budget -> (batch, bytes)."""
    nbytes = size_fn(eng.handle, C.byref(plan), int(engine_batch))
    budget = min(WORKSPACE_BUDGET, int(0.6 * torch.cuda.mem_get_info(dev)[0]) + _held_bytes(model))
    while nbytes > budget and engine_batch > 4:
        engine_batch = max(4, int(engine_batch) // 2)
        nbytes = size_fn(eng.handle, C.byref(plan), int(engine_batch))
    if nbytes == 0:
        raise _lib.CellSegHipError(f"swi workspace: {lib.cs_last_error().decode()}")
    return int(engine_batch), int(nbytes)


def swi_infer_strip(model: UNet, image: torch.Tensor, roi_size, overlap: float, rank: int, world: int, *,
                    engine_batch: Optional[int] = None, streams: int = DEFAULT_STREAMS, exchange=None,
                    want_logits: bool = False, want_prob: bool = False, want_mask: bool = True):
    """ONE image split over `world` ranks by WINDOW rows (multi-GPU tiled inference; the reference is single-device,
    baseline/predict.py:109).  This rank evaluates its own window rows (``cs_swi_windows``), obtains the lower part of the earlier
    ranks' window rows that still cover its first output rows (the halo) through ``exchange(win, plan, rank, world)`` -
    ``parallel.exchange_seam`` under torch.distributed; ``None`` evaluates the halo rows locally (single process) - and finalises
    its output rows (``cs_swi_finalize``).  Because the reference adds window predictions in window-index order, the halo rows are
    exactly the prefix of every seam pixel's sum: the result is bit-identical to the single-GPU one.
    -> (strip, logits | None, prob1 | None, mask | None); only rows [strip.row_begin, strip.row_end) of the outputs are written."""
    _lib.require_gpu(image)
    if image.dim() != 3:
        raise ValueError("image must be [C,H,W]")
    if image.dtype != torch.float32:
        image = image.float()
    _, h, w = image.shape
    plan = make_plan(h, w, roi_size, overlap)
    st = strip_partition(plan, rank, world)
    eng = model.engine()
    dev = image.device
    logits = torch.empty((model.out_channels, h, w), dtype=torch.float32, device=dev) if want_logits else None
    prob = torch.empty((h, w), dtype=torch.float32, device=dev) if want_prob else None
    mask = torch.empty((h, w), dtype=torch.uint8, device=dev) if want_mask else None
    if st.win_row_end == st.win_row_begin:
        if exchange is not None:        # fewer window rows than ranks: still enter the (possibly collective) seam exchange, see swi_infer_share
            exchange(torch.empty((0, plan.n_w, plan.roi_h, plan.roi_w, model.out_channels), dtype=torch.float32, device=dev), plan, rank, world)
        return st, logits, prob, mask
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()   # noqa: E731
    with torch.cuda.device(dev):
        check(lib.cs_unet_set_streams(eng.handle, int(streams)), "cs_unet_set_streams")
        if not engine_batch:                                   # this rank's own windows in equal chunks (the halo rows, evaluated locally without
            engine_batch = balanced_batch((st.win_row_end - st.win_row_begin) * plan.n_w, streams)     # an exchange, run at the same batch)
        engine_batch, nbytes = _capped_batch(model, eng, plan, engine_batch, dev, lib.cs_swi_windows_workspace_bytes)
        ws = model.workspace(("swistrip", h, w, plan.roi_h, plan.roi_w, plan.n_h, plan.n_w, engine_batch, int(streams)), nbytes, dev)
        n_rows = st.win_row_end - st.halo_begin
        win_bytes = n_rows * plan.n_w * plan.roi_h * plan.roi_w * model.out_channels * 4
        win = model.workspace(("swiwin", n_rows, plan.n_w, plan.roi_h, plan.roi_w), win_bytes, dev)[:win_bytes]
        win = win.view(torch.float32).view(n_rows, plan.n_w, plan.roi_h, plan.roi_w, model.out_channels)
        sc, sh, sw = image.stride()
        halo = st.win_row_begin - st.halo_begin

        def run(row_lo, row_hi):
            check(lib.cs_swi_windows(eng.handle, C.byref(plan), ptr(image), sc, sh, sw, engine_batch, int(row_lo), int(row_hi),
                                     ptr(win[row_lo - st.halo_begin]), ptr(ws), ws.numel(), _lib.stream_ptr()), "cs_swi_windows")

        run(st.win_row_begin, st.win_row_end)
        if exchange is not None:
            exchange(win, plan, rank, world)
        elif halo:
            run(st.halo_begin, st.win_row_begin)
        check(lib.cs_swi_finalize(C.byref(plan), ptr(win), model.out_channels, st.halo_begin, st.row_begin, st.row_end,
                                  ptr(logits), ptr(prob), ptr(mask), _lib.stream_ptr()), "cs_swi_finalize")
    return st, logits, prob, mask


def swi_infer_share(model: UNet, image: torch.Tensor, roi_size, overlap: float, rank: int, world: int, *,
                    engine_batch: Optional[int] = None, streams: int = DEFAULT_STREAMS, exchange=None,
                    want_logits: bool = False, want_prob: bool = False, want_mask: bool = True):
    """ONE image split over `world` ranks by the window LIST (``share_partition``; multi-GPU tiled inference; the
    reference is single-device, baseline/predict.py:109).  This rank evaluates its own windows (``cs_swi_windows_range``), obtains the
    parts of earlier ranks' windows that cover its rectangles through ``exchange(win, plan, rank, world)`` - ``parallel.exchange_share``
    under torch.distributed; ``None`` evaluates them locally (single process) - and finalises its rectangles (``cs_swi_finalize_rect``).
    Every window a rank lacks has a LOWER index than its own, and the reference adds window predictions in window-index order, so the
    received parts are exactly the prefix of each pixel's sum: bit-identical to the single-GPU result.
    -> (share, logits | None, prob1 | None, mask | None); only the share's rectangles of the outputs are written."""
    _lib.require_gpu(image)
    if image.dim() != 3:
        raise ValueError("image must be [C,H,W]")
    if image.dtype != torch.float32:
        image = image.float()
    _, h, w = image.shape
    plan = make_plan(h, w, roi_size, overlap)
    sh = share_partition(plan, rank, world)
    eng = model.engine()
    dev = image.device
    logits = torch.empty((model.out_channels, h, w), dtype=torch.float32, device=dev) if want_logits else None
    prob = torch.empty((h, w), dtype=torch.float32, device=dev) if want_prob else None
    mask = torch.empty((h, w), dtype=torch.uint8, device=dev) if want_mask else None
    if sh.win_end == sh.win_begin:
        # fewer windows than ranks: this rank holds nothing, but the seam exchange may be ONE collective every rank has to enter
        # (transport="allgather"), so it still takes part - with a zero-slot tensor - before it returns
        if exchange is not None:
            exchange(torch.empty((0, plan.roi_h, plan.roi_w, model.out_channels), dtype=torch.float32, device=dev), plan, rank, world)
        return sh, logits, prob, mask
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()   # noqa: E731
    with torch.cuda.device(dev):
        check(lib.cs_unet_set_streams(eng.handle, int(streams)), "cs_unet_set_streams")
        if not engine_batch:
            engine_batch = balanced_batch(sh.win_end - sh.win_begin, streams)
        engine_batch, nbytes = _capped_batch(model, eng, plan, engine_batch, dev, lib.cs_swi_windows_workspace_bytes)
        ws = model.workspace(("swistrip", h, w, plan.roi_h, plan.roi_w, plan.n_h, plan.n_w, engine_batch, int(streams)), nbytes, dev)
        base = sh.slot_row_begin * plan.n_w
        n_slots = (sh.slot_row_end - sh.slot_row_begin) * plan.n_w
        win_bytes = n_slots * plan.roi_h * plan.roi_w * model.out_channels * 4
        win = model.workspace(("swiwin", n_slots, plan.roi_h, plan.roi_w), win_bytes, dev)[:win_bytes]
        win = win.view(torch.float32).view(n_slots, plan.roi_h, plan.roi_w, model.out_channels)
        sc, sh_, sw = image.stride()

        def run(k_lo, k_hi):
            check(lib.cs_swi_windows_range(eng.handle, C.byref(plan), ptr(image), sc, sh_, sw, engine_batch, int(k_lo), int(k_hi),
                                           ptr(win[k_lo - base]), ptr(ws), ws.numel(), _lib.stream_ptr()), "cs_swi_windows_range")

        run(sh.win_begin, sh.win_end)
        if exchange is not None:
            exchange(win, plan, rank, world)
        elif sh.win_begin > base:
            run(base, sh.win_begin)                              # single process: the earlier windows of the held rows, evaluated here
        for r0, r1, c0, c1 in sh.rects():
            check(lib.cs_swi_finalize_rect(C.byref(plan), ptr(win), model.out_channels, sh.slot_row_begin, r0, r1, c0, c1,
                                           ptr(logits), ptr(prob), ptr(mask), _lib.stream_ptr()), "cs_swi_finalize_rect")
    return sh, logits, prob, mask


def sliding_window_inference(inputs: torch.Tensor, roi_size: Union[Sequence[int], int], sw_batch_size: int,
                             predictor: Callable[[torch.Tensor], torch.Tensor], overlap: float = 0.25,
                             mode: str = "constant", sigma_scale: float = 0.125, padding_mode: str = "constant",
                             cval: float = 0.0, sw_device=None, device=None, *args, **kwargs) -> torch.Tensor:
    """inputs [B,C,H,W] -> [B,Cout,H,W] fp32 logits (the reference always passes B == 1)."""
    if str(getattr(mode, "value", mode)).lower() != "constant" or str(getattr(padding_mode, "value", padding_mode)).lower() != "constant" or cval != 0.0:
        raise NotImplementedError("only the reference's settings are implemented: mode='constant', padding_mode='constant', cval=0")
    if args or kwargs:
        raise NotImplementedError("extra predictor arguments are not supported")
    _lib.require_gpu(inputs)
    if isinstance(roi_size, int):
        roi_size = (roi_size, roi_size)
    if inputs.dim() != 4:
        raise ValueError("inputs must be [B,C,H,W]")
    outs = []
    for b in range(inputs.shape[0]):
        if isinstance(predictor, UNet):
            # sw_batch_size only groups windows per predictor call in MONAI; results do not depend on it, so the
            # engine picks its own (balanced_batch) unless the caller asks for more than MAX_AUTO_BATCH windows per call.
            logits, _, _ = swi_infer_image(predictor, inputs[b], roi_size, overlap,
                                           engine_batch=int(sw_batch_size) if int(sw_batch_size) > MAX_AUTO_BATCH else None)
        else:
            logits = _generic(inputs[b], roi_size, int(sw_batch_size), predictor, overlap)
        outs.append(logits)
    return torch.stack(outs, 0)


def _generic(image: torch.Tensor, roi_size, sw_batch_size: int, predictor, overlap: float) -> torch.Tensor:
    """Arbitrary predictor: torch gathers the windows (plumbing), the HIP kernel finalizes."""
    image = image.float()
    c, h, w = image.shape
    plan = make_plan(h, w, roi_size, overlap)
    x = torch.zeros((c, plan.pad_h, plan.pad_w), dtype=torch.float32, device=image.device)
    x[:, plan.pad_top:plan.pad_top + h, plan.pad_left:plan.pad_left + w] = image
    hs, ws = window_starts(plan)
    wins = [(a, b) for a in hs for b in ws]
    preds = []
    for g in range(0, len(wins), sw_batch_size):
        batch = torch.stack([x[:, a:a + plan.roi_h, b:b + plan.roi_w] for a, b in wins[g:g + sw_batch_size]], 0)
        preds.append(predictor(batch).float())
    pred = torch.cat(preds, 0)                                     # [nwin, Cout, roi, roi]
    if pred.shape[1] != 3:
        raise NotImplementedError("the finalize kernel implements the 3-class contract")
    win_logits = pred.permute(0, 2, 3, 1).contiguous()             # NHWC, as the engine stores window predictions
    out = torch.empty((3, h, w), dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        check(lib.cs_swi_finalize(C.byref(plan), C.c_void_p(win_logits.data_ptr()), 3, 0, 0, h, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(), C.c_void_p(), _lib.stream_ptr()), "cs_swi_finalize")
    return out
