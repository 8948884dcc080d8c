"""CPU restatement of ``monai.inferers.sliding_window_inference`` as the reference calls it.

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Call sites followed:
  /root/reference/baseline/predict.py:83-84,109          (roi 256, sw_batch_size 4)
  /root/reference/baseline/model_training_3class.py:289-293
with MONAI defaults ``overlap=0.25, mode="constant", padding_mode="constant",
cval=0``.  MONAI is third-party and absent (requirements.txt:1), so this is the
published algorithm as restated in SURVEY.md Appendix B.  PARITY UNPINNED
against MONAI itself; self-consistency (partition of unity, window counts
441/961/25/9/1) is checked in tests/test_oracle_tiler.py.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F


def scan_starts(size: int, roi: int, overlap: float) -> List[int]:
    """Window start offsets along one (already padded, size >= roi) dimension."""
    interval = roi if size == roi else int(roi * (1.0 - overlap))
    interval = max(interval, 1)
    n = 0
    while n * interval + roi < size:
        n += 1
    return [min(i * interval, size - roi) for i in range(n + 1)]


def window_grid(h: int, w: int, roi: Tuple[int, int], overlap: float) -> List[Tuple[int, int]]:
    """Row-major (H outer, W inner) list of window origins."""
    return [(hs, ws) for hs in scan_starts(h, roi[0], overlap) for ws in scan_starts(w, roi[1], overlap)]


def sliding_window_inference_ref(inputs: torch.Tensor, roi_size: Sequence[int], sw_batch_size: int,
                                 predictor: Callable[[torch.Tensor], torch.Tensor],
                                 overlap: float = 0.25) -> torch.Tensor:
    """inputs [B,C,H,W] fp32 -> [B,Cout,H,W]; constant blending, zero padding."""
    b, _, h0, w0 = inputs.shape
    rh, rw = int(roi_size[0]), int(roi_size[1])
    ph, pw = max(rh - h0, 0), max(rw - w0, 0)
    pt, pl = ph // 2, pw // 2                       # symmetric split: half before, rest after
    x = F.pad(inputs, (pl, pw - pl, pt, ph - pt), mode="constant", value=0.0)
    h, w = x.shape[-2:]
    wins = window_grid(h, w, (rh, rw), overlap)
    total = b * len(wins)
    out = cnt = None
    for g in range(0, total, sw_batch_size):
        idx = range(g, min(g + sw_batch_size, total))
        crops = [x[i // len(wins), :, wins[i % len(wins)][0]:wins[i % len(wins)][0] + rh,
                   wins[i % len(wins)][1]:wins[i % len(wins)][1] + rw] for i in idx]
        pred = predictor(torch.stack(crops, 0))
        if out is None:
            out = torch.zeros((b, pred.shape[1], h, w), dtype=pred.dtype, device=pred.device)
            cnt = torch.zeros((1, 1, h, w), dtype=pred.dtype, device=pred.device)
        for k, i in enumerate(idx):
            hs, ws = wins[i % len(wins)]
            out[i // len(wins), :, hs:hs + rh, ws:ws + rw] += pred[k]      # importance map == 1
            if i // len(wins) == 0:
                cnt[0, 0, hs:hs + rh, ws:ws + rw] += 1.0
    out = out / cnt
    return out[:, :, pt:pt + h0, pl:pl + w0]
