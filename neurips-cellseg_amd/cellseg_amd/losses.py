"""``DiceCELoss`` / ``one_hot`` - host-side mirror of what the reference's training step calls
(/root/reference/baseline/model_training_3class.py:234, :256-259: ``monai.losses.DiceCELoss(softmax=True)`` on
``monai.networks.one_hot(labels, 3)``).

Row L of SURVEY.md 8(a).  Semantics restated from MONAI: Dice over (n, c) with include_background=True,
squared_pred=False, smooth 1e-5 in numerator and denominator, mean over n*c; CE on argmax(one-hot) targets with mean
reduction; loss = dice + ce.  Forward and backward are two HIP kernels behind the C-ABI (``cs_dice_ce_forward`` /
``cs_dice_ce_backward``, csrc/kernels_loss.hip): one pass over logits and target with fp64 per-(image, class) sums, one pass
that writes d loss / d logits in the logits' own (NHWC) layout - the ~15 torch launches they replace were 0.4 ms of a 7 ms
training step.  ``one_hot`` stays a torch scatter (it runs once per batch, outside the step's critical path).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib


def one_hot(labels: torch.Tensor, num_classes: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """labels [N,1,H,W] (any numeric dtype) -> [N,C,H,W] one-hot."""
    if labels.dim() != 4 or labels.shape[1] != 1:
        raise AssertionError("labels should have a channel with length equal to one.")
    out = torch.zeros((labels.shape[0], num_classes) + tuple(labels.shape[2:]), dtype=dtype, device=labels.device)
    return out.scatter_(1, labels.long(), 1.0)


class DiceCELoss(nn.Module):
    def __init__(self, softmax: bool = True, include_background: bool = True, smooth_nr: float = 1e-5, smooth_dr: float = 1e-5,
                 lambda_dice: float = 1.0, lambda_ce: float = 1.0):
        super().__init__()
        if not softmax or not include_background:
            raise NotImplementedError("only the reference's DiceCELoss(softmax=True) configuration is implemented")
        self.smooth_nr, self.smooth_dr, self.lambda_dice, self.lambda_ce = smooth_nr, smooth_dr, lambda_dice, lambda_ce

    def forward(self, logits: torch.Tensor, target_onehot: torch.Tensor) -> torch.Tensor:
        if logits.shape != target_onehot.shape:
            raise ValueError(f"shape mismatch {tuple(logits.shape)} vs {tuple(target_onehot.shape)}")
        _lib.require_gpu(logits)
        if logits.dim() != 4 or not (2 <= logits.shape[1] <= 4):
            raise ValueError("DiceCELoss expects [N,C,H,W] logits with 2..4 classes")
        return _DiceCEFunction.apply(logits.float(), target_onehot.float().to(logits.device), self.smooth_nr, self.smooth_dr,
                                     self.lambda_dice, self.lambda_ce)


def _strides(t: torch.Tensor):
    return (C.c_int64 * 4)(*[int(v) for v in t.stride()])


class _DiceCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, nr, dr, ld, lc):
        n, c, h, w = logits.shape
        with torch.cuda.device(logits.device):
            ws = torch.empty(int(lib.cs_dice_ce_workspace_bytes(n, c)), dtype=torch.uint8, device=logits.device)
            loss = torch.empty((), dtype=torch.float32, device=logits.device)
            check(lib.cs_dice_ce_forward(C.c_void_p(logits.data_ptr()), _strides(logits), C.c_void_p(target.data_ptr()), _strides(target),
                                         n, c, h, w, nr, dr, ld, lc, C.c_void_p(loss.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                         _lib.stream_ptr()), "cs_dice_ce_forward")
        ctx.save_for_backward(logits, target, ws)
        ctx.cfg = (nr, dr, ld, lc)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        logits, target, ws = ctx.saved_tensors
        nr, dr, ld, lc = ctx.cfg
        n, c, h, w = logits.shape
        g = grad_loss.contiguous().float()
        with torch.cuda.device(logits.device):
            grad = torch.empty_like(logits)                  # same strides as the logits: the engine's NHWC buffer, no copy in UNet.backward
            if grad.stride() != logits.stride():
                grad = torch.empty_strided(logits.shape, logits.stride(), dtype=torch.float32, device=logits.device)
            check(lib.cs_dice_ce_backward(C.c_void_p(logits.data_ptr()), _strides(logits), C.c_void_p(target.data_ptr()), _strides(target),
                                          n, c, h, w, nr, dr, ld, lc, C.c_void_p(g.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                          C.c_void_p(grad.data_ptr()), _lib.stream_ptr()), "cs_dice_ce_backward")
        return grad, None, None, None, None, None
