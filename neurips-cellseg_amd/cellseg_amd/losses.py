"""``DiceCELoss`` / ``one_hot`` - host-side mirror of what the reference's training step calls
(/root/reference/baseline/model_training_3class.py:234, :256-259: ``monai.losses.DiceCELoss(softmax=True)`` on
``monai.networks.one_hot(labels, 3)``).

Row L of SURVEY.md 8(a): negligible next to the network (a few reductions over [N,3,256,256]), so it stays in
torch ops on the GPU (SURVEY: "may stay in torch"); the network forward/backward underneath is the HIP engine.
Semantics restated from MONAI: Dice over (n, c) with include_background=True, squared_pred=False, smooth 1e-5 in
numerator and denominator, mean over n*c; CE on argmax(one-hot) targets with mean reduction; loss = dice + ce.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        p = torch.softmax(logits, 1)
        dims = (2, 3)
        inter = (p * target_onehot).sum(dims)
        denom = target_onehot.sum(dims) + p.sum(dims)
        dice = (1.0 - (2.0 * inter + self.smooth_nr) / (denom + self.smooth_dr)).mean()
        ce = F.cross_entropy(logits, target_onehot.argmax(1))
        return self.lambda_dice * dice + self.lambda_ce * ce
