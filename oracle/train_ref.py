"""CPU restatement of the reference's training step (rows L and O of SURVEY.md 8(a)).

TEST INFRASTRUCTURE (oracle) - never imported by the product path.

Follows /root/reference/baseline/model_training_3class.py:234-236 (DiceCELoss(softmax=True), AdamW(lr)) and
:250-262 (zero_grad, forward, one_hot, loss, backward, step).  MONAI's DiceCELoss / one_hot are third-party and
absent; restated from the published definition: PARITY UNPINNED against MONAI itself.
Written independently of cellseg_amd/losses.py (explicit loops over classes, log_softmax + nll) so the two can
cross-check each other.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def one_hot_ref(labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    return torch.cat([(labels.long() == c).float() for c in range(num_classes)], dim=1)


def dice_ce_ref(logits: torch.Tensor, onehot: torch.Tensor) -> torch.Tensor:
    n, c = logits.shape[:2]
    prob = torch.softmax(logits, dim=1)
    f = []
    for i in range(n):
        for k in range(c):
            inter = (prob[i, k] * onehot[i, k]).sum()
            den = onehot[i, k].sum() + prob[i, k].sum()
            f.append(1.0 - (2.0 * inter + 1e-5) / (den + 1e-5))
    dice = torch.stack(f).mean()
    target = onehot.argmax(dim=1)
    ce = F.nll_loss(F.log_softmax(logits, dim=1), target, reduction="mean")
    return dice + ce


def train_steps_ref(model, inputs, labels, steps: int, lr: float = 6e-4):
    """`steps` AdamW steps on fixed (inputs, labels); returns the list of losses (one per step, before the update)."""
    opt = torch.optim.AdamW(model.parameters(), lr)
    model.train()
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = dice_ce_ref(model(inputs), one_hot_ref(labels, 3))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses
