"""Multi-GPU partitioning for the hot path: one process per GPU, RCCL (torch "nccl") over xGMI.

* slides / tiles are independent units -> ``shard_items`` deals them to ranks, no data-path collective;
* one whole slide across ranks -> every rank owns a contiguous strip of output rows (``rank_rows``), runs every
  window that intersects it (InstanceNorm is per window, so windows never exchange state) and the uint8 mask
  strips are all-gathered (``gather_row_strips``: 2.1 MB per rank at 4096^2) before the connected-component pass.
The functions are device-agnostic plumbing over ``torch.distributed`` so the partition logic is covered by
world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .inferers import rank_rows  # noqa: F401  (re-export)


def shard_items(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of item indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def gather_row_strips(strip: torch.Tensor, img_h: int, group=None) -> torch.Tensor:
    """all-gather row strips [rows_r, W] (rows_r per ``rank_rows``) into the full [img_h, W] tensor on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return strip
    max_rows = -(-img_h // world)
    padded = strip.new_zeros((max_rows,) + tuple(strip.shape[1:]))
    padded[: strip.shape[0]] = strip
    parts: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    rows = [rank_rows(img_h, r, world) for r in range(world)]
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(rows)], 0)


def sync_gradients(model, group=None, reduce: bool = True) -> None:
    """Data-parallel gradient exchange: ONE RCCL all-reduce over the engine's flat fp32
    gradient bucket (6.5 MB at the reference width), then the mean.  `loss.backward()` leaves every parameter
    gradient as a view of ``model.flat_grad``; this re-binds the views after the in-place reduction so the optimiser
    reads the averaged values whatever autograd did with the originals.  ``reduce=False`` only re-binds (the caller has
    already averaged ``model.flat_grad`` in place, e.g. through a host-side backend)."""
    import torch.distributed as dist
    flat = model.flat_grad
    if flat is None:
        raise RuntimeError("sync_gradients: run backward first")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 and reduce:
        if flat.is_cuda and dist.get_backend(group) == "gloo":     # functional checks on a shared GPU: gloo reduces on the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)
    for key, p in model.named_parameters():
        off, numel, shape = model._grad_slots[key]
        p.grad = flat[off:off + numel].view(shape)
