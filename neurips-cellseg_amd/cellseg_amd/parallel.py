"""Multi-GPU partitioning for the hot path: one process per GPU, RCCL (torch "nccl") over xGMI.

* slides / tiles are independent units -> ``shard_items`` deals them to ranks, no data-path collective;
* one whole slide across ranks -> every rank owns a contiguous strip of output rows (``rank_rows``), runs every
  window that intersects it (InstanceNorm is per window, so windows never exchange state) and the uint8 mask
  strips are all-gathered (``gather_row_strips``: 2.1 MB per rank at 4096^2) before the connected-component pass.
The functions are device-agnostic plumbing over ``torch.distributed`` so the partition logic is covered by
world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .inferers import rank_rows  # noqa: F401  (re-export)


def shard_items(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of item indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def _collectives_on(group=None) -> bool:
    """True when a process group exists: every collective below then really runs, also at world size 1 (so the RCCL call path
    can be executed and tested on a one-GPU box); without a process group the functions are local no-ops."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _host_bounce(t: torch.Tensor, group=None) -> bool:
    """gloo cannot reduce / gather device tensors: functional checks on a shared GPU go through the host."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_row_strips(strip: torch.Tensor, img_h: int, group=None, rows: Optional[Sequence[Tuple[int, int]]] = None) -> torch.Tensor:
    """all-gather row strips [rows_r, W] into the full [img_h, W] tensor on every rank (ONE collective; RCCL on device tensors).
    `rows`: the (begin, end) rows of every rank - default ``rank_rows`` strips; ``strip_rows`` for the window-row partition."""
    import torch.distributed as dist
    if not _collectives_on(group):
        return strip
    world = dist.get_world_size(group)
    if rows is None:
        rows = [rank_rows(img_h, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in rows)
    bounce = _host_bounce(strip, group)
    src = strip.cpu() if bounce else strip
    padded = src.new_zeros((max_rows,) + tuple(src.shape[1:]))
    padded[: src.shape[0]] = src
    out = padded.new_empty((world * max_rows,) + tuple(src.shape[1:]))
    dist.all_gather_into_tensor(out, padded, group=group)
    if all(hi - lo == max_rows for lo, hi in rows):
        full = out
    else:
        full = torch.cat([out[r * max_rows: r * max_rows + hi - lo] for r, (lo, hi) in enumerate(rows)], 0)
    return full.to(strip.device) if bounce else full


def strip_rows(plan, world: int) -> List[Tuple[int, int]]:
    """Output rows of every rank under the window-row partition (``inferers.strip_partition``)."""
    from .inferers import strip_partition
    return [(st.row_begin, st.row_end) for st in (strip_partition(plan, r, world) for r in range(world))]


def seam_transfers(plan, world: int) -> List[Tuple[int, int, int, int]]:
    """Every (src_rank, dst_rank, window_row, first_local_row) the window-row partition needs: window row `window_row`, owned by
    src_rank, still covers dst_rank's first output rows - its window-local rows [first_local_row, roi_h) are what dst_rank reads.
    Deterministic and identical on every rank (both ends of a transfer derive it from the plan alone)."""
    from .inferers import strip_partition
    from ._lib import lib
    strips = [strip_partition(plan, r, world) for r in range(world)]
    owner = {}
    for r, st in enumerate(strips):
        for j in range(st.win_row_begin, st.win_row_end):
            owner[j] = r
    out = []
    for dst, st in enumerate(strips):
        for j in range(st.halo_begin, st.win_row_begin):
            hs = lib.cs_swi_window_start(j, plan.interval_h, plan.pad_h, plan.roi_h)
            out.append((owner[j], dst, j, max(0, st.row_begin + plan.pad_top - hs)))
    return out


def p2p_selfcheck(device, group=None) -> bool:
    """A 1 KB ring of batched point-to-point calls (every rank sends to rank + 1, receives from rank - 1): True when it completes
    and the payload arrives.  `bench.py` runs it before the strips leg so that a point-to-point failure of the collective library
    shows up here, and the seam exchange then takes its all-gather transport instead of ending the job.  Collective: every rank
    calls it; the verdict is agreed over the group (one failing rank switches all of them)."""
    import torch.distributed as dist
    if not _collectives_on(group):
        return True
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok = 1
    if world > 1:
        try:
            bounce = dist.get_backend(group) == "gloo"
            dev = torch.device("cpu") if bounce else device
            out = torch.full((256,), float(rank), dtype=torch.float32, device=dev)
            inp = torch.full((256,), -1.0, dtype=torch.float32, device=dev)
            ops = [dist.P2POp(dist.isend, out, (rank + 1) % world, group), dist.P2POp(dist.irecv, inp, (rank - 1) % world, group)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if inp.is_cuda:
                torch.cuda.synchronize(inp.device)
            ok = int(bool((inp == float((rank - 1) % world)).all().item()))
        except Exception:                                           # noqa: BLE001  (any failure of the library = "use the other transport")
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def all_ok(ok: bool, device, group=None) -> bool:
    """True when EVERY rank reports ok (one all-reduce MIN of a flag).  ``predict --split_slides`` calls it once per image before the first
    data-path collective: a rank that could not read or prepare an image makes all ranks skip it together instead of leaving its peers
    inside the seam exchange until the process-group timeout."""
    import torch.distributed as dist
    if not _collectives_on(group):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    bounce = _host_bounce(flag, group)
    flag = flag.cpu() if bounce else flag
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def exchange_seam(win: torch.Tensor, plan, rank: int, world: int, group=None, transport: str = "p2p") -> int:
    """Seam exchange of the window-row partition ("exchange seam partial sums with one neighbour send/recv").
    win: this rank's window logits [(win_row_end - halo_begin), n_w, roi_h, roi_w, C] with its own rows behind the halo slots
    (``inferers.swi_infer_strip``).  Sends the lower part of its last window row(s) to the rank(s) below and receives its halo
    from the rank above.  transport "p2p": batched point-to-point (RCCL send/recv over the direct xGMI link of the pair; 12 MB per
    seam at 4096^2 / stride 128).  transport "allgather": the same halves through ONE all-gather of equally sized slots (every
    rank contributes the halves it owes, zero padded; a receiver copies its halves out of the sender's slot) - world x the bytes,
    for a collective library whose point-to-point path fails (``p2p_selfcheck``).  Returns the number of transfers this rank
    took part in."""
    import torch.distributed as dist
    from .inferers import strip_partition
    if transport not in ("p2p", "allgather"):
        raise ValueError(f"exchange_seam: unknown transport {transport!r}")
    me = strip_partition(plan, rank, world)
    transfers = [t for t in seam_transfers(plan, world) if t[0] != t[1]]
    if transport == "allgather":
        if not transfers or not _collectives_on(group):
            return 0
        bounce = _host_bounce(win, group)
        numel = lambda r0: plan.n_w * (plan.roi_h - r0) * plan.roi_w * win.shape[-1]      # noqa: E731
        # slot layout, identical on every rank: the transfers of one source back to back, in seam_transfers order
        offs, fill = {}, [0] * world
        for src, dst, j, r0 in transfers:
            offs[(src, dst, j)] = fill[src]
            fill[src] += numel(r0)
        slot = max(fill)
        mine = torch.zeros(slot, dtype=win.dtype, device="cpu" if bounce else win.device)
        n = 0
        for src, dst, j, r0 in transfers:
            if src == rank:
                view = win[j - me.halo_begin, :, r0:]
                mine[offs[(src, dst, j)]: offs[(src, dst, j)] + view.numel()].copy_(view.reshape(-1))
                n += 1
        out = mine.new_empty(world * slot)
        dist.all_gather_into_tensor(out, mine, group=group)
        for src, dst, j, r0 in transfers:
            if dst == rank:
                view = win[j - me.halo_begin, :, r0:]
                o = src * slot + offs[(src, dst, j)]
                view.copy_(out[o: o + view.numel()].view(view.shape), non_blocking=True)
                n += 1
        return n
    ops, landing = [], []
    for src, dst, j, r0 in transfers:
        if rank not in (src, dst):
            continue
        view = win[j - me.halo_begin, :, r0:]                       # [n_w, roi_h - r0, roi_w, C]: contiguous per window
        if rank == src:
            buf = view.contiguous()
            buf = buf.cpu() if _host_bounce(buf, group) else buf
            ops.append(dist.P2POp(dist.isend, buf, dst, group))
        else:
            buf = torch.empty(view.shape, dtype=view.dtype, device="cpu" if _host_bounce(view, group) else view.device)
            ops.append(dist.P2POp(dist.irecv, buf, src, group))
            landing.append((view, buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for view, buf in landing:
        view.copy_(buf, non_blocking=True)
    return len(ops)


# ----------------------------------------------------------------------------------------------------------------------
# window-LIST partition (inferers.share_partition): shares of 120 / 121 windows instead of 3 / 4 window rows
# ----------------------------------------------------------------------------------------------------------------------
def share_transfers(plan, world: int) -> List[Tuple[int, int, int, int, int]]:
    """Every (src_rank, dst_rank, window, row_lo, row_hi) the window-list partition needs: window `window`, evaluated by src_rank,
    intersects a rectangle dst_rank finalises - its window-local rows [row_lo, row_hi) (all columns: one contiguous block) are what
    dst_rank reads.  src_rank < dst_rank always (a pixel belongs to the rank of its LAST covering window).  Deterministic and
    identical on every rank."""
    from .inferers import share_partition, window_starts
    shares = [share_partition(plan, r, world) for r in range(world)]
    hs, ws = window_starts(plan)                                    # padded coordinates
    out = []
    for dst, sh in enumerate(shares):
        rects = [(r0 + plan.pad_top, r1 + plan.pad_top, c0 + plan.pad_left, c1 + plan.pad_left) for r0, r1, c0, c1 in sh.rects()]
        for k in range(sh.slot_row_begin * plan.n_w, sh.win_begin):
            i, j = divmod(k, plan.n_w)
            lo, hi = None, None
            for r0, r1, c0, c1 in rects:
                a, b = max(r0, hs[i]), min(r1, hs[i] + plan.roi_h)
                if a < b and max(c0, ws[j]) < min(c1, ws[j] + plan.roi_w):
                    lo = a if lo is None else min(lo, a)
                    hi = b if hi is None else max(hi, b)
            if lo is not None:
                src = next(r for r, s in enumerate(shares) if s.win_begin <= k < s.win_end)
                out.append((src, dst, k, lo - hs[i], hi - hs[i]))
    return out


_SHARE_RUNS = {}


def _share_runs(plan, world: int):
    """share_transfers grouped for the wire: {(src, dst): [(row_lo, row_hi, first_window, count), ...]} - consecutive windows of one pair
    that want the same rows travel as ONE tensor [count, rows, roi_w, C], a plain slice of the window slots on both sides (two runs per
    pair at overlap 0.5: the lower halves of the n_w windows above the cut and the left neighbour of the first own window).  Cached per
    plan: the partition is fixed for a slide size."""
    key = (tuple(getattr(plan, f) for f, _ in plan._fields_), world)
    runs = _SHARE_RUNS.get(key)
    if runs is None:
        runs = {}
        for src, dst, k, lo, hi in share_transfers(plan, world):
            lst = runs.setdefault((src, dst), [])
            if lst and lst[-1][0] == lo and lst[-1][1] == hi and lst[-1][2] + lst[-1][3] == k:
                lst[-1] = (lo, hi, lst[-1][2], lst[-1][3] + 1)
            else:
                lst.append((lo, hi, k, 1))
        _SHARE_RUNS[key] = runs
    return runs


def exchange_share(win: torch.Tensor, plan, rank: int, world: int, group=None, transport: str = "p2p") -> int:
    """Seam exchange of the window-LIST partition.  win: this rank's window slots [n_slots, roi_h, roi_w, C] (``inferers.swi_infer_share``:
    slot of window k = k - slot_row_begin * n_w).  Sends the parts of its windows that later ranks finalise under and receives the parts
    of earlier ranks' windows that cover its own rectangles; transports as ``exchange_seam`` (batched RCCL send/recv, or ONE all-gather
    of equally sized slots for a library whose point-to-point path fails).  Returns the number of tensors this rank sent or received."""
    import torch.distributed as dist
    from .inferers import share_partition
    if transport not in ("p2p", "allgather"):
        raise ValueError(f"exchange_share: unknown transport {transport!r}")
    runs = _share_runs(plan, world)
    if not runs or not _collectives_on(group):
        return 0
    base = share_partition(plan, rank, world).slot_row_begin * plan.n_w
    bounce = _host_bounce(win, group)
    wire = torch.device("cpu") if bounce else win.device
    row_floats = plan.roi_w * win.shape[-1]
    n = 0
    if transport == "allgather":
        offs, fill = {}, [0] * world
        for (src, dst), rs in sorted(runs.items()):
            for lo, hi, k0, cnt in rs:
                offs[(src, dst, k0)] = fill[src]
                fill[src] += cnt * (hi - lo) * row_floats
        slot = max(fill)
        mine = torch.zeros(slot, dtype=win.dtype, device=wire)
        for (src, dst), rs in sorted(runs.items()):
            if src == rank:
                for lo, hi, k0, cnt in rs:
                    o, numel = offs[(src, dst, k0)], cnt * (hi - lo) * row_floats
                    mine[o:o + numel].view(cnt, hi - lo, plan.roi_w, win.shape[-1]).copy_(win[k0 - base:k0 - base + cnt, lo:hi])
                    n += 1
        out = mine.new_empty(world * slot)
        dist.all_gather_into_tensor(out, mine, group=group)
        for (src, dst), rs in sorted(runs.items()):
            if dst == rank:
                for lo, hi, k0, cnt in rs:
                    o, numel = src * slot + offs[(src, dst, k0)], cnt * (hi - lo) * row_floats
                    win[k0 - base:k0 - base + cnt, lo:hi].copy_(out[o:o + numel].view(cnt, hi - lo, plan.roi_w, win.shape[-1]), non_blocking=True)
                    n += 1
        return n
    ops, landing = [], []
    for (src, dst), rs in sorted(runs.items()):
        if rank not in (src, dst):
            continue
        for lo, hi, k0, cnt in rs:
            view = win[k0 - base:k0 - base + cnt, lo:hi]             # [cnt, rows, roi_w, C]: contiguous per window
            if rank == src:
                buf = view.contiguous()
                ops.append(dist.P2POp(dist.isend, buf.cpu() if bounce else buf, dst, group))
            else:
                buf = torch.empty(view.shape, dtype=win.dtype, device=wire)
                ops.append(dist.P2POp(dist.irecv, buf, src, group))
                landing.append((view, buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for view, buf in landing:
        view.copy_(buf, non_blocking=True)
    return len(ops)


def gather_shares(mask: torch.Tensor, plan, world: int, group=None, assemble: bool = True) -> Optional[torch.Tensor]:
    """all-gather of the rectangles every rank finalised (``inferers.share_partition``) into the full [img_h, img_w] tensor: ONE
    collective over equally sized flat slots (RCCL on device tensors).  mask: this rank's full-size tensor with its rectangles filled in.
    Every rank takes part; ``assemble=False`` skips the unpacking on ranks that do not need the full tensor (only the rank that labels
    this slide does) and returns None."""
    import torch.distributed as dist
    from .inferers import share_partition
    if not _collectives_on(group):
        return mask
    shares = [share_partition(plan, r, world) for r in range(world)]
    sizes = [sum((r1 - r0) * (c1 - c0) for r0, r1, c0, c1 in s.rects()) for s in shares]
    slot = max(max(sizes), 1)
    rank = dist.get_rank(group)
    bounce = _host_bounce(mask, group)
    wire = torch.device("cpu") if bounce else mask.device
    mine = torch.zeros(slot, dtype=mask.dtype, device=wire)
    o = 0
    for r0, r1, c0, c1 in shares[rank].rects():
        part = mask[r0:r1, c0:c1].reshape(-1)
        mine[o:o + part.numel()].copy_(part)
        o += part.numel()
    out = mine.new_empty(world * slot)
    dist.all_gather_into_tensor(out, mine, group=group)
    if not assemble:
        return None
    full = torch.empty_like(mask)
    for r, s in enumerate(shares):
        o = r * slot
        for r0, r1, c0, c1 in s.rects():
            n = (r1 - r0) * (c1 - c0)
            full[r0:r1, c0:c1] = out[o:o + n].view(r1 - r0, c1 - c0).to(mask.device)
            o += n
    return full


# ----------------------------------------------------------------------------------------------------------------------
# strip-local connected components with a seam merge (third row, first option)
# ----------------------------------------------------------------------------------------------------------------------
def _all_gather_flat(t: torch.Tensor, group=None) -> torch.Tensor:
    """[n] -> [world, n] on every rank (ONE collective; through the host under gloo)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bounce = _host_bounce(t, group)
    src = (t.cpu() if bounce else t).contiguous().view(-1)
    out = src.new_empty(world * src.numel())                        # flat on both sides: gloo accepts nothing else
    dist.all_gather_into_tensor(out, src, group=group)
    out = out.view((world,) + tuple(t.shape))
    return out.to(t.device) if bounce else out


def label_strips(mask: torch.Tensor, group=None, hole_area: int = 64, min_size: int = 16, labeller=None):
    """``measure.label(remove_small_objects(remove_small_holes(mask), 16))`` (predict.py:113) of ONE slide over the ranks of `group`:
    rank r filters and labels its ``rank_rows`` strip of the thresholded mask (+ an 80-row halo) on its own GPU, the strips' first / last
    rows are exchanged in two small all-gathers (64 KB + 32 KB per rank at 4096 columns) around the seam union-find, and every rank ends
    with the FINAL int32 labels of its rows - bit-identical to the single-GPU pass, numbering included.  mask: the full [H, W] mask
    (every rank has it after ``gather_shares``) - only the strip's rows and halo are read.
    -> (labels [rows, W] int32, (row_begin, row_end), n_labels device int32 scalar = components of the whole slide).
    Without a process group: one strip = the whole slide."""
    import torch.distributed as dist
    from .postprocess import StripLabeller
    on = _collectives_on(group)
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if on else (1, 0)
    if labeller is None:
        labeller = StripLabeller(mask.shape[0], mask.shape[1], rank, world, mask.device, hole_area, min_size)
    rec = labeller.local(mask)
    recs = _all_gather_flat(rec, group) if on else rec[None]
    ids = labeller.resolve(recs)
    all_ids = _all_gather_flat(ids, group) if on else ids[None]
    labels = labeller.finish(all_ids)
    return labels, (labeller.row_begin, labeller.row_end), labeller.n_labels


def gather_label_strips(strip: torch.Tensor, img_h: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """The ranks' label strips ([rows_r, W] int32, ``rank_rows`` strips) -> the full [img_h, W] map on rank `dst` (one gather: 8 MB per
    rank at 4096^2, straight onto dst's links); None on the other ranks."""
    import torch.distributed as dist
    if not _collectives_on(group):
        return strip
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = [rank_rows(img_h, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in rows)
    bounce = _host_bounce(strip, group)
    src = strip.cpu() if bounce else strip
    if src.shape[0] != max_rows:
        padded = src.new_zeros((max_rows,) + tuple(src.shape[1:]))
        padded[: src.shape[0]] = src
        src = padded
    parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
    dist.gather(src.contiguous(), parts, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(rows)], 0)
    return full.to(strip.device) if bounce else full


def sync_gradients(model, group=None, reduce: bool = True, rebind: bool = True) -> None:
    """Data-parallel gradient exchange: ONE RCCL all-reduce over the engine's flat fp32
    gradient bucket (6.5 MB at the reference width), then the mean.  `loss.backward()` leaves every parameter
    gradient as a view of ``model.flat_grad``; this re-binds the views after the in-place reduction so the optimiser
    reads the averaged values whatever autograd did with the originals.  ``reduce=False`` only re-binds (the caller has
    already averaged ``model.flat_grad`` in place, e.g. through a host-side backend); ``rebind=False`` skips the re-binding
    (``TrainStep``: the gradients live in one static bucket)."""
    import torch.distributed as dist
    flat = model.flat_grad
    if flat is None:
        raise RuntimeError("sync_gradients: run backward first")
    if _collectives_on(group) and reduce:                          # also at world size 1: the collective really runs
        world = dist.get_world_size(group)
        if _host_bounce(flat, group):                               # functional checks on a shared GPU: gloo reduces on the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
            if world > 1:
                flat.mul_(1.0 / world)
        elif dist.get_backend(group) == "nccl":
            # RCCL divides inside the reduction: one call, no separate scaling launch behind it (AVG at world size 1 is the identity)
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)      # gloo has no AVG
            if world > 1:
                flat.mul_(1.0 / world)
    if not rebind:
        return
    for key, p in model.named_parameters():
        off, numel, shape = model._grad_slots[key]
        p.grad = flat[off:off + numel].view(shape)
