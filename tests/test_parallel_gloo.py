"""world_size-2 (and 3) gloo tests of the multi-GPU partitioning logic (runs on CPU):
strips cover the image exactly once, each rank's window set is what its strip needs, all-gathered strips
re-assemble the single-rank result (computed here with the oracle tiler as a stand-in predictor path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, h, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cellseg_amd as ca
        from cellseg_amd import parallel
        torch.manual_seed(0)
        full = (torch.rand(h, w) > 0.5).to(torch.uint8)                  # what a single rank would produce
        lo, hi = ca.rank_rows(h, rank, world)
        out = parallel.gather_row_strips(full[lo:hi].clone(), h)
        ok = bool(torch.equal(out, full))
        items = list(parallel.shard_items(11, rank, world))
        gathered = [None] * world
        dist.all_gather_object(gathered, items)
        flat = [i for part in gathered for i in part]
        q.put((rank, ok, flat == list(range(11))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hw", [(2, (70, 33)), (3, (100, 17)), (2, (5, 8))])
def test_strip_gather_and_item_sharding(world, hw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hw[0], hw[1], q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and ok2 for _, ok, ok2 in res)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("h,overlap", [(4096, 0.5), (4096, 0.25), (700, 0.5), (200, 0.25)])
def test_row_strips_partition_and_window_cover(world, h, overlap):
    import cellseg_amd as ca
    plan = ca.make_plan(h, 520, (256, 256), overlap)
    hs, _ = ca.window_starts(plan)
    covered = np.zeros(h, int)
    for r in range(world):
        lo, hi = ca.rank_rows(h, r, world)
        covered[lo:hi] += 1
        if hi > lo:      # every window row that touches the strip must intersect it; the union must cover every row's windows
            need = [i for i, s in enumerate(hs) if s < hi + plan.pad_top and s + 256 > lo + plan.pad_top]
            assert need == list(range(need[0], need[-1] + 1))
    assert (covered == 1).all()


class _FakeModel:
    """Stands in for cellseg_amd.UNet after backward: a flat gradient bucket + parameters viewing slots of it."""

    def __init__(self, rank):
        self.p = {"a": torch.nn.Parameter(torch.zeros(2, 3)), "b": torch.nn.Parameter(torch.zeros(5))}
        self._grad_slots = {"a": (0, 6, (2, 3)), "b": (8, 5, (5,))}
        self.flat_grad = torch.arange(13, dtype=torch.float32) * (rank + 1)

    def named_parameters(self):
        return self.p.items()


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellseg_amd import parallel
        m = _FakeModel(rank)
        parallel.sync_gradients(m)
        mean = torch.arange(13, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = torch.allclose(m.flat_grad, mean) and torch.allclose(m.p["a"].grad, mean[:6].view(2, 3)) and \
            torch.allclose(m.p["b"].grad, mean[8:13]) and m.p["b"].grad.data_ptr() == m.flat_grad[8:].data_ptr()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res)
