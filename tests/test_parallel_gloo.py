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
