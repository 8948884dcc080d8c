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


# ----------------------------------------------------------------------------------------------------------------------
# window-row partition of ONE image (cs_swi_partition / parallel.seam_transfers / parallel.exchange_seam)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8, 40])
@pytest.mark.parametrize("h,overlap", [(4096, 0.5), (4096, 0.25), (700, 0.5), (1000, 0.5), (200, 0.25), (257, 0.5), (8192, 0.5)])
def test_window_row_partition_covers_every_pixel_once(world, h, overlap):
    import cellseg_amd as ca
    from cellseg_amd import parallel
    roi = 512 if h == 8192 else 256
    plan = ca.make_plan(h, 520, (roi, roi), overlap)
    hs, _ = ca.window_starts(plan)
    parts = [ca.strip_partition(plan, r, world) for r in range(world)]
    # window rows: contiguous, disjoint, complete, shares differ by at most one
    assert parts[0].win_row_begin == 0 and parts[-1].win_row_end == plan.n_h
    sizes = [p.win_row_end - p.win_row_begin for p in parts]
    assert all(a.win_row_end == b.win_row_begin for a, b in zip(parts, parts[1:])) and max(sizes) - min(sizes) <= 1
    # output rows: contiguous, disjoint, complete
    assert parts[0].row_begin == 0 and parts[-1].row_end == h
    assert all(a.row_end == b.row_begin for a, b in zip(parts, parts[1:]))
    transfers = parallel.seam_transfers(plan, world)
    for r, p in enumerate(parts):
        if p.win_row_end == p.win_row_begin:
            assert p.row_end == p.row_begin
            continue
        assert p.halo_begin <= p.win_row_begin
        incoming = {j: r0 for _, dst, j, r0 in transfers if dst == r}
        assert sorted(incoming) == list(range(p.halo_begin, p.win_row_begin))
        for y in (p.row_begin, (p.row_begin + p.row_end) // 2, p.row_end - 1):
            py = y + plan.pad_top
            cover = [i for i, s in enumerate(hs) if s <= py < s + roi]
            assert cover and p.halo_begin <= cover[0] and cover[-1] < p.win_row_end, (r, y, cover, p.halo_begin, p.win_row_end)
            for j in cover:
                if j < p.win_row_begin:                   # a halo row: the pixel lies in the part that is transferred
                    assert py - hs[j] >= incoming[j]
    for src, dst, j, r0 in transfers:
        assert src < dst and parts[src].win_row_begin <= j < parts[src].win_row_end and 0 <= r0 < roi
    if (h, overlap, world) == (4096, 0.5, 8):
        assert sizes == [3, 4, 4, 4, 4, 4, 4, 4] and plan.n_h == 31
        assert [(s, d) for s, d, _, _ in transfers] == [(r, r + 1) for r in range(7)] and all(r0 == 128 for *_, r0 in transfers)


def _seam_worker(rank, world, port, h, overlap, q, transport="p2p"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cellseg_amd as ca
        from cellseg_amd import parallel
        roi = 32
        plan = ca.make_plan(h, 70, (roi, roi), overlap)
        hs, _ = ca.window_starts(plan)
        g = torch.Generator().manual_seed(5)
        every = torch.rand(plan.n_h, plan.n_w, roi, roi, 3, generator=g)      # what all windows of the image predict
        me = ca.strip_partition(plan, rank, world)
        win = torch.full((me.win_row_end - me.halo_begin, plan.n_w, roi, roi, 3), float("nan"))
        win[me.win_row_begin - me.halo_begin:] = every[me.win_row_begin:me.win_row_end]
        ok = parallel.p2p_selfcheck(torch.device("cpu"))                      # the 1 KB ring bench.py runs before the strips leg
        n = parallel.exchange_seam(win, plan, rank, world, transport=transport)
        for j in range(me.halo_begin, me.win_row_begin):
            r0 = max(0, me.row_begin + plan.pad_top - hs[j])
            ok &= bool(torch.equal(win[j - me.halo_begin, :, r0:], every[j, :, r0:]))
        # and the unequal-height mask strips re-assemble
        full = (torch.rand(h, 70, generator=torch.Generator().manual_seed(6)) > 0.5).to(torch.uint8)
        rows = parallel.strip_rows(plan, world)
        out = parallel.gather_row_strips(full[me.row_begin:me.row_end].clone(), h, rows=rows)
        q.put((rank, ok, bool(torch.equal(out, full)), n))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["p2p", "allgather"])
@pytest.mark.parametrize("world,h,overlap", [(2, 200, 0.5), (3, 200, 0.5), (3, 171, 0.25), (4, 97, 0.5), (1, 120, 0.5)])
def test_seam_exchange_and_unequal_strip_gather(world, h, overlap, transport):
    """Both transports of the seam exchange (batched send/recv; the all-gather fallback bench.py takes when the point-to-point
    self-check fails) deliver exactly the halo rows the partition asks for."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seam_worker, args=(r, world, port, h, overlap, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and ok2 for _, ok, ok2, _ in res), res
    if world > 1:
        assert sum(n for *_, n in res) >= 2 * (world - 1)          # every seam has a sender and a receiver


def test_bench_launcher_stops_all_ranks_when_one_dies(tmp_path):
    """bench.py --gpus N: a rank that dies must not leave the others waiting in a barrier."""
    import importlib.util
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    sys.stderr.write('rank one gives up\\n'); sys.exit(3)\n"
                      "time.sleep(600)\n")
    os.environ["BENCH_DEVICE"] = "0"
    try:
        t0 = time.time()
        rc = bench.launch_ranks(3, script=str(script), argv=[])
        took = time.time() - t0
    finally:
        os.environ.pop("BENCH_DEVICE")
    assert rc == 1 and took < 30


# ----------------------------------------------------------------------------------------------------------------------
# window-LIST partition of ONE image (cs_swi_partition_windows / parallel.share_transfers / exchange_share / gather_shares)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8, 40])
@pytest.mark.parametrize("h,w,overlap", [(4096, 520, 0.5), (1000, 700, 0.25), (700, 520, 0.5), (200, 300, 0.25), (257, 513, 0.5), (8192, 1100, 0.5),
                                         (300, 260, 0.75)])
def test_window_list_partition_covers_every_pixel_once_and_names_every_needed_window(world, h, w, overlap):
    import cellseg_amd as ca
    from cellseg_amd import parallel
    roi = 512 if h == 8192 else 256
    plan = ca.make_plan(h, w, (roi, roi), overlap)
    hs, ws = ca.window_starts(plan)
    nwin = plan.n_h * plan.n_w
    shares = [ca.share_partition(plan, r, world) for r in range(world)]
    # windows: contiguous, disjoint, complete, shares differ by at most ONE window
    assert shares[0].win_begin == 0 and shares[-1].win_end == nwin
    sizes = [s.win_end - s.win_begin for s in shares]
    assert all(a.win_end == b.win_begin for a, b in zip(shares, shares[1:])) and max(sizes) - min(sizes) <= 1
    # rectangles: every pixel exactly once
    cover = np.zeros((h, w), np.int32)
    owner = np.full((h, w), -1, np.int32)
    for r, s in enumerate(shares):
        assert (s.n_rects == 0) == (s.win_end == s.win_begin) or s.n_rects > 0
        for r0, r1, c0, c1 in s.rects():
            assert 0 <= r0 < r1 <= h and 0 <= c0 < c1 <= w
            cover[r0:r1, c0:c1] += 1
            owner[r0:r1, c0:c1] = r
    assert (cover == 1).all()
    # a pixel belongs to the rank of its LAST covering window; every other covering window is its own or named by share_transfers
    transfers = parallel.share_transfers(plan, world)
    have = {}
    for src, dst, k, lo, hi in transfers:
        assert src < dst and shares[src].win_begin <= k < shares[src].win_end and 0 <= lo < hi <= roi
        assert shares[dst].slot_row_begin * plan.n_w <= k < shares[dst].win_begin
        have[(dst, k)] = (lo, hi)
    rng = np.random.default_rng(0)
    ys = np.unique(np.concatenate([rng.integers(0, h, 24), [0, h - 1], np.clip(np.array(hs) - plan.pad_top, 0, h - 1)]))
    xs = np.unique(np.concatenate([rng.integers(0, w, 24), [0, w - 1], np.clip(np.array(ws) - plan.pad_left, 0, w - 1)]))
    for y in ys:
        rows = [i for i, s0 in enumerate(hs) if s0 <= y + plan.pad_top < s0 + roi]
        for x in xs:
            cols = [j for j, s0 in enumerate(ws) if s0 <= x + plan.pad_left < s0 + roi]
            r = int(owner[y, x])
            s = shares[r]
            last = rows[-1] * plan.n_w + cols[-1]
            assert s.win_begin <= last < s.win_end, (y, x, r, last)
            for i in rows:
                assert s.slot_row_begin <= i < s.slot_row_end
                for j in cols:
                    k = i * plan.n_w + j
                    if not (s.win_begin <= k < s.win_end):
                        lo, hi = have[(r, k)]
                        assert lo <= y + plan.pad_top - hs[i] < hi, (y, x, r, k, lo, hi)
    if (h, overlap, world) == (4096, 0.5, 8):
        plan2 = ca.make_plan(4096, 4096, (256, 256), 0.5)
        sizes2 = [s.win_end - s.win_begin for s in (ca.share_partition(plan2, r, 8) for r in range(8))]
        assert sorted(sizes2) == [120] * 7 + [121] and sum(sizes2) == 961


def _share_worker(rank, world, port, h, w, overlap, q, transport):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cellseg_amd as ca
        from cellseg_amd import parallel
        roi = 32
        plan = ca.make_plan(h, w, (roi, roi), overlap)
        g = torch.Generator().manual_seed(5)
        every = torch.rand(plan.n_h * plan.n_w, roi, roi, 3, generator=g)       # what all windows of the image predict
        me = ca.share_partition(plan, rank, world)
        base = me.slot_row_begin * plan.n_w
        n_slots = (me.slot_row_end - me.slot_row_begin) * plan.n_w
        if me.win_end == me.win_begin:
            win = torch.empty((0, roi, roi, 3))          # fewer windows than ranks: what inferers.swi_infer_share enters the exchange with
        else:
            win = torch.full((max(n_slots, 1), roi, roi, 3), float("nan"))
            win[me.win_begin - base:me.win_end - base] = every[me.win_begin:me.win_end]
        n = parallel.exchange_share(win, plan, rank, world, transport=transport)
        # the flag agreement predict --split_slides runs before an image's first collective: one rank's failure is everyone's
        ok_all = parallel.all_ok(True, torch.device("cpu")) and not parallel.all_ok(rank != world - 1, torch.device("cpu"))
        assert ok_all or world == 1
        ok = True
        for src, dst, k, lo, hi in parallel.share_transfers(plan, world):
            if dst == rank:
                ok &= bool(torch.equal(win[k - base, lo:hi], every[k, lo:hi]))
        # the rectangles re-assemble into the full mask on every rank (and assemble=False returns nothing)
        full = (torch.rand(h, w, generator=torch.Generator().manual_seed(6)) > 0.5).to(torch.uint8)
        mine = torch.full((h, w), 7, dtype=torch.uint8)
        for r0, r1, c0, c1 in me.rects():
            mine[r0:r1, c0:c1] = full[r0:r1, c0:c1]
        out = parallel.gather_shares(mine, plan, world)
        none = parallel.gather_shares(mine, plan, world, assemble=False)
        q.put((rank, ok, bool(torch.equal(out, full)) and none is None, n))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["p2p", "allgather"])
@pytest.mark.parametrize("world,h,w,overlap", [(2, 200, 70, 0.5), (3, 200, 130, 0.5), (3, 171, 90, 0.25), (4, 97, 97, 0.5), (1, 120, 70, 0.5), (5, 64, 200, 0.75),
                                                (3, 30, 30, 0.5), (4, 40, 33, 0.5), (5, 48, 48, 0.5)])     # fewer windows than ranks: 1, 2 and 4 windows
def test_share_exchange_and_rectangle_gather(world, h, w, overlap, transport):
    """Both transports of the window-list seam exchange deliver exactly the window parts the partition asks for, and the all-gather of
    the finalised rectangles re-assembles the image."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, h, w, overlap, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and ok2 for _, ok, ok2, _ in res), res


# ----------------------------------------------------------------------------------------------------------------------
# bench.StripsGuard: an ASYMMETRIC failure inside the strips leg (one rank raises, its peer sits in a collective)
# ----------------------------------------------------------------------------------------------------------------------
_GUARD_SCRIPT = r"""
import json, os, sys, time, importlib.util
import torch, torch.distributed as dist
spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join({root!r}, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
guard = bench.StripsGuard(dist, rank, lambda msg: {{"value": 1.0, "scaling": "weak", "strips_error": msg}}, float(os.environ.get("DEADLINE", "60")))
mode = os.environ["MODE"]
if mode == "raise" and rank == 1:
    time.sleep(0.3)
    guard.fail("RuntimeError: provoked")                 # does not return
if mode == "hang" and rank == 1:
    time.sleep(3600)
t = torch.zeros(4)
dist.all_reduce(t)                                       # rank 0 waits here for a peer that never comes
print(json.dumps({{"unexpected": "the collective completed"}}))
"""


@pytest.mark.parametrize("mode", ["raise", "hang"])
def test_strips_guard_releases_both_ranks_with_one_line(tmp_path, mode):
    """One rank fails (or hangs) inside the guarded leg while the other is blocked in a collective: rank 0 prints the fallback line with
    strips_error, both processes exit with code 3 (a failed leg is a failed job) within seconds - nobody waits for the process-group timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "guard.py"
    script.write_text(_GUARD_SCRIPT.format(root=root))
    port = _free_port()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MODE=mode,
                                       DEADLINE="6" if mode == "hang" else "60")) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [3, 3], [o[1][-800:] for o in outs]
    lines = [ln for ln in outs[0][0].strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # (gloo announces its peers on stdout)
    import json
    line = json.loads(lines[0])
    assert line["scaling"] == "weak" and ("provoked" in line["strips_error"] if mode == "raise" else "did not finish within" in line["strips_error"])
    assert time.time() - t0 < 60


# ------------------------------------------------------------------------------------------- strip-local labelling: the collectives
class _StubLabeller:
    """Stands in for cellseg_amd.StripLabeller (device code, tests/test_gpu_ccl_strips.py) so that the two all-gathers and the gather of
    parallel.label_strips / gather_label_strips run over gloo on CPU tensors: every step checks what it is handed."""

    def __init__(self, h, w, rank, world):
        import cellseg_amd as ca
        self.rank, self.world, self.w = rank, world, w
        self.row_begin, self.row_end = ca.rank_rows(h, rank, world)
        self.n_labels = torch.tensor(7, dtype=torch.int32)

    def local(self, mask):
        return torch.arange(4 * self.w + 4, dtype=torch.int32) + 1000 * self.rank

    def resolve(self, recs):
        assert recs.shape == (self.world, 4 * self.w + 4)
        assert all(int(recs[r, 0]) == 1000 * r for r in range(self.world))
        return torch.arange(2 * self.w, dtype=torch.int32) - 1000 * self.rank

    def finish(self, ids):
        assert ids.shape == (self.world, 2 * self.w) and all(int(ids[r, 0]) == -1000 * r for r in range(self.world))
        return torch.full((self.row_end - self.row_begin, self.w), self.rank + 1, dtype=torch.int32)


def _label_strips_worker(rank, world, port, h, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cellseg_amd as ca
        from cellseg_amd import parallel
        lab, rows, n = parallel.label_strips(torch.zeros(h, w, dtype=torch.uint8), labeller=_StubLabeller(h, w, rank, world))
        ok = rows == ca.rank_rows(h, rank, world) and int(n) == 7
        dst = world - 1
        full = parallel.gather_label_strips(lab, h, dst=dst)
        if rank == dst:
            want = torch.cat([torch.full((ca.rank_rows(h, r, world)[1] - ca.rank_rows(h, r, world)[0], w), r + 1, dtype=torch.int32)
                              for r in range(world)])
            ok = ok and torch.equal(full, want)
        else:
            ok = ok and full is None
        q.put((rank, bool(ok)))
    except Exception:                                                 # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hw", [(2, (31, 4)), (3, (100, 17)), (4, (3, 5))])
def test_label_strips_collectives(world, hw):
    """Uneven strips (31 rows over 2, 3 rows over 4 ranks: an empty strip) through the record all-gather, the id all-gather and the
    gather of the label strips to the last rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_label_strips_worker, args=(r, world, port, hw[0], hw[1], q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok is True for _, ok in res), res
