#!/usr/bin/env python3
"""bench.py - headline benchmark: Mpixels/s of whole-slide sliding-window inference on MI355X.

One "step" = the reference's per-image hot path (baseline/predict.py:107-113) on one synthetic 3x4096x4096 slide that is already resident
in HBM: sliding-window U-Net inference (256 window / 128 stride = BASELINE.json configs[2], 961 windows) -> softmax -> class-1 > 0.5 ->
remove_small_holes -> remove_small_objects -> 8-connected label -> int32 label map copied to (pinned) host memory, ONE SLIDE AT A TIME:
every timed step ends with its label map on the host (SURVEY.md 8(d)'s per-image figure; round 6 - before, `value` was the two-slide
software pipeline, which is still measured and reported as `value_pipelined`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train] [--full]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one rank per GPU, RCCL)

`--gpus N` with N > 1 and no torchrun environment: this process starts N child ranks itself (before it has touched the GPU) and forwards
rank 0's JSON line.

Multi-GPU legs:
  slides  (weak scaling)             every rank segments its own slide, no data-path collective
  strips  (headline for N > 1,       ONE slide split by its WINDOW LIST (shares differ by one window at most; no window evaluated twice),
           strong scaling)           window parts sent to the later neighbour (RCCL send/recv), all-gather (RCCL) of the finalised uint8
                                     rectangles, CC + label copy on rank (slide index mod N) on its own stream
  train   (`train_step` object; `--mode train` makes it the headline) data-parallel training step, batch 32x3x256x256 per
          GPU (BASELINE.json configs[1]/[3]): forward + DiceCE + backward + ONE flat-bucket RCCL all-reduce + AdamW

Output (benchlib/line.py): the LAST stdout line is ONE compact JSON object (< 4 kB: the contract keys, `roofline` of the dominant kernel
from HIP events on the launch stream, `cpu_baseline` = the oracle on the host cores, one scalar per secondary leg); the full record of
every leg goes to gpurun_out/bench_full.json and to stderr.  The default run keeps every secondary leg that costs seconds; `--full` adds
the long ones (150-step trainings for the accuracy of the low-precision modes, the PyTorch-ROCm eager comparator, the CPU tile / training
baselines).  The legs live in benchlib/.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "neurips-cellseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

from benchlib import line as benchline  # noqa: E402
from benchlib.accuracy import lowprec_accuracy  # noqa: E402
from benchlib.common import (DTYPE, MFMA_PEAK, PEAK_F32_MFMA_TFLOPS, PEAK_FP8_MFMA_TFLOPS, PEAK_HBM_GBS, WIDTHS, WINDOW_GFLOP,  # noqa: E402,F401
                             effective_cpus, make_model, synthetic_wsi, time_steps)
from benchlib.cpu_baseline import cpu_baseline, cpu_baseline_tiles_and_train  # noqa: E402
from benchlib.eager import eager_rocm  # noqa: E402
from benchlib.ranks import StripsGuard, launch_ranks  # noqa: E402,F401
from benchlib.roofline import dominant_kernel, kernel_shares  # noqa: E402
from benchlib.traffic import live_pmc_traffic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: whole-slide inference is the headline (train step reported as `train_step`); train: the "
                         "data-parallel training step is the headline")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--roi", type=int, default=256)
    ap.add_argument("--overlap", type=float, default=0.5, help="0.5 = stride 128 (BASELINE configs[2]); reference default 0.25")
    ap.add_argument("--width", choices=sorted(WIDTHS), default="baseline32")
    ap.add_argument("--engine-batch", type=int, default=0, help="windows per engine call; 0 = the engine's own choice (equal chunks, one per internal "
                                                                "stream where that fits 512 windows: cellseg_amd.balanced_batch)")
    ap.add_argument("--streams", type=int, default=2, help="internal streams the window chunks are dealt to (1 = serial)")
    ap.add_argument("--precision", choices=sorted(MFMA_PEAK), default="fp32",
                    help="matrix-core arithmetic of the convolutions")
    ap.add_argument("--shard", choices=["slides", "strips"], default=None,
                    help="headline multi-GPU partition: ONE slide split by its window list over the ranks with the seam exchange and the "
                         "all-gather of the uint8 mask strips before the connected-component pass (strong scaling; the default for "
                         "--gpus N > 1: north_star's '>= 6x at 8 GPUs' is about one WSI) or one slide per rank (weak scaling, no "
                         "collective); the other partition is reported as a secondary object")
    ap.add_argument("--strip-batch", type=int, default=0, help="engine batch of the strips leg; 0 = the engine's own choice (a rank's windows in "
                                                               "equal chunks, one per internal stream: 124 windows at 8 ranks -> 2 x 62)")
    ap.add_argument("--train-batch", type=int, default=32, help="tiles per GPU of the training step (BASELINE configs[1]/[3]: 32)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--cc-stream", type=int, default=1, help="1: the connected-component pass of slide k runs on its own stream beside "
                                                              "the convolutions of slide k + 1 (0: same stream)")
    ap.add_argument("--cc", choices=["turns", "strips"], default="turns",
                    help="connected components of the strips leg: turns = slide k's gathered mask is labelled whole by rank k mod N (no label "
                         "traffic; best throughput); strips = every rank filters and labels its own rows of every slide, two small all-gathers "
                         "around the seam union-find, label strips gathered to rank k mod N (shortest latency of ONE slide)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic: quote profiles/pmc_traffic.json instead of running the two "
                                                                   "rocprofv3 --pmc child passes (default command at N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (reference width, other precisions, "
                                                                "the other partition, the training step)")
    ap.add_argument("--full", action="store_true", help="also run the long secondary legs: accuracy of the low-precision modes on nets trained for 150 "
                                                        "steps (32-base and configs[4]'s 64-base), the split-precision modes, the PyTorch-ROCm eager "
                                                        "comparator of configs[1], the CPU tile / training-step baselines")
    args = ap.parse_args()
    t_bench0 = time.time()
    leg_seconds = {}

    class leg:
        """with leg("name"): wall seconds of every leg of the run land in the full record (`leg_seconds`), so that the default run's length
        can be budgeted leg by leg."""
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            torch.cuda.synchronize()
            self.t0 = time.time()

        def __exit__(self, *exc):
            torch.cuda.synchronize()
            leg_seconds[self.name] = round(leg_seconds.get(self.name, 0.0) + time.time() - self.t0, 2)
            return False

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))           # nothing above has touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.shard is None:
        args.shard = "strips" if world > 1 else "slides"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the hot path")
    # BENCH_DEVICE / BENCH_BACKEND exist only to exercise the multi-rank code path on a 1-GPU box (ranks share cuda:0, gloo)
    local = int(os.environ.get("BENCH_DEVICE", local))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # BENCH_FORCE_DIST=1: initialise the process group at world size 1 too, so that the RCCL calls of the data path (flat-bucket
    # all-reduce, mask all-gather) really execute on a one-GPU box
    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # the line contract is ONE JSON line on stdout: whatever the collective library prints while it connects (gloo announces its
        # peers on stdout) goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            pg_timeout = datetime.timedelta(seconds=int(os.environ.get("BENCH_PG_TIMEOUT_S", "240")))      # a hung collective fails, it does not stall
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
            else:
                dist.init_process_group(backend, timeout=pg_timeout)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    barrier = (lambda: dist.barrier()) if dist else (lambda: None)

    def max_over_ranks(vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=red_dev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    import cellseg_amd as ca
    from cellseg_amd import _lib, parallel

    seed = 2022
    roi = (args.roi, args.roi)
    plan = ca.make_plan(args.size, args.size, roi, args.overlap)
    n_windows = plan.n_h * plan.n_w
    mpix = args.size * args.size / 1e6
    host = torch.empty((args.size, args.size), dtype=torch.int32, pin_memory=True)    # where "labels on host" land
    copy_stream, done = torch.cuda.Stream(dev), torch.cuda.Event()
    cc_stream, mask_ready = torch.cuda.Stream(dev), torch.cuda.Event()

    # ------------------------------------------------------------------------------------------------------------------
    # inference legs
    # ------------------------------------------------------------------------------------------------------------------
    images = {}

    def slide_pair(strips):
        """Two DIFFERENT slides per leg, alternating step by step (strips: every rank holds the SAME pair; slides: every rank its own)."""
        keys = (seed, seed + 1000) if strips else (seed + rank, seed + 1000 + rank)
        for key in keys:
            if key not in images:
                images[key] = torch.from_numpy(synthetic_wsi(args.size, key)).to(dev)      # resident in HBM before timing
        return [images[k] for k in keys]

    share_parts = [ca.share_partition(plan, r, world) for r in range(world)]

    no_labels, no_num = torch.zeros((1, 1), dtype=torch.int32, device=dev), torch.zeros((), dtype=torch.int32, device=dev)
    strip_labeller = [None]

    # before the strips leg: a 1 KB ring of batched send/recv; when the library's point-to-point path fails the seam halves go
    # through one all-gather instead (reported as seam_transport) - the job does not end there
    # (the check runs right before the first strips leg, inside the guarded region - StripsGuard below)
    seam = {"transport": None}
    fail_rank = os.environ.get("BENCH_STRIPS_FAIL_RANK")        # test hook: this rank raises inside its seam exchange (an ASYMMETRIC failure)

    def ensure_seam_transport():
        if dist is not None and seam["transport"] is None:
            seam["transport"] = os.environ.get("BENCH_SEAM_TRANSPORT") or ("p2p" if parallel.p2p_selfcheck(dev) else "allgather")

    # phase timing of the strips leg (exchange_ms / gather_ms / cc_ms of the line): HIP events on the launch stream around the three
    # phases, only while phases["on"] (a short pass after the timed loops, everything on one stream)
    phases = {"on": False, "ev": {"exchange": [], "gather": [], "cc": []}}

    class phase:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            if phases["on"]:
                self.a = torch.cuda.Event(enable_timing=True)
                self.a.record()

        def __exit__(self, *exc):
            if phases["on"]:
                b = torch.cuda.Event(enable_timing=True)
                b.record()
                phases["ev"][self.name].append((self.a, b))
            return False

    def seam_exchange(win, pl, r, w):
        if fail_rank is not None and int(fail_rank) == rank:
            raise RuntimeError(f"provoked failure on rank {rank} (BENCH_STRIPS_FAIL_RANK)")
        if os.environ.get("BENCH_STRIPS_HANG_RANK") == str(rank):     # test hook: this rank never leaves its exchange
            time.sleep(3600)
        with phase("exchange"):
            return parallel.exchange_share(win, pl, r, w, transport=seam["transport"] or "p2p")

    def make_step(model, strips, d2h, cc_overlap=True, sim=None):
        imgs = slide_pair(strips)
        it = [0]
        force_label = [None]                       # strips leg: None = the ranks take turns; True / False = this rank labels / does not
        pipelined = bool(args.cc_stream) and cc_overlap

        def labelled(mask):
            """The connected-component pass of slide k runs on its own stream (ordered after the slide's mask by an event), so its
            latency-bound kernels execute beside the convolutions of slide k + 1 - the same software pipeline the label copy
            already is part of.  Every timed slide is complete when the timed region ends (torch.cuda.synchronize())."""
            if not pipelined:
                with phase("cc"):
                    return ca.mask_to_instances(mask, return_num=True)
            mask_ready.record()
            cc_stream.wait_event(mask_ready)
            with torch.cuda.stream(cc_stream):
                lab, num = ca.mask_to_instances(mask, return_num=True)
            mask.record_stream(cc_stream)
            return lab, num

        def labelled_strips(full, dst, mine):
            """--cc strips: this rank's rows of the slide are filtered and labelled here (cs_ccl_strip_*, SURVEY.md 8(e) third row: strip-local
            union-find + seam merge; two all-gathers of edge rows), the int32 strips are gathered to rank dst, which holds the slide's label map."""
            if strip_labeller[0] is None:
                strip_labeller[0] = ca.StripLabeller(args.size, args.size, rank, world, dev)
            if pipelined:
                mask_ready.record()
                cc_stream.wait_event(mask_ready)
            with torch.cuda.stream(cc_stream if pipelined else torch.cuda.current_stream()), phase("cc"):
                lab, _, num = parallel.label_strips(full, labeller=strip_labeller[0])
                whole = parallel.gather_label_strips(lab, args.size, dst=dst)
            if pipelined:
                full.record_stream(cc_stream)
            return (whole, num) if mine else (no_labels, no_num)

        def step_slide():
            image = imgs[it[0] % 2]
            it[0] += 1
            _, _, mask = ca.swi_infer_image(model, image, roi, args.overlap, engine_batch=args.engine_batch or None, want_logits=False,
                                            want_mask=True, streams=args.streams)
            return labelled(mask)

        def step_strips():
            """One slide across ranks, split by the window LIST (961 windows over 8 ranks = 120 / 121 each; no window evaluated twice):
            every rank runs its windows, the parts later ranks finalise under go to them (RCCL send/recv), every rank finalises its
            rectangles, the uint8 rectangles are all-gathered (RCCL) and ONE rank - slide k: rank k mod N, so that the connected-component
            pass and the label copy cost every rank 1/N of a slide instead of rank 0 all of it - labels the full mask on its CC stream
            beside its next slide; its label map (copied to its pinned host buffer) is the job's output for that slide."""
            k = it[0]
            image = imgs[k % 2]
            it[0] += 1
            mine = force_label[0] if force_label[0] is not None else (k % world == rank)
            _, _, _, mask = ca.swi_infer_share(model, image, roi, args.overlap, rank, world, engine_batch=args.strip_batch or None,
                                               streams=args.streams, exchange=seam_exchange)
            if args.cc == "strips":
                with phase("gather"):
                    full = parallel.gather_shares(mask, plan, world, assemble=True)
                return labelled_strips(full, k % world, mine)
            with phase("gather"):
                full = parallel.gather_shares(mask, plan, world, assemble=mine)
            return labelled(full) if mine else (no_labels, no_num)

        def step_sim():
            """ONE GPU doing the work of rank sim[0] of sim[1] of the strips leg: its windows, the local pack / unpack copies of the
            exchange (the link transfer itself, 12 MB over one xGMI link, is not simulated), the finalize of its rectangles, a device-local
            copy of the gathered mask's size in place of the all-gather, and - on its turn, every sim[1]-th slide - the connected-component
            pass over a FULL mask (sim[2], from the single-GPU leg) on the CC stream."""
            k = it[0]
            image = imgs[k % 2]
            it[0] += 1
            r, w, full_mask = sim[:3]

            def fake_exchange(win, pl, rr, ww):
                base = sh_me.slot_row_begin * pl.n_w
                own0 = sh_me.win_begin - base                      # stand-in source on this GPU: the rank's own first windows
                for (src, dst), runs in parallel._share_runs(pl, ww).items():
                    for lo, hi, k0, cnt in runs:
                        if dst == rr:            # pack (stood in for locally), a device-local copy in place of the link, unpack
                            win[k0 - base:k0 - base + cnt, lo:hi].copy_(win[own0:own0 + cnt, lo:hi].contiguous().clone())
                        elif src == rr:
                            win[k0 - base:k0 - base + cnt, lo:hi].contiguous().clone()
            sh_me = ca.share_partition(plan, r, w)
            ca.swi_infer_share(model, image, roi, args.overlap, r, w, engine_batch=args.strip_batch or None, streams=args.streams,
                               exchange=fake_exchange)
            gathered = full_mask.clone()         # a device-local copy of the gathered mask's size in place of the all-gather
            if len(sim) > 3:                     # --cc strips: this rank's strip of EVERY slide (records / ids of the other strips precomputed)
                lab_s, all_recs, all_ids = sim[3]
                if pipelined:
                    mask_ready.record()
                    cc_stream.wait_event(mask_ready)
                with torch.cuda.stream(cc_stream if pipelined else torch.cuda.current_stream()):
                    lab_s.local(gathered)
                    lab_s.resolve(all_recs.clone())      # clones: device-local copies in place of the two all-gathers
                    lab = lab_s.finish(all_ids.clone())
                    whole = torch.cat([lab] * w)[:args.size] if k % w == r else None      # in place of the gather: the strips' bytes land here
                if pipelined:
                    gathered.record_stream(cc_stream)
                return (whole, lab_s.n_labels) if k % w == r else (no_labels, no_num)
            return labelled(gathered) if k % w == r else (no_labels, no_num)

        inner = step_sim if sim else (step_strips if (strips and dist is not None) else step_slide)
        inner.it, inner.force_label = it, force_label
        if not d2h:
            return inner

        def step_d2h():
            """The label copy runs on its own stream (ordered after this slide's CC by an event), so the PCIe transfer of slide k
            overlaps the inference of slide k + 1 - what the predict CLI does with its writer thread.  Every timed step's labels
            are on the host when the timed region ends (torch.cuda.synchronize() drains both streams)."""
            lab, num = inner()
            if lab is no_labels:
                return lab, num                              # strips leg: only rank 0 labels (and copies) the gathered mask
            done.record(cc_stream if pipelined else None)
            copy_stream.wait_event(done)
            with torch.cuda.stream(copy_stream):
                host.copy_(lab, non_blocking=True)
            lab.record_stream(copy_stream)
            return lab, num
        step_d2h.it, step_d2h.force_label = it, force_label
        return step_d2h

    def measure_infer(width, precision, strips, with_prof, extras=False):
        """-> dict(dt (ONE slide at a time, labels on the host at the end of every step: the headline), dt_pipe (the same K steps as a
        software pipeline: copy and connected components of slide k beside the windows of slide k + 1), dt_dev (pipelined, labels left in
        HBM), prof, ...): K timed steps each, max over ranks.
        extras (rank 0's primary leg): the reference's timer scope, the strips-leg simulation, the 10 GB envelope."""
        model = make_model(WIDTHS[width], seed).to(dev)
        model.precision = precision
        step_host, step_dev = make_step(model, strips, True), make_step(model, strips, False)
        step_one = make_step(model, strips, True, cc_overlap=False)
        step_prof = make_step(model, strips, False, cc_overlap=False)      # per-kernel HIP events: everything on one stream
        res = {}
        per_step = []

        def step_image():
            """image tensor resident in HBM -> int32 label map in (pinned) host memory, synchronised: nothing of slide
            k + 1 starts before slide k is complete on the host."""
            t0 = time.perf_counter()
            out = step_one()
            torch.cuda.synchronize()
            per_step.append((time.perf_counter() - t0) * 1e3)
            return out

        with torch.no_grad():
            torch.cuda.reset_peak_memory_stats(dev)
            for _ in range(args.warmup):
                step_image()
            per_step.clear()
            dt, _ = time_steps(step_image, args.steps, barrier)
            ms_sorted = sorted(per_step)
            res["latency"] = {"latency_ms": round(ms_sorted[len(ms_sorted) // 2], 3), "min_ms": round(ms_sorted[0], 3), "max_ms": round(ms_sorted[-1], 3),
                              "reps": len(ms_sorted)}
            res["workspace_bytes"] = int(sum(v.numel() * v.element_size() for v in getattr(model, "_ws", {}).values()))
            res["peak_allocated_bytes"] = int(torch.cuda.max_memory_allocated(dev))
            for _ in range(min(args.warmup, 2)):
                step_host()
            dt_pipe, _ = time_steps(step_host, args.steps, barrier)
            dt_dev, _ = time_steps(step_dev, args.steps, barrier)
            # one more step, untimed, on the FIRST slide of the pair with rank 0 labelling it: the label map the legs are compared by
            step_dev.it[0], step_dev.force_label[0] = 0, rank == 0
            labels, num = step_dev()
            step_dev.force_label[0] = None
            dt_prof = 0.0
            if with_prof:
                # same K steps again with HIP events around every kernel launch (on the launch stream) for the roofline
                _lib.lib.cs_prof_reset()
                _lib.lib.cs_prof_enable(1)
                dt_prof, _ = time_steps(step_prof, args.steps, barrier)
                _lib.lib.cs_prof_enable(0)
                res["prof"] = _lib.prof_collect()
            if strips and dist is not None:
                res["phases"] = measure_phases(model)
            if extras:
                with leg("reference_scope_tiff"):
                    res["incl_tiff"] = measure_reference_scope(model)
                if world == 1 and dist is None:
                    with leg("strips_sim"):
                        res["strips_sim"] = measure_strips_sim(model, dt_dev / args.steps)
                    with leg("envelope_10gb"):
                        res["envelope"] = measure_envelope(model)
        dt, dt_pipe, dt_dev, dt_prof = max_over_ranks([dt, dt_pipe, dt_dev, dt_prof])
        res.update(dt=dt, dt_pipe=dt_pipe, dt_dev=dt_dev, dt_prof=dt_prof, n_instances=int(num.item()),
                   fg_frac=float((labels > 0).float().mean().item()), labels=labels)
        return res

    def measure_phases(model, reps=3):
        """Strips leg: where a slide's time goes on this rank - seam exchange, mask all-gather, connected components - from HIP events on the
        launch stream (one stream, no overlap between slides), mean over `reps` slides, then the MAX over ranks (rank k labels slide k)."""
        step = make_step(model, True, False, cc_overlap=False)
        step()
        torch.cuda.synchronize()
        for v in phases["ev"].values():
            v.clear()
        phases["on"] = True
        for _ in range(max(reps, world)):                       # at least one labelling turn per rank
            step()
        phases["on"] = False
        torch.cuda.synchronize()
        n = max(reps, world)
        mine = [sum(a.elapsed_time(b) for a, b in phases["ev"][k]) / n for k in ("exchange", "gather")]
        cc = [a.elapsed_time(b) for a, b in phases["ev"]["cc"]]
        mine.append(sum(cc) / len(cc) if cc else 0.0)           # per labelled slide
        ex, ga, c = max_over_ranks(mine)
        return {"exchange_ms": round(ex, 3), "gather_ms": round(ga, 3), "cc_ms": round(c, 3),
                "note": "HIP events on the launch stream, one stream; exchange / gather: per slide, cc: per slide this rank labels; max over ranks"}

    def measure_envelope(model, budget=10 << 30, reps=3):
        """The reference states < 10 GB of GPU memory for inference (README.md:127).  The default engine batch takes what 288 GB offer
        (`workspace_bytes`); this leg finds the largest engine batch whose tiler workspace + the slide's own tensors stay under 10 GiB and
        times the same per-image step with it -> value_10gb."""
        import ctypes as C
        eng = model.engine()
        fixed = 3 * args.size * args.size * 4 + args.size * args.size * (1 + 4)          # image, mask, labels
        size_of = lambda b: int(_lib.lib.cs_swi_workspace_bytes(eng.handle, C.byref(plan), int(b)))      # noqa: E731
        lo, hi = 1, args.engine_batch or ca.balanced_batch(n_windows, args.streams)
        if size_of(hi) + fixed <= budget:
            lo = hi
        else:
            while hi - lo > 1:
                mid = (lo + hi) // 2
                lo, hi = (mid, hi) if size_of(mid) + fixed <= budget else (lo, mid)
        imgs = slide_pair(False)
        ms = []
        for i in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, mask = ca.swi_infer_image(model, imgs[i % 2], roi, args.overlap, engine_batch=lo, want_logits=False, want_mask=True, streams=args.streams)
            lab = ca.mask_to_instances(mask)
            host.copy_(lab, non_blocking=True)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        med = sorted(ms[1:])[len(ms[1:]) // 2]
        return {"engine_batch": int(lo), "workspace_bytes": size_of(lo), "fixed_bytes": fixed, "ms_per_image": round(med, 3), "value_10gb": round(mpix / med * 1e3, 3),
                "budget_bytes": budget, "note": "largest engine batch whose window workspace + image + mask + label map stay under 10 GiB "
                                                "(the reference's stated inference envelope, README.md:127); per-image step as the headline"}

    def measure_reference_scope(model, reps=3):
        """The reference's own timer scope (baseline/predict.py:106-116), one image at a time as its loop does: the uint8 HWC image on
        the host -> `/ max`, float tensor on the device -> sliding-window inference -> softmax / threshold / post-process -> labels
        on the host -> `<name>_label.tiff` (int32, deflate) written to disk."""
        import tempfile
        from cellseg_amd.preprocess import write_label_tiff
        imgs = slide_pair(False)
        pre = [(im.permute(1, 2, 0) * 255).round().to(torch.uint8).cpu().pin_memory() for im in imgs]      # pre_img_data, uint8 HWC
        ms = []
        with tempfile.TemporaryDirectory() as td:
            for i in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                u8 = pre[i % 2].to(dev, non_blocking=True)
                x = (u8.float() / u8.max()).permute(2, 0, 1)                   # [3,H,W] view of the HWC tensor, read in place
                _, _, mask = ca.swi_infer_image(model, x, roi, args.overlap, engine_batch=args.engine_batch or None, want_logits=False,
                                                want_mask=True, streams=args.streams)
                lab = ca.mask_to_instances(mask)
                write_label_tiff(os.path.join(td, f"slide{i}_label.tiff"), lab)        # deflated on the device, strips to the host
                ms.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ms[1:])
        med = ms[len(ms) // 2]
        return {"value_incl_tiff": round(mpix / med * 1e3, 3), "ms_per_image": round(med, 3), "min_ms": round(ms[0], 3), "max_ms": round(ms[-1], 3),
                "scope": "baseline/predict.py:106-116: uint8 HWC image on the host -> /max -> device -> sliding-window inference -> softmax, "
                         "threshold, post-process -> int32 deflate TIFF on disk (label map deflated on the device, compressed strips to the host), one image at a time"}

    def measure_strips_sim(model, full_s, sim_world=8):
        """How long ONE rank of the 8-GPU strips leg works per slide, measured on this GPU (see step_sim): rank 0 (sends only), an
        interior rank (receives and sends) and the last rank (the share with one window more); every rank labels every 8th slide.  The
        job runs at the slowest of them."""
        _, _, full_mask = ca.swi_infer_image(model, slide_pair(True)[0], roi, args.overlap, engine_batch=args.engine_batch or None,
                                             want_logits=False, want_mask=True, streams=args.streams)
        parts = [ca.share_partition(plan, r, sim_world) for r in range(sim_world)]
        n_sim = -(-args.steps // sim_world) * sim_world              # whole rounds of the labelling turn
        ranks = {}
        labellers = [ca.StripLabeller(args.size, args.size, r, sim_world, dev) for r in range(sim_world)]
        all_recs = torch.stack([lb.local(full_mask) for lb in labellers])
        all_ids = torch.stack([lb.resolve(all_recs) for lb in labellers])
        for sim_rank in sorted({0, min(3, sim_world - 1), sim_world - 1}):
            step = make_step(model, True, False, sim=(sim_rank, sim_world, full_mask))
            for _ in range(2):
                step()
            step.it[0] = 0
            dt_sim, _ = time_steps(step, n_sim, barrier)
            ranks[str(sim_rank)] = {"windows": parts[sim_rank].win_end - parts[sim_rank].win_begin, "labels_every": sim_world,
                                    "ms_per_slide": round(dt_sim / n_sim * 1e3, 3)}
            # the same share with --cc strips: the rank labels its 1/8 of the rows (+ halo) of EVERY slide instead of a whole mask every 8th
            step = make_step(model, True, False, sim=(sim_rank, sim_world, full_mask, (labellers[sim_rank], all_recs, all_ids)))
            for _ in range(2):
                step()
            step.it[0] = 0
            dt_sim, _ = time_steps(step, n_sim, barrier)
            ranks[str(sim_rank)]["ms_per_slide_cc_strips"] = round(dt_sim / n_sim * 1e3, 3)
        worst = max(v["ms_per_slide"] for v in ranks.values())
        worst_cc = max(v["ms_per_slide_cc_strips"] for v in ranks.values())
        return {"of": sim_world, "ranks": ranks, "windows_per_rank": [p.win_end - p.win_begin for p in parts],
                "max_windows_per_rank": max(p.win_end - p.win_begin for p in parts), "steps": n_sim,
                "ms_per_slide": worst, "full_slide_ms": round(full_s * 1e3, 3), "ratio": round(full_s * 1e3 / worst, 3),
                "ms_per_slide_cc_strips": worst_cc, "ratio_cc_strips": round(full_s * 1e3 / worst_cc, 3),
                "note": "one GPU doing one rank's share of the 8-rank strips leg (window-list partition: its 120 / 121 windows, the exchange's pack / "
                        "unpack copies, the finalize of its rectangles, and on its turn - every 8th slide - the CC pass over a full mask on the CC "
                        "stream); ms_per_slide = the slowest of the simulated ranks; the seam transfer (12 MB over one xGMI link) and the mask "
                        "all-gather (16 MB gathered) are stood in for by device-local copies of the same sizes, so `ratio` is an UPPER bound of "
                        "the 8-GPU speed-up: link latency, RCCL launch cost and rank skew are not in it; *_cc_strips = the same with the strip-local "
                        "connected components (--cc strips: every rank labels its rows + 80-row halo of every slide, seam union-find, device-local "
                        "copies in place of the two edge all-gathers and the label-strip gather)"}

    def measure_tiles(width, precision, n, k=20):
        """SURVEY.md 8(d) 'tile throughput': forward-only U-Net on N x 3 x 256 x 256 synthetic tiles (the predictor call of
        model_training_3class.py:255 / :291), logits left in HBM; Mpixels/s of tile pixels + the dominant kernel's roofline."""
        model = make_model(WIDTHS[width], seed).to(dev)
        model.precision = precision
        x = torch.rand(n, 3, 256, 256, generator=torch.Generator().manual_seed(seed)).to(dev)
        with torch.no_grad():
            for _ in range(3):
                model(x)
            dt, _ = time_steps(lambda: model(x), k, barrier)
            _lib.lib.cs_prof_reset()
            _lib.lib.cs_prof_enable(1)
            time_steps(lambda: model(x), k, barrier)
            _lib.lib.cs_prof_enable(0)
            prof = _lib.prof_collect()
        ms = dt / k * 1e3
        return {"value": round(n * 65536 / 1e6 / (dt / k), 1), "unit": "Mpixels/s (tile pixels, forward only)", "ms_per_forward": round(ms, 4),
                "tiles": n, "tflops": round(n * WINDOW_GFLOP[width] / ms, 2), "roofline": roofline({"prof": prof}, precision)}

    def measure_configs4():
        """BASELINE.json configs[4] on ONE GPU as a child process of this bench (its own slide size, window and net): 64...1024 U-Net,
        512 x 512 windows, fp8 weights / e4m3-staged activations, one 8192 x 8192 slide, 3 timed steps.  The child's full record is read
        from the file it writes (BENCH_FULL_NAME)."""
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", "fp8", "--width", "deep64", "--roi", "512", "--size", "8192",
               "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"]
        name = f"bench_full_configs4_{os.getpid()}.json"
        path = os.path.join(ROOT, "gpurun_out", name)
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DIST")}
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, BENCH_FULL_NAME=name))
            line = json.loads(cp.stdout.strip().splitlines()[-1])
            full = json.load(open(path)) if os.path.exists(path) else line
        except (subprocess.SubprocessError, ValueError, IndexError, OSError) as e:
            return {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            if os.path.exists(path):
                os.remove(path)
        return {k: full[k] for k in ("metric", "value", "unit", "ms_per_step", "value_pipelined", "ms_per_step_pipelined", "steps", "warmup", "dtype", "value_device",
                                     "roofline", "roofline_by_dtype", "frac_of_fp8_peak_whole_slide", "bench_seconds") if k in full} | {
            "workload": full["config"]["workload"], "windows": full["config"]["windows"], "workspace_bytes": full["config"].get("workspace_bytes"),
            "kernels": dict(list(full.get("kernels", {}).items())[:12]),
            "note": "python bench.py --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1 as a child process on the same GPU"}

    def infer_summary(res, strips):
        slides = 1 if strips else world
        return {"value": round(slides * mpix * args.steps / res["dt"], 3), "ms_per_step": round(res["dt"] / args.steps * 1e3, 3),
                "value_pipelined": round(slides * mpix * args.steps / res["dt_pipe"], 3), "ms_per_step_pipelined": round(res["dt_pipe"] / args.steps * 1e3, 3),
                "value_device": round(slides * mpix * args.steps / res["dt_dev"], 3), "instances_found": res["n_instances"]}

    # ------------------------------------------------------------------------------------------------------------------
    # training leg (BASELINE.json configs[1] at N=1, configs[3] at N=8): the reference loop, model_training_3class.py:247-267
    # ------------------------------------------------------------------------------------------------------------------
    def measure_train(width, with_prof=True, precision="fp32", batch=None, eager_too=True):
        torch.manual_seed(seed)                                   # identical replicas on every rank
        model = ca.UNet(channels=WIDTHS[width], strides=(2, 2, 2, 2), num_res_units=2).to(dev)
        model.precision = precision                               # matrix arithmetic of forward and backward (bf16: bf16 activations on the tape too)
        g = torch.Generator(device="cpu").manual_seed(seed + 1 + rank)      # every rank its own shard of the global batch
        b = batch or args.train_batch
        x = torch.rand(b, 3, 256, 256, generator=g).to(dev)
        y = torch.randint(0, 3, (b, 1, 256, 256), generator=g).to(dev)
        onehot = ca.one_hot(y, 3)
        loss_fn, opt = ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4)      # torch.optim.AdamW's rule, one launch for the 63 tensors
        model.train()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        k = args.train_steps

        def eager_step():
            """The loop body as the reference writes it (model_training_3class.py:250-262), through autograd, launch by launch."""
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(model(x), onehot)
            loss.backward()
            if dist is not None:                              # also at world size 1 under BENCH_FORCE_DIST: the RCCL all-reduce really runs
                ev[0].record()
                parallel.sync_gradients(model)
                ev[1].record()
            opt.step()
            return loss

        eager_ms = None
        if eager_too:
            for _ in range(max(args.warmup, 2)):
                eager_step()
            dt_e, _ = time_steps(eager_step, k, barrier)
            (dt_e,) = max_over_ranks([dt_e])
            eager_ms = dt_e / k * 1e3
        allreduce_ms = None
        if dist is not None and eager_too:
            torch.cuda.synchronize()
            allreduce_ms = round(ev[0].elapsed_time(ev[1]), 4)   # the last eager step's exchange (events on the launch stream)
        # the same body as one lean launch sequence on static buffers (ca.TrainStep), and replayed as a captured hipGraph
        graph_ms = None
        if eager_too:
            captured = ca.TrainStep(model, loss_fn, opt, capture=True)
            for _ in range(max(args.warmup, 3)):
                captured(x, onehot)
            dt_g, _ = time_steps(lambda: captured(x, onehot), k, barrier)
            (dt_g,) = max_over_ranks([dt_g])
            graph_ms, graph_nodes = dt_g / k * 1e3, captured.graph_nodes()
            del captured
        lean = ca.TrainStep(model, loss_fn, opt)

        def step():
            return lean(x, onehot)

        for _ in range(max(args.warmup, 3)):
            step()
        dt, loss = time_steps(step, k, barrier)
        prof = {}
        if with_prof:                                          # per-kernel HIP events: eager launches on one stream (a capture holds no events)
            _lib.lib.cs_prof_reset()
            _lib.lib.cs_prof_enable(1)
            time_steps(step, k, barrier)
            _lib.lib.cs_prof_enable(0)
            prof = _lib.prof_collect()
        (dt,) = max_over_ranks([dt])
        ms = dt / k * 1e3
        tot = sum(v["ms"] for v in prof.values()) or 1.0
        out = {"images_per_s": round(world * b * k / dt, 1), "ms_per_step": round(ms, 3),
               "eager_ms_per_step": round(eager_ms, 3) if eager_ms else None,
               "graph_ms_per_step": round(graph_ms, 3) if graph_ms else None, "graph_nodes": graph_nodes if graph_ms else None,
               "tflops": round(world * 3 * b * WINDOW_GFLOP[width] / ms, 2),
               "frac_of_fp32_mfma_peak": round(3 * b * WINDOW_GFLOP[width] / ms / PEAK_F32_MFMA_TFLOPS, 4),
               "global_batch": world * b, "per_gpu_batch": b, "width": list(WIDTHS[width]),
               "dtype": {"fp32": "f32", "fp16x3": "f32-grade split fp16 (2 planes x 3 terms): forward convolutions and 3x3 weight gradients on "
                                                   "v_mfma_f32_16x16x32_f16, data gradients f32",
                         "bf16": "bf16 forward convolutions, weight gradients and data gradients (v_mfma_f32_16x16x32_bf16; bf16 activations on the tape - the forward's "
                                 "bf16 tensors are what the norm-backward and weight-gradient kernels read -, fp32 gradients, norm / loss / optimiser arithmetic and master weights)"
                         }.get(precision, f"{precision} forward convolutions (fp32 activations), f32 backward"),
               "loss": round(float(loss.item()), 5), "grad_bucket_bytes": int(model.flat_grad.numel() * 4),
               "allreduce_ms": allreduce_ms,
               "roofline": roofline({"prof": prof}, precision if precision in MFMA_PEAK else "fp32") if prof else None,
               "engine_gpu_ms_per_step": round(tot / k, 3) if prof else None,
               "kernels": {n: {"ms_per_step": round(v["ms"] / k, 4), "share": round(v["ms"] / tot, 4),
                               "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] else None}
                           for n, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:10]},
               "note": "forward + DiceCE(softmax) + backward + flat-bucket all-reduce (when a process group exists) + AdamW + weight re-pack, all "
                       "through the C-ABI; ms_per_step = cellseg_amd.TrainStep (the launches on static buffers, frozen re-pack table), "
                       "graph_ms_per_step = the same sequence replayed as a captured hipGraph (TrainStep(capture=True)), eager_ms_per_step = "
                       "the body through autograd as the reference writes it; FLOPs = 3 x forward conv FLOPs (SURVEY 8(d))"}
        return out

    def roofline(res, precision=None, only=None):
        """benchlib.roofline.dominant_kernel on this leg's HIP-event table + (primary leg of the default workload) the HBM bytes per launch
        of that group: measured by THIS run where it can be (the default command on one GPU: two child passes of rocprofv3 --pmc FETCH_SIZE /
        WRITE_SIZE, gfx950 correction applied - benchlib.traffic), else quoted from the committed passes of the same command
        (profiles/pmc_traffic.json, tools/pmc_summary.py) and labelled as a quote."""
        rl = dominant_kernel(res.get("prof"), precision or args.precision, only)
        if rl is None:
            return None
        if res.get("primary") and args.engine_batch == 0 and args.size == 4096 and args.width == "baseline32" and args.precision == "fp32":
            live = None
            if world == 1 and dist is None and not args.no_secondary and not args.no_live_traffic:
                with leg("live_traffic"):
                    live, detail = live_pmc_traffic(rl["kernel"])
                rl["traffic_live"] = detail
            if live is not None:
                rl["traffic"] = live
                rl["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of `bench.py --steps 1 --warmup 0 "
                                        "--streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary`, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch")
            else:
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(rl["kernel"], {})
                    rl["traffic"] = pmc.get("hbm_bytes_per_launch")
                    rl["traffic_source"] = "QUOTED from profiles/pmc_traffic.json (committed rocprofv3 PMC passes of the default command), not measured by this run"
                except (OSError, ValueError, KeyError):
                    pass
        return rl

    def kernel_table(res):
        return kernel_shares(res["prof"], args.steps)

    def base_line(main_res, head_strips):
        """The full record's contract keys for one measured inference leg (slides or strips)."""
        main_res["primary"] = True
        rccl_ranks = dist.get_world_size() if dist else 1
        slides = 1 if head_strips else world               # slides segmented per step by the whole job
        ms_step = main_res["dt"] / args.steps * 1e3
        out = {
            "metric": f"Mpixels/s sliding-window inference ({args.size}^2 WSI) incl. GPU connected components, label map on host; per image: every "
                      f"timed step ends with its labels on the host (the two-slide software pipeline is value_pipelined)",
            "value": round(slides * mpix * args.steps / main_res["dt"], 3),
            "unit": "Mpixels/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "collective_backend": backend if dist else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong" if head_strips else "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {"workload": f"sliding-window inference on 3x{args.size}x{args.size} synthetic WSI, {args.roi} window / "
                                   f"{plan.interval_h} stride ({n_windows} windows), 3-class residual U-Net channels={list(WIDTHS[args.width])} "
                                   f"num_res_units=2 InstanceNorm+PReLU random-init, softmax + >0.5 + remove_small_holes(64) + "
                                   f"remove_small_objects(16) + 8-conn label on GPU, int32 labels copied to pinned host memory; "
                                   + ("one slide cut by its window list over the ranks" if head_strips else "one slide per GPU"),
                       "windows": n_windows, "engine_batch": args.engine_batch or ca.balanced_batch(n_windows, args.streams), "engine_batch_auto": not args.engine_batch,
                       "streams": args.streams, "precision": args.precision,
                       "parallelism": (f"1 slide split by its window list over {world} ranks + seam send/recv + mask all-gather, labelling rank round-robin" if head_strips
                                       else f"slides x{world}"),
                       "workspace_bytes": main_res.get("workspace_bytes"), "peak_allocated_bytes": main_res.get("peak_allocated_bytes"),
                       "instances_found": main_res["n_instances"], "foreground_fraction": round(main_res["fg_frac"], 4)},
            "latency_ms": main_res["latency"]["latency_ms"],
            "latency": dict(main_res["latency"], note="the timed steps one by one (median / min / max): image tensor resident in HBM -> int32 labels in pinned "
                                                      "host memory, synchronised, no overlap with a neighbour slide (SURVEY 8(d))"),
            "value_pipelined": round(slides * mpix * args.steps / main_res["dt_pipe"], 3),
            "ms_per_step_pipelined": round(main_res["dt_pipe"] / args.steps * 1e3, 3),
            "value_pipelined_note": "the same K steps without the per-step synchronisation: label copy and connected components of slide k run on their own "
                                    "streams beside the windows of slide k + 1 (rounds 1-5 quoted this as `value`)",
            "value_device": round(slides * mpix * args.steps / main_res["dt_dev"], 3),
            "value_device_note": "pipelined loop ending when the int32 label map is complete in HBM (no copy to the host)",
            "value_with_events": round(slides * mpix * args.steps / main_res["dt_prof"], 3) if main_res["dt_prof"] else None,
            "roofline": roofline(main_res),
            "kernels": kernel_table(main_res),
        }
        if main_res.get("phases"):
            out.update({k: main_res["phases"][k] for k in ("exchange_ms", "gather_ms", "cc_ms")}, strips_phases=main_res["phases"])
        if args.precision == "fp8":
            # mixed mode: the e4m3 launches against the fp8 peak, the bf16 launches of the narrow layers against the bf16 peak - one record per group
            out["roofline_by_dtype"] = {"fp8": roofline(main_res, only=lambda n: n.endswith("_fp8")),
                                        "bf16": roofline(main_res, only=lambda n: not n.endswith("_fp8") and "wino" not in n and "mfma" in n)}
            flops = n_windows * WINDOW_GFLOP.get(args.width, 0.0) * (args.roi / 256) ** 2 * 1e9
            out["frac_of_fp8_peak_whole_slide"] = round(flops / (main_res["dt_pipe"] / args.steps) / 1e12 / PEAK_FP8_MFMA_TFLOPS, 4) if flops else None
        return out

    def finish(out):
        out["leg_seconds"] = leg_seconds
        out["bench_seconds"] = round(time.time() - t_bench0, 1)
        benchline.emit(out, ROOT)
        if dist:
            dist.destroy_process_group()

    # ------------------------------------------------------------------------------------------------------------------
    head_strips = args.shard == "strips" and dist is not None
    single = world == 1 and dist is None
    secondary = not args.no_secondary
    main_res = sec_res = slides_res = None
    modes, other_part, train_res = {}, None, None
    strips_equal = None
    train_ref = None
    strips_error = None
    if args.mode == "infer":
        if head_strips:
            # the strips leg is the only one with data-path collectives (seam send/recv, mask all-gather).  The collective-free slides
            # leg runs FIRST and is kept: should the strips leg fail on this node (it has never run on real multi-GPU hardware here), the
            # job still prints the slides figure - marked as such, and the job exits non-zero - instead of ending without a line
            with leg("slides"):
                slides_res = measure_infer(args.width, args.precision, False, True, extras=secondary and rank == 0)

            def fallback_line(msg):
                full = base_line(slides_res, False)
                full["strips_error"] = msg
                full["config"]["note"] = "the strong-scaling strips leg FAILED on this node (strips_error; exit code 3); this line is the collective-free slides leg"
                full["leg_seconds"] = leg_seconds
                return benchline.compact_line(full, benchline.write_full(full, ROOT))

            # A failure inside the strips leg need not be symmetric: a rank that raises leaves its peers inside the seam send/recv or the
            # mask all-gather, and any later collective then mismatches until the process-group timeout aborts the job without a line.  So
            # from here to the end of the leg every rank runs a watcher thread: a rank that fails posts the reason in the process group's
            # store, every watcher sees it within half a second (or the leg's deadline passes), rank 0 prints the slides line with
            # `strips_error` set and every rank leaves at once with exit code 3 - no collective is entered after a failure.
            guard = StripsGuard(dist, rank, fallback_line, float(os.environ.get("BENCH_STRIPS_DEADLINE_S", "150")))
            try:
                ensure_seam_transport()
                with leg("strips"):
                    main_res = measure_infer(args.width, args.precision, True, True)      # ends with a collective: every rank got this far
            except Exception as e:                                  # noqa: BLE001
                guard.fail(f"{type(e).__name__}: {e}"[:500])        # does not return
            guard.disarm()
            if secondary or os.environ.get("BENCH_COMPARE_LEGS") == "1":
                other_part = infer_summary(slides_res, False)
                strips_equal = bool(torch.equal(slides_res["labels"], main_res["labels"])) if rank == 0 else None
        else:
            with leg("headline"):
                main_res = measure_infer(args.width, args.precision, False, True, extras=secondary and rank == 0)
        if secondary:
            if dist is not None and not head_strips and strips_error is None:
                # the secondary strips leg is guarded like the headline one: a rank that fails in it must not leave its peers in a collective
                ensure_seam_transport()

                def slides_line(msg):
                    full = base_line(main_res, False)
                    full["strips_error"] = msg
                    return benchline.compact_line(full, benchline.write_full(full, ROOT))

                guard = StripsGuard(dist, rank, slides_line, float(os.environ.get("BENCH_STRIPS_DEADLINE_S", "150")))
                try:
                    with leg("strips"):
                        other = measure_infer(args.width, args.precision, True, False)
                except Exception as e:                              # noqa: BLE001
                    guard.fail(f"{type(e).__name__}: {e}"[:500])
                guard.disarm()
                other_part = dict(infer_summary(other, True), **(other.get("phases") or {}))
                # rank 0's slides-leg slide pair is the strips leg's pair: the two partitions must produce the same label map
                strips_equal = bool(torch.equal(other["labels"], main_res["labels"])) if rank == 0 else None
            if single and args.width != "reference16":
                with leg("reference_width"):
                    sec_res = measure_infer("reference16", args.precision, False, True)
            if single and args.precision == "fp32":
                for prec in ("fp16x3", "bf16x6", "bf16", "fp8") if args.full else ("bf16", "fp8"):
                    with leg(f"slide_{prec}"):
                        r = measure_infer(args.width, prec, False, False)
                    modes[prec] = dict(infer_summary(r, False), foreground_agreement_vs_primary_random_init=round(
                        float(((r["labels"] > 0) == (main_res["labels"] > 0)).float().mean().item()), 6))
                if args.full:
                    # ... and what the modes do to the RESULT, on a net whose logits are decisive (trained on synthetic cells)
                    with leg("accuracy_32base"):
                        acc = lowprec_accuracy(dev)
                    modes["accuracy_on_trained_net"] = acc.pop("trained_net")
                    for prec, a in acc.items():
                        modes[prec].update(a)
            tw = args.width if args.width in WINDOW_GFLOP and args.roi == 256 else "baseline32"
            with leg("train_fp32"):
                train_res = measure_train(tw)
            if single:                                            # configs[1]: the same step with the convolutions on the 16-bit pipe
                for prec in ("fp16x3", "bf16") if args.full else ("bf16",):
                    with leg(f"train_{prec}"):
                        t16 = measure_train(tw, with_prof=False, precision=prec, eager_too=False)
                    train_res[prec] = {k: t16[k] for k in ("images_per_s", "ms_per_step", "tflops", "loss", "dtype")}
                # the reference's own training configuration (model_training_3class.py:201-208, --batch_size 8): 16...256 channels, batch 8
                with leg("train_reference"):
                    tref = measure_train("reference16", with_prof=False, batch=8)
                    train_ref = {k: tref[k] for k in ("images_per_s", "ms_per_step", "eager_ms_per_step", "graph_ms_per_step", "graph_nodes", "tflops", "loss",
                                                      "per_gpu_batch", "width")}
                    tref16 = measure_train("reference16", with_prof=False, batch=8, precision="bf16", eager_too=False)      # the same step in the bf16 mode
                    train_ref["bf16"] = {k: tref16[k] for k in ("images_per_s", "ms_per_step", "tflops", "loss", "dtype")}
    else:
        with leg("train"):
            train_res = measure_train(args.width, precision=args.precision if args.precision in ("fp32", "bf16", "fp16x3", "bf16x6") else "fp32")

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    rccl_ranks = dist.get_world_size() if dist else 1
    if args.mode == "train":
        out = {
            "metric": "images/s data-parallel training step (3x256x256 tiles, batch 32 per GPU)",
            "value": train_res["images_per_s"], "unit": "images/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "collective_backend": backend if dist else None,
            "steps": args.train_steps, "warmup": max(args.warmup, 3), "ms_per_step": train_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": train_res["dtype"] if args.precision != "fp32" else "f32", "data": "synthetic",
            "config": {"workload": f"training step on {args.train_batch}x3x256x256 synthetic tiles per GPU, 3-class residual U-Net channels="
                                   f"{list(WIDTHS[args.width])} num_res_units=2 InstanceNorm+PReLU random-init, DiceCE(softmax) + AdamW(6e-4), "
                                   f"one flat fp32 gradient bucket ({train_res['grad_bucket_bytes']} B) all-reduced per step",
                       "global_batch": world * args.train_batch, "parallelism": f"dp{world}", "precision": args.precision},
            "roofline": train_res.pop("roofline", None),
            "allreduce_ms": train_res.get("allreduce_ms"),
            "train_step": train_res,
        }
        return finish(out)

    out = base_line(main_res, head_strips)
    strips_info = {"seam_transport": seam["transport"], "strips_error": strips_error, "cc": args.cc,
                   "partition": "window list (cs_swi_partition_windows): contiguous shares that differ by at most one window",
                   "windows_per_rank": [sh.win_end - sh.win_begin for sh in share_parts],
                   "output_pixels_per_rank": [sum((r1 - r0) * (c1 - c0) for r0, r1, c0, c1 in sh.rects()) for sh in share_parts],
                   "labelling": ("round-robin over the ranks: slide k's gathered mask is labelled whole by rank k mod N" if args.cc == "turns" else
                                 "strip-local: every rank labels its rows of every slide, seam union-find, label strips gathered to rank k mod N"),
                   "strip_batch": args.strip_batch or "auto",
                   "equal_to_single_gpu_labels": strips_equal}
    if other_part:
        out["strips" if not head_strips else "slides"] = dict(
            other_part, scaling="strong" if not head_strips else "weak",
            note=("ONE slide split by its window list across the ranks (shares differ by one window at most; no window evaluated twice), window parts sent to the later neighbour, "
                  "finalised uint8 rectangles all-gathered, CC + label copy on rank (slide index mod N) on its CC stream: value = one slide's pixels / time"
                  if not head_strips else "one slide per rank, no collective: value = N slides' pixels / time"))
        out["strips" if not head_strips else "slides"].update(strips_info if not head_strips else {})
        out["value_slides"] = out["value"] if not head_strips else other_part["value"]
        out["value_strips"] = other_part["value"] if not head_strips else out["value"]
        if not head_strips:
            out.update({k: other_part[k] for k in ("exchange_ms", "gather_ms", "cc_ms") if k in other_part})
    if dist is not None:
        out["config"].update(strips_info)
    extras_res = slides_res if head_strips else main_res          # reference scope / simulation / envelope are per-image figures of the slides leg
    if extras_res.get("incl_tiff"):
        out.update(value_incl_tiff=extras_res["incl_tiff"]["value_incl_tiff"], reference_timer_scope=extras_res["incl_tiff"])
    if extras_res.get("strips_sim"):
        out["strips_sim"] = extras_res["strips_sim"]
    if extras_res.get("envelope"):
        env = extras_res["envelope"]
        out.update(value_10gb=env["value_10gb"], memory_envelope=env)
        out["config"].update(engine_batch_10gb=env["engine_batch"], workspace_bytes_10gb=env["workspace_bytes"])
    if sec_res:
        out["reference_width"] = dict(infer_summary(sec_res, False), channels=list(WIDTHS["reference16"]), roofline=roofline(sec_res))
    if modes:
        out["other_precisions"] = {"note": "same workload, convolutions on the 16-bit matrix pipe (v_mfma_f32_16x16x32_{f16,bf16}, fp32 "
                                           "accumulate): fp16x3 = 2 fp16 planes (22 mantissa bits) x 3 terms and bf16x6 = 3 bf16 planes x "
                                           "6 terms are fp32-grade (logits within 1e-4 of the oracle in the GPU tests; --full); bf16 = the bf16 datapath "
                                           "(bf16 weights, bf16 activations in HBM); fp8 = e4m3 weights with per-channel scales, e4m3-staged "
                                           "activations, v_mfma_scale_f32_16x16x128_f8f6f4 (built for the 64...1024 net of configs[4]: at this "
                                           "width most layers fill a fraction of its K = 128).  --full adds f1_vs_fp32 / pixel_agreement / delta_f1_vs_truth: "
                                           "instance F1@0.5 and interior-mask agreement against the fp32 path, and the change of the F1 against "
                                           "the synthetic ground truth, on the synthetic-trained net described under accuracy_on_trained_net",
                                   **modes}
    if train_res:
        out["train_step"] = train_res
    if train_ref:
        out["train_step_reference"] = dict(train_ref, note="the reference's own training setting: channels 16...256, batch 8 x 3 x 256 x 256 "
                                                            "(baseline/model_training_3class.py:201-208, 247-267)")
    if single and secondary and args.width in WINDOW_GFLOP and args.roi == 256:
        # forward-only tile throughput (N = 8: configs[0]'s batch, N = 32: configs[1]'s), exact fp32 and the bf16 datapath
        with leg("tiles"):
            out["tiles"] = {prec: {f"n{n}": measure_tiles(args.width, prec, n) for n in (8, 32)} for prec in ("fp32", "bf16")}
        out["tiles"]["note"] = ("forward-only U-Net on N x 3 x 256 x 256 synthetic tiles (torch.rand, seed 2022), logits left in HBM; value = tile "
                                "pixels / time; roofline = the dominant kernel of that forward (HIP events on the launch stream)")
        if args.precision == "fp32" and args.size == 4096:
            with leg("configs4_fp8_child"):
                out["configs4_fp8"] = measure_configs4()
            if args.full:
                try:      # ... and its accuracy claim at its OWN width and window: the 64...1024 net trained on synthetic cells, 512^2 windows
                    with leg("accuracy_configs4"):
                        acc4 = lowprec_accuracy(dev, n_images=16, size=2048, precisions=("bf16", "fp8"), width="deep64", roi=512)      # ~31 000 cells: 1e-3 = 31 cells
                    out["configs4_fp8"]["accuracy_on_trained_net"] = acc4
                except Exception as e:                              # noqa: BLE001
                    out["configs4_fp8"]["accuracy_on_trained_net"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if single and secondary and args.full and args.precision == "fp32" and "tiles" in out and train_res and modes:
        try:
            engine_ms = {"tiles_fp32": out["tiles"]["fp32"]["n32"]["ms_per_forward"], "tiles_bf16": out["tiles"]["bf16"]["n32"]["ms_per_forward"],
                         "train_fp32": train_res["ms_per_step"], "train_bf16": train_res.get("bf16", {}).get("ms_per_step"),
                         "slide_fp32": out["ms_per_step"], "slide_bf16": modes["bf16"]["ms_per_step"]}
            with leg("eager_rocm"):
                out["eager_rocm"] = eager_rocm(dev, args.size, args.roi, args.overlap, args.width, seed, engine_ms)
        except Exception as e:                                      # noqa: BLE001  (a comparator must not cost the line)
            out["eager_rocm"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and not args.no_cpu_baseline:
        with leg("cpu_baseline"):
            out["cpu_baseline"] = cpu_baseline(args.size, args.roi, args.overlap, WIDTHS[args.width], seed)
        if secondary and args.full:
            with leg("cpu_tiles_train"):
                out["cpu_baseline"].update(cpu_baseline_tiles_and_train(seed))
    finish(out)


if __name__ == "__main__":
    main()
