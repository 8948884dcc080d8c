#!/usr/bin/env python3
"""bench.py - headline benchmark: Mpixels/s of whole-slide sliding-window inference on MI355X.

One "step" = the reference's per-image hot path (baseline/predict.py:107-113) on one synthetic 3x4096x4096
slide that is already resident in HBM: sliding-window U-Net inference (256 window / 128 stride = BASELINE.json
configs[2], 961 windows) -> softmax -> class-1 > 0.5 -> remove_small_holes -> remove_small_objects -> 8-connected
label -> int32 label map copied to (pinned) host memory.  SURVEY.md 8(d) ends the metric at "label map available on
host", so `value` includes that copy; `value_device` is the same loop stopping when the labels are complete in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one rank per GPU, RCCL)

`--gpus N` with N > 1 and no torchrun environment: this process starts N child ranks itself (before it has touched the
GPU) and forwards rank 0's JSON line.

Multi-GPU legs:
  slides  (weak scaling)             every rank segments its own slide, no data-path collective
  strips  (headline for N > 1,       ONE slide split by its WINDOW LIST (shares differ by one window at most; no window evaluated twice),
           strong scaling)           window parts sent to the later neighbour (RCCL send/recv), all-gather (RCCL) of the finalised uint8
                                     rectangles, CC + label copy on rank (slide index mod N) on its own stream
  train   (`train_step` object; `--mode train` makes it the headline) data-parallel training step, batch 32x3x256x256 per
          GPU (BASELINE.json configs[1]/[3]): forward + DiceCE + backward + ONE flat-bucket RCCL all-reduce + AdamW
Prints ONE JSON line on rank 0 (contract in the task description), including `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the torch-CPU / C oracle on a bounded sample, host cores).
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

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA peak; a split-bf16 product costs 6 / 3 / 1 bf16 MFMA terms
PEAK_FP8_MFMA_TFLOPS = 5000.0    # dense fp8 MFMA peak
MFMA_PEAK = {"fp32": PEAK_F32_MFMA_TFLOPS, "fp16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3,
             "bf16": PEAK_BF16_MFMA_TFLOPS, "fp8": PEAK_FP8_MFMA_TFLOPS}
DTYPE = {"fp32": "f32", "fp16x3": "f32 (split-fp16, 2 planes x 3 MFMA terms, fp32 accumulate)", "bf16x6": "f32 (split-bf16, 3 planes x 6 MFMA terms, fp32 accumulate)",
         "bf16x3": "split-bf16 (2 planes x 3 terms)", "bf16": "bf16 (weights, activations in HBM; fp32 accumulate)",
         "fp8": "fp8 e4m3 for the layers with >= 128 channels (weights with per-channel scales; their inputs materialised once as activated e4m3, "
                "v_mfma_scale_f32_16x16x128_f8f6f4), bf16 for the narrower ones; conv outputs bf16; fp32 accumulate"}
PEAK_HBM_GBS = 8000.0
WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256),
          "deep64": (64, 128, 256, 512, 1024)}          # BASELINE.json configs[4] width (use with --roi 512)
# conv FLOPs of one forward over one 256^2 window (training step ~ 3x forward)
WINDOW_GFLOP = {"reference16": 1.556, "baseline32": 6.080, "deep64": 24.06}


def synthetic_wsi(n: int, seed: int) -> np.ndarray:
    """Gaussian-filtered (sigma 6) uniform noise, min-max scaled, 3 channels + small noise."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.random((n, n), dtype=np.float32), 6)
    g = (g - g.min()) / (g.max() - g.min())
    x = np.stack([g, g, g], 0) + rng.normal(0, 0.02, (3, n, n)).astype(np.float32)
    return np.clip(x, 0, 1).astype(np.float32)


def make_model(channels, seed=2022):
    import cellseg_amd as ca
    torch.manual_seed(seed)
    return ca.UNet(spatial_dims=2, in_channels=3, out_channels=3, channels=channels, strides=(2, 2, 2, 2), num_res_units=2).eval()


def time_steps(fn, steps, barrier):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0, out


def effective_cpus() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box reports 256
    logical CPUs under a 16-CPU quota; 256 OpenMP threads on 16 CPUs stall for minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


CONDA_PY = "/opt/conda/bin/python3.9"        # the image's second interpreter: real scikit-image 0.18.3 (no torch)


def skimage_chain_seconds(mask_u8: np.ndarray):
    """The reference's own post-process (predict.py:113) run with REAL scikit-image under the image's conda interpreter,
    when it is present; returns seconds or None.  (BASELINE.md 3(iii) prescribes this chain for the CPU leg.)"""
    if not os.path.exists(CONDA_PY):
        return None
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "mask.npy")
        np.save(path, mask_u8)
        try:
            out = subprocess.run([CONDA_PY, os.path.join(ROOT, "oracle", "skimage_chain.py"), path], capture_output=True, text=True,
                                 timeout=300, env={k: v for k, v in os.environ.items() if not k.startswith("PYTHON")})
            return float(json.loads(out.stdout.strip().splitlines()[-1])["seconds"]) if out.returncode == 0 else None
        except (subprocess.SubprocessError, ValueError, KeyError, IndexError, OSError):
            return None


def cpu_baseline(size, roi, overlap, channels, seed):
    """Oracle (torch-CPU restatement + plain-C CC) on a bounded sample of the same workload, extrapolated; the real
    scikit-image chain is timed beside the C oracle when the conda interpreter exists."""
    from oracle.postprocess_ref import postprocess_c, softmax_interior_ref
    from oracle.sliding_window_ref import sliding_window_inference_ref, window_grid
    from oracle.unet_ref import make_reference_unet
    threads = effective_cpus()
    torch.set_num_threads(threads)
    net = make_reference_unet(seed, channels=channels)
    interval = int(roi * (1 - overlap))
    crop = roi + 11 * interval                    # 12 x 12 = 144 windows
    crop = min(crop, size)
    img = torch.from_numpy(synthetic_wsi(size, seed)[:, :crop, :crop])[None]
    n_sample = len(window_grid(crop, crop, (roi, roi), overlap))
    n_full = len(window_grid(size, size, (roi, roi), overlap))
    t_reps = []
    with torch.no_grad():
        sliding_window_inference_ref(img[:, :, :roi, :roi], (roi, roi), 4, net, overlap)          # warm up
        while sum(t_reps) < 10.0 or len(t_reps) < 3:                                              # >= 10 s of windows, >= 3 repeats
            t0 = time.perf_counter()
            logits = sliding_window_inference_ref(img, (roi, roi), 4, net, overlap)
            t_reps.append(time.perf_counter() - t0)
            if len(t_reps) >= 24:
                break
    t_net = sorted(t_reps)[len(t_reps) // 2]                                                      # median repeat
    prob = softmax_interior_ref(logits[0].numpy())
    reps = int(np.ceil(size / crop))
    prob_full = np.tile(prob, (reps, reps))[:size, :size]                                       # full-size CC input
    t0 = time.perf_counter()
    postprocess_c(prob_full)
    t_cc = time.perf_counter() - t0
    t_sk = skimage_chain_seconds((prob_full > 0.5).astype(np.uint8))
    t_full = t_net * n_full / n_sample + t_cc
    spread = [round(size * size / 1e6 / (t * n_full / n_sample + t_cc), 4) for t in (max(t_reps), min(t_reps))]
    out = {"value": round(size * size / 1e6 / t_full, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
           "value_spread": spread,
           "sample": f"torch-CPU oracle on {n_sample} of {n_full} windows ({crop}x{crop} crop), {len(t_reps)} repeats = {sum(t_reps):.2f}s, median "
                     f"{t_net:.2f}s (value_spread = slowest / fastest repeat), {threads} threads, scaled to the slide + plain-C CC oracle on a "
                     f"full {size}x{size} mask ({t_cc:.2f}s, 1 thread)"}
    if t_sk is not None:
        out["value_skimage_chain"] = round(size * size / 1e6 / (t_net * n_full / n_sample + t_sk), 4)
        out["skimage_chain_s"] = round(t_sk, 3)
        out["sample"] += f"; value_skimage_chain replaces the C oracle by real scikit-image 0.18.3 remove_small_holes/objects + label on the same mask ({t_sk:.2f}s, {CONDA_PY})"
    return out


def _median_seconds(fn, min_s, min_reps=3, max_reps=12):
    ts = []
    while (sum(ts) < min_s or len(ts) < min_reps) and len(ts) < max_reps:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], len(ts), sum(ts)


def cpu_baseline_tiles_and_train(seed):
    """BASELINE.md 3 (i) and (iv) on the host cores beside the GPU legs: the oracle net's forward on N x 3 x 256 x 256 tiles (N = 4:
    the reference's sw_batch_size; N = 32: configs[1]) and one oracle training step (forward + DiceCE + backward + AdamW,
    oracle/train_ref.py) at batch 8, for the BASELINE.json width (32...512) and the reference's own (16...256)."""
    from oracle.train_ref import dice_ce_ref, one_hot_ref
    from oracle.unet_ref import make_reference_unet
    threads = effective_cpus()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    x32 = torch.rand(32, 3, 256, 256, generator=g)
    y8 = torch.randint(0, 3, (8, 1, 256, 256), generator=g)
    tiles, train = {}, {}
    for wname in ("baseline32", "reference16"):
        net = make_reference_unet(seed, channels=WIDTHS[wname]).eval()
        with torch.no_grad():
            net(x32[:1])                                                                        # warm up
            for n, min_s in ((4, 1.0), (32, 2.0)):
                t, reps, total = _median_seconds(lambda: net(x32[:n]), min_s)
                tiles[f"{wname}_n{n}"] = {"value": round(n * 65536 / 1e6 / t, 3), "unit": "Mpixels/s (tile pixels, forward only)",
                                          "ms": round(t * 1e3, 2), "reps": reps, "timed_s": round(total, 2)}
        net.train()
        opt = torch.optim.AdamW(net.parameters(), 6e-4)

        def one_step():
            opt.zero_grad()
            loss = dice_ce_ref(net(x32[:8]), one_hot_ref(y8, 3))
            loss.backward()
            opt.step()

        one_step()
        t, reps, total = _median_seconds(one_step, 3.0, min_reps=3, max_reps=6)
        train[wname] = {"images_per_s": round(8 / t, 2), "ms_per_step": round(t * 1e3, 1), "batch": 8, "reps": reps, "timed_s": round(total, 2)}
    note = (f"torch-CPU oracle (oracle/unet_ref.py, oracle/train_ref.py), {threads} threads, fp32, seeded uniform tiles; median of the repeats")
    return {"tiles": dict(tiles, cores=threads, kind="port", note=note), "train_step": dict(train, cores=threads, kind="port", note=note)}


def eager_rocm(dev, size, roi, overlap, width, seed, engine_ms):
    """BASELINE.json configs[1]'s comparator, "HIP conv/BN kernels vs PyTorch-ROCm eager": what the reference's own code executes on this
    GPU - the restated network (oracle/unet_ref.py, torch.nn modules -> MIOpen / rocBLAS through PyTorch-ROCm eager) moved to the device,
    in fp32 and under torch.autocast(bfloat16): the forward over 32 x 3 x 256 x 256 tiles (model_training_3class.py:255), one training
    step at batch 32 (:250-262: zero_grad, forward, DiceCE, backward, AdamW) and the sliding-window pass with sw_batch_size 4
    (predict.py:109-111, up to the class-1 probability map on the host; the reference then post-processes on the host cores, which is
    cpu_baseline's skimage chain).  Runs after every timed region of the engine, like cpu_baseline; `engine_ms` are this run's figures."""
    from oracle.sliding_window_ref import sliding_window_inference_ref
    from oracle.train_ref import dice_ce_ref, one_hot_ref
    from oracle.unet_ref import make_reference_unet

    def timed(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {"torch": torch.__version__, "network": "oracle/unet_ref.py (MONAI UNet restated as torch.nn modules) on cuda:0, eager",
           "note": "ratio = eager ms / engine ms of this same run (engine: cellseg_amd through the C-ABI); bf16 = torch.autocast(bfloat16) "
                   "against the engine's bf16 datapath"}
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(32, 3, 256, 256, generator=g).to(dev)
    y = torch.randint(0, 3, (32, 1, 256, 256), generator=g).to(dev)
    net = make_reference_unet(seed, channels=WIDTHS[width]).to(dev).eval()
    modes = (("fp32", None), ("bf16", torch.bfloat16))

    def ctx(dt):
        import contextlib
        return torch.autocast("cuda", dtype=dt) if dt is not None else contextlib.nullcontext()

    def pair(key, eager):
        eng = engine_ms.get(key)
        return {"eager_ms": round(eager, 3), "engine_ms": round(eng, 3) if eng else None, "ratio": round(eager / eng, 2) if eng else None}

    tiles, train, slide = {}, {}, {}
    with torch.no_grad():
        for name, dt in modes:
            with ctx(dt):
                tiles[name] = pair(f"tiles_{name}", timed(lambda: net(x), 10))
    # the slide as predict.py runs it: windows in groups of 4, logits accumulated on the device, softmax, class-1 map to the host
    img = torch.from_numpy(synthetic_wsi(size, seed)).to(dev)[None]

    def slide_once():
        logits = sliding_window_inference_ref(img, (roi, roi), 4, net, overlap)
        return torch.softmax(logits, 1)[0, 1].cpu()

    with torch.no_grad():
        for name, dt in modes:
            with ctx(dt):
                ms = timed(slide_once, 2, warm=1)
            slide[name] = dict(pair(f"slide_{name}", ms), eager_mpixels_s=round(size * size / 1e3 / ms, 2))
    slide["scope"] = ("eager: predict.py:109-111 (sliding_window_inference sw_batch_size 4 -> softmax -> class-1 probabilities on the host), WITHOUT the "
                      "host post-process that follows in the reference; engine: the whole headline step incl. connected components and the label copy")
    net.train()
    opt = torch.optim.AdamW(net.parameters(), 6e-4)
    onehot = one_hot_ref(y, 3)
    for name, dt in modes:
        def step():
            opt.zero_grad()
            with ctx(dt):
                loss = dice_ce_ref(net(x).float(), onehot)
            loss.backward()
            opt.step()
        train[name] = pair(f"train_{name}", timed(step, 10))
    out.update(tiles_forward_n32=tiles, train_step_batch32=train, slide=slide)
    return out


def lowprec_accuracy(dev, steps=150, n_images=8, size=1024, precisions=("fp16x3", "bf16x6", "bf16", "fp8"), seed=7, width="baseline32", roi=256):
    """Accuracy of the low-precision datapaths where it means something: on a TRAINED net (random-init logits sit at the 0.5
    threshold).  A 32-base U-Net is trained for `steps` AdamW steps on seeded synthetic cells (fp32 engine, ca.TrainStep), then
    `n_images` held-out synthetic images of size^2 are segmented in fp32 and in every mode: instance F1@0.5 against the fp32
    instances (compute_metric.py semantics), pixel agreement of the interior masks, and F1 against the synthetic ground truth
    (the reference's accuracy target is "F1 within 1e-3 of the reference path")."""
    import cellseg_amd as ca
    from cellseg_amd import data, metrics
    torch.manual_seed(seed)
    model = ca.UNet(channels=WIDTHS[width], strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
    ds = data.SyntheticCells(64, 256, 2022)
    batches = [(torch.stack([ds[i]["img"] for i in range(b, b + 8)]).to(dev),
                ca.one_hot(torch.stack([ds[i]["label"] for i in range(b, b + 8)]).to(dev), 3)) for b in range(0, 64, 8)]
    step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
    first = last = None
    for it in range(steps):
        loss = step(*batches[it % len(batches)])
        if it == 0:
            first = loss.item()
    last = loss.item()
    model.eval()
    held = data.SyntheticCells(n_images, size, 9900)
    sums = {p: dict(f1_vs_fp32=0.0, pixel_agreement=0.0, f1_vs_truth=0.0) for p in precisions}
    f1_truth_fp32, cells = 0.0, 0
    with torch.no_grad():
        for i in range(n_images):
            item = held[i]
            img = item["img"].to(dev)
            truth = ca.label((item["label"][0] == 1).to(dev).to(torch.uint8), connectivity=1)      # ground-truth interiors as instances
            res = {}
            for prec in ("fp32",) + tuple(precisions):
                model.precision = prec
                _, _, mask = ca.swi_infer_image(model, img, (roi, roi), 0.5, want_logits=False, want_mask=True)
                res[prec] = (mask.clone(), ca.mask_to_instances(mask))
            cells += int(res["fp32"][1].max())
            f1_truth_fp32 += metrics.f1_score(truth, res["fp32"][1])[5]
            for prec in precisions:
                sums[prec]["f1_vs_fp32"] += metrics.f1_score(res["fp32"][1], res[prec][1])[5]
                sums[prec]["pixel_agreement"] += float((res["fp32"][0] == res[prec][0]).float().mean())
                sums[prec]["f1_vs_truth"] += metrics.f1_score(truth, res[prec][1])[5]
    model.precision = "fp32"
    out = {"trained_net": {"channels": list(WIDTHS[width]), "window": roi, "steps": steps, "loss_first": round(first, 4), "loss_last": round(last, 4),
                           "held_out_images": n_images, "image_size": size, "cells_found_fp32": cells,
                           "f1_vs_truth_fp32": round(f1_truth_fp32 / n_images, 5)}}
    for prec in precisions:
        out[prec] = {k: round(v / n_images, 5) for k, v in sums[prec].items()}
        out[prec]["delta_f1_vs_truth"] = round(out[prec]["f1_vs_truth"] - out["trained_net"]["f1_vs_truth_fp32"], 5)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a torchrun environment starts the N ranks itself
# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(n: int, script: str = None, argv=None) -> int:
    """Start n children (one per GPU) running this same script; the parent never initialises the GPU
    (torch.cuda.device_count() does not, on this image) and never re-execs itself.  All children are polled: when any of them
    exits non-zero the others are terminated within seconds (a rank that died at start-up would otherwise leave the rest in
    dist.barrier() until the collective timeout) and every rank's stderr tail is forwarded."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_dev = torch.cuda.device_count()
    if n_dev < n and "BENCH_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but only {n_dev} HIP device(s) visible "
              "(BENCH_DEVICE=0 BENCH_BACKEND=gloo shares one GPU between the ranks for a functional check)", file=sys.stderr)
        return 2
    procs, logs = [], []
    out0 = tempfile.TemporaryFile("w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        logs.append(tempfile.TemporaryFile("w+"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=logs[r], text=True))
    codes = [None] * n
    failed_first = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed_first is None:
                    failed_first = r
        if failed_first is not None:
            deadline = time.time() + 5.0
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()                      # exact children of this process, by handle
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text0 = out0.read()
    if text0:
        sys.stdout.write(text0)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    for r, log in enumerate(logs):
        log.seek(0)
        tail = log.read()[-(4000 if bad else 1500):]
        if tail.strip():
            print(f"---- rank {r} stderr (tail) ----\n{tail}", file=sys.stderr)
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}; first failure: rank {failed_first}; the other ranks were terminated",
              file=sys.stderr)
        return 1
    return 0


class StripsGuard:
    """Watcher thread of the multi-GPU strips leg (armed on construction).  `fail(msg)` posts msg in the process group's store and leaves
    through `_bail`; the watcher polls the store (and the deadline) twice a second and leaves the same way when any rank has failed or
    the leg hangs: rank 0 prints `line_fn(msg)` - the slides-leg line with `strips_error` - and every rank exits with code 0 at once,
    without entering another collective.  Covers asymmetric failures (one rank raises, its peers sit in a collective)."""
    KEY = "bench_strips_failed"

    def __init__(self, dist, rank, line_fn, deadline_s):
        import threading
        self.rank, self.line_fn, self.deadline_s = rank, line_fn, deadline_s
        self.lock, self.done = threading.Lock(), False
        try:
            self.store = dist.distributed_c10d._get_default_store()
        except Exception:                                           # noqa: BLE001  (no store: the deadline alone guards the leg)
            self.store = None
        self.t_end = time.time() + deadline_s
        threading.Thread(target=self._watch, daemon=True).start()

    def _watch(self):
        while not self.done:
            time.sleep(0.5)
            msg = None
            try:
                if self.store is not None and self.store.check([self.KEY]):
                    msg = self.store.get(self.KEY).decode(errors="replace")
            except Exception:                                       # noqa: BLE001  (rank 0 hosts the store: it has left already)
                msg = "the strips leg failed on another rank"
            if msg is None and time.time() > self.t_end:
                msg = f"the strips leg did not finish within {self.deadline_s:.0f} s (BENCH_STRIPS_DEADLINE_S)"
            if msg is not None:
                self._bail(msg)

    def fail(self, msg):
        try:
            if self.store is not None:
                self.store.set(self.KEY, f"rank {self.rank}: {msg}")
        except Exception:                                           # noqa: BLE001
            pass
        self._bail(f"rank {self.rank}: {msg}")
        time.sleep(3600)                                            # the watcher is already leaving for this process

    def _bail(self, msg):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                sys.stdout.write(json.dumps(self.line_fn(msg)) + "\n")
            else:
                time.sleep(1.0)                                     # rank 0 hosts the store: let the other watchers read the key first
            sys.stdout.flush()
            sys.stderr.write(f"bench.py: rank {self.rank} leaves the strips leg: {msg}\n")
            sys.stderr.flush()
            os._exit(0)

    def disarm(self):
        with self.lock:
            self.done = True


def live_pmc_traffic(group_name, timeout_s=120):
    """HBM bytes per launch of one kernel group, MEASURED by this run: two child passes of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    (counter passes on their own, no tracing flags, the program itself behind `--`) around ONE slide of the default command on one stream,
    corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE reports half the
    bytes of wide coalesced reads) - tools/pmc_summary.py's arithmetic, grouping the kernel symbols under the engine's profiling ids.
    Children of this process (never an exec: the GPU is initialised here); -> (bytes_per_launch, detail dict) or (None, {"error": ...})."""
    import csv
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, {"error": "this run is itself under a profiler"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import group
    finally:
        sys.path.pop(0)
    out_dir = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    got, t0 = {}, time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out_dir, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "1",
                   "--warmup", "0", "--streams", "1", "--cc-stream", "0", "--no-cpu-baseline", "--no-secondary"]
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            files = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if cp.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {counter}: rc {cp.returncode}, {len(files)} counter files: {cp.stderr[-200:]}"}
            tot, cnt = 0.0, 0
            for path in files:
                for r in csv.DictReader(open(path)):
                    if r.get("Counter_Name", counter) == counter and group(r["Kernel_Name"]) == group_name:
                        tot += float(r["Counter_Value"])
                        cnt += 1
            if not cnt:
                return None, {"error": f"no {group_name} launches in the {counter} pass"}
            got[counter] = (tot / cnt, cnt)
    except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
        return None, {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    read_b = 2.0 * got["FETCH_SIZE"][0] * 1024
    write_b = got["WRITE_SIZE"][0] * 1024
    return round(read_b + write_b), {"hbm_read_bytes_per_launch": round(read_b), "hbm_write_bytes_per_launch": round(write_b),
                                     "launches_counted": got["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}


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
    args = ap.parse_args()

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

    def seam_exchange(win, pl, r, w):
        if fail_rank is not None and int(fail_rank) == rank:
            raise RuntimeError(f"provoked failure on rank {rank} (BENCH_STRIPS_FAIL_RANK)")
        if os.environ.get("BENCH_STRIPS_HANG_RANK") == str(rank):     # test hook: this rank never leaves its exchange
            time.sleep(3600)
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
            with torch.cuda.stream(cc_stream if pipelined else torch.cuda.current_stream()):
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
                full = parallel.gather_shares(mask, plan, world, assemble=True)
                return labelled_strips(full, k % world, mine)
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
        """-> dict(dt (labels on host), dt_dev (labels in HBM), prof, ...): K timed steps each, max over ranks.
        extras (rank 0's primary leg): per-image latency, the reference's timer scope, the strips-leg simulation."""
        model = make_model(WIDTHS[width], seed).to(dev)
        model.precision = precision
        step_host, step_dev = make_step(model, strips, True), make_step(model, strips, False)
        step_prof = make_step(model, strips, False, cc_overlap=False)      # per-kernel HIP events: everything on one stream
        res = {}
        with torch.no_grad():
            for _ in range(args.warmup):
                step_host()
            dt, _ = time_steps(step_host, args.steps, barrier)
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
            if extras:
                res["latency"] = measure_latency(model)
                res["incl_tiff"] = measure_reference_scope(model)
                if world == 1 and dist is None:
                    res["strips_sim"] = measure_strips_sim(model, dt_dev / args.steps)
        dt, dt_dev, dt_prof = max_over_ranks([dt, dt_dev, dt_prof])
        res.update(dt=dt, dt_dev=dt_dev, dt_prof=dt_prof, n_instances=int(num.item()),
                   fg_frac=float((labels > 0).float().mean().item()), labels=labels)
        return res

    def measure_latency(model, reps=6):
        """per-image latency = image tensor resident in HBM -> int32 label map on the host, ONE slide at a time,
        synchronised on both sides, no overlap with a neighbour slide (the two slides alternate)."""
        imgs = slide_pair(False)
        ms = []
        for i in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, mask = ca.swi_infer_image(model, imgs[i % 2], roi, args.overlap, engine_batch=args.engine_batch or None, want_logits=False,
                                            want_mask=True, streams=args.streams)
            lab = ca.mask_to_instances(mask)
            host.copy_(lab, non_blocking=True)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ms[1:])
        return {"latency_ms": round(ms[len(ms) // 2], 3), "min_ms": round(ms[0], 3), "max_ms": round(ms[-1], 3), "reps": reps}

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
        512 x 512 windows, fp8 weights / e4m3-staged activations, one 8192 x 8192 slide, 3 timed steps."""
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", "fp8", "--width", "deep64", "--roi", "512", "--size", "8192",
               "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"]
        try:
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                                env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DIST")})
            line = json.loads(cp.stdout.strip().splitlines()[-1])
        except (subprocess.SubprocessError, ValueError, IndexError, OSError) as e:
            return {"error": f"{type(e).__name__}: {e}"[:300]}
        return {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "value_device", "roofline", "roofline_by_dtype") if k in line} | {
            "workload": line["config"]["workload"], "windows": line["config"]["windows"],
            "kernels": dict(list(line["kernels"].items())[:12]),
            "note": "python bench.py --precision fp8 --width deep64 --roi 512 --size 8192 --steps 3 --warmup 1 as a child process on the same GPU"}

    def infer_summary(res, strips):
        slides = 1 if strips else world
        return {"value": round(slides * mpix * args.steps / res["dt"], 3), "ms_per_step": round(res["dt"] / args.steps * 1e3, 3),
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
        prof = res.get("prof")
        if not prof:
            return None
        precision = precision or args.precision
        total_ms = sum(v["ms"] for v in prof.values())
        if only is not None:
            prof = {n: v for n, v in prof.items() if only(n) and v["flops"] > 0}
            if not prof:
                return None
        name, k = max(prof.items(), key=lambda kv: kv[1]["ms"])
        mfma = k["flops"] > 0 and ("mfma" in name or "wino" in name or name.endswith("_fp8"))
        sec = k["ms"] / 1e3
        if mfma:
            achieved = k["flops"] / sec / 1e12
            # the peak of the instruction the launches of this group really issue: the fp32 Winograd / direct kernels run on the fp32 matrix
            # instruction whatever the mode of the other layers; in fp8 mode the e4m3 launches are their own groups (*_fp8) and every other
            # matrix kernel is a bf16 one
            peak_of = "fp32" if ("wino" in name or precision == "fp32") else ("fp8" if name.endswith("_fp8") else ("bf16" if precision == "fp8" else precision))
            peak = round(MFMA_PEAK[peak_of], 1)
            rl = {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                  "frac": round(achieved / peak, 4), "traffic": None, "peak_of": peak_of}
        else:
            achieved = k["bytes"] / sec / 1e9
            rl = {"bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
        # HBM bytes per launch of this group: measured by THIS run where it can be (the default command on one GPU: two child passes of
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction applied - live_pmc_traffic), else quoted from the committed passes
        # of the same command (profiles/pmc_traffic.json, tools/pmc_summary.py) and labelled as a quote
        if res.get("primary") and args.engine_batch == 0 and args.size == 4096 and args.width == "baseline32" and args.precision == "fp32":
            rl["algorithmic_bytes_per_launch"] = round(k["bytes"] / k["launches"])
            live = None
            if world == 1 and dist is None and not args.no_secondary and not args.no_live_traffic:
                live, detail = live_pmc_traffic(name)
                rl["traffic_live"] = detail
            if live is not None:
                rl["traffic"] = live
                rl["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of `bench.py --steps 1 --warmup 0 "
                                        "--streams 1 --cc-stream 0 --no-cpu-baseline --no-secondary`, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch")
            else:
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(name, {})
                    rl["traffic"] = pmc.get("hbm_bytes_per_launch")
                    rl["traffic_source"] = "QUOTED from profiles/pmc_traffic.json (committed rocprofv3 PMC passes of the default command), not measured by this run"
                except (OSError, ValueError, KeyError):
                    pass
        if "wino" in name:
            # Winograd: `achieved` counts the multiplies the algorithm really issues on the matrix pipe (F(2x2,3x3): 16 per 2x2 output
            # tile and channel pair; F(4x4,3x3): 36 per 4x4 tile); the same launches expressed in direct-convolution FLOPs (9 per
            # output pixel: SURVEY 8(d)'s figure)
            f4 = name.endswith("wino4")
            factor = 144 / 36 if f4 else 36 / 16
            rl["algorithm"] = ("Winograd F(4x4,3x3), fp32: 36 instead of 144 multiplies per 4x4 output tile and channel pair (every stride-1 3x3 "
                               "layer with >= 32 input channels)" if f4 else
                               "Winograd F(2x2,3x3), fp32: 16 instead of 36 multiplies per output tile and channel pair")
            rl["direct_conv_equivalent"] = round(rl["achieved"] * factor, 3)
            rl["frac_direct_conv_equivalent"] = round(rl["achieved"] * factor / rl["peak"], 4)
        rl.update({"kernel": name, "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 5),
                   "algorithmic_per_launch": (k["flops"] if mfma else k["bytes"]) / k["launches"],
                   "share_of_gpu_time": round(k["ms"] / total_ms, 4)})
        return rl

    def kernel_table(res):
        tot = sum(v["ms"] for v in res["prof"].values())
        return {n: {"ms_per_step": round(v["ms"] / args.steps, 4), "share": round(v["ms"] / tot, 4),
                    "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] else None,
                    "gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} for n, v in sorted(res["prof"].items(), key=lambda kv: -kv[1]["ms"])}

    def base_line(main_res, head_strips):
        """The contract keys of the JSON line for one measured inference leg (slides or strips)."""
        main_res["primary"] = True
        rccl_ranks = dist.get_world_size() if dist else 1
        slides = 1 if head_strips else world               # slides segmented per step by the whole job
        ms_step = main_res["dt"] / args.steps * 1e3
        out = {
            "metric": f"Mpixels/s sliding-window inference ({args.size}^2 WSI) incl. GPU connected components, label map on host; pipelined steady state "
                      f"(the label copy and connected components of slide k overlap the windows of slide k + 1; per-image figure: latency_ms)",
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
                                   + ("one slide cut into row strips" if head_strips else "one slide per GPU"),
                       "windows": n_windows, "engine_batch": args.engine_batch or ca.balanced_batch(n_windows, args.streams), "engine_batch_auto": not args.engine_batch,
                       "streams": args.streams, "precision": args.precision,
                       "parallelism": (f"1 slide split by its window list over {world} ranks + seam send/recv + mask all-gather, labelling rank round-robin" if head_strips
                                       else f"slides x{world}"),
                       "instances_found": main_res["n_instances"], "foreground_fraction": round(main_res["fg_frac"], 4)},
            "value_device": round(slides * mpix * args.steps / main_res["dt_dev"], 3),
            "value_device_note": "same loop ending when the int32 label map is complete in HBM (no copy to the host)",
            "value_with_events": round(slides * mpix * args.steps / main_res["dt_prof"], 3),
            "roofline": roofline(main_res),
            "kernels": kernel_table(main_res),
        }
        if args.precision == "fp8":
            # mixed mode: the e4m3 launches against the fp8 peak, the bf16 launches of the narrow layers against the bf16 peak - one record per group
            out["roofline_by_dtype"] = {"fp8": roofline(main_res, only=lambda n: n.endswith("_fp8")),
                                        "bf16": roofline(main_res, only=lambda n: not n.endswith("_fp8") and "wino" not in n and "mfma" in n)}
        return out

    # ------------------------------------------------------------------------------------------------------------------
    head_strips = args.shard == "strips" and dist is not None
    main_res = sec_res = slides_res = None
    modes, other_part, train_res = {}, None, None
    strips_equal = None
    train_ref = None
    strips_error = None
    if args.mode == "infer":
        if head_strips:
            # the strips leg is the only one with data-path collectives (seam send/recv, mask all-gather).  The collective-free slides
            # leg runs FIRST and is kept: should the strips leg fail on this node (it has never run on real multi-GPU hardware here), the
            # job still reports the slides figure - marked as such - instead of exiting without a line
            slides_res = measure_infer(args.width, args.precision, False, True, extras=not args.no_secondary and rank == 0)

            def fallback_line(msg):
                line = base_line(slides_res, False)
                line["strips_error"] = msg
                line["config"]["note"] = "the strong-scaling strips leg failed on this node (strips_error); the headline is the collective-free slides leg"
                return line

            # A failure inside the strips leg need not be symmetric: a rank that raises leaves its peers inside the seam send/recv or the
            # mask all-gather, and any later collective then mismatches until the process-group timeout aborts the job without a line.  So
            # from here to the end of the leg every rank runs a watcher thread: a rank that fails posts the reason in the process group's
            # store, every watcher sees it within half a second (or the leg's deadline passes), rank 0 prints the slides line with
            # `strips_error` set and every rank leaves at once - no collective is entered after a failure.
            guard = StripsGuard(dist, rank, fallback_line, float(os.environ.get("BENCH_STRIPS_DEADLINE_S", "150")))
            try:
                ensure_seam_transport()
                main_res = measure_infer(args.width, args.precision, True, True)      # ends with a collective: every rank got this far
            except Exception as e:                                  # noqa: BLE001
                guard.fail(f"{type(e).__name__}: {e}"[:500])        # does not return
            guard.disarm()
            if not args.no_secondary or os.environ.get("BENCH_COMPARE_LEGS") == "1":
                other_part = infer_summary(slides_res, False)
                strips_equal = bool(torch.equal(slides_res["labels"], main_res["labels"])) if rank == 0 else None
        else:
            main_res = measure_infer(args.width, args.precision, False, True, extras=not args.no_secondary and rank == 0)
        if not args.no_secondary:
            if dist is not None and not head_strips and strips_error is None:
                ensure_seam_transport()
                other = measure_infer(args.width, args.precision, True, False)
                other_part = infer_summary(other, True)
                # rank 0's slides-leg slide pair is the strips leg's pair: the two partitions must produce the same label map
                strips_equal = bool(torch.equal(other["labels"], main_res["labels"])) if rank == 0 else None
            if world == 1 and dist is None and args.width != "reference16":
                sec_res = measure_infer("reference16", args.precision, False, True)
            if world == 1 and dist is None and args.precision == "fp32":
                for prec in ("fp16x3", "bf16x6", "bf16", "fp8"):
                    r = measure_infer(args.width, prec, False, False)
                    modes[prec] = dict(infer_summary(r, False), foreground_agreement_vs_primary_random_init=round(
                        float(((r["labels"] > 0) == (main_res["labels"] > 0)).float().mean().item()), 6))
                # ... and what the modes do to the RESULT, on a net whose logits are decisive (trained on synthetic cells, ~1 s)
                acc = lowprec_accuracy(dev)
                modes["accuracy_on_trained_net"] = acc.pop("trained_net")
                for prec, a in acc.items():
                    modes[prec].update(a)
            tw = args.width if args.width in WINDOW_GFLOP and args.roi == 256 else "baseline32"
            train_res = measure_train(tw)
            if world == 1 and dist is None:                      # configs[1]: the same step with the forward convolutions on the bf16 pipe
                for prec in ("fp16x3", "bf16"):                  # the same step with forward convolutions + weight gradients on the 16-bit pipe
                    t16 = measure_train(tw, with_prof=False, precision=prec, eager_too=False)
                    train_res[prec] = {k: t16[k] for k in ("images_per_s", "ms_per_step", "tflops", "loss", "dtype")}
                # the reference's own training configuration (model_training_3class.py:201-208, --batch_size 8): 16...256 channels, batch 8
                tref = measure_train("reference16", with_prof=False, batch=8)
                train_ref = {k: tref[k] for k in ("images_per_s", "ms_per_step", "eager_ms_per_step", "graph_ms_per_step", "graph_nodes", "tflops", "loss",
                                                  "per_gpu_batch", "width")}
                tref16 = measure_train("reference16", with_prof=False, batch=8, precision="bf16", eager_too=False)      # the same step in the bf16 mode
                train_ref["bf16"] = {k: tref16[k] for k in ("images_per_s", "ms_per_step", "tflops", "loss", "dtype")}
    else:
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
            "steps": args.train_steps, "warmup": max(args.warmup, 2), "ms_per_step": train_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"training step on {args.train_batch}x3x256x256 synthetic tiles per GPU, 3-class residual U-Net channels="
                                   f"{list(WIDTHS[args.width])} num_res_units=2 InstanceNorm+PReLU random-init, DiceCE(softmax) + AdamW(6e-4), "
                                   f"one flat fp32 gradient bucket all-reduced per step",
                       "global_batch": world * args.train_batch, "parallelism": f"dp{world}"},
            "train_step": train_res,
        }
        print(json.dumps(out))
        if dist:
            dist.destroy_process_group()
        return

    out = base_line(main_res, head_strips)
    strips_info = {"seam_transport": seam["transport"], "strips_error": strips_error,
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
        if head_strips:
            out["config"].update(strips_info)
    if strips_error:
        out["strips_error"] = strips_error
        out["config"]["note"] = "the strong-scaling strips leg failed on this node (strips_error); the headline is the collective-free slides leg"
    extras_res = slides_res if head_strips else main_res          # latency / reference scope are per-image figures of the slides leg
    if extras_res.get("latency"):
        out.update(latency_ms=extras_res["latency"]["latency_ms"], latency=dict(
            extras_res["latency"], note="one slide at a time: image tensor resident in HBM -> int32 labels in pinned host memory, synchronised "
                                      "on both sides, no overlap with a neighbour slide (SURVEY 8(d) per-image latency)"))
    if extras_res.get("incl_tiff"):
        out.update(value_incl_tiff=extras_res["incl_tiff"]["value_incl_tiff"], reference_timer_scope=extras_res["incl_tiff"])
    if extras_res.get("strips_sim"):
        out["strips_sim"] = extras_res["strips_sim"]
    if sec_res:
        out["reference_width"] = dict(infer_summary(sec_res, False), channels=list(WIDTHS["reference16"]), roofline=roofline(sec_res))
    if modes:
        out["other_precisions"] = {"note": "same workload, convolutions on the 16-bit matrix pipe (v_mfma_f32_16x16x32_{f16,bf16}, fp32 "
                                           "accumulate): fp16x3 = 2 fp16 planes (22 mantissa bits) x 3 terms and bf16x6 = 3 bf16 planes x "
                                           "6 terms are fp32-grade (logits within 1e-4 of the oracle in the GPU tests); bf16 = the bf16 datapath "
                                           "(bf16 weights, bf16 activations in HBM); fp8 = e4m3 weights with per-channel scales, e4m3-staged "
                                           "activations, v_mfma_scale_f32_16x16x128_f8f6f4 (built for the 64...1024 net of configs[4]: at this "
                                           "width most layers fill a fraction of its K = 128).  f1_vs_fp32 / pixel_agreement / delta_f1_vs_truth: "
                                           "instance F1@0.5 and interior-mask agreement against the fp32 path, and the change of the F1 against "
                                           "the synthetic ground truth, on the synthetic-trained net described under accuracy_on_trained_net",
                                   **modes}
    if train_res:
        out["train_step"] = train_res
    if train_ref:
        out["train_step_reference"] = dict(train_ref, note="the reference's own training setting: channels 16...256, batch 8 x 3 x 256 x 256 "
                                                            "(baseline/model_training_3class.py:201-208, 247-267)")
    if world == 1 and dist is None and not args.no_secondary and args.width in WINDOW_GFLOP and args.roi == 256:
        # forward-only tile throughput (N = 8: configs[0]'s batch, N = 32: configs[1]'s), exact fp32 and the bf16 datapath
        out["tiles"] = {prec: {f"n{n}": measure_tiles(args.width, prec, n) for n in (8, 32)} for prec in ("fp32", "bf16")}
        out["tiles"]["note"] = ("forward-only U-Net on N x 3 x 256 x 256 synthetic tiles (torch.rand, seed 2022), logits left in HBM; value = tile "
                                "pixels / time; roofline = the dominant kernel of that forward (HIP events on the launch stream)")
        if args.precision == "fp32" and args.size == 4096:
            out["configs4_fp8"] = measure_configs4()
            try:          # ... and its accuracy claim at its OWN width and window: the 64...1024 net trained on synthetic cells, 512^2 windows
                acc4 = lowprec_accuracy(dev, n_images=16, size=2048, precisions=("bf16", "fp8"), width="deep64", roi=512)      # ~31 000 cells: 1e-3 = 31 cells
                out["configs4_fp8"]["accuracy_on_trained_net"] = acc4
            except Exception as e:                                  # noqa: BLE001
                out["configs4_fp8"]["accuracy_on_trained_net"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and dist is None and not args.no_secondary and args.precision == "fp32" and "tiles" in out and train_res and modes:
        try:
            engine_ms = {"tiles_fp32": out["tiles"]["fp32"]["n32"]["ms_per_forward"], "tiles_bf16": out["tiles"]["bf16"]["n32"]["ms_per_forward"],
                         "train_fp32": train_res["ms_per_step"], "train_bf16": train_res.get("bf16", {}).get("ms_per_step"),
                         "slide_fp32": out["ms_per_step"], "slide_bf16": modes["bf16"]["ms_per_step"]}
            out["eager_rocm"] = eager_rocm(dev, args.size, args.roi, args.overlap, args.width, seed, engine_ms)
        except Exception as e:                                      # noqa: BLE001  (a comparator must not cost the line)
            out["eager_rocm"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.roi, args.overlap, WIDTHS[args.width], seed)
        if not args.no_secondary:
            out["cpu_baseline"].update(cpu_baseline_tiles_and_train(seed))
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
