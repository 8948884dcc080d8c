#!/usr/bin/env python3
"""bench.py - headline benchmark: Mpixels/s of whole-slide sliding-window inference on MI355X.

One "step" = the reference's per-image hot path (baseline/predict.py:107-113) on one synthetic 3x4096x4096
slide that is already resident in HBM: sliding-window U-Net inference (256 window / 128 stride = BASELINE.json
configs[2], 961 windows) -> softmax -> class-1 > 0.5 -> remove_small_holes -> remove_small_objects -> 8-connected
label.  The timed region ends when the int32 label map is complete in HBM (`value`); the variant that also copies
the labels to pinned host memory is reported as `value_incl_d2h`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one rank per GPU, RCCL)

Multi-GPU (weak scaling): slides are independent units, every rank segments its own slide, no data-path
collective; value = N slides' pixels / max-over-ranks time.
Prints ONE JSON line on rank 0 (contract in the task description), including `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the torch-CPU / C oracle on a bounded sample, host cores).
"""
from __future__ import annotations

import argparse
import json
import os
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
MFMA_PEAK = {"fp32": PEAK_F32_MFMA_TFLOPS, "fp16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3,
             "bf16": PEAK_BF16_MFMA_TFLOPS}
DTYPE = {"fp32": "f32", "fp16x3": "f32 (split-fp16, 2 planes x 3 MFMA terms, fp32 accumulate)", "bf16x6": "f32 (split-bf16, 3 planes x 6 MFMA terms, fp32 accumulate)",
         "bf16x3": "split-bf16 (2 planes x 3 terms)", "bf16": "bf16"}
PEAK_HBM_GBS = 8000.0
WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256),
          "deep64": (64, 128, 256, 512, 1024)}          # BASELINE.json configs[4] width (use with --roi 512)


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


STREAMS = 2


def run_step(ca, model, image, roi, overlap, engine_batch):
    _, _, mask = ca.swi_infer_image(model, image, roi, overlap, engine_batch=engine_batch, want_logits=False, want_mask=True,
                                    streams=STREAMS)
    return ca.mask_to_instances(mask, return_num=True)


def time_steps(fn, steps, barrier):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
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


def cpu_baseline(size, roi, overlap, channels, seed):
    """Oracle (torch-CPU restatement + plain-C CC) on a bounded sample of the same workload, extrapolated."""
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
    with torch.no_grad():
        sliding_window_inference_ref(img[:, :, :roi, :roi], (roi, roi), 4, net, overlap)          # warm up
        t0 = time.perf_counter()
        logits = sliding_window_inference_ref(img, (roi, roi), 4, net, overlap)
        t_net = time.perf_counter() - t0
    prob = softmax_interior_ref(logits[0].numpy())
    reps = int(np.ceil(size / crop))
    prob_full = np.tile(prob, (reps, reps))[:size, :size]                                       # full-size CC input
    t0 = time.perf_counter()
    postprocess_c(prob_full)
    t_cc = time.perf_counter() - t0
    t_full = t_net * n_full / n_sample + t_cc
    return {"value": round(size * size / 1e6 / t_full, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU oracle on {n_sample} of {n_full} windows ({crop}x{crop} crop, {t_net:.2f}s, {threads} threads) "
                      f"scaled to the slide + plain-C CC oracle on a full {size}x{size} mask ({t_cc:.2f}s, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--roi", type=int, default=256)
    ap.add_argument("--overlap", type=float, default=0.5, help="0.5 = stride 128 (BASELINE configs[2]); reference default 0.25")
    ap.add_argument("--width", choices=sorted(WIDTHS), default="baseline32")
    ap.add_argument("--engine-batch", type=int, default=256)
    ap.add_argument("--streams", type=int, default=2, help="internal streams the window chunks are dealt to (1 = serial)")
    ap.add_argument("--precision", choices=sorted(MFMA_PEAK), default="fp32",
                    help="matrix-core arithmetic of the convolutions (activations stay fp32 in HBM)")
    ap.add_argument("--shard", choices=["slides", "strips"], default="slides",
                    help="multi-GPU partition: one slide per rank (weak scaling, no collective) or ONE slide cut into row strips "
                         "with an all-gather of the uint8 mask strips before the connected-component pass (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the reference-width (16..256) secondary measurement")
    args = ap.parse_args()

    global STREAMS
    STREAMS = args.streams
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the hot path")
    # BENCH_DEVICE / BENCH_BACKEND exist only to exercise the multi-rank code path on a 1-GPU box (ranks share cuda:0, gloo)
    local = int(os.environ.get("BENCH_DEVICE", local))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    barrier = (lambda: dist.barrier()) if dist else (lambda: None)

    import cellseg_amd as ca
    from cellseg_amd import _lib

    seed = 2022
    strips = args.shard == "strips" and world > 1
    image = torch.from_numpy(synthetic_wsi(args.size, seed + (0 if strips else rank))).to(dev)   # resident in HBM before timing
    from cellseg_amd import parallel
    row_lo, row_hi = ca.rank_rows(args.size, rank, world) if strips else (0, args.size)

    def run_step_strips(model):
        """One slide across ranks: every rank runs the windows that touch its row strip, finalises its rows, the uint8 mask
        strips are all-gathered (RCCL), and every rank labels the full mask (2.1 MB per rank at 4096^2)."""
        _, _, mask = ca.swi_infer_image(model, image, roi, args.overlap, engine_batch=args.engine_batch, rows=(row_lo, row_hi),
                                        want_logits=False, want_mask=True, streams=STREAMS)
        strip = mask[row_lo:row_hi]
        if backend != "nccl":
            full = parallel.gather_row_strips(strip.cpu(), args.size).to(dev)
        else:
            full = parallel.gather_row_strips(strip, args.size)
        return ca.mask_to_instances(full, return_num=True)
    roi = (args.roi, args.roi)
    plan = ca.make_plan(args.size, args.size, roi, args.overlap)
    n_windows = plan.n_h * plan.n_w
    mpix = args.size * args.size / 1e6

    def measure_mode(width, precision, ref_labels=None):
        """Timed loop only (same K steps) for a secondary matrix-core mode; agreement with the primary run's labels."""
        model = make_model(WIDTHS[width], seed).to(dev)
        model.precision = precision
        step = (lambda: run_step_strips(model)) if strips else (lambda: run_step(ca, model, image, roi, args.overlap, args.engine_batch))
        with torch.no_grad():
            for _ in range(max(args.warmup, 1)):
                step()
            dt, (labels, num) = time_steps(step, args.steps, barrier)
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out = {"value": round((1 if strips else world) * mpix * args.steps / t.item(), 3), "ms_per_step": round(t.item() / args.steps * 1e3, 3),
               "instances_found": int(num.item())}
        if ref_labels is not None:
            out["foreground_agreement_vs_primary"] = round(float(((labels > 0) == (ref_labels > 0)).float().mean().item()), 6)
        return out

    def measure(width):
        model = make_model(WIDTHS[width], seed).to(dev)
        model.precision = args.precision
        step = (lambda: run_step_strips(model)) if strips else (lambda: run_step(ca, model, image, roi, args.overlap, args.engine_batch))
        with torch.no_grad():
            for _ in range(max(args.warmup, 1) if args.warmup else 0):
                step()
            dt, (labels, num) = time_steps(step, args.steps, barrier)
            # same K steps again with HIP events around every kernel launch (on the launch stream) for the roofline
            _lib.lib.cs_prof_reset()
            _lib.lib.cs_prof_enable(1)
            dt_prof, _ = time_steps(step, args.steps, barrier)
            _lib.lib.cs_prof_enable(0)
            prof = _lib.prof_collect()
            # variant: labels on the host (pinned buffer)
            host = torch.empty((args.size, args.size), dtype=torch.int32, pin_memory=True)

            def step_d2h():
                lab, _ = step()
                host.copy_(lab, non_blocking=True)
                return lab
            dt_d2h, _ = time_steps(step_d2h, args.steps, barrier)
        t = torch.tensor([dt, dt_prof, dt_d2h], dtype=torch.float64, device=red_dev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_prof, dt_d2h = t.tolist()
        return dict(dt=dt, dt_prof=dt_prof, dt_d2h=dt_d2h, prof=prof, n_instances=int(num.item()),
                    fg_frac=float((labels > 0).float().mean().item()), labels=labels.clone())

    main_res = measure(args.width)
    sec_res, modes = None, {}
    if not args.no_secondary and args.width != "reference16":
        sec_res = measure("reference16")
    if not args.no_secondary and args.precision == "fp32":
        for prec in ("fp16x3", "bf16x6", "bf16"):
            modes[prec] = measure_mode(args.width, prec, main_res["labels"])

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    def roofline(res):
        prof = res["prof"]
        if not prof:
            return None
        name, k = max(prof.items(), key=lambda kv: kv[1]["ms"])
        mfma = k["flops"] > 0 and "mfma" in name
        sec = k["ms"] / 1e3
        if mfma:
            achieved = k["flops"] / sec / 1e12
            peak = round(MFMA_PEAK[args.precision], 1)
            rl = {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                  "frac": round(achieved / peak, 4), "traffic": None}
        else:
            achieved = k["bytes"] / sec / 1e9
            rl = {"bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
        # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # gfx950 correction applied; tools/pmc_summary.py).  bench.py cannot read PMCs itself, so it quotes the summary.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(name, {})
            if res is main_res and args.engine_batch == 256 and args.size == 4096 and args.width == "baseline32" and args.precision == "fp32":
                rl["traffic"] = pmc.get("hbm_bytes_per_launch")
                rl["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 PMC passes of the default command)"
                rl["algorithmic_bytes_per_launch"] = round(k["bytes"] / k["launches"])
        except (OSError, ValueError, KeyError):
            pass
        rl.update({"kernel": name, "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 5),
                   "algorithmic_per_launch": (k["flops"] if mfma else k["bytes"]) / k["launches"],
                   "share_of_gpu_time": round(k["ms"] / sum(v["ms"] for v in prof.values()), 4)})
        return rl

    def kernel_table(res):
        tot = sum(v["ms"] for v in res["prof"].values())
        return {n: {"ms_per_step": round(v["ms"] / args.steps, 4), "share": round(v["ms"] / tot, 4),
                    "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] else None,
                    "gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} for n, v in sorted(res["prof"].items(), key=lambda kv: -kv[1]["ms"])}

    slides = 1 if strips else world               # slides segmented per step by the whole job
    ms_step = main_res["dt"] / args.steps * 1e3
    out = {
        "metric": "Mpixels/s sliding-window inference (4096^2 WSI) incl. GPU connected components",
        "value": round(slides * mpix * args.steps / main_res["dt"], 3),
        "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "strong" if strips else "weak", "vs_baseline": None,
        "dtype": DTYPE[args.precision], "data": "synthetic",
        "config": {"workload": f"sliding-window inference on 3x{args.size}x{args.size} synthetic WSI, {args.roi} window / "
                               f"{plan.interval_h} stride ({n_windows} windows), 3-class residual U-Net channels={list(WIDTHS[args.width])} "
                               f"num_res_units=2 InstanceNorm+PReLU random-init, softmax + >0.5 + remove_small_holes(64) + "
                               f"remove_small_objects(16) + 8-conn label on GPU; " + ("one slide cut into row strips" if strips else "one slide per GPU"),
                   "windows": n_windows, "engine_batch": args.engine_batch, "streams": args.streams, "precision": args.precision, "parallelism": (f"1 slide in {world} row strips + mask all-gather" if strips else f"slides x{world}"),
                   "instances_found": main_res["n_instances"], "foreground_fraction": round(main_res["fg_frac"], 4)},
        "value_incl_d2h": round(slides * mpix * args.steps / main_res["dt_d2h"], 3),
        "value_with_events": round(slides * mpix * args.steps / main_res["dt_prof"], 3),
        "roofline": roofline(main_res),
        "kernels": kernel_table(main_res),
    }
    if sec_res:
        out["reference_width"] = {"channels": list(WIDTHS["reference16"]), "value": round(slides * mpix * args.steps / sec_res["dt"], 3),
                                  "ms_per_step": round(sec_res["dt"] / args.steps * 1e3, 3), "roofline": roofline(sec_res),
                                  "instances_found": sec_res["n_instances"]}
    if modes:
        out["other_precisions"] = {"note": "same workload, convolutions on the 16-bit matrix pipe (v_mfma_f32_16x16x32_{f16,bf16}, fp32 "
                                           "accumulate): fp16x3 = 2 fp16 planes (22 mantissa bits) x 3 terms and bf16x6 = 3 bf16 planes x "
                                           "6 terms are fp32-grade (logits within 1e-4 of the oracle in the GPU tests), bf16 = plain bf16 "
                                           "products; activations stay fp32 in HBM", **modes}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.roi, args.overlap, WIDTHS[args.width], seed)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
