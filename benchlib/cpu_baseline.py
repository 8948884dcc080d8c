"""bench.py's `cpu_baseline` leg: the oracle (torch-CPU restatement + plain-C connected components) timed on the GPU box's host cores.
This is the one leg of the bench that may import `oracle/` (the checker, timed here as the reported CPU baseline - never the product path)."""
from __future__ import annotations

import json
import os
import subprocess
import time

import numpy as np
import torch

from .common import ROOT, WIDTHS, effective_cpus, synthetic_wsi

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

