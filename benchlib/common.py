"""Constants and small helpers shared by the legs of bench.py."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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

