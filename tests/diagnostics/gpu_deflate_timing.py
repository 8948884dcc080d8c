"""Diagnostic (GPU): time of the device label-map writer on a 4096^2 cell map against the host zlib writer.

    python tests/diagnostics/gpu_deflate_timing.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "neurips-cellseg_amd"))
from cellseg_amd.preprocess import deflate_labels_gpu, write_label_tiff  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    h = w = 4096
    lab = np.zeros((h, w), np.int32)
    yy, xx = np.mgrid[0:64, 0:64]
    for k in range(1, 8436):
        cy, cx, r = rng.integers(0, h - 64), rng.integers(0, w - 64), rng.integers(4, 14)
        m = (yy - 32) ** 2 + (xx - 32) ** 2 <= r * r
        lab[cy:cy + 64, cx:cx + 64][m] = k
    t = torch.from_numpy(lab).cuda()
    host = torch.empty((h, w), dtype=torch.int32).pin_memory()
    for _ in range(2):
        deflate_labels_gpu(t, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    for _ in range(10):
        buf, sizes = deflate_labels_gpu(t, 64)
    dt = (time.perf_counter() - t0) / 10 * 1e3
    print(f"device encoder + copies to the host: {dt:.3f} ms per map, {sum(sizes)} bytes ({sum(sizes) / lab.nbytes * 100:.2f} % of the map)")
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "a_label.tiff")
        t0 = time.perf_counter()
        for _ in range(5):
            write_label_tiff(p, t)
        print(f"write_label_tiff from the device: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms, file {os.path.getsize(p)} bytes")
        t0 = time.perf_counter()
        for _ in range(3):
            host.copy_(t, non_blocking=True)
            torch.cuda.synchronize()
            write_label_tiff(p, host.numpy())
        print(f"copy to the host + zlib level 6 on {min(16, len(os.sched_getaffinity(0)))} threads: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms, "
              f"file {os.path.getsize(p)} bytes")


if __name__ == "__main__":
    main()
