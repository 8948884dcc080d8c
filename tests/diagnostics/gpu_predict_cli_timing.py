#!/usr/bin/env python3
"""Diagnostic (GPU): wall time of the ready-made `predict` CLI on a folder of 4096^2 RGB slides (file reading, device percentile
normalisation, sliding-window inference, post-process, label TIFF), per image and overall.

    python tests/diagnostics/gpu_predict_cli_timing.py [n_images]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np
from scipy.ndimage import gaussian_filter
from cellseg_amd import preprocess

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with tempfile.TemporaryDirectory() as td:
    inp, outp = os.path.join(td, "in"), os.path.join(td, "out")
    os.makedirs(inp)
    rng = np.random.default_rng(0)
    g = gaussian_filter(rng.random((4096, 4096), dtype=np.float32), 6)
    g = (g - g.min()) / (g.max() - g.min())
    for k in range(n):
        img = (np.stack([np.roll(g, 97 * k, 0), np.roll(g, 131 * k, 1), g], -1) * 255).astype(np.uint8)
        preprocess.write_tiff(os.path.join(inp, f"slide_{k:02d}.tiff"), img, level=1)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "neurips-cellseg_amd"), ROOT, os.environ.get("PYTHONPATH", "")]))
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-m", "cellseg_amd.predict", "-i", inp, "-o", outp, "--random_init", "--model_name", "unet"],
                         capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    print(out.stdout[-1500:])
    if out.returncode:
        print(out.stderr[-2000:])
        sys.exit(1)
    costs = [float(l.split("costing:")[1].strip().rstrip("s")) for l in out.stdout.splitlines() if "costing:" in l]
    sizes = [os.path.getsize(os.path.join(outp, f)) for f in sorted(os.listdir(outp))]
    print(f"{n} slides of 4096^2: process wall time {dt:.2f} s (incl. interpreter + library start-up), per-image 'costing' "
          f"median {np.median(costs) * 1e3:.0f} ms (first {costs[0] * 1e3:.0f} ms), label files {min(sizes)} .. {max(sizes)} bytes")
