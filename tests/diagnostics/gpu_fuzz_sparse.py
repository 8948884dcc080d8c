#!/usr/bin/env python3
"""Randomised fp32 parity sweep on SPARSE inputs (TEST INFRASTRUCTURE: compares against the oracle).

    python tests/diagnostics/gpu_fuzz_sparse.py [--seconds 120] [--seed 0] [--out file.jsonl]

Round 6 found the engine off by up to 0.3 logits on blank windows (tests/test_gpu_blank_windows.py); this sweep looks for whatever else
lives in that corner: random images composed of an exactly-zero (or constant) background, rectangles / blobs / single pixels / noise
patches, optionally quantised through uint8 (predict.py:94-107), random sizes, window sizes and overlaps, on random-depth nets - through
the tiler (cs_swi_infer) against the torch-CPU oracle AND a float64 evaluation of it.  Bars as in the test: |engine - float64| <=
max(1e-4, 4e-6 max|logit|), 4e-4 where torch-fp32 itself is more than 1e-4 from float64 (half of the random cases; median 1.3e-2 there); and within 1e-4 (+ its own distance) of
torch-fp32 wherever that is within 1e-4 of float64.  Prints one JSON line per case to --out and a summary; exits non-zero on a miss."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np
import torch
import cellseg_amd as ca
from oracle.sliding_window_ref import sliding_window_inference_ref
from oracle.unet_ref import UNetRef

TOL, REL = 1e-4, 4e-6


def random_sparse_image(rng, h, w):
    from scipy.ndimage import gaussian_filter
    bg = rng.choice([0.0, 0.0, 0.0, float(rng.uniform(0.05, 0.9))])
    x = np.full((3, h, w), bg, np.float32)
    if bg and rng.random() < 0.5:
        x *= rng.uniform(0.3, 1.0, (3, 1, 1)).astype(np.float32)
    for _ in range(int(rng.integers(0, 6))):
        kind = int(rng.integers(0, 5))
        y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
        a, b = int(rng.integers(1, max(2, h // 2))), int(rng.integers(1, max(2, w // 2)))
        sl = (slice(None), slice(y0, min(h, y0 + a)), slice(x0, min(w, x0 + b)))
        shape = x[sl].shape
        if kind == 0:
            x[sl] = rng.random(shape, dtype=np.float32)
        elif kind == 1:
            x[sl] = float(rng.uniform(0, 1))
        elif kind == 2:
            g = gaussian_filter(rng.random(shape[1:], dtype=np.float32), float(rng.uniform(1, 6)))
            g = (g - g.min()) / max(float(g.max() - g.min()), 1e-6)
            x[sl] = np.where(g > np.quantile(g, rng.uniform(0.3, 0.9)), g, 0.0)[None] * rng.uniform(0.3, 1.0, (3, 1, 1)).astype(np.float32)
        elif kind == 3:
            x[:, y0, x0] = 1.0
        else:
            x[sl] = 0.0
    if rng.random() < 0.6:                                   # the reference's uint8 pipeline: pre / max(pre)
        u8 = np.round(np.clip(x, 0, 1) * 255).astype(np.uint8)
        x = (u8 / max(int(u8.max()), 1)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


def run(seconds=120.0, seed=0, out=None):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    nets = []
    for ch in ((16, 32, 64), (16, 32, 64, 128, 256), (32, 64, 128, 256, 512)):
        torch.manual_seed(seed + len(ch) + ch[0])
        strides = (2,) * (len(ch) - 1)
        ref = UNetRef(channels=ch, strides=strides, num_res_units=2).eval()
        for n, p in ref.named_parameters():                  # non-trivial biases and slopes: blank windows are driven by them
            if n.endswith("bias"):
                p.data.uniform_(-0.3, 0.3)
            if n.endswith("A.weight"):
                p.data.fill_(float(rng.uniform(0.05, 0.45)))
        ref64 = UNetRef(channels=ch, strides=strides, num_res_units=2).eval().double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        hip = ca.UNet(channels=ch, strides=strides, num_res_units=2)
        hip.load_state_dict(ref.state_dict())
        nets.append((ch, ref, ref64, hip.to(dev).eval(), 1 << len(strides)))
    rows, worst = [], {"e64_over_bar": 0.0}
    f = open(out, "a") if out else None
    t_end = time.time() + seconds
    while time.time() < t_end:
        ch, ref, ref64, hip, mult = nets[int(rng.integers(0, len(nets)))]
        roi = int(rng.choice([64, 96, 128, 256])) // mult * mult
        ih, iw = int(rng.integers(8, 420)), int(rng.integers(8, 420))
        ov = float(rng.choice([0.0, 0.25, 0.5]))
        x = random_sparse_image(rng, ih, iw)[None]
        with torch.no_grad():
            t32 = sliding_window_inference_ref(x, (roi, roi), 4, ref, overlap=ov)
            f64 = sliding_window_inference_ref(x.double(), (roi, roi), 4, ref64, overlap=ov)
            got = ca.sliding_window_inference(x.to(dev), (roi, roi), 4, hip, overlap=ov).cpu()
        t64 = float((t32.double() - f64).abs().max())
        e64, e32 = float((got.double() - f64).abs().max()), float((got - t32).abs().max())
        scale = float(f64.abs().max())
        bar = max(4 * TOL, REL * scale) if t64 > TOL else max(TOL, REL * scale)
        ok = e64 <= bar and (t64 > TOL or e32 <= bar + t64)
        row = {"channels": list(ch), "image": [ih, iw], "roi": roi, "overlap": ov, "zero_fraction": round(float((x == 0).float().mean()), 3),
               "engine_vs_fp64": e64, "engine_vs_torch32": e32, "torch32_vs_fp64": t64, "logit_scale": scale, "bar": bar, "ok": ok}
        rows.append(row)
        if f:
            f.write(json.dumps(row) + "\n")
            f.flush()
        if e64 / bar > worst["e64_over_bar"]:
            worst = dict(row, e64_over_bar=e64 / bar)
        if not ok:
            torch.save({"x": x, "row": row}, os.path.join(ROOT, "gpurun_out", "fuzz_sparse_fail.pt"))
            print("SPARSE MISMATCH", json.dumps(row))
            raise AssertionError("gpu_fuzz_sparse mismatch (see the line above)")
    ill = [r for r in rows if r["torch32_vs_fp64"] > TOL]
    return {"seed": seed, "seconds": seconds, "cases": len(rows), "cases_where_torch32_is_off_float64_by_more_than_1e-4": len(ill),
            "worst_engine_vs_fp64": max(r["engine_vs_fp64"] for r in rows), "worst_torch32_vs_fp64": max(r["torch32_vs_fp64"] for r in rows),
            "worst_case": worst, "status": "all within the bars"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    print(json.dumps(run(a.seconds, a.seed, a.out)))
