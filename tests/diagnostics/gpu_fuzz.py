#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (TEST INFRASTRUCTURE: compares against the oracle).

    python tests/diagnostics/gpu_fuzz.py [--seconds 120] [--seed 0]

* connected-components post-process vs the plain-C oracle on random masks of random shapes / densities / structures - as one pass and as
  1...12 strips of rows (even and random, also empty, cuts) with the seam merge;
* tiler (+ net) vs the torch-CPU oracle on random image sizes, roi sizes and overlaps with a small net;
* create_interior_map, remove_boundary_cells / label_overlap vs their oracles on random instance maps.
Prints one JSON line with the case counts; exits non-zero on the first mismatch (after printing the failing case)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np
import torch
import cellseg_amd as ca
from cellseg_amd import metrics
from oracle import postprocess_ref as P, f1_ref, label_contract_ref
from oracle.unet_ref import make_reference_unet
from oracle.sliding_window_ref import sliding_window_inference_ref



def run(seconds: float = 120.0, seed: int = 0, channels=(16, 32, 64), precision: str = "fp32") -> dict:
    """Returns the case counts; raises AssertionError (after saving the failing input under gpurun_out/) on a mismatch."""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    P.build_c()
    counts = {"ccl": 0, "ccl_strips": 0, "tiler": 0, "interior": 0, "metric": 0}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


    def random_mask(h, w):
        kind = rng.integers(0, 5)
        if kind == 0:
            return (rng.random((h, w)) < rng.uniform(0.02, 0.98)).astype(np.uint8)
        if kind == 1:
            from scipy.ndimage import gaussian_filter
            g = gaussian_filter(rng.random((h, w)), rng.uniform(0.7, 6.0))
            return (g > np.quantile(g, rng.uniform(0.1, 0.9))).astype(np.uint8)
        if kind == 2:                                             # stripes / serpentines
            m = np.zeros((h, w), np.uint8)
            step = int(rng.integers(2, 6))
            m[::step, :] = 1
            m[:, :: int(rng.integers(3, 40))] ^= 1
            return m
        if kind == 3:                                             # sparse specks near the 16 / 64 thresholds
            m = np.zeros((h, w), np.uint8)
            for _ in range(int(rng.integers(1, 60))):
                y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
                a, b = int(rng.integers(1, 10)), int(rng.integers(1, 10))
                m[y:y + a, x:x + b] = 1
            return m if rng.random() < 0.5 else 1 - m
        return np.ones((h, w), np.uint8) if rng.random() < 0.5 else np.zeros((h, w), np.uint8)


    strides = (2,) * (len(channels) - 1)
    ref = make_reference_unet(5, channels=tuple(channels), strides=strides, num_res_units=2).eval()
    hip = ca.UNet(channels=tuple(channels), strides=strides, num_res_units=2)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev).eval()
    hip.precision = precision
    mult = 1 << len(strides)                                  # window sizes must be multiples of 2^levels

    t_end = time.time() + seconds
    while time.time() < t_end:
        # ---- CC chain
        h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        m = random_mask(h, w)
        hole, mins = int(rng.choice([0, 1, 7, 64, 200])), int(rng.choice([0, 1, 16, 50]))
        want = P.mask_postprocess_c(m, hole, mins)
        got = ca.mask_to_instances(torch.from_numpy(m).to(dev), hole_area=hole, min_size=mins).cpu().numpy()
        if not np.array_equal(want, got):
            np.save(os.path.join(ROOT, "gpurun_out") + "/fuzz_fail_mask.npy", m)
            print("CCL MISMATCH", h, w, hole, mins)
            raise AssertionError("gpu_fuzz mismatch (see the line above)")
        counts["ccl"] += 1
        # ---- the same chain as strips of rows with a seam merge (cs_ccl_strip_*): random strip counts and random (also empty) cuts
        world = int(rng.integers(1, 13))
        rows = None
        if rng.random() < 0.5:
            cuts = np.sort(rng.integers(0, h + 1, size=world - 1)).tolist() if world > 1 else []
            cuts = [0] + cuts + [h]
            rows = list(zip(cuts[:-1], cuts[1:]))
        got_s = ca.mask_to_instances_strips(torch.from_numpy(m).to(dev), world, hole_area=hole, min_size=mins, rows=rows).cpu().numpy()
        if not np.array_equal(want, got_s):
            np.save(os.path.join(ROOT, "gpurun_out") + "/fuzz_fail_mask.npy", m)
            print("CCL STRIPS MISMATCH", h, w, hole, mins, world, rows)
            raise AssertionError("gpu_fuzz mismatch (see the line above)")
        counts["ccl_strips"] += 1
        # ---- interior map + metric pieces on the labelled result
        if want.max() > 0 and min(h, w) >= 6:
            inst = torch.from_numpy(want).to(dev)
            if not np.array_equal(ca.create_interior_map(inst).cpu().numpy(), label_contract_ref.create_interior_map_ref(want)):
                np.save(os.path.join(ROOT, "gpurun_out") + "/fuzz_fail_inst.npy", want)
                print("INTERIOR MISMATCH", h, w)
                raise AssertionError("gpu_fuzz mismatch (see the line above)")
            counts["interior"] += 1
            other = np.roll(want, int(rng.integers(-3, 4)), int(rng.integers(0, 2)))
            a, b = inst, torch.from_numpy(other).to(dev)
            if not np.array_equal(metrics.label_overlap(a, b), f1_ref.label_overlap_ref(want, other)) or not np.array_equal(
                    metrics.remove_boundary_cells(a).cpu().numpy(), f1_ref.remove_boundary_cells_ref(want)):
                np.save(os.path.join(ROOT, "gpurun_out") + "/fuzz_fail_inst.npy", want)
                print("METRIC MISMATCH", h, w)
                raise AssertionError("gpu_fuzz mismatch (see the line above)")
            counts["metric"] += 1
        # ---- tiler + net (every 4th round: the CPU oracle is the slow side)
        if counts["ccl"] % 4 == 0:
            roi = (int(rng.choice([32, 64, 96, 128])) // mult * mult or mult, int(rng.choice([32, 64, 96, 128])) // mult * mult or mult)
            ih, iw = int(rng.integers(8, 300)), int(rng.integers(8, 300))
            ov = float(rng.choice([0.0, 0.25, 0.5, 0.6]))
            x = torch.rand(1, 3, ih, iw, generator=torch.Generator().manual_seed(int(rng.integers(0, 1 << 30))))
            with torch.no_grad():
                want_l = sliding_window_inference_ref(x, roi, 4, ref, overlap=ov)
                got_l = ca.sliding_window_inference(x.to(dev), roi, 4, hip, overlap=ov).cpu()
            err = (want_l - got_l).abs().max().item()
            if not (err <= (0.5 if precision in ("bf16", "fp8") else 1e-4)):    # the fp32 / fp32-grade budget, or the low-precision datapaths' own
                print("TILER MISMATCH", ih, iw, roi, ov, err)
                raise AssertionError("gpu_fuzz mismatch (see the line above)")
            counts["tiler"] += 1
    return {"seed": seed, "seconds": seconds, **counts, "status": "all equal"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--channels", type=int, nargs="+", default=[16, 32, 64])
    ap.add_argument("--precision", default="fp32")
    a = ap.parse_args()
    print(json.dumps(run(a.seconds, a.seed, tuple(a.channels), a.precision)))
