#!/usr/bin/env python3
"""Tile-throughput / training-step benchmark (BASELINE.json configs[1] and [3] shapes): batch x 3 x 256 x 256 tiles.

    python tools/bench_train.py [--batch 32] [--steps 10] [--width reference16|baseline32]

Prints one JSON line: forward-only tile throughput (Mpixels/s of computed window pixels) and the full training step
(forward + DiceCE + backward + AdamW) in ms and images/s, with the engine's per-kernel breakdown (HIP events)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca
from cellseg_amd import _lib

WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--width", choices=sorted(WIDTHS), default="reference16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(2022)
    model = ca.UNet(channels=WIDTHS[args.width], strides=(2, 2, 2, 2), num_res_units=2).to(dev)
    x = torch.rand(args.batch, 3, 256, 256, device=dev)
    y = torch.randint(0, 3, (args.batch, 1, 256, 256), device=dev)
    loss_fn, opt = ca.DiceCELoss(softmax=True), torch.optim.AdamW(model.parameters(), 6e-4)
    onehot = ca.one_hot(y, 3)

    def fwd():
        with torch.no_grad():
            return model(x)

    def step():
        opt.zero_grad()
        loss = loss_fn(model(x), onehot)
        loss.backward()
        opt.step()
        return loss

    def timeit(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    model.eval(); t_fwd = timeit(fwd, args.steps)
    model.train(); t_step = timeit(step, args.steps)
    _lib.lib.cs_prof_reset(); _lib.lib.cs_prof_enable(1)
    timeit(step, args.steps)
    _lib.lib.cs_prof_enable(0)
    prof = _lib.prof_collect()
    tot = sum(v["ms"] for v in prof.values())
    n_runs = args.steps + 2
    kern = {k: {"ms_per_step": round(v["ms"] / n_runs, 3), "share": round(v["ms"] / tot, 3),
                "tflops": round(v["flops"] / v["ms"] / 1e9, 1) if v["flops"] else None}
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]}
    mpx = args.batch * 256 * 256 / 1e6
    print(json.dumps({"width": args.width, "batch": args.batch, "forward_ms": round(t_fwd * 1e3, 3),
                      "forward_tile_Mpixels_per_s": round(mpx / t_fwd, 1), "train_step_ms": round(t_step * 1e3, 3),
                      "train_images_per_s": round(args.batch / t_step, 1), "engine_gpu_ms_per_step": round(tot / n_runs, 3),
                      "kernels": kern}))


if __name__ == "__main__":
    main()
