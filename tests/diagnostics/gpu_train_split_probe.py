#!/usr/bin/env python3
"""Probe for a batch-split training step: how long do two half-batch steps take when they run concurrently (two engine handles, two
streams, two host threads) against one full-batch step?  InstanceNorm nets only (per-image statistics: the halves are independent)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca

WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256)}
dev = torch.device("cuda:0")


def make(width, batch, precision):
    torch.manual_seed(2022)
    m = ca.UNet(channels=WIDTHS[width], strides=(2, 2, 2, 2), num_res_units=2).to(dev)
    m.precision = precision
    m.train()
    x = torch.rand(batch, 3, 256, 256, device=dev)
    t = ca.one_hot(torch.randint(0, 3, (batch, 1, 256, 256), device=dev), 3)
    return ca.TrainStep(m, ca.DiceCELoss(softmax=True), ca.AdamW(m, 6e-4)), x, t


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for width, batch, prec in (("reference16", 8, "fp32"), ("baseline32", 32, "bf16"), ("baseline32", 32, "fp32")):
    s8, x8, t8 = make(width, batch, prec)
    full = timed(lambda: s8(x8, t8))
    s4, x4, t4 = make(width, batch // 2, prec)
    half = timed(lambda: s4(x4, t4))
    # host enqueue time of one full step (no sync between steps: how far ahead of the GPU the host can run)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        s8(x8, t8)
    enq = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    # two half steps concurrently: two handles, two streams, two threads
    sa, xa, ta = make(width, batch // 2, prec)
    sb, xb, tb = make(width, batch // 2, prec)
    st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    N = 40

    def worker(step, x, t, stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                step(x, t)

    for n in (5, N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(sa, xa, ta, st[0], n)), threading.Thread(target=worker, args=(sb, xb, tb, st[1], n))]
        [t_.start() for t_ in th]
        [t_.join() for t_ in th]
        torch.cuda.synchronize()
        pair = (time.perf_counter() - t0) / n * 1e3
    print(f"{width} x {batch} {prec}: full step {full:.3f} ms (host enqueue {enq:.3f} ms), half-batch step {half:.3f} ms, two half-batch steps side by side {pair:.3f} ms per pair")
