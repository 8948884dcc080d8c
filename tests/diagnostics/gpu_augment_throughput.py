#!/usr/bin/env python3
"""Diagnostic (GPU): samples/s of the device augmentation chain (cellseg_amd.gpu_data) beside the host chain (cellseg_amd.data, one
process), and images/s of the reference-setting training step fed by it.

    python tests/diagnostics/gpu_augment_throughput.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np
import torch
import cellseg_amd as ca
from cellseg_amd import data, gpu_data

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
samples = []
for k in range(64):
    h, w = int(rng.integers(600, 1100)), int(rng.integers(600, 1100))
    samples.append((rng.random((3, h, w), dtype=np.float32), (rng.random((1, h, w)) * 3).astype(np.uint8)))
ds = gpu_data.DeviceCellDataset(samples, dev)
out = {"dataset": {"images": len(ds), "MiB_in_HBM": round(ds.nbytes() / 2 ** 20, 1)}}
for bs in (8, 32):
    aug = gpu_data.GpuAugmenter(ds, 256, seed=1)
    order = [int(i) for i in rng.integers(0, len(ds), bs * 60)]
    for b in range(0, bs * 10, bs):
        aug.batch(order[b:b + bs])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(bs * 10, bs * 60, bs):
        aug.batch(order[b:b + bs])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    params = aug.draw(order[:bs])
    e0.record()
    for _ in range(20):
        aug.batch(order[:bs], params)
    e1.record()
    torch.cuda.synchronize()
    out[f"batch_{bs}"] = {"samples_per_s_incl_host_draws": round(bs * 50 / dt), "ms_per_batch_incl_host_draws": round(dt / 50 * 1e3, 3),
                          "ms_per_batch_stream_time": round(e0.elapsed_time(e1) / 20, 3)}
hds = data.CellDataset([("a", "b")], 256, True, seed=1)
t0 = time.perf_counter()
for k in range(200):
    hds._augment(*samples[k % len(samples)])
out["host_chain_samples_per_s_one_process"] = round(200 / (time.perf_counter() - t0))
model = ca.UNet(spatial_dims=2, in_channels=3, out_channels=3, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
aug = gpu_data.GpuAugmenter(ds, 256, seed=2)
order = [int(i) for i in rng.integers(0, len(ds), 8 * 120)]
for b in range(0, 160, 8):
    img, lab = aug.batch(order[b:b + 8])
    step(img, ca.one_hot(lab, 3))
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(160, 960, 8):
    img, lab = aug.batch(order[b:b + 8])
    loss = step(img, ca.one_hot(lab, 3))
    loss.item()                                    # the reference loop reads the loss every step (model_training_3class.py:263)
dt = time.perf_counter() - t0
out["training_loop_batch_8_with_loss_item"] = {"images_per_s": round(800 / dt), "ms_per_step": round(dt / 100 * 1e3, 3)}
acc = torch.zeros((), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(160, 960, 8):
    img, lab = aug.batch(order[b:b + 8])
    acc += step(img, ca.one_hot(lab, 3)).double()   # what cellseg_amd.model_training_3class does: the epoch's sum stays on the device
acc.item()
dt = time.perf_counter() - t0
out["training_loop_batch_8_loss_summed_on_device"] = {"images_per_s": round(800 / dt), "ms_per_step": round(dt / 100 * 1e3, 3)}
print(json.dumps(out, indent=1))
