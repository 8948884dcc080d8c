"""bench.py --full: what the low-precision datapaths do to the RESULT, on a net trained on synthetic cells."""
from __future__ import annotations

import torch

from .common import WIDTHS

def lowprec_accuracy(dev, steps=150, n_images=8, size=1024, precisions=("fp16x3", "bf16x6", "bf16", "fp8"), seed=7, width="baseline32", roi=256, per_image=False):
    """Accuracy of the low-precision datapaths where it means something: on a TRAINED net (random-init logits sit at the 0.5
    threshold).  A 32-base U-Net is trained for `steps` AdamW steps on seeded synthetic cells (fp32 engine, ca.TrainStep), then
    `n_images` held-out synthetic images of size^2 are segmented in fp32 and in every mode: instance F1@0.5 against the fp32
    instances (compute_metric.py semantics), pixel agreement of the interior masks, and F1 against the synthetic ground truth
    (the reference's accuracy target is "F1 within 1e-3 of the reference path")."""
    import cellseg_amd as ca
    from cellseg_amd import data, metrics
    torch.manual_seed(seed)
    model = ca.UNet(channels=WIDTHS[width], strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
    ds = data.SyntheticCells(64, 256, 2022)
    batches = [(torch.stack([ds[i]["img"] for i in range(b, b + 8)]).to(dev),
                ca.one_hot(torch.stack([ds[i]["label"] for i in range(b, b + 8)]).to(dev), 3)) for b in range(0, 64, 8)]
    step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
    first = last = None
    for it in range(steps):
        loss = step(*batches[it % len(batches)])
        if it == 0:
            first = loss.item()
    last = loss.item()
    model.eval()
    held = data.SyntheticCells(n_images, size, 9900)
    sums = {p: dict(f1_vs_fp32=0.0, pixel_agreement=0.0, f1_vs_truth=0.0) for p in precisions}
    deltas = {p: [] for p in precisions}                    # per held-out image: F1 vs truth in the mode - F1 vs truth in fp32
    f1_truth_fp32, cells = 0.0, 0
    with torch.no_grad():
        for i in range(n_images):
            item = held[i]
            img = item["img"].to(dev)
            truth = ca.label((item["label"][0] == 1).to(dev).to(torch.uint8), connectivity=1)      # ground-truth interiors as instances
            res = {}
            for prec in ("fp32",) + tuple(precisions):
                model.precision = prec
                _, _, mask = ca.swi_infer_image(model, img, (roi, roi), 0.5, want_logits=False, want_mask=True)
                res[prec] = (mask.clone(), ca.mask_to_instances(mask))
            cells += int(res["fp32"][1].max())
            f1_32 = metrics.f1_score(truth, res["fp32"][1])[5]
            f1_truth_fp32 += f1_32
            for prec in precisions:
                sums[prec]["f1_vs_fp32"] += metrics.f1_score(res["fp32"][1], res[prec][1])[5]
                sums[prec]["pixel_agreement"] += float((res["fp32"][0] == res[prec][0]).float().mean())
                f1_p = metrics.f1_score(truth, res[prec][1])[5]
                sums[prec]["f1_vs_truth"] += f1_p
                deltas[prec].append(f1_p - f1_32)
    model.precision = "fp32"
    out = {"trained_net": {"channels": list(WIDTHS[width]), "window": roi, "steps": steps, "loss_first": round(first, 4), "loss_last": round(last, 4),
                           "held_out_images": n_images, "image_size": size, "cells_found_fp32": cells,
                           "f1_vs_truth_fp32": round(f1_truth_fp32 / n_images, 5)}}
    for prec in precisions:
        out[prec] = {k: round(v / n_images, 5) for k, v in sums[prec].items()}
        out[prec]["delta_f1_vs_truth"] = round(out[prec]["f1_vs_truth"] - out["trained_net"]["f1_vs_truth_fp32"], 5)
        d = torch.tensor(deltas[prec], dtype=torch.float64)
        out[prec]["delta_f1_mean"] = round(float(d.mean()), 6)
        out[prec]["delta_f1_se"] = round(float(d.std(unbiased=True) / len(d) ** 0.5), 6) if len(d) > 1 else None
        if per_image:
            out[prec]["delta_f1_per_image"] = [round(float(v), 6) for v in d]
    return out

