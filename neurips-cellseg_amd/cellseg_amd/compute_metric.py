"""F1 scoring CLI with the reference's flags and CSV columns (/root/reference/baseline/compute_metric.py:135-262).

    python -m cellseg_amd.compute_metric -g <gt dir> -s <seg dir> [--gt_suffix .tif] [--seg_suffix _label.tiff]
                                         [-thre 0.5 ...] [-o ./] [-n demo] [--count_bd_cells]

Per image: dice of the foreground masks, then F1 at each IoU threshold; the label-sized work (boundary-cell removal,
relabelling, the overlap histogram) runs on the GPU, matching stays on the host (cellseg_amd.metrics).
"""
from __future__ import annotations

import argparse
import csv
import os
import traceback

import numpy as np
import torch

from . import metrics
from .preprocess import read_image

COLUMNS = ["names", "true_num", "pred_num", "correct_num(TP)", "missed_num(FN)", "wrong_num(FP)", "precision", "recall", "dice", "F1"]


def dice(gt: torch.Tensor, seg: torch.Tensor) -> float:
    """compute_metric.py:72-81 on boolean foreground masks."""
    g, s = int(gt.count_nonzero()), int(seg.count_nonzero())
    if g == 0 and s == 0:
        return 1.0
    if g == 0:
        return 0.0
    return 2.0 * int((gt & seg).count_nonzero()) / (g + s)


def score_pair(gt: np.ndarray, seg: np.ndarray, threshold: float, count_bd_cells: bool, device) -> dict:
    g = torch.from_numpy(np.ascontiguousarray(gt.astype(np.int32))).to(device)
    s = torch.from_numpy(np.ascontiguousarray(seg.astype(np.int32))).to(device)
    d = dice(g > 0, s > 0)
    tp, fp, fn, n_true, n_pred = metrics.cell_counts(g, s, threshold, count_bd_cells)
    precision, recall, f1 = metrics.f1_from_counts(tp, n_true, n_pred)
    return {"true_num": n_true, "pred_num": n_pred, "correct_num(TP)": tp, "missed_num(FN)": fn, "wrong_num(FP)": fp,
            "precision": np.round(precision, 4), "recall": np.round(recall, 4), "dice": np.round(d, 4), "F1": np.round(f1, 4)}


def main(argv=None):
    ap = argparse.ArgumentParser("Compute F1 score for cell segmentation results")
    ap.add_argument("-g", "--gt_path", default="labelsTr_GT", type=str)
    ap.add_argument("-s", "--seg_path", default="", type=str)
    ap.add_argument("--gt_suffix", default=".tif", type=str)
    ap.add_argument("--seg_suffix", default="_label.tiff", type=str)
    ap.add_argument("-thre", "--thresholds", nargs="+", default=[0.5], type=float)
    ap.add_argument("-o", "--output_path", default="./", type=str)
    ap.add_argument("-n", "--save_name", default="demo", type=str)
    ap.add_argument("--count_bd_cells", default=False, action="store_true")
    args = ap.parse_args(argv)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    names = sorted(n for n in os.listdir(args.seg_path) if n.endswith(args.seg_suffix))
    results = {}
    for threshold in args.thresholds:
        print("compute metrics at threshold:", threshold)
        rows, failed = [], []
        for name in names:
            try:
                gt_name = name.split(args.seg_suffix)[0] + args.gt_suffix
                gt = read_image(os.path.join(args.gt_path, gt_name))
                seg = read_image(os.path.join(args.seg_path, name))
                rows.append({"names": name, **score_pair(gt, seg, threshold, args.count_bd_cells, device)})
            except Exception:
                print("!" * 20)
                print(name, "evaluation error!")
                traceback.print_exc()
                failed.append(name)
        save_name = args.save_name[:-4] if args.save_name.endswith(".csv") else args.save_name
        path = os.path.join(args.output_path, f"{save_name}-{threshold}.csv")
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=COLUMNS)
            w.writeheader()
            w.writerows(rows)
        f1s = [r["F1"] for r in rows]
        print("threshold:", threshold, "mean F1 Score:", np.mean(f1s) if f1s else float("nan"),
              "median F1 Score:", np.median(f1s) if f1s else float("nan"))
        print("failed cases:", failed)
        results[threshold] = rows
    return results


if __name__ == "__main__":
    main()
