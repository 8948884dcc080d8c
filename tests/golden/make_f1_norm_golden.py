"""Generate tests/golden/{f1,normalize,label_contract}_golden.npz by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):
    /opt/conda/bin/python3.9 tests/golden/make_f1_norm_golden.py

* f1_golden.npz            <- /root/reference/baseline/compute_metric.py
                              (eval_tp_fp_fn, remove_boundary_cells, dice); numba is
                              broken in this image, so an identity ``jit`` stub is
                              injected (the decorated function is plain Python).
* normalize_golden.npz     <- /root/reference/data/pre_process_3class.py:22-29
                              normalize_channel (same text as baseline/predict.py:14-21)
* label_contract_golden.npz<- /root/reference/data/pre_process_3class.py:31-56
                              create_interior_map
Only inputs and expected outputs are stored - no reference source.
"""
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

numba_stub = types.ModuleType("numba")
numba_stub.jit = lambda *a, **k: (lambda f: f)
sys.modules["numba"] = numba_stub
sys.path.insert(0, os.path.join(REF, "baseline"))
sys.path.insert(0, os.path.join(REF, "data"))
import compute_metric as cm          # noqa: E402
import pre_process_3class as pp      # noqa: E402
from skimage import measure          # noqa: E402


def random_instances(rng, h, w, sigma, q):
    from scipy.ndimage import gaussian_filter
    g = gaussian_filter(rng.random((h, w)), sigma)
    return measure.label(g > np.quantile(g, q)).astype(np.int32)


def perturb(rng, lab, drop=0.15, shift=2, split=True):
    """A plausible 'segmentation': shifted, some cells dropped, some merged/split."""
    seg = np.roll(lab, (shift, -shift), axis=(0, 1)).copy()
    ids = np.unique(seg)[1:]
    for i in ids:
        if rng.random() < drop:
            seg[seg == i] = 0
    if split and len(ids) > 2:
        i = ids[len(ids) // 2]
        ys, xs = np.nonzero(seg == i)
        if len(ys):
            cut = ys > np.median(ys)
            seg[ys[cut], xs[cut]] = seg.max() + 1
    # erode a few px randomly
    seg[rng.random(seg.shape) < 0.03] = 0
    return seg.astype(np.int32)


def main():
    rng = np.random.default_rng(2022)
    # ------------------------------------------------------------------ F1
    f1 = {}
    cases = []
    for k, (h, w, s, q) in enumerate([(96, 128, 3.0, 0.6), (200, 200, 2.5, 0.55), (64, 64, 2.0, 0.7),
                                      (150, 90, 4.0, 0.5)]):
        gt = random_instances(rng, h, w, s, q)
        cases.append((f"synth{k}", gt, perturb(rng, gt)))
    gt = random_instances(rng, 80, 80, 3.0, 0.6)
    cases.append(("identical", gt, gt.copy()))
    cases.append(("empty_seg", gt, np.zeros_like(gt)))
    cases.append(("empty_gt", np.zeros_like(gt), gt))
    for name, gt, seg in cases:
        tp, fp, fn = cm.eval_tp_fp_fn(gt, seg, threshold=0.5)
        gt_rb = cm.remove_boundary_cells(gt.copy())
        seg_rb = cm.remove_boundary_cells(seg.copy())
        tp2, fp2, fn2 = cm.eval_tp_fp_fn(gt_rb, seg_rb, threshold=0.5)
        f1[name + "/gt"] = gt
        f1[name + "/seg"] = seg
        f1[name + "/tpfpfn"] = np.array([tp, fp, fn], np.int64)
        f1[name + "/gt_rb"] = gt_rb.astype(np.int32)
        f1[name + "/seg_rb"] = seg_rb.astype(np.int32)
        f1[name + "/tpfpfn_rb"] = np.array([tp2, fp2, fn2], np.int64)
        f1[name + "/dice"] = np.array(cm.dice(gt, seg), np.float64)
        print("f1", name, (tp, fp, fn), (tp2, fp2, fn2), float(f1[name + "/dice"]))
    np.savez_compressed(os.path.join(HERE, "f1_golden.npz"), **f1)

    # ----------------------------------------------------------- normalise
    nm = {}
    from scipy.ndimage import gaussian_filter
    a8 = (gaussian_filter(rng.random((70, 90)), 2.0) * 255).astype(np.uint8)
    a8[:10, :10] = 0
    a16 = (gaussian_filter(rng.random((64, 48)), 1.5) * 40000 + 300).astype(np.uint16)
    a16[5:9, :] = 0
    flat = np.full((32, 32), 7, np.uint8); flat[0, 0] = 0          # p99 - p1 <= 0.001 branch
    flat16 = np.full((16, 16), 300, np.uint16)                       # else-branch astype(uint8) wrap quirk
    f32 = gaussian_filter(rng.random((40, 40)), 1.0).astype(np.float32)
    for name, a in [("u8", a8), ("u16", a16), ("flat_u8", flat), ("flat_u16", flat16), ("f32", f32)]:
        nm[name + "/in"] = a
        nm[name + "/out"] = pp.normalize_channel(a, lower=1, upper=99)
        print("norm", name, nm[name + "/out"].dtype, int(nm[name + "/out"].min()), int(nm[name + "/out"].max()))
    np.savez_compressed(os.path.join(HERE, "normalize_golden.npz"), **nm)

    # ------------------------------------------------------ label contract
    lc = {}
    for k, (h, w, s, q) in enumerate([(96, 128, 3.0, 0.6), (120, 80, 2.0, 0.55), (64, 64, 5.0, 0.5)]):
        inst = random_instances(rng, h, w, s, q).astype(np.int16)
        lc[f"c{k}/inst"] = inst
        lc[f"c{k}/interior"] = pp.create_interior_map(inst)
        print("label", k, np.bincount(lc[f"c{k}/interior"].ravel(), minlength=3))
    # touching instances (shared borders) stress find_boundaries(mode='inner')
    inst = np.zeros((40, 60), np.int16)
    inst[5:25, 5:25] = 1; inst[5:25, 25:45] = 2; inst[25:38, 10:50] = 3; inst[2:4, 50:58] = 4
    lc["touch/inst"] = inst
    lc["touch/interior"] = pp.create_interior_map(inst)
    np.savez_compressed(os.path.join(HERE, "label_contract_golden.npz"), **lc)


if __name__ == "__main__":
    main()
