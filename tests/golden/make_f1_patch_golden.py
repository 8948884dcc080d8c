"""Generate tests/golden/f1_patch_golden.npz: the reference's large-image (>= 25 Mpx) patch-wise F1 bookkeeping.

Run in the build container only:   /opt/conda/bin/python3.9 tests/golden/make_f1_patch_golden.py

The reference does the patch walk inline in compute_metric.py main() (:191-232) with roi_size = 2000, so it cannot be
called as a function.  This script drives the reference's OWN per-patch functions (remove_boundary_cells, eval_tp_fp_fn,
imported from /root/reference/baseline/compute_metric.py, and skimage's relabel_sequential) over zero-padded patches of a
configurable edge, so that small fixtures exercise the same arithmetic: pad to a multiple of the patch edge, per patch
drop boundary cells, relabel, count cells, match, and sum tp / fp / fn / cell counts.
Only inputs and expected outputs are stored.
"""
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
numba_stub = types.ModuleType("numba")
numba_stub.jit = lambda *a, **k: (lambda f: f)
sys.modules["numba"] = numba_stub
sys.path.insert(0, "/root/reference/baseline")
import compute_metric as cm                      # noqa: E402
from skimage import measure, segmentation        # noqa: E402
from scipy.ndimage import gaussian_filter        # noqa: E402


def patch_counts(gt, seg, edge):
    h, w = gt.shape
    ph, pw = -(-h // edge) * edge, -(-w // edge) * edge
    g = np.zeros((ph, pw), gt.dtype); g[:h, :w] = gt
    s = np.zeros((ph, pw), gt.dtype); s[:h, :w] = seg
    tot = np.zeros(5, np.int64)                  # tp, fp, fn, true cells, predicted cells
    for y in range(0, ph, edge):
        for x in range(0, pw, edge):
            a = cm.remove_boundary_cells(g[y:y + edge, x:x + edge])
            b = cm.remove_boundary_cells(s[y:y + edge, x:x + edge])
            a = segmentation.relabel_sequential(a)[0]
            b = segmentation.relabel_sequential(b)[0]
            tp, fp, fn = cm.eval_tp_fp_fn(a, b, threshold=0.5)
            tot += np.array([tp, fp, fn, a.max(), b.max()], np.int64)
    return tot


def main():
    rng = np.random.default_rng(77)
    out = {}
    for k, (h, w, edge) in enumerate([(250, 330, 100), (200, 200, 100), (130, 90, 64)]):
        f = gaussian_filter(rng.random((h, w)), 2.5)
        gt = measure.label(f > np.quantile(f, 0.55)).astype(np.int32)
        seg = np.roll(gt, (1, -2), axis=(0, 1)).copy()
        for i in np.unique(seg)[1:]:
            if rng.random() < 0.15:
                seg[seg == i] = 0
        seg[rng.random(seg.shape) < 0.03] = 0
        seg = measure.label(seg > 0).astype(np.int32)
        out[f"c{k}/gt"], out[f"c{k}/seg"] = gt, seg
        out[f"c{k}/edge"] = np.array(edge)
        out[f"c{k}/totals"] = patch_counts(gt, seg, edge)
        print(k, (h, w, edge), out[f"c{k}/totals"])
    np.savez_compressed(os.path.join(HERE, "f1_patch_golden.npz"), **out)


if __name__ == "__main__":
    main()
