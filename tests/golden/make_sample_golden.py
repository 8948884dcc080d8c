"""Generate tests/golden/sample_golden.npz: the reference's OWN functions run on the reference's own three sample pairs
(/root/reference/data/Train_Labeled/{images,labels}: cell_00225.png 1024x1024 RGB uint8, cell_00528.tif 200x200 gray uint16 - the
pad-to-256 path of the tiler - and cell_00849.png 512x512 RGB uint8 with 187 cells).  They are the only real inputs the reference ships.

Run in the build container only (the reference does not exist on the GPU box):
    /opt/conda/bin/python3.9 tests/golden/make_sample_golden.py

Per sample the file holds INPUT arrays (the decoded image - cell_00225 as its central 512 x 512 crop, to keep the fixture small - and the
int16 instance map) and the OUTPUT arrays of the imported reference code:
  <stem>/pre       uint8 [H,W,3]   data/pre_process_3class.py:84-95 (channel fix-up + normalize_channel per channel)
                                   = baseline/predict.py:94-104, the same text
  <stem>/interior  uint8 [H,W]     data/pre_process_3class.py:98  create_interior_map(gt.astype(int16))
  <stem>/seg       int32 [H,W]     measure.label(interior == 1): the interiors as a 'segmentation' 
  <stem>/tpfpfn    int64 [3]       baseline/compute_metric.py eval_tp_fp_fn(gt, seg, 0.5)           (whole image, :107-122 without the
  <stem>/tpfpfn_rb int64 [3]       ... after remove_boundary_cells on both                              boundary rule / with it, :225-241)
  <stem>/label_m   int32 [H,W]     measure.label(morphology.remove_small_objects(morphology.remove_small_holes(interior == 1), 16)):
                                   real scikit-image 0.18.3 on a REAL interior mask - the post-process of baseline/predict.py:113
Only arrays are stored - no reference source text.  numba is broken in this image: an identity ``jit`` stub is injected (the
decorated function is plain Python), as in make_f1_norm_golden.py.
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
import tifffile as tif               # noqa: E402
from skimage import io, measure, morphology   # noqa: E402


def main():
    out = {}
    root = os.path.join(REF, "data", "Train_Labeled")
    for name in sorted(os.listdir(os.path.join(root, "images"))):
        stem = name.split(".")[0]
        path = os.path.join(root, "images", name)
        img = tif.imread(path) if name.endswith((".tif", ".tiff")) else io.imread(path)
        gt = tif.imread(os.path.join(root, "labels", stem + "_label.tiff"))
        if img.shape[0] > 512:               # fixture size: the 1024 x 1024 photograph enters the normalisation as its central 512 x 512 crop
            o = (img.shape[0] - 512) // 2      # (the instance map, interior map and F1 counts below are those of the whole image)
            img = np.ascontiguousarray(img[o:o + 512, o:o + 512])
        data = img
        if data.ndim == 2:
            data = np.repeat(np.expand_dims(data, axis=-1), 3, axis=-1)
        elif data.ndim == 3 and data.shape[-1] > 3:
            data = data[:, :, :3]
        pre = np.zeros(data.shape, dtype=np.uint8)
        for i in range(3):
            ch = data[:, :, i]
            if len(ch[np.nonzero(ch)]) > 0:
                pre[:, :, i] = pp.normalize_channel(ch, lower=1, upper=99)
        interior = pp.create_interior_map(gt.astype(np.int16))
        seg = measure.label(interior == 1).astype(np.int32)
        tp, fp, fn = cm.eval_tp_fp_fn(gt, seg, threshold=0.5)
        gt_rb, seg_rb = cm.remove_boundary_cells(gt.copy()), cm.remove_boundary_cells(seg.copy())
        tp2, fp2, fn2 = cm.eval_tp_fp_fn(gt_rb, seg_rb, threshold=0.5)
        label_m = measure.label(morphology.remove_small_objects(morphology.remove_small_holes(interior == 1), 16)).astype(np.int32)
        out[stem + "/img"] = img
        out[stem + "/gt"] = gt.astype(np.int16)
        out[stem + "/pre"] = pre
        out[stem + "/interior"] = interior.astype(np.uint8)
        out[stem + "/seg"] = seg
        out[stem + "/tpfpfn"] = np.array([tp, fp, fn], np.int64)
        out[stem + "/tpfpfn_rb"] = np.array([tp2, fp2, fn2], np.int64)
        out[stem + "/label_m"] = label_m
        print(stem, img.shape, img.dtype, "cells", int(gt.max()), "tp/fp/fn", (tp, fp, fn), "after boundary rule", (tp2, fp2, fn2),
              "interior classes", np.bincount(interior.ravel(), minlength=3), "label_m max", int(label_m.max()))
    np.savez_compressed(os.path.join(HERE, "sample_golden.npz"), **out)
    print("wrote sample_golden.npz,", os.path.getsize(os.path.join(HERE, "sample_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
