"""TEST / BENCH INFRASTRUCTURE (oracle side) - times the reference's own post-process chain with REAL scikit-image.

Run under the image's second interpreter (/opt/conda/bin/python3.9: scikit-image 0.18.3, no torch):
    python3.9 oracle/skimage_chain.py mask.npy  ->  one JSON line {"seconds": ..., "n_labels": ...}
The chain is /root/reference/baseline/predict.py:113 applied to an already thresholded mask:
    measure.label(morphology.remove_small_objects(morphology.remove_small_holes(mask), 16))
Only bench.py's cpu_baseline leg executes this file (BASELINE.md 3(iii) prescribes the skimage chain for the CPU leg)."""
import json
import sys
import time

import numpy as np
from skimage import measure, morphology


def main():
    mask = np.load(sys.argv[1]).astype(bool)
    t0 = time.perf_counter()
    lab = measure.label(morphology.remove_small_objects(morphology.remove_small_holes(mask), 16))
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "n_labels": int(lab.max()), "skimage": __import__("skimage").__version__}))


if __name__ == "__main__":
    main()
