"""Golden fixtures for the image readers / writers / overlay (predict.py:88-91, :114, :118-122).

Run under the image's second interpreter, which has the reference's own I/O stack (tifffile 2021.7.2, scikit-image 0.18.3):
    /opt/conda/bin/python3.9 tests/golden/make_io_golden.py
Writes tests/golden/tiff/*.tif (small files produced by REAL tifffile: the formats the challenge data uses) and
tests/golden/io_golden.npz (what tifffile.imread returns for each, and skimage's overlay boundary masks for small label maps).
The fixtures are data (inputs + expected outputs); no reference source text is stored."""
import os

import numpy as np
import tifffile
from skimage import morphology, segmentation

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tiff")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(2022)


def blobs(h, w, n):
    lab = np.zeros((h, w), np.int32)
    for k in range(1, n + 1):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 9)
        y, x = np.ogrid[:h, :w]
        lab[(y - cy) ** 2 + (x - cx) ** 2 <= r * r] = k
    return lab


cases = {
    "u8_rgb_none": (rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), dict()),
    "u16_rgb_zlib": (rng.integers(0, 65536, (41, 29, 3), dtype=np.uint16), dict(compression="zlib")),
    "u16_gray_tiled": (rng.integers(0, 65536, (50, 70), dtype=np.uint16), dict(tile=(16, 16))),
    "u16_gray_tiled_zlib": (rng.integers(0, 4096, (50, 70), dtype=np.uint16), dict(tile=(32, 16), compression="zlib")),
    "f32_gray": (rng.random((20, 31), dtype=np.float32), dict()),
    "f32_3ch_zlib": (rng.random((20, 31, 3), dtype=np.float32), dict(compression="zlib", photometric="rgb")),
    "u8_planar": (rng.integers(0, 256, (3, 30, 40), dtype=np.uint8), dict(planarconfig="separate", photometric="rgb")),
    "u8_pages": (rng.integers(0, 256, (4, 24, 18), dtype=np.uint8), dict(photometric="minisblack")),
    "u8_stack": (rng.integers(0, 256, (3, 20, 26), dtype=np.uint8), dict(photometric="minisblack")),
    "u16_4ch": (rng.integers(0, 65536, (19, 23, 4), dtype=np.uint16), dict(photometric="rgb")),
    "u16_bigendian": (rng.integers(0, 65536, (21, 17, 3), dtype=np.uint16), dict(byteorder=">")),
    "u16_predictor": (rng.integers(0, 65536, (33, 47, 3), dtype=np.uint16), dict(compression="zlib", predictor=True)),
    "i32_label_zlib": (blobs(64, 48, 9), dict(compression="zlib")),
    "u8_bigtiff": (rng.integers(0, 256, (25, 35, 3), dtype=np.uint8), dict(bigtiff=True)),
}
golden = {}
for name, (arr, kw) in cases.items():
    path = os.path.join(OUT, name + ".tif")
    bo = kw.pop("byteorder", None)
    big = kw.pop("bigtiff", False)
    with tifffile.TiffWriter(path, byteorder=bo, bigtiff=big) as tw:
        if name == "u8_pages":
            for pg in arr:
                tw.write(pg, **kw)
        else:
            tw.write(arr, **kw)
    golden["tiff_" + name] = tifffile.imread(path)

# overlay: predict.py:119-120 with real scikit-image
for k, (h, w, n) in enumerate([(64, 64, 6), (40, 90, 12), (17, 23, 3), (8, 8, 1)]):
    lab = blobs(h, w, n)
    b = segmentation.find_boundaries(lab, connectivity=1, mode="inner")
    golden[f"overlay_labels_{k}"] = lab
    golden[f"overlay_inner_{k}"] = b
    golden[f"overlay_dilated_{k}"] = morphology.binary_dilation(b, morphology.disk(2))
golden["disk2"] = morphology.disk(2)
np.savez_compressed(os.path.join(HERE, "io_golden.npz"), **golden)
print("wrote", len(cases), "tiffs and", len(golden), "arrays")
