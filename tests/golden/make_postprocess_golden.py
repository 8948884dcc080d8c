"""Generate tests/golden/postprocess_golden.npz with REAL scikit-image 0.18.3.

Run in the build container (not on the GPU box, not at test time):
    /opt/conda/bin/python3.9 tests/golden/make_postprocess_golden.py

For every synthetic mask it stores the input (bit-packed) and the outputs of
the exact calls the reference makes at /root/reference/baseline/predict.py:113:
    holes  = morphology.remove_small_holes(mask)            # area 64, conn 1
    objs   = morphology.remove_small_objects(holes, 16)     # conn 1
    labels = measure.label(objs)                            # 8-connected
plus measure.label(mask, connectivity=1) / (…=2) of the raw mask so the 4- and
8-connected labelers are pinned separately.  Only data is written - no
reference or scikit-image source.
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage import measure, morphology  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def blobs(rng, h, w, sigma, q):
    from scipy.ndimage import gaussian_filter
    g = gaussian_filter(rng.random((h, w)), sigma)
    return g > np.quantile(g, q)


def serpentine(h, w):
    m = np.zeros((h, w), bool)
    for y in range(0, h, 2):
        m[y, :] = True
        if y + 1 < h:
            m[y + 1, (w - 1) if (y // 2) % 2 == 0 else 0] = True
    return m


def threshold_cases():
    """Holes of area 63 / 64 / 65 (incl. border-touching), objects of size 15 / 16 / 17,
    diagonal-only touching pairs (one 8-connected, two 4-connected components)."""
    m = np.zeros((96, 160), bool)
    m[4:40, 4:150] = True
    # interior holes 7x9=63, 8x8=64, 5x13=65
    m[8:15, 8:17] = False
    m[8:16, 24:32] = False
    m[8:13, 40:53] = False
    # L-shaped hole of area 63 and 64
    m[20:27, 60:68] = False; m[26, 67] = True          # 56-1=55.. then extend
    m[27:35, 60] = False                                # +8 = 63
    m[20:27, 80:88] = False; m[27:35, 80] = False       # 56+8 = 64
    # border-touching background pockets: area 63 and 64 at the image edge
    m2 = np.zeros((96, 160), bool)
    m2[50:96, 0:160] = True
    m2[89:96, 0:9] = False                              # 7x9=63 touching bottom-left corner
    m2[88:96, 20:28] = False                            # 8x8=64 touching bottom edge
    m2[60:67, 151:160] = False                          # 63 touching right edge
    m |= m2
    # small objects in the background band rows 41..49: sizes 15, 16, 17 (bars)
    m[40:50, :] = False
    m[42, 4:19] = True                                  # 15
    m[42, 24:40] = True                                 # 16
    m[42, 44:61] = True                                 # 17
    # 4x4 = 16 block and 3x5 = 15 block
    m[44:48, 70:74] = True
    m[44:47, 80:85] = True
    # diagonal chains: two 4x2=8 blocks touching only diagonally -> each <16 (removed by 4-conn filter)
    m[44:46, 100:104] = True; m[46:48, 104:108] = True
    # two 4x4 blocks touching only at a corner: both survive, merge under 8-conn label
    m[41:45, 120:124] = True; m[45:49, 124:128] = True
    return m


def main():
    rng = np.random.default_rng(2022)
    cases = {
        "blobs_128x192": blobs(rng, 128, 192, 3.0, 0.6),
        "blobs_ragged_257x193": blobs(rng, 257, 193, 2.0, 0.5),
        "noise_p50_64x80": rng.random((64, 80)) > 0.5,
        "noise_p30_97x61": rng.random((97, 61)) > 0.7,
        "noise_p80_90x90": rng.random((90, 90)) > 0.2,
        "checkerboard_40x56": (np.indices((40, 56)).sum(0) % 2).astype(bool),
        "checker4_64x64": ((np.indices((64, 64)) // 4).sum(0) % 2).astype(bool),
        "serpentine_33x47": serpentine(33, 47),
        "all_ones_48x48": np.ones((48, 48), bool),
        "all_zeros_48x48": np.zeros((48, 48), bool),
        "single_px_1x1_on": np.ones((1, 1), bool),
        "single_px_1x1_off": np.zeros((1, 1), bool),
        "row_1x200": rng.random((1, 200)) > 0.3,
        "col_200x1": rng.random((200, 1)) > 0.3,
        "thresholds_96x160": threshold_cases(),
        "blobs_fine_300x300": blobs(rng, 300, 300, 1.2, 0.55),
        "blobs_512x512": blobs(rng, 512, 512, 4.0, 0.6),
    }
    out = {}
    for name, m in cases.items():
        m = np.ascontiguousarray(m, dtype=bool)
        holes = morphology.remove_small_holes(m)
        objs = morphology.remove_small_objects(holes, 16)
        lab = measure.label(objs)
        out[name + "/shape"] = np.array(m.shape, np.int32)
        out[name + "/mask"] = np.packbits(m)
        out[name + "/holes"] = np.packbits(holes)
        out[name + "/objs"] = np.packbits(objs)
        out[name + "/labels"] = lab.astype(np.int32)
        out[name + "/label4_raw"] = measure.label(m, connectivity=1).astype(np.int32)
        out[name + "/label8_raw"] = measure.label(m, connectivity=2).astype(np.int32)
        print(f"{name:28s} fg={int(m.sum()):7d} n4={int(out[name + '/label4_raw'].max()):5d} "
              f"n8={int(out[name + '/label8_raw'].max()):5d} final={int(lab.max()):5d}")
    np.savez_compressed(os.path.join(HERE, "postprocess_golden.npz"), **out)


if __name__ == "__main__":
    main()
