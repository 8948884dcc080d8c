"""GPU tests of the device-side label-map writer (csrc/kernels_deflate.hip).

The contract is the reference's `tif.imwrite(<name>_label.tiff, labels, compression='zlib')` (baseline/predict.py:114): an int32
deflate TIFF whose pixels are exactly the label map.  The checker is an independent inflate: every strip the device produced goes
through `zlib.decompress` (which also verifies the Adler-32 the device combined) and must give back the strip's bytes, bit for
bit; whole files are read back with the package's reader and with Pillow.

Run on the MI355X box with ``pytest -m gpu``.  Everything goes through the C-ABI (libcellseg_hip.so)."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips-cellseg_amd"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _cells(h, w, seed, n_cells=None):
    """A label map that looks like the path's output: numbered discs on background 0."""
    rng = np.random.default_rng(seed)
    lab = np.zeros((h, w), np.int32)
    n = n_cells if n_cells is not None else max(1, h * w // 2000)
    yy, xx = np.mgrid[0:h, 0:w]
    for k in range(1, min(n, 400) + 1):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 14)
        lab[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = k
    return lab


def _strips_decode(lab, rps):
    from cellseg_amd.preprocess import deflate_labels_gpu
    t = torch.from_numpy(lab).to(_dev())
    buf, sizes = deflate_labels_gpu(t, rps)
    raw = buf.numpy().tobytes()
    h = lab.shape[0]
    rps = max(1, min(rps, h))
    assert len(sizes) == (h + rps - 1) // rps
    pos, total = 0, 0
    for s, sz in enumerate(sizes):
        assert pos % 2 == 0, "strips start on even offsets"
        stream = raw[pos:pos + sz]
        assert stream[:2] == b"\x78\x01"
        want = lab[s * rps:(s + 1) * rps].astype("<i4").tobytes()
        got = zlib.decompress(stream)                     # raises on a bad Adler-32 or a malformed block
        assert got == want, f"strip {s} of {lab.shape} at {rps} rows per strip does not decode to the labels"
        pos += sz + (sz & 1)
        total += sz
    assert pos == len(raw)
    return total


@pytest.mark.parametrize("shape,rps", [((1, 1), 64), ((3, 5), 2), ((64, 1024), 64), ((65, 1025), 64), ((200, 3000), 7),
                                       ((257, 2049), 1), ((130, 700), 1000), ((512, 512), 64)])
def test_every_strip_inflates_to_the_labels(shape, rps):
    total = _strips_decode(_cells(*shape, seed=shape[0] + shape[1]), rps)
    assert total > 0


def test_background_only_and_single_label_maps():
    for value in (0, 1, 143, 144, 255, 256, 70000, 2 ** 31 - 1):
        lab = np.full((70, 2500), value, np.int32)
        total = _strips_decode(lab, 64)
        assert total < lab.nbytes // 50                  # pure runs: 13 bits per 258 bytes + the chunk markers


def test_worst_case_labels_fit_the_bound():
    """Every label differs from its neighbour and every byte takes a 9-bit literal: the scratch slot and the output bound hold."""
    rng = np.random.default_rng(5)
    lab = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(96, 2048), dtype=np.int64).astype(np.int32)
    lab |= np.int32(-0x6F6F6F70)                          # 0x90909090: all four bytes >= 144
    total = _strips_decode(lab, 32)
    from cellseg_amd._lib import lib
    assert lab.nbytes < total <= lib.cs_label_deflate_bound(96, 2048, 32)
    # alternating two labels (runs of length 1), negative labels, the shortest runs that need the tail rule (260 bytes = 256 + 4)
    alt = np.tile(np.array([7, 9], np.int32), (33, 600))
    _strips_decode(alt, 5)
    _strips_decode(-_cells(80, 900, 3), 16)
    for run in (2, 3, 64, 65, 66, 67, 129, 130, 131, 1023, 1024):
        row = np.repeat(np.arange(1, 40, dtype=np.int32), run)[None, :]
        _strips_decode(np.repeat(row, 3, 0), 2)


def test_label_tiff_written_from_the_device_reads_back_identically(tmp_path):
    from PIL import Image

    from cellseg_amd.preprocess import read_image, write_label_tiff
    Image.MAX_IMAGE_PIXELS = None
    for shape, rps in [((300, 1500), 64), ((1000, 1111), 100), ((5, 7), 64)]:
        lab = _cells(*shape, seed=11)
        p = str(tmp_path / f"l{shape[0]}_label.tiff")
        write_label_tiff(p, torch.from_numpy(lab).to(_dev()), rows_per_strip=rps)
        got = read_image(p)
        assert got.dtype == np.int32 and np.array_equal(got, lab)
        with Image.open(p) as im:
            assert im.mode == "I" and np.array_equal(np.array(im), lab)
        # the host writer of the same array gives a file with the same pixels (different deflate bytes are fine)
        q = str(tmp_path / "host_label.tiff")
        write_label_tiff(q, lab, rows_per_strip=rps)
        assert np.array_equal(read_image(q), got)
        # int64 labels on the device are narrowed like the host path does
        write_label_tiff(p, torch.from_numpy(lab.astype(np.int64)).to(_dev()), rows_per_strip=rps)
        assert np.array_equal(read_image(p), lab)


def test_full_slide_label_map_round_trip_and_size():
    """BASELINE configs[1] size: 4096^2 labels.  Round trip through zlib per strip; a cell map shrinks to a few percent."""
    lab = np.tile(_cells(1024, 1024, seed=2, n_cells=400), (4, 4))
    lab[lab > 0] += (np.arange(16, dtype=np.int32).reshape(4, 4).repeat(1024, 0).repeat(1024, 1) * 400)[lab > 0]
    total = _strips_decode(lab, 64)
    assert total < lab.nbytes * 0.04, f"{total} bytes for a {lab.nbytes}-byte label map"


def test_deflate_rejects_bad_arguments():
    from cellseg_amd.preprocess import deflate_labels_gpu
    with pytest.raises(ValueError):
        deflate_labels_gpu(torch.zeros((4, 4), dtype=torch.int64, device=_dev()))
    with pytest.raises(ValueError):
        deflate_labels_gpu(torch.zeros((0, 4), dtype=torch.int32, device=_dev()))
    with pytest.raises(Exception):
        deflate_labels_gpu(torch.zeros((4, 4), dtype=torch.int32))          # host tensor: the device encoder refuses it
