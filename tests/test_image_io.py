"""Image readers / writers / overlay against fixtures produced by the reference's own I/O stack (real tifffile 2021.7.2 and
scikit-image 0.18.3; tests/golden/make_io_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from cellseg_amd import preprocess as P

G = os.path.join(os.path.dirname(__file__), "golden")
GOLD = np.load(os.path.join(G, "io_golden.npz"))
TIFFS = sorted(k[5:] for k in GOLD.files if k.startswith("tiff_"))


@pytest.mark.parametrize("name", TIFFS)
def test_own_tiff_reader_equals_tifffile(name):
    want = GOLD["tiff_" + name]
    got = P.read_tiff(os.path.join(G, "tiff", name + ".tif"))
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", TIFFS)
def test_read_image_dispatch(name):
    assert np.array_equal(P.read_image(os.path.join(G, "tiff", name + ".tif")), GOLD["tiff_" + name])


def test_read_image_failure_names_the_file(tmp_path):
    bad = tmp_path / "broken.tif"
    bad.write_bytes(b"II*\0garbage")
    with pytest.raises(OSError, match="broken.tif"):
        P.read_image(str(bad))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int32, np.float32])
@pytest.mark.parametrize("shape", [(37, 53), (37, 53, 3), (70, 20, 4), (1, 1), (130, 7, 3)])
def test_tiff_writer_round_trip(tmp_path, dtype, shape):
    a = (np.random.default_rng(1).random(shape) * 200).astype(dtype)
    P.write_tiff(str(tmp_path / "a.tif"), a, rows_per_strip=16)
    b = P.read_tiff(str(tmp_path / "a.tif"))
    assert b.dtype == a.dtype and np.array_equal(a, b)


def test_lzw_and_packbits_strips_from_libtiff(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    a = (rng.random((100, 130, 3)) * 255).astype(np.uint8)
    a[20:60] = 7                                            # long runs: codes grow past 9 bits, table resets
    for comp in ("tiff_lzw", "packbits", "tiff_adobe_deflate"):
        Image.fromarray(a).save(str(tmp_path / "l.tif"), compression=comp)
        assert np.array_equal(P.read_tiff(str(tmp_path / "l.tif")), a), comp
    big = (rng.random((300, 400)) * 65535).astype(np.uint16)
    Image.fromarray(big).save(str(tmp_path / "g.tif"), compression="tiff_lzw")
    assert np.array_equal(P.read_tiff(str(tmp_path / "g.tif")), big)


@pytest.mark.parametrize("dtype,c", [(np.uint8, 1), (np.uint8, 3), (np.uint8, 4), (np.uint16, 1)])
def test_png_writer_read_back_by_pillow(tmp_path, dtype, c):
    from PIL import Image
    a = (np.random.default_rng(2).random((33, 21, c)) * 250).astype(dtype)
    P.write_png(str(tmp_path / "a.png"), a)
    b = np.array(Image.open(str(tmp_path / "a.png")))
    assert np.array_equal(b.reshape(a.shape), a)


def test_png_writer_16bit_rgb_bytes(tmp_path):
    """Pillow cannot decode 16-bit RGB: check the stream itself (IHDR + inflated scanlines)."""
    import struct
    import zlib
    a = (np.random.default_rng(4).random((9, 5, 3)) * 65535).astype(np.uint16)
    P.write_png(str(tmp_path / "a.png"), a)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and raw[12:16] == b"IHDR"
    w, h, depth, ctype = struct.unpack(">IIBB", raw[16:26])
    assert (w, h, depth, ctype) == (5, 9, 16, 2)
    n = struct.unpack(">I", raw[33:37])[0]
    assert raw[37:41] == b"IDAT"
    lines = np.frombuffer(zlib.decompress(raw[41:41 + n]), np.uint8).reshape(9, 1 + 5 * 3 * 2)
    assert (lines[:, 0] == 0).all()
    assert np.array_equal(lines[:, 1:].copy().view(">u2").reshape(9, 5, 3), a)


@pytest.mark.parametrize("k", range(4))
def test_overlay_boundaries_equal_skimage(k):
    lab = GOLD[f"overlay_labels_{k}"]
    assert np.array_equal(P.find_boundaries_inner(lab), GOLD[f"overlay_inner_{k}"])
    assert np.array_equal(P.disk(2), GOLD["disk2"].astype(bool))
    img = np.full(lab.shape, 7, np.uint16)
    out = P.overlay_boundaries(img, lab)
    assert out.shape == lab.shape + (3,) and out.dtype == np.uint16
    assert np.array_equal(out[:, :, 0] == 255, GOLD[f"overlay_dilated_{k}"])
    assert (out[~GOLD[f"overlay_dilated_{k}"]] == 7).all()


def test_save_image_keeps_the_input_format(tmp_path):
    a16 = (np.random.default_rng(5).random((20, 30, 3)) * 60000).astype(np.uint16)
    P.save_image(str(tmp_path / "overlay_x.tif"), a16)
    assert np.array_equal(P.read_image(str(tmp_path / "overlay_x.tif")), a16)
    a8 = (a16 >> 8).astype(np.uint8)
    for ext in (".png", ".bmp"):
        P.save_image(str(tmp_path / ("overlay_x" + ext)), a8)
        assert np.array_equal(P.read_image(str(tmp_path / ("overlay_x" + ext))), a8)
