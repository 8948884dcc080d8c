"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/cellseg_hip.h declares,
the window plan matches the oracle tiler, MONAI state-dict compatibility, loud failure without a GPU, host
pre/post-processing against the reference-generated goldens.  No GPU compute is called."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    import cellseg_amd as ca
    header = open(os.path.join(ROOT, "include", "cellseg_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(ca.lib, name), f"{name} declared in include/cellseg_hip.h but not exported"
    assert declared == set(ca._lib.EXPORTED)
    assert ca.lib.cs_version() >= 100


def test_augmentation_record_layout_matches_the_c_header(tmp_path):
    """`cs_aug_params` is filled by ctypes on the host and read by the kernels: same size and the same offset for every field (gcc)."""
    import ctypes as C
    import subprocess
    from cellseg_amd import gpu_data
    fields = [f[0] for f in gpu_data.AugParams._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cellseg_hip.h"\nint main(void) {\n  printf("%zu", sizeof(cs_aug_params));\n'
                   + "".join(f'  printf(" %zu", offsetof(cs_aug_params, {f}));\n' for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == C.sizeof(gpu_data.AugParams)
    assert got[1:] == [getattr(gpu_data.AugParams, f).offset for f in fields]


def test_draw_params_follows_the_reference_chain_and_is_reproducible():
    """One record per sample, drawn in the order of the reference's transform list (model_training_3class.py:119-162) with its
    probabilities; the host pipeline applies it to the crop size whatever the source size."""
    from cellseg_amd import data
    a = [data.draw_params(np.random.default_rng(4), 300, 200, 256) for _ in range(2)]
    assert a[0].keys() == a[1].keys() and all(np.array_equal(a[0][k], a[1][k]) if k == "hist" else a[0][k] == a[1][k] for k in a[0])
    rng = np.random.default_rng(0)
    recs = [data.draw_params(rng, 300, 200, 256) for _ in range(4000)]
    frac = lambda f: np.mean([bool(f(r)) for r in recs])          # noqa: E731
    assert abs(frac(lambda r: r["flip_axis"]) - 0.5) < 0.03 and abs(frac(lambda r: r["rot_k"]) - 0.5) < 0.03
    for key, prob in (("gamma", 0.25), ("sigma_y", 0.25), ("zoom", 0.15)):
        assert abs(frac(lambda r: r[key] > 0) - prob) < 0.03, key
    assert abs(frac(lambda r: r["noise_std"] >= 0) - 0.25) < 0.03 and abs(frac(lambda r: r["hist"] is not None) - 0.25) < 0.03
    assert all(r["pad_w"] == 56 and r["pad_left"] == 28 and r["pad_h"] == 0 and 0 <= r["crop_y"] <= 44 and r["crop_x"] == 0 for r in recs)
    assert all(1.0 <= r["gamma"] < 2.0 for r in recs if r["gamma"] > 0) and all(0.8 <= r["zoom"] < 1.5 for r in recs if r["zoom"] > 0)
    img, lab = np.random.default_rng(1).random((3, 300, 200), dtype=np.float32), np.ones((1, 300, 200), np.uint8)
    for r in recs[:40]:
        oi, ol = data.apply_params(img, lab, r, 256)
        assert oi.shape == (3, 256, 256) and ol.shape == (1, 256, 256) and oi.dtype == np.float32 and ol.dtype == np.uint8


@pytest.mark.parametrize("hw", [(4096, 4096), (1024, 1000), (512, 512), (200, 200), (256, 300), (257, 255), (2000, 333)])
@pytest.mark.parametrize("overlap", [0.25, 0.5, 0.0])
def test_plan_matches_oracle_tiler(hw, overlap):
    import cellseg_amd as ca
    from oracle.sliding_window_ref import scan_starts
    plan = ca.make_plan(hw[0], hw[1], (256, 256), overlap)
    hs, ws = ca.window_starts(plan)
    assert hs == scan_starts(max(hw[0], 256), 256, overlap) and ws == scan_starts(max(hw[1], 256), 256, overlap)
    assert plan.pad_top == max(256 - hw[0], 0) // 2 and plan.pad_left == max(256 - hw[1], 0) // 2
    assert (plan.n_h, plan.n_w) == (len(hs), len(ws))


def test_plan_window_counts_of_the_configs():
    import cellseg_amd as ca
    assert (lambda p: p.n_h * p.n_w)(ca.make_plan(4096, 4096, (256, 256), 0.25)) == 441
    assert (lambda p: p.n_h * p.n_w)(ca.make_plan(4096, 4096, (256, 256), 0.5)) == 961
    assert (lambda p: p.n_h * p.n_w)(ca.make_plan(8192, 8192, (512, 512), 0.5)) == 961


def test_balanced_batch_deals_equal_chunks_to_the_streams():
    """engine_batch=None: the window list is cut into EQUAL chunks, one round per internal stream where that fits 512 windows (the slide's
    961 windows -> 2 x 481, a rank's 124 of the 8-rank strips leg -> 2 x 62), a handful of windows stays one launch, and no chunk exceeds
    the cap; every window is covered."""
    import cellseg_amd as ca
    assert ca.balanced_batch(961, 2) == 481 and ca.balanced_batch(124, 2) == 62 and ca.balanced_batch(480, 2) == 240
    assert ca.balanced_batch(20, 2) == 20 and ca.balanced_batch(1, 2) == 1 and ca.balanced_batch(961, 1) == 481
    for n in (1, 31, 32, 33, 441, 961, 1024, 1025, 3969, 10000):
        for s in (1, 2, 3, 4):
            b = ca.balanced_batch(n, s)
            chunks = -(-n // b)
            assert 1 <= b <= 512 or n < 16 * s
            assert chunks * b >= n and (chunks - 1) * b < n
            if n >= 16 * s:
                assert chunks % s == 0 or chunks * b - n < b, (n, s, b)          # whole rounds over the streams (the last chunk may be short)


def test_state_dict_is_monai_compatible_both_ways():
    import cellseg_amd as ca
    from oracle.unet_ref import UNetRef
    for kw in (dict(), dict(channels=(32, 64, 128, 256, 512)), dict(num_res_units=1), dict(norm="batch", act="relu")):
        ref = UNetRef(**kw)
        hip = ca.UNet(**{**kw, "strides": (2,) * (len(kw.get("channels", (16, 32, 64, 128, 256))) - 1)})
        a, b = hip.state_dict(), ref.state_dict()
        assert set(a) == set(b) and all(a[k].shape == b[k].shape for k in a)
        hip.load_state_dict(b)
        ref.load_state_dict(hip.state_dict())
    assert sum(p.numel() for p in ca.UNet().parameters()) == 1_626_072


def test_no_cpu_fallback():
    import cellseg_amd as ca
    m = ca.UNet()
    with pytest.raises(ca.CellSegHipError):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ca.CellSegHipError):
        ca.sliding_window_inference(torch.zeros(1, 3, 300, 300), (256, 256), 4, m)
    with pytest.raises(ca.CellSegHipError):
        ca.mask_to_instances(torch.zeros(8, 8, dtype=torch.bool))
    with pytest.raises(ca.CellSegHipError):
        ca.label(torch.zeros(8, 8, dtype=torch.bool))


def test_unsupported_arguments_are_refused():
    import cellseg_amd as ca
    with pytest.raises(ValueError):
        ca.UNet(spatial_dims=3)
    with pytest.raises(ValueError):
        ca.UNet(channels=(16, 32, 64), strides=(2, 1))
    with pytest.raises(ValueError):
        ca.UNet(channels=(10, 20, 40), strides=(2, 2))
    with pytest.raises(NotImplementedError):
        ca.sliding_window_inference(torch.zeros(1, 3, 8, 8), (256, 256), 4, ca.UNet(), mode="gaussian")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neurips-cellseg_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_preprocess_matches_reference_golden():
    from cellseg_amd import preprocess
    from oracle import normalize_ref
    g = np.load(os.path.join(G, "normalize_golden.npz"))
    for n in sorted({k.split("/")[0] for k in g.files}):
        assert np.array_equal(preprocess.normalize_channel(g[n + "/in"]), g[n + "/out"])
    rng = np.random.default_rng(0)
    for img in (rng.integers(0, 4000, (50, 60), dtype=np.uint16), rng.integers(0, 255, (40, 30, 4), dtype=np.uint8)):
        pre = preprocess.prepare_image(img)
        assert np.array_equal(pre, normalize_ref.preprocess_image_ref(img))
        assert np.array_equal(preprocess.to_unit_float(pre).transpose(2, 0, 1)[None], normalize_ref.to_network_input_ref(pre))


def test_label_tiff_round_trip(tmp_path):
    from cellseg_amd import preprocess
    lab = (np.arange(300 * 200).reshape(300, 200) % 70001).astype(np.int32)
    p = str(tmp_path / preprocess.label_name("cell_001.png"))
    assert p.endswith("cell_001_label.tiff")
    preprocess.write_label_tiff(p, lab)
    back = preprocess.read_image(p)
    assert back.dtype == np.int32 and np.array_equal(back, lab)
    # odd shapes, single strip, one row, negative ids (int32 is signed), and what another reader sees
    from PIL import Image
    for shape, rps in [((1, 1), 64), ((7, 13), 64), ((65, 33), 64), ((130, 5), 1), ((64, 64), 64)]:
        lab2 = (np.random.default_rng(shape[0]).integers(-5, 1 << 20, shape)).astype(np.int32)
        p2 = str(tmp_path / f"s{shape[0]}_{shape[1]}_label.tiff")
        preprocess.write_label_tiff(p2, lab2, rows_per_strip=rps)
        with Image.open(p2) as im:
            assert im.mode == "I" and im.size == (shape[1], shape[0])
            assert np.array_equal(np.array(im), lab2)
    conda = "/opt/conda/bin/python3.9"
    if os.path.exists(conda):                                   # tifffile (the reference's reader/writer) must agree too
        import subprocess
        out = subprocess.run([conda, "-c", f"import tifffile,sys; a=tifffile.imread(r'{p}'); print(a.dtype, a.shape, int(a.sum()))"],
                             capture_output=True, text=True)
        if out.returncode == 0:
            assert out.stdout.split() == ["int32", "(300,", "200)", str(int(lab.astype(np.int64).sum()))], out.stdout


# ------------------------------------------------------------------------------------------- training augmentations (host)
def test_augmentations_keep_shapes_and_label_values():
    from cellseg_amd import data
    rng = np.random.default_rng(3)
    img = rng.random((3, 64, 80), dtype=np.float32)
    lab = rng.integers(0, 3, (1, 64, 80)).astype(np.uint8)
    for z in (0.8, 1.0, 1.37, 1.5):
        zi, zl = data.zoom_keep_size(img, lab, z)
        assert zi.shape == img.shape and zl.shape == lab.shape and set(np.unique(zl)) <= {0, 1, 2}
    zi, zl = data.zoom_keep_size(img, lab, 1.0)
    assert np.allclose(zi, img) and np.array_equal(zl, lab)
    k = data.gaussian_kernel_1d(1.5)
    assert len(k) == 2 * 6 + 1 and abs(k.sum() - 1.0) < 1e-3 and np.allclose(k, k[::-1])
    sm = data.gaussian_smooth(np.ones((1, 40, 40), np.float32), 1.5, 1.0)
    assert abs(sm[0, 20, 20] - 1.0) < 2e-3 and sm[0, 0, 0] < 0.5            # zero padding darkens the corners
    fl = data.random_control_points(rng, 3)
    assert fl[0] == 0.0 and fl[-1] == 1.0 and 0.0 <= fl[1] <= 1.0
    hs = data.histogram_shift(img, fl)
    assert hs.min() == pytest.approx(img.min()) and hs.max() == pytest.approx(img.max())
    order = np.argsort(img.ravel())
    assert np.all(np.diff(hs.ravel()[order]) >= -1e-7)                        # monotone remap


def test_cell_dataset_train_pipeline(tmp_path):
    from PIL import Image
    from cellseg_amd import data
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "labels")
    rng = np.random.default_rng(0)
    for i in range(4):
        Image.fromarray((rng.random((90, 70, 3)) * 255).astype(np.uint8)).save(tmp_path / "images" / f"c{i}.png")
        Image.fromarray(rng.integers(0, 3, (90, 70)).astype(np.uint8)).save(tmp_path / "labels" / f"c{i}_label.png")
    train, val = data.split_files(str(tmp_path), 0, 0.25)
    assert len(train) == 3 and len(val) == 1
    ds = data.CellDataset(train, 64, True, seed=1)
    for rep in range(40):                                                     # exercises every augmentation branch
        item = ds[rep % 3]
        assert item["img"].shape == (3, 64, 64) and item["label"].shape == (1, 64, 64)
        assert item["img"].dtype == torch.float32 and set(item["label"].unique().tolist()) <= {0, 1, 2}


# ---- deterministic parts of the augmentations against independent restatements written down here (MONAI is absent:
# ---- model_training_3class.py:119-162 is anchored on what MONAI's transforms are documented to compute)
def test_gaussian_smooth_equals_direct_erf_convolution():
    """RandGaussianSmoothd's deterministic part: separable correlation with pixel-integrated Gaussian taps
    0.5 (erf((x + 0.5) / (sigma sqrt 2)) - erf((x - 0.5) / (sigma sqrt 2))), tail int(max(4 sigma, 0.5) + 0.5), zero padding."""
    import math
    from cellseg_amd import data
    rng = np.random.default_rng(5)
    img = rng.random((2, 19, 23)).astype(np.float32)
    sy, sx = 1.3, 0.6

    def taps(sigma):
        tail = int(max(sigma * 4.0, 0.5) + 0.5)
        return tail, [0.5 * (math.erf((x + 0.5) / (sigma * math.sqrt(2))) - math.erf((x - 0.5) / (sigma * math.sqrt(2)))) for x in range(-tail, tail + 1)]
    ty, ky = taps(sy)
    tx, kx = taps(sx)
    want = np.zeros_like(img, dtype=np.float64)
    tmp = np.zeros_like(img, dtype=np.float64)
    c, h, w = img.shape
    for y in range(h):
        for j, wgt in enumerate(ky):
            yy = y + j - ty
            if 0 <= yy < h:
                tmp[:, y] += wgt * img[:, yy]
    for x in range(w):
        for j, wgt in enumerate(kx):
            xx = x + j - tx
            if 0 <= xx < w:
                want[:, :, x] += wgt * tmp[:, :, xx]
    got = data.gaussian_smooth(img, sy, sx)
    assert np.abs(got - want).max() <= 5e-6


def test_histogram_shift_equals_piecewise_linear_formula():
    """RandHistogramShiftd's deterministic part with 3 control points: x in [lo, mid] maps linearly onto [lo, f], x in [mid, hi]
    onto [f, hi] where f = lo + fl[1] (hi - lo)."""
    from cellseg_amd import data
    rng = np.random.default_rng(6)
    img = (rng.random((3, 17, 11)) * 3 - 1).astype(np.float32)
    fl = np.array([0.0, 0.73, 1.0])
    lo, hi = float(img.min()), float(img.max())
    mid, f = 0.5 * (lo + hi), lo + 0.73 * (hi - lo)
    want = np.where(img <= mid, lo + (img - lo) / (mid - lo) * (f - lo), f + (img - mid) / (hi - mid) * (hi - f))
    assert np.abs(data.histogram_shift(img, fl) - want).max() <= 1e-5


def test_zoom_keep_size_equals_block_average_and_nearest():
    """RandZoomd(mode=("area", "nearest"), keep_size=True, padding_mode="edge") for zoom factors with integer ratios: zoom 0.5
    is a 2x2 block average (image) / the top-left sample of each block (label: nearest picks floor(i / zoom)), edge-padded back to
    the input size; zoom 2.0 replicates every pixel 2x2 and crops the centre."""
    from cellseg_amd import data
    rng = np.random.default_rng(7)
    img = rng.random((3, 16, 24)).astype(np.float32)
    lab = rng.integers(0, 3, (1, 16, 24)).astype(np.uint8)
    zi, zl = data.zoom_keep_size(img, lab, 0.5)
    small = img.reshape(3, 8, 2, 12, 2).mean((2, 4))
    small_l = lab[:, ::2, ::2]
    assert np.abs(zi[:, 4:12, 6:18] - small).max() <= 1e-6 and np.array_equal(zl[:, 4:12, 6:18], small_l)
    assert np.allclose(zi[:, 0, 6:18], small[:, 0]) and np.allclose(zi[:, 15, 6:18], small[:, -1])      # edge padding, rows
    assert np.allclose(zi[:, 4:12, 0], small[:, :, 0]) and np.array_equal(zl[:, 4:12, 23], small_l[:, :, -1])
    zi, zl = data.zoom_keep_size(img, lab, 2.0)
    big = np.repeat(np.repeat(img, 2, 1), 2, 2)
    big_l = np.repeat(np.repeat(lab, 2, 1), 2, 2)
    assert np.abs(zi - big[:, 8:24, 12:36]).max() <= 1e-6 and np.array_equal(zl, big_l[:, 8:24, 12:36])


def test_worker_seeding_gives_distinct_streams():
    """forked DataLoader workers all replayed the parent's generator state."""
    from cellseg_amd import data
    ds = data.CellDataset([], 64, True, seed=3)

    class Info:
        def __init__(self, seed, ds):
            self.seed, self.dataset = seed, ds
    draws = []
    for wid, seed in ((0, 1000), (1, 1001), (0, 2000)):
        real = torch.utils.data.get_worker_info
        torch.utils.data.get_worker_info = lambda s=seed: Info(s, ds)
        try:
            data.seed_worker(wid)
        finally:
            torch.utils.data.get_worker_info = real
        draws.append(ds.rng.random(4).tolist())
    assert draws[0] != draws[1] and draws[0] != draws[2] and draws[1] != draws[2]


def test_augmentation_kernels_match_the_primitives_monai_calls():
    """zoom / histogram shift / Gaussian smoothing against fixtures produced by the primitives MONAI's transforms call
    (torch.nn.functional.interpolate area + nearest with MONAI's crop / edge-pad rule, np.interp, a separable torch conv with
    MONAI's erf taps): tests/golden/make_augment_golden.py.  Reference: baseline/model_training_3class.py:119-162."""
    from cellseg_amd import data
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_golden.npz"))
    img, lab = g["zoom_img"], g["zoom_lab"]
    for i, z in enumerate(g["zoom_factors"]):
        zi, zl = data.zoom_keep_size(img, lab, float(z))
        assert zi.shape == img.shape and zl.shape == lab.shape and zl.dtype == lab.dtype
        assert np.array_equal(zi, g[f"zoom_area_{i}"]), ("area", z)
        assert np.array_equal(zl, g[f"zoom_nearest_{i}"]), ("nearest", z)
    for i, fl in enumerate(g["hist_floating"]):
        assert np.array_equal(data.histogram_shift(g["hist_img"], fl), g[f"hist_out_{i}"]), fl
    for i, (s0, s1) in enumerate(g["gauss_sigmas"]):
        taps = data.gaussian_kernel_1d(float(s0))
        assert taps.shape == g[f"gauss_taps_{i}"].shape and np.abs(taps - g[f"gauss_taps_{i}"]).max() <= 1e-7
        got = data.gaussian_smooth(g["gauss_img"], float(s0), float(s1))
        assert np.abs(got - g[f"gauss_out_{i}"]).max() <= 2e-6, (s0, s1)
