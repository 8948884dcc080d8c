"""GPU tests of the device-side training augmentations (csrc/kernels_augment.hip).

The checker is the host pipeline (`cellseg_amd.data.apply_params`: numpy / scipy / torch-CPU, its deterministic kernels pinned to the
primitives MONAI calls by tests/golden/augment_golden.npz) applied to the SAME parameter records: pad / crop / flip / rotate must agree
exactly, the floating-point stages to float rounding, the labels exactly.  The Gaussian noise is checked by its moments (the device
generator is not numpy's stream).

Run on the MI355X box with ``pytest -m gpu``.  Everything goes through the C-ABI (libcellseg_hip.so)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips-cellseg_amd"))

SIZE = 256


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _samples(seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for h, w in [(300, 420), (200, 180), (256, 256), (257, 511), (120, 400)]:
        img = rng.random((3, h, w), dtype=np.float32)
        img[:, : h // 3] *= 0.2                                   # structure, so that flips / rotations / crops are told apart
        lab = (rng.random((1, h, w)) * 3).astype(np.uint8)
        out.append((img, lab))
    return out


def _noop(src_h, src_w):
    ph, pw = max(SIZE - src_h, 0), max(SIZE - src_w, 0)
    return {"src_h": src_h, "src_w": src_w, "pad_top": ph // 2, "pad_left": pw // 2, "pad_h": ph, "pad_w": pw, "crop_y": 0, "crop_x": 0,
            "flip_axis": 0, "rot_k": 0, "noise_std": -1.0, "noise_seed": 0, "gamma": 0.0, "sigma_y": 0.0, "sigma_x": 0.0, "hist": None,
            "zoom": 0.0}


def _run(samples, indices, params):
    from cellseg_amd import data, gpu_data
    ds = gpu_data.DeviceCellDataset(samples, _dev())
    aug = gpu_data.GpuAugmenter(ds, SIZE, seed=1)
    img, lab = aug.batch(indices, params)
    want = [data.apply_params(samples[i][0], samples[i][1], p, SIZE) for i, p in zip(indices, params)]
    return img.cpu().numpy(), lab.cpu().numpy(), want


def test_pad_crop_flip_rotate_are_exact():
    samples = _samples()
    rng = np.random.default_rng(3)
    indices, params = [], []
    for i, (img, _) in enumerate(samples):
        for flip in (0, 1, 2):
            for rot in (0, 1, 2, 3):
                p = _noop(*img.shape[1:])
                p["crop_y"] = int(rng.integers(0, img.shape[1] + p["pad_h"] - SIZE + 1))
                p["crop_x"] = int(rng.integers(0, img.shape[2] + p["pad_w"] - SIZE + 1))
                p["flip_axis"], p["rot_k"] = flip, rot
                indices.append(i)
                params.append(p)
    img, lab, want = _run(samples, indices, params)
    for k, (wi, wl) in enumerate(want):
        assert np.array_equal(img[k], wi), (indices[k], params[k])
        assert np.array_equal(lab[k], wl), (indices[k], params[k])


@pytest.mark.parametrize("stage", ["gamma", "smooth", "hist", "zoom_in", "zoom_out"])
def test_each_floating_point_stage_matches_the_host_pipeline(stage):
    samples = _samples(1)
    rng = np.random.default_rng(11)
    from cellseg_amd import data
    indices, params = [], []
    for i, (img, _) in enumerate(samples):
        for rep in range(4):
            p = _noop(*img.shape[1:])
            p["crop_y"] = int(rng.integers(0, img.shape[1] + p["pad_h"] - SIZE + 1))
            p["crop_x"] = int(rng.integers(0, img.shape[2] + p["pad_w"] - SIZE + 1))
            if stage == "gamma":
                p["gamma"] = float(rng.uniform(1, 2))
            elif stage == "smooth":
                p["sigma_y"], p["sigma_x"] = float(rng.uniform(1, 2)), float(rng.uniform(0.25, 1.5))
            elif stage == "hist":
                p["hist"] = data.random_control_points(rng, 3)
            elif stage == "zoom_in":
                p["zoom"] = [1.5, 1.0039, float(rng.uniform(1.0, 1.5)), 1.25][rep]
            else:
                p["zoom"] = [0.8, 0.9961, float(rng.uniform(0.8, 1.0)), 0.875][rep]
            indices.append(i)
            params.append(p)
    img, lab, want = _run(samples, indices, params)
    for k, (wi, wl) in enumerate(want):
        assert np.array_equal(lab[k], wl), (stage, params[k])
        err = float(np.abs(img[k] - wi).max())
        assert err <= 2e-6, (stage, params[k], err)


def test_whole_chain_on_drawn_records_matches_the_host_pipeline():
    """Records drawn by `data.draw_params` (the reference chain's own probabilities), noise switched off so that the comparison is exact
    to rounding: every combination the training loop meets, on images larger and smaller than the crop."""
    from cellseg_amd import data
    samples = _samples(2)
    rng = np.random.default_rng(5)
    indices = [int(i) for i in rng.integers(0, len(samples), 96)]
    params = [data.draw_params(rng, *samples[i][0].shape[1:], SIZE) for i in indices]
    fired = {k: 0 for k in ("gamma", "sigma_y", "zoom", "hist", "flip_axis", "rot_k")}
    for p in params:
        p["noise_std"] = -1.0
        for k in fired:
            fired[k] += (p[k] is not None) if k == "hist" else bool(p[k])
    assert all(v > 0 for v in fired.values()), fired
    img, lab, want = _run(samples, indices, params)
    for k, (wi, wl) in enumerate(want):
        assert np.array_equal(lab[k], wl), params[k]
        err = float(np.abs(img[k] - wi).max())
        assert err <= 5e-6, (params[k], err)


def test_noise_has_the_drawn_standard_deviation():
    samples = _samples(3)
    params, indices = [], []
    for std in (0.01, 0.05, 0.1):
        for seed in (1, 2):
            p = _noop(*samples[0][0].shape[1:])
            p["noise_std"], p["noise_seed"] = std, seed
            params.append(p)
            indices.append(0)
    clean = _noop(*samples[0][0].shape[1:])
    img, _, _ = _run(samples, indices + [0], params + [clean])
    for k, p in enumerate(params):
        d = (img[k] - img[-1]).ravel().astype(np.float64)
        assert abs(d.mean()) < 4 * p["noise_std"] / np.sqrt(d.size) + 1e-6
        assert abs(d.std() / p["noise_std"] - 1) < 0.01
        assert abs(np.mean(d ** 4) / d.std() ** 4 - 3) < 0.1                    # Gaussian kurtosis
    assert not np.array_equal(img[0], img[1])                                   # different seeds, different fields


def test_batches_from_the_augmenters_own_stream_train_a_step():
    """`GpuAugmenter.batch` feeds `TrainStep` directly (device tensors, no host round trip); consecutive calls reuse the two staging buffers."""
    import cellseg_amd as ca
    from cellseg_amd import gpu_data
    dev = _dev()
    ds = gpu_data.DeviceCellDataset(_samples(4), dev)
    assert len(ds) == 5 and ds.nbytes() > 0
    aug = gpu_data.GpuAugmenter(ds, SIZE, seed=9)
    torch.manual_seed(0)
    model = ca.UNet(spatial_dims=2, in_channels=3, out_channels=3, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
    step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
    losses = []
    for it in range(6):
        img, lab = aug.batch([it % 5, (it + 1) % 5, (it + 2) % 5, (it + 3) % 5])
        assert img.shape == (4, 3, SIZE, SIZE) and lab.shape == (4, 1, SIZE, SIZE) and lab.dtype == torch.uint8
        assert int(lab.max()) <= 2 and torch.isfinite(img).all()
        losses.append(float(step(img, ca.one_hot(lab, 3))))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def _env():
    pkg = os.path.join(ROOT, "neurips-cellseg_amd")
    return dict(os.environ, PYTHONPATH=os.pathsep.join([pkg, ROOT, os.environ.get("PYTHONPATH", "")]), HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_training_cli_with_the_dataset_resident_in_hbm(tmp_path):
    """`model_training_3class --gpu_data`: the reference's directory layout (images/, labels/<stem>_label.png) decoded once into HBM,
    batches augmented on the device, checkpoints written as before; the loss falls over the epochs."""
    import subprocess
    from scipy.ndimage import binary_erosion, gaussian_filter
    from cellseg_amd import preprocess
    data_dir = tmp_path / "Train_Pre_3class"
    (data_dir / "images").mkdir(parents=True)
    (data_dir / "labels").mkdir()
    rng = np.random.default_rng(0)
    for k in range(20):
        h, w = int(rng.integers(200, 400)), int(rng.integers(260, 420))
        g = gaussian_filter(rng.random((h, w)), 5)
        g = (g - g.min()) / (g.max() - g.min())
        fg = g > np.quantile(g, 0.6)
        lab = np.zeros((h, w), np.uint8)
        lab[fg] = 2
        lab[binary_erosion(fg, iterations=2)] = 1
        rgb = np.stack([g, g, g], -1) + rng.normal(0, 0.03, (h, w, 3))
        preprocess.write_png(str(data_dir / "images" / f"cell_{k:03d}.png"), (np.clip(rgb, 0, 1) * 255).astype(np.uint8))
        preprocess.write_png(str(data_dir / "labels" / f"cell_{k:03d}_label.png"), lab)
    work = tmp_path / "work"
    out = subprocess.run([sys.executable, "-m", "cellseg_amd.model_training_3class", "--data_path", str(data_dir), "--gpu_data",
                          "--max_epochs", "7", "--val_start_epoch", "3", "--val_interval", "2", "--batch_size", "6", "--num_workers", "0",
                          "--work_dir", str(work)], capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    assert "training set resident in HBM: 18 images" in out.stdout, out.stdout[-1500:]
    losses = [float(l.split("average loss:")[1].split()[0]) for l in out.stdout.splitlines() if "average loss" in l]
    assert len(losses) == 6 and losses[-1] < losses[0], losses
    ck = torch.load(os.path.join(work, "unet_3class", "final_model.pth"), map_location="cpu")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
