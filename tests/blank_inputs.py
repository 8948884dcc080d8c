"""Inputs and nets of the blank-window parity tests (tests/test_gpu_blank_windows.py, tests/diagnostics/gpu_blank_windows.py).

The reference normalises every channel over its NON-ZERO pixels only and leaves zero background at exactly 0
(/root/reference/baseline/predict.py:14-21, 94-108), so real slides are full of all-zero, constant and nearly empty windows: behind such
a window a conv output is constant (or constant away from the border / a small patch), InstanceNorm's variance collapses and its rstd
climbs towards 1/sqrt(eps) = 316 - every rounding difference of the layer in front is amplified by that factor."""
import numpy as np
import torch

SEED = 2022


def blank_windows(size=256):
    """{name: [N,3,size,size] fp32} - the window kinds VERDICT r05 names, plus a uniform-noise control."""
    g = torch.Generator().manual_seed(11)
    s = size
    zero = torch.zeros(1, 3, s, s)
    const = torch.full((1, 3, s, s), 0.37)
    const3 = torch.tensor([0.2, 0.5, 0.9]).view(1, 3, 1, 1).expand(1, 3, s, s).clone()
    patch = torch.zeros(1, 3, s, s)
    patch[0, :, 97:97 + 22, 140:140 + 13] = torch.rand(3, 22, 13, generator=g)
    corner = torch.zeros(1, 3, s, s)                       # a patch that touches the window's corner (zero padding next to data)
    corner[0, :, :9, :17] = torch.rand(3, 9, 17, generator=g)
    half = torch.rand(1, 3, s, s, generator=g)
    half[..., : s // 2] = 0.0
    pixel = torch.zeros(1, 3, s, s)
    pixel[0, :, 128, 128] = 1.0
    # cells on an exactly-zero background, quantised like the reference's uint8 pipeline (pre / max(pre))
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(5)
    f = gaussian_filter(rng.random((s, s), dtype=np.float32), 6)
    f = (f - f.min()) / (f.max() - f.min())
    cells = np.where(f > np.quantile(f, 0.8), f, 0.0)
    u8 = np.round(np.stack([cells, cells * 0.7, cells * 0.4]) * 255).astype(np.uint8)
    sparse = torch.from_numpy((u8 / u8.max()).astype(np.float32))[None]
    noise = torch.rand(1, 3, s, s, generator=g)
    # one batch holding blank AND busy windows: a chunk of a real slide (statistics are per window - nothing may leak between them)
    mixed = torch.cat([zero, noise, const, sparse, patch], 0)
    return {"zero": zero, "const": const, "const3": const3, "patch22x13": patch, "corner9x17": corner, "half_zero": half, "one_pixel": pixel,
            "sparse_cells_u8": sparse, "noise": noise, "mixed_batch": mixed}


def trained_state(channels, steps=150, seed=7):
    """A net whose logits are decisive: `steps` AdamW steps on seeded synthetic cells with the engine (fp32); -> CPU state dict."""
    import cellseg_amd as ca
    from cellseg_amd import data
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    model = ca.UNet(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=2).to(dev).train()
    ds = data.SyntheticCells(32, 256, SEED)
    batches = [(torch.stack([ds[i]["img"] for i in range(b, b + 8)]).to(dev),
                ca.one_hot(torch.stack([ds[i]["label"] for i in range(b, b + 8)]).to(dev), 3)) for b in range(0, 32, 8)]
    step = ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model, 6e-4))
    for it in range(steps):
        step(*batches[it % len(batches)])
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def make_nets(trained=True):
    """{name: {"channels", "state"}}: the seeded 16- and 32-base nets of the parity tests and (trained=True, needs the GPU) a 16...256 net
    trained on synthetic cells."""
    from oracle.unet_ref import make_reference_unet
    nets = {}
    for name, ch in (("seeded16", (16, 32, 64, 128, 256)), ("seeded32", (32, 64, 128, 256, 512))):
        nets[name] = {"channels": list(ch), "state": {k: v.clone() for k, v in make_reference_unet(SEED, channels=ch).state_dict().items()}}
    if trained:
        nets["trained16"] = {"channels": [16, 32, 64, 128, 256], "state": trained_state((16, 32, 64, 128, 256))}
    return nets


def zero_background_slide(n=4096, seed=SEED, keep=0.6):
    """[3,n,n] fp32: Gaussian-filtered blobs (SURVEY 8(d)'s generator) inside low-frequency 'tissue' regions, everything else EXACTLY zero
    (>= 55 % of the slide, with whole windows blank), values quantised through uint8 as predict.py:94-107 does (pre / max(pre))."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.random((n, n), dtype=np.float32), 6)
    g = (g - g.min()) / (g.max() - g.min())
    t = gaussian_filter(rng.random((n // 16, n // 16), dtype=np.float32), 10)
    tissue = np.kron(t > np.quantile(t, 0.45), np.ones((16, 16), bool))
    fg = (g > np.quantile(g, 1.0 - keep)) & tissue
    x = np.stack([g, g * 0.8, g * 0.6]) + rng.normal(0, 0.02, (3, n, n)).astype(np.float32)
    x = np.where(fg[None], np.clip(x, 1.0 / 255, 1.0), 0.0)
    u8 = np.round(x * 255).astype(np.uint8)
    return torch.from_numpy((u8 / u8.max()).astype(np.float32))
