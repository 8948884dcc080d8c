"""bench.py --full: BASELINE.json configs[1]'s comparator, the reference's network through PyTorch-ROCm eager on the same GPU."""
from __future__ import annotations

import time

import torch

from .common import WIDTHS, synthetic_wsi

def eager_rocm(dev, size, roi, overlap, width, seed, engine_ms):
    """BASELINE.json configs[1]'s comparator, "HIP conv/BN kernels vs PyTorch-ROCm eager": what the reference's own code executes on this
    GPU - the restated network (oracle/unet_ref.py, torch.nn modules -> MIOpen / rocBLAS through PyTorch-ROCm eager) moved to the device,
    in fp32 and under torch.autocast(bfloat16): the forward over 32 x 3 x 256 x 256 tiles (model_training_3class.py:255), one training
    step at batch 32 (:250-262: zero_grad, forward, DiceCE, backward, AdamW) and the sliding-window pass with sw_batch_size 4
    (predict.py:109-111, up to the class-1 probability map on the host; the reference then post-processes on the host cores, which is
    cpu_baseline's skimage chain).  Runs after every timed region of the engine, like cpu_baseline; `engine_ms` are this run's figures."""
    from oracle.sliding_window_ref import sliding_window_inference_ref
    from oracle.train_ref import dice_ce_ref, one_hot_ref
    from oracle.unet_ref import make_reference_unet

    def timed(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {"torch": torch.__version__, "network": "oracle/unet_ref.py (MONAI UNet restated as torch.nn modules) on cuda:0, eager",
           "note": "ratio = eager ms / engine ms of this same run (engine: cellseg_amd through the C-ABI); bf16 = torch.autocast(bfloat16) "
                   "against the engine's bf16 datapath"}
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(32, 3, 256, 256, generator=g).to(dev)
    y = torch.randint(0, 3, (32, 1, 256, 256), generator=g).to(dev)
    net = make_reference_unet(seed, channels=WIDTHS[width]).to(dev).eval()
    modes = (("fp32", None), ("bf16", torch.bfloat16))

    def ctx(dt):
        import contextlib
        return torch.autocast("cuda", dtype=dt) if dt is not None else contextlib.nullcontext()

    def pair(key, eager):
        eng = engine_ms.get(key)
        return {"eager_ms": round(eager, 3), "engine_ms": round(eng, 3) if eng else None, "ratio": round(eager / eng, 2) if eng else None}

    tiles, train, slide = {}, {}, {}
    with torch.no_grad():
        for name, dt in modes:
            with ctx(dt):
                tiles[name] = pair(f"tiles_{name}", timed(lambda: net(x), 10))
    # the slide as predict.py runs it: windows in groups of 4, logits accumulated on the device, softmax, class-1 map to the host
    img = torch.from_numpy(synthetic_wsi(size, seed)).to(dev)[None]

    def slide_once():
        logits = sliding_window_inference_ref(img, (roi, roi), 4, net, overlap)
        return torch.softmax(logits, 1)[0, 1].cpu()

    with torch.no_grad():
        for name, dt in modes:
            with ctx(dt):
                ms = timed(slide_once, 2, warm=1)
            slide[name] = dict(pair(f"slide_{name}", ms), eager_mpixels_s=round(size * size / 1e3 / ms, 2))
    slide["scope"] = ("eager: predict.py:109-111 (sliding_window_inference sw_batch_size 4 -> softmax -> class-1 probabilities on the host), WITHOUT the "
                      "host post-process that follows in the reference; engine: the whole headline step incl. connected components and the label copy")
    net.train()
    opt = torch.optim.AdamW(net.parameters(), 6e-4)
    onehot = one_hot_ref(y, 3)
    for name, dt in modes:
        def step():
            opt.zero_grad()
            with ctx(dt):
                loss = dice_ce_ref(net(x).float(), onehot)
            loss.backward()
            opt.step()
        train[name] = pair(f"train_{name}", timed(step, 10))
    out.update(tiles_forward_n32=tiles, train_step_batch32=train, slide=slide)
    return out

