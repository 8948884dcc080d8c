#!/usr/bin/env python3
"""The reference's own way of running on this GPU (PyTorch-ROCm eager through MIOpen) timed beside the engine.

TEST INFRASTRUCTURE: uses the oracle's restated network (oracle/unet_ref.py) moved to cuda:0, i.e. exactly what
``monai.networks.nets.UNet(...).to("cuda")`` + ``sliding_window_inference(..., sw_batch_size=4, ...)`` would execute
(BASELINE.json configs[1]: "HIP conv/BN kernels vs PyTorch-ROCm eager").  Not part of the product and not a bench line;
the numbers go into DESIGN.md section 5.

    python tests/diagnostics/gpu_eager_compare.py [--size 4096] [--width baseline32] > gpurun_out/eager_compare.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np
import torch
import cellseg_amd as ca
from bench import synthetic_wsi
from oracle.unet_ref import make_reference_unet
from oracle.sliding_window_ref import sliding_window_inference_ref
from oracle.postprocess_ref import postprocess_c

WIDTHS = {"baseline32": (32, 64, 128, 256, 512), "reference16": (16, 32, 64, 128, 256)}


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--width", choices=sorted(WIDTHS), default="baseline32")
    ap.add_argument("--overlap", type=float, default=0.5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ch = WIDTHS[args.width]
    ref = make_reference_unet(2022, channels=ch, strides=(2, 2, 2, 2), num_res_units=2).eval().to(dev)
    hip = ca.UNet(channels=ch, strides=(2, 2, 2, 2), num_res_units=2)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev).eval()
    out = {"width": args.width, "torch": torch.__version__}

    # --- configs[1]: forward (and train step) over batch x 3 x 256 x 256 tiles ---
    x = torch.rand(args.batch, 3, 256, 256, device=dev)
    mpx = args.batch * 256 * 256 / 1e6
    with torch.no_grad():
        t_ref = timed(lambda: ref(x), 10)
        t_hip = timed(lambda: hip(x), 10)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t_ref_bf16 = timed(lambda: ref(x), 10)
        xc = x.contiguous(memory_format=torch.channels_last)
        refc = ref.to(memory_format=torch.channels_last)
        t_ref_cl = timed(lambda: refc(xc), 10)
        err = (hip(x) - ref(x)).abs().max().item()
        hip.precision = "bf16"
        t_hip_bf16 = timed(lambda: hip(x), 10)
        hip.precision = "fp32"
    out["tiles_forward"] = {"batch": args.batch, "eager_fp32_ms": t_ref * 1e3, "eager_fp32_channels_last_ms": t_ref_cl * 1e3,
                            "eager_bf16_autocast_ms": t_ref_bf16 * 1e3, "engine_fp32_ms": t_hip * 1e3,
                            "engine_bf16_ms": t_hip_bf16 * 1e3, "eager_fp32_mpx_s": mpx / t_ref, "engine_fp32_mpx_s": mpx / t_hip,
                            "speedup_fp32": t_ref / t_hip, "max_abs_logit_diff": err}

    # training step: forward + DiceCE + backward + AdamW
    ref_t = make_reference_unet(2022, channels=ch, strides=(2, 2, 2, 2), num_res_units=2).train().to(dev)
    hip_t = ca.UNet(channels=ch, strides=(2, 2, 2, 2), num_res_units=2)
    hip_t.load_state_dict(ref_t.state_dict())
    hip_t = hip_t.to(dev).train()
    y = torch.randint(0, 3, (args.batch, 1, 256, 256), device=dev)
    onehot = ca.one_hot(y, 3)
    loss_fn = ca.DiceCELoss(softmax=True)

    def make_step(m):
        opt = torch.optim.AdamW(m.parameters(), 6e-4)

        def step():
            opt.zero_grad()
            loss_fn(m(x), onehot).backward()
            opt.step()
        return step
    t_ref_train = timed(make_step(ref_t), 10)
    engine_step = ca.TrainStep(hip_t, loss_fn, ca.AdamW(hip_t, 6e-4))        # the engine's own loss / optimiser kernels (round 2) and lean step (round 3)
    t_hip_train = timed(lambda: engine_step(x, onehot), 10)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t_ref_train_bf16 = timed(make_step(ref_t), 10)
    out["train_step"] = {"batch": args.batch, "eager_fp32_ms": t_ref_train * 1e3, "eager_bf16_autocast_ms": t_ref_train_bf16 * 1e3,
                         "engine_fp32_ms": t_hip_train * 1e3, "speedup": t_ref_train / t_hip_train}
    del ref_t, hip_t
    torch.cuda.empty_cache()

    # --- configs[2]: the whole-slide path exactly as predict.py runs it, on this GPU ---
    img = torch.from_numpy(synthetic_wsi(args.size, 2022)).to(dev)
    n = args.size

    def ref_slide():
        with torch.no_grad():
            logits = sliding_window_inference_ref(img[None], (256, 256), 4, ref, overlap=args.overlap)   # predict.py:109
            prob = torch.softmax(logits, 1)[0, 1].cpu().numpy()                                          # predict.py:110-111
        return postprocess_c(prob)                                                                       # predict.py:113

    def hip_slide():
        _, _, mask = ca.swi_infer_image(hip, img, (256, 256), args.overlap, want_logits=False, want_mask=True)
        return ca.mask_to_instances(mask).cpu()

    t0 = time.perf_counter(); lab_ref = ref_slide(); torch.cuda.synchronize(); t_ref_slide_first = time.perf_counter() - t0
    t_ref_slide = timed(ref_slide, 1, warm=0)
    lab_hip = hip_slide().numpy()
    t_hip_slide = timed(hip_slide, 3, warm=1)
    out["slide"] = {"size": n, "overlap": args.overlap, "eager_sw_batch_4_s": t_ref_slide, "eager_first_call_s": t_ref_slide_first,
                    "engine_s": t_hip_slide, "eager_mpx_s": n * n / 1e6 / t_ref_slide, "engine_mpx_s": n * n / 1e6 / t_hip_slide,
                    "speedup": t_ref_slide / t_hip_slide, "labels_identical": bool(np.array_equal(lab_ref, lab_hip)),
                    "components": int(lab_hip.max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
