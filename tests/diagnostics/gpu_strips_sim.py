#!/usr/bin/env python3
"""Diagnostic: how long one rank of the 8-rank window-row split works per 4096^2 slide, for several engine batches / stream counts
(one GPU doing rank R's share; see bench.py `strips_sim`).  python tests/diagnostics/gpu_strips_sim.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch  # noqa: E402

import cellseg_amd as ca  # noqa: E402
from bench import WIDTHS, make_model, synthetic_wsi  # noqa: E402

dev = torch.device("cuda:0")
model = make_model(WIDTHS["baseline32"]).to(dev)
img = torch.from_numpy(synthetic_wsi(4096, 2022)).to(dev)
roi, ov = (256, 256), 0.5
cc_stream, ev = torch.cuda.Stream(dev), torch.cuda.Event()


def timed(fn, k=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


with torch.no_grad():
    _, _, full_mask = ca.swi_infer_image(model, img, roi, ov, want_logits=False, want_mask=True)

    def full():
        _, _, m = ca.swi_infer_image(model, img, roi, ov, want_logits=False, want_mask=True)
        ev.record(); cc_stream.wait_event(ev)
        with torch.cuda.stream(cc_stream):
            ca.mask_to_instances(m)
        m.record_stream(cc_stream)
    t_full = timed(full)
    print(f"full slide {t_full:.3f} ms")
    for cc in (True, False):
        for streams in (1, 2):
            for eb in (31, 62, 64, 124):
                def strip():
                    ca.swi_infer_strip(model, img, roi, ov, 3, 8, engine_batch=eb, streams=streams, exchange=lambda *a: None)
                    if cc:
                        ev.record(); cc_stream.wait_event(ev)
                        with torch.cuda.stream(cc_stream):
                            ca.mask_to_instances(full_mask)
                t = timed(strip)
                print(f"cc={cc} streams={streams} engine_batch={eb}: {t:.3f} ms  ratio {t_full / t:.2f}")
