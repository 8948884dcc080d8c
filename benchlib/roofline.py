"""`roofline` of the JSON line: the dominant kernel group of a leg, from the engine's HIP-event table.

The engine brackets every kernel launch with HIP events ON THE LAUNCH STREAM while profiling is enabled (cs_prof_enable; torch.cuda.Event
would only see torch's current stream) and accumulates, per kernel group, the elapsed milliseconds, the launches, and the ALGORITHMIC
flops / bytes of each launch (2 x multiply-adds of the layer as the kernel issues them; for HBM-bound groups one read of the
inputs + one write of the outputs).  `prof` = {group: {"ms", "launches", "flops", "bytes"}} (cellseg_amd._lib.prof_collect).  Pure python.
"""
from __future__ import annotations

from .common import MFMA_PEAK, PEAK_HBM_GBS


def dominant_kernel(prof, precision, only=None):
    """-> {"bound", "achieved", "peak", "unit", "frac", "traffic": None, "kernel", "launches", "avg_launch_ms", ...} of the group with the most
    GPU time (restricted to the groups `only(name)` accepts), or None without a table."""
    if not prof:
        return None
    total_ms = sum(v["ms"] for v in prof.values())
    if only is not None:
        prof = {n: v for n, v in prof.items() if only(n) and v["flops"] > 0}
        if not prof:
            return None
    name, k = max(prof.items(), key=lambda kv: kv[1]["ms"])
    mfma = k["flops"] > 0 and ("mfma" in name or "wino" in name or name.endswith("_fp8"))
    sec = k["ms"] / 1e3
    if mfma:
        achieved = k["flops"] / sec / 1e12
        # the peak of the instruction the launches of this group really issue: the fp32 Winograd / direct kernels run on the fp32 matrix
        # instruction whatever the mode of the other layers; in fp8 mode the e4m3 launches are their own groups (*_fp8) and every other
        # matrix kernel is a bf16 one
        peak_of = "fp32" if ("wino" in name or precision == "fp32") else ("fp8" if name.endswith("_fp8") else ("bf16" if precision == "fp8" else precision))
        peak = round(MFMA_PEAK[peak_of], 1)
        rl = {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
              "frac": round(achieved / peak, 4), "traffic": None, "peak_of": peak_of}
    else:
        achieved = k["bytes"] / sec / 1e9
        rl = {"bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
    rl["algorithmic_bytes_per_launch"] = round(k["bytes"] / k["launches"]) if k.get("bytes") else None
    if "wino" in name:
        # Winograd: `achieved` counts the multiplies the algorithm really issues on the matrix pipe (F(2x2,3x3): 16 per 2x2 output
        # tile and channel pair; F(4x4,3x3): 36 per 4x4 tile); the same launches expressed in direct-convolution FLOPs (9 per
        # output pixel: SURVEY 8(d)'s figure)
        f4 = name.endswith("wino4")
        factor = 144 / 36 if f4 else 36 / 16
        rl["algorithm"] = ("Winograd F(4x4,3x3), fp32: 36 instead of 144 multiplies per 4x4 output tile and channel pair (every stride-1 3x3 "
                           "layer with >= 32 input channels)" if f4 else
                           "Winograd F(2x2,3x3), fp32: 16 instead of 36 multiplies per output tile and channel pair")
        rl["direct_conv_equivalent"] = round(rl["achieved"] * factor, 3)
        rl["frac_direct_conv_equivalent"] = round(rl["achieved"] * factor / rl["peak"], 4)
    rl.update({"kernel": name, "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 5),
               "algorithmic_per_launch": (k["flops"] if mfma else k["bytes"]) / k["launches"],
               "share_of_gpu_time": round(k["ms"] / total_ms, 4)})
    return rl


def kernel_shares(prof, steps):
    """Per group: ms per step, share of the GPU time, TFLOP/s and GB/s of its algorithmic work; sorted by time."""
    tot = sum(v["ms"] for v in prof.values())
    return {n: {"ms_per_step": round(v["ms"] / steps, 4), "share": round(v["ms"] / tot, 4),
                "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] else None,
                "gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} for n, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
