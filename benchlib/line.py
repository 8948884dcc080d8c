"""The ONE stdout line of bench.py.

The driver keeps only a tail of a run's stdout and parses its last line: round 5's record had grown to 20 kB and was not parsed.  So the
legs build a FULL record (every object they measured, written to `gpurun_out/bench_full.json` and to stderr) and this module derives the
COMPACT line from it: the contract keys, `roofline` and `cpu_baseline` whole (minus their prose), the workload named in one sentence, and
ONE scalar per secondary leg.  Pure python on plain dicts: tests/test_host_cpu.py runs it on a canned record without a GPU.
"""
from __future__ import annotations

import json
import os

MAX_LINE_BYTES = 4096
# the contract of the task description + what VERDICT r05 asks the line to carry
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "peak_of", "avg_launch_ms", "launches",
                 "share_of_gpu_time", "direct_conv_equivalent", "frac_direct_conv_equivalent", "traffic_source")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "value_skimage_chain")
SCALARS = ("rccl_ranks", "collective_backend", "latency_ms", "value_pipelined", "ms_per_step_pipelined", "value_device", "value_incl_tiff", "value_10gb",
           "value_slides", "value_strips", "strips_error", "exchange_ms", "gather_ms", "cc_ms", "allreduce_ms", "bench_seconds")
CONFIG_KEYS = ("workload", "windows", "windows_per_rank", "engine_batch", "streams", "precision", "parallelism", "global_batch", "workspace_bytes",
               "engine_batch_10gb", "workspace_bytes_10gb", "seam_transport", "cc", "instances_found", "note")


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short_traffic_source(s):
    if not s:
        return None
    return "live" if s.startswith("measured by this run") else ("quoted:profiles/pmc_traffic.json" if s.startswith("QUOTED") else _clip(s, 40))


def secondary_scalars(full: dict) -> dict:
    """One number per secondary leg (ms per step unless the key says otherwise); legs that did not run are left out."""
    pick = {
        "reference_width_ms": _get(full, "reference_width", "ms_per_step"),
        "reference_width_frac": _get(full, "reference_width", "roofline", "frac"),
        "bf16_ms": _get(full, "other_precisions", "bf16", "ms_per_step"),
        "fp8_ms": _get(full, "other_precisions", "fp8", "ms_per_step"),
        "reference_width_ms_pipelined": _get(full, "reference_width", "ms_per_step_pipelined"),
        "bf16_ms_pipelined": _get(full, "other_precisions", "bf16", "ms_per_step_pipelined"),
        "fp8_ms_pipelined": _get(full, "other_precisions", "fp8", "ms_per_step_pipelined"),
        "configs4_fp8_ms_pipelined": _get(full, "configs4_fp8", "ms_per_step_pipelined"),
        "fp16x3_ms": _get(full, "other_precisions", "fp16x3", "ms_per_step"),
        "bf16x6_ms": _get(full, "other_precisions", "bf16x6", "ms_per_step"),
        "configs4_fp8_ms": _get(full, "configs4_fp8", "ms_per_step"),
        "configs4_fp8_mpx": _get(full, "configs4_fp8", "value"),
        "configs4_fp8_frac": _get(full, "configs4_fp8", "roofline_by_dtype", "fp8", "frac"),
        "configs4_fp8_frac_slide": _get(full, "configs4_fp8", "frac_of_fp8_peak_whole_slide"),
        "train_fp32_ms": _get(full, "train_step", "ms_per_step"),
        "train_bf16_ms": _get(full, "train_step", "bf16", "ms_per_step"),
        "train_fp16x3_ms": _get(full, "train_step", "fp16x3", "ms_per_step"),
        "train_images_per_s": _get(full, "train_step", "images_per_s"),
        "train_reference_ms": _get(full, "train_step_reference", "ms_per_step"),
        "train_reference_bf16_ms": _get(full, "train_step_reference", "bf16", "ms_per_step"),
        "tiles_fp32_n32_mpx": _get(full, "tiles", "fp32", "n32", "value"),
        "tiles_bf16_n32_mpx": _get(full, "tiles", "bf16", "n32", "value"),
        "tiles_fp32_n8_mpx": _get(full, "tiles", "fp32", "n8", "value"),
        "tiles_bf16_n8_mpx": _get(full, "tiles", "bf16", "n8", "value"),
        "strips_sim_x": _get(full, "strips_sim", "ratio"),
        "strips_sim_ms": _get(full, "strips_sim", "ms_per_slide"),
        "eager_slide_fp32_x": _get(full, "eager_rocm", "slide", "fp32", "ratio"),
        "eager_train_fp32_x": _get(full, "eager_rocm", "train_step_batch32", "fp32", "ratio"),
        "eager_tiles_bf16_x": _get(full, "eager_rocm", "tiles_forward_n32", "bf16", "ratio"),
        "bf16_delta_f1": _get(full, "other_precisions", "bf16", "delta_f1_vs_truth"),
        "fp8_delta_f1": _get(full, "other_precisions", "fp8", "delta_f1_vs_truth"),
        "configs4_bf16_delta_f1": _get(full, "configs4_fp8", "accuracy_on_trained_net", "bf16", "delta_f1_vs_truth"),
        "configs4_fp8_delta_f1": _get(full, "configs4_fp8", "accuracy_on_trained_net", "fp8", "delta_f1_vs_truth"),
        "cpu_train_ms": _get(full, "cpu_baseline", "train_step", "baseline32", "ms_per_step"),
        "strips_equal_to_single_gpu": _get(full, "strips", "equal_to_single_gpu_labels") if "strips" in full else _get(full, "config", "equal_to_single_gpu_labels"),
    }
    return {k: v for k, v in pick.items() if v is not None}


def compact_line(full: dict, full_path: str | None = None) -> dict:
    """full record -> the compact dict whose json.dumps is the last stdout line (always < MAX_LINE_BYTES: prose is clipped, and if a
    record still does not fit, the secondary scalars are dropped from the end - the contract keys never are)."""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        out[k] = full.get(k)
    out["metric"] = _clip(out["metric"], 200)
    out["dtype"] = _clip(out["dtype"], 80)
    cfg = full.get("config") or {}
    out["config"] = {k: (_clip(cfg[k], 420) if k == "workload" else _clip(cfg[k], 160)) for k in CONFIG_KEYS if cfg.get(k) is not None}
    rl = full.get("roofline")
    if rl:
        out["roofline"] = {k: rl[k] for k in ROOFLINE_KEYS if rl.get(k) is not None}
        out["roofline"].setdefault("traffic", None)                      # the contract names the key: null when no counter pass ran
        if "traffic_source" in out["roofline"]:
            out["roofline"]["traffic_source"] = _short_traffic_source(out["roofline"]["traffic_source"])
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (_clip(cb[k], 260) if k == "sample" else cb[k]) for k in CPU_KEYS if cb.get(k) is not None}
    else:
        out["cpu_baseline"] = None
    for k in SCALARS:
        if full.get(k) is not None:
            out[k] = _clip(full[k], 300)
    rbd = full.get("roofline_by_dtype")
    if rbd:
        out["roofline_by_dtype"] = {d: {k: r[k] for k in ("kernel", "achieved", "peak", "frac") if k in r} for d, r in rbd.items() if r}
    sec = secondary_scalars(full)
    if sec:
        out["secondary"] = sec
    if full_path:
        out["full_record"] = full_path
    # belt and braces: the line must fit whatever a future leg adds to the record
    while len(json.dumps(out)) >= MAX_LINE_BYTES and out.get("secondary"):
        out["secondary"].popitem()
    if len(json.dumps(out)) >= MAX_LINE_BYTES:
        out["config"] = {"workload": _clip(cfg.get("workload", ""), 200)}
        out.pop("roofline_by_dtype", None)
    assert len(json.dumps(out)) < MAX_LINE_BYTES
    return out


def write_full(full: dict, root: str) -> str | None:
    """The full record beside the run (gpurun_out/ is merged back from the GPU box); -> the path relative to the repo root, or None."""
    try:
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        name = os.environ.get("BENCH_FULL_NAME") or ("bench_full.json" if full.get("n_gpus", 1) == 1 and full.get("unit") == "Mpixels/s" else \
            f"bench_full_{'train' if full.get('unit') == 'images/s' else 'infer'}_n{full.get('n_gpus', 1)}.json")
        with open(os.path.join(d, name), "w") as f:
            json.dump(full, f, indent=1)
        return f"gpurun_out/{name}"
    except OSError:
        return None


def emit(full: dict, root: str, stream=None, err=None) -> dict:
    """Write the full record (file + stderr) and print the compact line as the LAST stdout line; -> the compact dict."""
    import sys
    stream, err = stream or sys.stdout, err or sys.stderr
    path = write_full(full, root)
    line = compact_line(full, path)
    err.write("bench.py full record: " + json.dumps(full) + "\n")
    err.flush()
    stream.write(json.dumps(line) + "\n")
    stream.flush()
    return line
