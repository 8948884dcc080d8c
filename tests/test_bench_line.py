"""The line contract of bench.py, on CPU (the round-5 line had grown to 20 kB and the driver could not parse it).
benchlib/line.py derives the LAST stdout line - one compact JSON object under 4 kB - from the full record the legs build; here it runs
on a canned full record of a real default run (profiles/bench_full_r06_first.json) and on inflated / degenerate variants."""
import copy
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import line as benchline  # noqa: E402

CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"}


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "bench_full_r06_first.json")))


def test_compact_line_of_a_real_default_run_fits_and_carries_the_contract():
    full = _canned()
    assert len(json.dumps(full)) > 8192                              # the record itself is what no longer fits a driver's stdout tail
    line = benchline.compact_line(full, "gpurun_out/bench_full.json")
    text = json.dumps(line)
    assert len(text) < benchline.MAX_LINE_BYTES == 4096 and "\n" not in text
    assert CONTRACT <= set(line) and json.loads(text) == line
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["unit"] == "Mpixels/s"
    assert line["config"]["workload"].startswith("sliding-window inference on 3x4096x4096") and line["config"]["windows"] == 961
    assert line["config"]["workspace_bytes"] > 0 and line["config"]["workspace_bytes_10gb"] <= 10 << 30 and line["value_10gb"] > 0
    rl = line["roofline"]
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "direct_conv_equivalent"} <= set(rl)
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3 and rl["traffic"] == full["roofline"]["traffic"]
    cb = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and len(cb["sample"]) <= 260
    # the per-image headline and the pipeline beside it; one scalar per secondary leg
    assert line["latency_ms"] == full["latency_ms"] and line["value_pipelined"] == full["value_pipelined"] > line["value"]
    sec = line["secondary"]
    for key in ("reference_width_ms", "bf16_ms", "fp8_ms", "configs4_fp8_ms", "configs4_fp8_frac", "train_fp32_ms", "train_bf16_ms", "train_reference_ms",
                "tiles_bf16_n32_mpx", "strips_sim_x"):
        assert isinstance(sec[key], (int, float)), key
    assert all(not isinstance(v, (dict, list)) for v in sec.values())


def test_compact_line_stays_under_the_limit_whatever_the_record_grows_to():
    full = _canned()
    full["metric"] += " and a lot more prose" * 200
    full["config"]["workload"] += " with further detail" * 300
    full["config"]["note"] = "n" * 5000
    full["cpu_baseline"]["sample"] += "; another sentence" * 400
    full["roofline"]["traffic_source"] = "measured by this run: " + "x" * 4000
    full["strips_error"] = "RuntimeError: " + "e" * 9000
    full["other_precisions"].update({f"mode{i}": {"ms_per_step": 1.0 + i} for i in range(300)})        # legs the assembler does not know stay out
    full["new_leg_of_a_later_round"] = {"table": list(range(5000))}
    line = benchline.compact_line(full, "gpurun_out/bench_full.json")
    assert len(json.dumps(line)) < 4096 and CONTRACT <= set(line)
    assert line["roofline"]["traffic_source"] == "live" and line["strips_error"].endswith("~") and "new_leg_of_a_later_round" not in line


def test_compact_line_without_secondary_legs_and_for_the_other_headlines():
    full = _canned()
    lean = {k: copy.deepcopy(v) for k, v in full.items() if k in CONTRACT or k in ("rccl_ranks", "latency_ms")}
    lean["cpu_baseline"] = None                                           # --no-cpu-baseline
    lean["roofline"].pop("traffic", None)                                 # no counter pass: the key stays, as null
    line = benchline.compact_line(lean)
    assert CONTRACT <= set(line) and line["cpu_baseline"] is None and line["roofline"]["traffic"] is None
    assert "secondary" not in line and "full_record" not in line
    # N > 1: ranks, per-rank windows and the three phases of the strips leg travel in the line
    multi = dict(copy.deepcopy(lean), n_gpus=8, rccl_ranks=8, scaling="strong", exchange_ms=0.41, gather_ms=0.22, cc_ms=1.9, value_slides=3600.0, value_strips=3100.0)
    multi["config"].update(windows_per_rank=[120] * 7 + [121], seam_transport="p2p", cc="turns")
    line = benchline.compact_line(multi)
    assert line["rccl_ranks"] == 8 and line["config"]["windows_per_rank"][-1] == 121 and line["config"]["seam_transport"] == "p2p"
    assert (line["exchange_ms"], line["gather_ms"], line["cc_ms"]) == (0.41, 0.22, 1.9)
    # --mode train
    train = {"metric": "images/s data-parallel training step", "value": 5600.0, "unit": "images/s", "n_gpus": 1, "rccl_ranks": 1, "steps": 20, "warmup": 3,
             "ms_per_step": 5.7, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
             "config": {"workload": "training step on 32x3x256x256", "global_batch": 32, "parallelism": "dp1"}, "roofline": copy.deepcopy(full["roofline"]),
             "allreduce_ms": 0.08, "train_step": {"ms_per_step": 5.7, "images_per_s": 5600.0, "kernels": {"wgrad_mfma": {}}}}
    line = benchline.compact_line(train)
    assert line["allreduce_ms"] == 0.08 and line["config"]["global_batch"] == 32 and line["secondary"]["train_fp32_ms"] == 5.7


def test_emit_prints_one_line_and_writes_the_full_record(tmp_path):
    full = _canned()
    out, err = io.StringIO(), io.StringIO()
    line = benchline.emit(full, str(tmp_path), stream=out, err=err)
    text = out.getvalue()
    assert text.count("\n") == 1 and json.loads(text) == line and len(text) < 4096
    assert line["full_record"] == "gpurun_out/bench_full.json"
    assert json.load(open(tmp_path / "gpurun_out" / "bench_full.json")) == full
    assert err.getvalue().startswith("bench.py full record: ") and json.loads(err.getvalue().split(": ", 1)[1]) == full


def test_roofline_of_the_dominant_kernel_from_a_hip_event_table():
    """benchlib.roofline.dominant_kernel on a canned HIP-event table: the group with the most GPU time, ALGORITHMIC flops / its launch time
    against the peak of the instruction it issues (MI355X_MICROARCH.md), Winograd also in direct-convolution FLOPs, HBM-bound groups in GB/s."""
    from benchlib.roofline import dominant_kernel, kernel_shares
    prof = {"conv3x3_s1_wino4": {"ms": 132.0, "launches": 180, "flops": 180 * 52.4e9, "bytes": 180 * 0.91e9},
            "conv3x3_s2_mfma": {"ms": 75.0, "launches": 80, "flops": 80 * 145e9, "bytes": 80 * 1.2e9},
            "conv3x3_s1_fp8": {"ms": 60.0, "launches": 100, "flops": 100 * 1.4e12, "bytes": 1e9},
            "norm_finalize": {"ms": 20.0, "launches": 340, "flops": 0.0, "bytes": 340 * 4e6}}
    rl = dominant_kernel(prof, "fp32")
    assert rl["kernel"] == "conv3x3_s1_wino4" and rl["bound"] == "mfma" and rl["peak"] == 157.3 and rl["peak_of"] == "fp32" and rl["traffic"] is None
    assert abs(rl["achieved"] - 180 * 52.4e9 / 0.132 / 1e12) < 0.01 and abs(rl["frac"] - rl["achieved"] / 157.3) < 1e-3
    assert abs(rl["direct_conv_equivalent"] - 4 * rl["achieved"]) < 0.01 and abs(rl["avg_launch_ms"] - 132.0 / 180) < 1e-4
    assert abs(rl["share_of_gpu_time"] - 132.0 / 287.0) < 1e-3 and rl["algorithmic_bytes_per_launch"] == round(0.91e9)
    fp8 = dominant_kernel(prof, "fp8", only=lambda n: n.endswith("_fp8"))
    assert fp8["kernel"] == "conv3x3_s1_fp8" and fp8["peak"] == 5000.0 and abs(fp8["achieved"] - 100 * 1.4e12 / 0.060 / 1e12) < 0.5
    bf = dominant_kernel(prof, "fp8", only=lambda n: "mfma" in n)
    assert bf["peak"] == 2500.0 and bf["peak_of"] == "bf16"              # in the mixed mode every other matrix kernel is a bf16 one
    hbm = dominant_kernel({"norm_finalize": prof["norm_finalize"]}, "fp32")
    assert hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and hbm["peak"] == 8000.0 and abs(hbm["achieved"] - 340 * 4e6 / 0.020 / 1e9) < 0.1
    assert dominant_kernel({}, "fp32") is None and dominant_kernel(prof, "fp32", only=lambda n: False) is None
    table = kernel_shares(prof, 10)
    assert list(table)[0] == "conv3x3_s1_wino4" and abs(sum(v["share"] for v in table.values()) - 1.0) < 1e-3 and table["norm_finalize"]["tflops"] is None


def test_bench_driver_is_small_and_its_legs_live_in_benchlib():
    """bench.py = argument parsing + leg selection; helpers and the line assembler are importable without a GPU."""
    import importlib
    for mod in ("common", "cpu_baseline", "eager", "accuracy", "ranks", "roofline", "traffic", "line"):
        importlib.import_module(f"benchlib.{mod}")
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "benchline.emit(" in src and src.count("print(json.dumps(") == 0          # one way out: the compact line
    assert len(src.splitlines()) < 1000


@pytest.mark.gpu
def test_the_default_run_ends_with_one_parsable_compact_line():
    """The REAL default command (what the driver runs, with its own flags): the last stdout line round-trips json.loads, is the only
    stdout line, stays under 4 kB and carries roofline + cpu_baseline; the run takes well under the 90 s asked for."""
    import time
    from conftest import run_bench
    t0 = time.time()
    line, full = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", timeout=600)
    took = time.time() - t0
    assert CONTRACT <= set(line) and line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    assert line["roofline"]["frac"] > 0.3 and line["roofline"]["traffic"] and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert abs(line["value"] - 4096 * 4096 / 1e6 / (line["ms_per_step"] / 1e3)) < 0.01 * line["value"]
    assert line["value"] > 50 and line["value_pipelined"] >= 0.98 * line["value"] and abs(line["latency_ms"] - line["ms_per_step"]) < 0.05 * line["ms_per_step"]
    assert line["secondary"]["configs4_fp8_ms"] > 0 and full["bench_seconds"] < 90 and took < 120, (full["bench_seconds"], took, full["leg_seconds"])
