#!/usr/bin/env python3
"""Summarise rocprofv3 output into profiles/: per-kernel time (kernel-trace --stats) and HBM traffic (PMC passes).

    python tools/pmc_summary.py <round-tag> <kernel_stats.csv> <fetch_counter.csv> <write_counter.csv>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, "HBM" section; the
swi_finalize kernel confirms it here: it reads the 756 MB window-logit buffer once and FETCH_SIZE says 378 MB).
Kernel symbols are grouped under the engine's profiling ids so bench.py can match its HIP-event timings.
"""
import collections
import csv
import json
import re
import sys

GROUPS = [(r"conv_wino4[xyz]?_kernel", "conv3x3_s1_wino4"), (r"conv_wino1?_kernel", "conv3x3_s1_wino"), (r"conv_(mfma|wreg|bf16s|fp8)_kernel<1, 3,|conv_bf16_rows(32|64)_kernel", "conv3x3_s1_mfma"), (r"conv_(mfma|wreg|bf16s|fp8)_kernel<2, 3,|conv_bf16_rows_s2_kernel", "conv3x3_s2_mfma"),
          (r"conv_(mfma|bf16s|fp8)_kernel<1, 1,", "conv1x1_mfma"), (r"conv_first(16)?(_rec)?_kernel|image_records_kernel", "conv_first_mfma"),
          (r"convt_(mfma|bf16s|fp8|wino|bf16_rows)_kernel", "convT3x3s2_mfma"), (r"convt_small_kernel|convt_top_gemm(16)?_kernel", "convT_top_mfma"),
          (r"conv_last_kernel", "conv_last_valu"), (r"norm_finalize_kernel", "norm_finalize"),
          (r"norm_act_add\w*_kernel|act_quant_kernel", "norm_act_add"), (r"swi_finalize_kernel", "swi_finalize"),
          (r"ccl_(\w+)_kernel", None)]


def group(name):
    for pat, g in GROUPS:
        m = re.search(pat, name)
        if m:
            return g or ("ccl_" + m.group(m.lastindex))
    return None


def counters(path):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        g = group(r["Kernel_Name"])
        if g:
            d[g][0] += float(r["Counter_Value"])
            d[g][1] += 1
    return d


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    extra = " ".join(sys.argv[5:])
    out = {"source": "rocprofv3 --kernel-trace --stats and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                     f"`python3 bench.py --streams 1 --no-cpu-baseline --no-secondary {extra}`".rstrip() + (" (default workload)" if not extra else ""),
           "kernels": {}}
    t = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(stats)):
        g = group(r["Name"])
        if g:
            t[g][0] += float(r["TotalDurationNs"])
            t[g][1] += int(r["Calls"])
    f, w = counters(fetch), counters(write)
    for g in sorted(t, key=lambda g: -t[g][0]):
        e = {"calls": t[g][1], "avg_duration_us": round(t[g][0] / t[g][1] / 1e3, 2), "total_ms": round(t[g][0] / 1e6, 3)}
        if g in f and g in w and f[g][1] and w[g][1]:
            fetch_b = 2.0 * f[g][0] / f[g][1] * 1024
            write_b = w[g][0] / w[g][1] * 1024
            e.update({"hbm_read_bytes_per_launch": round(fetch_b), "hbm_write_bytes_per_launch": round(write_b),
                      "hbm_bytes_per_launch": round(fetch_b + write_b)})
        out["kernels"][g] = e
    json.dump(out, open(f"profiles/pmc_traffic_{tag}.json", "w"), indent=1)
    if re.fullmatch(r"r\d+", tag):                      # the default (fp32) workload's summary is what bench.py quotes as `traffic`
        json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    for g, e in out["kernels"].items():
        print(f"{g:20s} calls {e['calls']:5d} avg {e['avg_duration_us']:9.2f} us  hbm/launch {e.get('hbm_bytes_per_launch', 0)/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
