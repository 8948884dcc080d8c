#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of the SQ / GRBM counters into profiles/pmc_mfma_util_<tag>.json.

    python tools/pmc_mfma_summary.py <round-tag> <counter_collection.csv>

Expected counters (one pass, <= 8 counters):
    GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
    SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / kernel time
(GRBM_GUI_ACTIVE is summed over the 8 XCDs; kernel time = End - Start of the dispatch under the counter pass).
"""
import collections
import csv
import json
import re
import sys


def main():
    tag, path = sys.argv[1:3]
    extra = " ".join(sys.argv[3:])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    secs = collections.defaultdict(float)
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "cs::" not in name:
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            secs[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            calls[name] += 1
    out = {"source": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES "
                     "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-secondary " + extra,
           "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE/8 / kernel time",
           "kernels": {}}
    for name in sorted(agg, key=lambda k: -secs[k]):
        c = agg[name]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        out["kernels"][name] = {"launches": calls[name], "seconds_under_pmc": round(secs[name], 5),
                                "clock_ghz": round(cyc / secs[name] * 1e-9, 3),
                                "mfma_busy_frac_of_simd_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
                                "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                "wait_any": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                                "active_inst": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)}
    json.dump(out, open(f"profiles/pmc_mfma_util_{tag}.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:60]:60s} mfma busy {v['mfma_busy_frac_of_simd_cycles']:.3f}  clock {v['clock_ghz']:.2f} GHz  launches {v['launches']}")


if __name__ == "__main__":
    main()
