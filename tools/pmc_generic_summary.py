#!/usr/bin/env python3
"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected (counter_collection.csv), as ratios to SQ_WAVE_CYCLES /
GRBM_GUI_ACTIVE where those are present.   python tools/pmc_generic_summary.py <counter_collection.csv> [out.json]"""
import collections, csv, json, re, sys


def main():
    path = sys.argv[1]
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
    out = {}
    for name in sorted(agg, key=lambda k: -secs[k])[:12]:
        c = dict(agg[name])
        row = {"launches": calls[name], "ms": round(secs[name] * 1e3, 3)}
        wc = c.get("SQ_WAVE_CYCLES")
        for k, v in c.items():
            row[k] = v
            if wc and k.startswith("SQ_") and k != "SQ_WAVE_CYCLES":
                row[k + "/WAVE_CYCLES"] = round(v / wc, 4)
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        out[name] = row
        print(name[:90])
        for k, v in row.items():
            if "/" in k or k in ("launches", "ms", "mfma_busy_frac"):
                print(f"     {k:45s} {v}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
