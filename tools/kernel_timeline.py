#!/usr/bin/env python3
"""Launch timeline out of a rocprofv3 --kernel-trace CSV: the last `--steps` repetitions of a periodic launch sequence.
    python tools/kernel_timeline.py <kernel_trace.csv> --period-kernel adamw [--steps 5] [--list]
Prints per-step wall time, the time no kernel is running, the time covered per queue and a per-kernel table (launches per step, mean
duration, mean gap to the previous kernel's end on the same queue)."""
import argparse, csv, collections, re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--period-kernel", default="adamw_dev")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--list", action="store_true", help="print every launch of the last step")
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void\s+", "", n).replace("(anonymous namespace)::", "")).replace("cs::", "")[:70]
marks = [i for i, r in enumerate(rows) if a.period_kernel in r[2]]
assert len(marks) > a.steps + 1, f"only {len(marks)} launches of a kernel matching {a.period_kernel!r}"
# a step = everything after the marker of step k-1 up to and including the marker of step k ... plus what follows the marker until the
# next step's first launch; simpler: cut at the marker kernel's START
cuts = [rows[i][0] for i in marks]
steps = []
for k in range(len(cuts) - a.steps - 1, len(cuts) - 1):
    steps.append([r for r in rows if cuts[k] <= r[0] < cuts[k + 1]])
walls = [(cuts[k + 1] - cuts[k]) / 1e3 for k in range(len(cuts) - a.steps - 1, len(cuts) - 1)]
print(f"steps analysed: {len(steps)}; period (marker start to marker start): mean {sum(walls) / len(walls):.1f} us, each {[round(w, 1) for w in walls]}")
idle, busy_q = [], collections.defaultdict(list)
for st in steps:
    ev = sorted((s, e) for s, e, _, _ in st)
    cover, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    span = max(e for _, e in ev) - ev[0][0]
    idle.append((span - cover) / 1e3)
    for q in {r[3] for r in st}:
        busy_q[q].append(sum(e - s for s, e, _, qq in st if qq == q) / 1e3)
print(f"launches per step: {[len(s) for s in steps]}")
print(f"time with no kernel running inside a step: mean {sum(idle) / len(idle):.1f} us")
for q, v in sorted(busy_q.items(), key=lambda kv: -sum(kv[1])):
    print(f"  queue {q}: kernel time {sum(v) / len(v):.1f} us per step")
agg = collections.OrderedDict()
for st in steps:
    last_end = {}
    for s, e, n, q in st:
        d = agg.setdefault(short(n), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        if q in last_end:
            d[2] += max(0, s - last_end[q]) / 1e3
        last_end[q] = max(last_end.get(q, 0), e)
print(f"{'kernel':72s} {'n/step':>7s} {'us each':>8s} {'us/step':>8s} {'gap each':>8s}")
for n, (c, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:72s} {c / len(steps):7.1f} {dur / c:8.2f} {dur / len(steps):8.1f} {gap / c:8.2f}")
if a.list:
    t0 = steps[-1][0][0]
    for s, e, n, q in steps[-1]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {short(n)}")
