"""roofline.traffic measured by the bench run itself: two `rocprofv3 --pmc` child passes of one slide."""
from __future__ import annotations

import os
import subprocess
import sys
import time

from .common import ROOT

def live_pmc_traffic(group_name, timeout_s=120):
    """HBM bytes per launch of one kernel group, MEASURED by this run: two child passes of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    (counter passes on their own, no tracing flags, the program itself behind `--`) around ONE slide of the default command on one stream,
    corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE reports half the
    bytes of wide coalesced reads) - tools/pmc_summary.py's arithmetic, grouping the kernel symbols under the engine's profiling ids.
    Children of this process (never an exec: the GPU is initialised here); -> (bytes_per_launch, detail dict) or (None, {"error": ...})."""
    import csv
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, {"error": "this run is itself under a profiler"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import group
    finally:
        sys.path.pop(0)
    out_dir = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    got, t0 = {}, time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out_dir, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1",
                   "--warmup", "0", "--streams", "1", "--cc-stream", "0", "--no-cpu-baseline", "--no-secondary"]
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            files = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if cp.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {counter}: rc {cp.returncode}, {len(files)} counter files: {cp.stderr[-200:]}"}
            tot, cnt = 0.0, 0
            for path in files:
                for r in csv.DictReader(open(path)):
                    if r.get("Counter_Name", counter) == counter and group(r["Kernel_Name"]) == group_name:
                        tot += float(r["Counter_Value"])
                        cnt += 1
            if not cnt:
                return None, {"error": f"no {group_name} launches in the {counter} pass"}
            got[counter] = (tot / cnt, cnt)
    except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
        return None, {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    read_b = 2.0 * got["FETCH_SIZE"][0] * 1024
    write_b = got["WRITE_SIZE"][0] * 1024
    return round(read_b + write_b), {"hbm_read_bytes_per_launch": round(read_b), "hbm_write_bytes_per_launch": round(write_b),
                                     "launches_counted": got["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}

