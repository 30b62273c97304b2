#!/usr/bin/env python3
"""Per-size kernel durations (or counters) of tools/measure_slab_fit.py from a rocprofv3 run of it.

    slab_fit_from_trace.py <rocprof dir> <the tool's own jsonl from that run> [pmc]

The tool launches 200 warm-up sweeps, then 3 x (3 + 20) sweeps per size in ascending size order; the trace's dispatches of the sweep kernel are
cut into those groups by order.  Kernel-trace mode prints the mean / min duration of the 60 timed dispatches of every size and the fit
duration = a + b * rounds over the whole-round sizes; pmc mode the per-size mean of every counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import numpy as np


def main():
    root, jl = sys.argv[1], sys.argv[2]
    pmc = len(sys.argv) > 3 and sys.argv[3] == "pmc"
    sizes = [json.loads(l) for l in open(jl) if l.startswith("{") and '"sites"' in l]
    pat = "*counter_collection.csv" if pmc else "*kernel_trace.csv"
    rows = []
    for path in glob.glob(os.path.join(root, "**", pat), recursive=True):
        rows += list(csv.DictReader(open(path)))
    rows = [r for r in rows if "sweep_kernel<2, 3" in r["Kernel_Name"]]
    if pmc:
        per = defaultdict(dict)
        for r in rows:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        order = [per[k] for k in sorted(per)]
    else:
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        order = [{"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6} for r in rows]
    warm, per_size = 200, 69
    if len(order) < warm + per_size * len(sizes):
        raise SystemExit(f"{len(order)} dispatches of the sweep kernel, expected {warm + per_size * len(sizes)}")
    out = []
    for i, rec in enumerate(sizes):
        chunk = order[warm + i * per_size: warm + (i + 1) * per_size]
        timed = [c for j, c in enumerate(chunk) if j % 23 >= 3]
        o = {"sites": rec["sites"], "what": rec["what"], "rounds": rec["rounds"]}
        for key in timed[0]:
            vals = [c[key] for c in timed]
            o[key + "_mean"] = float(np.mean(vals))
            if not pmc:
                o[key + "_min"] = float(np.min(vals))
        out.append(o)
        print(json.dumps(o))
    if not pmc:
        whole = [o for o in out if o["what"].endswith("whole rounds")]
        x, y = np.array([o["rounds"] for o in whole]), np.array([o["ms_mean"] for o in whole])
        b, a = np.polyfit(x, y, 1)
        print(json.dumps({"fit": "kernel-trace duration (ms) = a + b * rounds over the whole-round sizes", "a_ms": round(float(a), 5), "b_ms_per_round": round(float(b), 5)}))


if __name__ == "__main__":
    main()
