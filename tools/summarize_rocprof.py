#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  summarize_rocprof.py trace <dir> <out.csv>   kernel, calls, total_ms, avg_ms, min_ms, max_ms  (from *_kernel_trace.csv)
  summarize_rocprof.py pmc   <dir> <out.csv>   kernel, counter, dispatches, mean, min, max       (from *_counter_collection.csv)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no {pattern} under {root}")
    return hits


def short(name):
    return name.split("(")[0].strip()


def trace(root, out):
    acc = defaultdict(list)
    for path in find(root, "*kernel_trace.csv"):
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"])
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), f"{sum(v):.4f}", f"{sum(v) / len(v):.4f}", f"{min(v):.4f}", f"{max(v):.4f}"])


def pmc(root, out):
    acc = defaultdict(list)
    for path in find(root, "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
        for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
