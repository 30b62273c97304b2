#!/usr/bin/env python3
"""Prints kernel ms and HBM fraction of tools/measure_configs.py JSON lines (development helper)."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            j = json.loads(line)
            tag = f"mfma={j['FMH_COUNTS_MFMA']} " if "FMH_COUNTS_MFMA" in j else ""
            print(f"{tag}{j['config']} {j['layout']} kernel_ms {j['kernel_ms']:.4f} frac {j['frac_of_8TBs']:.3f}")
