#!/usr/bin/env python3
"""Transcribes the FORMAT of the reference's committed run_vcf outputs into tests/golden/format_exemplars.json
(header lines, a few data rows, FALSTA record names and token shapes).  Data, not source: the inputs that produced these
files are not in the reference tree, so they pin the output surface, not numbers.
Run in the build container only (reads /root/reference/data)."""
import gzip
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shape(tok: str) -> str:
    if re.fullmatch(r"-?\d+\.\d{6}", tok):
        return "d.dddddd"
    if re.fullmatch(r"-?\d+", tok):
        return "0" if tok == "0" else "int"
    return tok  # NA, Infinity, -Infinity, names


def main():
    out = {"source": "SauersML/ferromic data/output.csv, data/FST_data.tsv, data/per_site_diversity_output.falsta.gz (format exemplars)"}
    with open(os.path.join(REF, "output.csv")) as fh:
        lines = fh.read().splitlines()
    out["output_csv"] = {"header": lines[0], "columns": len(lines[0].split(",")), "rows": lines[1:4],
                         "cell_shapes_by_column": [sorted({shape(l.split(",")[c]) for l in lines[1:]}) for c in range(len(lines[0].split(",")))]}
    with open(os.path.join(REF, "FST_data.tsv")) as fh:
        lines = fh.read().splitlines()
    out["hudson_tsv"] = {"header": lines[0], "columns": len(lines[0].split("\t")), "rows": lines[1:4],
                         "cell_shapes_by_column": [sorted({shape(l.split("\t")[c]) for l in lines[1:]}) for c in range(len(lines[0].split("\t")))]}
    recs = []
    shapes = set()
    with gzip.open(os.path.join(REF, "per_site_diversity_output.falsta.gz"), "rt") as fh:
        name = None
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                name = line
                continue
            toks = line.split(",")
            shapes |= {shape(t) for t in toks}
            if len(recs) < 8:
                m = re.fullmatch(r">(\w+?)_chr_(\w+)_start_(\d+)_end_(\d+)_group_(\d)", name)
                recs.append({"header": name, "tokens": len(toks), "span": int(m.group(4)) - int(m.group(3)) + 1, "track": m.group(1), "group": int(m.group(5)),
                             "first_tokens": toks[:8]})
    out["diversity_falsta"] = {"records": recs, "token_shapes": sorted(shapes)}
    path = os.path.join(ROOT, "tests", "golden", "format_exemplars.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
