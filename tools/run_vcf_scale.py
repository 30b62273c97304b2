#!/usr/bin/env python3
"""End-to-end run_vcf at reduced C4 scale (SURVEY.md 8(d) "config 4 honesty note").

Writes a REAL synthetic VCF (default 200 000 sites x 2 500 diploid samples = 5 000 haplotypes, ~3.5 GB of
text), a matching reference FASTA/.fai and a one-region config, runs the run_vcf binary on it (text ingest ->
packed matrix -> GPU sweeps -> CSV/FALSTA/TSV writers) and checks the CSV row against the `ferromic` Python
drop-in fed the same genotypes through Population.from_numpy (itself parity-tested against the oracle).
Prints one JSON line with the stage timings.  Needs a GPU; nothing here reads /root/reference.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.environ.get("RUN_VCF_BIN") or os.path.join(ROOT, "ferromic_amd", "bin", "run_vcf")  # RUN_VCF_BIN: another build of the binary (A/Bs)


class BgzfWriter:
    """Streams bytes into BGZF blocks (independent gzip members with a 'BC' size subfield, then the EOF block)."""

    def __init__(self, path, block=65280):
        self.fh, self.block, self.buf = open(path, "wb"), block, bytearray()

    def _emit(self, chunk: bytes):
        import struct
        import zlib

        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(chunk) + comp.flush()
        bsize = 12 + 6 + len(body) + 8
        self.fh.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
        self.fh.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

    def write(self, data: bytes):
        self.buf += data
        while len(self.buf) >= self.block:
            self._emit(bytes(self.buf[:self.block]))
            del self.buf[:self.block]

    def close(self):
        if self.buf:
            self._emit(bytes(self.buf))
        self._emit(b"")
        self.fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_inputs(tmp: str, sites: int, samples: int, seed: int, compress: str = "none"):
    rng = np.random.default_rng(seed)
    gaps = rng.integers(1, int(os.environ.get("RUN_VCF_GAP_MAX", "7")), size=sites)  # RUN_VCF_GAP_MAX=200: a variant every 100 bp on average
    pos = np.cumsum(gaps)  # 1-based VCF positions
    length = int(pos[-1]) + 10
    names = [f"SYN{i:05d}" for i in range(samples)]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = bases[rng.integers(0, 4, size=length)].tobytes().decode()
    hdr = ">chr1\n"
    with open(os.path.join(tmp, "ref.fa"), "w") as fh:
        fh.write(hdr)
        for i in range(0, length, 60):
            fh.write(seq[i:i + 60] + "\n")
    with open(os.path.join(tmp, "ref.fa.fai"), "w") as fh:
        fh.write(f"chr1\t{length}\t{len(hdr)}\t60\t61\n")
    with open(os.path.join(tmp, "ann.gtf"), "w") as fh:
        fh.write('chr1\t.\tCDS\t1\t100\t.\t+\t0\tgene_id "g"; transcript_id "t";\n')
    os.makedirs(os.path.join(tmp, "vcfs"), exist_ok=True)
    half = samples // 2
    geno = np.empty((sites, samples, 2), dtype=np.uint8)
    vcf_path = os.path.join(tmp, "vcfs", "chr1.vcf" + ("" if compress == "none" else ".gz"))
    import gzip

    opener = {"none": lambda p: open(p, "wb"), "gzip": lambda p: gzip.open(p, "wb", compresslevel=1), "bgzf": BgzfWriter}[compress]
    with opener(vcf_path) as fh:
        fh.write(("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n").encode())
        chunk = 4096
        for c0 in range(0, sites, chunk):
            n = min(chunk, sites - c0)
            base = rng.beta(0.8, 0.8, size=(n, 1, 1))
            div = rng.normal(0.0, 0.05, size=(n, 1, 1))
            f = np.clip(np.where(np.arange(samples)[None, :, None] < half, base + div, base - div), 0.001, 0.999)
            g = (rng.random((n, samples, 2)) < f).astype(np.uint8)
            geno[c0:c0 + n] = g
            cells = np.empty((n, samples, 7), dtype=np.uint8)
            cells[:, :, 0] = g[:, :, 0] + ord("0")
            cells[:, :, 1] = ord("|")
            cells[:, :, 2] = g[:, :, 1] + ord("0")
            cells[:, :, 3] = ord(":")
            cells[:, :, 4] = ord("9")
            cells[:, :, 5] = ord("9")
            cells[:, :, 6] = ord("\t")
            cells[:, -1, 6] = ord("\n")
            for i in range(n):
                fh.write(f"chr1\t{pos[c0 + i]}\t.\tA\tC\t.\tPASS\t.\tGT:GQ\t".encode())
                fh.write(cells[i].tobytes())
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"
    cells = ["0|0" if i < half else "1|1" for i in range(samples)]
    cfg += f"chr1\t1\t{length}\t1\tid\tpass\tinv\t" + "\t".join(cells) + "\n"
    with open(os.path.join(tmp, "config.tsv"), "w") as fh:
        fh.write(cfg)
    return geno, pos, length, os.path.getsize(vcf_path)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=2_500)
    ap.add_argument("--seed", type=int, default=202_500)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--compress", choices=["none", "gzip", "bgzf"], default="none", help="how the synthetic VCF is stored")
    ap.add_argument("--populations", type=int, default=0, help="also pass --fst_populations with this many CSV-defined populations (equal contiguous blocks of samples): "
                                                                 "one W&C sweep over all of them and one Hudson sweep per population PAIR")
    ap.add_argument("--bin", default=BIN, help="the run_vcf binary to time")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="run_vcf_scale_")
    t0 = time.perf_counter()
    geno, pos, length, vcf_bytes = write_inputs(tmp, args.sites, args.samples, args.seed, args.compress)
    gen_s = time.perf_counter() - t0

    out_csv = os.path.join(tmp, "out", "results.csv")
    cmd = [args.bin, "--vcf_folder", os.path.join(tmp, "vcfs"), "--reference", os.path.join(tmp, "ref.fa"), "--gtf",
           os.path.join(tmp, "ann.gtf"), "--config_file", os.path.join(tmp, "config.tsv"), "--output_file", out_csv, "--fst"]
    if args.populations:
        names = [f"SYN{i:05d}" for i in range(args.samples)]
        with open(os.path.join(tmp, "pops.csv"), "w") as fh:
            for k in range(args.populations):
                fh.write(f"pop{k:02d}," + ",".join(names[args.samples * k // args.populations:args.samples * (k + 1) // args.populations]) + "\n")
        cmd += ["--fst_populations", os.path.join(tmp, "pops.csv")]
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_TIMING="1"))
    cli_s = time.perf_counter() - t0
    if res.returncode != 0:
        print(res.stderr[-4000:], file=sys.stderr)
        return 1
    header, row = [l.split(",") for l in open(out_csv).read().splitlines()[:2]]
    row = dict(zip(header, row))

    import ferromic as fm

    half = args.samples // 2
    t0 = time.perf_counter()
    pops = []
    for g, (a, b) in enumerate(((0, half), (half, args.samples))):
        haps = [(s, side) for s in range(a, b) for side in (0, 1)]
        pops.append(fm.Population.from_numpy(g, geno, (pos - 1).astype(np.int64), haps, length))
    seg = [p.segregating_sites() for p in pops]
    pi = [p.nucleotide_diversity() for p in pops]
    hud = fm.hudson_fst(pops[0], pops[1])
    api_s = time.perf_counter() - t0
    n_hap = 2 * half
    theta = [fm.watterson_theta(seg[0], n_hap, length), fm.watterson_theta(seg[1], 2 * (args.samples - half), length)]

    def close(csv_value: str, x: float) -> bool:
        return csv_value == f"{x:.6f}" or abs(float(csv_value) - x) <= 1.5e-6

    checks = {
        "0_segregating_sites_filtered": row["0_segregating_sites_filtered"] == str(seg[0]),
        "1_segregating_sites_filtered": row["1_segregating_sites_filtered"] == str(seg[1]),
        "0_pi_filtered": close(row["0_pi_filtered"], pi[0]),
        "1_pi_filtered": close(row["1_pi_filtered"], pi[1]),
        "0_w_theta_filtered": close(row["0_w_theta_filtered"], theta[0]),
        "1_w_theta_filtered": close(row["1_w_theta_filtered"], theta[1]),
        "hudson_fst_hap_group_0v1": close(row["hudson_fst_hap_group_0v1"], hud.fst),
        "hudson_dxy_hap_group_0v1": close(row["hudson_dxy_hap_group_0v1"], hud.d_xy),
    }
    timing = [l for l in res.stderr.splitlines() if l.startswith("[TIMING]")]
    print(json.dumps({
        "sites": args.sites, "samples": args.samples, "haplotypes": 2 * args.samples, "vcf_bytes": vcf_bytes, "vcf_storage": args.compress,
        "csv_populations": args.populations, "population_pairs": args.populations * (args.populations - 1) // 2, "binary": os.path.relpath(args.bin, ROOT),
        "generate_s": gen_s, "run_vcf_wall_s": cli_s, "vcf_MB_per_s": vcf_bytes / cli_s / 1e6,
        "sites_per_s_end_to_end": args.sites / cli_s, "api_from_numpy_s": api_s,
        "csv_matches_python_api": checks, "all_match": all(checks.values()), "run_vcf_timing": timing,
        "csv_row": {k: row[k] for k in ("0_pi_filtered", "1_pi_filtered", "0_segregating_sites_filtered",
                                         "haplotype_overall_fst_wc", "hudson_fst_hap_group_0v1")},
    }))
    if not args.keep:
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)
    return 0 if all(checks.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
