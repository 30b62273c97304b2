"""The text -> Variant stage of the run_vcf binary without a GPU (`--ingest_only` computes no statistic): config / BED /
FASTA(.fai) / VCF parsing, flags, sorting and the plain, gzip and BGZF readers, pinned to the oracle's parse by a digest
over every (position, flags, stride, genotype bytes)."""

import os
import re
import subprocess

import pytest

from oracle import run_vcf_ref as V
from tests.test_gpu_run_vcf import BIN, make_cohort, write_cli_integration_case


def oracle_digests(kw, min_gq=30, mask_file=None, allow_file=None, exclude=()):
    entries = V.parse_config_file(kw["config_file"])
    mask = V.parse_regions_file(mask_file) if mask_file else None
    allow = V.parse_regions_file(allow_file) if allow_file else None
    by_chr = {}
    for e in entries:
        by_chr.setdefault(e.seqname, []).append(e)
    out = {}
    for chrom in sorted(by_chr):
        try:
            seq = V.read_reference_sequence(kw["reference"], chrom)
        except Exception:
            continue
        final_mask = {k: list(v) for k, v in (mask or {}).items()}
        final_mask.setdefault(chrom, []).extend(V.find_n_regions(seq))
        hulls = [(max(e.interval[0] - 3_000_000, 0), min(e.interval[1] + 3_000_000, len(seq))) for e in by_chr[chrom]]
        try:
            vcf_path = V.find_vcf_file(kw["vcf_folder"], chrom)
        except Exception:
            continue  # no VCF for this chromosome: its entries are skipped (process.rs:2001-2010)
        variants, flags, names = V.process_vcf(vcf_path, chrom, V.merge_intervals(hulls), min_gq, final_mask, allow, set(exclude))
        h = 1469598103934665603
        for v, fl in zip(variants, flags):
            for b in list(int(v.position).to_bytes(8, "little", signed=True)) + [fl, v.genotypes.stride] + list(v.genotypes.data):
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        out[chrom] = (len(variants), len(names), f"{h:016x}")
    return out


@pytest.mark.parametrize("route", ["mmap", "stream", "gzip"])
@pytest.mark.parametrize("block,head", [("1000", "4096"), ("333", "40"), ("64", "0"), ("5000", "100")])
def test_ingest_across_block_borders(tmp_path, block, head, route):
    """A plain-text body is parsed in line-aligned windows of a mapping of the file; a compressed one (and a plain one
    under FERROMIC_NO_MMAP) is streamed in blocks whose unfinished last line is carried into the headroom in front of the
    next block (or, when it is longer than the headroom, through a spill buffer).  Tiny blocks put every line across a
    border on every route."""
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    kw, names = make_cohort(tmp_path, seed=93, n_samples=11, gz=route == "gzip")
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--config_file", kw["config_file"],
           "--output_file", str(tmp_path / "out" / "o.csv"), "--ingest_only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_THREADS="3", FERROMIC_INGEST_BLOCK=block, FERROMIC_INGEST_HEAD=head,
                                  **({"FERROMIC_NO_MMAP": "1"} if route == "stream" else {})))
    assert res.returncode == 0, res.stderr[-2000:]
    got = {m.group(1): (int(m.group(2)), int(m.group(3)), m.group(4))
           for m in re.finditer(r"\[INGEST\] chr (\S+): (\d+) variants x (\d+) samples digest ([0-9a-f]{16})", res.stdout)}
    assert got == oracle_digests(kw) and len(got) == 3


@pytest.mark.parametrize("storage", [False, True, "bgzf"])
def test_ingest_matches_oracle_parse(tmp_path, storage):
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    kw, names = make_cohort(tmp_path, seed=91, n_samples=17, gz=storage)
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--config_file", kw["config_file"],
           "--output_file", str(tmp_path / "out" / "o.csv"), "--mask_file", str(tmp_path / "mask.bed"), "--allow_file", str(tmp_path / "allow.tsv"),
           "--min_gq", "31", "--ingest_only"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_THREADS="3"), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = {m.group(1): (int(m.group(2)), int(m.group(3)), m.group(4))
           for m in re.finditer(r"\[INGEST\] chr (\S+): (\d+) variants x (\d+) samples digest ([0-9a-f]{16})", res.stdout)}
    exp = oracle_digests(kw, min_gq=31, mask_file=str(tmp_path / "mask.bed"), allow_file=str(tmp_path / "allow.tsv"))
    assert got == exp and len(got) == 3


def test_ingest_of_the_reference_cli_integration_input(tmp_path, kats):
    """The input of the reference's CLI integration test (filter_tests.rs:82-246) through the ingest stage: chromosomes without a VCF (chr1)
    are skipped, the others give the oracle's variants, flags (GQ < 30, outside the allow list) and sample lists."""
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    kw = write_cli_integration_case(tmp_path, kats["cli_integration_filtering"])
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--config_file", kw["config_file"],
           "--output_file", str(tmp_path / "out" / "o.csv"), "--allow_file", kw["allow_file"], "--min_gq", "30", "--ingest_only"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_THREADS="2"), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = {m.group(1): (int(m.group(2)), int(m.group(3)), m.group(4))
           for m in re.finditer(r"\[INGEST\] chr (\S+): (\d+) variants x (\d+) samples digest ([0-9a-f]{16})", res.stdout)}
    exp = oracle_digests(kw, min_gq=30, allow_file=kw["allow_file"])
    assert got == exp and set(got) == {"17", "22", "3"}, (got, exp)


def test_cli_region_argument_matches_reference_rules(tmp_path, kats):
    """--region goes through parse_region (parse.rs:241-261): the reference's valid/invalid cases at the CLI (no GPU needed)."""
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    kw, _ = make_cohort(tmp_path, seed=92, n_samples=6)
    base = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--output_file", str(tmp_path / "o" / "o.csv"),
            "--chr", "1", "--ingest_only"]
    env = dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_THREADS="2")
    for text in kats["parse_region"]["invalid"]:
        res = subprocess.run(base + [f"--region={text}"], capture_output=True, text=True, env=env, timeout=120)
        assert res.returncode != 0 and "InvalidRegion" in res.stderr, (text, res.stderr[-300:])
    res = subprocess.run(base + ["--region", "1-1000"], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0 and "[INGEST] chr 1:" in res.stdout, res.stderr[-500:]
