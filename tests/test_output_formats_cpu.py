"""The run_vcf output surface against the reference's committed exemplars:
tests/golden/format_exemplars.json holds the header lines, a few rows and the FALSTA record shapes of
data/output.csv, data/FST_data.tsv and data/per_site_diversity_output.falsta.gz (transcribed by
tools/make_format_exemplars.py; the inputs behind those files are not in the reference tree, so they pin format, not
numbers).  `run_vcf --print_formats` emits the binary's header lines and the records its writers produce for a tiny
made-up region - no GPU, no inputs."""

import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("FERROMIC_RUN_VCF_BIN") or os.path.join(ROOT, "ferromic_amd", "bin", "run_vcf")  # the override: the sanitizer build (make asan)


@pytest.fixture(scope="module")
def exemplars():
    with open(os.path.join(ROOT, "tests", "golden", "format_exemplars.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def printed():
    binary = os.path.join(ROOT, "ferromic_amd", "bin", "run_vcf")
    if not os.path.exists(binary):
        import __graft_entry__ as ge

        ge.build()
    res = subprocess.run([binary, "--print_formats"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    files = {}
    for line in res.stdout.splitlines():
        if "\t" in line and line.split("\t", 1)[0].endswith((".csv", ".tsv", ".falsta")):
            name, rest = line.split("\t", 1)
            files.setdefault(name, []).append(rest)
        else:  # the token line of the FALSTA record just opened
            files[name].append(line)
    return files


def shape(tok):
    if re.fullmatch(r"-?\d+\.\d{6}", tok):
        return "d.dddddd"
    if re.fullmatch(r"-?\d+", tok):
        return "0" if tok == "0" else "int"
    return tok


def test_csv_and_tsv_headers_equal_the_reference_files(exemplars, printed):
    assert exemplars["output_csv"]["columns"] == 34
    assert printed["output.csv"] == [exemplars["output_csv"]["header"]]
    assert exemplars["hudson_tsv"]["columns"] == 12
    assert printed["hudson_fst_results.tsv"] == [exemplars["hudson_tsv"]["header"]]
    assert printed["wc_fst_results.tsv"][0].split("\t")[:3] == ["chr", "region_start_1based", "region_end_1based"]


def test_oracle_headers_equal_the_reference_files(exemplars):
    from oracle import run_vcf_ref as O

    assert ",".join(O.CSV_HEADER) == exemplars["output_csv"]["header"]
    assert "\t".join(O.HUDSON_TSV_HEADER) == exemplars["hudson_tsv"]["header"]
    # the cell formatters produce exactly the token shapes the reference files hold: {:.6}, NA, NaN, integers
    assert O.fmt6(0.0013634) == "0.001363" and O.fmt6(float("nan")) == "NaN" and O.format_optional_float(None) == "NA"
    for row in exemplars["output_csv"]["rows"]:
        cells = row.split(",")
        assert len(cells) == 34
        for c, cell in enumerate(cells):
            assert shape(cell) in exemplars["output_csv"]["cell_shapes_by_column"][c]
    floats = {9, 10, 11, 12, 15, 16, 17, 18}
    for c, shapes in enumerate(exemplars["output_csv"]["cell_shapes_by_column"]):
        if c in floats:
            assert set(shapes) <= {"d.dddddd", "NaN"}
    for c in (7, 8, 9, 10):
        assert set(exemplars["hudson_tsv"]["cell_shapes_by_column"][c]) == {"d.dddddd"}
    assert set(exemplars["hudson_tsv"]["cell_shapes_by_column"][11]) <= {"d.dddddd", "NA"}


def test_falsta_records_have_the_reference_shape(exemplars, printed):
    ref = exemplars["diversity_falsta"]
    assert set(ref["token_shapes"]) == {"0", "NA", "d.dddddd"}
    # record order inside one (region, group) of the reference file, and one dense line per record
    assert [r["track"] for r in ref["records"][:4]] == ["unfiltered_pi", "unfiltered_theta", "filtered_pi", "filtered_theta"]
    for r in ref["records"]:
        assert r["tokens"] == r["span"]
    recs = printed["per_site_diversity_output.falsta"]
    heads, lines = recs[0::2], recs[1::2]
    assert [re.fullmatch(r">(\w+?)_chr_1_start_5_end_12_group_(\d)", h).groups() for h in heads] == \
           [(t, g) for g in "01" for t in ("unfiltered_pi", "unfiltered_theta", "filtered_pi", "filtered_theta")]
    for h, line in zip(heads, lines):
        pattern = re.sub(r"\d+", r"\\d+", re.escape(ref["records"][0]["header"]).replace("unfiltered_pi", r"\w+"))
        assert re.fullmatch(pattern, h), (pattern, h)
        toks = line.split(",")
        assert len(toks) == 12 - 5 + 1 and {shape(t) for t in toks} <= set(ref["token_shapes"])
    assert lines[0] == "0,0.289855,0,0,NA,0,0,0"  # default 0, masked site NA, a true zero stays 0 (process.rs:3786-3792)
    fst = printed["per_site_fst_output.falsta"]
    fheads, flines = fst[0::2], fst[1::2]
    assert fheads[0] == ">haplotype_overall_fst_summary_chr_1_start_5_end_12" and fheads[6] == ">hudson_pairwise_fst_hap_0v1_chr_1_start_5_end_12"
    assert len(fheads) == 9
    for line in flines:
        toks = line.split(",")
        assert len(toks) == 8 and {shape(t) for t in toks} <= {"0", "NA", "d.dddddd", "Infinity", "-Infinity"}
    assert flines[6] == "NA,1.000000,-0.500000,NA,NA,NA,NA,NA"  # FST tracks default to NA (process.rs:3842-3856)


def test_bench_tracks_diagnostic_runs_without_a_gpu():
    """`run_vcf --bench_tracks`: formats and deflates the 17 tracks of a made-up 15-kb region 500 times on one thread (what the writers cost per
    small region) - no inputs, no GPU.  Checked here: it runs, reports 17 members and a sane size."""
    import re
    import subprocess

    from tests.test_gpu_run_vcf import BIN

    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    res = subprocess.run([BIN, "--bench_tracks"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-1000:]
    m = re.search(r"tracks of one region: ([0-9.]+) ms, (\d+) members, (\d+) bytes", res.stdout)
    assert m and int(m.group(2)) == 17 and 1000 < int(m.group(3)) < 200000 and float(m.group(1)) < 50.0, res.stdout


def test_writer_self_checks():
    """`run_vcf --check_writers N` (no GPU): (1) the writers' `{:.6}` formatter (exact integer arithmetic on the binary value, no printf) against
    printf's %.6f on pseudo-random doubles of every magnitude, the exact ties of the sixth decimal (k / 128) and their one-ulp neighbours, values
    next to a carry, zeros, subnormals, NaN and infinities (20 M values were run once: none differ); (2) the slicing-by-8 CRC-32 against zlib's on
    random buffers; (3) the run-aware gzip writer (runs of the default token as back references, table-driven CRC steps, both Huffman code sets,
    value tokens as back references to their last occurrence) against the text writer on 600 random tracks - sorted and unsorted records, both
    default tokens, gaps at the match-length edges, empty lines, 700 000-position lines - each member inflated by zlib, which also verifies the
    trailer's CRC-32 and length."""
    from tests.test_gpu_run_vcf import BIN

    if not os.path.exists(BIN):
        pytest.skip("run_vcf not built")
    res = subprocess.run([BIN, "--check_writers", "300000"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count(" 0 differ") == 3, res.stdout


def test_run_aware_gzip_writer_round_trips_through_zlib_and_gzip(tmp_path):
    """The hand-written DEFLATE writer of the FALSTA tracks (deflate_runs.cpp: fixed and tuned Huffman codes, runs as back references, values seen
    before as back references, CRC-32 folded per run) on the writers' adversarial tracks - both default tokens, run lengths around 258 and its
    multiples, records at the first and last position, unsorted records, empty and 700 000-position regions.  Every member must inflate to
    the track's text with zlib (which also checks the trailer's CRC-32 and length) AND with Python's gzip module, and so must the member the
    zlib level-1 writer makes of the same text."""
    import gzip
    import zlib

    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    out = tmp_path / "cases"
    res = subprocess.run([BIN, "--dump_writer_cases", str(out), "60"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    texts = sorted(p for p in os.listdir(out) if p.endswith(".txt"))
    assert len(texts) == 60
    sizes, run_bytes, zlib_bytes = [], 0, 0
    for name in texts:
        base = str(out / name[:-4])
        text = open(base + ".txt", "rb").read()
        for kind in ("runs", "zlib"):
            member = open(f"{base}.{kind}.gz", "rb").read()
            d = zlib.decompressobj(31)  # gzip framing: header, CRC-32 and ISIZE are verified
            assert d.decompress(member) + d.flush() == text and d.eof and d.unused_data == b"", (name, kind, "zlib")
            assert gzip.decompress(member) == text, (name, kind, "gzip")
        sizes.append(len(text))
        run_bytes += os.path.getsize(base + ".runs.gz")
        zlib_bytes += os.path.getsize(base + ".zlib.gz")
    assert max(sizes) > 300_000 and min(sizes) < 32  # tracks of several hundred thousand positions and an empty region are among the cases
    assert run_bytes < 3 * zlib_bytes  # the run-aware members stay in the size class of zlib's (they are chosen for sparse tracks only)
