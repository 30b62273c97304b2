"""Adversarial VCF text through the run_vcf binary's ingest stage (`--ingest_only`, no GPU) and through the oracle's
restatement of process_variant / process_vcf (process.rs:4092-4768): odd FORMAT layouts, GQ spellings, ploidies,
half-missing calls, alleles beyond u8, signed / padded / zero positions, foreign chromosomes, short lines, CR-LF, indels,
symbolic ALTs, duplicate and unsorted positions.  The digest covers every (position, flags, stride, genotype bytes)."""

import os
import random
import re
import subprocess

import pytest

from tests.test_gpu_run_vcf import BIN
from tests.test_run_vcf_ingest_cpu import oracle_digests

CASES = int(os.environ.get("FERROMIC_FUZZ_INGEST_CASES", "12"))


def weird_cell(rng, n_alts):
    r = rng.random()
    a = lambda: str(rng.randint(0, n_alts))  # noqa: E731
    gq = rng.choice(["99", "60", "31", "30", "29", "5", "0", ".", "", "30.5", "70000", "+40", " 45", "45 ", "abc", "-3", "065"])
    if r < 0.45:
        return f"{a()}{rng.choice('|/')}{a()}:{gq}"
    if r < 0.52:
        return rng.choice(["./.", ".|.", ".", "./.:.", ".:.", ".|.:99", "./.:5"])
    if r < 0.58:
        return f"{a()}:{gq}"                                   # haploid
    if r < 0.62:
        return f"{a()}|{a()}|{a()}:{gq}"                       # triploid
    if r < 0.66:
        return rng.choice([f"{a()}|.:{gq}", f".|{a()}:{gq}", f"{a()}/.:{gq}"])   # half missing -> whole call None
    if r < 0.70:
        return rng.choice([f"256|0:{gq}", f"0|300:{gq}", f"-1|0:{gq}", f"+1|0:{gq}", f"1|+0:{gq}", f"0| 1:{gq}", f"0|1 :{gq}", f"00|01:{gq}", f"255|254:{gq}"])
    if r < 0.74:
        return f"{a()}|{a()}"                                  # no GQ part in the cell
    if r < 0.78:
        return f"{a()}|{a()}:{gq}:12,3:extra"
    if r < 0.80:
        return ""
    return f"{a()}|{a()}:{gq}"


def near_clean_cell(rng, n_alts, dirt):
    """The cell the binary's whole-line fast path takes ("a|b:GQ", one-digit alleles), with a small chance of one of its border cases."""
    a = lambda: str(rng.randint(0, min(n_alts, 9)))  # noqa: E731
    if rng.random() >= dirt:
        return f"{a()}{rng.choice('||||/')}{a()}:{rng.choice(['99', '60', '31', '30', '29', '5', '0', '100', '65535'])}"
    return rng.choice([f"{a()}|{a()}:65536", f"{a()}|{a()}:100000", f"{a()}|{a()}:0000031", f"{a()}|{a()}:", f"{a()}|{a()}:.", f"{a()}|{a()}:3x",
                       f"{a()}|{a()}:40:7,1", f"{a()}|{a()}:12:", f"10|{a()}:50", f"{a()}|10:50", f"{a()}:50", f"{a()}|{a()}", f"{a()}|{a()}|{a()}:50",
                       f"{a()}|{a()};50", f".|.:50", f"{a()}|{a()}:50 ", f" {a()}|{a()}:50", ""])


def weird_line(rng, chrom, pos, n_samples, clean=0.0):
    if rng.random() < clean:
        fmt = rng.choice(["GT:GQ"] * 8 + ["GT:GQ:AD", "GT:GQ:PL:DP"])
        dirt = rng.choice([0.0, 0.0, 0.0, 0.02, 0.2])
        cells = [near_clean_cell(rng, 3, dirt) for _ in range(n_samples)]
        if fmt != "GT:GQ":
            cells = [c + ":1,2" if rng.random() < 0.9 else c for c in cells]
        fields = [chrom, str(pos), ".", rng.choice("ACGT"), rng.choice(["A", "C", "A,C", "A,C,G", "GG"]), ".", "PASS", ".", fmt] + cells
        r = rng.random()
        if r < 0.03:
            fields = fields[:9 + rng.randint(0, n_samples - 1)]
        elif r < 0.06:
            fields.append("0|1:50")
        return "\t".join(fields) + ("\r\n" if rng.random() < 0.03 else "\n")
    fmt = rng.choice(["GT:GQ"] * 6 + ["GT:AD:GQ", "GT:GQ:PL", "GQ:GT", "GT", "GT:DP", "GT:gq", "GT:GQ:GQ"])
    ref = rng.choice(["A", "C", "G", "T", "a", "N", "AT", "", "*"])
    alt = rng.choice(["A", "C", "G", "T", "t", "A,C", "A,C,G", "GG", "A,GG", "<DEL>", "*", ".", "", "A,", ",A"])
    n_alts = max(1, alt.count(",") + 1)
    cells = []
    for _ in range(n_samples):
        c = weird_cell(rng, n_alts)
        if fmt == "GT:AD:GQ" and c.count(":") == 1:
            g, q = c.split(":")
            c = f"{g}:7,2:{q}"
        elif fmt == "GQ:GT" and c.count(":") == 1:
            g, q = c.split(":")
            c = f"{q}:{g}"
        cells.append(c)
    chrom_txt = rng.choice([chrom] * 8 + ["chr" + chrom, "Chr" + chrom, "CHR" + chrom, " " + chrom, chrom + " ", "chr2", "2", "", "chrchr" + chrom])
    pos_txt = rng.choice([str(pos)] * 10 + ["0", "-4", "+" + str(pos), " " + str(pos), str(pos) + " ", "1_0", "abc", "", "99999999999999999999", "007"])
    fields = [chrom_txt, pos_txt, ".", ref, alt, ".", "PASS", ".", fmt] + cells
    r = rng.random()
    if r < 0.02:
        fields = fields[:rng.randint(1, 8)]                    # fewer than the 9 fixed fields
    elif r < 0.05:
        fields = fields[:9 + rng.randint(0, n_samples - 1)]    # genotype columns missing
    elif r < 0.07:
        fields.append("0|1:50")                                # one column too many
    eol = "\r\n" if rng.random() < 0.05 else "\n"
    return "\t".join(fields) + eol


def build(tmp, seed, clean=0.0, max_samples=9):
    rng = random.Random(seed)
    n_samples = rng.randint(1, max_samples)
    names = [f"S{i:03d}" for i in range(n_samples)]
    length = 3000
    seq = "".join(rng.choice("ACGT") for _ in range(length))
    (tmp / "ref.fa").write_text(">chr1\n" + "\n".join(seq[i:i + 60] for i in range(0, length, 60)) + "\n")
    (tmp / "ref.fa.fai").write_text(f"chr1\t{length}\t6\t60\t61\n")
    (tmp / "ann.gtf").write_text('chr1\t.\tCDS\t1\t100\t.\t+\t0\tgene_id "g"; transcript_id "t";\n')
    os.makedirs(tmp / "vcfs", exist_ok=True)
    text = "##fileformat=VCFv4.2\n##contig=<ID=chr1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    pos = 0
    for _ in range(rng.randint(150, 400)):
        step = rng.choice([0, 1, 1, 2, 3, 5, 8, -2])          # duplicates and the odd step backwards
        pos = max(pos + step, 1)
        text += weird_line(rng, "1", pos, n_samples, clean)
    if rng.random() < 0.5:
        text = text.rstrip("\n")                               # no newline at the end of the file
    (tmp / "vcfs" / "chr1.vcf").write_text(text, newline="")
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"
    cfg += "chr1\t1\t2500\t1\tid\tpass\tinv\t" + "\t".join(rng.choice(["0|0", "0|1", "1|0", "1|1"]) for _ in names) + "\n"
    (tmp / "config.tsv").write_text(cfg)
    (tmp / "mask.bed").write_text("chr1\t100\t160\n1\t900\t905\n")
    (tmp / "allow.tsv").write_text("chr1\t1\t2000\n")
    return dict(vcf_folder=str(tmp / "vcfs"), reference=str(tmp / "ref.fa"), gtf=str(tmp / "ann.gtf"), config_file=str(tmp / "config.tsv"))


def run_and_compare(tmp_path, seed, kw, extra=()):
    min_gq = [30, 31, 0, 46][seed % 4]
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--config_file", kw["config_file"],
           "--output_file", str(tmp_path / "out" / "o.csv"), "--mask_file", str(tmp_path / "mask.bed"), "--allow_file", str(tmp_path / "allow.tsv"),
           "--min_gq", str(min_gq), "--ingest_only", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_THREADS=str(1 + seed % 3)), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = {m.group(1): (int(m.group(2)), int(m.group(3)), m.group(4))
           for m in re.finditer(r"\[INGEST\] chr (\S+): (\d+) variants x (\d+) samples digest ([0-9a-f]{16})", res.stdout)}
    exp = oracle_digests(kw, min_gq=min_gq, mask_file=str(tmp_path / "mask.bed"), allow_file=str(tmp_path / "allow.tsv"))
    assert got == exp
    return got


@pytest.mark.parametrize("seed", range(CASES))
def test_adversarial_vcf_text(tmp_path, seed):
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    run_and_compare(tmp_path, seed, build(tmp_path, 5000 + seed))


@pytest.mark.parametrize("seed", range(max(CASES, 60)))
def test_mostly_clean_lines(tmp_path, seed):
    """Lines made of the plain "a|b:GQ" cells the whole-line fast path of process_variant takes, with its border cases mixed in:
    GQ at and beyond 65535, empty / dotted / padded GQ, extra FORMAT parts, two-digit alleles, haploid and triploid cells, short and long lines,
    CR-LF, a file without a final newline."""
    if not os.path.exists(BIN):
        pytest.skip("run_vcf binary not built")
    got = run_and_compare(tmp_path, seed, build(tmp_path, 9000 + seed, clean=0.85, max_samples=40))
    assert got and all(v[0] > 0 for v in got.values())
