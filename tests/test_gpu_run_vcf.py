"""run_vcf CLI parity: the C++ binary (text ingest + GPU statistics + writers) against the oracle's
literal restatement of the reference pipeline, on the reference's own end-to-end cases and on a
synthetic multi-chromosome cohort with missing calls, low GQ, indels, multi-allelic sites, masks,
allow lists, N runs in the reference, suffix-tagged config genotypes and sample exclusions."""

import gzip
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import run_vcf_ref as V

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("FERROMIC_RUN_VCF_BIN") or os.path.join(ROOT, "ferromic_amd", "bin", "run_vcf")  # the override: the sanitizer build (make asan)


def run_binary(out_csv, **kw):
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--output_file", out_csv]
    for key in ("config_file", "mask_file", "allow_file", "fst_populations"):
        if kw.get(key):
            cmd += [f"--{key}", kw[key]]
    if kw.get("chrom"):
        cmd += ["--chr", kw["chrom"]]
    if kw.get("region"):
        cmd += ["--region", kw["region"]]
    if kw.get("min_gq") is not None:
        cmd += ["--min_gq", str(kw["min_gq"])]
    if kw.get("exclude"):
        cmd += ["--exclude", ",".join(kw["exclude"])]
    if kw.get("enable_fst"):
        cmd += ["--fst"]
    if kw.get("devices"):
        cmd += ["--devices", kw["devices"]]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_PROGRESS="0", **kw.get("env", {})), timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = {}
    d = os.path.dirname(out_csv)
    out[os.path.basename(out_csv)] = open(out_csv).read()
    for name in ("per_site_diversity_output.falsta.gz", "per_site_fst_output.falsta.gz", "hudson_fst_results.tsv.gz",
                 "wc_fst_results.tsv.gz"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            out[name] = gzip.open(p, "rt").read()
    return out


def assert_tables_close(got: str, exp: str, sep: str, what: str):
    g, e = got.splitlines(), exp.splitlines()
    assert len(g) == len(e), f"{what}: {len(g)} rows vs {len(e)}"
    for i, (gl, el) in enumerate(zip(g, e)):
        gf, ef = gl.split(sep), el.split(sep)
        assert len(gf) == len(ef), f"{what} row {i}"
        for a, b in zip(gf, ef):
            if a == b:
                continue
            try:
                fa, fb = float(a), float(b)
            except ValueError:
                raise AssertionError(f"{what} row {i}: {a!r} != {b!r}\n{gl}\n{el}") from None
            assert abs(fa - fb) <= 1.5e-6, f"{what} row {i}: {a} vs {b}"  # a last-digit flip of {:.6} from sum order


def compare(got, exp):
    assert set(got) == set(exp), (sorted(got), sorted(exp))
    for name in exp:
        if name.endswith(".csv"):
            assert_tables_close(got[name], exp[name], ",", name)
        elif name.startswith("hudson") or name.startswith("wc_fst"):
            assert_tables_close(got[name], exp[name], "\t", name)
        else:
            assert got[name] == exp[name], f"{name} differs"  # per-site tracks are bit-exact -> identical text


def write_case(tmp, k):
    os.makedirs(tmp / "vcf", exist_ok=True)
    (tmp / "vcf" / "chr1.vcf").write_text(k["vcf"])
    (tmp / "reference.fa").write_text(k["fasta"])
    (tmp / "reference.fa.fai").write_text(k["fai"])
    (tmp / "annotations.gtf").write_text(k["gtf"])
    (tmp / "config.tsv").write_text(k["config"])
    return dict(vcf_folder=str(tmp / "vcf"), reference=str(tmp / "reference.fa"), gtf=str(tmp / "annotations.gtf"),
                config_file=str(tmp / "config.tsv"), enable_fst=k["enable_fst"])


def write_cli_integration_case(tmp, k):
    """The files test_variant_filtering_cli_integration writes (filter_tests.rs:82-246), from the fixture's texts and recipes, plus the
    .fai that `samtools faidx` would put next to the FASTA (the reference opens it through an indexed reader, process.rs:1917)."""
    os.makedirs(tmp / "vcfs_test", exist_ok=True)
    (tmp / "test_allow.tsv").write_text(k["allow"])
    (tmp / "test_config.tsv").write_text(k["config"])
    for name, body in k["vcfs"].items():
        (tmp / "vcfs_test" / name).write_text(k["vcf_header"] + body)
    rec = k["fasta_recipe"]
    seq = rec["unit"] * rec["repeat"]
    fasta, fai, gtf, off = "", "", "", 0
    for c in rec["chromosomes"]:
        hdr = f">{c}\n"
        fai += f"{c}\t{len(seq)}\t{off + len(hdr)}\t{len(seq)}\t{len(seq) + 1}\n"
        fasta += hdr + seq + "\n"
        off += len(hdr) + len(seq) + 1
        gtf += f'{c}\t.\tgene\t1\t1000\t.\t+\t.\tgene_id "gene_{c}"; gene_name "gene_{c}";\n'
    (tmp / "reference.fasta").write_text(fasta)
    (tmp / "reference.fasta.fai").write_text(fai)
    (tmp / "annotations.gtf").write_text(gtf)
    return dict(vcf_folder=str(tmp / "vcfs_test"), reference=str(tmp / "reference.fasta"), gtf=str(tmp / "annotations.gtf"),
                config_file=str(tmp / "test_config.tsv"), allow_file=str(tmp / "test_allow.tsv"), min_gq=k["args"]["min_gq"])


def test_reference_cli_integration_case(tmp_path, kats):
    """The reference's own CLI integration input (12-row config with _lowconf suffixes and non-pass verdicts, an allow file with a chromosome
    that has no line for one entry, three tiny VCFs with GQ < 30 rows, entries on a chromosome without a VCF): run_vcf against the oracle's
    restatement cell for cell, the FALSTA tracks byte for byte, and the reference test's own (weak) asserts."""
    k = kats["cli_integration_filtering"]
    kw = write_cli_integration_case(tmp_path, k)
    exp = V.run(output_file=str(tmp_path / "oracle" / "output_stats.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "output_stats.csv"), **kw)
    compare(got, exp)
    csv_text = got["output_stats.csv"]
    assert csv_text and any(word in csv_text for word in k["expect"]["output_contains_any"])
    # without the index the reference stops at the indexed reader (process.rs:1917); so does run_vcf, with a message that names the file
    os.remove(kw["reference"] + ".fai")
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--config_file", kw["config_file"],
           "--output_file", str(tmp_path / "gpu2" / "o.csv"), "--min_gq", "30", "--allow_file", kw["allow_file"]]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_PROGRESS="0"), timeout=300)
    assert "reference.fasta.fai" in res.stderr + res.stdout


@pytest.mark.parametrize("case", ["falsta_zero_fill", "falsta_hudson_tracks"])
def test_reference_end_to_end_cases(tmp_path, kats, case):
    kw = write_case(tmp_path, kats[case])
    exp = V.run(output_file=str(tmp_path / "oracle" / "results.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "results.csv"), **kw)
    compare(got, exp)
    if case == "falsta_hudson_tracks":
        e = kats[case]["expect"]
        lines = got["per_site_fst_output.falsta.gz"].splitlines()
        vals = [float(x) for x in lines[lines.index(e["fst_header"]) + 1].split(",")]
        assert vals == pytest.approx(e["fst"], abs=1e-6)


def write_bgzf(path, data: bytes, block=60_000):
    """BGZF as bgzip / htslib write it: independent gzip members with a 'BC' extra subfield, plus the empty EOF block."""
    import struct
    import zlib

    with open(path, "wb") as fh:
        for off in list(range(0, len(data), block)) + [None]:
            chunk = b"" if off is None else data[off:off + block]
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = comp.compress(chunk) + comp.flush()
            bsize = 12 + 6 + len(body) + 8
            fh.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            fh.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def make_cohort(tmp, seed, n_samples=14, gz=False):
    rng = random.Random(seed)
    names = [f"POP_{'ABC'[i % 3]}_HG{i:05d}" for i in range(n_samples)]
    chroms = {"1": 6000, "7": 4000, "X": 3000}
    fasta, fai, off = "", "", 0
    for c, ln in chroms.items():
        seq = "".join(rng.choice("ACGT") for _ in range(ln))
        if c == "7":
            seq = seq[:1500] + "N" * 120 + seq[1620:]
        hdr = f">chr{c}\n"
        body = "\n".join(seq[i:i + 60] for i in range(0, ln, 60)) + "\n"
        fai += f"chr{c}\t{ln}\t{off + len(hdr)}\t60\t61\n"
        fasta += hdr + body
        off += len(hdr) + len(body)
    (tmp / "ref.fa").write_text(fasta)
    (tmp / "ref.fa.fai").write_text(fai)
    (tmp / "ann.gtf").write_text('chr1\t.\tCDS\t1\t100\t.\t+\t0\tgene_id "g"; transcript_id "t";\n')
    os.makedirs(tmp / "vcfs", exist_ok=True)
    header = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    for c, ln in chroms.items():
        lines = []
        pos = 0
        while True:
            pos += rng.randint(1, 9)
            if pos > ln:
                break
            kind = rng.random()
            ref, alt = rng.choice("ACGT"), rng.choice("ACGT")
            if kind < 0.05:
                ref = "AT"            # REF indel -> discarded
            elif kind < 0.10:
                alt = "GG"            # MNP -> discarded
            elif kind < 0.18:
                alt = alt + "," + rng.choice("ACGT")  # multi-allelic
            f = rng.betavariate(0.8, 0.8)
            cells = []
            for i in range(n_samples):
                r = rng.random()
                if r < 0.03:
                    cells.append("./.:.")
                    continue
                if r < 0.04:
                    cells.append(".:.")
                    continue
                amax = 2 if "," in alt else 1
                bias = 0.25 if i % 2 else -0.2
                a = [(rng.randint(1, amax) if rng.random() < min(max(f + bias, 0.02), 0.98) else 0) for _ in range(2)]
                gq = rng.choice([99, 60, 45, 31, 30]) if rng.random() > 0.04 else rng.choice([5, 29, "."])
                sep = "|" if rng.random() > 0.1 else "/"
                if c == "X" and i % 3 == 0 and rng.random() < 0.8:
                    cells.append(f"{a[0]}:{gq}")          # haploid call (hemizygous X): ragged ploidy inside one row
                else:
                    cells.append(f"{a[0]}{sep}{a[1]}:{gq}")
            prefix = "chr" if c != "7" else ""
            lines.append(f"{prefix}{c}\t{pos}\t.\t{ref}\t{alt}\t.\tPASS\t.\tGT:GQ\t" + "\t".join(cells) + "\n")
        name = {"1": "chr1.vcf", "7": "cohort.chr7.phased.vcf", "X": "chrX.vcf"}[c]
        text = header + "".join(lines)
        if gz == "bgzf" and c == "1":
            write_bgzf(tmp / "vcfs" / "chr1.vcf.gz", text.encode(), block=rng.choice([700, 5_000, 60_000]))
        elif gz and c == "1":
            with gzip.open(tmp / "vcfs" / "chr1.vcf.gz", "wt") as fh:
                fh.write(text)
        else:
            (tmp / "vcfs" / name).write_text(text)
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"

    def row(c, s, e):
        cells = []
        for i in range(n_samples):
            g = rng.choice(["0|0", "0|1", "1|0", "1|1"])
            if rng.random() < 0.15:
                g += "_lowconf"
            if rng.random() < 0.03:
                g = "."
            cells.append(g)
        return f"chr{c}\t{s}\t{e}\t{s}\tid\tpass\tinv\t" + "\t".join(cells) + "\n"

    cfg += row("1", 100, 2500) + row("1", 2000, 5900) + row("7", 1400, 1800) + row("7", 10, 3900) + row("X", 1, 3000)
    cfg += row("7", 1510, 1610)   # >= 99 % masked by the N run -> dropped
    cfg += row("9", 1, 100)       # chromosome absent from the reference -> skipped
    (tmp / "config.tsv").write_text(cfg)
    (tmp / "mask.bed").write_text("chr1\t300\t420\n1\t4000\t4100\nchrX\t0\t50\n")
    (tmp / "allow.tsv").write_text("chr1\t1\t5000\nchr7\t1\t4000\nchrX\t100\t2900\n")
    return dict(vcf_folder=str(tmp / "vcfs"), reference=str(tmp / "ref.fa"), gtf=str(tmp / "ann.gtf"),
                config_file=str(tmp / "config.tsv")), names


@pytest.mark.parametrize("variant", ["plain", "fst", "fst_mask_allow_exclude", "gz_min_gq", "bgzf"])
def test_synthetic_cohort(tmp_path, variant):
    kw, names = make_cohort(tmp_path, seed={"plain": 11, "fst": 12, "fst_mask_allow_exclude": 13, "gz_min_gq": 14, "bgzf": 15}[variant],
                            gz="bgzf" if variant == "bgzf" else variant == "gz_min_gq")
    if variant in ("fst", "fst_mask_allow_exclude", "gz_min_gq", "bgzf"):
        kw["enable_fst"] = True
    if variant == "fst_mask_allow_exclude":
        kw.update(mask_file=str(tmp_path / "mask.bed"), allow_file=str(tmp_path / "allow.tsv"), exclude=[names[3], "HG00007"])
    if variant == "gz_min_gq":
        kw["min_gq"] = 50
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
    rows = exp["out.csv"].splitlines()
    assert len(rows) == 6  # header + 5 surviving regions
    if variant in ("fst", "fst_mask_allow_exclude"):
        assert any("NA" not in r.split(",")[25:30] for r in rows[1:])  # some region has a calculable W&C FST


def make_big_cohort(tmp, seed, n_sites, n_samples, length):
    rng = np.random.default_rng(seed)
    names = [f"S{i:04d}" for i in range(n_samples)]
    seq = "".join(rng.choice(list("ACGT"), size=length))
    hdr = ">chr1\n"
    body = "\n".join(seq[i:i + 60] for i in range(0, length, 60)) + "\n"
    (tmp / "ref.fa").write_text(hdr + body)
    (tmp / "ref.fa.fai").write_text(f"chr1\t{length}\t{len(hdr)}\t60\t61\n")
    (tmp / "ann.gtf").write_text('chr1\t.\tCDS\t1\t100\t.\t+\t0\tgene_id "g"; transcript_id "t";\n')
    os.makedirs(tmp / "vcfs", exist_ok=True)
    pos = np.sort(rng.choice(np.arange(1, length + 1), size=n_sites, replace=False))
    f = rng.beta(0.8, 0.8, size=(n_sites, 1))
    shift = np.where(np.arange(n_samples) % 2 == 0, 0.15, -0.15)[None, :, None]
    g = (rng.random((n_sites, n_samples, 2)) < np.clip(f[:, :, None] + shift, 0.01, 0.99)).astype(np.uint8)
    multi = rng.random(n_sites) < 0.05
    g[multi] *= rng.integers(1, 3, size=(int(multi.sum()), n_samples, 2), dtype=np.uint8)
    miss = rng.random((n_sites, n_samples)) < 0.01
    lowgq = rng.random((n_sites, n_samples)) < 0.0005
    bases = np.array(list("ACGT"))
    ref = bases[rng.integers(0, 4, n_sites)]
    alt = bases[rng.integers(0, 4, n_sites)]
    lines = ["##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"]
    gt_txt = np.char.add(np.char.add(g[:, :, 0].astype(str), "|"), g[:, :, 1].astype(str))
    gt_txt = np.where(lowgq, np.char.add(gt_txt, ":12"), np.char.add(gt_txt, ":60"))
    gt_txt = np.where(miss, "./.:.", gt_txt)
    for s in range(n_sites):
        a = alt[s] + (",T" if multi[s] else "")
        lines.append(f"chr1\t{pos[s]}\t.\t{ref[s]}\t{a}\t.\tPASS\t.\tGT:GQ\t" + "\t".join(gt_txt[s]) + "\n")
    (tmp / "vcfs" / "chr1.vcf").write_text("".join(lines))
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"
    r = random.Random(seed)
    for (s, e) in ((1, length), (length // 4, length // 2), (length // 2, length - 10)):
        cells = [r.choice(["0|0", "0|1", "1|0", "1|1", "0|1_lowconf"]) for _ in range(n_samples)]
        cfg += f"chr1\t{s}\t{e}\t{s}\tid\tpass\tinv\t" + "\t".join(cells) + "\n"
    (tmp / "config.tsv").write_text(cfg)
    return dict(vcf_folder=str(tmp / "vcfs"), reference=str(tmp / "ref.fa"), gtf=str(tmp / "ann.gtf"),
                config_file=str(tmp / "config.tsv"), enable_fst=True)


def test_large_single_chromosome(tmp_path):
    """SURVEY 8(d) honesty note: a real synthetic VCF at reduced scale through text -> matrix -> all outputs.
    30 000 sites x 80 haplotypes, 3 overlapping regions (the first spans the chromosome), 5 % multi-allelic
    sites, 1 % missing calls, a few low-GQ cells, W&C + Hudson tracks; every file compared with the oracle."""
    n_sites = 30_000
    kw = make_big_cohort(tmp_path, 3, n_sites, 40, n_sites * 6)
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
    row = dict(zip(V.CSV_HEADER, got["out.csv"].splitlines()[1].split(",")))
    assert int(row["0_segregating_sites"]) > 20_000 and row["haplotype_overall_fst_wc"] != "NA"


def test_csv_defined_populations(tmp_path):
    """--fst_populations: W&C + Hudson between CSV-defined populations (stats.rs:816-1078, process.rs:3301-3392)
    and the fifth output file, wc_fst_results.tsv.gz."""
    kw, names = make_cohort(tmp_path, seed=21, n_samples=18)
    pops = {"AFR": [n for n in names if "_A_" in n], "EUR": [n for n in names if "_B_" in n] + ["NOT_IN_VCF"],
            "EAS": [n for n in names if "_C_" in n]}
    text = "# population,samples...\n\n" + "".join(f"{k}, " + " , ".join(v) + "\n" for k, v in pops.items()) + "EMPTY\n"
    (tmp_path / "pops.csv").write_text(text)
    kw.update(enable_fst=True, fst_populations=str(tmp_path / "pops.csv"), exclude=[names[4]])
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
    wc = got["wc_fst_results.tsv.gz"].splitlines()
    assert wc[0].split("\t") == V.WC_TSV_HEADER
    assert sum(1 for r in wc if "\tpairwise\t" in r) == 3 * 5 and sum(1 for r in wc if "\toverall\tALL\tALL\t" in r) == 5
    hud = got["hudson_fst_results.tsv.gz"]
    assert "NamedPopulation\tAFR\tNamedPopulation\tEAS" in hud and "HaplotypeGroup\t0\tHaplotypeGroup\t1" in hud


@pytest.mark.parametrize("seed", range(100, 116))
def test_random_flag_combinations(tmp_path, seed):
    """Seeded random cohorts x random CLI flags (mask / allow / exclude / min_gq / --fst / population CSV / gz input / workers)."""
    rng = random.Random(seed)
    n = rng.randint(6, 22)
    kw, names = make_cohort(tmp_path, seed=seed, n_samples=n, gz=rng.choice([False, False, True, "bgzf"]))
    if rng.random() < 0.7:
        kw["enable_fst"] = True
    if rng.random() < 0.5:
        kw["mask_file"] = str(tmp_path / "mask.bed")
    if rng.random() < 0.5:
        kw["allow_file"] = str(tmp_path / "allow.tsv")
    if rng.random() < 0.4:
        kw["exclude"] = rng.sample(names, rng.randint(1, 2))
    if rng.random() < 0.5:
        kw["min_gq"] = rng.choice([0, 20, 31, 46, 70])
    if kw.get("enable_fst") and rng.random() < 0.5:
        k = rng.randint(2, 4)
        (tmp_path / "pops.csv").write_text("".join(f"G{g}," + ",".join(names[g::k]) + "\n" for g in range(k)))
        kw["fst_populations"] = str(tmp_path / "pops.csv")
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), devices="0,0" if rng.random() < 0.3 else None, **kw)
    compare(got, exp)


def test_many_csv_populations(tmp_path):
    """More CSV populations than the fused sweep holds in registers (8): fmh_wc_sweep_many (counting sweeps in
    batches of 8 + counts kernel) and one Hudson sweep per population pair."""
    kw, names = make_cohort(tmp_path, seed=41, n_samples=33)
    pops = {f"P{k:02d}": names[k::11] for k in range(11)}
    (tmp_path / "pops.csv").write_text("".join(f"{k}," + ",".join(v) + "\n" for k, v in pops.items()))
    kw.update(enable_fst=True, fst_populations=str(tmp_path / "pops.csv"))
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
    wc = got["wc_fst_results.tsv.gz"].splitlines()
    assert sum(1 for r in wc if "\tpairwise\t" in r) == 55 * 5 and sum(1 for r in wc if "\toverall\tALL\tALL\t" in r) == 5


def test_region_workers_match_single_worker(tmp_path):
    """--devices: config regions dealt out to one worker thread per GPU ("whole config regions").
    A one-GPU box can only alias device 0, which still exercises the dynamic work queue, the ordered emit
    and the per-device sweep lock: the files must be identical to the single-worker run and to the oracle."""
    kw, names = make_cohort(tmp_path, seed=31, n_samples=16)
    kw.update(enable_fst=True, mask_file=str(tmp_path / "mask.bed"))
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    one = run_binary(str(tmp_path / "one" / "out.csv"), **kw)
    many = run_binary(str(tmp_path / "many" / "out.csv"), devices="0,0,0", **kw)
    assert many == one
    compare(many, exp)


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_large_region_split_into_site_slabs(tmp_path, devices):
    """--devices N on ONE large region ("sub-ranges of one large region"): the region's sites are split into one
    contiguous slab per listed GPU, every statistic's regional accumulators (summaries, W&C, Hudson) are summed through the library's
    communicator, the per-site tracks are assembled in site order.  A one-GPU box lists device 0 several times, which selects the
    communicator's in-process transport (RCCL refuses two ranks on one GPU); FERROMIC_SHARD_MIN_BYTES=1 makes every region "large".
    The files must equal the single-GPU run - per-site tracks byte for byte, regional cells within the last digit of {:.6} - and the
    oracle.  5 % multi-allelic sites, 1 % missing calls, overlapping regions, CSV-defined populations on top."""
    n_sites = 12_000
    kw = make_big_cohort(tmp_path, 17, n_sites, 30, n_sites * 5)
    pops = tmp_path / "pops.csv"
    pops.write_text("".join(f"pop{k},{','.join(f'S{i:04d}' for i in range(k, 30, 3))}\n" for k in range(3)))
    kw["fst_populations"] = str(pops)
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    one = run_binary(str(tmp_path / "one" / "out.csv"), **kw)
    split = run_binary(str(tmp_path / "split" / "out.csv"), devices=devices, env={"FERROMIC_SHARD_MIN_BYTES": "1"}, **kw)
    compare(split, one)
    compare(split, exp)
    for name in ("per_site_diversity_output.falsta.gz", "per_site_fst_output.falsta.gz"):
        assert split[name] == one[name]
    # the same command without the override leaves these small regions whole: identical to the single-worker files
    whole = run_binary(str(tmp_path / "whole" / "out.csv"), devices=devices, **kw)
    assert whole == one


@pytest.mark.parametrize("nth", [0, 2, 5])
def test_slab_failure_is_an_error_not_a_hang(tmp_path, nth):
    """one slab failing before its collective left its peers blocked for ever (the in-process rendezvous and RCCL have no
    timeout, and the region holds the shard lock).  FERROMIC_INJECT_SLAB_FAILURE=<slab>:<nth> makes slab 1 throw at the nth slab-parallel
    step of the run (matrix upload, a summaries / W&C / Hudson sweep): the failing thread aborts the communicator group, the peers
    return with an error, the region is logged as DROPPED and the run ENDS - and the regions after it are computed (the communicators
    are re-created): their rows equal the clean run's."""
    n_sites = 6_000
    kw = make_big_cohort(tmp_path, 23, n_sites, 24, n_sites * 5)
    clean = run_binary(str(tmp_path / "clean" / "out.csv"), devices="0,0,0", env={"FERROMIC_SHARD_MIN_BYTES": "1"}, **kw)
    out_csv = str(tmp_path / "hurt" / "out.csv")
    os.makedirs(os.path.dirname(out_csv), exist_ok=True)
    cmd = [BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"], "--output_file", out_csv,
           "--config_file", kw["config_file"], "--devices", "0,0,0", "--workers_per_device", "1"]
    if kw.get("enable_fst"):
        cmd += ["--fst"]
    env = dict(os.environ, FERROMIC_PROGRESS="0", FERROMIC_SHARD_MIN_BYTES="1", FERROMIC_INJECT_SLAB_FAILURE=f"1:{nth}")
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)  # a hang ends here
    log = res.stdout + res.stderr
    assert "injected failure of slab 1" in log and "DROPPED" in log, log[-3000:]
    assert res.returncode == 0, log[-3000:]
    hurt_rows = open(out_csv).read().splitlines()
    clean_rows = clean["out.csv"].splitlines()
    assert hurt_rows[0] == clean_rows[0] and len(hurt_rows) == len(clean_rows) - 1  # exactly one region was dropped
    assert set(hurt_rows[1:]) <= set(clean_rows[1:])  # every other region: the very rows of the clean run


def test_single_chromosome_mode(tmp_path):
    kw, _ = make_cohort(tmp_path, seed=5)
    kw.pop("config_file")
    kw.update(chrom="1", region="200-3000", enable_fst=True)
    exp = V.run(output_file=str(tmp_path / "oracle" / "o.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "o.csv"), **kw)
    compare(got, exp)
    row = dict(zip(V.CSV_HEADER, got["o.csv"].splitlines()[1].split(",")))
    assert row["1_num_hap_no_filter"] == "0" and row["hudson_fst_hap_group_0v1"] == "NA"  # everything is group 0


def test_cli_errors(tmp_path):
    res = subprocess.run([BIN, "--vcf_folder", "x"], capture_output=True, text=True)
    assert res.returncode != 0 and "required arguments" in res.stderr
    kw, _ = make_cohort(tmp_path, seed=6)
    res = subprocess.run([BIN, "--vcf_folder", kw["vcf_folder"], "--reference", kw["reference"], "--gtf", kw["gtf"]],
                         capture_output=True, text=True)
    assert res.returncode != 0 and "Either --config_file or --chr must be specified" in res.stderr


def test_reference_full_pipeline_shape(tmp_path):
    """src/tests/full_integration_test.rs: 25 chromosomes, 30 config regions of 4 kb (five chromosomes carry two), 8 samples
    with random haplotype groups, 15 random biallelic variants per region, no filter of any kind.  The reference's own
    assertions (30 rows, pi > 0 in both groups, filtered == unfiltered, segregating sites > 0) plus equality with the oracle."""
    rng = random.Random(2024)
    samples = [f"Sample{i}" for i in range(8)]
    chroms = [str(i) for i in range(1, 26)]
    length = 100_000
    fasta, fai, off = "", "", 0
    for c in chroms:
        seq = "ACTACGTACGGATCG" * (length // 15 + 1)
        seq = seq[:length]
        hdr = f">chr{c}\n"
        body = "\n".join(seq[i:i + 100] for i in range(0, length, 100)) + "\n"
        fai += f"chr{c}\t{length}\t{off + len(hdr)}\t100\t101\n"
        fasta += hdr + body
        off += len(hdr) + len(body)
    (tmp_path / "ref.fa").write_text(fasta)
    (tmp_path / "ref.fa.fai").write_text(fai)
    (tmp_path / "ann.gtf").write_text("".join(f'chr{c}\t.\tgene\t1\t1000\t.\t+\t.\tgene_id "g{c}"; gene_name "g{c}";\n' for c in chroms))
    os.makedirs(tmp_path / "vcfs")
    usage, regions = {}, {}
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(samples) + "\n"
    for i in range(30):
        c = chroms[i % 25]
        start = 1000 + usage.get(c, 0) * 5000   # 0-based half-open [start, start + 4000)
        usage[c] = usage.get(c, 0) + 1
        regions.setdefault(c, []).append((start, start + 4000))
        cells = [f"{rng.randint(0, 1)}|{rng.randint(0, 1)}" for _ in samples]
        cfg += f"chr{c}\t{start + 1}\t{start + 4000}\t{start}\tinv{i}\tpass\tinv\t" + "\t".join(cells) + "\n"
    (tmp_path / "config.tsv").write_text(cfg)
    for c, regs in regions.items():
        rows = []
        for a, b in regs:
            for _ in range(15):
                pos1 = rng.randrange(a, b) + 1
                gts = "\t".join(f"{rng.randint(0, 1)}|{rng.randint(0, 1)}:60" for _ in samples)
                rows.append((pos1, f"chr{c}\t{pos1}\t.\tA\tT\t.\tPASS\t.\tGT:GQ\t{gts}\n"))
        rows.sort(key=lambda r: r[0])
        (tmp_path / "vcfs" / f"chr{c}.vcf").write_text(
            "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n" + "".join(r[1] for r in rows))
    kw = dict(vcf_folder=str(tmp_path / "vcfs"), reference=str(tmp_path / "ref.fa"), gtf=str(tmp_path / "ann.gtf"),
              config_file=str(tmp_path / "config.tsv"))
    exp = V.run(output_file=str(tmp_path / "oracle" / "output.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "output.csv"), **kw)
    compare(got, exp)
    lines = got["output.csv"].splitlines()
    assert len(lines) - 1 == 30
    for line in lines[1:]:
        row = dict(zip(V.CSV_HEADER, line.split(",")))
        assert float(row["0_pi"]) > 0.0 and float(row["1_pi"]) > 0.0
        assert abs(float(row["0_pi"]) - float(row["0_pi_filtered"])) < 1e-9 and abs(float(row["1_pi"]) - float(row["1_pi_filtered"])) < 1e-9
        assert int(row["0_segregating_sites"]) > 0


def test_group1_example_through_the_cli(tmp_path, kats):
    """src/tests/stats_tests.rs setup_group1_test / test_group1_*: three samples, haplotype group 1 = {Sample1 R, Sample2 L,
    Sample3 R} -> 3 haplotypes, 2 segregating sites, theta_W = 2 / H_2 / L."""
    c = kats["process_variants"]["cases"][0]
    length = 4000
    seq = ("ACGT" * (length // 4 + 1))[:length]
    hdr = ">chr1\n"
    (tmp_path / "ref.fa").write_text(hdr + "\n".join(seq[i:i + 60] for i in range(0, length, 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text(f"chr1\t{length}\t{len(hdr)}\t60\t61\n")
    (tmp_path / "ann.gtf").write_text('chr1\t.\tCDS\t1\t100\t.\t+\t0\tgene_id "g"; transcript_id "t";\n')
    os.makedirs(tmp_path / "vcfs")
    names = c["sample_names"]
    vcf = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    for v in c["variants"]:
        vcf += f"chr1\t{v['pos'] + 1}\t.\tA\tT\t.\tPASS\t.\tGT:GQ\t" + "\t".join(f"{g[0]}|{g[1]}:40" for g in v["g"]) + "\n"
    (tmp_path / "vcfs" / "chr1.vcf").write_text(vcf)
    a, b = c["interval"]   # 0-based half-open -> 1-based inclusive config columns
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"
    cfg += f"chr1\t{a + 1}\t{b}\t1\tid\tpass\tinv\t" + "\t".join("|".join(str(x) for x in c["sample_filter"][n]) for n in names) + "\n"
    (tmp_path / "config.tsv").write_text(cfg)
    kw = dict(vcf_folder=str(tmp_path / "vcfs"), reference=str(tmp_path / "ref.fa"), gtf=str(tmp_path / "ann.gtf"),
              config_file=str(tmp_path / "config.tsv"))
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
    row = dict(zip(V.CSV_HEADER, got["out.csv"].splitlines()[1].split(",")))
    assert row["1_num_hap_no_filter"] == "3" and row["1_segregating_sites"] == "2"
    assert abs(float(row["1_w_theta"]) - 2.0 / 1.5 / (b - a)) < 1e-6


@pytest.mark.parametrize("seed", range(int(os.environ.get("FERROMIC_FUZZ_PIPELINE_CASES", "10"))))
def test_adversarial_text_full_pipeline(tmp_path, seed):
    """The adversarial VCF text of tests/test_run_vcf_ingest_fuzz_cpu.py (ragged ploidy, triploid and haploid calls, alleles
    up to 255, half-missing calls, duplicate and unsorted positions, broken lines) through the WHOLE binary - matrices,
    GPU sweeps, all writers - against the oracle pipeline."""
    from tests.test_run_vcf_ingest_fuzz_cpu import build

    kw = build(tmp_path, 7000 + seed)
    kw.update(enable_fst=True, min_gq=[30, 0, 46][seed % 3])
    if seed % 2:
        kw.update(mask_file=str(tmp_path / "mask.bed"), allow_file=str(tmp_path / "allow.tsv"))
    exp = V.run(output_file=str(tmp_path / "oracle" / "out.csv"), **kw)
    got = run_binary(str(tmp_path / "gpu" / "out.csv"), **kw)
    compare(got, exp)
