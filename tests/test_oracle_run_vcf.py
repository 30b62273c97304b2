"""Pins the run_vcf half of the oracle (oracle/run_vcf_ref.py) against the reference's end-to-end
tests (FALSTA zero fill, Hudson FALSTA tracks) and its output format exemplars."""

import os

import pytest

from oracle import run_vcf_ref as V


def write_case(tmp, k):
    os.makedirs(tmp / "vcf", exist_ok=True)
    (tmp / "vcf" / "chr1.vcf").write_text(k["vcf"])
    (tmp / "reference.fa").write_text(k["fasta"])
    (tmp / "reference.fa.fai").write_text(k["fai"])
    (tmp / "annotations.gtf").write_text(k["gtf"])
    (tmp / "config.tsv").write_text(k["config"])
    return dict(vcf_folder=str(tmp / "vcf"), reference=str(tmp / "reference.fa"), gtf=str(tmp / "annotations.gtf"),
                config_file=str(tmp / "config.tsv"), output_file=str(tmp / "out" / "results.csv"), enable_fst=k["enable_fst"])


def track(text, header):
    lines = text.splitlines()
    return lines[lines.index(header) + 1].split(",")


def test_falsta_zero_fill(tmp_path, kats):
    k = kats["falsta_zero_fill"]
    out = V.run(**write_case(tmp_path, k))
    e = k["expect"]
    for hdr in (e["pi_header"], e["theta_header"]):
        vals = track(out["per_site_diversity_output.falsta.gz"], hdr)
        assert len(vals) == e["length"]
        assert all(vals[i] == "0" for i in e["zero_positions"])
        assert all(vals[i] != "0" for i in e["nonzero_positions"])
    csv = out["results.csv"].splitlines()
    assert csv[0].split(",") == V.CSV_HEADER and len(V.CSV_HEADER) == 34
    row = dict(zip(V.CSV_HEADER, csv[1].split(",")))
    assert row["chr"] == "1" and row["region_start"] == "1" and row["region_end"] == "5"
    assert row["0_segregating_sites"] == "1" and row["0_num_hap_no_filter"] == "2"
    assert row["0_w_theta"] == "0.200000" and row["0_pi"] == "0.200000"   # S=1, n=2, L=5
    assert row["hudson_fst_hap_group_0v1"] == "NA"
    assert os.path.exists(tmp_path / "out" / "per_site_diversity_output.falsta.gz")


def test_falsta_hudson_tracks(tmp_path, kats):
    k = kats["falsta_hudson_tracks"]
    out = V.run(**write_case(tmp_path, k))
    e = k["expect"]
    text = out["per_site_fst_output.falsta.gz"]
    for hdr, key in ((e["fst_header"], "fst"), (e["num_header"], "num"), (e["den_header"], "den")):
        vals = [float(x) for x in track(text, hdr)]
        assert vals == pytest.approx(e[key], abs=e["abs_tol"])
    hud = out["hudson_fst_results.tsv.gz"].splitlines()
    assert hud[0].split("\t") == V.HUDSON_TSV_HEADER
    f = hud[1].split("\t")
    assert f[:7] == ["1", "0", "2", "HaplotypeGroup", "0", "HaplotypeGroup", "1"]
    assert ">haplotype_overall_fst_summary_chr_1_start_1_end_3" in text


def test_format_helpers_match_rust():
    assert V.fmt6(0.1234565) == "0.123457" or V.fmt6(0.1234565) == "0.123456"  # exact-decimal rounding
    assert V.fmt6(float("nan")) == "NaN" and V.fmt6(float("inf")) == "inf" and V.fmt6(-float("inf")) == "-inf"
    assert V.format_optional_float(None) == "NA" and V.format_optional_float(float("nan")) == "NA"
    assert V._falsta_value_fst(float("inf")) == "Infinity" and V._falsta_value_fst(-0.0) == "0"
    assert V.fmt6(-0.0000001) == "-0.000000"


def test_find_vcf_file_scoring(tmp_path):
    for name in ("chr22.test.vcf", "chr2.other.vcf.gz", "chr22.vcf.gz.tbi", "x_chr22_y.vcf"):
        (tmp_path / name).write_text("")
    assert os.path.basename(V.find_vcf_file(str(tmp_path), "22")) == "chr22.test.vcf"
    assert os.path.basename(V.find_vcf_file(str(tmp_path), "2")) == "chr2.other.vcf.gz"
    with pytest.raises(Exception):
        V.find_vcf_file(str(tmp_path), "7")


def test_config_parsing_rules(tmp_path):
    cfg = tmp_path / "c.tsv"
    cfg.write_text("seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\tA\tB\tC\n"
                   "chr3\t200100\t200900\t1\tid\tpass\tinv\t0|0_lowconf\t0|0_lowconf\t0|1\n"
                   "chr1\t5\t9\t1\tid\tpass\tinv\t2|0\tx\t.\n")
    entries = V.parse_config_file(str(cfg))
    assert len(entries) == 1  # second row has no valid unfiltered genotype -> skipped (parse.rs:206-215)
    e = entries[0]
    assert e.seqname == "3" and e.interval == (200099, 200900)
    assert e.samples_unfiltered == {"A": (0, 0), "B": (0, 0), "C": (0, 1)}
    assert e.samples_filtered == {"C": (0, 1)}  # suffixed genotypes are not exact matches


def test_interval_conversions(kats):
    """src/tests/interval_tests.rs: the 0-/1-based inclusive -> 0-based half-open conversions every coordinate goes through."""
    k = kats["interval_conversions"]
    for c in k["from_0based_inclusive"]:
        iv = V.from_0based_inclusive(*c["args"])
        assert iv == (c["start"], c["end"]) and V.half_open_len(iv) == c["len"]
    big = V.from_0based_inclusive((1 << 63) - 1, (1 << 63) - 1)
    assert big[1] >= big[0] and V.half_open_len(big) == k["from_0based_inclusive_i64_max_len"]
    for c in k["from_1based_inclusive"]:
        iv = V.from_1based_inclusive(*c["args"])
        assert iv == (c["start"], c["end"])
        if "len" in c:
            assert V.half_open_len(iv) == c["len"]
    for c in k["from_0based_point"]:
        assert V.from_0based_point(c["arg"]) == (c["start"], c["end"]) == V.from_0based_inclusive(c["arg"], c["arg"])
    assert V.half_open_len(tuple(k["reversed_len"]["interval"])) == k["reversed_len"]["len"]
    a, b = V.from_0based_inclusive(*k["slice"]["args"])
    assert k["slice"]["dna"][a:b] == k["slice"]["expect"]
    assert V.from_0based_inclusive(4, 9) == V.from_1based_inclusive(5, 10)


def test_process_variant_reference_cases(kats):
    """src/tests/filter_tests.rs (GQ filter flags, 1-based -> 0-based) and src/tests/mnp_test.rs (a mixed SNP/MNP ALT is dropped)."""
    for c in kats["process_variant_cases"]["cases"]:
        got = V.process_variant(c["line"], c["chrom"], [tuple(r) for r in c["regions"]], c["kept"], c["min_gq"], None, None)
        if c["expect"] is None:
            assert got is None, c["name"]
            continue
        variant, flags = got
        assert variant.position == c["expect"]["position"], c["name"]
        assert (flags == 0) == c["expect"]["flags_zero"], c["name"]


def test_sample_name_mapping(kats):
    """src/tests/sample_mapping_tests.rs: the core ID resolves to the index of the full VCF name."""
    from oracle import ferromic_ref as R

    k = kats["sample_mapping"]
    mapping = R.map_sample_names_to_indices(k["samples"])
    for name, idx in k["expect"].items():
        assert mapping.get(name) == idx


def test_core_sample_id(kats):
    for name, core in kats["core_sample_id"]["cases"]:
        assert V.core_sample_id(name) == core


def test_parse_region(kats):
    from oracle import ferromic_ref as R

    k = kats["parse_region"]
    for text, (a, b) in k["valid"]:
        assert V.parse_region(text) == (a, b)
    for text in k["invalid"]:
        with pytest.raises(R.VcfError) as info:
            V.parse_region(text)
        assert info.value.kind == "InvalidRegion", text


def test_validate_vcf_header(kats):
    from oracle import ferromic_ref as R

    for h in kats["vcf_header"]["valid"]:
        V.validate_vcf_header(h)
    for h in kats["vcf_header"]["invalid"]:
        with pytest.raises(R.VcfError) as info:
            V.validate_vcf_header(h)
        assert info.value.kind == "InvalidVcfFormat"


def test_find_vcf_file_reference_cases(tmp_path, kats):
    k = kats["find_vcf_file"]
    for name in k["files"]:
        (tmp_path / name).write_text("")
    for chrom, name in k["expect"].items():
        assert V.find_vcf_file(str(tmp_path), chrom).endswith(name)
    for chrom in k["missing"]:
        with pytest.raises(Exception):
            V.find_vcf_file(str(tmp_path), chrom)
    with pytest.raises(Exception):
        V.find_vcf_file("/non/existent/path", "1")


def test_config_with_placeholder_columns(tmp_path, kats):
    k = kats["config_with_noreads"]
    (tmp_path / "c.tsv").write_text(k["config"])
    assert len(V.parse_config_file(str(tmp_path / "c.tsv"))) == k["entries"]


def test_process_variants_reference_cases(kats):
    """src/tests/stats_tests.rs:1016-1481: haplotypes of a config group, segregating sites inside the group, theta_W with
    n = assigned haplotypes and L = region length unless an adjusted length is given."""
    from oracle import ferromic_ref as R

    for c in kats["process_variants"]["cases"]:
        variants = [R.make_variant(v["pos"], v["g"]) for v in c["variants"]]
        sample_filter = {k: tuple(v) for k, v in c["sample_filter"].items()}
        res = V.process_variants(variants, c["sample_names"], c["group"], sample_filter, tuple(c["interval"]), c["adjusted_len"],
                                 c["is_filtered"], frozenset(), None, None)
        assert res is not None, c["name"]
        segsites, theta, pi, n_hap, _ = res
        e = c["expect"]
        if "n_hap" in e:
            assert n_hap == e["n_hap"], c["name"]
        if "segsites" in e:
            assert segsites == e["segsites"], c["name"]
        if "theta" in e:
            num, h, L = e["theta"].split("/")
            expected = {"2/H2/2001": 2.0 / R.harmonic(2) / 2001.0, "12/11/100": 12.0 / 11.0 / 100.0}[e["theta"]]
            assert abs(theta - expected) < e["theta_abs_tol"], c["name"]
        if "inversion_frequency" in e:
            assert abs(R.calculate_inversion_allele_frequency(sample_filter) - e["inversion_frequency"]) < 1e-6
