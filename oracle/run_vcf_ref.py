"""CPU oracle for the `run_vcf` CLI surface: literal Python restatement of the statistic-bearing
spine of src/run_vcf.rs, src/parse.rs and src/process.rs (config TSV / BED / VCF text ->
variants + filter flags -> per-region statistics -> output.csv, FALSTA tracks, Hudson / W&C TSV).

TEST INFRASTRUCTURE ONLY (see oracle/ferromic_ref.py).  PHYLIP / CDS side effects, PCA, progress
bars and log files are outside the path and are not restated.  Pinned by the reference's own
end-to-end tests (tests/golden/reference_kats.json: falsta_zero_fill, falsta_hudson_tracks).
"""

from __future__ import annotations

import gzip
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import ferromic_ref as R

FLAG_PASS, FLAG_MASK, FLAG_ALLOW, FLAG_LOW_GQ, FLAG_MISSING = 0, 1, 2, 4, 8  # process.rs:785-789


# ---- parse.rs ---------------------------------------------------------------------------------


def parse_regions_file(path: str) -> Dict[str, List[Tuple[int, int]]]:
    """parse.rs:15-88 -> chr -> [(start, end)] 0-based half-open, sorted by start."""
    is_bed = path.endswith(".bed") and os.path.splitext(path)[1] == ".bed"
    regions: Dict[str, List[Tuple[int, int]]] = {}
    with open(path) as fh:
        for line in fh:
            fields = line.split()
            if len(fields) < 3:
                continue
            chrom = _trim_start_matches(fields[0], "chr")
            raw_start, raw_end = _parse_i64(fields[1]), _parse_i64(fields[2])
            if raw_start is None or raw_end is None:
                continue
            if is_bed:
                iv = (_as_usize(raw_start), _as_usize(raw_end))
            else:
                iv = R._hal_from_1based_inclusive(raw_start, raw_end)
            regions.setdefault(chrom, []).append(iv)
    for v in regions.values():
        v.sort(key=lambda iv: iv[0])
    return regions


def _as_usize(x: int) -> int:
    return x & ((1 << 64) - 1)


def _wrap_i64(x: int) -> int:
    """i64 arithmetic of a release build (no overflow checks): wraps."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


_I64_MAX = (1 << 63) - 1


def from_1based_inclusive(start_inclusive: int, end_inclusive: int) -> Tuple[int, int]:
    """ZeroBasedHalfOpen::from_1based_inclusive, process.rs:193-206 -> (start, end) 0-based half-open."""
    a = max(start_inclusive, 1)
    b = max(end_inclusive, a)
    return a - 1, b


def from_0based_inclusive(start_inclusive: int, end_inclusive: int) -> Tuple[int, int]:
    """ZeroBasedHalfOpen::from_0based_inclusive, process.rs:210-222 (saturating end + 1)."""
    a = max(start_inclusive, 0)
    b = a if end_inclusive < a else max(min(end_inclusive + 1, _I64_MAX), a)
    return a, b


def from_0based_point(p: int) -> Tuple[int, int]:
    """process.rs:225-231."""
    a = max(p, 0)
    return a, a + 1


def half_open_len(iv: Tuple[int, int]) -> int:
    """ZeroBasedHalfOpen::len, process.rs:234-240."""
    return iv[1] - iv[0] if iv[1] > iv[0] else 0


def _trim_start_matches(s: str, prefix: str) -> str:
    while s.startswith(prefix):
        s = s[len(prefix):]
    return s


@dataclass
class ConfigEntry:  # process.rs:397-403
    seqname: str
    interval: Tuple[int, int]  # ZeroBasedHalfOpen
    samples_unfiltered: Dict[str, Tuple[int, int]]
    samples_filtered: Dict[str, Tuple[int, int]]


def parse_config_file(path: str) -> List[ConfigEntry]:
    """parse.rs:91-239."""
    with open(path) as fh:
        lines = [ln.rstrip("\n").rstrip("\r") for ln in fh if ln.strip("\r\n") != ""]
    headers = lines[0].split("\t")
    sample_names = headers[7:]
    if not sample_names:
        raise R.VcfError("Parse", "No sample names found in config file header.")
    entries = []
    for line_num, line in enumerate(lines[1:]):
        record = line.split("\t")
        if len(record) != len(headers):
            raise R.VcfError("Parse", f"Mismatched number of fields in record on line {line_num + 2}")
        seqname = _trim_start_matches(record[0].strip(), "chr")
        start_pos, end_pos = _parse_i64(record[1]), _parse_i64(record[2])
        if start_pos is None:
            raise R.VcfError("Parse", "Invalid start")
        if end_pos is None:
            raise R.VcfError("Parse", "Invalid end")
        interval = R._hal_from_1based_inclusive(start_pos, end_pos)
        unf: Dict[str, Tuple[int, int]] = {}
        fil: Dict[str, Tuple[int, int]] = {}
        for i, fld in enumerate(record[7:]):
            name = sample_names[i]
            g = fld.split("_")[0]
            if len(g) >= 3 and g[1] == "|" and g[0] in "0123456789" and g[2] in "0123456789":
                left, right = int(g[0]), int(g[2])
                if left <= 1 and right <= 1:
                    unf[name] = (left, right)
            if fld in ("0|0", "0|1", "1|0", "1|1"):
                fil[name] = (int(fld[0]), int(fld[2]))
        if not unf:
            continue
        entries.append(ConfigEntry(seqname, interval, unf, fil))
    return entries


def parse_region(region: str) -> Tuple[int, int]:  # parse.rs:241-261
    parts = region.split("-")
    if len(parts) != 2:
        raise R.VcfError("InvalidRegion", "Invalid region format. Use start-end")
    def parse_i64(text: str, what: str) -> int:  # Rust str::parse::<i64>: optional sign, ASCII digits only
        body = text[1:] if text[:1] in "+-" else text
        if not body or not body.isascii() or not body.isdigit() or not -(1 << 63) <= int(text) < (1 << 63):
            raise R.VcfError("InvalidRegion", f"Invalid {what} position")
        return int(text)

    s, e = parse_i64(parts[0], "start"), parse_i64(parts[1], "end")
    if s >= e:
        raise R.VcfError("InvalidRegion", "Start position must be less than end position")
    return R._hal_from_1based_inclusive(s, e)


def validate_vcf_header(header: str) -> None:
    """parse.rs:529-543."""
    req = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]
    fields = header.split("\t")
    if len(fields) < len(req) or fields[:len(req)] != req:
        raise R.VcfError("InvalidVcfFormat", "Invalid VCF header format")


def core_sample_id(name: str) -> str:
    """stats.rs:1010-1012 -> normalize_sample_name_for_lookup (process.rs:1192-1196)."""
    return name[:-2] if name.endswith("_L") or name.endswith("_R") else name


def find_vcf_file(folder: str, chrom: str) -> str:
    """parse.rs:263-515."""
    if not os.path.isdir(folder):
        raise R.VcfError("Io", f"VCF folder does not exist: {folder}")
    for pattern in (f"chr{chrom}.vcf.gz", f"chr{chrom}.vcf", f"{chrom}.vcf.gz", f"{chrom}.vcf"):
        p = os.path.join(folder, pattern)
        if os.path.exists(p):
            return p
    invalid = (".csi", ".tbi", ".idx", ".md5", ".bai")

    def boundary_match(name: str) -> bool:
        for pattern in (f"chr{chrom}", chrom):
            start = 0
            while True:
                idx = name.find(pattern, start)
                if idx < 0:
                    break
                after = name[idx + len(pattern): idx + len(pattern) + 1]
                before = name[idx - 1: idx] if idx > 0 else ""
                if (after == "" or not after.isdigit()) and (before == "" or not before.isdigit()):
                    return True
                start = idx + 1
        return False

    def prefix_boundary(name: str, prefix: str) -> bool:
        if not name.startswith(prefix):
            return False
        rest = name[len(prefix):]
        return rest == "" or not rest[0].isdigit()

    cands = []
    for name in os.listdir(folder):
        if not (name.endswith(".vcf") or name.endswith(".vcf.gz")):
            continue
        if any(name.endswith(x) for x in invalid) or not boundary_match(name):
            continue
        score = 0
        if name == f"chr{chrom}.vcf.gz":
            score += 100
        elif name == f"chr{chrom}.vcf":
            score += 90
        elif name == f"{chrom}.vcf.gz":
            score += 80
        elif name == f"{chrom}.vcf":
            score += 70
        if name.endswith(".vcf.gz"):
            score += 15
        if prefix_boundary(name, f"chr{chrom}"):
            score += 10
        elif prefix_boundary(name, chrom):
            score += 5
        score -= len(name) // 5
        cands.append((-score, os.path.join(folder, name)))
    if not cands:
        raise R.VcfError("NoVcfFiles", "No VCF files found")
    cands.sort()
    return cands[0][1]


def open_text(path: str):
    # newline="\n": lines end at LF only and keep a CR, as BufRead::read_line does
    return gzip.open(path, "rt", newline="\n") if path.endswith(".gz") else open(path, newline="\n")


def read_fai(reference_path: str) -> Dict[str, Tuple[int, int, int, int]]:
    out = {}
    with open(reference_path + ".fai") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) >= 5:
                out[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
    return out


def read_reference_sequence(reference_path: str, chrom: str) -> bytes:
    """Whole-chromosome fetch through the .fai index (parse.rs:545-650, process.rs:1915-1952)."""
    fai = read_fai(reference_path)
    name = chrom if chrom in fai else f"chr{chrom}"
    if name not in fai:
        raise R.VcfError("Io", f"Chromosome {chrom} not found in reference")
    length, offset, line_bases, line_width = fai[name]
    out = bytearray()
    with open(reference_path, "rb") as fh:
        pos = 0
        while pos < length:
            line_idx, col = divmod(pos, line_bases)
            fh.seek(offset + line_idx * line_width + col)
            take = min(line_bases - col, length - pos)
            out += fh.read(take)
            pos += take
    return bytes(out)


def find_n_regions(seq: bytes, start_offset: int = 0) -> List[Tuple[int, int]]:  # process.rs:1849-1874
    regions = []
    in_n = False
    start_n = 0
    for i, b in enumerate(seq):
        is_n = b in (78, 110)
        if is_n and not in_n:
            in_n, start_n = True, i
        elif not is_n and in_n:
            in_n = False
            regions.append((start_offset + start_n, start_offset + i))
    if in_n:
        regions.append((start_offset + start_n, start_offset + len(seq)))
    return regions


def position_in_regions(pos: int, regions: Sequence[Tuple[int, int]]) -> bool:  # process.rs:738-744
    return any(s <= pos < e for s, e in regions)


def merge_intervals(intervals: List[Tuple[int, int]]) -> List[Tuple[int, int]]:  # process.rs:762-783
    if not intervals:
        return []
    intervals = sorted(intervals, key=lambda iv: iv[0])
    merged = []
    cs, ce = intervals[0]
    for s, e in intervals[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            merged.append((cs, ce))
            cs, ce = s, e
    merged.append((cs, ce))
    return merged


def _normalize_chr_prefix(c: str) -> str:
    for p in ("chr", "Chr", "CHR"):
        if c.startswith(p):
            return c[len(p):]
    return c


def _parse_i64(s: str) -> Optional[int]:
    """str::parse::<i64>: optional sign, ASCII digits only (no whitespace, no underscores), range checked."""
    body = s[1:] if s[:1] in ("+", "-") else s
    if not body or not body.isascii() or not body.isdigit():
        return None
    v = int(s)
    return v if -(1 << 63) <= v < (1 << 63) else None


def _parse_u8(s: str) -> Optional[int]:
    if s.startswith("+"):
        s = s[1:]
    if not s or not s.isdigit() or not s.isascii():
        return None
    v = int(s)
    return v if v <= 255 else None


def _parse_u16(s: str) -> Optional[int]:
    if s.startswith("+"):
        s = s[1:]
    if not s or not s.isdigit() or not s.isascii():
        return None
    v = int(s)
    return v if v <= 65535 else None


def process_variant(line: str, chrom: str, regions, kept_cols, min_gq: int, allow, mask):
    """process.rs:4471-4768 -> None | (Variant, flags)."""
    fields = line.split("\t")
    if len(fields) < 9:
        raise R.VcfError("Parse", "Invalid VCF line format")
    if kept_cols and len(fields) <= max(kept_cols):
        raise R.VcfError("Parse", "Invalid VCF line format: missing genotype column")
    vcf_chr = _normalize_chr_prefix(fields[0].strip())
    if vcf_chr != _normalize_chr_prefix(chrom.strip()):
        return None
    pos1 = _parse_i64(fields[1])
    if pos1 is None:
        raise R.VcfError("Parse", "Invalid position")
    if pos1 < 1:
        raise R.VcfError("Parse", f"Invalid 1-based pos: {pos1}")
    pos0 = pos1 - 1
    if not any(s <= pos0 < e for s, e in regions):  # position_in_zero_based_regions
        return None
    flags = FLAG_PASS
    if allow is not None:
        a = allow.get(vcf_chr)
        if a is not None:
            if not position_in_regions(pos0, a):
                flags |= FLAG_ALLOW
        else:
            flags |= FLAG_ALLOW
    if mask is not None:
        m = mask.get(vcf_chr)
        if m is not None and any(max(pos0, s) < min(pos0 + 1, e) for s, e in m):
            flags |= FLAG_MASK
    alts = fields[4].split(",")
    indel = len(fields[3]) != 1 or any(len(a) != 1 for a in alts)
    fmt = fields[8].split(":")
    if "GQ" not in fmt:
        raise R.VcfError("Parse", "GQ field not found in FORMAT")
    gq_index = fmt.index("GQ")
    raw: List[Optional[List[int]]] = []
    for idx in kept_cols:
        alleles_str = fields[idx].split(":")[0]
        if alleles_str in (".", "./.", ".|."):
            raw.append(None)
            continue
        parts = alleles_str.replace("/", "|").split("|")
        vals = [_parse_u8(p) for p in parts]
        raw.append(None if any(v is None for v in vals) else vals)
    low_gq = False
    for i, idx in enumerate(kept_cols):
        if raw[i] is None:
            continue
        parts = fields[idx].split(":")
        if gq_index >= len(parts):
            raise R.VcfError("Parse", "GQ value missing in sample genotype field")
        gq_str = parts[gq_index].strip()
        gq = 0 if gq_str in (".", "") else (_parse_u16(gq_str) or 0)
        if gq < min_gq:
            low_gq = True
    if low_gq:
        flags |= FLAG_LOW_GQ
    if any(g is None for g in raw):
        flags |= FLAG_MISSING
    if indel:
        return None
    return R.make_variant(pos0, raw), flags


def process_vcf(path: str, chrom: str, regions, min_gq: int, mask, allow, exclusion_set):
    """process.rs:4092-4469 -> (variants, flags, sample_names); sorted by (position, genotype bytes)."""
    items = []
    sample_names: List[str] = []
    kept: List[int] = []
    with open_text(path) as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            if line.startswith("#CHROM"):
                hdr = line.split("\t")
                req = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]
                if len(hdr) < 9 or hdr[:9] != req:
                    raise R.VcfError("InvalidVcfFormat", "Invalid VCF header format")
                for idx, name in enumerate(line.split()):
                    if idx >= 9 and name not in exclusion_set:
                        sample_names.append(name)
                        kept.append(idx)
                break
        if not sample_names:
            raise R.VcfError("Parse", "No samples remain after applying exclusions")
        for line in fh:
            try:
                res = process_variant(line, chrom, regions, kept, min_gq, allow, mask)
            except R.VcfError:
                continue  # the collector prints the error and carries on (process.rs:4358-4360)
            if res is not None:
                items.append(res)
    items.sort(key=lambda it: (it[0].position, bytes(it[0].genotypes.data)))
    return [v for v, _ in items], [f for _, f in items], sample_names


def get_haplotype_indices_for_group(group: int, sample_filter, index_map):  # process.rs:1279-1333
    out = []
    for name, (left, right) in sample_filter.items():
        idx = index_map.get(R.normalize_sample_name_for_lookup(name))
        if idx is None:
            continue
        if left == group:
            out.append((idx, R.LEFT))
        if right == group:
            out.append((idx, R.RIGHT))
    return out


def process_variants(variants, sample_names, group, sample_filter, interval, adjusted_len, is_filtered,
                     filtered_positions, mask_intervals, dense_matrix):
    """process.rs:821-1188 (statistics only) -> None | (segsites, theta, pi, n_hap, site_diversities)."""
    index_map = R.map_sample_names_to_indices(sample_names)
    group_haps = get_haplotype_indices_for_group(group, sample_filter, index_map)
    if not group_haps:
        return None
    n_hap = len(group_haps)
    if not variants:
        theta, pi = (R.NAN, R.NAN) if n_hap < 2 else (0.0, 0.0)
        return 0, theta, pi, n_hap, []
    in_region = [v for v in variants if interval[0] <= v.position < interval[1]]
    segsites = 0
    for v in in_region:
        vals = []
        for idx, side in group_haps:
            g = v.genotypes.get(idx)
            if g is not None and side < len(g):
                vals.append(g[side])
        if len(set(vals)) > 1:
            segsites += 1
    qr = R.QueryRegion(interval[0], interval[1] - 1 if interval[1] > 0 else -1)
    L = adjusted_len if adjusted_len is not None else max(interval[1] - interval[0], 0)
    theta = R.calculate_watterson_theta(segsites, n_hap, L)
    if dense_matrix is not None:
        ctx = R.PopulationContext(group, list(group_haps), in_region, sample_names, L, dense_matrix, None)
        pi = R.calculate_pi_for_population(ctx)
    else:
        pi = R.calculate_pi(in_region, group_haps, L)
    site_divs = R.calculate_per_site_diversity(variants, group_haps, qr, filtered_positions, mask_intervals)
    return segsites, theta, pi, n_hap, site_divs


# ---- formatting (Rust `{:.6}` etc.) ----------------------------------------------------------------


def fmt6(x: float) -> str:
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return f"{x:.6f}"


def format_optional_float(v: Optional[float]) -> str:  # process.rs:3702-3713
    if v is None or v != v:
        return "NA"
    return fmt6(v)


def format_optional_usize(v: Optional[int]) -> str:
    return "NA" if v is None else str(v)


CSV_HEADER = [  # process.rs:1745-1787
    "chr", "region_start", "region_end", "0_sequence_length", "1_sequence_length", "0_sequence_length_adjusted",
    "1_sequence_length_adjusted", "0_segregating_sites", "1_segregating_sites", "0_w_theta", "1_w_theta", "0_pi", "1_pi",
    "0_segregating_sites_filtered", "1_segregating_sites_filtered", "0_w_theta_filtered", "1_w_theta_filtered",
    "0_pi_filtered", "1_pi_filtered", "0_num_hap_no_filter", "1_num_hap_no_filter", "0_num_hap_filter", "1_num_hap_filter",
    "inversion_freq_no_filter", "inversion_freq_filter", "haplotype_overall_fst_wc", "haplotype_between_pop_variance_wc",
    "haplotype_within_pop_variance_wc", "haplotype_num_informative_sites_wc", "hudson_fst_hap_group_0v1",
    "hudson_dxy_hap_group_0v1", "hudson_pi_hap_group_0", "hudson_pi_hap_group_1", "hudson_pi_avg_hap_group_0v1",
]


@dataclass
class RegionResult:
    csv_row: List[str]
    seqname: str
    region_start: int  # 1-based inclusive
    region_end: int
    diversity: List[Tuple[int, float, float, int, bool]]
    wc_sites: List[Tuple[int, float, float, float, float, float, float]]
    hudson_sites: List[Tuple[int, float, float, float]]
    hudson_rows: List[List[str]] = field(default_factory=list)
    wc_rows: List[List[str]] = field(default_factory=list)


def _falsta_value_div(v: float) -> str:  # process.rs:3786-3792
    if v != v:
        return "NA"
    if v == 0.0:
        return "0"
    return fmt6(v)


def _falsta_value_fst(v: float) -> str:  # process.rs:3842-3856
    if v != v:
        return "NA"
    if math.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    if v == 0.0:
        return "0"
    return fmt6(v)


def diversity_falsta_text(r: RegionResult) -> str:
    """append_diversity_falsta, process.rs:3740-3806."""
    if not r.diversity:
        return ""
    region = R._hal_from_1based_inclusive(r.region_start, r.region_end)
    n = R._hal_len(region)
    out = []
    for g in sorted({d[3] for d in r.diversity}):
        for is_filtered, which, prefix in ((False, "pi", "unfiltered_pi_"), (False, "theta", "unfiltered_theta_"),
                                           (True, "pi", "filtered_pi_"), (True, "theta", "filtered_theta_")):
            line = ["0"] * n
            any_ = False
            for pos1, pi, th, gg, filt in r.diversity:
                if gg != g or filt != is_filtered:
                    continue
                p = pos1 - 1
                if region[0] <= p < region[1]:
                    line[p - region[0]] = _falsta_value_div(pi if which == "pi" else th)
                    any_ = True
            if any_:
                out.append(f">{prefix}chr_{r.seqname}_start_{r.region_start}_end_{r.region_end}_group_{g}")
                out.append(",".join(line))
    return "".join(x + "\n" for x in out)


def fst_falsta_text(r: RegionResult) -> str:
    """append_fst_falsta, process.rs:3809-4003."""
    if not r.wc_sites and not r.hudson_sites:
        return ""
    region = R._hal_from_1based_inclusive(r.region_start, r.region_end)
    n = R._hal_len(region)
    out = []
    suffix = f"chr_{r.seqname}_start_{r.region_start}_end_{r.region_end}"

    def track(header, sites, col):
        v = ["NA"] * n
        for s in sites:
            p = s[0] - 1
            if region[0] <= p < region[1]:
                v[p - region[0]] = _falsta_value_fst(s[col])
        out.append(f">{header}_{suffix}")
        out.append(",".join(v))

    if r.wc_sites:
        track("haplotype_overall_fst_summary", r.wc_sites, 1)
        track("haplotype_overall_fst_numerator", r.wc_sites, 2)
        track("haplotype_overall_fst_denominator", r.wc_sites, 3)
        track("haplotype_0v1_pairwise_fst_summary", r.wc_sites, 4)
        track("haplotype_0v1_pairwise_fst_numerator", r.wc_sites, 5)
        track("haplotype_0v1_pairwise_fst_denominator", r.wc_sites, 6)
    if r.hudson_sites:
        track("hudson_pairwise_fst_hap_0v1", r.hudson_sites, 1)
        track("hudson_pairwise_fst_hap_0v1_numerator", r.hudson_sites, 2)
        track("hudson_pairwise_fst_hap_0v1_denominator", r.hudson_sites, 3)
    return "".join(x + "\n" for x in out)


def _pop_id_fmt(pid):  # process.rs:3692-3698
    if pid is None:
        return "NA", "NA"
    if isinstance(pid, int):
        return "HaplotypeGroup", str(pid)
    return "NamedPopulation", str(pid)


HUDSON_TSV_HEADER = ["chr", "region_start_0based", "region_end_0based", "pop1_id_type", "pop1_id_name", "pop2_id_type",
                     "pop2_id_name", "Dxy", "pi_pop1", "pi_pop2", "pi_xy_avg", "FST"]
WC_TSV_HEADER = ["chr", "region_start_1based", "region_end_1based", "comparison_type", "pop1", "pop2", "fst",
                 "numerator_a", "denominator_a_plus_b", "informative_sites"]


def process_single_config_entry(entry: ConfigEntry, all_variants, all_flags, sample_names, mask, allow, chr_length,
                                chrom, enable_fst, csv_populations=None, csv_path=None) -> Optional[RegionResult]:
    """process.rs:2468-3653 (statistics and records only)."""
    ext = R._hal_from_1based_inclusive(max(entry.interval[0] - 3_000_000, 0), min(_wrap_i64(entry.interval[1] + 3_000_000), chr_length))
    sl = [(v, f) for v, f in zip(all_variants, all_flags) if ext[0] <= v.position < ext[1]]
    allow_chr = allow.get(chrom) if allow is not None else None
    mask_chr = mask.get(chrom) if mask is not None else None

    def in_iv(p):
        return entry.interval[0] <= p < entry.interval[1]

    unf = [v for v, _ in sl if in_iv(v.position)
           and (allow_chr is None or position_in_regions(v.position, allow_chr))
           and (mask_chr is None or not position_in_regions(v.position, mask_chr))]
    fil = [v for v, f in sl if f == FLAG_PASS and in_iv(v.position)]
    dense_unf = R.DenseGenotypeMatrix.from_variants(unf, len(sample_names))
    dense_fil = R.DenseGenotypeMatrix.from_variants(fil, len(sample_names))
    filtered_positions_in_region: set = set()  # FilteringStats.filtered_positions is never filled here (2568-2571)
    num_excluded = 0
    index_map = R.map_sample_names_to_indices(sample_names)

    wc = None
    pop_wc = None
    if enable_fst:
        qr = R.QueryRegion(entry.interval[0], entry.interval[1] - 1 if entry.interval[1] > 0 else -1)
        wc = R.calculate_fst_wc_haplotype_groups(fil, sample_names, entry.samples_filtered, qr)
        if csv_path is not None:  # process.rs:2769-2802
            try:
                pop_wc = R.calculate_fst_wc_csv_populations(fil, sample_names, csv_path, qr)
            except (R.VcfError, OSError):
                pop_wc = None
    sequence_length = entry.interval[1] - entry.interval[0]
    adj = R.calculate_adjusted_sequence_length(entry.interval[0] + 1, entry.interval[1], allow_chr, mask_chr)
    callable_fraction = adj / sequence_length if sequence_length > 0 else R.NAN
    masked_fraction = 1.0 - callable_fraction
    if not math.isfinite(callable_fraction) or masked_fraction >= 0.99:
        return None
    fil_adj = R.saturating_sub_i64(adj, num_excluded)
    calls = [
        (0, True, fil, entry.samples_filtered, fil_adj, filtered_positions_in_region, dense_fil),
        (1, True, fil, entry.samples_filtered, fil_adj, filtered_positions_in_region, dense_fil),
        (0, False, unf, entry.samples_unfiltered, adj, set(), dense_unf),
        (1, False, unf, entry.samples_unfiltered, adj, set(), dense_unf),
    ]
    results = [process_variants(vs, sample_names, g, sf, entry.interval, L, isf, fp, mask_chr, dm)
               for g, isf, vs, sf, L, fp, dm in calls]
    if all(r is None for r in results):
        return None
    dflt = (0, 0.0, 0.0, 0, [])
    s0f, t0f, p0f, n0f, d0f = results[0] or dflt
    s1f, t1f, p1f, n1f, d1f = results[1] or dflt
    s0u, t0u, p0u, n0u, d0u = results[2] or dflt
    s1u, t1u, p1u, n1u, d1u = results[3] or dflt
    inv_f = R.calculate_inversion_allele_frequency(entry.samples_filtered)
    inv_u = R.calculate_inversion_allele_frequency(entry.samples_unfiltered)
    inv_f = -1.0 if inv_f is None else inv_f
    inv_u = -1.0 if inv_u is None else inv_u

    hud = dict(fst=None, dxy=None, pi0=None, pi1=None, avg=None)
    hudson_sites: List[Tuple[int, float, float, float]] = []
    hudson_rows: List[List[str]] = []
    region_start0, region_end0 = entry.interval[0], entry.interval[1] - 1
    hqr = R.QueryRegion(entry.interval[0], entry.interval[1] - 1) if entry.interval[1] > entry.interval[0] else R.QueryRegion(0, -1)

    def hudson_row(outcome):
        t1, n1 = _pop_id_fmt(outcome.pop1_id)
        t2, n2 = _pop_id_fmt(outcome.pop2_id)
        return [entry.seqname, str(region_start0), str(region_end0), t1, n1, t2, n2, format_optional_float(outcome.d_xy),
                format_optional_float(outcome.pi_pop1), format_optional_float(outcome.pi_pop2),
                format_optional_float(outcome.pi_xy_avg), format_optional_float(outcome.fst)]

    if enable_fst:
        h0 = get_haplotype_indices_for_group(0, entry.samples_filtered, index_map)
        h1 = get_haplotype_indices_for_group(1, entry.samples_filtered, index_map)
        if len(h0) >= 2 and len(h1) >= 2 and hqr.start <= hqr.end:
            p0 = R.PopulationContext(0, h0, fil, sample_names, fil_adj, dense_fil, None)
            p1 = R.PopulationContext(1, h1, fil, sample_names, fil_adj, dense_fil, None)
            try:
                outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p0, p1, hqr)
                hudson_rows.append(hudson_row(outcome))
                informative = sum(1 for s in sites if s.den_component is not None and math.isfinite(s.den_component) and s.den_component > 0.0)
                if informative > 0:
                    for s in sites:
                        hudson_sites.append((s.position, R.NAN if s.fst is None else s.fst,
                                             R.NAN if s.num_component is None else s.num_component,
                                             R.NAN if s.den_component is None else s.den_component))
                hud = dict(fst=outcome.fst, dxy=outcome.d_xy, pi0=outcome.pi_pop1, pi1=outcome.pi_pop2, avg=outcome.pi_xy_avg)
            except R.VcfError:
                pass
        if csv_populations is not None:
            pop_haps = {}
            for name, ids in csv_populations.items():
                hl = []
                for sid in ids:
                    if sid in index_map:
                        hl += [(index_map[sid], R.LEFT), (index_map[sid], R.RIGHT)]
                if hl:
                    pop_haps[name] = hl
            names = sorted(pop_haps)
            for i in range(len(names)):
                for j in range(i + 1, len(names)):
                    a, b = pop_haps[names[i]], pop_haps[names[j]]
                    if len(a) >= 2 and len(b) >= 2 and hqr.start <= hqr.end:
                        pa = R.PopulationContext(names[i], a, fil, sample_names, fil_adj, dense_fil, None)
                        pb = R.PopulationContext(names[j], b, fil, sample_names, fil_adj, dense_fil, None)
                        try:
                            outcome, _ = R.calculate_hudson_fst_for_pair_with_sites(pa, pb, hqr)
                            hudson_rows.append(hudson_row(outcome))
                        except R.VcfError:
                            pass

    if wc is not None:
        e = wc.overall_fst
        hap = (e.value, e.sum_a, e.sum_b, e.sites)
    else:
        hap = (None, 0.0, 0.0, 0)
    rs1, re1 = entry.interval[0] + 1, entry.interval[1]
    row = [entry.seqname, str(rs1), str(re1), str(sequence_length), str(sequence_length), str(adj), str(adj),
           str(s0u), str(s1u), fmt6(t0u), fmt6(t1u), fmt6(p0u), fmt6(p1u), str(s0f), str(s1f), fmt6(t0f), fmt6(t1f),
           fmt6(p0f), fmt6(p1f), str(n0u), str(n1u), str(n0f), str(n1f), fmt6(inv_u), fmt6(inv_f),
           format_optional_float(hap[0]), format_optional_float(hap[1]), format_optional_float(hap[2]),
           format_optional_usize(hap[3]), format_optional_float(hud["fst"]), format_optional_float(hud["dxy"]),
           format_optional_float(hud["pi0"]), format_optional_float(hud["pi1"]), format_optional_float(hud["avg"])]
    diversity = ([(d.position, d.pi, d.watterson_theta, 0, False) for d in d0u]
                 + [(d.position, d.pi, d.watterson_theta, 1, False) for d in d1u]
                 + [(d.position, d.pi, d.watterson_theta, 0, True) for d in d0f]
                 + [(d.position, d.pi, d.watterson_theta, 1, True) for d in d1f])
    wc_sites = []
    if wc is not None:
        for s in wc.site_fst:
            ov = s.overall_fst.value if s.overall_fst.state == "calculable" else R.NAN
            pe = s.pairwise_fst.get("0_vs_1")
            pv = pe.value if (pe is not None and pe.state == "calculable") else R.NAN
            on, ow = s.variance_components
            pn, pw = s.pairwise_variance_components.get("0_vs_1", (R.NAN, R.NAN))
            wc_sites.append((s.position, ov, on, on + ow, pv, pn, pn + pw))
    wc_rows: List[List[str]] = []
    if pop_wc is not None:  # RegionalWcFSTOutcome rows, process.rs:1662-1716
        def wc_row(kind, p1, p2, est):
            denom = est.sum_a + est.sum_b
            return [entry.seqname, str(rs1), str(re1), kind, p1, p2, format_optional_float(est.value),
                    format_optional_float(est.sum_a), format_optional_float(denom), format_optional_usize(est.sites)]
        wc_rows.append(wc_row("overall", "ALL", "ALL", pop_wc.overall_fst))
        for key in sorted(pop_wc.pairwise_fst):
            parts = key.split("_vs_")
            p1, p2 = (parts[0], parts[1]) if len(parts) == 2 else ("unknown", "unknown")
            wc_rows.append(wc_row("pairwise", p1, p2, pop_wc.pairwise_fst[key]))
    return RegionResult(row, entry.seqname, rs1, re1, diversity, wc_sites, hudson_sites, hudson_rows, wc_rows)


def resolve_sample_exclusions(vcf_folder, chrom, requested, config_entries):
    """run_vcf.rs:24-187: exact name, else every known sample containing the request as a substring."""
    if not requested:
        return set()
    vcf_ids, cfg_ids = set(), set()
    try:
        with open_text(find_vcf_file(vcf_folder, chrom)) as fh:
            for line in fh:
                if line.startswith("#CHROM"):
                    vcf_ids.update(line.split()[9:])
                    break
    except (R.VcfError, OSError):
        pass
    if config_entries is not None:
        for e in config_entries:
            cfg_ids.update(e.samples_unfiltered)
            cfg_ids.update(e.samples_filtered)
    if not vcf_ids and not cfg_ids:
        return set(requested)
    resolved = set()
    for req in sorted(requested):
        t = req.strip()
        for ids in (vcf_ids, cfg_ids):
            if t in ids:
                resolved.add(t)
            else:
                resolved.update(s for s in ids if t in s)
    return resolved


def run(vcf_folder: str, reference: str, gtf: str, output_file: str, config_file: Optional[str] = None,
        chrom: Optional[str] = None, region: Optional[str] = None, min_gq: int = 30, mask_file: Optional[str] = None,
        allow_file: Optional[str] = None, exclude: Sequence[str] = (), enable_fst: bool = False,
        fst_populations: Optional[str] = None) -> Dict[str, str]:
    """run_vcf.rs:216-486 + process_config_entries (process.rs:1335-1730).  Writes the output files
    next to `output_file` and returns their decompressed text by name."""
    mask = parse_regions_file(mask_file) if mask_file else None
    allow = parse_regions_file(allow_file) if allow_file else None
    exclusion = set(exclude)
    if config_file:
        entries = parse_config_file(config_file)
        if entries:
            exclusion = resolve_sample_exclusions(vcf_folder, entries[0].seqname, exclusion, entries)
        for e in entries:
            for name in list(e.samples_unfiltered):
                if name in exclusion:
                    del e.samples_unfiltered[name]
            for name in list(e.samples_filtered):
                if name in exclusion:
                    del e.samples_filtered[name]
    elif chrom:
        exclusion = resolve_sample_exclusions(vcf_folder, chrom, exclusion, None)
        interval = parse_region(region) if region else R._hal_from_1based_inclusive(1, (1 << 63) - 1)
        path = find_vcf_file(vcf_folder, chrom)
        names = []
        with open_text(path) as fh:
            for line in fh:
                if line.startswith("#CHROM"):
                    names = [n for n in line.split()[9:] if n not in exclusion]
                    break
        if not names:
            raise R.VcfError("Parse", "No samples remain after applying exclusions")
        g = {n: (0, 0) for n in names}
        entries = [ConfigEntry(chrom, interval, dict(g), dict(g))]
    else:
        raise R.VcfError("Parse", "Either --config_file or --chr must be specified")

    csv_populations = None
    if enable_fst and fst_populations:  # process.rs:1394-1426
        try:
            csv_populations = {k: [x for x in v if x not in exclusion] for k, v in R.parse_population_csv(fst_populations).items()}
        except (R.VcfError, OSError):
            csv_populations = None
    wc_rows: List[List[str]] = []
    by_chr: Dict[str, List[ConfigEntry]] = {}
    for e in entries:
        by_chr.setdefault(e.seqname, []).append(e)
    out_dir = os.path.dirname(os.path.abspath(output_file))
    os.makedirs(out_dir, exist_ok=True)
    csv_lines = [",".join(CSV_HEADER)]
    div_text, fst_text = [], []
    hudson_rows: List[List[str]] = []
    for chrom_name in sorted(by_chr):
        chr_entries = by_chr[chrom_name]
        try:
            if not os.path.exists(gtf):
                raise R.VcfError("Io", "GTF not found")
            ref_seq = read_reference_sequence(reference, chrom_name)
            chr_length = len(ref_seq)
            gmask = {k: list(v) for k, v in (mask or {}).items()}
            gmask.setdefault(chrom_name, []).extend(find_n_regions(ref_seq, 0))
            try:
                vcf_path = find_vcf_file(vcf_folder, chrom_name)
            except R.VcfError:
                continue
            hulls = [(max(e.interval[0] - 3_000_000, 0), _as_usize(min(_wrap_i64(e.interval[1] + 3_000_000), chr_length))) for e in chr_entries]
            merged = merge_intervals(hulls)
            try:
                variants, flags, sample_names = process_vcf(vcf_path, chrom_name, merged, min_gq, gmask, allow, exclusion)
            except R.VcfError:
                continue
        except (R.VcfError, OSError):
            continue
        for e in chr_entries:
            try:
                res = process_single_config_entry(e, variants, flags, sample_names, gmask, allow, chr_length, chrom_name, enable_fst,
                                                  csv_populations, fst_populations if enable_fst else None)
            except R.VcfError:
                continue
            if res is None:
                continue
            csv_lines.append(",".join(res.csv_row))
            div_text.append(diversity_falsta_text(res))
            fst_text.append(fst_falsta_text(res))
            hudson_rows += res.hudson_rows
            wc_rows += res.wc_rows
    outputs = {os.path.basename(output_file): "".join(x + "\n" for x in csv_lines)}
    with open(output_file, "w") as fh:
        fh.write(outputs[os.path.basename(output_file)])
    for name, chunks in (("per_site_diversity_output.falsta.gz", div_text), ("per_site_fst_output.falsta.gz", fst_text)):
        text = "".join(chunks)
        if text:
            outputs[name] = text
            with gzip.open(os.path.join(out_dir, name), "wt") as fh:
                fh.write(text)
    if enable_fst:
        text = "".join("\t".join(r) + "\n" for r in [HUDSON_TSV_HEADER] + hudson_rows)
        outputs["hudson_fst_results.tsv.gz"] = text
        with gzip.open(os.path.join(out_dir, "hudson_fst_results.tsv.gz"), "wt") as fh:
            fh.write(text)
        if wc_rows:  # process.rs:1628-1726
            text = "".join("\t".join(r) + "\n" for r in [WC_TSV_HEADER] + wc_rows)
            outputs["wc_fst_results.tsv.gz"] = text
            with gzip.open(os.path.join(out_dir, "wc_fst_results.tsv.gz"), "wt") as fh:
                fh.write(text)
    return outputs
