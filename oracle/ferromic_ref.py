"""CPU oracle: literal Python restatement of ferromic's per-site diversity / FST path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``ferromic_amd`` / ``libferromic_hip.so``) never
imports, links or executes this file.

Every function cites the reference file:line it restates (paths relative to the reference
repository, ``src/stats.rs`` unless said otherwise).  The restatement is deliberately literal:
same branch order, same formulas and operation order, same epsilon, same ``None``/NaN rules.
Python floats are IEEE-754 binary64 and Python does not contract a*b+c into an FMA, so every
per-site value below is bit-identical to what the Rust code computes for the same counts.

Pinning: checked in ``tests/test_oracle_golden.py`` against the known-answer vectors the
reference's own tests hold (``tests/golden/reference_kats.json``, transcribed from
``src/tests/stats_tests.rs``, ``src/tests/hudson_fst_tests.rs``, ``src/pytests/*.py``).
Weir & Cockerham has no reference test anywhere ("parity unpinned" for W&C): it is pinned
only by stats.rs:1781-2374 and analytic vectors derived from those formulas.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

FST_EPSILON = 1e-12  # stats.rs:26
MISSING = 0xFF  # process.rs:438
LEFT, RIGHT = 0, 1  # process.rs:391-394 HaplotypeSide
INVALID_GROUP = 0xFFFF  # stats.rs:1080

NAN = float("nan")
INF = float("inf")


# ----------------------------------------------------------------------------------------------
# Data model (process.rs:428-585)
# ----------------------------------------------------------------------------------------------


class CompressedGenotypes:
    """process.rs:431-512 — flat u8 rows, 0xFF sentinel, stride = max ploidy seen (>=1)."""

    def __init__(self, raw: Sequence[Optional[Sequence[int]]]):
        self.num_samples = len(raw)
        max_ploidy = max((len(g) for g in raw if g is not None), default=0)
        if self.num_samples > 0:
            max_ploidy = max(max_ploidy, 1)
        self.stride = max_ploidy
        if self.num_samples == 0 or max_ploidy == 0:
            self.data = bytearray()
            return
        flat = bytearray([MISSING]) * (self.num_samples * max_ploidy)
        for sample_idx, genotype in enumerate(raw):
            start = sample_idx * max_ploidy
            if genotype is not None:
                for offset, allele in enumerate(genotype):
                    if offset >= max_ploidy:
                        break
                    flat[start + offset] = int(allele) & 0xFF
        self.data = flat

    def get(self, index: int) -> Optional[List[int]]:
        # process.rs:479-496
        if index >= self.num_samples or self.stride == 0:
            return None
        start = index * self.stride
        if start >= len(self.data) or self.data[start] == MISSING:
            return None
        genotype = []
        for offset in range(self.stride):
            byte = self.data[start + offset]
            if byte == MISSING:
                break
            genotype.append(byte)
        return genotype

    def __iter__(self):
        for idx in range(self.num_samples):
            yield self.get(idx)

    def __len__(self):
        return self.num_samples


@dataclass
class Variant:  # process.rs:532-536
    position: int
    genotypes: CompressedGenotypes


def make_variant(position: int, genotypes: Sequence[Optional[Sequence[int]]]) -> Variant:
    return Variant(int(position), CompressedGenotypes(genotypes))


@dataclass
class QueryRegion:  # process.rs:559-585 — 0-based inclusive
    start: int
    end: int

    def contains(self, pos: int) -> bool:
        return self.start <= pos <= self.end

    def len(self) -> int:
        if self.start > self.end:
            return 0
        # ZeroBasedHalfOpen::from_0based_inclusive (process.rs:210-222)
        adjusted_start = max(self.start, 0)
        if self.end < adjusted_start:
            adjusted_end = adjusted_start
        else:
            adjusted_end = max(self.end + 1, adjusted_start)
        return adjusted_end - adjusted_start if adjusted_end > adjusted_start else 0


class DenseGenotypeMatrix:
    """stats.rs:250-331.  ``data`` is bytes-like of length S*N*ploidy; ``missing`` is a list of
    u64 words (LSB-first bit per linear entry) or None."""

    def __init__(self, data, missing, variant_count, sample_count, ploidy, max_allele):
        assert len(data) == variant_count * sample_count * ploidy  # stats.rs:276-284
        self.data = data
        self.missing = missing
        self.variant_count = variant_count
        self.sample_count = sample_count
        self.ploidy = ploidy
        self.stride = sample_count * ploidy
        self.max_allele = max_allele

    @staticmethod
    def from_variants(variants: Sequence[Variant], sample_count: int) -> Optional["DenseGenotypeMatrix"]:
        # stats.rs:339-500
        if not variants:
            return None
        max_ploidy = 0
        for v in variants:
            for g in v.genotypes:
                if g is not None:
                    max_ploidy = max(max_ploidy, len(g))
        if max_ploidy == 0:
            return None
        variant_count = len(variants)
        stride = sample_count * max_ploidy
        total = variant_count * stride
        data = bytearray(total)
        missing_bytes = bytearray(total)
        for vi, variant in enumerate(variants):
            base = vi * stride
            for sample_idx in range(sample_count):
                offset = sample_idx * max_ploidy
                gt = variant.genotypes.get(sample_idx)
                if gt is not None:
                    ln = len(gt)
                    for i in range(min(ln, max_ploidy)):
                        data[base + offset + i] = gt[i]
                    for i in range(ln, max_ploidy):
                        missing_bytes[base + offset + i] = 1
                else:
                    for i in range(max_ploidy):
                        missing_bytes[base + offset + i] = 1
        words = (total + 63) // 64
        missing = [0] * words
        for idx in range(total):
            if missing_bytes[idx] == 1:
                missing[idx >> 6] |= 1 << (idx & 63)
        global_max = max(data) if total else 0
        return DenseGenotypeMatrix(bytes(data), missing, variant_count, sample_count, max_ploidy, global_max)


def dense_missing(bits, idx: int) -> bool:  # stats.rs:1298-1302
    return ((bits[idx >> 6] >> (idx & 63)) & 1) == 1


# ----------------------------------------------------------------------------------------------
# Memberships (stats.rs:1093-1295)
# ----------------------------------------------------------------------------------------------


@dataclass
class HapMembership:  # stats.rs:1204-1244
    left: List[bool]
    right: List[bool]
    total: int

    @staticmethod
    def build(sample_count: int, haplotypes: Iterable[Tuple[int, int]]) -> "HapMembership":
        left = [False] * sample_count
        right = [False] * sample_count
        total = 0
        for sample_idx, side in haplotypes:
            if sample_idx >= sample_count:
                continue
            if side == LEFT:
                if not left[sample_idx]:
                    left[sample_idx] = True
                    total += 1
            else:
                if not right[sample_idx]:
                    right[sample_idx] = True
                    total += 1
        return HapMembership(left, right, total)


def dense_membership_offsets(matrix: DenseGenotypeMatrix, haplotypes: Iterable[Tuple[int, int]]) -> List[int]:
    """DenseMembership::build, stats.rs:1252-1284 — dedup'd sorted byte offsets within a row."""
    sample_count = matrix.sample_count
    ploidy = matrix.ploidy
    left = [False] * sample_count
    right = [False] * sample_count
    offsets = []
    for sample_idx, side in haplotypes:
        if sample_idx >= sample_count:
            continue
        if side == LEFT:
            if not left[sample_idx]:
                left[sample_idx] = True
                offsets.append(sample_idx * ploidy)
        else:
            if ploidy <= 1:
                continue
            if not right[sample_idx]:
                right[sample_idx] = True
                offsets.append(sample_idx * ploidy + 1)
    offsets.sort()
    return offsets


@dataclass
class SubpopulationMembership:  # stats.rs:1093-1159
    left: List[int]
    right: List[int]
    labels: List[str]
    pair_keys: List[Tuple[int, int, str]]

    @staticmethod
    def from_map(sample_count: int, map_subpop: Dict[Tuple[int, int], str]) -> "SubpopulationMembership":
        labels = sorted(set(map_subpop.values()))
        label_to_index = {label: idx for idx, label in enumerate(labels)}
        left = [INVALID_GROUP] * sample_count
        right = [INVALID_GROUP] * sample_count
        for (sample_idx, side), pop_id in map_subpop.items():
            if sample_idx >= sample_count:
                continue
            group_idx = label_to_index.get(pop_id)
            if group_idx is not None:
                if side == LEFT:
                    left[sample_idx] = group_idx
                else:
                    right[sample_idx] = group_idx
        pair_keys = []
        for i in range(len(labels)):
            for j in range(i + 1, len(labels)):
                pair_keys.append((i, j, f"{labels[i]}_vs_{labels[j]}"))
        return SubpopulationMembership(left, right, labels, pair_keys)

    def group_count(self) -> int:
        return len(self.labels)


def normalize_sample_name_for_lookup(name: str) -> str:  # process.rs:1192-1196
    if name.endswith("_L"):
        return name[:-2]
    if name.endswith("_R"):
        return name[:-2]
    return name


def map_sample_names_to_indices(sample_names: Sequence[str]) -> Dict[str, int]:  # process.rs:1198-1241
    exact_map: Dict[str, int] = {}
    alias: Dict[str, Optional[int]] = {}
    for i, name in enumerate(sample_names):
        exact_map[name] = i
        suffix = name.rsplit("_", 1)[-1]
        if suffix != name:
            if suffix not in alias:
                alias[suffix] = i
            elif alias[suffix] != i:
                alias[suffix] = None
    for a, target in alias.items():
        if target is not None and a not in exact_map:
            exact_map[a] = target
    return exact_map


def map_samples_to_haplotype_groups(sample_names, sample_to_group_map) -> Dict[Tuple[int, int], str]:
    # stats.rs:1036-1052
    out: Dict[Tuple[int, int], str] = {}
    idx_of = map_sample_names_to_indices(sample_names)
    for config_name, (left_group, right_group) in sample_to_group_map.items():
        lookup = normalize_sample_name_for_lookup(config_name)
        vcf_idx = idx_of.get(lookup)
        if vcf_idx is not None:
            out[(vcf_idx, LEFT)] = str(left_group)
            out[(vcf_idx, RIGHT)] = str(right_group)
    return out


# ----------------------------------------------------------------------------------------------
# Dense population summary (stats.rs:1311-1542, 1665-1709)
# ----------------------------------------------------------------------------------------------


def dense_sum_alt_no_missing(data, base, offsets) -> int:  # stats.rs:1665-1674
    s = 0
    for off in offsets:
        s += data[base + off]
    return s


def dense_sum_alt_with_missing(data, base, offsets, bits) -> Tuple[int, int]:  # stats.rs:1677-1697
    alt = 0
    total = 0
    for off in offsets:
        idx = base + off
        if dense_missing(bits, idx):
            continue
        alt += data[idx]
        total += 1
    return total, alt


def dense_pi_from_counts(total_called: int, alt_count: int) -> Optional[float]:  # stats.rs:1700-1709
    if total_called < 2:
        return None
    n = float(total_called)
    alt = float(alt_count)
    ref_count = float(total_called - alt_count)
    sum_sq = ref_count * ref_count + alt * alt
    return n / (n - 1.0) * (1.0 - sum_sq / (n * n))


@dataclass
class DensePopulationSummary:  # stats.rs:1311-1317
    alt_counts: List[int]
    called_counts: List[int]
    haplotype_capacity: int
    segregating_sites: int
    pi_sum: float


def build_dense_population_summary(matrix: DenseGenotypeMatrix, haplotypes) -> DensePopulationSummary:
    # stats.rs:1367-1470 (serial arm; the rayon arm differs only in f64 summation order)
    offsets = dense_membership_offsets(matrix, haplotypes)
    S = matrix.variant_count
    stride = matrix.stride
    alt_counts = [0] * S
    called_counts = [0] * S
    seg = 0
    pi_total = 0.0
    if matrix.missing is not None:
        for vi in range(S):
            called, alt = dense_sum_alt_with_missing(matrix.data, vi * stride, offsets, matrix.missing)
            alt_counts[vi] = alt
            called_counts[vi] = called
            if called >= 2 and alt > 0 and alt < called:
                seg += 1
            v = dense_pi_from_counts(called, alt)
            if v is not None:
                pi_total += v
    else:
        total = len(offsets)
        for vi in range(S):
            alt = dense_sum_alt_no_missing(matrix.data, vi * stride, offsets)
            alt_counts[vi] = alt
            called_counts[vi] = total
            if alt > 0 and alt < total:
                seg += 1
            v = dense_pi_from_counts(total, alt)
            if v is not None:
                pi_total += v
    return DensePopulationSummary(alt_counts, called_counts, len(offsets), seg, pi_total)


def saturating_sub_i64(a: int, b: int) -> int:
    r = a - b
    lo, hi = -(1 << 63), (1 << 63) - 1
    return lo if r < lo else hi if r > hi else r


def calculate_pi_from_summary(summary: DensePopulationSummary, seq_length: int, precomputed: Optional[float] = None) -> float:
    # stats.rs:1476-1542
    if summary.haplotype_capacity <= 1:
        return NAN
    if seq_length < 0:
        return 0.0
    if seq_length == 0:
        return INF
    uncallable = sum(1 for c in summary.called_counts if c < 2)
    effective_length = saturating_sub_i64(seq_length, uncallable)
    if effective_length == 0:
        return NAN
    sum_pi = precomputed if precomputed is not None else summary.pi_sum
    return sum_pi / float(effective_length)


@dataclass
class HudsonSummaryTotals:  # stats.rs:1545-1552
    numerator_sum: float = 0.0
    denominator_sum: float = 0.0
    pi1_sum: float = 0.0
    pi2_sum: float = 0.0
    dxy_sum_all: float = 0.0
    dxy_uncallable_sites: int = 0


def aggregate_hudson_components_from_summaries(pop1: DensePopulationSummary, pop2: DensePopulationSummary) -> HudsonSummaryTotals:
    # stats.rs:1554-1623
    ln = min(len(pop1.alt_counts), len(pop2.alt_counts))
    t = HudsonSummaryTotals()
    for idx in range(ln):
        n1 = pop1.called_counts[idx]
        n2 = pop2.called_counts[idx]
        if n1 == 0 or n2 == 0:
            t.dxy_uncallable_sites += 1
            continue
        a1 = pop1.alt_counts[idx]
        a2 = pop2.alt_counts[idx]
        r1 = n1 - a1
        r2 = n2 - a2
        denom_pairs = float(n1 * n2)
        if denom_pairs == 0.0:
            continue
        dxy = float(a1 * r2 + r1 * a2) / denom_pairs
        if dxy < 0.0:
            dxy = 0.0
        elif dxy > 1.0:
            dxy = 1.0
        t.dxy_sum_all += dxy
        if n1 < 2 or n2 < 2:
            continue
        denom1 = float(n1 * (n1 - 1))
        denom2 = float(n2 * (n2 - 1))
        pi1 = 2.0 * float(a1) * float(r1) / denom1 if denom1 > 0.0 else 0.0
        pi2 = 2.0 * float(a2) * float(r2) / denom2 if denom2 > 0.0 else 0.0
        t.pi1_sum += pi1
        t.pi2_sum += pi2
        if dxy > FST_EPSILON:
            t.numerator_sum += dxy - 0.5 * (pi1 + pi2)
            t.denominator_sum += dxy
    return t


def dxy_from_summaries(pop1, pop2, sequence_length: int) -> Optional[float]:  # stats.rs:1637-1662
    if sequence_length <= 0:
        return None
    totals = aggregate_hudson_components_from_summaries(pop1, pop2)
    effective_length = saturating_sub_i64(sequence_length, totals.dxy_uncallable_sites)
    if effective_length > 0:
        return totals.dxy_sum_all / float(effective_length)
    return None


def dense_dxy_from_biallelic_counts(n1, alt1, n2, alt2) -> Optional[float]:  # stats.rs:1712-1733
    if n1 == 0 or n2 == 0:
        return None
    n1_f = float(n1)
    n2_f = float(n2)
    alt1_f = float(alt1) / n1_f
    alt2_f = float(alt2) / n2_f
    ref1 = 1.0 - alt1_f
    ref2 = 1.0 - alt2_f
    dot = ref1 * ref2 + alt1_f * alt2_f
    if dot < 0.0:
        dot = 0.0
    dxy = 1.0 - dot
    if dxy < 0.0:
        dxy = 0.0
    elif dxy > 1.0:
        dxy = 1.0
    return dxy


def fst_components(dxy: Optional[float], pi1: Optional[float], pi2: Optional[float]):
    """dense_fst_components_from_biallelic (stats.rs:1736-1757) == the match in
    hudson_site_from_variant (2984-3001) == dense general (3143-3158)."""
    if dxy is not None and pi1 is not None and pi2 is not None:
        if dxy > FST_EPSILON:
            num = dxy - 0.5 * (pi1 + pi2)
            return num / dxy, num, dxy
        pi_avg = 0.5 * (pi1 + pi2)
        if abs(pi_avg) <= FST_EPSILON:
            return None, 0.0, 0.0
        return None, None, None
    return None, None, None


# ----------------------------------------------------------------------------------------------
# Weir & Cockerham (stats.rs:1781-2374)
# ----------------------------------------------------------------------------------------------


@dataclass
class FstEstimate:  # stats.rs:37-126 (+ lib.rs:76-165 state strings)
    state: str  # calculable | components_yield_indeterminate_ratio | no_inter_population_variance | insufficient_data_for_estimation
    value: Optional[float]
    sum_a: float
    sum_b: float
    sites: int


def _div(a: float, b: float) -> float:
    """IEEE-754 division (Rust f64 `/`): never raises."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return NAN
        neg = (a < 0.0) != (math.copysign(1.0, b) < 0.0)
        return -INF if neg else INF


def classify_fst(a: float, b: float, sites: int) -> FstEstimate:
    """fst_estimate_from_components (stats.rs:1781-1812) and the identical regional ladder
    (2234-2270, 2294-2328)."""
    denominator = a + b
    eps = FST_EPSILON
    if denominator > eps:
        return FstEstimate("calculable", _div(a, denominator), a, b, sites)
    if denominator < -eps:
        return FstEstimate("components_yield_indeterminate_ratio", None, a, b, sites)
    if abs(a) > eps:
        return FstEstimate("calculable", _div(a, denominator), a, b, sites)
    return FstEstimate("no_inter_population_variance", None, a, b, sites)


def insufficient(sites: int) -> FstEstimate:
    return FstEstimate("insufficient_data_for_estimation", None, 0.0, 0.0, sites)


def calculate_variance_components(pop_stats: Sequence[Tuple[int, float]], global_freq: float) -> Tuple[float, float]:
    # stats.rs:2034-2127; pop_stats = [(n_i, p_i)]
    r = float(len(pop_stats))
    if r < 2.0:
        return 0.0, 0.0
    n_values = [float(n) for n, _ in pop_stats]
    total_haplotypes = sum(n for n, _ in pop_stats)
    n_bar = float(total_haplotypes) / r
    if (n_bar - 1.0) < 1e-9:
        return 0.0, 0.0
    global_p = global_freq
    sum_sq_diff_n = 0.0
    for n_i in n_values:
        diff = n_i - n_bar
        sum_sq_diff_n += diff * diff
    c_squared = sum_sq_diff_n / (r * n_bar * n_bar) if (r > 0.0 and n_bar > 0.0) else 0.0
    numerator_s_squared = 0.0
    for n, p in pop_stats:
        diff_p = p - global_p
        numerator_s_squared += float(n) * diff_p * diff_p
    if (r - 1.0) > 1e-9 and n_bar > 1e-9:
        s_squared = numerator_s_squared / ((r - 1.0) * n_bar)
    else:
        s_squared = 0.0
    x_wc = global_p * (1.0 - global_p) - ((r - 1.0) / r) * s_squared
    a_numerator_term = s_squared - (x_wc / (n_bar - 1.0))
    a_denominator_factor = 1.0 - (c_squared / (r - 1.0))
    a = _div(a_numerator_term, a_denominator_factor)
    b = (n_bar / (n_bar - 1.0)) * x_wc
    return a, b


@dataclass
class SiteFstWc:  # stats.rs:194-215
    position: int
    overall_fst: FstEstimate
    pairwise_fst: Dict[str, FstEstimate]
    variance_components: Tuple[float, float]
    population_sizes: Dict[str, int]
    pairwise_variance_components: Dict[str, Tuple[float, float]]


def calculate_fst_wc_at_site_with_membership(variant: Variant, membership: SubpopulationMembership):
    # stats.rs:1814-2032
    alleles_present = set()
    for g in variant.genotypes:
        if g is not None:
            for a in g:
                alleles_present.add(a)
    unique_alleles = sorted(alleles_present)
    G = membership.group_count()
    sum_site_a = 0.0
    sum_site_b = 0.0
    sum_pairwise: Dict[str, Tuple[float, float]] = {}
    pop_sizes: Dict[str, int] = {}
    pop_sizes_populated = False
    for target in unique_alleles:
        total_counts = [0] * G
        alt_counts = [0] * G
        for sample_idx, g in enumerate(variant.genotypes):
            if g is None:
                continue
            if len(g) > 0:
                group = membership.left[sample_idx] if sample_idx < len(membership.left) else INVALID_GROUP
                if group != INVALID_GROUP:
                    total_counts[group] += 1
                    if g[0] == target:
                        alt_counts[group] += 1
            if len(g) > 1:
                group = membership.right[sample_idx] if sample_idx < len(membership.right) else INVALID_GROUP
                if group != INVALID_GROUP:
                    total_counts[group] += 1
                    if g[1] == target:
                        alt_counts[group] += 1
        total_called = 0
        total_target = 0
        valid_groups = 0
        stats = []
        for idx in range(G):
            total = total_counts[idx]
            if total == 0:
                continue
            valid_groups += 1
            tc = alt_counts[idx]
            total_called += total
            total_target += tc
            stats.append((total, float(tc) / float(total)))
            if not pop_sizes_populated:
                pop_sizes[membership.labels[idx]] = total
        pop_sizes_populated = True
        if valid_groups < 2:
            continue
        global_freq = float(total_target) / float(total_called) if total_called > 0 else 0.0
        comp_a, comp_b = calculate_variance_components(stats, global_freq)
        sum_site_a += comp_a
        sum_site_b += comp_b
        for ia, ib, key in membership.pair_keys:
            ta = total_counts[ia]
            tb = total_counts[ib]
            if ta == 0 or tb == 0:
                continue
            xa = alt_counts[ia]
            xb = alt_counts[ib]
            fa = float(xa) / float(ta)
            fb = float(xb) / float(tb)
            pair_total = ta + tb
            pair_global = float(xa + xb) / float(pair_total) if pair_total > 0 else 0.0
            pw_a, pw_b = calculate_variance_components([(ta, fa), (tb, fb)], pair_global)
            e = sum_pairwise.get(key, (0.0, 0.0))
            sum_pairwise[key] = (e[0] + pw_a, e[1] + pw_b)
    if not pop_sizes_populated:
        return insufficient(1), {}, (0.0, 0.0), pop_sizes, {}
    overall = classify_fst(sum_site_a, sum_site_b, 1)
    pairwise_est: Dict[str, FstEstimate] = {}
    for _, _, key in membership.pair_keys:
        if key in sum_pairwise:
            pw_a, pw_b = sum_pairwise[key]
            pairwise_est[key] = classify_fst(pw_a, pw_b, 1)
        else:
            pairwise_est[key] = insufficient(1)
            sum_pairwise[key] = (0.0, 0.0)
    return overall, pairwise_est, (sum_site_a, sum_site_b), pop_sizes, sum_pairwise


def calculate_overall_fst_wc(site_fst_values: Sequence[SiteFstWc]):
    # stats.rs:2145-2374
    if not site_fst_values:
        return insufficient(0), {}, {}
    num_insufficient = 0
    overall_components: List[Tuple[float, float]] = []
    pairwise_components: Dict[str, List[Tuple[float, float]]] = {}
    all_keys = set()
    for site in site_fst_values:
        if site.overall_fst.state == "insufficient_data_for_estimation":
            num_insufficient += 1
        else:
            overall_components.append(site.variance_components)
        for key, (a_xy, b_xy) in site.pairwise_variance_components.items():
            all_keys.add(key)
            est = site.pairwise_fst.get(key)
            if est is not None and est.state != "insufficient_data_for_estimation":
                pairwise_components.setdefault(key, []).append((a_xy, b_xy))
    total_attempted = len(site_fst_values)
    contributing = total_attempted - num_insufficient
    if contributing == 0:
        overall = insufficient(total_attempted)
    else:
        sum_a = 0.0
        for a, _ in overall_components:
            sum_a += a
        sum_b = 0.0
        for _, b in overall_components:
            sum_b += b
        overall = classify_fst(sum_a, sum_b, len(overall_components))
    pairwise_est: Dict[str, FstEstimate] = {}
    agg: Dict[str, Tuple[float, float]] = {}
    for key in sorted(all_keys):
        comps = pairwise_components.get(key)
        if comps is not None:
            sa = 0.0
            for a, _ in comps:
                sa += a
            sb = 0.0
            for _, b in comps:
                sb += b
            agg[key] = (sa, sb)
            pairwise_est[key] = classify_fst(sa, sb, len(comps))
        else:
            attempted = sum(
                1 for s in site_fst_values if key in s.pairwise_variance_components or key in s.pairwise_fst
            )
            pairwise_est[key] = insufficient(attempted)
            agg[key] = (0.0, 0.0)
    return overall, pairwise_est, agg


@dataclass
class FstWcResults:  # stats.rs:559-579
    overall_fst: FstEstimate
    pairwise_fst: Dict[str, FstEstimate]
    pairwise_variance_components: Dict[str, Tuple[float, float]]
    site_fst: List[SiteFstWc]
    fst_type: str


def calculate_fst_wc_haplotype_groups(variants, sample_names, sample_to_group_map, region: QueryRegion) -> FstWcResults:
    # stats.rs:675-806
    hap_to_group = map_samples_to_haplotype_groups(sample_names, sample_to_group_map)
    membership = SubpopulationMembership.from_map(len(sample_names), hap_to_group)
    sites = []
    for variant in variants:
        if not region.contains(variant.position):
            continue
        overall, pw, comps, sizes, pw_comps = calculate_fst_wc_at_site_with_membership(variant, membership)
        sites.append(SiteFstWc(variant.position + 1, overall, pw, comps, sizes, pw_comps))
    overall, pw, agg = calculate_overall_fst_wc(sites)
    return FstWcResults(overall, pw, agg, sites, "haplotype_groups")


def parse_population_csv(csv_path: str) -> Dict[str, List[str]]:
    """stats.rs:951-1007: population label, then its sample IDs; '#' comments and blank lines skipped."""
    population_map: Dict[str, List[str]] = {}
    with open(csv_path) as fh:
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if line.strip() == "" or line.startswith("#"):
                continue
            parts = [x.strip() for x in line.split(",")]
            if not parts or parts[0] == "":
                continue
            samples = [x for x in parts[1:] if x != ""]
            if samples:
                population_map[parts[0]] = samples
    if not population_map:
        raise VcfError("Parse", f"Population CSV file '{csv_path}' contains no valid population data after parsing.")
    return population_map


def map_samples_to_populations(sample_names, population_assignments) -> Dict[Tuple[int, int], str]:
    # stats.rs:1054-1078
    out: Dict[Tuple[int, int], str] = {}
    idx_of = map_sample_names_to_indices(sample_names)
    csv_sample_to_pop: Dict[str, str] = {}
    for pop_name, samples in population_assignments.items():
        for sid in samples:
            csv_sample_to_pop[sid] = pop_name
    for csv_name, pop_name in csv_sample_to_pop.items():
        vcf_idx = idx_of.get(normalize_sample_name_for_lookup(csv_name))
        if vcf_idx is not None:
            out[(vcf_idx, LEFT)] = pop_name
            out[(vcf_idx, RIGHT)] = pop_name
    return out


def calculate_fst_wc_csv_populations(variants, sample_names, csv_path: str, region: QueryRegion) -> FstWcResults:
    # stats.rs:816-934
    assignments = parse_population_csv(csv_path)
    membership = SubpopulationMembership.from_map(len(sample_names), map_samples_to_populations(sample_names, assignments))
    sites = []
    for variant in variants:
        if not region.contains(variant.position):
            continue
        overall, pw, comps, sizes, pw_comps = calculate_fst_wc_at_site_with_membership(variant, membership)
        sites.append(SiteFstWc(variant.position + 1, overall, pw, comps, sizes, pw_comps))
    overall, pw, agg = calculate_overall_fst_wc(sites)
    return FstWcResults(overall, pw, agg, sites, "population_groups")


def extract_wc_fst_components(e: FstEstimate):  # stats.rs:4860-4914
    return e.value, e.sum_a, e.sum_b, e.sites


# ----------------------------------------------------------------------------------------------
# Sparse allele summaries, pi, Dxy, Hudson per-site (stats.rs:2403-3058)
# ----------------------------------------------------------------------------------------------


@dataclass
class AlleleCountSummary:  # stats.rs:2632-2681
    total_called: int = 0
    sum_counts_sq: float = 0.0
    counts: List[List[int]] = field(default_factory=list)  # sorted [allele, count]

    def record(self, allele: int) -> None:
        self.total_called += 1
        lo, hi = 0, len(self.counts)
        while lo < hi:
            mid = (lo + hi) // 2
            if self.counts[mid][0] < allele:
                lo = mid + 1
            else:
                hi = mid
        if lo < len(self.counts) and self.counts[lo][0] == allele:
            entry = self.counts[lo]
            self.sum_counts_sq += float(2 * entry[1] + 1)
            entry[1] += 1
        else:
            self.counts.insert(lo, [allele, 1])
            self.sum_counts_sq += 1.0


def freq_summary_for_pop(variant: Variant, membership: HapMembership) -> AlleleCountSummary:  # stats.rs:2683-2701
    summary = AlleleCountSummary()
    for idx, g in enumerate(variant.genotypes):
        if g is None:
            continue
        if idx < len(membership.left) and membership.left[idx]:
            if len(g) > 0:
                summary.record(g[0])
        if idx < len(membership.right) and membership.right[idx]:
            if len(g) > 1:
                summary.record(g[1])
    return summary


def pi_from_components(total_called: int, sum_counts_sq: float) -> Optional[float]:  # stats.rs:2723-2733
    if total_called < 2:
        return None
    n = float(total_called)
    inv_n = 1.0 / n
    sum_p2 = sum_counts_sq * inv_n * inv_n
    return n / (n - 1.0) * (1.0 - sum_p2)


def compute_pi_metrics_fast(variant: Variant, membership: HapMembership):
    # stats.rs:2761-2821 -> (total_called, sum_counts_sq, distinct_alleles)
    counts: Dict[int, int] = {}
    total_called = 0
    for sample_index, g in enumerate(variant.genotypes):
        if g is None:
            continue
        if sample_index < len(membership.left) and membership.left[sample_index]:
            if len(g) > 0:
                counts[g[0]] = counts.get(g[0], 0) + 1
                total_called += 1
        if sample_index < len(membership.right) and membership.right[sample_index]:
            if len(g) > 1:
                counts[g[1]] = counts.get(g[1], 0) + 1
                total_called += 1
    sum_counts_sq = 0.0
    for c in counts.values():  # insertion order == used_indices order
        sum_counts_sq += float(c) * float(c)
    return total_called, sum_counts_sq, len(counts)


def dxy_from_counts(c1: AlleleCountSummary, c2: AlleleCountSummary) -> Optional[float]:  # stats.rs:2907-2935
    n1 = c1.total_called
    n2 = c2.total_called
    if n1 == 0 or n2 == 0:
        return None
    dot = 0.0
    i = j = 0
    inv1 = 1.0 / float(n1)
    inv2 = 1.0 / float(n2)
    e1, e2 = c1.counts, c2.counts
    while i < len(e1) and j < len(e2):
        if e1[i][0] < e2[j][0]:
            i += 1
        elif e1[i][0] > e2[j][0]:
            j += 1
        else:
            dot += (float(e1[i][1]) * inv1) * (float(e2[j][1]) * inv2)
            i += 1
            j += 1
    dxy = 1.0 - dot
    return min(max(dxy, 0.0), 1.0)


@dataclass
class SiteFstHudson:  # stats.rs:536-555
    position: int
    fst: Optional[float]
    d_xy: Optional[float]
    pi_pop1: Optional[float]
    pi_pop2: Optional[float]
    n1_called: int
    n2_called: int
    num_component: Optional[float]
    den_component: Optional[float]


def hudson_site_from_variant(variant: Variant, pop1_mem: HapMembership, pop2_mem: HapMembership) -> SiteFstHudson:
    # stats.rs:2969-3014
    c1 = freq_summary_for_pop(variant, pop1_mem)
    c2 = freq_summary_for_pop(variant, pop2_mem)
    pi1 = pi_from_components(c1.total_called, c1.sum_counts_sq)
    pi2 = pi_from_components(c2.total_called, c2.sum_counts_sq)
    dxy = dxy_from_counts(c1, c2)
    fst, num_c, den_c = fst_components(dxy, pi1, pi2)
    return SiteFstHudson(variant.position + 1, fst, dxy, pi1, pi2, c1.total_called, c2.total_called, num_c, den_c)


@dataclass
class PopulationContext:  # stats.rs:231-247
    id: object
    haplotypes: List[Tuple[int, int]]
    variants: Sequence[Variant]
    sample_names: Sequence[str]
    sequence_length: int
    dense_genotypes: Optional[DenseGenotypeMatrix] = None
    dense_summary: Optional[DensePopulationSummary] = None


def variants_compatible(a: Sequence[Variant], b: Sequence[Variant]) -> bool:  # stats.rs:3399-3401
    return len(a) == len(b) and all(x.position == y.position for x, y in zip(a, b))


def calculate_hudson_fst_per_site(pop1: PopulationContext, pop2: PopulationContext, region: QueryRegion) -> List[SiteFstHudson]:
    # stats.rs:3021-3058
    if not variants_compatible(pop1.variants, pop2.variants):
        return []
    m1 = HapMembership.build(len(pop1.sample_names), pop1.haplotypes)
    m2 = HapMembership.build(len(pop2.sample_names), pop2.haplotypes)
    return [hudson_site_from_variant(v, m1, m2) for v in pop1.variants if region.contains(v.position)]


def dense_collect_counts(matrix: DenseGenotypeMatrix, offsets, variant_idx):
    """stats.rs:2823-2880 -> (total_called, sum_counts_sq, counts dict in first-seen order)."""
    base = variant_idx * matrix.stride
    counts: Dict[int, int] = {}
    if matrix.missing is not None:
        called = 0
        for off in offsets:
            idx = base + off
            if dense_missing(matrix.missing, idx):
                continue
            a = matrix.data[idx]
            counts[a] = counts.get(a, 0) + 1
            called += 1
        total_called = called
    else:
        for off in offsets:
            a = matrix.data[base + off]
            counts[a] = counts.get(a, 0) + 1
        total_called = len(offsets)
    sum_counts_sq = 0.0
    for c in counts.values():
        sum_counts_sq += float(c) * float(c)
    return total_called, sum_counts_sq, counts


def _dense_dot(counts1, counts2, n1, n2) -> float:
    # shared by stats.rs:2557-2590 and 3106-3139
    inv1 = 1.0 / float(n1)
    inv2 = 1.0 / float(n2)
    dot = 0.0
    if len(counts1) <= len(counts2):
        for allele, c1 in counts1.items():
            if c1 == 0:
                continue
            c2 = counts2.get(allele, 0)
            if c2 != 0:
                dot += (float(c1) * inv1) * (float(c2) * inv2)
    else:
        for allele, c2 in counts2.items():
            if c2 == 0:
                continue
            c1 = counts1.get(allele, 0)
            if c1 != 0:
                dot += (float(c1) * inv1) * (float(c2) * inv2)
    return dot


def dense_hudson_sites_general(matrix, variants, off1, off2) -> List[SiteFstHudson]:  # stats.rs:3072-3177
    sites = []
    for vi, variant in enumerate(variants):
        n1, ss1, c1 = dense_collect_counts(matrix, off1, vi)
        n2, ss2, c2 = dense_collect_counts(matrix, off2, vi)
        pi1 = (float(n1) / (float(n1) - 1.0) * (1.0 - ss1 / (float(n1) * float(n1)))) if n1 >= 2 else None
        pi2 = (float(n2) / (float(n2) - 1.0) * (1.0 - ss2 / (float(n2) * float(n2)))) if n2 >= 2 else None
        if n1 == 0 or n2 == 0:
            dxy = None
        else:
            dxy = min(max(1.0 - _dense_dot(c1, c2, n1, n2), 0.0), 1.0)
        fst, num_c, den_c = fst_components(dxy, pi1, pi2)
        sites.append(SiteFstHudson(variant.position + 1, fst, dxy, pi1, pi2, n1, n2, num_c, den_c))
    return sites


def dense_hudson_sites_biallelic(matrix, variants, off1, off2) -> List[SiteFstHudson]:  # stats.rs:3179-3278
    sites = []
    stride = matrix.stride
    if matrix.missing is not None:
        for vi, variant in enumerate(variants):
            base = vi * stride
            n1, a1 = dense_sum_alt_with_missing(matrix.data, base, off1, matrix.missing)
            n2, a2 = dense_sum_alt_with_missing(matrix.data, base, off2, matrix.missing)
            pi1 = dense_pi_from_counts(n1, a1)
            pi2 = dense_pi_from_counts(n2, a2)
            dxy = dense_dxy_from_biallelic_counts(n1, a1, n2, a2)
            fst, num_c, den_c = fst_components(dxy, pi1, pi2)
            sites.append(SiteFstHudson(variant.position + 1, fst, dxy, pi1, pi2, n1, n2, num_c, den_c))
    else:
        n1t, n2t = len(off1), len(off2)
        n1f, n2f = float(n1t), float(n2t)
        scale1 = (n1f / (n1f - 1.0), 1.0 / (n1f * n1f)) if n1t >= 2 else None
        scale2 = (n2f / (n2f - 1.0), 1.0 / (n2f * n2f)) if n2t >= 2 else None

        def pi_scaled(scale, alt, total):
            if scale is None:
                return None
            if alt == 0 or alt == total:
                return 0.0
            alt_f = float(alt)
            ref_f = float(total - alt)
            return scale[0] * (1.0 - (ref_f * ref_f + alt_f * alt_f) * scale[1])

        for vi, variant in enumerate(variants):
            base = vi * stride
            a1 = dense_sum_alt_no_missing(matrix.data, base, off1)
            a2 = dense_sum_alt_no_missing(matrix.data, base, off2)
            pi1 = pi_scaled(scale1, a1, n1t)
            pi2 = pi_scaled(scale2, a2, n2t)
            dxy = dense_dxy_from_biallelic_counts(n1t, a1, n2t, a2)
            fst, num_c, den_c = fst_components(dxy, pi1, pi2)
            sites.append(SiteFstHudson(variant.position + 1, fst, dxy, pi1, pi2, n1t, n2t, num_c, den_c))
    return sites


def dense_hudson_sites(matrix, variants, off1, off2):  # stats.rs:3060-3070
    if matrix.max_allele <= 1:
        return dense_hudson_sites_biallelic(matrix, variants, off1, off2)
    return dense_hudson_sites_general(matrix, variants, off1, off2)


def hudson_component_sums(sites: Sequence[SiteFstHudson]) -> Tuple[float, float]:  # stats.rs:1625-1635
    num_sum = 0.0
    den_sum = 0.0
    for s in sites:
        if s.num_component is not None and s.den_component is not None:
            num_sum += s.num_component
            den_sum += s.den_component
    return num_sum, den_sum


def aggregate_hudson_from_sites(sites) -> Optional[float]:  # stats.rs:3309-3316
    num_sum, den_sum = hudson_component_sums(sites)
    return num_sum / den_sum if den_sum > FST_EPSILON else None


def calculate_dxy_dense(matrix, off1, off2, sequence_length: int) -> Optional[float]:  # stats.rs:2526-2611
    if len(off1) == 0 or len(off2) == 0:
        return None
    if sequence_length <= 0:
        return None
    sum_dxy = 0.0
    skipped = 0
    for vi in range(matrix.variant_count):
        n1, _, c1 = dense_collect_counts(matrix, off1, vi)
        n2, _, c2 = dense_collect_counts(matrix, off2, vi)
        if n1 == 0 or n2 == 0:
            skipped += 1
            continue
        sum_dxy += min(max(1.0 - _dense_dot(c1, c2, n1, n2), 0.0), 1.0)
    effective = saturating_sub_i64(sequence_length, skipped)
    return sum_dxy / float(effective) if effective > 0 else None


class VcfError(Exception):  # process.rs:632-640; Debug-formatted like lib.rs:1551
    def __init__(self, kind: str, msg: str):
        super().__init__(f'{kind}("{msg}")')
        self.kind = kind
        self.msg = msg


def calculate_d_xy_hudson(pop1: PopulationContext, pop2: PopulationContext) -> Optional[float]:
    # stats.rs:2403-2524 -> DxyHudsonResult.d_xy
    if pop1.sequence_length <= 0:
        raise VcfError("InvalidRegion", "Sequence length must be positive for Dxy calculation")
    if pop1.sequence_length != pop2.sequence_length:
        raise VcfError("Parse", "Sequence length mismatch in Dxy calculation")
    if not variants_compatible(pop1.variants, pop2.variants):
        raise VcfError("Parse", "Variant slices differ in positions/length for Dxy calculation")
    if not pop1.haplotypes or not pop2.haplotypes:
        return None
    if pop1.dense_summary is not None and pop2.dense_summary is not None:
        return dxy_from_summaries(pop1.dense_summary, pop2.dense_summary, pop1.sequence_length)
    m1, m2 = pop1.dense_genotypes, pop2.dense_genotypes
    if m1 is not None and m2 is not None and m1 is m2 and m1.ploidy == 2:
        return calculate_dxy_dense(
            m1, dense_membership_offsets(m1, pop1.haplotypes), dense_membership_offsets(m2, pop2.haplotypes), pop1.sequence_length
        )
    mem1 = HapMembership.build(len(pop1.sample_names), pop1.haplotypes)
    mem2 = HapMembership.build(len(pop2.sample_names), pop2.haplotypes)
    sum_dxy = 0.0
    skipped = 0
    for variant in pop1.variants:
        d = dxy_from_counts(freq_summary_for_pop(variant, mem1), freq_summary_for_pop(variant, mem2))
        if d is not None:
            sum_dxy += d
        else:
            skipped += 1
    effective = saturating_sub_i64(pop1.sequence_length, skipped)
    return sum_dxy / float(effective) if effective > 0 else None


@dataclass
class HudsonFSTOutcome:  # stats.rs:515-532
    pop1_id: object = None
    pop2_id: object = None
    fst: Optional[float] = None
    d_xy: Optional[float] = None
    pi_pop1: Optional[float] = None
    pi_pop2: Optional[float] = None
    pi_xy_avg: Optional[float] = None


def calculate_hudson_fst_for_pair_core(pop1: PopulationContext, pop2: PopulationContext, region: Optional[QueryRegion]):
    # stats.rs:3435-3599
    if pop1.sequence_length <= 0:
        raise VcfError("InvalidRegion", "Sequence length must be positive for Hudson FST calculation.")
    if pop1.sequence_length != pop2.sequence_length:
        raise VcfError("Parse", "Sequence length mismatch between population contexts for Hudson FST calculation.")
    if not variants_compatible(pop1.variants, pop2.variants):
        raise VcfError("Parse", "Variant slices differ in positions/length.")
    summary_pair = (
        (pop1.dense_summary, pop2.dense_summary)
        if pop1.dense_summary is not None and pop2.dense_summary is not None
        else None
    )
    summary_totals = None
    m1, m2 = pop1.dense_genotypes, pop2.dense_genotypes
    dense_shared = m1 if (m1 is not None and m2 is not None and m1 is m2 and m1.ploidy == 2) else None
    site_values: List[SiteFstHudson] = []
    if region is not None:
        site_values = calculate_hudson_fst_per_site(pop1, pop2, region)
        num_sum, den_sum = hudson_component_sums(site_values)
    elif summary_pair is not None:
        summary_totals = aggregate_hudson_components_from_summaries(*summary_pair)
        num_sum, den_sum = summary_totals.numerator_sum, summary_totals.denominator_sum
    elif dense_shared is not None:
        if not pop1.variants:
            num_sum, den_sum = 0.0, 0.0
        else:
            o1 = dense_membership_offsets(dense_shared, pop1.haplotypes)
            o2 = dense_membership_offsets(dense_shared, pop2.haplotypes)
            site_values = dense_hudson_sites(dense_shared, pop1.variants, o1, o2)
            num_sum, den_sum = hudson_component_sums(site_values)
    elif not pop1.variants:
        num_sum, den_sum = 0.0, 0.0
    else:
        mem1 = HapMembership.build(len(pop1.sample_names), pop1.haplotypes)
        mem2 = HapMembership.build(len(pop2.sample_names), pop2.haplotypes)
        site_values = [hudson_site_from_variant(v, mem1, mem2) for v in pop1.variants]
        num_sum, den_sum = hudson_component_sums(site_values)
    regional_fst = num_sum / den_sum if den_sum > FST_EPSILON else None
    if summary_totals is not None:
        pi1_raw = calculate_pi_from_summary(summary_pair[0], pop1.sequence_length, summary_totals.pi1_sum)
        pi2_raw = calculate_pi_from_summary(summary_pair[1], pop2.sequence_length, summary_totals.pi2_sum)
        if not pop1.haplotypes or not pop2.haplotypes:
            dxy_value = None
        else:
            effective = saturating_sub_i64(pop1.sequence_length, summary_totals.dxy_uncallable_sites)
            dxy_value = summary_totals.dxy_sum_all / float(effective) if effective > 0 else None
    else:
        pi1_raw = calculate_pi_for_population(pop1)
        pi2_raw = calculate_pi_for_population(pop2)
        dxy_value = calculate_d_xy_hudson(pop1, pop2)
    pi1_opt = pi1_raw if math.isfinite(pi1_raw) else None
    pi2_opt = pi2_raw if math.isfinite(pi2_raw) else None
    outcome = HudsonFSTOutcome(pop1.id, pop2.id, regional_fst, dxy_value, pi1_opt, pi2_opt, None)
    if pi1_opt is not None and pi2_opt is not None:
        outcome.pi_xy_avg = 0.5 * (pi1_opt + pi2_opt)
    return outcome, site_values


def calculate_hudson_fst_for_pair_with_sites(pop1, pop2, region):  # stats.rs:3619-3625
    return calculate_hudson_fst_for_pair_core(pop1, pop2, region)


def calculate_hudson_fst_for_pair(pop1, pop2):  # stats.rs:3636-3641
    return calculate_hudson_fst_for_pair_core(pop1, pop2, None)[0]


# ----------------------------------------------------------------------------------------------
# Adjusted length, inversion freq, segregating sites (stats.rs:3644-4084)
# ----------------------------------------------------------------------------------------------


def _hal_from_1based_inclusive(start, end):  # process.rs:193-206
    a = max(start, 1)
    b = end if end >= a else a
    return a - 1, b


def _hal_len(iv):
    return iv[1] - iv[0] if iv[1] > iv[0] else 0


def subtract_regions(intervals, masks):  # stats.rs:3739-3775
    if masks is None:
        return list(intervals)
    out = []
    for a_start, a_end in intervals:
        parts = [(a_start, a_end)]
        for m_start, m_end in masks:
            nxt = []
            for s, e in parts:
                if m_end < s or m_start > e:
                    nxt.append((s, e))
                    continue
                if m_start > s:
                    left_end = m_start - 1
                    if left_end >= s:
                        nxt.append((s, left_end))
                if m_end < e:
                    right_start = m_end + 1
                    if right_start <= e:
                        nxt.append((right_start, e))
            parts = nxt
            if not parts:
                break
        out.extend(parts)
    return out


def calculate_adjusted_sequence_length(region_start: int, region_end: int, allow=None, mask=None) -> int:
    # stats.rs:3644-3736
    region = _hal_from_1based_inclusive(region_start, region_end)
    allowed = []
    if allow is not None:
        for start, end in allow:
            s = max(region[0], start)
            e = min(region[1], end)
            if s < e:
                allowed.append((s + 1, e))  # to_1based_inclusive_tuple
    else:
        allowed.append((region_start, region_end))
    converted = None
    if mask is not None:
        converted = [(s + 1, e) for s, e in mask]
    unmasked = subtract_regions(allowed, converted)
    return sum(_hal_len(_hal_from_1based_inclusive(s, e)) for s, e in unmasked)


def calculate_inversion_allele_frequency(sample_filter: Dict[str, Tuple[int, int]]) -> Optional[float]:
    # stats.rs:3778-3805
    num_ones = 0
    total = 0
    for _, (h1, h2) in sample_filter.items():
        for allele in (h1, h2):
            if allele == 0 or allele == 1:
                total += 1
                if allele == 1:
                    num_ones += 1
    return float(num_ones) / float(total) if total > 0 else None


def variant_is_segregating(variant: Variant) -> bool:  # stats.rs:3815-3829
    first = None
    for g in variant.genotypes:
        if g is not None:
            for allele in g:
                if first is None:
                    first = allele
                elif first != allele:
                    return True
    return False


def count_segregating_sites(variants) -> int:  # stats.rs:3808-3813
    return sum(1 for v in variants if variant_is_segregating(v))


def variant_is_segregating_in_haplotypes(variant: Variant, haplotypes) -> bool:  # stats.rs:3868-3889
    first = None
    for sample_idx, side in haplotypes:
        g = variant.genotypes.get(sample_idx)
        if g is None:
            continue
        if side >= len(g):
            continue
        allele = g[side]
        if first is None:
            first = allele
        elif first != allele:
            return True
    return False


def count_segregating_sites_for_haplotypes(variants, haplotypes) -> int:  # stats.rs:3858-3866
    return sum(1 for v in variants if variant_is_segregating_in_haplotypes(v, haplotypes))


def count_segregating_sites_dense(matrix: DenseGenotypeMatrix, offsets) -> int:  # stats.rs:3891-4084
    stride = matrix.stride
    if matrix.max_allele <= 1:
        if len(offsets) < 2:
            return 0
        total = len(offsets)
        seg = 0
        for vi in range(matrix.variant_count):
            base = vi * stride
            if matrix.missing is not None:
                called, alt = dense_sum_alt_with_missing(matrix.data, base, offsets, matrix.missing)
                if called >= 2 and alt > 0 and alt < called:
                    seg += 1
            else:
                alt = dense_sum_alt_no_missing(matrix.data, base, offsets)
                if alt > 0 and alt < total:
                    seg += 1
        return seg
    if not offsets:
        return 0
    seg = 0
    for vi in range(matrix.variant_count):
        base = vi * stride
        first = None
        poly = False
        for off in offsets:
            idx = base + off
            if matrix.missing is not None and dense_missing(matrix.missing, idx):
                continue
            allele = matrix.data[idx]
            if first is None:
                first = allele
            elif allele != first:
                poly = True
                break
        if poly:
            seg += 1
    return seg


def count_segregating_sites_for_population(ctx: PopulationContext) -> int:  # stats.rs:3831-3851
    if ctx.dense_summary is not None:
        return ctx.dense_summary.segregating_sites
    m = ctx.dense_genotypes
    if m is not None and m.ploidy == 2:
        offsets = dense_membership_offsets(m, ctx.haplotypes)
        if len(offsets) <= 1:
            return 0
        return count_segregating_sites_dense(m, offsets)
    return count_segregating_sites_for_haplotypes(ctx.variants, ctx.haplotypes)


# ----------------------------------------------------------------------------------------------
# harmonic, theta, pi, per-site diversity, pairwise differences (stats.rs:4106-4806)
# ----------------------------------------------------------------------------------------------


def harmonic(n: int) -> float:  # stats.rs:4234-4240
    s = 0.0
    for k in range(1, n + 1):
        s += 1.0 / float(k)
    return s


def calculate_watterson_theta(seg_sites: int, n: int, seq_length: int) -> float:  # stats.rs:4243-4307
    if n <= 1:
        return NAN if seg_sites == 0 else INF
    if seq_length <= 0:
        return NAN if seg_sites == 0 else INF
    h = harmonic(n - 1)
    if h > 0.0:
        return float(seg_sites) / h / float(seq_length)
    return NAN if seg_sites == 0 else INF


def calculate_pi(variants, haplotypes_in_group, seq_length: int) -> float:  # stats.rs:4317-4432
    if len(haplotypes_in_group) <= 1:
        return NAN
    if seq_length < 0:
        return 0.0
    if seq_length == 0:
        return INF
    variant_sample_count = len(variants[0].genotypes) if variants else 0
    hap_sample_count = max((s + 1 for s, _ in haplotypes_in_group), default=0)
    sample_count = max(variant_sample_count, hap_sample_count)
    membership = HapMembership.build(sample_count, haplotypes_in_group)
    if membership.total <= 1:
        return NAN
    sum_pi = 0.0
    skipped = 0
    for variant in variants:
        total_called, ssq, _ = compute_pi_metrics_fast(variant, membership)
        pi_site = pi_from_components(total_called, ssq)
        if pi_site is not None:
            sum_pi += pi_site
        elif total_called < 2:
            skipped += 1
    effective = saturating_sub_i64(seq_length, skipped)
    if effective == 0:
        return NAN
    return sum_pi / float(effective)


def calculate_pi_dense(matrix: DenseGenotypeMatrix, offsets, seq_length: int) -> float:  # stats.rs:4434-4597
    if len(offsets) <= 1:
        return NAN
    if seq_length < 0:
        return 0.0
    if seq_length == 0:
        return INF
    stride = matrix.stride
    sum_pi = 0.0
    skipped = 0
    if matrix.max_allele <= 1:
        if matrix.missing is not None:
            for vi in range(matrix.variant_count):
                called, alt = dense_sum_alt_with_missing(matrix.data, vi * stride, offsets, matrix.missing)
                v = dense_pi_from_counts(called, alt)
                if v is not None:
                    sum_pi += v
                else:
                    skipped += 1
        else:
            total = len(offsets)
            n = float(total)
            scale = n / (n - 1.0)
            inv_n_sq = 1.0 / (n * n)
            for vi in range(matrix.variant_count):
                alt = dense_sum_alt_no_missing(matrix.data, vi * stride, offsets)
                if alt == 0 or alt == total:
                    continue
                alt_f = float(alt)
                ref_f = float(total - alt)
                sum_sq = ref_f * ref_f + alt_f * alt_f
                sum_pi += scale * (1.0 - sum_sq * inv_n_sq)
    else:
        for vi in range(matrix.variant_count):
            total_called, ssq, _ = dense_collect_counts(matrix, offsets, vi)
            if total_called >= 2:
                n = float(total_called)
                sum_p2 = ssq / (n * n)
                sum_pi += n / (n - 1.0) * (1.0 - sum_p2)
            else:
                skipped += 1
    effective = saturating_sub_i64(seq_length, skipped)
    if effective == 0:
        return NAN
    return sum_pi / float(effective)


def calculate_pi_for_population(ctx: PopulationContext) -> float:  # stats.rs:4599-4614
    if ctx.dense_summary is not None:
        return calculate_pi_from_summary(ctx.dense_summary, ctx.sequence_length)
    m = ctx.dense_genotypes
    if m is not None and m.ploidy == 2:
        return calculate_pi_dense(m, dense_membership_offsets(m, ctx.haplotypes), ctx.sequence_length)
    return calculate_pi(ctx.variants, ctx.haplotypes, ctx.sequence_length)


@dataclass
class SiteDiversity:  # stats.rs:185-190
    position: int
    pi: float
    watterson_theta: float


def calculate_per_site_diversity(variants, haplotypes_in_group, region: QueryRegion, filtered_positions=frozenset(), mask_intervals=None):
    # stats.rs:4628-4806
    sample_count = len(variants[0].genotypes) if variants else 0
    membership = HapMembership.build(sample_count, haplotypes_in_group)
    if region.len() <= 0:
        return []
    out: List[SiteDiversity] = []
    if len(haplotypes_in_group) < 2:
        return out
    for variant in variants:
        if not region.contains(variant.position):
            continue
        total_called, ssq, distinct = compute_pi_metrics_fast(variant, membership)
        if total_called < 2:
            pi_value, theta_value = NAN, NAN
        else:
            if distinct > 1:
                denom = harmonic(total_called - 1)
                theta_value = 1.0 / denom if denom > 0.0 else 0.0
            else:
                theta_value = 0.0
            p = pi_from_components(total_called, ssq)
            pi_value = p if p is not None else 0.0
        pos0 = variant.position
        masked = mask_intervals is not None and any(s <= pos0 < e for s, e in mask_intervals)
        if pos0 in filtered_positions or masked:
            pi_value, theta_value = NAN, NAN
        out.append(SiteDiversity(pos0 + 1, pi_value, theta_value))
    return out


def calculate_pairwise_differences(variants, number_of_samples: int, sequence_length: int):
    # stats.rs:4106-4231
    if sequence_length <= 0:
        return []
    hap_counts: List[Optional[int]] = [None] * number_of_samples
    for variant in variants:
        for idx, g in enumerate(variant.genotypes):
            if idx >= number_of_samples:
                break
            if hap_counts[idx] is None and g is not None:
                hap_counts[idx] = len(g)
        if all(c is not None for c in hap_counts):
            break
    base_sites = sequence_length
    result = []
    for i in range(number_of_samples):
        for j in range(i + 1, number_of_samples):
            hi = hap_counts[i] or 0
            hj = hap_counts[j] or 0
            if hi == 0 or hj == 0:
                result.append(((i, j), 0, 0))
                continue
            product = hi * hj
            diff = 0
            comparable = base_sites * product
            for variant in variants:
                gi = variant.genotypes.get(i)
                gj = variant.genotypes.get(j)
                if gi is not None and gj is not None:
                    for a in gi:
                        for b in gj:
                            if a != b:
                                diff += 1
                else:
                    comparable = max(comparable - product, 0)
            result.append(((i, j), diff, comparable))
    return result


# ----------------------------------------------------------------------------------------------
# lib.rs helpers the Python API layer relies on
# ----------------------------------------------------------------------------------------------


def convert_numeric_array(genotypes, positions):
    """lib.rs:1135-1227 — numpy (S,N,ploidy) array -> (variants, dense matrix or None).

    Signed dtypes: negative = missing.  Sparse marks the WHOLE sample None if any allele is
    missing (1195-1199); dense marks only that allele's bit (1180-1191).  Dense matrix only if
    ploidy == 2 (1208)."""
    import numpy as np

    g = np.asarray(genotypes)
    S, N, ploidy = g.shape
    if len(positions) != S:
        raise ValueError(f"positions length {len(positions)} does not match variant dimension {S}")
    signed = g.dtype.kind == "i"
    if g.dtype.itemsize > 1 and (g > 255).any():
        raise ValueError("allele values must be <= 255")
    variants = []
    total = S * N * ploidy
    dense = bytearray(total)
    missing_bits = None
    max_allele = 0
    lin = 0
    for vi in range(S):
        row = []
        for si in range(N):
            alleles = []
            miss = False
            for ai in range(ploidy):
                val = int(g[vi, si, ai])
                if signed and val < 0:
                    miss = True
                    if missing_bits is None:
                        missing_bits = [0] * ((total + 63) // 64)
                    missing_bits[lin // 64] |= 1 << (lin % 64)
                else:
                    alleles.append(val)
                    dense[lin] = val
                    if val > max_allele:
                        max_allele = val
                lin += 1
            row.append(None if miss else alleles)
        variants.append(make_variant(int(positions[vi]), row))
    matrix = DenseGenotypeMatrix(bytes(dense), missing_bits, S, N, ploidy, max_allele) if ploidy == 2 else None
    return variants, matrix


def population_context_like_lib(pop_id, variants, haplotypes, sample_names, sequence_length, dense=None) -> PopulationContext:
    """OwnedPopulationContext::as_population_context, lib.rs:777-799: a summary is attached iff a
    dense matrix exists and max_allele <= 1."""
    summary = None
    if dense is not None and dense.max_allele <= 1:
        summary = build_dense_population_summary(dense, haplotypes)
    return PopulationContext(pop_id, list(haplotypes), variants, sample_names, sequence_length, dense, summary)
