"""GPU parity proper: every sweep of libferromic_hip.so (called through the C-ABI) against the CPU
oracle on the same seeded inputs.  Integer outputs and per-site f64 records must be bit-exact;
regional sums within 1e-9 relative (1e-12 absolute near zero), the tolerance north_star states
(reference sums are rayon-ordered, i.e. order-dependent themselves)."""

import random

import math

import numpy as np
import pytest

from oracle import ferromic_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from ferromic_amd import device

    return device


def upload(dev, m: R.DenseGenotypeMatrix):
    return dev.DeviceMatrix.from_host(np.frombuffer(m.data, dtype=np.uint8), H.missing_words_np(m), m.variant_count,
                                      m.sample_count, m.ploidy, m.max_allele)


SHAPES = [
    # (sites, samples, max_allele, p_missing)
    (300, 37, 1, 0.0),     # ragged: H = 74 (not a multiple of 16), partial last tile
    (64, 8, 1, 0.0),       # exactly one tile, H = 16
    (1, 3, 1, 0.0),        # single site
    (257, 500, 1, 0.0),    # C2-like width
    (190, 61, 1, 0.07),    # biallelic with missing
    (130, 45, 3, 0.0),     # multi-allelic
    (130, 45, 5, 0.1),     # multi-allelic with missing
    (70, 2500, 1, 0.0),    # C4-like width (H = 5000)
    (40, 5000, 1, 0.0),    # C5 width (H = 10 000)
    (40, 5000, 1, 0.01),
    (70, 7, 7, 0.0),       # narrow rows (one 16-byte vector), alleles >= 4: the per-allele fallback of the general path
    (70, 8, 5, 0.05),
    (33, 100, 9, 0.0),
    (6, 100_000, 1, 0.0),  # H = 200 000: the masks exceed the LDS budget -> global-mask route of the same kernels
    (5, 90_001, 2, 0.02),  # the same, multi-allelic with missing calls, ragged width
]


@pytest.mark.parametrize("sites,samples,max_allele,p_missing", SHAPES)
def test_summaries_and_hudson_dense(dev, sites, samples, max_allele, p_missing):
    rng = np.random.default_rng(sites * 131 + samples)
    m = H.random_dense_matrix(rng, sites, samples, 2, max_allele, p_missing)
    dm = upload(dev, m)
    assert dm.scan_max_allele() == m.max_allele
    half = samples // 2
    h1 = H.haps_for_samples(range(0, half)) + [(0, 0), (samples + 5, 1)]  # duplicate + out of range
    h2 = H.haps_for_samples(range(half, samples - (1 if samples > 2 else 0)))  # last sample ungrouped
    g = dev.Groups.from_haplotype_lists(dm, [h1, h2])
    off1 = R.dense_membership_offsets(m, h1)
    off2 = R.dense_membership_offsets(m, h2)
    assert g.sizes == [len(off1), len(off2)]

    # ---- (a3) build_dense_population_summary, stats.rs:1367-1470 ----
    if max_allele <= 1:
        s = dev.population_summaries(dm, g, dev.FORMULA_SUMMARY)
        for p, hl in enumerate((h1, h2)):
            exp = R.build_dense_population_summary(m, hl)
            assert np.array_equal(s.alt[p], np.array(exp.alt_counts, dtype=np.uint32))
            assert np.array_equal(s.called[p], np.array(exp.called_counts, dtype=np.uint32))
            assert s.totals[p]["segregating_sites"] == exp.segregating_sites
            assert s.totals[p]["haplotype_capacity"] == exp.haplotype_capacity
            assert s.totals[p]["uncallable_sites"] == sum(1 for c in exp.called_counts if c < 2)
            assert H.rel_close(s.totals[p]["pi_sum"], exp.pi_sum)

    # ---- dense paths: count_segregating_sites_dense 3891, calculate_pi_dense 4534 ----
    d = dev.population_summaries(dm, g, dev.FORMULA_DENSE)
    L = 10 * sites + 7
    for p, off in enumerate((off1, off2)):
        assert d.totals[p]["segregating_sites"] == R.count_segregating_sites_dense(m, off)
        exp_pi = R.calculate_pi_dense(m, off, L)
        eff = L - d.totals[p]["uncallable_sites"]
        assert H.rel_close(d.totals[p]["pi_sum"] / eff, exp_pi)

    # ---- dense_hudson_sites 3060-3278: per-site records bit-exact ----
    variants = [R.Variant(7 * i, None) for i in range(sites)]
    exp_sites = R.dense_hudson_sites(m, variants, off1, off2)
    hs = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE)
    # Multi-allelic dense arm (3106-3139): the reference adds the p1a*p2a products in the order the alleles first occur along the driving
    # population's ascending column offsets; the kernel re-adds them in that order wherever three or more alleles are shared (first_member_column,
    # sweep_kernels.hpp), so every D_xy-derived track is the reference's bits at every max_allele.
    H.assert_bits_equal(hs.sites["fst"], [H.opt(x.fst) for x in exp_sites], "fst")
    H.assert_bits_equal(hs.sites["dxy"], [H.opt(x.d_xy) for x in exp_sites], "dxy")
    H.assert_bits_equal(hs.sites["pi1"], [H.opt(x.pi_pop1) for x in exp_sites], "pi1")
    H.assert_bits_equal(hs.sites["pi2"], [H.opt(x.pi_pop2) for x in exp_sites], "pi2")
    H.assert_bits_equal(hs.sites["num"], [H.opt(x.num_component) for x in exp_sites], "num")
    H.assert_bits_equal(hs.sites["den"], [H.opt(x.den_component) for x in exp_sites], "den")
    assert np.array_equal(hs.sites["called"][0], np.array([x.n1_called for x in exp_sites], dtype=np.uint32))
    assert np.array_equal(hs.sites["called"][1], np.array([x.n2_called for x in exp_sites], dtype=np.uint32))
    num_sum, den_sum = R.hudson_component_sums(exp_sites)
    assert H.rel_close(hs.totals["site_num_sum"], num_sum)
    assert H.rel_close(hs.totals["site_den_sum"], den_sum)
    assert hs.totals["sites_with_components"] == sum(1 for x in exp_sites if x.num_component is not None)
    # calculate_dxy_dense 2526-2611
    exp_dxy = R.calculate_dxy_dense(m, off1, off2, L)
    eff = L - hs.totals["site_dxy_skipped"]
    assert H.rel_close(hs.totals["site_dxy_sum"] / eff, exp_dxy)

    # ---- aggregate_hudson_components_from_summaries 1554-1623 ----
    if max_allele <= 1:
        t = R.aggregate_hudson_components_from_summaries(R.build_dense_population_summary(m, h1),
                                                         R.build_dense_population_summary(m, h2))
        for k in ("numerator_sum", "denominator_sum", "pi1_sum", "pi2_sum", "dxy_sum_all"):
            assert H.rel_close(hs.totals[k], getattr(t, k)), k
        assert hs.totals["dxy_uncallable_sites"] == t.dxy_uncallable_sites


SPARSE_CASES = [
    # (sites, samples, max_allele, p_missing, p_haploid, all_missing_rows)
    (150, 20, 1, 0.0, 0.0, 0),
    (150, 21, 1, 0.15, 0.05, 2),
    (90, 33, 3, 0.1, 0.1, 1),
    (40, 300, 2, 0.02, 0.0, 0),
    (5, 90_000, 1, 0.02, 0.0, 0),  # 180 000 columns: bit masks in LDS for the Hudson pair and the one-group diversity sweep
]


@pytest.mark.parametrize("sites,samples,max_allele,p_missing,p_haploid,dead", SPARSE_CASES)
def test_sparse_formulas(dev, sites, samples, max_allele, p_missing, p_haploid, dead):
    """hudson_site_from_variant (2969), calculate_pi (4317), calculate_d_xy_hudson sparse fold (2476),
    calculate_per_site_diversity (4628), count_segregating_sites (3808) through the dense+mask
    representation the host layer ships to the GPU."""
    rng = random.Random(sites * 7 + samples)
    variants = H.random_sparse_variants(rng, sites, samples, max_allele, p_missing, p_haploid, dead)
    m = H.dense_from_variants(variants, samples)
    dm = upload(dev, m)
    third = samples // 3
    h1 = H.haps_for_samples(range(0, third))
    h2 = H.haps_for_samples(range(third, 2 * third)) + [(2 * third, 0)]  # one half-sample
    g = dev.Groups.from_haplotype_lists(dm, [h1, h2])
    names = [f"s{i}" for i in range(samples)]
    L = variants[-1].position + 10
    p1 = R.PopulationContext(0, h1, variants, names, L)
    p2 = R.PopulationContext(1, h2, variants, names, L)
    region = R.QueryRegion(0, L)
    exp_sites = R.calculate_hudson_fst_per_site(p1, p2, region)
    hs = dev.hudson_sweep(dm, g, dev.FORMULA_SPARSE)
    H.assert_bits_equal(hs.sites["fst"], [H.opt(x.fst) for x in exp_sites], "fst")
    H.assert_bits_equal(hs.sites["dxy"], [H.opt(x.d_xy) for x in exp_sites], "dxy")
    H.assert_bits_equal(hs.sites["pi1"], [H.opt(x.pi_pop1) for x in exp_sites], "pi1")
    H.assert_bits_equal(hs.sites["pi2"], [H.opt(x.pi_pop2) for x in exp_sites], "pi2")
    H.assert_bits_equal(hs.sites["num"], [H.opt(x.num_component) for x in exp_sites], "num")
    H.assert_bits_equal(hs.sites["den"], [H.opt(x.den_component) for x in exp_sites], "den")
    assert np.array_equal(hs.sites["called"][0], np.array([x.n1_called for x in exp_sites], dtype=np.uint32))
    outcome, _ = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, region)
    fst = hs.totals["site_num_sum"] / hs.totals["site_den_sum"] if hs.totals["site_den_sum"] > 1e-12 else None
    assert (fst is None) == (outcome.fst is None)
    if fst is not None:
        assert H.rel_close(fst, outcome.fst)
    for p, hl in enumerate((h1, h2)):
        exp_pi = R.calculate_pi(variants, hl, L)
        assert H.rel_close(hs.pop[p]["pi_sum"] / (L - hs.pop[p]["uncallable_sites"]), exp_pi)
        assert hs.pop[p]["segregating_sites"] == R.count_segregating_sites_for_haplotypes(variants, hl)
    exp_dxy = R.calculate_d_xy_hudson(p1, p2)
    assert H.rel_close(hs.totals["site_dxy_sum"] / (L - hs.totals["site_dxy_skipped"]), exp_dxy)

    # per-site diversity of population 1
    g1 = dev.Groups.from_haplotype_lists(dm, [h1])
    dv = dev.diversity_sites(dm, g1)
    exp_div = R.calculate_per_site_diversity(variants, h1, region)
    H.assert_bits_equal(dv.pi, [x.pi for x in exp_div], "site pi")
    H.assert_bits_equal(dv.theta, [x.watterson_theta for x in exp_div], "site theta")

    # cohort-wide segregating sites: every column is a member (stats.rs:3808-3829)
    gall = dev.Groups(dm, np.ones((1, dm.columns), dtype=np.uint8))
    sall = dev.population_summaries(dm, gall, dev.FORMULA_SPARSE, want_sites=False)
    assert sall.totals[0]["segregating_sites"] == R.count_segregating_sites(variants)


WC_CASES = [
    # (sites, samples, n_groups, max_allele, p_missing, p_haploid, dead_rows, ungrouped)
    (120, 16, 2, 1, 0.0, 0.0, 0, 0),
    (120, 18, 2, 1, 0.2, 0.1, 2, 3),
    (100, 40, 4, 1, 0.05, 0.0, 1, 4),
    (80, 30, 3, 3, 0.1, 0.05, 1, 2),
    (60, 50, 5, 2, 0.05, 0.0, 0, 5),
    (4, 60_000, 3, 1, 0.02, 0.0, 0, 7),  # 120 000 columns: fused W&C with bit masks in LDS
    # biallelic, nothing missing, five to eight groups: the kernels instantiated for EXACTLY five, six and seven groups (round 4) and the eight-group one
    (150, 40, 5, 1, 0.0, 0.0, 0, 3),
    (130, 45, 6, 1, 0.0, 0.0, 0, 0),
    (140, 50, 7, 1, 0.0, 0.0, 0, 2),
    (90, 64, 8, 1, 0.0, 0.0, 0, 1),
    (70, 2600, 5, 1, 0.0, 0.0, 0, 10),  # 5 200 columns: sixteen lanes per row
    (70, 2100, 7, 1, 0.0, 0.0, 0, 0),
]


@pytest.mark.parametrize("sites,samples,G,max_allele,p_missing,p_haploid,dead,ungrouped", WC_CASES)
def test_wc_sweep(dev, sites, samples, G, max_allele, p_missing, p_haploid, dead, ungrouped):
    """calculate_fst_wc_at_site_with_membership (1814-2032) + calculate_overall_fst_wc sums (2145-2374)."""
    rng = random.Random(sites + 1000 * G)
    variants = H.random_sparse_variants(rng, sites, samples, max_allele, p_missing, p_haploid, dead)
    # make one group entirely missing at one site, and uneven group sizes
    names = [f"s{i}" for i in range(samples)]
    sample_to_group = {}
    for i in range(samples - ungrouped):
        left = i % G
        right = (i // 2) % G if i % 5 == 0 else left  # some samples straddle two groups
        sample_to_group[names[i]] = (left, right)
    m = H.dense_from_variants(variants, samples)
    dm = upload(dev, m)
    hap_to_group = R.map_samples_to_haplotype_groups(names, sample_to_group)
    mem = R.SubpopulationMembership.from_map(samples, hap_to_group)
    assert mem.group_count() == G
    masks = np.zeros((G, dm.columns), dtype=np.uint8)
    for i in range(samples):
        if mem.left[i] != R.INVALID_GROUP:
            masks[mem.left[i], i * dm.ploidy] = 1
        if mem.right[i] != R.INVALID_GROUP and dm.ploidy > 1:
            masks[mem.right[i], i * dm.ploidy + 1] = 1
    g = dev.Groups(dm, masks)
    res = R.calculate_fst_wc_haplotype_groups(variants, names, sample_to_group, R.QueryRegion(0, 10 ** 9))
    w = dev.wc_sweep(dm, g)
    keys = [k for _, _, k in mem.pair_keys]
    state_code = {s: i for i, s in enumerate(dev.WC_STATES)}
    exp_a = [[s.variance_components[0] for s in res.site_fst]]
    exp_b = [[s.variance_components[1] for s in res.site_fst]]
    exp_s = [[state_code[s.overall_fst.state] for s in res.site_fst]]
    for k in keys:
        exp_a.append([s.pairwise_variance_components.get(k, (0.0, 0.0))[0] for s in res.site_fst])
        exp_b.append([s.pairwise_variance_components.get(k, (0.0, 0.0))[1] for s in res.site_fst])
        exp_s.append([state_code[s.pairwise_fst[k].state] if k in s.pairwise_fst else 3 for s in res.site_fst])
    for slot in range(1 + len(keys)):
        assert np.array_equal(w.state[slot], np.array(exp_s[slot], dtype=np.uint8)), f"state slot {slot}"
        H.assert_bits_equal(w.a[slot], exp_a[slot], f"a slot {slot}")
        H.assert_bits_equal(w.b[slot], exp_b[slot], f"b slot {slot}")
    for gi, label in enumerate(mem.labels):
        exp_n = [s.population_sizes.get(label, 0) for s in res.site_fst]
        assert np.array_equal(w.group_called[gi], np.array(exp_n, dtype=np.uint32))
    # regional: sums in site order vs GPU tree order -> 1e-9
    est = [res.overall_fst] + [res.pairwise_fst[k] for k in keys]
    for slot, e in enumerate(est):
        if e.state == "insufficient_data_for_estimation":
            assert w.informative_sites[slot] == 0
            continue
        assert w.informative_sites[slot] == e.sites
        assert H.rel_close(float(w.sum_a[slot]), e.sum_a)
        assert H.rel_close(float(w.sum_b[slot]), e.sum_b)


def test_row_ranges_and_empty(dev):
    rng = np.random.default_rng(5)
    m = H.random_dense_matrix(rng, 500, 24, 2, 1, 0.0)
    dm = upload(dev, m)
    g = dev.Groups.from_haplotype_lists(dm, [H.haps_for_samples(range(12)), H.haps_for_samples(range(12, 24))])
    full = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE)
    a = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, 0, 123)
    b = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, 123, 377)
    H.assert_bits_equal(np.concatenate([a.sites["fst"], b.sites["fst"]]), full.sites["fst"], "fst split")
    for k in ("numerator_sum", "denominator_sum", "site_num_sum", "dxy_sum_all"):
        assert H.rel_close(a.totals[k] + b.totals[k], full.totals[k])
    assert a.pop[0]["segregating_sites"] + b.pop[0]["segregating_sites"] == full.pop[0]["segregating_sites"]
    empty = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, 10, 0)
    assert empty.totals["denominator_sum"] == 0.0 and empty.sites["fst"].size == 0
    from ferromic_amd import _abi

    with pytest.raises(_abi.FerromicHipError):
        dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, 400, 200)


def test_generator_matches_host_hash(dev):
    """The on-device synthetic cohort equals the same counter-based stream evaluated on the host."""
    S, N = 200, 33
    dm = dev.DeviceMatrix.alloc(S, N, 2, True)
    rng = np.random.default_rng(1)
    thr = (rng.random((2, S)) * (1 << 24)).astype(np.uint32)
    poc = np.repeat(np.arange(N) >= N // 2, 2).astype(np.uint8)
    dm.generate(1234, 77, thr, poc, missing_threshold24=int(0.05 * (1 << 24)))
    data, words = dm.download()
    data = data.reshape(S, 2 * N)

    def hash24(seed, site, col):
        M = (1 << 64) - 1
        z = (seed + 0x9E3779B97F4A7C15 * ((site * 0x100000001B3 + col + 1) & M)) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z = z ^ (z >> 31)
        return z >> 40

    for s in (0, 1, 57, 199):
        for h in (0, 1, 31, 32, 65):
            miss = hash24(1234 ^ 0xA5A5A5A5DEADBEEF, 77 + s, h) < int(0.05 * (1 << 24))
            bit = 1 if hash24(1234, 77 + s, h) < int(thr[poc[h], s]) else 0
            idx = s * 2 * N + h
            got_miss = (int(words[idx >> 6]) >> (idx & 63)) & 1
            assert got_miss == int(miss)
            assert data[s, h] == (0 if miss else bit)


def test_rows_wider_than_lds_for_all_masks(dev):
    """24 000 haplotypes x 8 groups: the eight byte masks (8 x 24 000 B) do not fit the 150 KiB of LDS the fused sweep
    may use, so W&C re-batches the groups (counts path) and the summaries sweep the populations in batches - same
    numbers as the oracle / as the fused kernel on a subset that fits."""
    rng = np.random.default_rng(99)
    S, N, G = 24, 12_000, 8
    m = H.random_dense_matrix(rng, S, N, 2, 1, 0.01)
    dm = upload(dev, m)
    pop_of_sample = rng.integers(0, G, size=N)
    lists = [H.haps_for_samples(np.nonzero(pop_of_sample == g)[0].tolist()) for g in range(G)]
    masks = np.stack([dev.Groups.mask_from_haplotypes(dm, hl) for hl in lists])
    groups = dev.Groups(dm, masks)
    got = dev.population_summaries(dm, groups, dev.FORMULA_SUMMARY)
    for g in range(G):
        exp = R.build_dense_population_summary(m, lists[g])
        assert got.totals[g]["segregating_sites"] == exp.segregating_sites
        assert np.array_equal(got.alt[g], np.array(exp.alt_counts, dtype=np.uint32))
        assert np.array_equal(got.called[g], np.array(exp.called_counts, dtype=np.uint32))
        assert H.rel_close(got.totals[g]["pi_sum"], exp.pi_sum)
    w = dev.wc_sweep(dm, groups)          # too wide for the fused kernel -> counts path inside the library
    w8 = dev.wc_sweep_many(dm, masks)
    assert np.array_equal(w.a, w8.a) and np.array_equal(w.b, w8.b) and np.array_equal(w.state, w8.state)
    assert np.array_equal(w.group_called, got.called)
    # four of the groups fit the fused kernel: the shared pair slots must agree bit for bit
    keep = [0, 3, 5, 6]
    w4 = dev.wc_sweep(dm, dev.Groups(dm, masks[keep]))
    slot8 = {}
    k = 1
    for i in range(G):
        for j in range(i + 1, G):
            slot8[(i, j)] = k
            k += 1
    k4 = 1
    for x in range(4):
        for y in range(x + 1, 4):
            k8 = slot8[(keep[x], keep[y])]
            assert np.array_equal(w4.a[k4], w.a[k8]) and np.array_equal(w4.b[k4], w.b[k8])
            k4 += 1


@pytest.mark.parametrize("p_missing", [0.0, 0.04])
@pytest.mark.parametrize("formula", ["FORMULA_SUMMARY", "FORMULA_DENSE", "FORMULA_SPARSE"])
def test_hudson_pair_from_count_tables(dev, p_missing, formula):
    """fmh_hudson_from_counts (two populations' per-site count tables instead of a matrix: the reference's aggregate over two
    DensePopulationSummary objects, stats.rs:1554-1623) against fmh_hudson_sweep on the matrix both tables came from: every per-site
    record bit for bit, integer totals exactly, f64 totals to the order of their additions.  (Tables of two DIFFERENT matrices: the API
    fuzz, test_two_separately_built_numpy_populations, against the oracle.)"""
    rng = np.random.default_rng(31 + int(p_missing * 100))
    S, N = 5000, 70
    f = getattr(dev, formula)
    m = H.random_dense_matrix(rng, S, N, 2, 1, p_missing)
    dm = upload(dev, m)
    lists = [H.haps_for_samples(range(0, 30)), H.haps_for_samples(range(25, N))]
    g = dev.Groups.from_haplotype_lists(dm, lists)
    ref = dev.hudson_sweep(dm, g, f)
    summ = dev.population_summaries(dm, g, dev.FORMULA_SUMMARY)
    caps = [2 * 30, 2 * (N - 25)]
    tabs = [dev.DeviceBuffer.from_numpy(dm.device, x) for x in (summ.called[0], summ.alt[0], summ.called[1], summ.alt[1])]
    for (r0, rows) in ((0, S), (0, 1), (0, 777)):
        want = ref if rows == S else dev.hudson_sweep(dm, g, f, 0, rows)
        got = dev.hudson_from_counts(dm.device, tabs[0], tabs[1], caps[0], tabs[2], tabs[3], caps[1], rows, f, any_missing=p_missing > 0)
        for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            H.assert_bits_equal(got.sites[k], want.sites[k], k)
        assert np.array_equal(got.sites["called"], want.sites["called"]) and np.array_equal(got.sites["alt"], want.sites["alt"])
        for k, v in want.totals.items():
            if isinstance(v, int): assert got.totals[k] == v, k
            else: assert H.rel_close(got.totals[k], v, 1e-11), k
        for p in range(2):
            for k, v in want.pop[p].items():
                if isinstance(v, int): assert got.pop[p][k] == v, (p, k)
                else: assert H.rel_close(got.pop[p][k], v, 1e-11), (p, k)
    empty = dev.hudson_from_counts(dm.device, tabs[0], tabs[1], caps[0], tabs[2], tabs[3], caps[1], 0, f, want_sites=False)
    assert empty.totals["sites_with_components"] == 0 and empty.pop[0]["haplotype_capacity"] == caps[0]


@pytest.mark.parametrize("G,max_allele,p_missing,S", [(9, 1, 0.0, 3000), (12, 1, 0.02, 5000), (12, 1, 0.0, 2000), (26, 1, 0.0, 40_000), (26, 3, 0.01, 2500), (40, 6, 0.0, 700), (70, 1, 0.3, 900)])
def test_many_groups_totals_without_tracks(dev, G, max_allele, p_missing, S):
    """fmh_wc_sweep_many with no per-site track asked for (run_vcf's CSV populations): the regional sums come straight from the count
    tables (one thread per pair over the sites of a chunk) and equal the sums of the per-site route's tracks - informative sites exactly,
    sum a / sum b to the order of the additions."""
    rng = np.random.default_rng(1000 * G + max_allele)
    N = 180
    m = H.random_dense_matrix(rng, S, N, 2, max_allele, p_missing)
    dm = upload(dev, m)
    pop_of_sample = rng.integers(0, G, size=N)
    pop_of_sample[:G] = np.arange(G)  # no empty group ...
    if G == 12:
        pop_of_sample[pop_of_sample == 5] = 6  # ... but one here: a group without members takes part in no slot
    masks = np.stack([dev.Groups.mask_from_haplotypes(dm, H.haps_for_samples(np.nonzero(pop_of_sample == g)[0].tolist())) for g in range(G)])
    for (r0, rows) in ((0, S), (S // 3, S // 2 + 5), (7, 3)):
        full = dev.wc_sweep_many(dm, masks, r0, rows)
        tot = dev.wc_sweep_many(dm, masks, r0, rows, sites=False)
        assert tot.a is None and tot.sites_attempted == rows
        assert np.array_equal(tot.informative_sites, full.informative_sites)
        counted = full.state != 3
        assert np.array_equal(tot.informative_sites, counted.sum(axis=1).astype(np.uint64))
        exp_a = np.where(counted, full.a, 0.0).sum(axis=1)
        exp_b = np.where(counted, full.b, 0.0).sum(axis=1)
        assert np.allclose(tot.sum_a, exp_a, rtol=1e-10, atol=1e-10) and np.allclose(tot.sum_b, exp_b, rtol=1e-10, atol=1e-10)
        assert np.allclose(tot.sum_a, full.sum_a, rtol=1e-10, atol=1e-10) and np.allclose(tot.sum_b, full.sum_b, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("G,S", [(9, 3000), (12, 2111), (26, 20_000), (64, 1500)])
def test_biallelic_pair_totals_kernel_against_the_general_one(dev, fmh_opts, G, S):
    """More than eight groups, biallelic, nothing missing, no track asked for: wc_pair_totals_biallelic_kernel (frequencies per site and
    group staged once, R threads per pair) against the general pair kernel (FMH_WC_BI_TOTALS=0) and against the sums of the per-site
    route - informative sites exactly, the sums to the order of the additions, for every replica count and ragged row ranges."""
    rng = np.random.default_rng(77 * G)
    N = 200
    m = H.random_dense_matrix(rng, S, N, 2, 1, 0.0)
    dm = upload(dev, m)
    pop_of_sample = rng.integers(0, G, size=N)
    pop_of_sample[:G] = np.arange(G)
    if G == 12:
        pop_of_sample[pop_of_sample == 5] = 6  # a group without members
    masks = np.stack([dev.Groups.mask_from_haplotypes(dm, H.haps_for_samples(np.nonzero(pop_of_sample == g)[0].tolist())) for g in range(G)])
    for (r0, rows) in ((0, S), (S // 3, S // 2 + 5), (7, 3), (1, 33)):
        fmh_opts.setenv("FMH_WC_BI_TOTALS", "0")
        general = dev.wc_sweep_many(dm, masks, r0, rows, sites=False)
        fmh_opts.delenv("FMH_WC_BI_TOTALS")
        full = dev.wc_sweep_many(dm, masks, r0, rows)
        for replicas in ("0", "1", "2", "4", "8"):
            fmh_opts.setenv("FMH_WC_BI_REPLICAS", replicas)
            tot = dev.wc_sweep_many(dm, masks, r0, rows, sites=False)
            assert np.array_equal(tot.informative_sites, general.informative_sites), replicas
            assert np.array_equal(tot.informative_sites, full.informative_sites), replicas
            for got, want in ((tot.sum_a, general.sum_a), (tot.sum_b, general.sum_b), (tot.sum_a, full.sum_a), (tot.sum_b, full.sum_b)):
                assert np.allclose(got, want, rtol=1e-11, atol=1e-12), replicas
        fmh_opts.delenv("FMH_WC_BI_REPLICAS")
        for chunks in ("1", "7"):  # one workgroup walks every tile / ragged last chunk; the XCD padding of the grid leaves at once
            fmh_opts.setenv("FMH_WC_BI_CHUNKS", chunks)
            tot = dev.wc_sweep_many(dm, masks, r0, rows, sites=False)
            assert np.array_equal(tot.informative_sites, general.informative_sites), chunks
            assert np.allclose(tot.sum_a, general.sum_a, rtol=1e-11, atol=1e-12) and np.allclose(tot.sum_b, general.sum_b, rtol=1e-11, atol=1e-12), chunks
        fmh_opts.delenv("FMH_WC_BI_CHUNKS")


def _same_result(a, b, what):
    """Two results of the same call on two images of one matrix: every array the same bits, every scalar equal."""
    if isinstance(a, dict):
        assert a.keys() == b.keys(), what
        for k in a:
            _same_result(a[k], b[k], f"{what}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), what
        for i, (x, y) in enumerate(zip(a, b)):
            _same_result(x, y, f"{what}[{i}]")
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and a.dtype == b.dtype, what
        if a.dtype.kind == "f":
            H.assert_bits_equal(a.reshape(-1), b.reshape(-1), what)
        else:
            assert np.array_equal(a, b), what
    elif isinstance(a, float):
        assert (math.isnan(a) and math.isnan(b)) or a == b, what
    elif a is None or isinstance(a, (int, str, bool, np.integer)):
        assert a == b, what
    else:
        _same_result(vars(a), vars(b), what)


@pytest.mark.parametrize("S,N,max_allele,p_missing,p_multi", [(700, 60, 2, 0.0, 0.03), (450, 2100, 3, 0.02, 0.02), (260, 2100, 5, 0.0, 0.05),
                                                              (400, 150, 6, 0.03, 0.1), (200, 40, 2, 0.0, 0.0), (130, 70, 3, 0.0, 1.0),
                                                              (5000, 30, 2, 0.0, 0.01)])  # the last one: above the default size threshold
def test_rows_without_alleles_above_one_skip_the_upper_planes(dev, fmh_opts, S, N, max_allele, p_missing, p_multi):
    """A packed multi-allelic matrix carries a table of the rows that have a bit in plane 1 or 2 (row_hi_kernel); the sweeps read the upper
    planes of those rows only and run the one-plane core on steps without any.  Same matrix uploaded with the table (default) and without
    (FMH_ROW_HI=0): every track and every total the same bits, on four- and sixteen-lane rows, two and three planes, with missing calls,
    on unaligned row ranges; and the dense Hudson tracks against the oracle."""
    rng = np.random.default_rng(S * 7 + N + max_allele)
    Hc = 2 * N
    freq = rng.beta(0.8, 0.8, size=(S, 1))
    data = (rng.random((S, Hc)) < freq).astype(np.uint8)
    multi_rows = np.nonzero(rng.random(S) < p_multi)[0]
    if p_multi > 0 and len(multi_rows) == 0:
        multi_rows = np.array([S // 2])
    for r in multi_rows:
        vals = rng.integers(0, max_allele + 1, size=Hc, dtype=np.uint8)
        vals[rng.random(Hc) < 0.5] = 0
        data[r] = vals
    if len(multi_rows):
        data[multi_rows[0], 3] = max_allele  # the matrix's max_allele is met
    missing = None
    if p_missing > 0:
        miss = rng.random((S, Hc)) < p_missing
        data[miss] = 0
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        words = np.frombuffer(np.concatenate([bits, np.zeros((-len(bits)) % 8, np.uint8)]).tobytes(), dtype="<u8")
        missing = [int(w) for w in words]
    declared = max(int(data.max()), 2)  # (p_multi = 0: a matrix declared multi-allelic whose rows are all biallelic)
    m = R.DenseGenotypeMatrix(bytes(data.reshape(-1)), missing, S, N, 2, declared)
    fmh_opts.setenv("FMH_ROW_HI", "2")  # the table at any size (by default matrices below 4 096 rows go without)
    with_table = upload(dev, m)
    fmh_opts.setenv("FMH_ROW_HI", "0")
    without = upload(dev, m)
    fmh_opts.delenv("FMH_ROW_HI")
    by_default = upload(dev, m) if S >= 4096 else None
    third = N // 3
    lists2 = [H.haps_for_samples(range(0, N // 2)), H.haps_for_samples(range(N // 2, N - 1))]
    lists3 = [H.haps_for_samples(range(i * third, (i + 1) * third)) for i in range(3)]
    results = []
    for dm in (with_table, without):
        g2, g3, g1 = dev.Groups.from_haplotype_lists(dm, lists2), dev.Groups.from_haplotype_lists(dm, lists3), dev.Groups.from_haplotype_lists(dm, lists2[:1])
        out = []
        for (r0, rows) in ((0, S), (5, S - 9), (S // 2 + 1, min(70, S - S // 2 - 1)), (3, 1)):
            out.append(dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE, r0, rows))
            out.append(dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE, r0, rows))
            out.append(dev.wc_sweep(dm, g3, r0, rows))
            out.append(dev.population_summaries(dm, g3, dev.FORMULA_DENSE, r0, rows))
            out.append(dev.diversity_sites(dm, g1, r0, rows))
        results.append(out)
    for i, (a, b) in enumerate(zip(*results)):
        _same_result(a, b, f"call {i}")
    if by_default is not None:
        g2 = dev.Groups.from_haplotype_lists(by_default, lists2)
        _same_result(dev.hudson_sweep(by_default, g2, dev.FORMULA_DENSE, 0, S), results[1][0], "default options")
    if S * N <= 60_000:  # the oracle's dense Hudson sites on the small shapes
        off1, off2 = R.dense_membership_offsets(m, lists2[0]), R.dense_membership_offsets(m, lists2[1])
        exp = R.dense_hudson_sites(m, [R.Variant(7 * i, None) for i in range(S)], off1, off2)
        hs = results[0][0]
        H.assert_bits_equal(hs.sites["dxy"], [H.opt(e.d_xy) for e in exp], "dxy vs oracle")
        H.assert_bits_equal(hs.sites["fst"], [H.opt(e.fst) for e in exp], "fst vs oracle")


@pytest.mark.parametrize("S,N,p_gap_rows,max_allele", [(900, 60, 0.03, 1), (700, 2100, 0.02, 1), (300, 45, 0.0, 1), (260, 50, 1.0, 1), (5000, 40, 0.01, 1),
                                                        (600, 80, 0.05, 3), (400, 2100, 0.04, 2)])
def test_rows_without_uncalled_columns_skip_the_called_plane(dev, fmh_opts, S, N, p_gap_rows, max_allele):
    """A packed matrix with a called plane carries a table of the rows that have an uncalled column (row_gap_kernel); the sweeps read the
    called plane of those rows only, and a counting step of complete rows runs the no-missing core with the group sizes as called counts.
    Same matrix uploaded with the table and without: every track and every total the same bits (deferring two-group kernels, four- and
    sixteen-lane rows, biallelic and multi-allelic, unaligned row ranges); dense Hudson tracks against the oracle on the small shapes."""
    rng = np.random.default_rng(S * 3 + N)
    Hc = 2 * N
    freq = rng.beta(0.8, 0.8, size=(S, 1))
    data = (rng.random((S, Hc)) < freq).astype(np.uint8)
    if max_allele > 1:
        for r in np.nonzero(rng.random(S) < 0.05)[0]:
            data[r] = np.where(rng.random(Hc) < 0.5, 0, rng.integers(0, max_allele + 1, size=Hc)).astype(np.uint8)
        data[S // 3, 1] = max_allele
    miss = np.zeros((S, Hc), dtype=bool)
    gap_rows = np.nonzero(rng.random(S) < p_gap_rows)[0]
    if p_gap_rows > 0 and len(gap_rows) == 0:
        gap_rows = np.array([S // 2])
    for r in gap_rows:
        miss[r] = rng.random(Hc) < (0.5 if r % 7 == 0 else 0.03)
        miss[r, int(rng.integers(0, Hc))] = True
    data[miss] = 0
    bits = np.packbits(miss.reshape(-1), bitorder="little")
    words = np.frombuffer(np.concatenate([bits, np.zeros((-len(bits)) % 8, np.uint8)]).tobytes(), dtype="<u8")
    m = R.DenseGenotypeMatrix(bytes(data.reshape(-1)), [int(w) for w in words], S, N, 2, max(int(data.max()), max_allele))
    fmh_opts.setenv("FMH_ROW_HI", "2")
    with_table = upload(dev, m)
    fmh_opts.setenv("FMH_ROW_HI", "0")
    without = upload(dev, m)
    fmh_opts.delenv("FMH_ROW_HI")
    by_default = upload(dev, m) if S >= 4096 else None
    third = N // 3
    lists2 = [H.haps_for_samples(range(0, N // 2)), H.haps_for_samples(range(N // 2, N - 1))]
    lists3 = [H.haps_for_samples(range(i * third, (i + 1) * third)) for i in range(3)]
    results = []
    for dm in (with_table, without):
        g2, g3, g1 = dev.Groups.from_haplotype_lists(dm, lists2), dev.Groups.from_haplotype_lists(dm, lists3), dev.Groups.from_haplotype_lists(dm, lists2[:1])
        out = []
        for (r0, rows) in ((0, S), (5, S - 9), (S // 2 + 1, min(70, S - S // 2 - 1)), (3, 1)):
            out.append(dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE, r0, rows))
            out.append(dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE, r0, rows))
            out.append(dev.wc_sweep(dm, g3, r0, rows))
            out.append(dev.population_summaries(dm, g3, dev.FORMULA_DENSE, r0, rows))
            out.append(dev.diversity_sites(dm, g1, r0, rows))
        results.append(out)
    for i, (a, b) in enumerate(zip(*results)):
        _same_result(a, b, f"call {i}")
    if by_default is not None:
        g2 = dev.Groups.from_haplotype_lists(by_default, lists2)
        _same_result(dev.hudson_sweep(by_default, g2, dev.FORMULA_DENSE, 0, S), results[1][0], "default options")
    if S * N <= 60_000:
        off1, off2 = R.dense_membership_offsets(m, lists2[0]), R.dense_membership_offsets(m, lists2[1])
        exp = R.dense_hudson_sites(m, [R.Variant(7 * i, None) for i in range(S)], off1, off2)
        hs = results[0][0]
        H.assert_bits_equal(hs.sites["dxy"], [H.opt(e.d_xy) for e in exp], "dxy vs oracle")
        H.assert_bits_equal(hs.sites["fst"], [H.opt(e.fst) for e in exp], "fst vs oracle")


def test_mask_routes_agree(dev, fmh_opts):
    """The sweep keeps the group masks as bytes in LDS, as bits in LDS (rows too wide for bytes) or as bytes in global
    memory (rows too wide for bits); FMH_MASK_MODE forces the slower routes on rows that do not need them.  All three
    must give the same bits, with and without missing calls, on the biallelic and the general counting paths."""
    rng = np.random.default_rng(4242)
    for (S, N, max_allele, p_missing) in ((70, 900, 1, 0.0), (70, 901, 1, 0.03), (40, 1300, 3, 0.02), (33, 777, 6, 0.0)):
        m = H.random_dense_matrix(rng, S, N, 2, max_allele, p_missing)
        dm = upload(dev, m)
        cut = N // 3
        lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, N - 5))]
        g2 = dev.Groups.from_haplotype_lists(dm, lists)
        g1 = dev.Groups.from_haplotype_lists(dm, lists[:1])
        thirds = [H.haps_for_samples(range(i, N, 3)) for i in range(3)]
        g3 = dev.Groups.from_haplotype_lists(dm, thirds)

        def run():
            hs = dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE)
            dv = dev.diversity_sites(dm, g1)
            w = dev.wc_sweep(dm, g3)
            ps = dev.population_summaries(dm, g3, dev.FORMULA_SUMMARY)
            return hs, dv, w, ps

        fmh_opts.delenv("FMH_MASK_MODE", raising=False)
        base = run()
        for forced in ("2", "1"):
            fmh_opts.setenv("FMH_MASK_MODE", forced)
            got = run()
            for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                H.assert_bits_equal(got[0].sites[k], base[0].sites[k], f"{k} mode {forced}")
            assert got[0].totals == base[0].totals and got[0].pop == base[0].pop
            H.assert_bits_equal(got[1].pi, base[1].pi, "site pi")
            H.assert_bits_equal(got[1].theta, base[1].theta, "site theta")
            assert np.array_equal(got[2].a, base[2].a, equal_nan=True) and np.array_equal(got[2].b, base[2].b, equal_nan=True)
            assert np.array_equal(got[2].state, base[2].state) and np.array_equal(got[2].group_called, base[2].group_called)
            assert np.array_equal(got[3].alt, base[3].alt) and np.array_equal(got[3].called, base[3].called)
            assert got[3].totals == base[3].totals
        fmh_opts.delenv("FMH_MASK_MODE", raising=False)


def test_exact_group_count_kernels_are_the_padded_kernels_bits(dev, fmh_opts):
    """Five, six and seven W&C groups on a packed biallelic matrix with nothing missing run kernels instantiated for exactly that many groups;
    FMH_WC_EXACT=0 sends them through the padded eight-group kernel again.  Every per-site a, b, state and count must be the same bits, the
    informative-site counts equal, the regional sums equal to 1e-12 (the eight-group kernel sums through another LDS transposition layout)."""
    rng = np.random.default_rng(5678)
    for (S, N, G) in ((3000, 400, 5), (2500, 1250, 5), (2000, 640, 6), (2100, 900, 7), (1500, 2700, 6)):
        m = H.random_dense_matrix(rng, S, N, 2, 1, 0.0)
        dm = upload(dev, m)
        lists = [H.haps_for_samples(range(i, N - 3, G)) for i in range(G)]
        g = dev.Groups.from_haplotype_lists(dm, lists)
        for blocks in (None, "2"):
            if blocks:
                fmh_opts.setenv("FMH_GRID_BLOCKS", blocks)
            fmh_opts.setenv("FMH_WC_EXACT", "0")
            base = dev.wc_sweep(dm, g, 5, S - 9)
            fmh_opts.setenv("FMH_WC_EXACT", "1")
            got = dev.wc_sweep(dm, g, 5, S - 9)
            what = f"{S}x{N} {G} groups blocks {blocks}"
            assert np.array_equal(got.a.view(np.uint64), base.a.view(np.uint64)) and np.array_equal(got.b.view(np.uint64), base.b.view(np.uint64)), what
            assert np.array_equal(got.state, base.state) and np.array_equal(got.group_called, base.group_called), what
            assert np.array_equal(got.informative_sites, base.informative_sites) and got.sites_attempted == base.sites_attempted, what
            assert np.allclose(got.sum_a, base.sum_a, rtol=1e-12, atol=1e-15) and np.allclose(got.sum_b, base.sum_b, rtol=1e-12, atol=1e-15), what
            fmh_opts.delenv("FMH_GRID_BLOCKS", raising=False)
        fmh_opts.delenv("FMH_WC_EXACT", raising=False)


def test_deferred_epilogues_are_the_same_bits(dev, fmh_opts):
    """The biallelic kernels with one, two (and, packed, four) groups count several of a wave's tiles before they run those tiles' epilogues
    (DESIGN.md section 3 "Deferred epilogues"; depth chosen per launch, FMH_DEFER_TILES forces it).  The tiles a wave takes and their order
    are the same at every depth, so every per-site value AND every regional total must be the same bits as the undeferred order - checked
    here on ragged matrices with the grid held to a few workgroups (FMH_GRID_BLOCKS) so that a wave really sweeps dozens of tiles, on packed
    and u8 rows, four- and sixteen-lane rows, with and without missing calls, and against the oracle's counts."""
    rng = np.random.default_rng(1606)
    cases = ((5000, 450, 0.0, "packed"), (4097, 3000, 0.0, "packed"), (3001, 2700, 0.02, "packed"), (2500, 600, 0.03, "packed"),
             (3333, 640, 0.0, "bytes"), (2049, 700, 0.05, "bytes"))
    for (S, N, p_missing, layout) in cases:
        m = H.random_dense_matrix(rng, S, N, 2, 1, p_missing)
        if layout == "bytes":
            fmh_opts.setenv("FMH_LAYOUT", "bytes")
        dm = upload(dev, m)
        fmh_opts.delenv("FMH_LAYOUT", raising=False)
        cut = N // 3
        lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, N - 2))]
        quarters = [H.haps_for_samples(range(i, N, 4)) for i in range(4)]
        g2, g1, g4 = (dev.Groups.from_haplotype_lists(dm, x) for x in (lists, lists[:1], quarters))

        def run():
            return (dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE), dev.diversity_sites(dm, g1), dev.population_summaries(dm, g2, dev.FORMULA_SUMMARY),
                    dev.population_summaries(dm, g4, dev.FORMULA_SUMMARY), dev.wc_sweep(dm, g4), dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE, 7, S - 11))

        for blocks in ("1", "3"):
            fmh_opts.setenv("FMH_GRID_BLOCKS", blocks)
            fmh_opts.setenv("FMH_DEFER_TILES", "1")
            base = run()
            for depth in ("2", "5", "16"):
                fmh_opts.setenv("FMH_DEFER_TILES", depth)
                got = run()
                what = f"{S}x{N} {layout} missing {p_missing} blocks {blocks} depth {depth}"
                for i in (0, 5):
                    for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                        H.assert_bits_equal(got[i].sites[k], base[i].sites[k], k + " " + what)
                    assert np.array_equal(got[i].sites["alt"], base[i].sites["alt"]) and np.array_equal(got[i].sites["called"], base[i].sites["called"]), what
                    assert got[i].totals == base[i].totals and got[i].pop == base[i].pop, what
                H.assert_bits_equal(got[1].pi, base[1].pi, "site pi " + what)
                H.assert_bits_equal(got[1].theta, base[1].theta, "site theta " + what)
                assert got[1].totals == base[1].totals, what
                for i in (2, 3):
                    assert np.array_equal(got[i].alt, base[i].alt) and np.array_equal(got[i].called, base[i].called) and got[i].totals == base[i].totals, what
                assert np.array_equal(got[4].a, base[4].a, equal_nan=True) and np.array_equal(got[4].b, base[4].b, equal_nan=True), what
                assert np.array_equal(got[4].state, base[4].state) and np.array_equal(got[4].sum_a, base[4].sum_a) and np.array_equal(got[4].sum_b, base[4].sum_b), what
            fmh_opts.delenv("FMH_DEFER_TILES")
        fmh_opts.delenv("FMH_GRID_BLOCKS")
        exp = R.build_dense_population_summary(m, lists[0])
        assert np.array_equal(base[0].sites["alt"][0], np.array(exp.alt_counts, dtype=np.uint32))
        assert np.array_equal(base[0].sites["called"][0], np.array(exp.called_counts, dtype=np.uint32))


def test_matrix_core_counting_route_agrees(dev, fmh_opts):
    """BASELINE config C5: the counts as an int8 MFMA contraction (FMH_COUNTS_MFMA, u8 rows, biallelic, nothing missing).  Same
    integers as the dot4 route, the same epilogue code after them: every track and total must be the same bits - and against
    the oracle on the counts themselves.  Shapes exercise ragged K (columns not a multiple of 16 / 64 / 256), tiles with rows
    past the end, a row range that starts inside the matrix, and one to four groups."""
    from oracle import ferromic_ref as R

    rng = np.random.default_rng(90210)
    fmh_opts.setenv("FMH_LAYOUT", "bytes")
    for (S, N) in ((70, 900), (129, 37), (64, 128), (1000, 1301), (3, 8), (257, 5000)):
        m = H.random_dense_matrix(rng, S, N, 2, 1, 0.0)
        dm = upload(dev, m)
        cut = N // 3
        lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, max(cut + 1, N - 5)))]
        g2 = dev.Groups.from_haplotype_lists(dm, lists)
        g1 = dev.Groups.from_haplotype_lists(dm, lists[:1])
        thirds = [H.haps_for_samples(range(i, N, 3)) for i in range(3)]
        g3 = dev.Groups.from_haplotype_lists(dm, thirds)
        quarters = [H.haps_for_samples(range(i, N, 4)) for i in range(4)]
        g4 = dev.Groups.from_haplotype_lists(dm, quarters)
        r0, rc = (S // 5, S - S // 5 - 1) if S > 10 else (0, S)

        def run():
            hs = dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE)
            hr = dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE, r0, rc)
            dv = dev.diversity_sites(dm, g1)
            w3, w4, w2 = dev.wc_sweep(dm, g3), dev.wc_sweep(dm, g4), dev.wc_sweep(dm, g2)
            ps = dev.population_summaries(dm, g4, dev.FORMULA_SUMMARY)
            return hs, hr, dv, w3, w4, w2, ps

        fmh_opts.delenv("FMH_COUNTS_MFMA", raising=False)
        base = run()
        s1 = R.build_dense_population_summary(m, lists[0])
        assert np.array_equal(base[0].sites["alt"][0], np.array(s1.alt_counts, dtype=np.uint32))
        for unroll in ("1", "2"):
            fmh_opts.setenv("FMH_COUNTS_MFMA", unroll)
            got = run()
            for i in (0, 1):
                for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                    H.assert_bits_equal(got[i].sites[k], base[i].sites[k], f"{k} mfma {unroll} {S}x{N}")
                assert np.array_equal(got[i].sites["alt"], base[i].sites["alt"]) and np.array_equal(got[i].sites["called"], base[i].sites["called"])
                assert got[i].totals == base[i].totals and got[i].pop == base[i].pop
            H.assert_bits_equal(got[2].pi, base[2].pi, "site pi")
            H.assert_bits_equal(got[2].theta, base[2].theta, "site theta")
            for i in (3, 4, 5):
                assert np.array_equal(got[i].a, base[i].a, equal_nan=True) and np.array_equal(got[i].b, base[i].b, equal_nan=True)
                assert np.array_equal(got[i].state, base[i].state) and np.array_equal(got[i].group_called, base[i].group_called)
                assert np.array_equal(got[i].sum_a, base[i].sum_a) and np.array_equal(got[i].sum_b, base[i].sum_b)
                assert np.array_equal(got[i].informative_sites, base[i].informative_sites)
            assert np.array_equal(got[6].alt, base[6].alt) and np.array_equal(got[6].called, base[6].called)
            assert got[6].totals == base[6].totals
        fmh_opts.delenv("FMH_COUNTS_MFMA", raising=False)


def test_eight_groups_on_rows_beyond_the_bit_mask_budget(dev):
    """160 000 haplotypes x 8 groups: even as bits the eight masks pass the LDS budget (8 x 10 048 x 2 B > 150 KiB),
    so W&C goes through the counts path in batches of four groups; a fused four-group sweep (bit masks) must give the
    same numbers on the slots they share."""
    rng = np.random.default_rng(77)
    S, N, G = 6, 80_000, 8
    m = H.random_dense_matrix(rng, S, N, 2, 1, 0.01)
    dm = upload(dev, m)
    pop_of_sample = rng.integers(0, G, size=N)
    lists = [H.haps_for_samples(np.nonzero(pop_of_sample == g)[0].tolist()) for g in range(G)]
    masks = np.stack([dev.Groups.mask_from_haplotypes(dm, hl) for hl in lists])
    groups = dev.Groups(dm, masks)
    got = dev.population_summaries(dm, groups, dev.FORMULA_SUMMARY)
    w = dev.wc_sweep(dm, groups)
    assert np.array_equal(w.group_called, got.called)
    keep = [1, 2, 4, 7]
    w4 = dev.wc_sweep(dm, dev.Groups(dm, masks[keep]))
    slot8, k = {}, 1
    for i in range(G):
        for j in range(i + 1, G):
            slot8[(i, j)] = k
            k += 1
    k4 = 1
    for x in range(4):
        for y in range(x + 1, 4):
            k8 = slot8[(keep[x], keep[y])]
            assert np.array_equal(w4.a[k4], w.a[k8], equal_nan=True) and np.array_equal(w4.b[k4], w.b[k8], equal_nan=True)
            k4 += 1
    for g in (0, 5):
        exp = R.build_dense_population_summary(m, lists[g])
        assert np.array_equal(got.alt[g], np.array(exp.alt_counts, dtype=np.uint32))
        assert np.array_equal(got.called[g], np.array(exp.called_counts, dtype=np.uint32))


def test_packed_and_byte_layouts_agree(dev, fmh_opts):
    """A generated cohort is swept from its u8 rows (FMH_LAYOUT=bytes), from the bit-packed image next to them, and from
    the packed image alone (bytes released); every output must be the same bits.  Then the packed-only matrix is
    downloaded (unpack), scanned for its max allele and sent through the pairwise Gram (unpack staging)."""
    rng = np.random.default_rng(31)
    for (S, N, max_allele, p_missing) in ((300, 700, 1, 0.0), (257, 333, 1, 0.04), (130, 1100, 3, 0.0), (90, 260, 2, 0.1), (70, 40, 3, 0.02),
                                          (150, 900, 7, 0.0), (97, 333, 5, 0.06), (64, 2700, 4, 0.01), (40, 70, 6, 0.0)):  # three planes
        m = H.random_dense_matrix(rng, S, N, 2, max_allele, p_missing)
        fmh_opts.setenv("FMH_LAYOUT", "bytes")
        dm = upload(dev, m)                       # u8 rows only
        fmh_opts.delenv("FMH_LAYOUT")
        cut = N // 3
        lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, N - 3))]
        thirds = [H.haps_for_samples(range(i, N, 3)) for i in range(3)]
        g2, g1, g3 = (dev.Groups.from_haplotype_lists(dm, x) for x in (lists, lists[:1], thirds))

        def run():
            return (dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE), dev.diversity_sites(dm, g1), dev.wc_sweep(dm, g3),
                    dev.population_summaries(dm, g3, dev.FORMULA_SUMMARY), dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE))

        base = run()
        host_before = dm.download()
        pd_before = dev.pairwise_differences(dm, N)

        def same(got):
            for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                H.assert_bits_equal(got[0].sites[k], base[0].sites[k], k)
                H.assert_bits_equal(got[4].sites[k], base[4].sites[k], k + " sparse")
            assert np.array_equal(got[0].sites["alt"], base[0].sites["alt"]) and np.array_equal(got[0].sites["called"], base[0].sites["called"])
            assert got[0].totals == base[0].totals and got[0].pop == base[0].pop
            H.assert_bits_equal(got[1].pi, base[1].pi, "site pi")
            H.assert_bits_equal(got[1].theta, base[1].theta, "site theta")
            assert np.array_equal(got[1].distinct, base[1].distinct)
            assert np.array_equal(got[2].a, base[2].a, equal_nan=True) and np.array_equal(got[2].b, base[2].b, equal_nan=True)
            assert np.array_equal(got[2].state, base[2].state) and np.array_equal(got[2].group_called, base[2].group_called)
            assert np.array_equal(got[3].alt, base[3].alt) and np.array_equal(got[3].called, base[3].called)
            assert got[3].totals == base[3].totals

        dm.pack()                                  # packed image next to the bytes: the sweeps switch to it
        same(run())
        for lpr, unroll in (("4", "1"), ("4", "5"), ("16", "2"), ("16", "4")):   # both lane layouts of the packed cores, odd batch depths
            fmh_opts.setenv("FMH_PACKED_LPR", lpr)
            fmh_opts.setenv("FMH_PACKED_UNROLL", unroll)
            same(run())
        fmh_opts.delenv("FMH_PACKED_LPR")
        fmh_opts.delenv("FMH_PACKED_UNROLL")
        fmh_opts.setenv("FMH_LAYOUT", "bytes")  # ... unless told otherwise
        same(run())
        fmh_opts.delenv("FMH_LAYOUT")
        dm.pack(release_bytes=True)                # packed image alone
        same(run())
        host_after = dm.download()
        assert np.array_equal(host_after[0], host_before[0])
        if host_before[1] is not None:
            assert np.array_equal(host_after[1], host_before[1])
        exp_max = int(np.asarray(host_before[0]).max()) if p_missing == 0.0 else None
        if exp_max is not None:
            assert dm.scan_max_allele() == exp_max
        pd_after = dev.pairwise_differences(dm, N)
        iu = np.triu_indices(N, k=1)
        assert np.array_equal(pd_after[0][iu], pd_before[0][iu]) and np.array_equal(pd_after[1][iu], pd_before[1][iu])


def test_pack_api_contract(dev):
    """fmh_matrix_pack: refuses what the planes cannot hold, is idempotent, and a matrix that released its bytes says so
    (no byte pointers, no generator) while every read path keeps working."""
    from ferromic_amd import _abi
    import ctypes as C

    rng = np.random.default_rng(8)
    wide_alleles = H.random_dense_matrix(rng, 20, 30, 2, 9, 0.0)
    dm = upload(dev, wide_alleles)                 # max_allele 9: stays on u8 rows (three planes hold alleles 0..7)
    with pytest.raises(_abi.FerromicHipError):
        dm.pack()
    # a max_allele below the data would silently drop allele bits: the packer detects it on the device and refuses
    lying = np.frombuffer(wide_alleles.data, dtype=np.uint8)
    for claimed in (1, 3, 7):
        with pytest.raises(_abi.FerromicHipError, match="above max_allele"):
            dev.DeviceMatrix.from_host(lying, None, 20, 30, 2, claimed)
    # ... but only for CALLED entries: a missing entry may hold any byte
    masked = lying.copy().reshape(20, 60)
    words = np.zeros((20 * 60 + 63) // 64, dtype=np.uint64)
    big = np.argwhere(masked > 3)
    for (r, c) in big:
        idx = int(r) * 60 + int(c)
        words[idx >> 6] |= np.uint64(1) << np.uint64(idx & 63)
    ok3 = dev.DeviceMatrix.from_host(masked.reshape(-1), words, 20, 30, 2, 3)
    back = ok3.download()
    called = ~np.unpackbits(np.asarray(back[1]).view(np.uint8), bitorder="little")[:1200].astype(bool)
    assert np.array_equal(np.asarray(back[0])[called], masked.reshape(-1)[called])
    # wrap validation (header: bits_pitch % 4 == 0, 4-byte aligned called rows)
    buf = dev.DeviceBuffer(0, 4096)
    with pytest.raises(_abi.FerromicHipError, match="multiple of 4"):
        dev.DeviceMatrix.wrap(buf.ptr, 64, buf.ptr + 2048, 6, 4, 30, 2, 1)
    with pytest.raises(_abi.FerromicHipError, match="4-byte aligned"):
        dev.DeviceMatrix.wrap(buf.ptr, 64, buf.ptr + 2050, 8, 4, 30, 2, 1)
    ok = dev.DeviceMatrix.alloc(40, 25, 2, with_missing=True)
    thr = (rng.random((1, 40)) * (1 << 24)).astype(np.uint32)
    ok.generate(3, 0, thr, np.zeros(50, dtype=np.uint8), missing_threshold24=int(0.1 * (1 << 24)))
    before = ok.download()
    ok.pack()
    ok.pack()                                      # again: same planes
    ok.generate(3, 0, thr, np.zeros(50, dtype=np.uint8), missing_threshold24=int(0.1 * (1 << 24)))  # bytes present: re-packed
    ok.pack(release_bytes=True)
    ok.pack(release_bytes=True)                    # nothing left to pack: a no-op
    d, b = C.c_void_p(1), C.c_void_p(1)
    _abi.check(_abi.load().fmh_matrix_device_ptrs(ok._h, C.byref(d), C.byref(b)))
    assert not d.value and not b.value
    with pytest.raises(_abi.FerromicHipError):
        ok.generate(3, 0, thr, np.zeros(50, dtype=np.uint8), 0)
    after = ok.download()
    assert np.array_equal(after[0], before[0]) and np.array_equal(after[1], before[1])
    g = dev.Groups(ok, np.ones((1, 50), dtype=np.uint8))
    s = dev.population_summaries(ok, g, dev.FORMULA_DENSE)
    data = np.asarray(before[0]).reshape(40, 50)
    miss = np.unpackbits(np.asarray(before[1]).view(np.uint8), bitorder="little")[:40 * 50].reshape(40, 50).astype(bool)
    assert np.array_equal(s.called[0], (~miss).sum(axis=1).astype(np.uint32))
    assert np.array_equal(s.alt[0], ((data == 1) & ~miss).sum(axis=1).astype(np.uint32))


@pytest.mark.parametrize("S,N,max_allele,p_missing", [(300, 333, 1, 0.0), (257, 100, 1, 0.07), (130, 700, 3, 0.02), (97, 45, 6, 0.1), (1, 3, 2, 0.0),
                                                       (40_000, 640, 1, 0.01)])
def test_upload_routes_agree(dev, S, N, max_allele, p_missing):
    """Three ways into the same resident image: fmh_matrix_create (u8 rows packed on the HOST into pinned staging - threads, SSE2, ragged
    tails, the linear missing bitset at arbitrary bit offsets; the last shape spans several staging slabs and threads), fmh_matrix_create_packed
    (host bit planes, pitched copies) and the device-side packer (u8 rows uploaded with FMH_LAYOUT=bytes, then fmh_matrix_pack).  Downloads
    and sweeps must be identical."""
    import os

    rng = np.random.default_rng(S + N)
    m = H.random_dense_matrix(rng, S, N, 2, max_allele, p_missing)
    cols = 2 * N
    data = np.frombuffer(m.data, dtype=np.uint8).reshape(S, cols)
    words = H.missing_words_np(m)
    a = upload(dev, m)                                     # host-packed
    from ferromic_amd import _abi

    with _abi.options(FMH_LAYOUT="bytes"):
        b = upload(dev, m)                                 # u8 rows on the device ...
    b.pack(release_bytes=True)                             # ... packed there
    nplanes = 1 if m.max_allele <= 1 else (2 if m.max_allele <= 3 else 3)
    pitch = (cols + 7) // 8 + 5                            # a host pitch of its own
    planes = []
    for k in range(nplanes):
        bits = np.packbits((data >> k) & 1, axis=1, bitorder="little")
        planes.append(np.pad(bits, ((0, 0), (0, pitch - bits.shape[1]))))
    called = None
    if words is not None:
        miss = np.unpackbits(words.view(np.uint8), bitorder="little")[:S * cols].reshape(S, cols)
        cb = np.packbits(1 - miss, axis=1, bitorder="little")
        called = np.pad(cb, ((0, 0), (0, pitch - cb.shape[1])))
    c = dev.DeviceMatrix.from_host_planes(planes, called, S, N, 2, m.max_allele)
    ref = a.download()
    for other in (b, c):
        got = other.download()
        assert np.array_equal(got[0], ref[0])
        assert (got[1] is None) == (ref[1] is None) and (ref[1] is None or np.array_equal(got[1], ref[1]))
    assert np.array_equal(ref[0].reshape(S, cols), data)
    if N >= 3:
        lists = [H.haps_for_samples(range(0, N // 3)), H.haps_for_samples(range(N // 3, N))]
        base = dev.hudson_sweep(a, dev.Groups.from_haplotype_lists(a, lists), dev.FORMULA_SPARSE)
        for other in (b, c):
            got = dev.hudson_sweep(other, dev.Groups.from_haplotype_lists(other, lists), dev.FORMULA_SPARSE)
            for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                H.assert_bits_equal(got.sites[k], base.sites[k], k)
            assert got.totals == base.totals and got.pop == base.pop
    with pytest.raises(Exception, match="allele plane"):
        dev.DeviceMatrix.from_host_planes(planes[:1], called, S, N, 2, 3)


@pytest.mark.parametrize("S,N,max_allele,p_missing", [(300, 37, 1, 0.0), (190, 61, 1, 0.07), (130, 45, 3, 0.0), (130, 45, 5, 0.1), (70, 2500, 1, 0.0),
                                                      (5000, 450, 1, 0.0), (4097, 3000, 1, 0.02), (33, 70000, 1, 0.0), (64, 8, 2, 0.3), (1, 3, 1, 0.0)])
@pytest.mark.parametrize("layout", ["packed", "bytes"])
def test_fused_region_sweep_equals_the_separate_calls(dev, fmh_opts, S, N, max_allele, p_missing, layout):
    """fmh_pair_region_sweep reads the matrix once for what fmh_population_summaries + 2 x fmh_diversity_sites + fmh_hudson_sweep read four
    times: every per-site track must be the same bits, every integer total equal, the f64 totals equal to 1e-12 (another grid sums the
    partials in another order) - with one formula for both parts and with the region driver's pair (dense summaries, sparse Hudson), with
    and without the Hudson part, on a row range that starts inside the matrix."""
    import ctypes as C

    from ferromic_amd import _abi

    lib = _abi.load()
    rng = np.random.default_rng(S * 31 + N + max_allele)
    m = H.random_dense_matrix(rng, S, N, 2, max_allele, p_missing)
    if layout == "bytes":
        fmh_opts.setenv("FMH_LAYOUT", "bytes")
    dm = upload(dev, m)
    cut = max(1, N // 3)
    lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, max(cut + 1, N - 1)))]
    g2 = dev.Groups.from_haplotype_lists(dm, lists)
    g1 = [dev.Groups.from_haplotype_lists(dm, lists[:1]), dev.Groups.from_haplotype_lists(dm, lists[1:])]
    r0, rows = (0, S) if S < 20 else (7, S - 11)
    for summary_formula, hudson_formula in ((dev.FORMULA_SPARSE, dev.FORMULA_SPARSE), (dev.FORMULA_DENSE, dev.FORMULA_SPARSE), (dev.FORMULA_DENSE, -1)):
        bufs = {k: dev.DeviceBuffer(dm.device, 8 * 2 * max(rows, 1)) for k in ("pi", "theta")}
        for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            bufs[k] = dev.DeviceBuffer(dm.device, 8 * max(rows, 1))
        for k in ("alt", "called"):
            bufs[k] = dev.DeviceBuffer(dm.device, 4 * 2 * max(rows, 1))
        div = _abi.PairDiversitySites(bufs["pi"].ptr, bufs["theta"].ptr)
        sites = _abi.HudsonSites(*(bufs[k].ptr for k in ("fst", "dxy", "pi1", "pi2", "num", "den", "alt", "called")))
        tot = _abi.HudsonTotals()
        _abi.check(lib.fmh_pair_region_sweep(dm._h, g2._h, r0, rows, summary_formula, hudson_formula, C.byref(div), C.byref(sites), C.byref(tot), None))
        what = f"{S}x{N} alleles<={max_allele} missing {p_missing} {layout} formulas {summary_formula}/{hudson_formula}"
        ps = dev.population_summaries(dm, g2, summary_formula, r0, rows)
        for p in range(2):
            for k in ("haplotype_capacity", "segregating_sites", "uncallable_sites"):
                assert getattr(tot.pop[p], k) == ps.totals[p][k], (what, p, k)
            assert tot.pop[p].pi_sum == pytest.approx(ps.totals[p]["pi_sum"], rel=1e-12, abs=1e-300), what
            dv = dev.diversity_sites(dm, g1[p], r0, rows)
            H.assert_bits_equal(bufs["pi"].to_numpy(np.float64, 2 * rows).reshape(2, rows)[p], dv.pi, f"site pi group {p} {what}")
            H.assert_bits_equal(bufs["theta"].to_numpy(np.float64, 2 * rows).reshape(2, rows)[p], dv.theta, f"site theta group {p} {what}")
        assert np.array_equal(bufs["alt"].to_numpy(np.uint32, 2 * rows).reshape(2, rows), ps.alt) or max_allele > 1, what
        assert np.array_equal(bufs["called"].to_numpy(np.uint32, 2 * rows).reshape(2, rows), ps.called), what
        if hudson_formula >= 0:
            hs = dev.hudson_sweep(dm, g2, hudson_formula, r0, rows)
            for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                H.assert_bits_equal(bufs[k].to_numpy(np.float64, rows), hs.sites[k], f"{k} {what}")
            for k, v in hs.totals.items():
                got = getattr(tot, k)
                if isinstance(v, int):
                    assert got == v, (what, k)
                else:
                    assert got == pytest.approx(v, rel=1e-12, abs=1e-300), (what, k)
        else:
            assert tot.sites_with_components == 0 and tot.numerator_sum == 0.0 and tot.site_num_sum == 0.0, what
    one = dev.Groups.from_haplotype_lists(dm, lists[:1])
    assert lib.fmh_pair_region_sweep(dm._h, one._h, 0, S, 0, 0, None, None, None, None) == _abi.FMH_ERR_INVALID
    assert lib.fmh_pair_region_sweep(dm._h, g2._h, 0, S, 7, 0, None, None, None, None) == _abi.FMH_ERR_INVALID
