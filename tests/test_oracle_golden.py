"""Pins the CPU oracle (oracle/ferromic_ref.py) against the known-answer vectors the reference's
own tests hold (tests/golden/reference_kats.json) and against brute-force pair enumeration
(the definition scikit-allel implements, which the reference's Python tests compare with)."""

import itertools
import math

import pytest

from oracle import ferromic_ref as R


def mk(variants):
    return [R.make_variant(v["pos"], v["g"]) for v in variants]


def haps(lst):
    return [(int(s), int(side)) for s, side in lst]


def close(actual, expected, abs_tol=None, rel_tol=None):
    if isinstance(expected, str):
        return (math.isnan(actual) if expected == "nan" else math.isinf(actual))
    if abs_tol is not None:
        return abs(actual - expected) <= abs_tol
    return math.isclose(actual, expected, rel_tol=rel_tol or 1e-12, abs_tol=0.0)


def test_harmonic(kats):
    for c in kats["harmonic"]["cases"]:
        assert close(R.harmonic(c["n"]), c["expected"], abs_tol=c["abs_tol"])


def test_watterson_theta(kats):
    for c in kats["watterson_theta"]["cases"]:
        got = R.calculate_watterson_theta(c["S"], c["n"], c["L"])
        assert close(got, c["expected"], c.get("abs_tol"), c.get("rel_tol")), (c, got)


def test_count_segregating_sites(kats):
    for c in kats["count_segregating_sites"]["cases"]:
        assert R.count_segregating_sites(mk(c["variants"])) == c["expected"]


def test_segregating_sites_dense_sparse_parity(kats):
    k = kats["segregating_sites_population_dense_sparse_parity"]
    names = ["s0", "s1"]
    for c in k["cases"]:
        variants = mk(c["variants"])
        for dense in (False, True):
            matrix = R.DenseGenotypeMatrix.from_variants(variants, k["sample_count"]) if dense else None
            ctx = R.PopulationContext(0, haps(k["haplotypes"]), variants, names, 1, matrix, None)
            assert R.count_segregating_sites_for_population(ctx) == c["expected"]


def test_calculate_pi(kats):
    for c in kats["calculate_pi"]["cases"]:
        got = R.calculate_pi(mk(c["variants"]), haps(c["haplotypes"]), c["L"])
        if "expected" in c:
            assert close(got, c["expected"], c.get("abs_tol")), (c, got)
        if "expected_gt" in c:
            assert got > c["expected_gt"]
        if "expected_lt" in c:
            assert got < c["expected_lt"]


def _ctx_pair(k, variants, L, variants2=None):
    names = [f"s{i}" for i in range(k["sample_count"])]
    p1 = R.PopulationContext(0, haps(k["pop1"]), variants, names, L)
    p2 = R.PopulationContext(1, haps(k["pop2"]), variants2 if variants2 is not None else variants, names, L)
    return p1, p2


def test_hudson_per_site(kats):
    k = kats["hudson_per_site"]
    for c in k["cases"]:
        p1, p2 = _ctx_pair(k, mk(c["variants"]), c["L"])
        outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(*c["region"]))
        by_pos = {s.position: s for s in sites}
        for e in c["sites"]:
            s = by_pos[e["position"]]
            for key, attr in (("fst", "fst"), ("num", "num_component"), ("den", "den_component"),
                              ("dxy", "d_xy"), ("pi1", "pi_pop1"), ("pi2", "pi_pop2")):
                if key in e:
                    assert abs(getattr(s, attr) - e[key]) < 1e-12, (c["name"], key)
        assert abs(outcome.fst - c["regional_fst"]) < 1e-12
        assert abs(R.aggregate_hudson_from_sites(sites) - c["regional_fst"]) < 1e-12
        # the region-less entry point must agree (sparse path, stats.rs:3490-3503)
        assert abs(R.calculate_hudson_fst_for_pair(p1, p2).fst - c["regional_fst"]) < 1e-12


def test_hudson_degenerate(kats):
    k = kats["hudson_degenerate"]
    p1, p2 = _ctx_pair(k, [], k["no_variants"]["L"])
    assert R.calculate_hudson_fst_for_pair(p1, p2).fst is None
    c = k["no_variants_with_region"]
    p1, p2 = _ctx_pair(k, [], c["L"])
    outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(*c["region"]))
    assert len(sites) == 0 and outcome.fst is None
    c = k["incompatible"]
    p1, p2 = _ctx_pair(k, mk(c["variants1"]), c["L"], mk(c["variants2"]))
    assert R.calculate_hudson_fst_per_site(p1, p2, R.QueryRegion(*c["region"])) == []
    with pytest.raises(R.VcfError):
        R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(*c["region"]))


def test_hudson_pi_dxy_consistency(kats):
    k = kats["hudson_pi_dxy_consistency"]
    p1, p2 = _ctx_pair(k, mk(k["variants"]), k["L"])
    outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(*k["region"]))
    vs = [s for s in sites if s.fst is not None]
    assert len(vs) == 2
    L = float(k["L"])
    assert abs(outcome.pi_pop1 - sum(s.pi_pop1 for s in vs) / L) < 1e-12
    assert abs(outcome.pi_pop2 - sum(s.pi_pop2 for s in vs) / L) < 1e-12
    assert abs(outcome.d_xy - sum(s.d_xy for s in vs) / L) < 1e-12


def test_hudson_missing_data(kats):
    k = kats["hudson_missing_data"]
    p1, p2 = _ctx_pair(k, mk(k["variants"]), k["L"])
    assert R.calculate_hudson_fst_for_pair(p1, p2).fst > k["expected_fst_gt"]


def test_hudson_from_summaries(kats):
    for c in kats["hudson_from_summaries"]["cases"]:
        s1 = R.DensePopulationSummary(c["alt1"], c["called1"], c["cap1"], 1, 1.0)
        s2 = R.DensePopulationSummary(c["alt2"], c["called2"], c["cap2"], 1, 1.0)
        h = [(0, 0), (0, 1)]
        p1 = R.PopulationContext(0, h, [], [], c["L"], None, s1)
        p2 = R.PopulationContext(1, h, [], [], c["L"], None, s2)
        got = R.calculate_d_xy_hudson(p1, p2)
        assert got == R.dxy_from_summaries(s1, s2, c["L"]) or (got is None and c["expected_dxy"] is None)
        if c["expected_dxy"] is None:
            assert got is None
        else:
            assert abs(got - c["expected_dxy"]) < c["abs_tol"]


# ---- brute-force definitions (what scikit-allel's mean_pairwise_difference[_between] compute) ----


def _called(variant, hap_list):
    out = []
    for s, side in hap_list:
        g = variant.genotypes.get(s)
        if g is not None and side < len(g):
            out.append(g[side])
    return out


def brute_mpd(variant, hap_list):
    a = _called(variant, hap_list)
    if len(a) < 2:
        return None
    pairs = list(itertools.combinations(a, 2))
    return sum(1 for x, y in pairs if x != y) / len(pairs)


def brute_mpd_between(variant, h1, h2):
    a, b = _called(variant, h1), _called(variant, h2)
    if not a or not b:
        return None
    return sum(1 for x in a for y in b if x != y) / (len(a) * len(b))


def _two_pops(k):
    h1 = [(s, side) for s in k["pop1_samples"] for side in (0, 1)]
    h2 = [(s, side) for s in k["pop2_samples"] for side in (0, 1)]
    return h1, h2


def test_hudson_scikit_allel_dataset(kats):
    """src/pytests/test_hudson_fst_integration.py: allel.hudson_fst num = dxy - (mpd1+mpd2)/2,
    den = dxy (per variant); fst = sum(num)/sum(den); d_xy = sum(den)/L."""
    k = kats["hudson_scikit_allel_dataset"]
    variants = mk(k["variants"])
    h1, h2 = _two_pops(k)
    p1 = R.PopulationContext("pop1", h1, variants, k["sample_names"], k["L"])
    p2 = R.PopulationContext("pop2", h2, variants, k["sample_names"], k["L"])
    outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(0, len(variants) - 1))
    nums, dens = [], []
    for v, s, e in zip(variants, sites, k["sites"]):
        den = brute_mpd_between(v, h1, h2)
        num = den - 0.5 * (brute_mpd(v, h1) + brute_mpd(v, h2))
        nums.append(num)
        dens.append(den)
        assert s.position == v.position + 1
        assert s.num_component == pytest.approx(num, rel=1e-12)
        assert s.den_component == pytest.approx(den, rel=1e-12)
        assert s.fst == pytest.approx(num / den, rel=1e-12)
        assert s.d_xy == pytest.approx(e["dxy"], rel=1e-12)
        assert s.pi_pop1 == pytest.approx(e["pi1"], abs=1e-15)
        assert s.pi_pop2 == pytest.approx(e["pi2"], rel=1e-12)
        assert s.num_component == pytest.approx(e["num"], rel=1e-12)
    assert outcome.fst == pytest.approx(sum(nums) / sum(dens), rel=1e-12)
    assert outcome.fst == pytest.approx(k["fst"], rel=1e-12)
    res = R.calculate_hudson_fst_for_pair(p1, p2)
    assert res.fst == pytest.approx(k["fst"], rel=1e-12)
    assert res.d_xy == pytest.approx(sum(dens) / k["L"], rel=1e-12)
    assert res.d_xy == pytest.approx(k["dxy"], rel=1e-12)


def test_diversity_scikit_allel_dataset(kats):
    """src/pytests/test_diversity_integration.py: nansum(mean_pairwise_difference)/L etc."""
    k = kats["diversity_scikit_allel_dataset"]
    variants = mk(k["variants"])
    h1, h2 = _two_pops(k)
    L = k["L"]
    for hl, key in ((h1, "pop1_pi"), (h2, "pop2_pi"), (h1 + h2, "combined_pi")):
        expected = sum(x for x in (brute_mpd(v, hl) for v in variants) if x is not None) / L
        assert R.calculate_pi(variants, hl, L) == pytest.approx(expected, rel=1e-12)
        assert expected == pytest.approx(k[key], rel=1e-12)
    sites = R.calculate_per_site_diversity(variants, h1, R.QueryRegion(0, L - 1))
    by_pos = {s.position: s for s in sites}
    for v, e in zip(variants, k["pop1_site_pi"]):
        assert by_pos[v.position + 1].pi == pytest.approx(brute_mpd(v, h1) or 0.0, rel=1e-12)
        assert by_pos[v.position + 1].pi == pytest.approx(e, rel=1e-12)
    p1 = R.PopulationContext("pop1", h1, variants, k["sample_names"], L)
    p2 = R.PopulationContext("pop2", h2, variants, k["sample_names"], L)
    expected = sum(x for x in (brute_mpd_between(v, h1, h2) for v in variants) if x is not None) / L
    assert R.calculate_d_xy_hudson(p1, p2) == pytest.approx(expected, rel=1e-12)
    assert expected == pytest.approx(k["dxy"], rel=1e-12)


def test_adjusted_length_and_inversion_freq(kats):
    for c in kats["adjusted_sequence_length"]["cases"]:
        allow = [tuple(x) for x in c["allow"]] if c["allow"] else None
        mask = [tuple(x) for x in c["mask"]] if c["mask"] else None
        assert R.calculate_adjusted_sequence_length(c["start"], c["end"], allow, mask) == c["expected"]
    for c in kats["inversion_allele_frequency"]["cases"]:
        m = {k: tuple(v) for k, v in c["map"].items()}
        assert R.calculate_inversion_allele_frequency(m) == pytest.approx(c["expected"])


def test_wc_analytic(kats):
    """W&C has no reference test (parity unpinned by tests): analytic vectors from the
    stats.rs:2034-2127 formulas."""
    for c in kats["wc_analytic"]["cases"]:
        n1, n2 = c["n"]
        a1, a2 = c["alt"]
        # n haplotypes per group laid out as diploid samples
        g = []
        for n, a in ((n1, a1), (n2, a2)):
            alleles = [1] * a + [0] * (n - a)
            g += [alleles[i:i + 2] for i in range(0, n, 2)]
        names = [f"s{i}" for i in range(len(g))]
        groups = {names[i]: ((0, 0) if i < n1 // 2 else (1, 1)) for i in range(len(g))}
        res = R.calculate_fst_wc_haplotype_groups([R.make_variant(5, g)], names, groups, R.QueryRegion(0, 10))
        site = res.site_fst[0]
        assert site.position == 6
        assert site.variance_components[0] == pytest.approx(c["a"], abs=1e-12)
        assert site.variance_components[1] == pytest.approx(c["b"], abs=1e-12)
        assert site.overall_fst.state == c["state"]
        assert res.overall_fst.state == c["state"]
        assert res.overall_fst.sites == 1
        if c["fst"] is None:
            assert site.overall_fst.value is None
        else:
            assert site.overall_fst.value == pytest.approx(c["fst"], rel=1e-12)
            assert res.pairwise_fst["0_vs_1"].value == pytest.approx(c["fst"], rel=1e-12)


def test_dense_paths_agree_with_sparse_random():
    """Internal consistency of the restatement: summary / dense / sparse paths on one random
    cohort (biallelic with missing; multi-allelic)."""
    import random

    rng = random.Random(7)
    N = 12
    for max_allele in (1, 3):
        rows = []
        for s in range(40):
            row = []
            for i in range(N):
                if rng.random() < 0.1:
                    row.append(None)
                else:
                    row.append([rng.randint(0, max_allele), rng.randint(0, max_allele)])
            rows.append(row)
        variants = [R.make_variant(10 * i, r) for i, r in enumerate(rows)]
        names = [f"s{i}" for i in range(N)]
        h1 = [(s, side) for s in range(0, 6) for side in (0, 1)]
        h2 = [(s, side) for s in range(6, 12) for side in (0, 1)]
        L = 500
        matrix = R.DenseGenotypeMatrix.from_variants(variants, N)
        sparse = (R.PopulationContext(0, h1, variants, names, L), R.PopulationContext(1, h2, variants, names, L))
        dense = (R.PopulationContext(0, h1, variants, names, L, matrix), R.PopulationContext(1, h2, variants, names, L, matrix))
        o_s = R.calculate_hudson_fst_for_pair(*sparse)
        o_d = R.calculate_hudson_fst_for_pair(*dense)
        for f in ("fst", "d_xy", "pi_pop1", "pi_pop2"):
            assert getattr(o_s, f) == pytest.approx(getattr(o_d, f), rel=1e-12)
        assert R.count_segregating_sites_for_population(sparse[0]) == R.count_segregating_sites_for_population(dense[0])
        if max_allele <= 1:
            summ = tuple(R.population_context_like_lib(i, v.variants, v.haplotypes, names, L, matrix) for i, v in enumerate(dense))
            o_q = R.calculate_hudson_fst_for_pair(*summ)
            assert o_q.fst == pytest.approx(o_s.fst, rel=1e-12)
            assert o_q.d_xy == pytest.approx(o_s.d_xy, rel=1e-12)
            # quirk 4 (stats.rs:1591-1609): the summaries path drops pi_k at sites where the
            # OTHER population has n<2, so pi may differ from the per-population value.
            assert R.calculate_pi_for_population(summ[0]) == pytest.approx(o_s.pi_pop1, rel=1e-12)


def test_hudson_reference_property_cases(kats):
    """src/tests/hudson_fst_tests.rs:20-298, 1009-1100: the reference's range assertions on single-site cohorts."""
    k = kats["hudson_properties"]
    h1, h2 = [tuple(h) for h in k["pop1"]], [tuple(h) for h in k["pop2"]]
    for c in k["cases"]:
        variants = [R.make_variant(v["pos"], v["g"]) for v in c["variants"]]
        p1 = R.PopulationContext(0, h1, variants, k["sample_names"], c["L"])
        p2 = R.PopulationContext(1, h2, variants, k["sample_names"], c["L"])
        if "region" not in c:
            fst = R.calculate_hudson_fst_for_pair(p1, p2).fst
            assert fst is not None and c["fst_min"] <= fst <= c["fst_max"], c["name"]
            continue
        outcome, sites = R.calculate_hudson_fst_for_pair_with_sites(p1, p2, R.QueryRegion(*c["region"]))
        if "site_position" in c:
            site = next(s for s in sites if s.position == c["site_position"])
            assert abs(site.d_xy - c["dxy"]) < c["tol"] and abs(site.pi_pop1 - c["pi1"]) < c["tol"] and abs(site.pi_pop2 - c["pi2"]) < c["tol"]
            assert abs(outcome.fst - site.fst) < c["tol"]
        else:
            assert len(sites) == c["n_sites"]
            s = sites[0]
            assert s.d_xy is not None and s.pi_pop1 is not None and s.pi_pop2 is not None and s.fst is not None
            assert c["site_fst_min"] <= s.fst <= c["site_fst_max"]


def test_pairwise_differences_reference_cases(kats):
    """src/tests/stats_tests.rs:368-470, literally: per-haplotype mismatch counts and comparable sites of sample pairs."""
    for case in kats["pairwise_differences"]["cases"]:
        variants = [R.make_variant(pos, genos) for pos, genos in case["variants"]]
        res = R.calculate_pairwise_differences(variants, case["sample_count"], case["sequence_length"])
        assert len(res) == case["result_len"]
        got = {f"{i},{j}": [d, c] for (i, j), d, c in res}
        for key, exp in case["expected"].items():
            assert got[key] == exp, (case["name"], key)
