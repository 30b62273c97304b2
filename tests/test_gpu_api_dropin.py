"""Drop-in API parity: the `ferromic` module (ferromic._core, C++ -> C-ABI -> HIP) exercised the way the
reference's own Python tests do (src/pytests/test_ferromic.py, test_hudson_fst_integration.py,
test_diversity_integration.py, src/pybenches/test_population_statistics_benchmarks.py), with the
CPU oracle standing in for scikit-allel (absent here)."""

import math
import random

import numpy as np
import pytest

import ferromic as fm
from oracle import ferromic_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def variant_record(pos, calls):
    """One variant as the module's input coercion takes it (lib.rs:825-1367): a mapping with `position` and per-sample `genotypes`."""
    return dict(position=pos, genotypes=calls)


build_variant = variant_record


# ---- src/pytests/test_ferromic.py ----------------------------------------------------------------


def test_segregating_sites_on_the_reference_vectors(kats):
    for case in kats["count_segregating_sites"]["cases"]:
        records = [variant_record(v["pos"], v["g"]) for v in case["variants"]]
        assert fm.segregating_sites(records) == case["expected"], case


def test_population_from_numpy_accepts_python_positions(kats):
    k = kats["from_numpy_echo"]
    population = fm.Population.from_numpy(
        "demo",
        genotypes=np.array(k["genotypes"], dtype=np.uint8),
        positions=k["positions"],
        haplotypes=[tuple(h) for h in k["haplotypes"]],
        sequence_length=k["sequence_length"],
        sample_names=k["sample_names"],
    )
    assert population.variant_count == k["variant_count"]
    assert population.sample_names == k["sample_names"]
    assert population.haplotypes == [(0, 0), (0, 1)]
    assert population.segregating_sites() == 0  # sample 0 is 0|0
    assert population.nucleotide_diversity() == 0.0


# ---- src/pytests/test_hudson_fst_integration.py ---------------------------------------------------


def _population_dict(pop_id, sample_indices, variants, L, names):
    return {
        "id": pop_id,
        "haplotypes": [(s, side) for s in sample_indices for side in (0, 1)],
        "variants": variants,
        "sequence_length": L,
        "sample_names": names,
    }


def test_hudson_fst_matches_reference_dataset(kats):
    k = kats["hudson_scikit_allel_dataset"]
    variants = [build_variant(v["pos"], v["g"]) for v in k["variants"]]
    p1 = _population_dict("pop1", k["pop1_samples"], variants, k["L"], k["sample_names"])
    p2 = _population_dict("pop2", k["pop2_samples"], variants, k["L"], k["sample_names"])
    result = fm.hudson_fst(p1, p2)
    assert result.fst == pytest.approx(k["fst"], rel=1e-12)
    assert result.d_xy == pytest.approx(k["dxy"], rel=1e-12)
    assert result.population1_label == "pop1" and result.population1_haplotype_group is None
    result2, sites = fm.hudson_fst_with_sites(p1, p2, (0, len(variants) - 1))
    assert result2.fst == pytest.approx(k["fst"], rel=1e-12)
    informative = [s for s in sites if s.numerator_component is not None and s.denominator_component is not None]
    assert len(informative) == len(variants)
    for idx, (site, e) in enumerate(zip(informative, k["sites"])):
        assert site.position == variants[idx]["position"] + 1
        assert site.numerator_component == pytest.approx(e["num"], rel=1e-12)
        assert site.denominator_component == pytest.approx(e["dxy"], rel=1e-12)
        assert site.fst == pytest.approx(e["num"] / e["dxy"], rel=1e-12)
    assert [s.position for s in fm.hudson_fst_sites(p1, p2, (1, 2))] == [2, 3]


# ---- src/pytests/test_diversity_integration.py ----------------------------------------------------


def test_diversity_dataset(kats):
    k = kats["diversity_scikit_allel_dataset"]
    variants = [build_variant(v["pos"], v["g"]) for v in k["variants"]]
    h1 = [(s, side) for s in k["pop1_samples"] for side in (0, 1)]
    h2 = [(s, side) for s in k["pop2_samples"] for side in (0, 1)]
    L = k["L"]
    assert fm.nucleotide_diversity(variants, h1, L) == pytest.approx(k["pop1_pi"], rel=1e-12)
    assert fm.nucleotide_diversity(variants, h2, L) == pytest.approx(k["pop2_pi"], rel=1e-12)
    assert fm.nucleotide_diversity(variants, h1 + h2, L) == pytest.approx(k["combined_pi"], rel=1e-12)
    sites = fm.per_site_diversity(variants, h1, (0, L - 1))
    by_pos = {s.position: s for s in sites}
    for v, e in zip(k["variants"], k["pop1_site_pi"]):
        assert by_pos[v["pos"] + 1].pi == pytest.approx(e, rel=1e-12)
    p1 = _population_dict("pop1", k["pop1_samples"], variants, L, k["sample_names"])
    p2 = _population_dict("pop2", k["pop2_samples"], variants, L, k["sample_names"])
    assert fm.hudson_dxy(p1, p2).d_xy == pytest.approx(k["dxy"], rel=1e-12)
    # region=None -> [min position, max position] (lib.rs:1412-1441)
    assert [s.position for s in fm.per_site_diversity(variants, h1)] == [1, 4, 6, 8]


# ---- reference Rust KATs through the Python surface ------------------------------------------------


def test_hudson_rust_kats_through_api(kats):
    k = kats["hudson_per_site"]
    names = [f"s{i}" for i in range(k["sample_count"])]
    for c in k["cases"]:
        variants = [(v["pos"], v["g"]) for v in c["variants"]]  # tuple form (lib.rs:836-848)
        p1 = fm.Population(0, variants, [tuple(h) for h in k["pop1"]], c["L"], names)
        p2 = fm.Population(1, variants, [tuple(h) for h in k["pop2"]], c["L"], names)
        result, sites = fm.hudson_fst_with_sites(p1, p2, tuple(c["region"]))
        by_pos = {s.position: s for s in sites}
        for e in c["sites"]:
            s = by_pos[e["position"]]
            for key, attr in (("fst", "fst"), ("num", "numerator_component"), ("den", "denominator_component"),
                              ("dxy", "d_xy"), ("pi1", "pi_pop1"), ("pi2", "pi_pop2")):
                if key in e:
                    assert abs(getattr(s, attr) - e[key]) < 1e-12, (c["name"], key)
        assert abs(result.fst - c["regional_fst"]) < 1e-12
        assert abs(fm.hudson_fst(p1, p2).fst - c["regional_fst"]) < 1e-12
        assert result.population1_label == "haplotype_group_0" and result.population2_haplotype_group == 1


def test_hudson_degenerate_and_errors(kats):
    k = kats["hudson_degenerate"]
    names = [f"s{i}" for i in range(4)]
    h1, h2 = [tuple(h) for h in k["pop1"]], [tuple(h) for h in k["pop2"]]
    p1, p2 = fm.Population(0, [], h1, 1000, names), fm.Population(1, [], h2, 1000, names)
    assert fm.hudson_fst(p1, p2).fst is None
    res, sites = fm.hudson_fst_with_sites(fm.Population(0, [], h1, 3, names), fm.Population(1, [], h2, 3, names), (100, 102))
    assert sites == [] and res.fst is None
    c = k["incompatible"]
    q1 = fm.Population(0, [(v["pos"], v["g"]) for v in c["variants1"]], h1, 2, names)
    q2 = fm.Population(1, [(v["pos"], v["g"]) for v in c["variants2"]], h2, 2, names)
    assert fm.hudson_fst_sites(q1, q2, tuple(c["region"])) == []
    with pytest.raises(ValueError, match="VCF error: Parse"):
        fm.hudson_fst_with_sites(q1, q2, tuple(c["region"]))
    with pytest.raises(ValueError, match="Sequence length mismatch"):
        fm.hudson_fst(fm.Population(0, [], h1, 5, names), fm.Population(1, [], h2, 6, names))
    with pytest.raises(ValueError, match="region end"):
        fm.hudson_fst_sites(q1, q1, (5, 1))
    with pytest.raises(ValueError, match="at least two haplotypes"):
        fm.per_site_diversity([(1, [[0, 1]])], [(0, 0)])
    with pytest.raises(ValueError, match="sample_names must contain"):
        fm.wc_fst([], [], {}, (0, 1))


# ---- from_numpy populations: summary / dense / sparse selection (lib.rs:777-799, stats.rs:3435-3599) ----


def _oracle_pops(genotypes, positions, hap_lists, L, names):
    variants, dense = R.convert_numeric_array(genotypes, positions)
    return [R.population_context_like_lib(i, variants, hl, names, L, dense) for i, hl in enumerate(hap_lists)]


def _check_pair(fm_pops, or_pops):
    for a, b in zip(fm_pops, or_pops):
        assert a.segregating_sites() == R.count_segregating_sites_for_population(b)
        got, exp = a.nucleotide_diversity(), R.calculate_pi_for_population(b)
        assert H.rel_close(got, exp), (got, exp)
    exp = R.calculate_hudson_fst_for_pair(or_pops[0], or_pops[1])
    got = fm.hudson_fst(fm_pops[0], fm_pops[1])
    for f in ("fst", "d_xy", "pi_pop1", "pi_pop2", "pi_xy_avg"):
        e, g = getattr(exp, f), getattr(got, f)
        assert (e is None) == (g is None), f
        if e is not None:
            assert H.rel_close(g, e), (f, g, e)
    assert (fm.hudson_dxy(fm_pops[0], fm_pops[1]).d_xy is None) == (R.calculate_d_xy_hudson(or_pops[0], or_pops[1]) is None)


@pytest.mark.parametrize("dtype,ploidy,max_allele,p_missing", [
    (np.uint8, 2, 1, 0.0),    # summary path (the benchmark path)
    (np.int8, 2, 1, 0.08),    # summary path with per-allele missing bits
    (np.int16, 2, 3, 0.05),   # dense general path (max_allele > 1)
    (np.uint16, 2, 2, 0.0),
    (np.int8, 1, 1, 0.1),     # ploidy != 2 -> no dense matrix -> sparse path
    (np.int8, 3, 2, 0.1),
])
def test_from_numpy_paths(dtype, ploidy, max_allele, p_missing):
    rng = np.random.default_rng(int(np.dtype(dtype).num) * 10 + ploidy)
    S, N = 150, 26
    g = rng.integers(0, max_allele + 1, size=(S, N, ploidy)).astype(dtype)
    g[rng.random((S, N, ploidy)) < 0.5] = 0
    if p_missing > 0:
        g[rng.random((S, N, ploidy)) < p_missing] = -1
    positions = np.cumsum(rng.integers(1, 50, size=S)).astype(np.int64)
    L = int(positions[-1] - positions[0] + 1)
    names = [f"sample_{i}" for i in range(N)]
    all_haps = [(s, side) for s in range(N) for side in range(min(ploidy, 2))]
    h1 = [h for h in all_haps if h[0] < N // 2]
    h2 = [h for h in all_haps if h[0] >= N // 2]
    base = fm.Population.from_numpy("all_samples", g, positions, all_haps, L, sample_names=names)
    p1 = base.with_haplotypes("population_1", h1)
    p2 = base.with_haplotypes("population_2", h2)
    o_all, o1, o2 = _oracle_pops(g, positions, [all_haps, h1, h2], L, names)
    assert base.segregating_sites() == R.count_segregating_sites_for_population(o_all)
    assert H.rel_close(base.nucleotide_diversity(), R.calculate_pi_for_population(o_all))
    _check_pair([p1, p2], [o1, o2])
    # region-restricted per-site path on the same populations (always the sparse model)
    reg = (int(positions[20]), int(positions[100]))
    exp_o, exp_sites = R.calculate_hudson_fst_for_pair_with_sites(o1, o2, R.QueryRegion(*reg))
    got_o, got_sites = fm.hudson_fst_with_sites(p1, p2, reg)
    assert len(got_sites) == len(exp_sites) == 81
    for gs, es in zip(got_sites, exp_sites):
        assert gs.position == es.position and gs.n1_called == es.n1_called and gs.n2_called == es.n2_called
        for ga, ea in ((gs.fst, es.fst), (gs.d_xy, es.d_xy), (gs.pi_pop1, es.pi_pop1),
                       (gs.numerator_component, es.num_component), (gs.denominator_component, es.den_component)):
            assert (ga is None) == (ea is None)
            if ea is not None:
                assert ga == ea  # bit-exact per-site values
    for f in ("fst", "d_xy", "pi_pop1", "pi_pop2"):
        e, gq = getattr(exp_o, f), getattr(got_o, f)
        assert (e is None) == (gq is None)
        if e is not None:
            assert H.rel_close(gq, e), f


def test_synthetic_benchmark_cohort_like_pybench():
    """src/pybenches/test_population_statistics_benchmarks.py:113-261 recipe at 4096 x 96."""
    S, N, scale = 4096, 96, 0.05
    rng = np.random.default_rng(seed=S + N)
    half = N // 2
    base_freq = rng.beta(0.8, 0.8, size=S)
    divergence = rng.normal(0.0, scale, size=S)
    f1 = np.clip(base_freq + divergence, 0.001, 0.999)
    f2 = np.clip(base_freq - divergence, 0.001, 0.999)
    h1m = rng.binomial(1, f1[:, None], size=(S, half * 2)).astype(np.int8)
    h2m = rng.binomial(1, f2[:, None], size=(S, half * 2)).astype(np.int8)
    genotypes = np.concatenate([h1m.reshape(S, half, 2), h2m.reshape(S, half, 2)], axis=1)
    genotypes[0, :half, :] = 0
    genotypes[0, half:, :] = 1
    genotypes[1, :half, 0] = 0
    genotypes[1, :half, 1] = 1
    genotypes[1, half:, :] = 1
    positions = np.cumsum(rng.integers(1, 50, size=S, dtype=np.int64), dtype=np.int64)
    L = int(positions[-1]) + 1 - int(positions[0])
    haps = [(s, side) for s in range(N) for side in (0, 1)]
    names = [f"sample_{i}" for i in range(N)]
    population = fm.Population.from_numpy("all_samples", genotypes, positions, haps, L, sample_names=names)
    p1 = population.with_haplotypes("population_1", [h for h in haps if h[0] < half])
    p2 = population.with_haplotypes("population_2", [h for h in haps if h[0] >= half])
    # the oracle's dense summary path on the same matrix, C restatement for speed
    from oracle import dense as D

    flat = genotypes.astype(np.uint8).reshape(-1)
    off_all = np.arange(2 * N, dtype=np.uint64)
    out = D.hudson_sweep(flat, None, S, 2 * N, off_all[: 2 * half], off_all[2 * half:], 2)
    both = D.hudson_sweep(flat, None, S, 2 * N, off_all, off_all[:2], 1)
    assert population.segregating_sites() == both.pop[0]["segregating_sites"]
    assert H.rel_close(population.nucleotide_diversity(), both.pop[0]["pi_sum"] / L)
    theta = fm.watterson_theta(population.segregating_sites(), 2 * N, L)
    assert H.rel_close(theta, R.calculate_watterson_theta(both.pop[0]["segregating_sites"], 2 * N, L))
    res = fm.hudson_fst(p1, p2)
    t = out.totals
    assert H.rel_close(res.fst, t["numerator_sum"] / t["denominator_sum"])
    assert H.rel_close(res.d_xy, t["dxy_sum_all"] / L)
    assert H.rel_close(res.pi_pop1, t["pi1_sum"] / L) and H.rel_close(res.pi_pop2, t["pi2_sum"] / L)
    assert H.rel_close(p1.nucleotide_diversity(), out.pop[0]["pi_sum"] / L)
    assert p2.segregating_sites() == out.pop[1]["segregating_sites"]


# ---- Weir & Cockerham through the API ---------------------------------------------------------------


def _est_equal(got, exp):
    assert got.state == exp.state and got.sites == exp.sites
    assert (got.value is None) == (exp.value is None)
    if exp.value is not None:
        assert H.rel_close(got.value, exp.value) or (math.isinf(got.value) and got.value == exp.value)
    assert H.rel_close(got.sum_a, exp.sum_a) and H.rel_close(got.sum_b, exp.sum_b)


@pytest.mark.parametrize("G,max_allele,p_missing", [(2, 1, 0.0), (2, 1, 0.2), (4, 1, 0.05), (3, 3, 0.1), (1, 1, 0.1),
                                                  (9, 1, 0.0), (11, 1, 0.05), (13, 3, 0.1)])  # > 8 groups: counts path
def test_wc_fst_api(G, max_allele, p_missing):
    rng = random.Random(G * 100 + max_allele)
    S, N = 90, 24
    variants = H.random_sparse_variants(rng, S, N, max_allele, p_missing, 0.05, 1)
    names = [f"EUR_GBR_HG{i:05d}" for i in range(N)]
    sample_to_group = {names[i].rsplit("_", 1)[-1] + ("_L" if i % 2 else ""): (i % G, (i + (i % 7 == 0)) % G) for i in range(N - 3)}
    region = (variants[5].position, variants[-3].position)
    py_variants = [build_variant(v.position, [g for g in v.genotypes]) for v in variants]
    got = fm.wc_fst(py_variants, names, sample_to_group, region)
    exp = R.calculate_fst_wc_haplotype_groups(variants, names, sample_to_group, R.QueryRegion(*region))
    assert got.fst_type == "haplotype_groups"
    _est_equal(got.overall_fst, exp.overall_fst)
    assert set(got.pairwise_fst) == set(exp.pairwise_fst)
    for key in exp.pairwise_fst:
        _est_equal(got.pairwise_fst[key], exp.pairwise_fst[key])
        assert H.rel_close(got.pairwise_variance_components[key][0], exp.pairwise_variance_components[key][0])
    assert len(got.site_fst) == len(exp.site_fst)
    for gs, es in zip(got.site_fst, exp.site_fst):
        assert gs.position == es.position
        assert gs.overall_fst.state == es.overall_fst.state
        assert gs.variance_components() == es.variance_components
        assert gs.population_sizes == es.population_sizes
        assert gs.pairwise_variance_components == es.pairwise_variance_components
        assert {k: v.state for k, v in gs.pairwise_fst.items()} == {k: v.state for k, v in es.pairwise_fst.items()}
    assert fm.wc_fst_components(got.overall_fst) == got.overall_fst.components()


def test_wc_analytic_through_api(kats):
    for c in kats["wc_analytic"]["cases"]:
        n1, n2 = c["n"]
        a1, a2 = c["alt"]
        g = []
        for n, a in ((n1, a1), (n2, a2)):
            alleles = [1] * a + [0] * (n - a)
            g += [alleles[i:i + 2] for i in range(0, n, 2)]
        names = [f"s{i}" for i in range(len(g))]
        groups = {names[i]: ((0, 0) if i < n1 // 2 else (1, 1)) for i in range(len(g))}
        res = fm.wc_fst([(5, g)], names, groups, (0, 10))
        site = res.site_fst[0]
        assert site.position == 6
        assert site.variance_components_a == pytest.approx(c["a"], abs=1e-12)
        assert site.variance_components_b == pytest.approx(c["b"], abs=1e-12)
        assert site.overall_fst.state == c["state"] == res.overall_fst.state
        if c["fst"] is not None:
            assert res.pairwise_fst["0_vs_1"].value == pytest.approx(c["fst"], rel=1e-12)


# ---- pairwise differences (stats_tests.rs:368-470 cases + random cohorts) -----------------------------


@pytest.mark.parametrize("N,S,max_allele,p_missing,p_haploid", [(3, 3, 1, 0.0, 0.0), (9, 150, 1, 0.1, 0.1), (70, 300, 3, 0.05, 0.05),
                                                                 (130, 64, 2, 0.0, 0.0), (150, 200, 2, 0.03, 0.02), (20, 60, 40, 0.05, 0.0)])
def test_pairwise_differences(N, S, max_allele, p_missing, p_haploid):
    rng = random.Random(N * 31 + S)
    variants = H.random_sparse_variants(rng, S, N, max_allele, p_missing, p_haploid, 1 if p_missing else 0)
    py_variants = [build_variant(v.position, [g for g in v.genotypes]) for v in variants]
    L = variants[-1].position + 5
    for n in (N, N + 2, max(N - 1, 2)):
        got = fm.pairwise_differences(py_variants, n, L)
        exp = R.calculate_pairwise_differences(variants, n, L)
        assert len(got) == len(exp) == n * (n - 1) // 2
        for g, ((i, j), d, c) in zip(got, exp):
            assert (g.sample_i, g.sample_j, g.differences, g.comparable_sites) == (i, j, d, c)
    assert fm.pairwise_differences([], 3, 10)[0].comparable_sites == 0
    with pytest.raises(ValueError, match="sequence_length"):
        fm.pairwise_differences(py_variants, N, 0)


def test_pairwise_differences_reference_kat():
    """src/tests/stats_tests.rs:368-470: three samples, three variants."""
    variants = [build_variant(1, [[0, 0], [0, 1], [1, 1]]), build_variant(2, [[0, 0], [0, 0], [0, 0]]),
                build_variant(3, [[0, 1], [1, 1], None])]
    res = {(p.sample_i, p.sample_j): (p.differences, p.comparable_sites) for p in fm.pairwise_differences(variants, 3, 10)}
    exp = {k: (d, c) for k, d, c in R.calculate_pairwise_differences([R.make_variant(v["position"], v["genotypes"]) for v in variants], 3, 10)}
    assert res == exp
    assert res[(0, 1)] == (4, 40) and res[(0, 2)][1] == 36  # one variant lacks sample 2 -> 4 comparisons removed


def test_concurrent_python_threads_share_populations():
    """The module releases the GIL around device calls (lib.rs `py.allow_threads`): threads hammering the same Population
    objects (lazy uploads, cached summaries, one sweep at a time per device inside the library) must all see the
    single-threaded answers."""
    import threading

    rng = np.random.default_rng(17)
    S, N = 3000, 40
    g = (rng.random((S, N, 2)) < rng.beta(0.8, 0.8, size=(S, 1, 1))).astype(np.int8)
    g[rng.random((S, N, 2)) < 0.02] = -1
    positions = np.cumsum(rng.integers(1, 40, size=S)).astype(np.int64)
    L = int(positions[-1] - positions[0] + 1)
    haps = [(s, side) for s in range(N) for side in (0, 1)]
    names = [f"s{i}" for i in range(N)]

    def fresh():
        base = fm.Population.from_numpy("all", g, positions, haps, L, sample_names=names)
        return base, base.with_haplotypes(1, haps[:N]), base.with_haplotypes(2, haps[N:])

    def answers(base, p1, p2):
        r = fm.hudson_fst(p1, p2)
        _, sites = fm.hudson_fst_with_sites(p1, p2, (int(positions[100]), int(positions[900])))
        return (base.segregating_sites(), base.nucleotide_diversity(), p1.nucleotide_diversity(), r.fst, r.d_xy,
                fm.hudson_dxy(p1, p2).d_xy, len(sites), sites[17].fst, sites[400].d_xy)

    expected = answers(*fresh())
    shared = fresh()   # untouched: the first uploads and summaries happen under contention
    results, errors = [], []

    def work():
        try:
            for _ in range(6):
                results.append(answers(*shared))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:1]
    assert len(results) == 48 and all(r == expected for r in results)


def test_hudson_reference_property_cases_through_api(kats):
    """src/tests/hudson_fst_tests.rs:20-298, 1009-1100 through the drop-in module."""
    k = kats["hudson_properties"]
    h1, h2 = [tuple(h) for h in k["pop1"]], [tuple(h) for h in k["pop2"]]
    for c in k["cases"]:
        variants = [build_variant(v["pos"], v["g"]) for v in c["variants"]]
        p1 = fm.Population(0, variants, h1, c["L"], k["sample_names"])
        p2 = fm.Population(1, variants, h2, c["L"], k["sample_names"])
        if "region" not in c:
            fst = fm.hudson_fst(p1, p2).fst
            assert fst is not None and c["fst_min"] <= fst <= c["fst_max"], c["name"]
            continue
        outcome, sites = fm.hudson_fst_with_sites(p1, p2, tuple(c["region"]))
        if "site_position" in c:
            site = next(s for s in sites if s.position == c["site_position"])
            assert abs(site.d_xy - c["dxy"]) < c["tol"] and abs(site.pi_pop1 - c["pi1"]) < c["tol"] and abs(site.pi_pop2 - c["pi2"]) < c["tol"]
            assert abs(outcome.fst - site.fst) < c["tol"]
        else:
            assert len(sites) == c["n_sites"]
            s = sites[0]
            assert None not in (s.d_xy, s.pi_pop1, s.pi_pop2, s.fst) and c["site_fst_min"] <= s.fst <= c["site_fst_max"]


def test_pairwise_differences_reference_cases_literal(kats):
    """The four pairwise tests of src/tests/stats_tests.rs:368-470 through the drop-in module (FP4 / int8 MFMA Gram)."""
    for case in kats["pairwise_differences"]["cases"]:
        variants = [build_variant(pos, genos) for pos, genos in case["variants"]]
        res = fm.pairwise_differences(variants, case["sample_count"], case["sequence_length"])
        assert len(res) == case["result_len"]
        got = {f"{p.sample_i},{p.sample_j}": [p.differences, p.comparable_sites] for p in res}
        for key, exp in case["expected"].items():
            assert got[key] == exp, (case["name"], key)


def test_records_with_numpy_genotypes_take_the_buffer_path():
    """A variant record whose `genotypes` is one numpy integer array (samples x ploidy, any of the integer dtypes, strided views too) is read
    through its buffer; the result must be what the same calls as nested lists give, and a value outside 0..255 the same OverflowError."""
    rng = np.random.default_rng(31)
    S, N = 400, 37
    calls = rng.integers(0, 3, size=(S, N, 2))
    haps = [(i, side) for i in range(N) for side in (0, 1)]
    as_lists = [variant_record(10 * i + 3, calls[i].tolist()) for i in range(S)]
    base = fm.per_site_diversity(as_lists, haps)
    for dtype in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.int64):
        arr = calls.astype(dtype)
        wide = np.zeros((S, N, 4), dtype=dtype)
        wide[:, :, ::2] = arr  # a strided view per record
        for records in ([variant_record(10 * i + 3, arr[i]) for i in range(S)], [variant_record(10 * i + 3, wide[i][:, ::2]) for i in range(S)],
                        [(10 * i + 3, arr[i]) for i in range(S)]):
            got = fm.per_site_diversity(records, haps)
            assert len(got) == len(base)
            for a, b in zip(got, base):
                assert a.position == b.position and a.pi == b.pi and a.watterson_theta == b.watterson_theta
    assert fm.segregating_sites([variant_record(1, np.array([0, 1, 1, 0], dtype=np.int8))]) == fm.segregating_sites([variant_record(1, [0, 1, 1, 0])])  # haploid calls
    with pytest.raises(OverflowError):
        fm.per_site_diversity([variant_record(5, np.array([[0, -1], [1, 0]], dtype=np.int8))], [(0, 0), (1, 1)])
    with pytest.raises(OverflowError):
        fm.per_site_diversity([variant_record(5, np.array([[0, 300], [1, 0]], dtype=np.int16))], [(0, 0), (1, 1)])
