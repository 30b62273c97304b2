"""Randomised drop-in parity: small cohorts in every shape the reference's Python surface accepts (None samples,
haploid ints, ragged ploidy, the 0xFF sentinel, ragged sample counts, duplicate / out-of-range haplotypes, absent
sample names, partial regions) go through `ferromic` (C++ host -> C-ABI -> HIP) and through the oracle's literal
restatement of src/stats.rs; every statistic of the path must agree (integers and per-site floats exactly, regional
floats to 1e-9)."""

import math
import os
import random

import pytest

import ferromic as fm
from oracle import ferromic_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("FERROMIC_FUZZ_CASES", "300"))


def random_cohort(rng: random.Random):
    S = rng.choice([0, 1, 2, 3, 5, 8, 12])
    N = rng.randint(1, 6) if rng.random() < 0.8 else rng.randint(7, 14)
    weird = rng.random() < 0.3
    variants = []
    pos = rng.randint(0, 5)
    for _ in range(S):
        n_g = N if not (weird and rng.random() < 0.15) else max(N - 1, 1)
        genos = []
        for _ in range(n_g):
            r = rng.random()

            def allele():
                if weird and rng.random() < 0.03:
                    return 255
                return rng.choice(([0, 0, 1, 1, 2] if rng.random() < 0.7 else [0, 1, 3, 5, 6]) if weird else [0, 1])

            if r < 0.10:
                genos.append(None)
            elif r < 0.18 and weird:
                genos.append(allele())           # haploid int
            elif r < 0.26 and weird:
                genos.append([allele()])
            elif r < 0.32 and weird:
                genos.append([allele(), allele(), allele()])
            else:
                genos.append([allele(), allele()])
        variants.append((pos, genos))
        pos += rng.randint(1, 9) if rng.random() > 0.05 else 0  # an occasional duplicate position
    L = pos + rng.randint(1, 20)
    names = [f"POP_S{i}" for i in range(N)] if rng.random() > 0.15 else []
    all_haps = [(s, side) for s in range(N) for side in (0, 1)]
    rng.shuffle(all_haps)
    cut = rng.randint(0, len(all_haps))
    h1, h2 = all_haps[:cut], all_haps[cut:]
    if rng.random() < 0.3 and h1:
        h1 = h1 + [h1[0], (N + 2, 0)]            # duplicate + out-of-range sample
    return variants, h1, h2, L, names, N


def as_oracle(variants):
    out = []
    for pos, genos in variants:
        out.append(R.make_variant(pos, [None if g is None else ([g] if isinstance(g, int) else list(g)) for g in genos]))
    return out


def same_float(a, b, exact=False):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, float) and isinstance(b, float) and (math.isnan(a) or math.isnan(b)):
        return math.isnan(a) and math.isnan(b)
    return a == b if exact else H.rel_close(a, b)


def call_both(fm_call, oracle_call):
    """Both sides raise, or both return."""
    try:
        exp = oracle_call()
    except R.VcfError as e:
        with pytest.raises(ValueError) as info:
            fm_call()
        assert str(info.value) == f"VCF error: {e}"
        return None, None
    return fm_call(), exp


@pytest.mark.parametrize("seed", range(CASES))
def test_random_cohort(seed):
    rng = random.Random(1000 + seed)
    variants, h1, h2, L, names, N = random_cohort(rng)
    ov = as_oracle(variants)
    dict_variants = [{"pos": p, "calls": g} for p, g in variants] if seed % 3 == 0 else variants

    assert fm.segregating_sites(dict_variants) == R.count_segregating_sites(ov)
    assert same_float(fm.nucleotide_diversity(dict_variants, h1, L), R.calculate_pi(ov, h1, L))

    p1, p2 = fm.Population(0, dict_variants, h1, L, names), fm.Population("second", variants, h2, L, names)
    o1, o2 = R.PopulationContext(0, h1, ov, names, L), R.PopulationContext("second", h2, ov, names, L)
    for a, b in ((p1, o1), (p2, o2)):
        assert a.segregating_sites() == R.count_segregating_sites_for_population(b)
        assert same_float(a.nucleotide_diversity(), R.calculate_pi_for_population(b))

    got, exp = call_both(lambda: fm.hudson_fst(p1, p2), lambda: R.calculate_hudson_fst_for_pair(o1, o2))
    if exp is not None:
        for f in ("fst", "d_xy", "pi_pop1", "pi_pop2", "pi_xy_avg"):
            assert same_float(getattr(got, f), getattr(exp, f)), f
    got, exp = call_both(lambda: fm.hudson_dxy(p1, p2).d_xy, lambda: R.calculate_d_xy_hudson(o1, o2))
    assert same_float(got, exp)

    lo = rng.randint(0, max(L // 2, 1))
    region = (lo, lo + rng.randint(0, L))
    got, exp = call_both(lambda: fm.hudson_fst_with_sites(p1, p2, region),
                         lambda: R.calculate_hudson_fst_for_pair_with_sites(o1, o2, R.QueryRegion(*region)))
    if exp is not None:
        (g_out, g_sites), (e_out, e_sites) = got, exp
        assert same_float(g_out.fst, e_out.fst)
        assert len(g_sites) == len(e_sites)
        for gs, es in zip(g_sites, e_sites):
            assert (gs.position, gs.n1_called, gs.n2_called) == (es.position, es.n1_called, es.n2_called)
            for ga, ea in ((gs.fst, es.fst), (gs.d_xy, es.d_xy), (gs.pi_pop1, es.pi_pop1), (gs.pi_pop2, es.pi_pop2),
                           (gs.numerator_component, es.num_component), (gs.denominator_component, es.den_component)):
                assert same_float(ga, ea, exact=True)
    assert len(fm.hudson_fst_sites(p1, p2, region)) == len(R.calculate_hudson_fst_per_site(o1, o2, R.QueryRegion(*region)))

    if len(h1) >= 2:
        use_region = region if (seed % 2 or not variants) else None
        if use_region is None:
            reg = R.QueryRegion(min(v.position for v in ov), max(v.position for v in ov))
        else:
            reg = R.QueryRegion(*use_region)
        e_div = R.calculate_per_site_diversity(ov, h1, reg)
        g_div = fm.per_site_diversity(dict_variants, h1, use_region)
        assert [(d.position) for d in g_div] == [d.position for d in e_div]
        for gd, ed in zip(g_div, e_div):
            assert same_float(gd.pi, ed.pi, exact=True) and same_float(gd.watterson_theta, ed.watterson_theta, exact=True)

    if names:
        groups = {}
        top_group = rng.choice([1, 2, 2, 4, 9])   # up to ten haplotype groups: beyond eight the counts path
        for i, name in enumerate(names):
            if rng.random() < 0.85:
                key = name if rng.random() < 0.7 else name.rsplit("_", 1)[-1]   # alias lookup (process.rs:1198-1241)
                groups[key] = (rng.randint(0, top_group), rng.randint(0, top_group))
        e_wc = R.calculate_fst_wc_haplotype_groups(ov, names, groups, R.QueryRegion(*region))
        g_wc = fm.wc_fst(dict_variants, names, groups, region)
        assert g_wc.overall_fst.state == e_wc.overall_fst.state and g_wc.overall_fst.sites == e_wc.overall_fst.sites
        assert same_float(g_wc.overall_fst.value, e_wc.overall_fst.value)
        assert set(g_wc.pairwise_fst) == set(e_wc.pairwise_fst)
        for key, e_est in e_wc.pairwise_fst.items():
            g_est = g_wc.pairwise_fst[key]
            assert (g_est.state, g_est.sites) == (e_est.state, e_est.sites), key
            assert same_float(g_est.value, e_est.value)
        assert len(g_wc.site_fst) == len(e_wc.site_fst)
        for gs, es in zip(g_wc.site_fst, e_wc.site_fst):
            assert gs.position == es.position and gs.overall_fst.state == es.overall_fst.state
            assert gs.variance_components() == es.variance_components
            assert gs.population_sizes == es.population_sizes
            assert gs.pairwise_variance_components == es.pairwise_variance_components

    n = rng.randint(0, N + 1)
    e_pd = R.calculate_pairwise_differences(ov, n, L)
    g_pd = fm.pairwise_differences(dict_variants, n, L)
    assert [(p.sample_i, p.sample_j, p.differences, p.comparable_sites) for p in g_pd] == [(i, j, d, c) for (i, j), d, c in e_pd]


NUMPY_CASES = int(os.environ.get("FERROMIC_FUZZ_NUMPY_CASES", "120"))


@pytest.mark.parametrize("seed", range(NUMPY_CASES))
def test_random_numpy_population(seed):
    """Population.from_numpy over random dtypes / ploidies / missing rates / allele ranges (incl. 255 and empty shapes):
    summary, dense and sparse arms of lib.rs:777-799 + stats.rs:3435-3599 against the oracle's same selection."""
    import numpy as np

    rng = np.random.default_rng(5000 + seed)
    dtype = [np.uint8, np.int8, np.uint16, np.int16][seed % 4]
    S = int(rng.choice([0, 1, 2, 7, 33]))
    N = int(rng.integers(1, 9))
    ploidy = int(rng.choice([1, 2, 2, 2, 3]))
    max_allele = int(rng.choice([1, 1, 2, 3]))
    g = rng.integers(0, max_allele + 1, size=(S, N, ploidy)).astype(dtype)
    if np.dtype(dtype).kind == "i" and rng.random() < 0.6:
        g[rng.random((S, N, ploidy)) < 0.12] = -1
    if np.dtype(dtype) in (np.dtype(np.uint8), np.dtype(np.uint16), np.dtype(np.int16)) and rng.random() < 0.15 and g.size:
        g.reshape(-1)[int(rng.integers(0, g.size))] = 255   # the sparse sentinel value as an allele
    positions = np.cumsum(rng.integers(1, 30, size=S)).astype(rng.choice([np.int64, np.int32, np.uint32, np.uint64]))
    L = int(positions[-1] - positions[0] + 1) if S else 10
    names = [f"n{i}" for i in range(N)]
    sides = range(min(ploidy, 2))
    all_haps = [(s, side) for s in range(N) for side in sides]
    cut = int(rng.integers(0, len(all_haps) + 1))
    h1, h2 = all_haps[:cut], all_haps[cut:]
    base = fm.Population.from_numpy("all", g, positions, all_haps, L, sample_names=names)
    p1, p2 = base.with_haplotypes(1, h1), base.with_haplotypes(2, h2)
    variants, dense = R.convert_numeric_array(g, [int(x) for x in positions])
    o_all, o1, o2 = (R.population_context_like_lib(i, variants, hl, names, L, dense) for i, hl in enumerate((all_haps, h1, h2)))
    for a, b in ((base, o_all), (p1, o1), (p2, o2)):
        assert a.variant_count == S
        assert a.segregating_sites() == R.count_segregating_sites_for_population(b)
        assert same_float(a.nucleotide_diversity(), R.calculate_pi_for_population(b))
    got, exp = call_both(lambda: fm.hudson_fst(p1, p2), lambda: R.calculate_hudson_fst_for_pair(o1, o2))
    if exp is not None:
        for f in ("fst", "d_xy", "pi_pop1", "pi_pop2", "pi_xy_avg"):
            assert same_float(getattr(got, f), getattr(exp, f)), f
    got, exp = call_both(lambda: fm.hudson_dxy(p1, p2).d_xy, lambda: R.calculate_d_xy_hudson(o1, o2))
    assert same_float(got, exp)
    if S:
        region = (int(positions[0]), int(positions[min(S - 1, 20)]))
        got, exp = call_both(lambda: fm.hudson_fst_with_sites(p1, p2, region),
                             lambda: R.calculate_hudson_fst_for_pair_with_sites(o1, o2, R.QueryRegion(*region)))
        if exp is not None:
            assert same_float(got[0].fst, exp[0].fst) and len(got[1]) == len(exp[1])
            for gs, es in zip(got[1], exp[1]):
                assert (gs.position, gs.n1_called, gs.n2_called) == (es.position, es.n1_called, es.n2_called)
                assert same_float(gs.fst, es.fst, exact=True) and same_float(gs.d_xy, es.d_xy, exact=True)
                assert same_float(gs.pi_pop1, es.pi_pop1, exact=True) and same_float(gs.pi_pop2, es.pi_pop2, exact=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FERROMIC_FUZZ_PAIR_CASES", "80"))))
def test_two_separately_built_numpy_populations(seed):
    """hudson_* between two Population.from_numpy objects that do NOT share their arrays (same positions): the summaries
    arm reads each population's own matrix (stats.rs:1554-1623), the per-site arms read population 1's variants for both
    (3050-3055), the dense arm is not available (different matrices)."""
    import numpy as np

    rng = np.random.default_rng(7000 + seed)
    S = int(rng.choice([1, 3, 9, 40]))
    N1, N2 = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    max_allele = int(rng.choice([1, 1, 1, 2]))
    dt = [np.int8, np.uint8, np.int16][seed % 3]

    def make(n):
        g = rng.integers(0, max_allele + 1, size=(S, n, 2)).astype(dt)
        if np.dtype(dt).kind == "i":
            g[rng.random((S, n, 2)) < 0.1] = -1
        return g

    g1, g2 = make(N1), make(N2 if seed % 2 else N1)
    n2 = g2.shape[1]
    positions = np.cumsum(rng.integers(1, 30, size=S)).astype(np.int64)
    L = int(positions[-1] - positions[0] + 1)
    names1, names2 = [f"a{i}" for i in range(N1)], [f"b{i}" for i in range(n2)]
    h1 = [(s, side) for s in range(N1) for side in (0, 1)]
    h2 = [(s, side) for s in range(n2) for side in (0, 1)]
    p1 = fm.Population.from_numpy(1, g1, positions, h1, L, sample_names=names1)
    p2 = fm.Population.from_numpy(2, g2, positions, h2, L, sample_names=names2)
    v1, d1 = R.convert_numeric_array(g1, [int(x) for x in positions])
    v2, d2 = R.convert_numeric_array(g2, [int(x) for x in positions])
    o1 = R.population_context_like_lib(1, v1, h1, names1, L, d1)
    o2 = R.population_context_like_lib(2, v2, h2, names2, L, d2)
    got, exp = call_both(lambda: fm.hudson_fst(p1, p2), lambda: R.calculate_hudson_fst_for_pair(o1, o2))
    if exp is not None:
        for f in ("fst", "d_xy", "pi_pop1", "pi_pop2", "pi_xy_avg"):
            assert same_float(getattr(got, f), getattr(exp, f)), f
    got, exp = call_both(lambda: fm.hudson_dxy(p1, p2).d_xy, lambda: R.calculate_d_xy_hudson(o1, o2))
    assert same_float(got, exp)
    region = (int(positions[0]), int(positions[-1]))
    got, exp = call_both(lambda: fm.hudson_fst_with_sites(p1, p2, region),
                         lambda: R.calculate_hudson_fst_for_pair_with_sites(o1, o2, R.QueryRegion(*region)))
    if exp is not None:
        assert same_float(got[0].fst, exp[0].fst) and len(got[1]) == len(exp[1])
        for gs, es in zip(got[1], exp[1]):
            assert (gs.position, gs.n1_called, gs.n2_called) == (es.position, es.n1_called, es.n2_called)
            assert same_float(gs.fst, es.fst, exact=True) and same_float(gs.d_xy, es.d_xy, exact=True)
