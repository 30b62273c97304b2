"""An INDEPENDENT pin for the Weir & Cockerham part of the oracle.

The reference holds no W&C test of any kind (no call of calculate_fst_wc*, no FstEstimate vector under src/tests, src/pytests or
src/pybenches), so nothing can be transcribed; oracle/ferromic_ref.py's W&C functions are a literal restatement of stats.rs:1781-2374
and until now were pinned only by vectors derived from that same source.  This file derives the estimator from the paper instead -
B. S. Weir & C. C. Cockerham, "Estimating F-statistics for the analysis of population structure", Evolution 38 (1984), eqs. 2-4 and 10 -
for groups of HAPLOID gene copies (what ferromic's haplotype groups are: the observed heterozygote frequency h-bar of eq. 3 is 0, and
so is the component c = h-bar / 2), in exact rational arithmetic, never looking at stats.rs:

    n-bar = sum n_i / r                  n_c = (r n-bar - sum n_i^2 / (r n-bar)) / (r - 1)            p-bar = sum n_i p_i / (r n-bar)
    s^2   = sum n_i (p_i - p-bar)^2 / ((r - 1) n-bar)
    a     = (n-bar / n_c) [ s^2 - (p-bar (1 - p-bar) - (r - 1) s^2 / r) / (n-bar - 1) ]
    b     = (n-bar / (n-bar - 1)) [ p-bar (1 - p-bar) - (r - 1) s^2 / r ]
    theta-hat (eq. 10, several alleles / loci) = sum a / sum (a + b)

and checks, on random cohorts of haplotype groups with missing calls, unequal group sizes, ungrouped samples and multi-allelic sites:
(1) the oracle's per-site (a, b) equals the SUM over the alleles present of the textbook components (at a biallelic site: exactly twice
the single-allele textbook values - the reference sums over both alleles, the quirk SURVEY.md a12 records - so theta-hat is unchanged);
(2) the same for every pair of groups; (3) the regional theta-hat = sum a / (sum a + sum b) to 1e-12 relative.  This is the strongest pin
available: it fixes the estimator's algebra independently of the reference's source, while bit-level details (operation order) stay
pinned by the literal restatement."""

import random
from fractions import Fraction

import pytest

from oracle import ferromic_ref as R


def textbook_components(ns, ps):
    """W&C (1984) eqs. 2-4 with h-bar = 0 for r groups of n_i haploid gene copies with allele frequencies p_i; exact rationals."""
    r = len(ns)
    nbar = Fraction(sum(ns), r)
    nc = (r * nbar - Fraction(sum(n * n for n in ns)) / (r * nbar)) / (r - 1)
    pbar = sum(n * p for n, p in zip(ns, ps)) / (r * nbar)
    s2 = sum(n * (p - pbar) ** 2 for n, p in zip(ns, ps)) / ((r - 1) * nbar)
    inner = pbar * (1 - pbar) - Fraction(r - 1, r) * s2
    a = (nbar / nc) * (s2 - inner / (nbar - 1))
    b = (nbar / (nbar - 1)) * inner
    return a, b


def random_cohort(rng, n_samples, n_groups, n_sites, max_allele, p_missing, p_ungrouped):
    names = [f"S{i:03d}" for i in range(n_samples)]
    group_of = {}  # (sample, side) -> label; haplotype groups: the two sides of a sample may sit in different groups
    for s in range(n_samples):
        for side in (0, 1):
            if rng.random() >= p_ungrouped:
                group_of[(s, side)] = f"g{rng.randrange(n_groups)}"
    variants = []
    for site in range(n_sites):
        f = [rng.random() for _ in range(n_groups)]
        gts = []
        for s in range(n_samples):
            if rng.random() < p_missing:
                gts.append(None)
                continue
            g = []
            for side in (0, 1):
                lab = group_of.get((s, side))
                base = f[int(lab[1:])] if lab else 0.5
                allele = 1 if rng.random() < base else 0
                if allele and max_allele > 1 and rng.random() < 0.3:
                    allele = rng.randint(2, max_allele)
                g.append(allele)
            gts.append(g)
        variants.append(R.make_variant(10 * site + 3, gts))
    return names, group_of, variants


def counts_by_group(variant, group_of, labels):
    """called gene copies and per-allele counts of every group at one site, straight from the genotypes"""
    n = {lab: 0 for lab in labels}
    c = {lab: {} for lab in labels}
    for s, g in enumerate(variant.genotypes):
        if g is None:
            continue
        for side, allele in enumerate(g):
            lab = group_of.get((s, side))
            if lab is None:
                continue
            n[lab] += 1
            c[lab][allele] = c[lab].get(allele, 0) + 1
    return n, c


def alleles_present(variant):
    out = set()
    for g in variant.genotypes:
        if g is not None:
            out.update(g)
    return sorted(out)


@pytest.mark.parametrize("seed,n_groups,max_allele,p_missing", [(1, 2, 1, 0.0), (2, 3, 1, 0.05), (3, 4, 1, 0.1), (4, 4, 3, 0.05), (5, 5, 2, 0.0), (6, 2, 1, 0.3)])
def test_oracle_wc_equals_the_1984_estimator(seed, n_groups, max_allele, p_missing):
    rng = random.Random(seed)
    names, group_of, variants = random_cohort(rng, n_samples=31, n_groups=n_groups, n_sites=60, max_allele=max_allele, p_missing=p_missing, p_ungrouped=0.1)
    membership = R.SubpopulationMembership.from_map(len(names), group_of)
    labels = membership.labels
    sum_a = sum_b = Fraction(0)
    pair_sums = {}
    sites = []
    checked_sites = checked_pairs = biallelic_doubles = 0
    for v in variants:
        overall, pw, comps, sizes, pw_comps = R.calculate_fst_wc_at_site_with_membership(v, membership)
        sites.append(R.SiteFstWc(v.position + 1, overall, pw, comps, sizes, pw_comps))
        n, c = counts_by_group(v, group_of, labels)
        present = alleles_present(v)
        live = [lab for lab in labels if n[lab] > 0]
        if len(live) < 2 or (sum(n[lab] for lab in live) == len(live)):
            continue  # fewer than two groups with data, or n-bar = 1: the estimator is undefined and the reference reports (0, 0) / insufficient
        # (1) overall: the sum over the alleles present of the textbook components
        ta = tb = Fraction(0)
        per_allele = []
        for allele in present:
            a, b = textbook_components([n[lab] for lab in live], [Fraction(c[lab].get(allele, 0), n[lab]) for lab in live])
            per_allele.append((a, b))
            ta += a
            tb += b
        assert comps[0] == pytest.approx(float(ta), rel=1e-12, abs=1e-14), (v.position, comps, float(ta))
        assert comps[1] == pytest.approx(float(tb), rel=1e-12, abs=1e-14)
        if len(present) == 2:  # the two alleles of a biallelic site carry the SAME components: the site holds twice the textbook value
            assert per_allele[0] == per_allele[1]
            assert comps[0] == pytest.approx(2 * float(per_allele[0][0]), rel=1e-12, abs=1e-14)
            biallelic_doubles += 1
        if overall.state != "insufficient_data_for_estimation":
            sum_a += ta
            sum_b += tb
        checked_sites += 1
        # (2) every pair of groups that both have data: the r = 2 estimator on the pair alone
        for i, j, key in membership.pair_keys:
            li, lj = labels[i], labels[j]
            if n[li] == 0 or n[lj] == 0 or n[li] + n[lj] == 2:
                continue
            pa = pb = Fraction(0)
            for allele in present:
                a, b = textbook_components([n[li], n[lj]], [Fraction(c[li].get(allele, 0), n[li]), Fraction(c[lj].get(allele, 0), n[lj])])
                pa += a
                pb += b
            got = pw_comps[key]
            assert got[0] == pytest.approx(float(pa), rel=1e-12, abs=1e-14), (v.position, key)
            assert got[1] == pytest.approx(float(pb), rel=1e-12, abs=1e-14)
            if pw[key].state != "insufficient_data_for_estimation":
                s = pair_sums.setdefault(key, [Fraction(0), Fraction(0)])
                s[0] += pa
                s[1] += pb
            checked_pairs += 1
    assert checked_sites >= 40 and checked_pairs >= 40 and (max_allele > 1 or biallelic_doubles >= 20)
    # (3) the regional estimate: theta-hat = sum a / (sum a + sum b) over the sites (eq. 10)
    overall, pairwise, agg = R.calculate_overall_fst_wc(sites)
    if sum_a + sum_b != 0:
        theta = float(sum_a / (sum_a + sum_b))
        assert overall.state == "calculable" and overall.value == pytest.approx(theta, rel=1e-12, abs=1e-14)
    for key, (pa, pb) in pair_sums.items():
        assert agg[key][0] == pytest.approx(float(pa), rel=1e-11, abs=1e-13) and agg[key][1] == pytest.approx(float(pb), rel=1e-11, abs=1e-13)
        if pa + pb != 0 and pairwise[key].state == "calculable":
            assert pairwise[key].value == pytest.approx(float(pa / (pa + pb)), rel=1e-11, abs=1e-13)


def test_textbook_estimator_sanity():
    """The restated equations behave as the paper says: identical frequencies in every group give theta-hat <= 0 (no structure, small
    negative by the sampling correction), fixed differences give theta-hat = 1."""
    a, b = textbook_components([10, 10], [Fraction(1, 2), Fraction(1, 2)])
    assert a < 0 and a + b > 0 and a / (a + b) < 0
    a, b = textbook_components([12, 8, 20], [Fraction(1), Fraction(0), Fraction(1)])
    assert b == 0 and a > 0  # theta-hat = a / (a + b) = 1
    a2, b2 = textbook_components([7, 9], [Fraction(3, 7), Fraction(2, 9)])
    a2r, b2r = textbook_components([7, 9], [Fraction(4, 7), Fraction(7, 9)])  # the other allele: p -> 1 - p
    assert (a2, b2) == (a2r, b2r)
