"""Pins the C half of the oracle (oracle/dense_oracle.c) to the Python restatement, which is itself
pinned by the reference's own KATs (tests/test_oracle_golden.py)."""

import numpy as np
import pytest

from oracle import dense as D
from oracle import ferromic_ref as R
from tests import helpers as H


@pytest.mark.parametrize("sites,samples,p_missing,threads", [(400, 31, 0.0, 1), (400, 31, 0.0, 3), (333, 26, 0.12, 4)])
def test_c_sweep_matches_python_oracle(sites, samples, p_missing, threads):
    rng = np.random.default_rng(sites + samples)
    m = H.random_dense_matrix(rng, sites, samples, 2, 1, p_missing)
    h1 = H.haps_for_samples(range(0, samples // 2))
    h2 = H.haps_for_samples(range(samples // 2, samples - 1))
    off1 = R.dense_membership_offsets(m, h1)
    off2 = R.dense_membership_offsets(m, h2)
    out = D.hudson_sweep(np.frombuffer(m.data, dtype=np.uint8), H.missing_words_np(m), sites, m.stride, off1, off2, threads)
    s1 = R.build_dense_population_summary(m, h1)
    s2 = R.build_dense_population_summary(m, h2)
    assert np.array_equal(out.alt[0], np.array(s1.alt_counts, dtype=np.uint32))
    assert np.array_equal(out.called[1], np.array(s2.called_counts, dtype=np.uint32))
    assert out.pop[0]["segregating_sites"] == s1.segregating_sites
    assert out.pop[1]["segregating_sites"] == s2.segregating_sites
    assert H.rel_close(out.pop[0]["pi_sum"], s1.pi_sum, 1e-12)
    t = R.aggregate_hudson_components_from_summaries(s1, s2)
    for k in ("numerator_sum", "denominator_sum", "pi1_sum", "pi2_sum", "dxy_sum_all"):
        assert H.rel_close(out.totals[k], getattr(t, k), 1e-12), k
    assert out.totals["dxy_uncallable_sites"] == t.dxy_uncallable_sites
    exp = R.dense_hudson_sites(m, [R.Variant(i, None) for i in range(sites)], off1, off2)
    H.assert_bits_equal(out.fst, [H.opt(x.fst) for x in exp], "fst")
    H.assert_bits_equal(out.dxy, [H.opt(x.d_xy) for x in exp], "dxy")
    H.assert_bits_equal(out.pi1, [H.opt(x.pi_pop1) for x in exp], "pi1")
    H.assert_bits_equal(out.pi2, [H.opt(x.pi_pop2) for x in exp], "pi2")
    H.assert_bits_equal(out.num, [H.opt(x.num_component) for x in exp], "num")
    H.assert_bits_equal(out.den, [H.opt(x.den_component) for x in exp], "den")
    ns, ds = R.hudson_component_sums(exp)
    assert H.rel_close(out.totals["site_num_sum"], ns, 1e-12) and H.rel_close(out.totals["site_den_sum"], ds, 1e-12)


def test_generator_is_deterministic_and_thread_independent():
    S, H_ = 300, 50
    rng = np.random.default_rng(3)
    thr = (rng.random((2, S)) * (1 << 24)).astype(np.uint32)
    poc = (np.arange(H_) >= H_ // 2).astype(np.uint8)
    a, wa = D.generate(S, H_, 99, 1000, thr, poc, int(0.1 * (1 << 24)), 1)
    b, wb = D.generate(S, H_, 99, 1000, thr, poc, int(0.1 * (1 << 24)), 5)
    assert np.array_equal(a, b) and np.array_equal(wa, wb)
    # slab property used for region sharding: rows [100,200) of the cohort == a slab generated alone
    c, wc = D.generate(100, H_, 99, 1100, thr[:, 100:200].copy(), poc, 0, 2)
    c2, _ = D.generate(S, H_, 99, 1000, thr, poc, 0, 2)
    assert np.array_equal(c, c2.reshape(S, H_)[100:200].reshape(-1))
    frac = a.reshape(S, H_)[:, : H_ // 2].mean(axis=1)
    assert abs(np.corrcoef(frac, thr[0] / float(1 << 24))[0, 1]) > 0.8


@pytest.mark.parametrize("sites,samples,G,max_allele,p_missing,ungrouped,threads", [
    (160, 24, 2, 1, 0.0, 0, 1), (140, 31, 4, 1, 0.08, 3, 3), (120, 26, 3, 3, 0.15, 5, 2), (90, 18, 5, 2, 0.5, 2, 4), (40, 9, 2, 1, 0.97, 0, 1)])
def test_c_wc_matches_python_oracle(sites, samples, G, max_allele, p_missing, ungrouped, threads):
    """fo_wc_sites_threaded against calculate_fst_wc_at_site_with_membership / calculate_overall_fst_wc, bit for bit:
    per-site a, b and state of the overall slot and of every pair, and the regional sums (serial, site order)."""
    rng = np.random.default_rng(1000 * sites + samples + G)
    m = H.random_dense_matrix(rng, sites, samples, 2, max_allele, p_missing)
    data = np.frombuffer(m.data, dtype=np.uint8).reshape(sites, 2 * samples)
    words = H.missing_words_np(m)
    miss = np.zeros((sites, 2 * samples), dtype=bool) if words is None else \
        np.unpackbits(words.view(np.uint8), bitorder="little")[: sites * 2 * samples].reshape(sites, 2 * samples).astype(bool)
    # make the dense mask expressible in the sparse model the Python restatement takes: a genotype whose FIRST allele is missing
    # is None as a whole (process.rs:479-496), one whose second allele is missing is a haploid call
    miss[:, 1::2] |= miss[:, 0::2]
    if words is not None:
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        words = np.frombuffer(np.concatenate([bits, np.zeros((-len(bits)) % 8, np.uint8)]).tobytes(), dtype="<u8").copy()
    group_of_sample = rng.integers(0, G, size=samples)
    group_of_sample[:G] = np.arange(G)            # every group has a member
    group_map = {(s, side): str(int(group_of_sample[s])) for s in range(samples - ungrouped) for side in (0, 1)}
    membership = R.SubpopulationMembership.from_map(samples, group_map)
    goc = np.full(2 * samples, 0xFF, dtype=np.uint8)
    for (s, side), label in group_map.items():
        goc[2 * s + side] = membership.labels.index(label)
    n_groups = membership.group_count()
    out = D.wc_sites(data.reshape(-1), words, sites, 2 * samples, goc, n_groups, threads)
    pairs = [(i, j) for i in range(n_groups) for j in range(i + 1, n_groups)]
    states = {"calculable": 0, "components_yield_indeterminate_ratio": 1, "no_inter_population_variance": 2, "insufficient_data_for_estimation": 3}
    site_records = []
    for s in range(sites):
        genos = []
        for smp in range(samples):
            g = [int(data[s, 2 * smp + k]) for k in (0, 1)]
            genos.append(None if miss[s, 2 * smp] else (g[:1] if miss[s, 2 * smp + 1] else g))
        overall, pw, comps, sizes, pcomps = R.calculate_fst_wc_at_site_with_membership(R.make_variant(s, genos), membership)
        site_records.append(R.SiteFstWc(s + 1, overall, pw, comps, sizes, pcomps))
        assert (float(out.a[0][s]), float(out.b[0][s])) == comps, s
        assert int(out.state[0][s]) == states[overall.state], s
        for k, (i, j) in enumerate(pairs, start=1):
            key = f"{membership.labels[i]}_vs_{membership.labels[j]}"
            if overall.state == "insufficient_data_for_estimation":
                assert int(out.state[k][s]) == 3
                continue
            assert (float(out.a[k][s]), float(out.b[k][s])) == pcomps[key], (s, key)
            assert int(out.state[k][s]) == states[pw[key].state], (s, key)
    overall, pw, agg = R.calculate_overall_fst_wc(site_records)
    if overall.state != "insufficient_data_for_estimation":
        assert (float(out.sum_a[0]), float(out.sum_b[0])) == (overall.sum_a, overall.sum_b) and int(out.informative[0]) == overall.sites
    for k, (i, j) in enumerate(pairs, start=1):
        key = f"{membership.labels[i]}_vs_{membership.labels[j]}"
        if key in agg and pw[key].state != "insufficient_data_for_estimation":
            assert (float(out.sum_a[k]), float(out.sum_b[k])) == agg[key], key
            assert int(out.informative[k]) == pw[key].sites
