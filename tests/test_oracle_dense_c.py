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
