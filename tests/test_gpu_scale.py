"""Parity at BASELINE.json's sizes.  C2 (1 M sites x 1 000 haplotypes) is compared in full with the
C oracle; at C4 width (5 000 haplotypes, 2 M sites = 10 GB) the checks are the size-independent
properties the domain offers — additivity over genomic slabs, symmetry under population swap, track
checksums against the regional accumulators — plus an oracle comparison on a slab in the middle of
the cohort regenerated on the host from the same counter-based stream."""

import numpy as np
import pytest

from oracle import dense as D
from oracle import ferromic_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from ferromic_amd import device

    return device


def thresholds(S, seed, sigma=0.05):
    rng = np.random.default_rng(seed)
    base = rng.beta(0.8, 0.8, size=S)
    div = rng.normal(0.0, sigma, size=S)
    thr = (np.stack([np.clip(base + div, 0.001, 0.999), np.clip(base - div, 0.001, 0.999)]) * (1 << 24)).astype(np.uint32)
    thr[0, 0], thr[1, 0] = 0, 1 << 24
    return thr


def two_pops(N):
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)
    return poc, np.stack([poc == 0, poc == 1]).astype(np.uint8)


LAYOUTS = ["packed", "bytes"]  # the bit-packed resident image (what fmh_matrix_create keeps) and the u8 rows


def settle(dm, layout):
    """A generated cohort holds u8 rows; `packed` turns it into what fmh_matrix_create would have left: planes only."""
    if layout == "packed":
        dm.pack(release_bytes=True)


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("missing", [0.0, 0.01])
def test_c2_full_against_c_oracle(dev, missing, layout):
    S, N = 1_000_000, 500
    seed = S + N
    thr = thresholds(S, seed)
    poc, masks = two_pops(N)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=missing > 0)
    mt = int(missing * (1 << 24))
    dm.generate(seed, 0, thr, poc, mt)
    settle(dm, layout)
    g = dev.Groups(dm, masks)
    got = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE)
    data, words = dm.download()
    # the host generator is the same stream
    hdata, hwords = D.generate(S, 2 * N, seed, 0, thr, poc, mt, 16)
    assert np.array_equal(data, hdata)
    if missing > 0:
        assert np.array_equal(words, hwords)
    off1 = np.nonzero(poc == 0)[0]
    off2 = np.nonzero(poc == 1)[0]
    exp = D.hudson_sweep(data, words, S, 2 * N, off1, off2, 16)
    assert np.array_equal(got.sites["alt"], exp.alt)            # bit-exact integer tracks
    assert np.array_equal(got.sites["called"], exp.called)
    for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):        # bit-exact f64 tracks
        H.assert_bits_equal(got.sites[name], getattr(exp, name), name)
    for p in range(2):
        assert got.pop[p]["segregating_sites"] == exp.pop[p]["segregating_sites"]
        assert got.pop[p]["uncallable_sites"] == exp.pop[p]["uncallable_sites"]
    for k in ("numerator_sum", "denominator_sum", "pi1_sum", "pi2_sum", "dxy_sum_all", "site_num_sum", "site_den_sum"):
        assert H.rel_close(got.totals[k], exp.totals[k]), k
    assert got.totals["dxy_uncallable_sites"] == exp.totals["dxy_uncallable_sites"]
    assert got.totals["sites_with_components"] == exp.totals["sites_with_components"]
    # summaries entry point agrees with the fused sweep (FORMULA_SUMMARY pi == dense_pi_from_counts)
    s = dev.population_summaries(dm, g, dev.FORMULA_SUMMARY, want_sites=False)
    for p in range(2):
        assert s.totals[p]["segregating_sites"] == exp.pop[p]["segregating_sites"]
        assert H.rel_close(s.totals[p]["pi_sum"], exp.pop[p]["pi_sum"])


@pytest.mark.parametrize("layout", LAYOUTS)
def test_c4_width_properties(dev, layout):
    S, N = 2_000_000, 2500
    seed = 10_002_500
    thr = thresholds(S, seed)
    poc, masks = two_pops(N)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=False)
    dm.generate(seed, 0, thr, poc, 0)
    settle(dm, layout)
    g = dev.Groups(dm, masks)
    full = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE)
    t = full.totals

    # (1) additivity over uneven genomic slabs (what region sharding relies on)
    cuts = [0, 333_337, 1_250_001, S]
    parts = [dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, a, b - a, want_sites=False) for a, b in zip(cuts, cuts[1:])]
    for k, v in t.items():
        tot = sum(p.totals[k] for p in parts)
        assert (tot == v) if isinstance(v, int) else H.rel_close(tot, v), k
    for p in range(2):
        assert sum(x.pop[p]["segregating_sites"] for x in parts) == full.pop[p]["segregating_sites"]

    # (2) symmetry: swapping the populations swaps pi and leaves Dxy / FST / components bit-identical
    sw = dev.hudson_sweep(dm, dev.Groups(dm, masks[::-1].copy()), dev.FORMULA_DENSE)
    H.assert_bits_equal(sw.sites["pi1"], full.sites["pi2"], "pi swap")
    H.assert_bits_equal(sw.sites["dxy"], full.sites["dxy"], "dxy swap")
    H.assert_bits_equal(sw.sites["fst"], full.sites["fst"], "fst swap")
    assert np.array_equal(sw.sites["alt"][0], full.sites["alt"][1])

    # (3) checksums of the tracks against the regional accumulators
    assert H.rel_close(float(np.nansum(full.sites["num"])), t["site_num_sum"])
    assert H.rel_close(float(np.nansum(full.sites["den"])), t["site_den_sum"])
    assert H.rel_close(float(np.nansum(full.sites["pi1"])), full.pop[0]["pi_sum"])
    a, n = full.sites["alt"].astype(np.int64), full.sites["called"].astype(np.int64)
    assert (n == 2500).all() and (a <= n).all()
    assert int(((a[0] > 0) & (a[0] < n[0])).sum()) == full.pop[0]["segregating_sites"]
    assert full.sites["fst"][0] == 1.0 and full.sites["dxy"][0] == 1.0  # forced informative first row
    # Hudson identities per site: num == dxy - (pi1+pi2)/2 exactly as computed, fst*den == num
    np.testing.assert_array_equal(full.sites["num"], full.sites["dxy"] - 0.5 * (full.sites["pi1"] + full.sites["pi2"]))

    # (4) a slab in the middle of the cohort against the oracle (regenerated on the host)
    b, e = 1_234_000, 1_254_000
    hdata, _ = D.generate(e - b, 2 * N, seed, b, np.ascontiguousarray(thr[:, b:e]), poc, 0, 16)
    exp = D.hudson_sweep(hdata, None, e - b, 2 * N, np.nonzero(poc == 0)[0], np.nonzero(poc == 1)[0], 16)
    assert np.array_equal(full.sites["alt"][:, b:e], exp.alt)
    for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
        H.assert_bits_equal(full.sites[name][b:e], getattr(exp, name), name)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_c3_wc_four_populations_properties(dev, layout):
    """W&C at C3 width (2 500 haplotypes, 4 populations): pair (i, j) of the 4-group sweep equals a
    2-group sweep of just those groups; regional sums equal the track sums."""
    S, N, P = 300_000, 1250, 4
    seed = 5_001_250
    base = thresholds(S, seed)
    thr = np.stack([base[p % 2] for p in range(P)])
    pop_of_sample = np.minimum(np.arange(N) * P // N, P - 1).astype(np.uint8)
    poc = np.repeat(pop_of_sample, 2)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=True)
    dm.generate(seed, 0, thr, poc, int(0.02 * (1 << 24)))
    settle(dm, layout)
    masks = np.stack([poc == p for p in range(P)]).astype(np.uint8)
    w4 = dev.wc_sweep(dm, dev.Groups(dm, masks))
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    for k, (i, j) in enumerate(pairs, start=1):
        if (i, j) not in ((0, 1), (1, 3)):
            continue
        w2 = dev.wc_sweep(dm, dev.Groups(dm, masks[[i, j]]))
        H.assert_bits_equal(w4.a[k], w2.a[0], f"pair {i}{j} a")   # overall of the 2-group sweep == that pair
        H.assert_bits_equal(w4.b[k], w2.b[1], f"pair {i}{j} b")
        assert np.array_equal(w4.state[k], w2.state[0])
    for slot in range(1 + len(pairs)):
        ok = w4.state[slot] != 3
        assert int(ok.sum()) == int(w4.informative_sites[slot])
        assert H.rel_close(float(w4.a[slot][ok].sum()), float(w4.sum_a[slot]))
        assert H.rel_close(float(w4.b[slot][ok].sum()), float(w4.sum_b[slot]))


@pytest.mark.parametrize("layout", LAYOUTS)
def test_c3_wc_sampled_sites_against_oracle(dev, layout):
    """SURVEY 8(d) parity gate for C3: a fixed subsample of sites of a C3-width cohort (2 500 haplotypes, 4 populations)
    against the oracle's literal per-site W&C (calculate_fst_wc_at_site_with_membership), a and b bit for bit; the
    regional sums against the oracle's sums over the SAME subsample read back from the GPU tracks."""
    S, N, P = 200_000, 1250, 4
    seed = 5_001_250
    base = thresholds(S, seed)
    thr = np.stack([base[p % 2] for p in range(P)])
    pop_of_sample = np.minimum(np.arange(N) * P // N, P - 1).astype(np.uint8)
    poc = np.repeat(pop_of_sample, 2)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=False)
    dm.generate(seed, 0, thr, poc, 0)
    settle(dm, layout)
    masks = np.stack([poc == p for p in range(P)]).astype(np.uint8)
    w = dev.wc_sweep(dm, dev.Groups(dm, masks))
    membership = R.SubpopulationMembership.from_map(N, {(s, side): str(pop_of_sample[s]) for s in range(N) for side in (0, 1)})
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)]
    rng = np.random.default_rng(1)
    rows = np.unique(np.concatenate([[0, 1, S - 1], rng.integers(0, S, size=150)]))
    for r in rows:
        hdata, _ = D.generate(1, 2 * N, seed, int(r), np.ascontiguousarray(thr[:, r:r + 1]), poc, 0, 1)
        g = hdata.reshape(N, 2)
        overall, _, components, _, pair_components = R.calculate_fst_wc_at_site_with_membership(R.make_variant(int(r), g.tolist()), membership)
        assert (float(w.a[0][r]), float(w.b[0][r])) == components, r
        for k, (i, j) in enumerate(pairs, start=1):
            assert (float(w.a[k][r]), float(w.b[k][r])) == pair_components[f"{i}_vs_{j}"], (r, i, j)
        assert dev.WC_STATES[w.state[0][r]] == overall.state


def test_c3_full_size(dev):
    """BASELINE config C3 at its full size - 5 M sites x 2 500 haplotypes, 4 populations, fused W&C sweep on the resident
    (bit-packed) layout - against the C oracle (oracle/dense_oracle.c fo_wc_sites_threaded, pinned bit for bit to the Python
    restatement of stats.rs:1814-2032) on the SAME cohort regenerated on the host: a, b and state of the overall slot and of
    all six pairs at EVERY site bit for bit (SURVEY 8(d) asks for 1 %), informative-site counts exact, regional sums 1e-9
    (the reference sums serially in site order, the GPU per lane / block / grid)."""
    S, N, P = 5_000_000, 1250, 4
    seed = 5_001_250
    base = thresholds(S, seed)
    thr = np.stack([base[p % 2] for p in range(P)])
    pop_of_sample = np.minimum(np.arange(N) * P // N, P - 1).astype(np.uint8)
    poc = np.repeat(pop_of_sample, 2)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=False)
    dm.generate(seed, 0, thr, poc, 0)
    settle(dm, "packed")
    masks = np.stack([poc == p for p in range(P)]).astype(np.uint8)
    w = dev.wc_sweep(dm, dev.Groups(dm, masks))
    hdata, _ = D.generate(S, 2 * N, seed, 0, thr, poc, 0, 16)
    exp = D.wc_sites(hdata, None, S, 2 * N, poc, P, 16)
    del hdata
    slots = 1 + P * (P - 1) // 2
    assert w.a.shape == (slots, S)
    assert np.array_equal(w.state, exp.state)
    assert np.array_equal(w.a.view(np.uint64), exp.a.view(np.uint64))   # bit for bit, 35 M values each
    assert np.array_equal(w.b.view(np.uint64), exp.b.view(np.uint64))
    assert (w.group_called == np.array([int((poc == p).sum()) for p in range(P)], dtype=np.uint32)[:, None]).all()
    for k in range(slots):
        assert int(w.informative_sites[k]) == int(exp.informative[k]) == S
        assert H.rel_close(float(w.sum_a[k]), float(exp.sum_a[k])), k
        assert H.rel_close(float(w.sum_b[k]), float(exp.sum_b[k])), k
    assert w.sites_attempted == S
    # the regional F_ST the API reports
    assert H.rel_close(float(w.sum_a[0] / (w.sum_a[0] + w.sum_b[0])), float(exp.sum_a[0] / (exp.sum_a[0] + exp.sum_b[0])))


def test_c5_full_size(dev, fmh_opts):
    """BASELINE config C5 at its full size - 2 M sites x 10 000 haplotypes, 2 populations, pi + Hudson - through all three counting
    routes: int8 MFMA contraction on u8 rows (the route the config names), v_dot4 on u8 rows, popcounts on the packed planes; each
    against the C oracle on the same cohort regenerated on the host: counts and all six f64 tracks bit for bit at every site,
    integer totals exact, f64 totals 1e-9; plus additivity over uneven slabs on the MFMA route."""
    S, N = 2_000_000, 5000
    seed = 2_005_000
    thr = thresholds(S, seed)
    poc, masks = two_pops(N)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=False)
    dm.generate(seed, 0, thr, poc, 0)
    g = dev.Groups(dm, masks)
    hdata, _ = D.generate(S, 2 * N, seed, 0, thr, poc, 0, 16)
    exp = D.hudson_sweep(hdata, None, S, 2 * N, np.nonzero(poc == 0)[0], np.nonzero(poc == 1)[0], 16)
    del hdata

    def check(got, what):
        assert np.array_equal(got.sites["alt"], exp.alt), what
        assert np.array_equal(got.sites["called"], exp.called), what
        for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            H.assert_bits_equal(got.sites[name], getattr(exp, name), f"{name} {what}")
        for p in range(2):
            assert got.pop[p]["segregating_sites"] == exp.pop[p]["segregating_sites"], what
            assert got.pop[p]["uncallable_sites"] == exp.pop[p]["uncallable_sites"], what
            assert H.rel_close(got.pop[p]["pi_sum"], exp.pop[p]["pi_sum"]), what
        for k in ("numerator_sum", "denominator_sum", "pi1_sum", "pi2_sum", "dxy_sum_all", "site_num_sum", "site_den_sum"):
            assert H.rel_close(got.totals[k], exp.totals[k]), (k, what)
        assert got.totals["dxy_uncallable_sites"] == exp.totals["dxy_uncallable_sites"]
        assert got.totals["sites_with_components"] == exp.totals["sites_with_components"]

    fmh_opts.setenv("FMH_LAYOUT", "bytes")
    fmh_opts.setenv("FMH_COUNTS_MFMA", "1")
    mfma = dev.hudson_sweep(dm, g, dev.FORMULA_DENSE)
    check(mfma, "int8 MFMA route")
    cuts = [0, 444_441, 1_500_007, S]
    parts = [dev.hudson_sweep(dm, g, dev.FORMULA_DENSE, a, b - a, want_sites=False) for a, b in zip(cuts, cuts[1:])]
    for k, v in mfma.totals.items():
        tot = sum(p.totals[k] for p in parts)
        assert (tot == v) if isinstance(v, int) else H.rel_close(tot, v), k
    del mfma, parts
    fmh_opts.delenv("FMH_COUNTS_MFMA")
    check(dev.hudson_sweep(dm, g, dev.FORMULA_DENSE), "dot4 route")
    fmh_opts.delenv("FMH_LAYOUT")
    dm.pack(release_bytes=True)
    check(dev.hudson_sweep(dm, g, dev.FORMULA_DENSE), "packed route")


def test_c4_full_size(dev):
    """BASELINE config C4 at its full size - 10 M sites x 5 000 haplotypes, 2 populations, the sweep bench.py times - on the resident
    (bit-packed) layout: ONE sweep over the whole cohort, then every site's counts and f64 tracks bit for bit against the C oracle,
    which regenerates the cohort on the host slab by slab (2 M sites = 10 GB at a time); integer totals exact, f64 totals 1e-9
    against the oracle's slab totals summed in slab order."""
    S, N = 10_000_000, 2500
    seed = S + N
    thr = thresholds(S, seed)
    poc, masks = two_pops(N)
    dm = dev.DeviceMatrix.alloc(S, N, 2, with_missing=False)
    dm.generate(seed, 0, thr, poc, 0)
    settle(dm, "packed")
    got = dev.hudson_sweep(dm, dev.Groups(dm, masks), dev.FORMULA_DENSE)
    off1, off2 = np.nonzero(poc == 0)[0], np.nonzero(poc == 1)[0]
    sums = {k: 0.0 for k in ("numerator_sum", "denominator_sum", "pi1_sum", "pi2_sum", "dxy_sum_all", "site_num_sum", "site_den_sum")}
    counts = {"dxy_uncallable_sites": 0, "sites_with_components": 0}
    seg, unc, pis = [0, 0], [0, 0], [0.0, 0.0]
    slab = 2_000_000
    for b in range(0, S, slab):
        e = min(S, b + slab)
        hdata, _ = D.generate(e - b, 2 * N, seed, b, np.ascontiguousarray(thr[:, b:e]), poc, 0, 16)
        exp = D.hudson_sweep(hdata, None, e - b, 2 * N, off1, off2, 16)
        del hdata
        assert np.array_equal(got.sites["alt"][:, b:e], exp.alt), b
        assert np.array_equal(got.sites["called"][:, b:e], exp.called), b
        for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            H.assert_bits_equal(got.sites[name][b:e], getattr(exp, name), f"{name} slab {b}")
        for k in sums:
            sums[k] += exp.totals[k]
        for k in counts:
            counts[k] += exp.totals[k]
        for p in range(2):
            seg[p] += exp.pop[p]["segregating_sites"]
            unc[p] += exp.pop[p]["uncallable_sites"]
            pis[p] += exp.pop[p]["pi_sum"]
    for k, v in sums.items():
        assert H.rel_close(got.totals[k], v), k
    for k, v in counts.items():
        assert got.totals[k] == v, k
    for p in range(2):
        assert got.pop[p]["segregating_sites"] == seg[p] and got.pop[p]["uncallable_sites"] == unc[p]
        assert H.rel_close(got.pop[p]["pi_sum"], pis[p])
    # the fused region sweep (summaries + both groups' per-site diversity + the Hudson pair, ONE read) over the same 10 M sites: its Hudson
    # tracks and counts must be the bits just checked against the oracle, its diversity tracks the bits of fmh_diversity_sites per group
    import ctypes as C

    from ferromic_amd import _abi

    lib = _abi.load()
    g2 = dev.Groups(dm, masks)
    bufs = {k: dev.DeviceBuffer(dm.device, 8 * S) for k in ("fst", "dxy", "pi1", "pi2", "num", "den")}
    bufs.update({k: dev.DeviceBuffer(dm.device, 8 * S) for k in ("alt", "called")})
    bufs.update({k: dev.DeviceBuffer(dm.device, 16 * S) for k in ("site_pi", "site_theta")})
    sites = _abi.HudsonSites(*(bufs[k].ptr for k in ("fst", "dxy", "pi1", "pi2", "num", "den", "alt", "called")))
    div = _abi.PairDiversitySites(bufs["site_pi"].ptr, bufs["site_theta"].ptr)
    tot = _abi.HudsonTotals()
    _abi.check(lib.fmh_pair_region_sweep(dm._h, g2._h, 0, S, dev.FORMULA_DENSE, dev.FORMULA_DENSE, C.byref(div), C.byref(sites), C.byref(tot), None))
    for name in ("fst", "dxy", "pi1", "pi2", "num", "den"):
        H.assert_bits_equal(bufs[name].to_numpy(np.float64, S), got.sites[name], f"fused {name}")
    assert np.array_equal(bufs["alt"].to_numpy(np.uint32, 2 * S).reshape(2, S), got.sites["alt"])
    assert tot.sites_with_components == got.totals["sites_with_components"] and tot.pop[0].segregating_sites == seg[0] and tot.pop[1].segregating_sites == seg[1]
    assert H.rel_close(tot.numerator_sum, got.totals["numerator_sum"]) and H.rel_close(tot.pop[1].pi_sum, pis[1])
    fused_pi = bufs["site_pi"].to_numpy(np.float64, 2 * S).reshape(2, S)
    fused_theta = bufs["site_theta"].to_numpy(np.float64, 2 * S).reshape(2, S)
    for p in range(2):
        dv = dev.diversity_sites(dm, dev.Groups(dm, masks[p:p + 1]))
        H.assert_bits_equal(fused_pi[p], dv.pi, f"fused site pi group {p}")
        H.assert_bits_equal(fused_theta[p], dv.theta, f"fused site theta group {p}")
