"""The LDS-staged flat-tile route of short packed rows (sweep_flat_kernels.hpp: one row per lane, scalar masks, LDS-DMA tiles) against the
four-lane route it replaces and against the oracle.  Per-site tracks and integer totals must be the SAME BITS on both routes (the same counts
go through the same epilogue code); f64 regional sums come from another grid (two waves per workgroup), so they are held to 1e-12 relative."""

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


def upload(dev, m):
    return dev.DeviceMatrix.from_host(np.frombuffer(m.data, dtype=np.uint8), H.missing_words_np(m), m.variant_count, m.sample_count, m.ploidy, m.max_allele)


def close_totals(a, b, what):
    assert a.keys() == b.keys(), what
    for k in a:
        if isinstance(a[k], float):
            assert math.isclose(a[k], b[k], rel_tol=1e-12, abs_tol=1e-15) or (math.isnan(a[k]) and math.isnan(b[k])), (what, k, a[k], b[k])
        else:
            assert a[k] == b[k], (what, k, a[k], b[k])


def run_all(dev, dm, g1, g2, g4, S):
    lo, cnt = (3, S - 5) if S > 8 else (0, S)
    return dict(
        hud=dev.hudson_sweep(dm, g2, dev.FORMULA_DENSE),
        hud_range=dev.hudson_sweep(dm, g2, dev.FORMULA_SPARSE, lo, cnt),
        div=dev.diversity_sites(dm, g1),
        sum2=dev.population_summaries(dm, g2, dev.FORMULA_SUMMARY),
        sum4=dev.population_summaries(dm, g4, dev.FORMULA_SUMMARY),
        sum1=dev.population_summaries(dm, g1, dev.FORMULA_DENSE),
        wc4=dev.wc_sweep(dm, g4),
        wc2=dev.wc_sweep(dm, g2, lo, cnt),
    )


def same(got, base, what):
    for i in ("hud", "hud_range"):
        for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
            H.assert_bits_equal(got[i].sites[k], base[i].sites[k], f"{i} {k} {what}")
        assert np.array_equal(got[i].sites["alt"], base[i].sites["alt"]) and np.array_equal(got[i].sites["called"], base[i].sites["called"]), what
        close_totals(got[i].totals, base[i].totals, f"{i} totals {what}")
        for p in (0, 1):
            close_totals(got[i].pop[p], base[i].pop[p], f"{i} pop {p} {what}")
    H.assert_bits_equal(got["div"].pi, base["div"].pi, "site pi " + what)
    H.assert_bits_equal(got["div"].theta, base["div"].theta, "site theta " + what)
    assert np.array_equal(got["div"].called, base["div"].called) and np.array_equal(got["div"].distinct, base["div"].distinct), what
    close_totals(got["div"].totals, base["div"].totals, "div totals " + what)
    for i in ("sum1", "sum2", "sum4"):
        assert np.array_equal(got[i].alt, base[i].alt) and np.array_equal(got[i].called, base[i].called), (i, what)
        for a, b in zip(got[i].totals, base[i].totals):
            close_totals(a, b, f"{i} totals {what}")
    for i in ("wc4", "wc2"):
        assert np.array_equal(got[i].a.view(np.uint64), base[i].a.view(np.uint64)), (i, what)
        assert np.array_equal(got[i].b.view(np.uint64), base[i].b.view(np.uint64)), (i, what)
        assert np.array_equal(got[i].state, base[i].state) and np.array_equal(got[i].group_called, base[i].group_called), (i, what)
        assert np.array_equal(got[i].informative_sites, base[i].informative_sites) and got[i].sites_attempted == base[i].sites_attempted, (i, what)
        assert np.allclose(got[i].sum_a, base[i].sum_a, rtol=1e-12, atol=1e-15) and np.allclose(got[i].sum_b, base[i].sum_b, rtol=1e-12, atol=1e-15), (i, what)


# (sites, samples): columns = 2 x samples, vectors per row = ceil(columns / 128): every swizzle class (odd, 2 mod 4, 4 mod 8, 8 mod 16, 16, 32),
# ragged widths, one-row and one-tile matrices, partial last tiles, a wave with dozens of tiles
CASES = [
    (1, 3), (63, 64), (64, 65), (65, 100), (200, 190), (333, 192), (129, 250), (257, 321), (130, 384), (70, 449), (1000, 500), (90, 575),
    (150, 640), (77, 700), (300, 768), (65, 1000), (4097, 1250), (100, 1500), (66, 1985), (2000, 2048),
]


@pytest.mark.parametrize("S,N", CASES)
def test_flat_tile_route_is_the_same_bits(dev, fmh_opts, S, N):
    rng = np.random.default_rng(7000 + S + N)
    m = H.random_dense_matrix(rng, S, N, 2, 1, 0.0)
    dm = upload(dev, m)
    cut = max(1, N // 3)
    lists = [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, max(cut + 1, N - 2)))]
    quarters = [H.haps_for_samples(range(i, N, 4)) for i in range(4)] if N >= 4 else [H.haps_for_samples(range(0, 1)), H.haps_for_samples(range(1, 2)), H.haps_for_samples(range(2, 3)), H.haps_for_samples(range(0, 2))]
    g2, g1, g4 = (dev.Groups.from_haplotype_lists(dm, x) for x in (lists, lists[:1], quarters))
    fmh_opts.setenv("FMH_FLAT", "0")
    base = run_all(dev, dm, g1, g2, g4, S)
    # the reference's counts, from the oracle
    exp = R.build_dense_population_summary(m, lists[0])
    assert np.array_equal(base["hud"].sites["alt"][0], np.array(exp.alt_counts, dtype=np.uint32))
    fmh_opts.setenv("FMH_FLAT", "1")
    # FMH_FLAT_SLOTS: 0 = the register-staged variant (with its deferral depths), 1 / 2 = the LDS-DMA variants
    for slots, defer in (("0", "1"), ("0", "3"), ("0", "8"), ("1", None), ("2", None)):
        fmh_opts.setenv("FMH_FLAT_SLOTS", slots)
        if defer is not None:
            fmh_opts.setenv("FMH_FLAT_DEFER", defer)
        for blocks in (None, "1", "3"):
            if blocks is None:
                fmh_opts.delenv("FMH_GRID_BLOCKS", raising=False)
            else:
                if S < 300:
                    continue
                fmh_opts.setenv("FMH_GRID_BLOCKS", blocks)
            got = run_all(dev, dm, g1, g2, g4, S)
            same(got, base, f"{S}x{N} slots {slots} defer {defer} blocks {blocks}")
        fmh_opts.delenv("FMH_GRID_BLOCKS", raising=False)
        fmh_opts.delenv("FMH_FLAT_DEFER", raising=False)


def test_flat_tile_route_fused_region_sweep(dev, fmh_opts):
    """fmh_pair_region_sweep (summaries + both groups' diversity + Hudson from one read of the matrix) on the flat route: every track the
    same bits as on the four-lane route, integer totals equal, f64 totals to 1e-12."""
    import ctypes as C

    from ferromic_amd import _abi

    lib = _abi.load()
    rng = np.random.default_rng(99)
    for (S, N) in ((500, 500), (131, 1250), (64, 60), (1000, 2048)):
        m = H.random_dense_matrix(rng, S, N, 2, 1, 0.0)
        dm = upload(dev, m)
        cut = N // 2
        g2 = dev.Groups.from_haplotype_lists(dm, [H.haps_for_samples(range(0, cut)), H.haps_for_samples(range(cut, N - 1))])
        r0, rows = 5, S - 9
        for summary_formula, hudson_formula in ((dev.FORMULA_DENSE, dev.FORMULA_SPARSE), (dev.FORMULA_DENSE, -1)):
            out = {}
            for flat in ("0", "1"):
                fmh_opts.setenv("FMH_FLAT", flat)
                bufs = {k: dev.DeviceBuffer(dm.device, 8 * 2 * rows) for k in ("pi", "theta")}
                for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                    bufs[k] = dev.DeviceBuffer(dm.device, 8 * rows)
                for k in ("alt", "called"):
                    bufs[k] = dev.DeviceBuffer(dm.device, 4 * 2 * rows)
                div = _abi.PairDiversitySites(bufs["pi"].ptr, bufs["theta"].ptr)
                sites = _abi.HudsonSites(*(bufs[k].ptr for k in ("fst", "dxy", "pi1", "pi2", "num", "den", "alt", "called")))
                tot = _abi.HudsonTotals()
                _abi.check(lib.fmh_pair_region_sweep(dm._h, g2._h, r0, rows, summary_formula, hudson_formula, C.byref(div), C.byref(sites), C.byref(tot), None))
                arrays = {k: bufs[k].to_numpy(np.float64, (2 if k in ("pi", "theta") else 1) * rows) for k in ("pi", "theta", "fst", "dxy", "pi1", "pi2", "num", "den")}
                arrays.update({k: bufs[k].to_numpy(np.uint32, 2 * rows) for k in ("alt", "called")})
                out[flat] = (arrays, dev.hudson_totals_dict(tot), [dev._pop_totals(tot.pop[p]) for p in (0, 1)])
            what = f"{S}x{N} formulas {summary_formula}/{hudson_formula}"
            for k, a in out["0"][0].items():
                if hudson_formula < 0 and k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
                    continue
                if a.dtype == np.float64:
                    H.assert_bits_equal(out["1"][0][k], a, f"fused {k} {what}")
                else:
                    assert np.array_equal(out["1"][0][k], a), (k, what)
            close_totals(out["1"][1], out["0"][1], "fused totals " + what)
            for p in (0, 1):
                close_totals(out["1"][2][p], out["0"][2][p], f"fused pop {p} {what}")
