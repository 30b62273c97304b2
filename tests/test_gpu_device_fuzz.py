"""Counting fuzz at the C-ABI: random matrix geometries around every boundary the kernels have (16-byte vectors,
4-vector batches, 16-lane rows, 64-site tiles, 4-wave workgroups), random overlapping memberships of 1..8 groups,
missing masks and allele ranges 1..9; the integer outputs of the sweeps (alt, called, segregating sites, uncallable
sites, W&C group sizes and informative-site counts) must equal a direct numpy count."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("FERROMIC_FUZZ_DEVICE_CASES", "60"))
WIDTHS = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 511, 513, 1023, 1025, 2500]
ROWS = [1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 255, 256, 257, 300, 1000]


@pytest.fixture(scope="module")
def dev():
    from ferromic_amd import device

    return device


# seeds beyond the default range that once failed: one-group general sweeps with missing calls on rows of >= 32 vectors
# lost the last v_dot4 of a row to a dot4 -> DPP hazard (tools/scan_dot4_hazard.py, DESIGN.md "Hazards")
REGRESSION_SEEDS = [185, 325, 369, 389, 419, 466, 759, 788]


@pytest.mark.parametrize("seed", list(range(CASES)) + [s for s in REGRESSION_SEEDS if s >= CASES])
def test_random_geometry_counts(dev, fmh_opts, seed):
    rng = np.random.default_rng(9000 + seed)
    if seed % 2 and seed not in REGRESSION_SEEDS:  # the tables of rows with upper-plane bits / uncalled columns at any matrix size (default: from 4 096 rows)
        fmh_opts.setenv("FMH_ROW_HI", "2")
    N = int(rng.choice(WIDTHS))
    S = int(rng.choice(ROWS))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    H = N * ploidy
    max_allele = int(rng.choice([1, 1, 2, 3, 4, 7, 9]))
    p_missing = float(rng.choice([0.0, 0.0, 0.03, 0.3]))
    data = rng.integers(0, max_allele + 1, size=(S, H), dtype=np.uint8)
    data[rng.random((S, H)) < 0.5] = 0
    miss = rng.random((S, H)) < p_missing if p_missing > 0 else np.zeros((S, H), dtype=bool)
    if seed % 4 == 1 and seed not in REGRESSION_SEEDS:  # a cohort that is mostly biallelic and mostly complete: alleles above 1 and missing calls in a tenth of the rows
        plain = rng.random(S) >= 0.1
        data[plain] = np.minimum(data[plain], 1)
        miss[plain] = False
    data[0, 0] = max_allele
    miss[0, 0] = False
    data[miss] = 0
    words = None
    if p_missing > 0:
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        pad = (-len(bits)) % 8
        words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8").copy()
    dm = dev.DeviceMatrix.from_host(data.reshape(-1), words, S, N, ploidy, max_allele)
    G = int(rng.integers(1, 9))
    masks = (rng.random((G, H)) < rng.choice([0.1, 0.5, 0.9])).astype(np.uint8)   # overlapping, possibly empty groups
    masks[0, 0] = 1
    called_ref = np.stack([((masks[g][None, :] == 1) & ~miss).sum(axis=1) for g in range(G)]).astype(np.uint32)
    counts = np.stack([np.stack([(((data == a) & ~miss) * masks[g][None, :]).sum(axis=1) for g in range(G)]) for a in range(max_allele + 1)])
    distinct = (counts > 0).sum(axis=0)                       # [G][S]
    r0, r1 = (0, S) if seed % 3 else (int(rng.integers(0, S)), int(rng.integers(0, S)) + 1)
    r0, r1 = min(r0, r1 - 1) if r1 > 0 else 0, max(r1, r0 + 1)
    r1 = min(r1, S)
    res = dev.population_summaries(dm, dev.Groups(dm, masks), dev.FORMULA_DENSE, r0, r1 - r0)
    assert np.array_equal(res.called, called_ref[:, r0:r1])
    if max_allele >= 1:
        assert np.array_equal(res.alt, counts[1][:, r0:r1].astype(np.uint32))
    for g in range(G):
        assert res.totals[g]["segregating_sites"] == int((distinct[g, r0:r1] >= 2).sum())
        assert res.totals[g]["uncallable_sites"] == int((called_ref[g, r0:r1] < 2).sum())
        assert res.totals[g]["haplotype_capacity"] == int(masks[g].sum())
    if G >= 2:
        w = dev.wc_sweep(dm, dev.Groups(dm, masks), r0, r1 - r0)
        assert np.array_equal(w.group_called, called_ref[:, r0:r1])
        any_allele = (~miss[r0:r1]).any(axis=1)
        assert np.array_equal(w.state[0] != 3, any_allele)
        assert int(w.informative_sites[0]) == int(any_allele.sum())
        k = 1
        for i in range(G):
            for j in range(i + 1, G):
                both = any_allele & (called_ref[i, r0:r1] > 0) & (called_ref[j, r0:r1] > 0)
                assert np.array_equal(w.state[k] != 3, both), (i, j)
                assert int(w.informative_sites[k]) == int(both.sum())
                k += 1
        wm = dev.wc_sweep_many(dm, masks, r0, r1 - r0)   # counts path: same per-site bits as the fused kernel
        assert np.array_equal(wm.a, w.a) and np.array_equal(wm.b, w.b) and np.array_equal(wm.state, w.state)
        wt = dev.wc_sweep_many(dm, masks, r0, r1 - r0, sites=False)   # no track asked for: regional sums straight from the count tables
        assert np.array_equal(wt.informative_sites, wm.informative_sites)
        assert np.allclose(wt.sum_a, wm.sum_a, rtol=1e-11, atol=1e-11) and np.allclose(wt.sum_b, wm.sum_b, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("seed", range(CASES))
def test_random_geometry_hudson_and_diversity(dev, seed):
    """The two-group Hudson sweep and the one-group diversity sweep over the same random geometries: counts exactly,
    per-site pi / theta / D_xy against the formulas evaluated in numpy float64 (1e-12: same operations, other order)."""
    rng = np.random.default_rng(77000 + seed)
    N = int(rng.choice(WIDTHS))
    S = int(rng.choice(ROWS))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    H = N * ploidy
    max_allele = int(rng.choice([1, 1, 2, 3, 4, 7, 9]))
    p_missing = float(rng.choice([0.0, 0.0, 0.03, 0.3]))
    data = rng.integers(0, max_allele + 1, size=(S, H), dtype=np.uint8)
    data[rng.random((S, H)) < 0.5] = 0
    data[0, 0] = max_allele
    miss = rng.random((S, H)) < p_missing if p_missing > 0 else np.zeros((S, H), dtype=bool)
    data[miss] = 0
    words = None
    if p_missing > 0:
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        pad = (-len(bits)) % 8
        words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8").copy()
    dm = dev.DeviceMatrix.from_host(data.reshape(-1), words, S, N, ploidy, max_allele)
    masks = (rng.random((2, H)) < rng.choice([0.1, 0.5, 0.9])).astype(np.uint8)
    masks[0, 0] = 1
    counts = np.stack([np.stack([(((data == a) & ~miss) * masks[g][None, :]).sum(axis=1) for g in range(2)])
                       for a in range(max_allele + 1)]).astype(np.float64)           # [A][2][S]
    n = counts.sum(axis=0)                                                            # [2][S]
    with np.errstate(divide="ignore", invalid="ignore"):
        freq = counts / n[None]
        pi = np.where(n >= 2, n / (n - 1.0) * (1.0 - (freq ** 2).sum(axis=0)), np.nan)
        dxy = np.where((n[0] > 0) & (n[1] > 0), np.clip(1.0 - (freq[:, 0] * freq[:, 1]).sum(axis=0), 0.0, 1.0), np.nan)
    distinct = (counts > 0).sum(axis=0)

    def close(got, exp, what):
        assert np.array_equal(np.isnan(got), np.isnan(exp)), what
        ok = ~np.isnan(exp)
        assert np.allclose(got[ok], exp[ok], rtol=1e-12, atol=1e-13), what

    hs = dev.hudson_sweep(dm, dev.Groups(dm, masks), dev.FORMULA_DENSE)
    assert np.array_equal(hs.sites["called"], n.astype(np.uint32))
    if max_allele >= 1:
        assert np.array_equal(hs.sites["alt"], counts[1].astype(np.uint32))
    close(hs.sites["dxy"], dxy, "dxy")
    close(hs.sites["pi1"], pi[0], "pi1")
    close(hs.sites["pi2"], pi[1], "pi2")
    for g in range(2):
        assert hs.pop[g]["segregating_sites"] == int((distinct[g] >= 2).sum())
    dv = dev.diversity_sites(dm, dev.Groups(dm, masks[:1]))
    assert np.array_equal(dv.called, n[0].astype(np.uint32))
    assert np.array_equal(dv.distinct, distinct[0].astype(np.uint32))
    close(dv.pi, pi[0], "site pi")
    harmonic = np.concatenate([[0.0], np.cumsum(1.0 / np.arange(1, H + 2))])
    theta = np.where(n[0] >= 2, np.where(distinct[0] >= 2, 1.0 / harmonic[np.maximum(n[0].astype(int) - 1, 1)], 0.0), np.nan)
    close(dv.theta, theta, "site theta")


@pytest.mark.parametrize("N,S,max_allele,p_missing", [(600, 300, 1, 0.0), (513, 257, 3, 0.05), (300, 1000, 2, 0.0), (257, 129, 1, 0.2), (300, 200, 6, 0.03), (130, 700, 5, 0.0)])
def test_pairwise_gram_multi_tile(dev, N, S, max_allele, p_missing):
    """fmh_pairwise_differences across several 256-sample tiles (diagonal and off-diagonal tile pairs, K slices, ragged edges)
    against the same Gram products in numpy int64: diff = sum_s len_i len_j - sum_a cnt_i(a) cnt_j(a), both = sum_s valid_i valid_j."""
    rng = np.random.default_rng(N * 7 + S)
    g = rng.integers(0, max_allele + 1, size=(S, N, 2), dtype=np.uint8)
    miss = rng.random((S, N, 2)) < p_missing if p_missing > 0 else np.zeros((S, N, 2), dtype=bool)
    g[miss] = 0
    words = None
    if p_missing > 0:
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        pad = (-len(bits)) % 8
        words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8").copy()
    dm = dev.DeviceMatrix.from_host(g.reshape(-1), words, S, N, 2, max_allele)
    diff, both = dev.pairwise_differences(dm, N)
    # sparse-model genotype length: the prefix of called alleles (CompressedGenotypes::get)
    called = ~miss
    length = called[:, :, 0].astype(np.int64) + (called[:, :, 0] & called[:, :, 1]).astype(np.int64)   # [S][N]
    valid = (length > 0).astype(np.int64)
    exp_diff = length.T @ length
    for a in range(max_allele + 1):
        in_prefix = np.stack([called[:, :, 0], called[:, :, 0] & called[:, :, 1]], axis=2)
        cnt = ((g == a) & in_prefix).sum(axis=2).astype(np.int64)
        exp_diff -= cnt.T @ cnt
    exp_both = valid.T @ valid
    iu = np.triu_indices(N, k=1)
    assert np.array_equal(diff[iu].astype(np.int64), exp_diff[iu])
    assert np.array_equal(both[iu].astype(np.int64), exp_both[iu])


def _gram_reference(g):
    cnt1 = g.sum(axis=2).astype(np.float64)          # biallelic, nothing missing: allele-1 count per (site, sample)
    cnt0 = 2.0 - cnt1
    S = g.shape[0]
    return 4.0 * S - (cnt0.T @ cnt0 + cnt1.T @ cnt1)  # exact in float64 (values << 2^53)


def test_pairwise_gram_many_k_slices(dev):
    """70 000 sites x 300 samples: several K slices per XCD, split-K atomics."""
    rng = np.random.default_rng(77)
    S, N = 70_000, 300
    g = (rng.random((S, N, 2)) < rng.beta(0.8, 0.8, size=(S, 1, 1))).astype(np.uint8)
    dm = dev.DeviceMatrix.from_host(g.reshape(-1), None, S, N, 2, 1)
    diff, both = dev.pairwise_differences(dm, N)
    iu = np.triu_indices(N, k=1)
    assert np.array_equal(diff[iu].astype(np.float64), _gram_reference(g)[iu])
    assert (both[iu] == S).all()


def test_pairwise_gram_site_slabs():
    """A planes budget of 4 MiB forces the site axis through many slabs (own process: the budget is read once)."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ferromic_amd import device as dev\n"
        "from tests.test_gpu_device_fuzz import _gram_reference\n"
        "rng = np.random.default_rng(3)\n"
        "S, N = 20000, 260\n"
        "g = (rng.random((S, N, 2)) < 0.3).astype(np.uint8)\n"
        "dm = dev.DeviceMatrix.from_host(g.reshape(-1), None, S, N, 2, 1)\n"
        "diff, both = dev.pairwise_differences(dm, N)\n"
        "iu = np.triu_indices(N, k=1)\n"
        "assert np.array_equal(diff[iu].astype(np.float64), _gram_reference(g)[iu])\n"
        "assert (both[iu] == S).all()\n"
        "print('slabs ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, FMH_PD_PLANES_BYTES=str(4 << 20)))
    assert res.returncode == 0 and "slabs ok" in res.stdout, res.stderr[-2000:]


@pytest.mark.parametrize("N", [255, 256, 300, 512])
def test_pairwise_single_plane_and_general_routes_agree(N):
    """Biallelic cohorts without missing calls take the one-plane Gram (diff = ploidy (T_i + T_j) - 2 G_ij, T from an
    all-ones row after the last sample - also when N is a multiple of the 256-sample tile and the row opens a new tile);
    FMH_PD_TWO_PLANES forces the general two-plane route.  Both must equal numpy (own processes: the switch is read once).
    Ploidy 1 and 3 go through the per-allele loop of the planes kernel instead of the diploid fast path."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ferromic_amd import device as dev\n"
        "rng = np.random.default_rng(%d)\n"
        "for ploidy, S in ((2, 3000), (1, 700), (3, 900)):\n"
        "    g = (rng.random((S, %d, ploidy)) < rng.beta(0.8, 0.8, size=(S, 1, 1))).astype(np.uint8)\n"
        "    dm = dev.DeviceMatrix.from_host(g.reshape(-1), None, S, %d, ploidy, 1)\n"
        "    diff, both = dev.pairwise_differences(dm, %d)\n"
        "    c1 = g.sum(axis=2).astype(np.int64); c0 = ploidy - c1\n"
        "    exp = S * ploidy * ploidy - c0.T @ c0 - c1.T @ c1\n"
        "    iu = np.triu_indices(%d, k=1)\n"
        "    assert np.array_equal(diff[iu].astype(np.int64), exp[iu]), ploidy\n"
        "    assert (both[iu] == S).all()\n"
        "print('routes ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), N, N, N, N, N)
    # ... and every Gram route: the phase-interleaved kernel with its slab epilogue (default), with 64-bit atomics instead (FMH_PD_SLABS=0, or slabs
    # beyond the budget), and round 3's two-stage kernel with its own planes layout (FMH_PD_PHASED=0)
    for env in ({}, {"FMH_PD_TWO_PLANES": "1"}, {"FMH_PD_INT8": "1"}, {"FMH_PD_INT8": "1", "FMH_PD_TWO_PLANES": "1"},
                {"FMH_PD_SLABS": "0"}, {"FMH_PD_SLAB_BYTES": "4096", "FMH_PD_TWO_PLANES": "1"}, {"FMH_PD_PHASED": "0"},
                {"FMH_PD_PHASED": "0", "FMH_PD_INT8": "1", "FMH_PD_TWO_PLANES": "1"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert res.returncode == 0 and "routes ok" in res.stdout, (env, res.stderr[-2000:])


def test_pairwise_fp4_sums_beyond_f32_integers_are_split():
    """The FP4 Gram accumulates in f32, exact only up to 2^24: with every genotype 1|1 over 4.5 M sites a pair's
    product sum is 18 M > 2^24, so the host must cut K into items below the cap even when asked for one huge chunk
    (FMH_PD_KCHUNK).  Multi-allelic data with missing calls takes the same route through all its planes."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ferromic_amd import device as dev\n"
        "S, N = 4_500_000, 6\n"
        "g = np.ones((S, N, 2), dtype=np.uint8)\n"
        "g[:, 4, :] = 0\n"
        "g[::3, 5, 0] = 0\n"
        "dm = dev.DeviceMatrix.from_host(g.reshape(-1), None, S, N, 2, 1)\n"
        "diff, both = dev.pairwise_differences(dm, N)\n"
        "c1 = g.sum(axis=2).astype(np.int64); c0 = 2 - c1\n"
        "exp = S * 4 - c0.T @ c0 - c1.T @ c1\n"
        "iu = np.triu_indices(N, k=1)\n"
        "assert np.array_equal(diff[iu].astype(np.int64), exp[iu]), (diff[iu], exp[iu])\n"
        "assert (both[iu] == S).all()\n"
        "print('cap ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in ({"FMH_PD_KCHUNK": "100000000"}, {"FMH_PD_KCHUNK": "100000000", "FMH_PD_TWO_PLANES": "1"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert res.returncode == 0 and "cap ok" in res.stdout, (env, res.stderr[-2000:])
