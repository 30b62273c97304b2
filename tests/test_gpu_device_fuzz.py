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


@pytest.mark.parametrize("seed", range(CASES))
def test_random_geometry_counts(dev, seed):
    rng = np.random.default_rng(9000 + seed)
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
