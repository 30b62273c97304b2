"""Shared builders for parity tests: random cohorts in the reference's data model plus the
oracle-side expectations, so GPU tests read like the reference's own tests."""

from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np

from oracle import ferromic_ref as R


def random_sparse_variants(rng: random.Random, n_sites: int, n_samples: int, max_allele: int = 1,
                           p_missing: float = 0.0, p_haploid: float = 0.0, all_missing_rows: int = 0):
    """Variants in the sparse model (None = missing sample, 1-allele genotype = haploid call)."""
    rows = []
    pos = 0
    for s in range(n_sites):
        pos += rng.randint(1, 49)
        freq = [rng.random() for _ in range(max_allele + 1)]
        tot = sum(freq)
        cum = np.cumsum([f / tot for f in freq])

        def draw():
            u = rng.random()
            return int(np.searchsorted(cum, u, side="right").clip(0, max_allele))

        row = []
        for i in range(n_samples):
            if s < all_missing_rows or rng.random() < p_missing:
                row.append(None)
            elif rng.random() < p_haploid:
                row.append([draw()])
            else:
                row.append([draw(), draw()])
        rows.append((pos, row))
    return [R.make_variant(p, g) for p, g in rows]


def dense_from_variants(variants, n_samples) -> R.DenseGenotypeMatrix:
    m = R.DenseGenotypeMatrix.from_variants(variants, n_samples)
    assert m is not None
    return m


def random_dense_matrix(rng: np.random.Generator, n_sites: int, n_samples: int, ploidy: int = 2,
                        max_allele: int = 1, p_missing: float = 0.0) -> R.DenseGenotypeMatrix:
    """A dense matrix in the reference host layout (per-allele missing bits, like from_numpy)."""
    H = n_samples * ploidy
    freq = rng.beta(0.8, 0.8, size=(n_sites, 1))
    if max_allele <= 1:
        data = (rng.random((n_sites, H)) < freq).astype(np.uint8)
    else:
        data = rng.integers(0, max_allele + 1, size=(n_sites, H), dtype=np.uint8)
        data[rng.random((n_sites, H)) < 0.6] = 0
        data[0, 0] = max_allele
    missing = None
    if p_missing > 0:
        miss = rng.random((n_sites, H)) < p_missing
        data[miss] = 0
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        pad = (-len(bits)) % 8
        words = np.frombuffer(np.concatenate([bits, np.zeros(pad, np.uint8)]).tobytes(), dtype="<u8")
        missing = [int(w) for w in words]
    return R.DenseGenotypeMatrix(bytes(data.reshape(-1)), missing, n_sites, n_samples, ploidy, int(data.max()) if n_sites else 0)


def missing_words_np(m: R.DenseGenotypeMatrix) -> Optional[np.ndarray]:
    return None if m.missing is None else np.array(m.missing, dtype=np.uint64)


def haps_for_samples(samples: Sequence[int]) -> List[Tuple[int, int]]:
    return [(s, side) for s in samples for side in (0, 1)]


def opt(x: Optional[float]) -> float:
    return float("nan") if x is None else x


def assert_bits_equal(actual: np.ndarray, expected: Sequence[float], what: str):
    exp = np.array(expected, dtype=np.float64)
    a = np.asarray(actual, dtype=np.float64)
    assert a.shape == exp.shape, what
    both_nan = np.isnan(a) & np.isnan(exp)
    same = (a.view(np.uint64) == exp.view(np.uint64)) | both_nan  # the sign of a zero counts: +0.0 and -0.0 are different bits
    if not same.all():
        i = int(np.argmin(same))
        raise AssertionError(f"{what}: first mismatch at {i}: gpu={a[i]!r} oracle={exp[i]!r}")


def rel_close(a: float, b: float, rel: float = 1e-9, abs_tol: float = 1e-12) -> bool:
    """north_star tolerance: 1e-9 relative (1e-12 absolute near zero)."""
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    return math.isclose(a, b, rel_tol=rel, abs_tol=abs_tol)


def assert_close_rel(actual: np.ndarray, expected: Sequence[float], what: str, rel: float = 1e-12):
    exp = np.array(expected, dtype=np.float64)
    a = np.asarray(actual, dtype=np.float64)
    assert a.shape == exp.shape, what
    assert np.array_equal(np.isnan(a), np.isnan(exp)), f"{what}: None pattern differs"
    ok = np.isnan(a) | np.isclose(a, exp, rtol=rel, atol=1e-15)
    if not ok.all():
        i = int(np.argmin(ok))
        raise AssertionError(f"{what}: first mismatch at {i}: gpu={a[i]!r} oracle={exp[i]!r}")
