"""Host-only parts of the drop-in `ferromic` module (no genotype data touches these): scalar
statistics, interval arithmetic, input validation and error texts the reference's tests grep for
(src/pytests/test_ferromic.py)."""

import math

import numpy as np
import pytest

import ferromic as fm


def test_module_surface_matches_reference():
    """lib.rs:2227-2270: 10 classes, 17 functions, 4 attributes."""
    classes = ["Population", "PairwiseDifference", "ChromosomePcaResult", "DiversitySite", "HudsonDxyResult",
               "HudsonFstSite", "HudsonFstResult", "FstEstimate", "WcFstSite", "WcFstResult"]
    functions = ["segregating_sites", "nucleotide_diversity", "watterson_theta", "pairwise_differences",
                 "per_site_diversity", "hudson_dxy", "hudson_fst", "hudson_fst_sites", "hudson_fst_with_sites",
                 "wc_fst", "wc_fst_components", "chromosome_pca", "chromosome_pca_to_file", "per_chromosome_pca",
                 "global_pca", "adjusted_sequence_length", "inversion_allele_frequency"]
    for name in classes:
        assert isinstance(getattr(fm, name), type), name
    for name in functions:
        assert callable(getattr(fm, name)), name
    for attr in ("__version__", "__rust_profile__", "__rust_opt_level__", "__debug_build__"):
        assert hasattr(fm, attr)
    # the host side is the compiled extension (C++/pybind11 over the C-ABI), not a Python re-implementation
    import ferromic._core as core
    assert core.__file__.endswith(".so") and fm.Population is core.Population and fm.hudson_fst is core.hudson_fst
    with pytest.raises(NotImplementedError):
        fm.global_pca({}, [], "out")
    with pytest.raises(ValueError, match="sequence_length"):
        fm.pairwise_differences([], 3, 0)


def test_watterson_theta_matches_rust_implementation(kats):
    theta = fm.watterson_theta(3, 4, 100)
    assert math.isclose(theta, 3 / (1 + 1 / 2 + 1 / 3) / 100, rel_tol=1e-12)
    for c in kats["watterson_theta"]["cases"]:
        if c["n"] > 1 and c["L"] > 0:
            got = fm.watterson_theta(c["S"], c["n"], c["L"])
            if "abs_tol" in c:
                assert abs(got - c["expected"]) < c["abs_tol"]
            else:
                assert math.isclose(got, c["expected"], rel_tol=c["rel_tol"])


def test_watterson_theta_requires_multiple_samples():
    with pytest.raises(ValueError) as excinfo:
        fm.watterson_theta(1, 1, 100)
    assert "sample_count" in str(excinfo.value)
    with pytest.raises(ValueError) as excinfo:
        fm.watterson_theta(1, 5, 0)
    assert "sequence_length" in str(excinfo.value)


def test_adjusted_sequence_length(kats):
    for c in kats["adjusted_sequence_length"]["cases"]:
        got = fm.adjusted_sequence_length(c["start"], c["end"], allow=c["allow"], mask=c["mask"])
        assert got == c["expected"]
    assert fm.adjusted_sequence_length(1, 100) == 100
    with pytest.raises(ValueError, match="end must be greater"):
        fm.adjusted_sequence_length(10, 5)
    with pytest.raises(ValueError, match="interval end"):
        fm.adjusted_sequence_length(1, 100, mask=[(5, 2)])


def test_population_rejects_non_positive_sequence_length():
    with pytest.raises(ValueError) as excinfo:
        fm.Population("demo", [], [], 0)
    assert "sequence_length" in str(excinfo.value)
    with pytest.raises(ValueError, match="sequence_length"):
        fm.Population.from_numpy("demo", np.zeros((1, 1, 2), np.uint8), [1], [(0, 0)], -3)
    with pytest.raises(ValueError, match="sequence_length"):
        fm.nucleotide_diversity([], [(0, 0), (0, 1)], 0)


def test_inversion_allele_frequency_counts_haplotypes():
    sample_map = {"sampleA": (0, 1), "sampleB": (1, 1), "sampleC": (2, 255)}
    assert fm.inversion_allele_frequency(sample_map) == pytest.approx(0.75)
    assert fm.inversion_allele_frequency({"Sample1": (0, 1), "Sample2": (0, 1), "Sample3": (0, 0)}) == pytest.approx(2.0 / 6.0, abs=1e-6)  # stats_tests.rs:1770-1826
    assert fm.inversion_allele_frequency({"x": (2, 3)}) is None
    with pytest.raises(ValueError, match="sample_to_group must be a dict"):
        fm.inversion_allele_frequency([("a", (0, 1))])


def test_population_attributes_and_coercions():
    pop = fm.Population({"haplotype_group": 1}, [], [(0, "L"), [1, "right"], (2, 1)], 10, ["a", "b", "c"])
    assert pop.id == 1 and pop.haplotype_group == 1 and pop.label is None
    assert pop.haplotypes == [(0, 0), (1, 1), (2, 1)]
    assert pop.sequence_length == 10 and pop.variant_count == 0 and pop.sample_names == ["a", "b", "c"]
    assert repr(pop) == "Population(haplotype_group 1, haplotypes=3, variants=0, sequence_length=10)"
    named = fm.Population("demo", [], [], 5)
    assert named.id == "demo" and named.label == "demo" and named.haplotype_group is None
    child = named.with_haplotypes(7, [(0, 0)])
    assert child.haplotype_group == 7 and child.haplotypes == [(0, 0)]
    with pytest.raises(ValueError, match="haplotype side"):
        fm.Population("x", [], [(0, 2)], 5)
    with pytest.raises(ValueError, match="haplotype_group ids must be <= 255"):
        fm.Population(300, [], [], 5)
    with pytest.raises(ValueError, match="variant tuples must have length 2"):
        fm.Population("x", [(1, [[0, 0]], 3)], [], 5)
    with pytest.raises(ValueError, match="mapping missing required field"):
        fm.Population("x", [{"genotypes": [[0, 0]]}], [], 5)


def test_from_numpy_validation():
    g = np.zeros((2, 3, 2), dtype=np.uint8)
    with pytest.raises(ValueError, match="positions length 1 does not match variant dimension 2"):
        fm.Population.from_numpy("p", g, [5], [(0, 0)], 10)
    with pytest.raises(ValueError, match="genotypes must be a numpy.ndarray"):
        fm.Population.from_numpy("p", g.astype(np.float32), [5, 6], [(0, 0)], 10)
    with pytest.raises(ValueError, match="allele values must be <= 255"):
        fm.Population.from_numpy("p", np.full((2, 3, 2), 300, dtype=np.uint16), [5, 6], [(0, 0)], 10)
    with pytest.raises(ValueError, match="positions must be a sequence of integers"):
        fm.Population.from_numpy("p", g, np.array([1.0, 2.0]), [(0, 0)], 10)
    pop = fm.Population.from_numpy("p", g, np.array([5, 6], dtype=np.uint32), [(0, 0), (0, 1)], 10)
    assert pop.variant_count == 2


def test_result_reprs():
    e = fm.FstEstimate("calculable", 0.25, 1.0, 3.0, 2)
    assert repr(e) == "FstEstimate(state='calculable', value=0.250000, sum_a=Some(1.0), sum_b=Some(3.0), sites=Some(2))"
    assert e.components() == (0.25, 1.0, 3.0, 2)
    with pytest.raises(AttributeError):
        e.value = 3
    assert repr(fm.HudsonDxyResult(None)) == "HudsonDxyResult(d_xy=None)"
    assert repr(fm.DiversitySite(5, 0.5, 0.25)) == "DiversitySite(position=5, pi=0.500000, watterson_theta=0.250000)"
