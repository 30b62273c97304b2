"""Drop-in alias: ``import ferromic`` resolves to the MI355X implementation, which mirrors the reference's PyO3 module
(src/lib.rs:2227-2270) name for name.  The host side is C++ (``ferromic._core``, pybind11: ferromic_amd/csrc/pymodule.cpp)
over the C-ABI of ``libferromic_hip.so``; every statistic is computed by the HIP kernels behind it."""

from ferromic_amd import _abi as _abi

# Maps the PyTorch wheel's HIP runtime first when one is installed (one runtime per process, see _abi.load) and loads
# libferromic_hip.so, so that _core's dependency on it resolves to that same library; raises ImportError if it is missing.
_abi.load()

from ._core import (  # noqa: E402,F401
    ChromosomePcaResult,
    DiversitySite,
    FstEstimate,
    HudsonDxyResult,
    HudsonFstResult,
    HudsonFstSite,
    PairwiseDifference,
    Population,
    WcFstResult,
    WcFstSite,
    adjusted_sequence_length,
    chromosome_pca,
    chromosome_pca_to_file,
    global_pca,
    hudson_dxy,
    hudson_fst,
    hudson_fst_sites,
    hudson_fst_with_sites,
    inversion_allele_frequency,
    nucleotide_diversity,
    pairwise_differences,
    per_chromosome_pca,
    per_site_diversity,
    segregating_sites,
    watterson_theta,
    wc_fst,
    wc_fst_components,
    get_device,
    set_device,
)

__version__ = "0.1.4"  # the reference crate's version (Cargo.toml:3), which lib.rs:2229 exports
# lib.rs:2229-2239 build attributes; the native layer is HIP/C++, not Rust, but tooling reads these
__rust_profile__ = "release"
__rust_opt_level__ = "3"
__debug_build__ = False
