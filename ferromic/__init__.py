"""Drop-in alias: ``import ferromic`` resolves to the MI355X implementation (ferromic_amd.api),
which mirrors the reference's PyO3 module (src/lib.rs:2227-2270) name for name."""

from ferromic_amd.api import (  # noqa: F401
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
    __version__,
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
)

# lib.rs:2229-2239 build attributes; the native layer is HIP/C++, not Rust, but tooling reads these
__rust_profile__ = "release"
__rust_opt_level__ = "3"
__debug_build__ = False
