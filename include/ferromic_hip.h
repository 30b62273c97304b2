/*
 * ferromic_hip.h — C-ABI of libferromic_hip.so: the MI355X (gfx950) device layer under the
 * ferromic per-site diversity / FST hot path.
 *
 * The reference (SauersML/ferromic) has no FFI for this path: PyO3 calls the Rust functions in
 * src/stats.rs directly.  This header is the boundary a Rust (or any) host would bind instead of
 * those functions; every entry point cites the reference code it replaces.  INTEGRATION.md shows
 * the `extern "C"` block a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes; no C++/torch types.  Pointers named d_* are DEVICE pointers
 *     (hipMalloc or a torch tensor's data_ptr), h_* are HOST pointers; every d_* output may be
 *     NULL to skip that track.
 *   - every function returns a status (FMH_OK == 0) and records a thread-local message readable
 *     through fmh_last_error().
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls that fill host
 *     totals synchronise that stream before returning.
 *   - per-site f64 tracks encode the reference's Option<f64>::None as NaN.
 *   - rows are variant sites in matrix order; a sweep over [row_begin, row_begin+row_count) writes
 *     per-site outputs at index (row - row_begin).
 */
#ifndef FERROMIC_HIP_H
#define FERROMIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMH_ABI_VERSION 3
#define FMH_MAX_GROUPS 8 /* populations per sweep */
#define FMH_MAX_PAIRS 28 /* FMH_MAX_GROUPS choose 2 */

enum fmh_status {
  FMH_OK = 0,
  FMH_ERR_INVALID = 1,     /* bad argument (message says which) */
  FMH_ERR_HIP = 2,         /* HIP runtime error */
  FMH_ERR_NO_DEVICE = 3,   /* no usable GPU: the product path has no CPU fallback */
  FMH_ERR_UNSUPPORTED = 4
};

/* Which of the reference's algebraically-equal formula sets a sweep reproduces bit-for-bit. */
enum fmh_formula {
  /* sparse Variant path: pi_from_components (stats.rs:2723-2733), dxy_from_counts (2907-2935),
   * hudson_site_from_variant (2969-3014), compute_pi_metrics_fast (2761-2821) */
  FMH_FORMULA_SPARSE = 0,
  /* dense matrix path: dense_pi_from_counts (1700-1709), dense_dxy_from_biallelic_counts
   * (1712-1733), dense_hudson_sites_{biallelic,general} (3072-3278), calculate_pi_dense
   * (4434-4597); the no-missing biallelic arms (3218-3273, 4485-4523) are taken automatically
   * when the matrix has no missing mask and max_allele <= 1, exactly as the reference does */
  FMH_FORMULA_DENSE = 1,
  /* build_dense_population_summary (1367-1470): like DENSE but per-site pi is always
   * dense_pi_from_counts (1392, 1409), also on a mask-free matrix */
  FMH_FORMULA_SUMMARY = 2
};

typedef struct fmh_matrix fmh_matrix; /* DenseGenotypeMatrix, stats.rs:250-331 */
typedef struct fmh_groups fmh_groups; /* P column memberships, stats.rs:1246-1295 / 1204-1244 / 1093-1159 */

const char* fmh_last_error(void);
int fmh_abi_version(void);
int fmh_device_count(int* h_count);
/* name, CU count and total memory of a device (for reports) */
int fmh_device_info(int device, char* h_name, size_t name_cap, int* h_compute_units, uint64_t* h_total_mem);

/* ---- options ------------------------------------------------------------------------------------
 * Per-process switches that route kernels or size launches (tests, measurements, and the two a deployment may want:
 * FMH_LAYOUT and FMH_COMM_TRANSPORT).  Keys are the names of the environment variables of the same meaning; the FMH_*
 * environment is read ONCE, when the library first needs an option, and never again - nothing on a launch path calls
 * getenv - so a later change goes through fmh_set_option (value NULL = back to what the process
 * started with: the environment's value, else the default; integers, or the words listed):
 *   FMH_LAYOUT (bytes | packed)   FMH_MASK_MODE (1 | 2)   FMH_DEFER_TILES (1..16)   FMH_PACKED_LPR (4 | 16)   FMH_PACKED_UNROLL
 *   FMH_PACKED_NO_PREFETCH   FMH_COUNTS_MFMA (1 | 2)   FMH_GRID_PER_CU   FMH_GRID_BLOCKS   FMH_MAX_OCC   FMH_UNROLL   FMH_PITCH_ALIGN
 *   FMH_COMM_TRANSPORT (host | rccl)   FMH_UPLOAD_THREADS   FMH_PD_TWO_PLANES   FMH_PD_INT8   FMH_PD_PLANES_BYTES   FMH_PD_KCHUNK
 *   FMH_PD_SB   FMH_PD_OCC   FMH_PIPE   FMH_GRAPH
 * Values are atomics: setting one while another thread launches is safe (that launch sees the old or the new value). */
int fmh_set_option(const char* key, const char* value_or_null);
int fmh_get_option(const char* key, long long* h_value);

/* ---- raw device memory helpers for hosts that do not bring their own allocator --------------- */
int fmh_device_alloc(int device, size_t bytes, void** d_out);
/* Blocks are recycled through a pool without stream ordering: the caller's outstanding stream work on the block must be
 * complete (every fmh_* call that fills host totals has synchronised its stream; fmh_device_zero has not). */
int fmh_device_free(int device, void* d_ptr);
/* `stream` arguments throughout this header are hipStream_t handles passed as void*.  NULL is HIP's legacy default stream: ordered against
 * every blocking stream of the device - which is what a host that also enqueues work of its own on the default stream wants, and a
 * device-wide serialisation point when several host threads use one GPU.  FMH_STREAM_PER_THREAD is the calling thread's own stream
 * (hipStreamPerThread): not ordered against the default stream or other threads.  run_vcf's region workers pass it to the copies of
 * results that a synchronous fmh_* call has already completed. */
#define FMH_STREAM_PER_THREAD ((void*)2)
int fmh_copy_to_host(int device, void* h_dst, const void* d_src, size_t bytes, void* stream);
int fmh_copy_to_device(int device, void* d_dst, const void* h_src, size_t bytes, void* stream);
/* zero-fills device memory, stream-ordered (no synchronisation): accumulators such as fmh_pairwise_differences' outputs */
int fmh_device_zero(int device, void* d_ptr, size_t bytes, void* stream);
int fmh_stream_synchronize(int device, void* stream);
/* Frees the scratch the library keeps between calls on `device` (the sample-major planes of
 * fmh_pairwise_differences, up to 8 GiB); it is re-created on demand. */
int fmh_device_release_scratch(int device);

/* ---- genotype matrix (replaces DenseGenotypeMatrix::new, stats.rs:261-296) ------------------- */
/*
 * Upload a site-major matrix in the reference's host layout: data[site*stride + sample*ploidy + side],
 * stride = samples*ploidy (stats.rs:293); optional missing bitset, one bit per linear entry,
 * LSB-first in u64 words (stats.rs:1298-1302; lib.rs:1188-1189).
 * Resident layout: with max_allele <= 7 (every biallelic and every SNP cohort) the matrix is kept BIT-PACKED - one bit
 * plane per allele bit (1, 2 or 3) plus one "called" plane, 128 columns per 16-byte vector - and the sweeps read 1/8 (1/4 with
 * alleles 2..3, 3/8 with 4..7) of the bytes the u8 layout would cost; the rows are packed on the host (threads, SSE2) into pinned
 * staging, so that only the planes cross PCIe.  Other matrices
 * (max_allele > 7, rows beyond 600 000 columns, FMH_LAYOUT=bytes) keep the u8 rows, padded to a 16-byte pitch, with the
 * missing bitset re-laid as one "called" bit-row per site.  A called value above max_allele is a caller error: the packed
 * layout would lose its high bits, so fmh_matrix_create (while it packs the rows on the host) and fmh_matrix_pack (on the
 * device) detect it and return FMH_ERR_INVALID; the u8 layout keeps the bytes as they are and only uses max_allele as a
 * loop bound.  fmh_matrix_create_packed takes the caller's planes at their word.
 */
int fmh_matrix_create(const uint8_t* h_data, const uint64_t* h_missing_or_null, size_t variants,
                      size_t samples, size_t ploidy, uint8_t max_allele, int device, fmh_matrix** out);
/*
 * The same for a host that already holds BIT PLANES (run_vcf after ingest, a cohort stored packed): plane k = bit k of the allele
 * value (h_plane1 iff max_allele >= 2, h_plane2 iff max_allele >= 4), h_called_or_null = 1 bits for called entries; a row is
 * h_pitch bytes, column c is bit (c & 7) of byte (c >> 3), bits past the last column zero.  No conversion: ceil(H / 8) bytes per site
 * and plane cross PCIe.  With h_pitch = the device's own plane pitch - ceil(H / 8) rounded up to 16 - each plane goes up in ONE copy
 * (then the padding bytes of a row must be zero as well); any other pitch takes a pitched copy, a descriptor per row, which is slow for
 * millions of short rows.  The planes are taken at their word (no max_allele check is possible).
 */
int fmh_matrix_create_packed(const uint8_t* h_plane0, const uint8_t* h_plane1_or_null, const uint8_t* h_plane2_or_null,
                             const uint8_t* h_called_or_null, size_t h_pitch, size_t variants, size_t samples, size_t ploidy,
                             uint8_t max_allele, int device, fmh_matrix** out);
/* Allocate an uninitialised device matrix (filled by fmh_matrix_generate or by the caller). */
int fmh_matrix_alloc(size_t variants, size_t samples, size_t ploidy, int with_missing, uint8_t max_allele,
                     int device, fmh_matrix** out);
/* Wrap caller-owned device memory (e.g. a torch uint8 tensor): d_data rows of `pitch` bytes
 * (pitch % 16 == 0, pitch >= samples*ploidy); d_called_bits_or_null rows of `bits_pitch` bytes
 * (bit h of a row set = entry h is called; bits_pitch % 4 == 0, 4-byte aligned). The wrapper never frees them. */
int fmh_matrix_wrap(void* d_data, size_t pitch, void* d_called_bits_or_null, size_t bits_pitch,
                    size_t variants, size_t samples, size_t ploidy, uint8_t max_allele, int device,
                    fmh_matrix** out);
/* Build the bit-packed image of a matrix that holds u8 rows (fmh_matrix_alloc + fmh_matrix_generate, fmh_matrix_wrap);
 * the sweeps use it from then on.  release_bytes != 0 frees the u8 rows of a matrix the library owns (a wrapped matrix
 * keeps the caller's memory).  FMH_ERR_UNSUPPORTED when max_allele > 7 or the rows are too wide; a no-op when the
 * matrix is already packed and holds no bytes. */
int fmh_matrix_pack(fmh_matrix* m, int release_bytes);
int fmh_matrix_destroy(fmh_matrix* m);
/* geometry: any pointer may be NULL */
int fmh_matrix_info(const fmh_matrix* m, size_t* variants, size_t* samples, size_t* ploidy, size_t* pitch,
                    size_t* bits_pitch, int* has_missing, uint8_t* max_allele, int* device);
/* the u8 rows and called bit-rows; NULL for a matrix that holds only its packed image */
int fmh_matrix_device_ptrs(const fmh_matrix* m, void** d_data, void** d_called_bits);
/* Copy back in the reference host layout (tests / oracle). h_missing_or_null must hold
 * ceil(variants*stride/64) words when the matrix has a mask.  The byte of a MISSING entry is 0 whichever route built the
 * matrix (the planes of a missing entry are masked with the called plane on the way out). */
int fmh_matrix_download(const fmh_matrix* m, uint8_t* h_data, uint64_t* h_missing_or_null);
/* max over called entries, computed on the device (what from_variants computes at stats.rs:490) */
int fmh_matrix_scan_max_allele(const fmh_matrix* m, uint8_t* h_max, void* stream);

/*
 * Counter-based synthetic cohort written straight into HBM (bench + parity at sizes no host could
 * hold): entry (site, column) = 1 iff hash24(seed, global_site, column) < threshold[pop_of_column][site],
 * and is missing iff hash24(seed ^ K, global_site, column) < missing_threshold24.  `h_thresholds24` is
 * [n_pops][variants] (values in [0, 2^24]); `h_pop_of_column` is [samples*ploidy] with entries < n_pops.
 * `first_global_site` lets a rank generate its slab of a larger cohort.  oracle/dense_oracle.c holds
 * the identical generator so CPU and GPU see the same matrix without moving it.
 */
int fmh_matrix_generate(fmh_matrix* m, uint64_t seed, uint64_t first_global_site,
                        const uint32_t* h_thresholds24, const uint8_t* h_pop_of_column, int n_pops,
                        uint32_t missing_threshold24, void* stream);

/* ---- memberships --------------------------------------------------------------------------- */
/*
 * P populations as 0/1 column masks, h_column_mask[p*H + h], H = samples*ploidy.  A mask is what
 * DenseMembership::build (stats.rs:1252-1284) / HapMembership::build (1212-1238) reduce a
 * (sample, side) list to: duplicates collapse, out-of-range samples are dropped, the Right side is
 * dropped when ploidy <= 1.  Populations may overlap.  1 <= P <= FMH_MAX_GROUPS.
 * Row width: a packed matrix is swept with bit masks in LDS (P_padded * H / 8 bytes <= 150 KiB, P_padded = 1, 2, 4, 8);
 * on u8 rows the masks sit in LDS as bytes while P_padded * round_up(H, 1024) <= 150 KiB, as bits up to 8x that width,
 * and in global memory beyond (one or two groups at a time) — fmh_population_summaries and fmh_wc_sweep re-batch their
 * groups by themselves, fmh_hudson_sweep / fmh_diversity_sites need no batching.
 */
int fmh_groups_create(const fmh_matrix* m, const uint8_t* h_column_mask, int n_groups, fmh_groups** out);
int fmh_groups_destroy(fmh_groups* g);
int fmh_groups_sizes(const fmh_groups* g, int* n_groups, uint64_t* h_sizes /* [n_groups] mask popcounts */);

/* ---- per-population summary sweep ------------------------------------------------------------ */
typedef struct {
  uint64_t haplotype_capacity; /* mask popcount — DensePopulationSummary.haplotype_capacity (1466) */
  uint64_t segregating_sites;  /* sites with >= 2 distinct called alleles inside the population
                                  (== called>=2 && 0<alt<called when biallelic: 1389, 4049, 3891-4026, 3868-3889) */
  uint64_t uncallable_sites;   /* sites with called < 2 (1512-1516, 4391-4393, 4464, 4586) */
  double pi_sum;               /* sum of per-site pi over sites with called >= 2 (1392-1394, 4388-4390, ...) */
} fmh_pop_totals;

/*
 * Replaces build_dense_population_summary (stats.rs:1367-1470) for all P populations in ONE pass,
 * and supplies the scalars count_segregating_sites_dense (3891), calculate_pi_dense (4534),
 * calculate_pi (4317) and count_segregating_sites_for_haplotypes (3858) reduce to.
 * d_alt / d_called: [P][row_count] u32 (alt = number of allele-1 calls; meaningful when max_allele <= 1).
 */
int fmh_population_summaries(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                             int formula, uint32_t* d_alt, uint32_t* d_called, fmh_pop_totals* h_totals,
                             void* stream);

/* ---- Hudson pair sweep (groups 0 and 1 of `g`) ----------------------------------------------- */
typedef struct {
  /* HudsonSummaryTotals, stats.rs:1545-1552, from aggregate_hudson_components_from_summaries
   * (1554-1623).  Only meaningful when max_allele <= 1. */
  double numerator_sum, denominator_sum, pi1_sum, pi2_sum, dxy_sum_all;
  uint64_t dxy_uncallable_sites;
  /* hudson_component_sums over the per-site records (stats.rs:1625-1635) */
  double site_num_sum, site_den_sum;
  uint64_t sites_with_components;
  /* sum of per-site d_xy over sites where it is Some, and the count where it is None
   * (calculate_d_xy_hudson sparse fold 2476-2496; calculate_dxy_dense 2546-2596) */
  double site_dxy_sum;
  uint64_t site_dxy_skipped;
  fmh_pop_totals pop[2];
} fmh_hudson_totals;

typedef struct { /* SiteFstHudson as structure-of-arrays, stats.rs:536-555; all nullable */
  double* d_fst;
  double* d_dxy;
  double* d_pi1;
  double* d_pi2;
  double* d_num;
  double* d_den;
  uint32_t* d_alt;    /* [2][row_count] */
  uint32_t* d_called; /* [2][row_count]  (n1_called, n2_called) */
} fmh_hudson_sites;

/*
 * One pass over the rows yields what the reference computes in up to four passes:
 * build_dense_population_summary x2 (1367), aggregate_hudson_components_from_summaries (1554),
 * dense_hudson_sites (3060) or hudson_site_from_variant per site (2969), and the pi/Dxy auxiliaries
 * of calculate_hudson_fst_for_pair_core (3435-3599).  This is the "pi + Hudson FST" sweep
 * BASELINE.json's metric is quoted on.
 */
int fmh_hudson_sweep(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                     int formula, const fmh_hudson_sites* sites_or_null, fmh_hudson_totals* h_totals,
                     void* stream);

/*
 * The same Hudson totals and per-site records from the two populations' per-site COUNT TABLES (d_called / d_alt as fmh_population_summaries
 * writes them, biallelic) instead of a matrix: aggregate_hudson_components_from_summaries (stats.rs:1554-1623) takes two
 * DensePopulationSummary objects that need not come from one matrix - ferromic.hudson_fst / hudson_dxy of two Population.from_numpy
 * objects.  capacity1/2 = haplotype_capacity of the two populations (copied into h_totals->pop[]); any_missing != 0 when either matrix
 * has missing calls (selects the kernel twin the fused sweep would have taken; it matters to FMH_FORMULA_DENSE only).  Same per-site code as
 * fmh_hudson_sweep after the counting: the per-site values are its bits, the regional sums differ in the order of their additions.
 */
int fmh_hudson_from_counts(int device, const uint32_t* d_called1, const uint32_t* d_alt1, uint64_t capacity1, const uint32_t* d_called2,
                           const uint32_t* d_alt2, uint64_t capacity2, size_t row_count, int formula, int any_missing,
                           const fmh_hudson_sites* sites_or_null, fmh_hudson_totals* h_totals, void* stream);

/* ---- per-site diversity (population 0 of `g`) ------------------------------------------------- */
/*
 * Replaces the loop of calculate_per_site_diversity (stats.rs:4693-4750): d_pi / d_theta per site;
 * called < 2 -> (NaN, NaN); theta = 1/H_{called-1} when >= 2 distinct alleles else 0 (4717-4722),
 * with H_k summed ascending exactly like harmonic() (4234-4240).  Position filtering / masking
 * (4731-4743) stays on the host.
 */
int fmh_diversity_sites(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                        double* d_pi, double* d_theta, uint32_t* d_called, uint32_t* d_distinct,
                        fmh_pop_totals* h_totals, void* stream);

/* ---- fused region sweep: summaries + per-site diversity of groups 0 and 1 (+ the Hudson pair) in ONE read ---------------------- */
typedef struct { /* calculate_per_site_diversity's SiteDiversity (stats.rs:185-190) of BOTH groups; nullable */
  double* d_pi;    /* [2][row_count] */
  double* d_theta; /* [2][row_count] */
} fmh_pair_diversity_sites;
/*
 * What the reference's region driver (process.rs:2468-3653) computes for haplotype groups 0 and 1 of one variant set in separate walks -
 * process_variants per group (segregating sites, pi: calculate_pi_for_population 4599-4614; per-site diversity: 4628-4806) and the
 * Hudson pair (calculate_hudson_fst_for_pair_with_sites 3619-3625) - from one pass over the matrix:
 *   h_totals->pop[0..1]  the population summaries by `summary_formula`
 *   diversity            per-site pi / theta of both groups (always the sparse formulas, as the reference)
 *   hudson_formula >= 0  per-site Hudson records into `sites` and the Hudson totals by THAT formula set (the driver uses
 *                        FMH_FORMULA_DENSE for the regional pi of a diploid dense matrix and FMH_FORMULA_SPARSE for Hudson);
 *   hudson_formula  < 0  no Hudson part (h_totals' Hudson fields are zero; sites->d_alt / d_called are still written when given).
 * Every value is the bits of the separate calls (same kernel code after the counts); fmh_population_summaries + 2 x fmh_diversity_sites +
 * fmh_hudson_sweep read the matrix four times, this reads it once.
 */
int fmh_pair_region_sweep(const fmh_matrix* m, const fmh_groups* g /* exactly 2 groups */, size_t row_begin, size_t row_count,
                          int summary_formula, int hudson_formula_or_negative, const fmh_pair_diversity_sites* diversity_or_null,
                          const fmh_hudson_sites* sites_or_null, fmh_hudson_totals* h_totals, void* stream);

/* ---- Weir & Cockerham sweep -------------------------------------------------------------------- */
enum fmh_wc_state { /* FstEstimate variants, stats.rs:37-126 */
  FMH_WC_CALCULABLE = 0,
  FMH_WC_INDETERMINATE = 1,
  FMH_WC_NO_VARIANCE = 2,
  FMH_WC_INSUFFICIENT = 3
};

typedef struct {
  /* slot 0 = overall, slot 1+k = pair k in (0,1),(0,2),...,(P-2,P-1) order (stats.rs:1133-1142) */
  double sum_a[1 + FMH_MAX_PAIRS];
  double sum_b[1 + FMH_MAX_PAIRS];
  uint64_t informative_sites[1 + FMH_MAX_PAIRS]; /* sites whose per-site state != insufficient (2172-2203) */
  uint64_t sites_attempted;                      /* rows swept */
} fmh_wc_totals;

/*
 * Replaces calculate_fst_wc_at_site_with_membership (stats.rs:1814-2032) per row and the sums of
 * calculate_overall_fst_wc (2145-2374).  Groups are the sorted labels of SubpopulationMembership
 * (1104-1150).  Per-site outputs, all [(1+npairs)][row_count]: a, b (summed over every allele
 * present at the site, 1859-1985) and the FstEstimate state code; d_group_called [P][row_count]
 * are the population_sizes (1919-1923).  Alleles are discovered over ALL columns of the row
 * (1826-1837), not only grouped ones.
 */
int fmh_wc_sweep(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                 double* d_a, double* d_b, uint8_t* d_state, uint32_t* d_group_called,
                 fmh_wc_totals* h_totals, void* stream);

/*
 * The same statistics for ANY number of groups (2 <= n_groups <= FMH_MAX_GROUPS_MANY; labels are u8 in the
 * reference, so 256 covers every input).  h_column_mask is a HOST array [n_groups][columns] of 0/1 bytes in
 * sorted-label order.  Counting runs as summary sweeps over batches of 8 groups, the per-site W&C arithmetic
 * in a counts kernel with the same operand order as fmh_wc_sweep (bit-identical per-site a, b).  Device outputs
 * are [(1 + n_groups(n_groups-1)/2)][row_count] (a, b, state; any may be NULL) and [n_groups][row_count]
 * (d_group_called, may be NULL); host outputs h_sum_a / h_sum_b / h_informative_sites have 1 + npairs entries.
 */
#define FMH_MAX_GROUPS_MANY 256
int fmh_wc_sweep_many(const fmh_matrix* m, const uint8_t* h_column_mask, int n_groups, size_t row_begin,
                      size_t row_count, double* d_a, double* d_b, uint8_t* d_state, uint32_t* d_group_called,
                      double* h_sum_a, double* h_sum_b, uint64_t* h_informative_sites, void* stream);

/* ---- pairwise differences ------------------------------------------------------------------------- */
/*
 * Replaces the nested sample-pair loops of calculate_pairwise_differences (stats.rs:4158-4221) for the first
 * n_samples samples of the matrix (sparse model: a genotype is Some iff its first allele is called, and ends
 * at its first missing allele).  Row-major [n_samples][n_samples] device outputs, filled for i < j only:
 *   d_diff[i*n + j] = sum over sites where both genotypes are Some of the all-vs-all allele mismatches
 *                     (len_i*len_j - sum_a cnt_i(a)*cnt_j(a)),
 *   d_both[i*n + j] = number of sites where both genotypes are Some
 * (comparable sites = L*h_i*h_j - (variants - both)*h_i*h_j is host arithmetic, stats.rs:4182-4208).
 * The call ADDS into both buffers (several matrices of the same samples can be accumulated): zero them first (fmh_device_zero).  Needs ploidy <= 127 (int8 MFMA operands; ploidy <= 4 runs on FP4 MFMA, equally exact);
 * one allele-count plane per allele value 0..max_allele, a single plane when the matrix is biallelic with nothing missing.
 */
int fmh_pairwise_differences(const fmh_matrix* m, size_t n_samples, unsigned long long* d_diff,
                             unsigned long long* d_both, void* stream);

/* ---- multi-GPU: region sharding + RCCL reduce of the regional accumulators -------------------------------- */
/*
 * Sites are independent: rank r of G sweeps only its contiguous slab of the region and writes its own
 * per-site tracks; the only exchange is the SUM of the regional accumulators - what the reference's rayon fold/reduce
 * does across its worker threads (stats.rs:1365-1461 for the summaries, the serial sums of 1554-1623 and 2145-2374
 * for Hudson and W&C).  A communicator carries that sum over RCCL (xGMI inside a node): one ncclAllReduce per vector
 * type, a few hundred bytes, latency-bound.  RCCL is bound at run time (dlopen), so single-GPU hosts never load it.
 * Integer totals are exact; f64 totals depend on the reduction order (1e-9 contract, as with rayon).
 *
 *   one process per GPU (MPI / torch.distributed.run):  rank 0 calls fmh_comm_get_unique_id, ships the 128 bytes to the
 *     other ranks over any channel, every rank calls fmh_comm_init_rank.
 *   one process, several GPUs (run_vcf --devices):       fmh_comm_init_all; each device's thread then uses its own handle
 *     concurrently.  When a device appears more than once in the list (rehearsal on a one-GPU box) or
 *     FMH_COMM_TRANSPORT=host is set, the handles share an in-process rendezvous that sums in rank order on the host
 *     instead - RCCL cannot place two ranks on one GPU.
 */
typedef struct fmh_comm fmh_comm;
#define FMH_COMM_ID_BYTES 128
int fmh_comm_get_unique_id(void* h_id /*[FMH_COMM_ID_BYTES]*/);
int fmh_comm_init_rank(const void* h_id, int world, int rank, int device, fmh_comm** out);
int fmh_comm_init_all(const int* h_devices, int n, fmh_comm** h_out /*[n]*/);
/* A communicator of ONE rank that needs no transport (and no RCCL): what a single GPU uses for the pipelined
 * fmh_hudson_sweep_sharded_begin / _end calls when it scans many windows. */
int fmh_comm_init_local(int device, fmh_comm** out);
int fmh_comm_destroy(fmh_comm* c);
/* transport: 0 = RCCL, 1 = in-process host rendezvous, 2 = local (one rank, nothing to exchange) */
int fmh_comm_info(const fmh_comm* c, int* world, int* rank, int* device, int* transport);
/* One line for reports: "transport=rccl world=8 rank=3 device=3 rccl_library=<file the process bound> rccl_version=<ncclGetVersion> ..." */
int fmh_comm_describe(const fmh_comm* c, char* h_text, size_t cap);
/* For a rank that FAILED before its collective (allocation, upload, validation): wakes the peers of an in-process group, which return
 * FMH_ERR_INVALID from the collective they are waiting in instead of blocking for ever, and aborts the RCCL communicator
 * (ncclCommAbort).  The communicator only accepts fmh_comm_destroy afterwards.  Every sharded _begin validates its arguments before
 * it enqueues anything, so a rank whose _begin returned an error has not entered the collective and may call this. */
int fmh_comm_abort(fmh_comm* c);

/* Element-wise sum over all ranks, in place, blocking (collective: every rank calls it with the same lengths;
 * n_f64, n_u64 <= FMH_COMM_MAX_VALUES). */
#define FMH_COMM_MAX_VALUES 512
int fmh_allreduce_totals(fmh_comm* c, double* h_f64, size_t n_f64, uint64_t* h_u64, size_t n_u64);
/* The same in two halves: _begin enqueues the reduce on the communicator's own stream and returns, _end waits for it
 * and writes the sums.  One reduce may be in flight per communicator. */
int fmh_allreduce_totals_begin(fmh_comm* c, const double* h_f64, size_t n_f64, const uint64_t* h_u64, size_t n_u64);
int fmh_allreduce_totals_end(fmh_comm* c, double* h_f64, uint64_t* h_u64);

/*
 * Region-sharded Hudson sweep: this rank's rows [row_begin, row_begin + row_count) of ITS slab matrix, then the sum of the
 * regional accumulators over all ranks - fmh_hudson_sweep + fmh_allreduce_totals without a host hop in between: the
 * partials are finalised on the device, reduced there by RCCL on the communicator's stream and land in pinned memory.
 * _begin returns as soon as the work is enqueued (up to FMH_SHARDED_IN_FLIGHT sweeps per communicator), _end waits for
 * the OLDEST outstanding one and returns its region-wide totals (pop[].haplotype_capacity is the local mask popcount,
 * identical on every rank).  A loop `begin(k); if (k) end(k-1)` overlaps the reduce of one window with the sweep of the
 * next, which is what keeps small slabs (1.25 M sites x 5 000 haplotypes is 0.16 ms of kernel) from paying the
 * collective's latency; keeping two or three sweeps ahead (`begin(k); if (k >= 2) end(k-2)`) also covers a reduce that can only
 * start once a persistent sweep grid frees compute units, or that takes longer than one sweep.  fmh_hudson_sweep_sharded = _begin + _end.
 */
#define FMH_SHARDED_IN_FLIGHT 4
int fmh_hudson_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                   int formula, const fmh_hudson_sites* sites_or_null, void* stream);
int fmh_hudson_sweep_sharded_end(fmh_comm* c, fmh_hudson_totals* h_global_totals);
int fmh_hudson_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                             int formula, const fmh_hudson_sites* sites_or_null, fmh_hudson_totals* h_global_totals, void* stream);

/*
 * The same pipeline for the other two regional reductions north_star names ("RCCL reduce ... for the sum-a / sum-b and global pi
 * accumulators"): sweep -> finalise on the device -> grouped ncclAllReduce of the 64 + 64 accumulators on the communicator's stream ->
 * one D2H; no host hop, FMH_SHARDED_IN_FLIGHT sweeps of ANY kind may be in flight per communicator and each _end collects the oldest
 * (it must be of its kind).  W&C: calculate_overall_fst_wc's sums (stats.rs:2145-2374) - two to eight groups run as one fused kernel (five
 * to eight: with alleles up to 3); more than 3 alleles with five to eight groups, or rows too wide for all masks, run the blocking fmh_wc_sweep
 * inside _begin and only the reduce is pipelined.
 * Population summaries: build_dense_population_summary's scalars (stats.rs:1367-1470) for up to FMH_MAX_GROUPS populations.
 * Per-site tracks stay on the rank that owns the rows; haplotype_capacity is the local mask popcount (identical on every rank).
 */
int fmh_wc_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, double* d_a,
                               double* d_b, uint8_t* d_state, uint32_t* d_group_called, void* stream);
int fmh_wc_sweep_sharded_end(fmh_comm* c, fmh_wc_totals* h_global_totals);
int fmh_wc_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, double* d_a, double* d_b,
                         uint8_t* d_state, uint32_t* d_group_called, fmh_wc_totals* h_global_totals, void* stream);
int fmh_population_summaries_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                           int formula, uint32_t* d_alt, uint32_t* d_called, void* stream);
int fmh_population_summaries_sharded_end(fmh_comm* c, fmh_pop_totals* h_global_totals /*[n_groups]*/);
int fmh_population_summaries_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int formula,
                                     uint32_t* d_alt, uint32_t* d_called, fmh_pop_totals* h_global_totals, void* stream);
/* fmh_pair_region_sweep over this rank's slab, totals (population summaries + Hudson) summed over the ranks the same way; collected by
 * fmh_hudson_sweep_sharded_end (the totals struct is the same) */
int fmh_pair_region_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula,
                                        int hudson_formula_or_negative, const fmh_pair_diversity_sites* diversity_or_null,
                                        const fmh_hudson_sites* sites_or_null, void* stream);
int fmh_pair_region_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula,
                                  int hudson_formula_or_negative, const fmh_pair_diversity_sites* diversity_or_null,
                                  const fmh_hudson_sites* sites_or_null, fmh_hudson_totals* h_global_totals, void* stream);

/*
 * Packing of the totals structs into the f64 + u64 vectors a sum-reduce moves (fmh_allreduce_totals, or MPI / any other
 * transport).  Everything packed is a plain sum over slabs, except haplotype_capacity / sites_attempted-style constants,
 * which unpack restores from the rank count carried in the last u64 slot.
 */
#define FMH_HUDSON_PACK_F64 10
#define FMH_HUDSON_PACK_U64 10
int fmh_hudson_totals_pack(const fmh_hudson_totals* t, double* h_f64 /*[FMH_HUDSON_PACK_F64]*/,
                           uint64_t* h_u64 /*[FMH_HUDSON_PACK_U64]*/);
int fmh_hudson_totals_unpack(fmh_hudson_totals* t, const double* h_f64, const uint64_t* h_u64);
/* per-population summaries (fmh_population_summaries, fmh_diversity_sites): n entries -> n f64 (pi_sum) and 3n + 1 u64
 * (segregating, uncallable, haplotype_capacity per population, then the rank count) */
#define FMH_POP_PACK_F64(n) ((size_t)(n))
#define FMH_POP_PACK_U64(n) (3 * (size_t)(n) + 1)
int fmh_pop_totals_pack(const fmh_pop_totals* t, int n, double* h_f64, uint64_t* h_u64);
int fmh_pop_totals_unpack(fmh_pop_totals* t, int n, const double* h_f64, const uint64_t* h_u64);
/* W&C regional sums (calculate_overall_fst_wc, stats.rs:2145-2374): slots = 1 + n_groups (n_groups - 1) / 2 ->
 * 2 * slots f64 (sum_a, sum_b) and slots + 1 u64 (informative_sites per slot, sites_attempted) */
#define FMH_WC_PACK_F64(slots) (2 * (size_t)(slots))
#define FMH_WC_PACK_U64(slots) ((size_t)(slots) + 1)
int fmh_wc_totals_pack(const fmh_wc_totals* t, int n_groups, double* h_f64, uint64_t* h_u64);
int fmh_wc_totals_unpack(fmh_wc_totals* t, int n_groups, const double* h_f64, const uint64_t* h_u64);

/* ---- measurement ---------------------------------------------------------------------------------- */
/* Accumulated HIP-event time (ms) and launch count of the dominant sweep kernel since the last
 * reset, measured on the stream the kernel ran on.  Timing is off unless enabled: the two event
 * records of a timed launch cost a few microseconds of stream time.  fmh_timing_enable(n) with
 * n > 1 times every n-th sweep only (the launch count returned is that of the timed ones). */
int fmh_timing_enable(int on);
int fmh_timing_reset(void);
int fmh_timing_read(double* h_total_ms, uint64_t* h_launches);
/* shortest and longest of the launches timed since the last reset (0, 0 when none) */
int fmh_timing_read_minmax(double* h_min_ms, double* h_max_ms);
/* The same for the grouped all-reduce of the timed sharded sweeps (RCCL transport): HIP events on the communicator's stream around
 * ncclGroupStart .. ncclGroupEnd - the device-side latency of one 2 x 512-byte reduce. */
int fmh_timing_read_reduce(double* h_total_ms, uint64_t* h_reduces);
int fmh_timing_reset_reduce(void);

#ifdef __cplusplus
}
#endif
#endif /* FERROMIC_HIP_H */
