/*
 * dense_oracle.c — CPU oracle, C half: literal restatement of the reference's DENSE paths
 * (src/stats.rs of SauersML/ferromic) for matrices too large for the Python restatement, plus a
 * pthread driver used as bench.py's `cpu_baseline` ("port": the reference itself is Rust and no Rust
 * toolchain exists on the build or GPU box).
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load liboracle_dense.so.  The product (libferromic_hip.so / ferromic_amd) never links,
 * imports or executes it.
 *
 * Pinning: tests/test_oracle_dense_c.py checks every function here against oracle/ferromic_ref.py,
 * which is itself pinned by the reference's own known-answer vectors (tests/golden/).
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fPIC -shared -pthread (oracle/Makefile).
 * -ffp-contract=off keeps a*b+c un-fused, as rustc does, so per-site f64 values are bit-identical
 * to the Rust expressions restated below.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FST_EPSILON 1e-12 /* stats.rs:26 */

/* ---- counter-based synthetic cohort: identical stream to generate_kernel (sweep_kernels.hpp) ---- */
static inline uint32_t hash24(uint64_t seed, uint64_t site, uint64_t column) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (site * 0x100000001B3ull + column + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 40);
}

typedef struct {
  uint8_t* data;
  uint64_t* missing; /* may be NULL; must be zeroed by the caller */
  size_t s0, s1, variants, columns;
  uint64_t seed, first_site;
  const uint32_t* thr;
  const uint8_t* pop_of_column;
  uint32_t missing_thr;
} gen_job;

static void* gen_worker(void* p) {
  gen_job* j = (gen_job*)p;
  for (size_t s = j->s0; s < j->s1; ++s) {
    for (size_t h = 0; h < j->columns; ++h) {
      const uint32_t thr = j->thr[(size_t)j->pop_of_column[h] * j->variants + s];
      uint8_t bit = hash24(j->seed, j->first_site + s, h) < thr ? 1 : 0;
      if (j->missing && hash24(j->seed ^ 0xA5A5A5A5DEADBEEFull, j->first_site + s, h) < j->missing_thr) {
        const size_t idx = s * j->columns + h;
        __atomic_fetch_or(&j->missing[idx >> 6], 1ull << (idx & 63), __ATOMIC_RELAXED);
        bit = 0;
      }
      j->data[s * j->columns + h] = bit;
    }
  }
  return NULL;
}

/* data: [variants*columns] in the reference host layout (stats.rs:293); missing: zeroed words or NULL */
void fo_generate(uint8_t* data, uint64_t* missing, size_t variants, size_t columns, uint64_t seed,
                 uint64_t first_site, const uint32_t* thresholds24, const uint8_t* pop_of_column,
                 uint32_t missing_thr, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  gen_job* jobs = (gen_job*)malloc(sizeof(gen_job) * nthreads);
  for (int t = 0; t < nthreads; ++t) {
    gen_job j = {data, missing, variants * t / nthreads, variants * (t + 1) / nthreads, variants, columns,
                 seed, first_site, thresholds24, pop_of_column, missing_thr};
    jobs[t] = j;
    pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

/* ---- stats.rs:1298-1302 ---- */
static inline int dense_missing(const uint64_t* bits, size_t idx) { return (int)((bits[idx >> 6] >> (idx & 63)) & 1); }

/* ---- stats.rs:1665-1674 ---- */
static inline size_t dense_sum_alt_no_missing(const uint8_t* data, size_t base, const size_t* offsets, size_t n_off) {
  size_t sum = 0;
  const uint8_t* ptr = data + base;
  for (size_t i = 0; i < n_off; ++i) sum += ptr[offsets[i]];
  return sum;
}

/* ---- stats.rs:1677-1697 ---- */
static inline void dense_sum_alt_with_missing(const uint8_t* data, size_t base, const size_t* offsets, size_t n_off,
                                              const uint64_t* bits, size_t* total_out, size_t* alt_out) {
  size_t alt = 0, total = 0;
  const uint8_t* ptr = data + base;
  for (size_t i = 0; i < n_off; ++i) {
    const size_t idx = base + offsets[i];
    if (dense_missing(bits, idx)) continue;
    alt += ptr[offsets[i]];
    total += 1;
  }
  *total_out = total;
  *alt_out = alt;
}

/* ---- stats.rs:1700-1709; returns 0 and leaves *out untouched for None ---- */
static inline int dense_pi_from_counts(size_t total_called, size_t alt_count, double* out) {
  if (total_called < 2) return 0;
  double n = (double)total_called;
  double alt = (double)alt_count;
  double ref_count = (double)(total_called - alt_count);
  double sum_sq = ref_count * ref_count + alt * alt;
  *out = n / (n - 1.0) * (1.0 - sum_sq / (n * n));
  return 1;
}

/* ---- stats.rs:1712-1733 ---- */
static inline int dense_dxy_from_biallelic_counts(size_t n1, size_t alt1, size_t n2, size_t alt2, double* out) {
  if (n1 == 0 || n2 == 0) return 0;
  double n1_f = (double)n1, n2_f = (double)n2;
  double alt1_f = (double)alt1 / n1_f;
  double alt2_f = (double)alt2 / n2_f;
  double ref1 = 1.0 - alt1_f;
  double ref2 = 1.0 - alt2_f;
  double dot = ref1 * ref2 + alt1_f * alt2_f;
  if (dot < 0.0) dot = 0.0;
  double dxy = 1.0 - dot;
  if (dxy < 0.0) dxy = 0.0; else if (dxy > 1.0) dxy = 1.0;
  *out = dxy;
  return 1;
}

typedef struct { /* DensePopulationSummary scalars, stats.rs:1311-1317 */
  uint64_t haplotype_capacity, segregating_sites, uncallable_sites;
  double pi_sum;
} fo_pop_totals;

typedef struct { /* HudsonSummaryTotals (1545-1552) + hudson_component_sums (1625) */
  double numerator_sum, denominator_sum, pi1_sum, pi2_sum, dxy_sum_all;
  uint64_t dxy_uncallable_sites;
  double site_num_sum, site_den_sum;
  uint64_t sites_with_components;
} fo_hudson_totals;

/*
 * build_dense_population_summary over sites [s0, s1), stats.rs:1367-1470 (one fold of the rayon arm;
 * the serial arm is the same loop over the whole range).
 */
void fo_population_summary_range(const uint8_t* data, const uint64_t* missing, size_t stride, size_t s0, size_t s1,
                                 const size_t* offsets, size_t n_off, uint32_t* alt_counts, uint32_t* called_counts,
                                 fo_pop_totals* t) {
  size_t seg = 0, unc = 0;
  double pi_total = 0.0;
  for (size_t v = s0; v < s1; ++v) {
    const size_t base = v * stride;
    size_t called, alt;
    if (missing) dense_sum_alt_with_missing(data, base, offsets, n_off, missing, &called, &alt);
    else { alt = dense_sum_alt_no_missing(data, base, offsets, n_off); called = n_off; }
    alt_counts[v] = (uint32_t)alt;
    called_counts[v] = (uint32_t)called;
    if (called >= 2 && alt > 0 && alt < called) seg += 1;
    double value;
    if (dense_pi_from_counts(called, alt, &value)) pi_total += value; else unc += 1;
  }
  t->haplotype_capacity = n_off;
  t->segregating_sites = seg;
  t->uncallable_sites = unc; /* what calculate_pi_from_summary recounts at 1512-1516 */
  t->pi_sum = pi_total;
}

/* aggregate_hudson_components_from_summaries over [s0, s1), stats.rs:1554-1623 */
void fo_hudson_from_summaries_range(const uint32_t* alt1, const uint32_t* called1, const uint32_t* alt2,
                                    const uint32_t* called2, size_t s0, size_t s1, fo_hudson_totals* t) {
  for (size_t idx = s0; idx < s1; ++idx) {
    const size_t n1 = called1[idx], n2 = called2[idx];
    if (n1 == 0 || n2 == 0) { t->dxy_uncallable_sites += 1; continue; }
    const size_t alt_count1 = alt1[idx], alt_count2 = alt2[idx];
    const size_t ref_count1 = n1 - alt_count1, ref_count2 = n2 - alt_count2;
    const double denom_pairs = (double)(n1 * n2);
    if (denom_pairs == 0.0) continue;
    double dxy = (double)(alt_count1 * ref_count2 + ref_count1 * alt_count2) / denom_pairs;
    if (dxy < 0.0) dxy = 0.0; else if (dxy > 1.0) dxy = 1.0;
    t->dxy_sum_all += dxy;
    if (n1 < 2 || n2 < 2) continue;
    const double denom1 = (double)(n1 * (n1 - 1));
    const double denom2 = (double)(n2 * (n2 - 1));
    const double pi1 = denom1 > 0.0 ? 2.0 * (double)alt_count1 * (double)ref_count1 / denom1 : 0.0;
    const double pi2 = denom2 > 0.0 ? 2.0 * (double)alt_count2 * (double)ref_count2 / denom2 : 0.0;
    t->pi1_sum += pi1;
    t->pi2_sum += pi2;
    if (dxy > FST_EPSILON) {
      t->numerator_sum += dxy - 0.5 * (pi1 + pi2);
      t->denominator_sum += dxy;
    }
  }
}

/*
 * dense_hudson_sites_biallelic over [s0, s1), stats.rs:3179-3278 (+ dense_fst_components_from_biallelic
 * 1736-1757).  Outputs are NaN for None.  Counts come from the summaries (same gather).
 */
void fo_dense_hudson_sites_biallelic_range(const uint32_t* alt1, const uint32_t* called1, const uint32_t* alt2,
                                           const uint32_t* called2, int has_missing, size_t s0, size_t s1,
                                           double* fst, double* dxy_out, double* pi1_out, double* pi2_out,
                                           double* num, double* den, fo_hudson_totals* t) {
  for (size_t v = s0; v < s1; ++v) {
    const size_t n1 = called1[v], n2 = called2[v], a1 = alt1[v], a2 = alt2[v];
    double pi1 = NAN, pi2 = NAN, dxy = NAN;
    int ok1, ok2, okd;
    if (has_missing) {
      ok1 = dense_pi_from_counts(n1, a1, &pi1);
      ok2 = dense_pi_from_counts(n2, a2, &pi2);
    } else { /* stats.rs:3219-3256 */
      ok1 = n1 >= 2;
      ok2 = n2 >= 2;
      if (ok1) {
        if (a1 == 0 || a1 == n1) pi1 = 0.0;
        else {
          double n1_f = (double)n1, scale = n1_f / (n1_f - 1.0), inv_n_sq = 1.0 / (n1_f * n1_f);
          double alt_f = (double)a1, ref_f = (double)(n1 - a1);
          pi1 = scale * (1.0 - (ref_f * ref_f + alt_f * alt_f) * inv_n_sq);
        }
      }
      if (ok2) {
        if (a2 == 0 || a2 == n2) pi2 = 0.0;
        else {
          double n2_f = (double)n2, scale = n2_f / (n2_f - 1.0), inv_n_sq = 1.0 / (n2_f * n2_f);
          double alt_f = (double)a2, ref_f = (double)(n2 - a2);
          pi2 = scale * (1.0 - (ref_f * ref_f + alt_f * alt_f) * inv_n_sq);
        }
      }
    }
    okd = dense_dxy_from_biallelic_counts(n1, a1, n2, a2, &dxy);
    double f = NAN, nc = NAN, dc = NAN;
    if (okd && ok1 && ok2) {
      if (dxy > FST_EPSILON) {
        double numv = dxy - 0.5 * (pi1 + pi2);
        f = numv / dxy; nc = numv; dc = dxy;
      } else {
        double pi_avg = 0.5 * (pi1 + pi2);
        if (fabs(pi_avg) <= FST_EPSILON) { nc = 0.0; dc = 0.0; }
      }
    }
    if (fst) fst[v] = f;
    if (dxy_out) dxy_out[v] = okd ? dxy : NAN;
    if (pi1_out) pi1_out[v] = ok1 ? pi1 : NAN;
    if (pi2_out) pi2_out[v] = ok2 ? pi2 : NAN;
    if (num) num[v] = nc;
    if (den) den[v] = dc;
    if (!isnan(nc) && !isnan(dc)) { t->site_num_sum += nc; t->site_den_sum += dc; t->sites_with_components += 1; }
  }
}

/* ---- the timed CPU baseline: "pi + Hudson FST" sweep, site ranges over pthreads ---------------- */
typedef struct {
  const uint8_t* data;
  const uint64_t* missing;
  size_t stride, s0, s1;
  const size_t *off1, *off2;
  size_t n1, n2;
  uint32_t *alt1, *called1, *alt2, *called2;
  double *fst, *dxy, *pi1, *pi2, *num, *den;
  fo_pop_totals p1, p2;
  fo_hudson_totals h;
} sweep_job;

static void* sweep_worker(void* p) {
  sweep_job* j = (sweep_job*)p;
  memset(&j->h, 0, sizeof j->h);
  /* one pass per population, as the reference does (one summary per Population object) */
  fo_population_summary_range(j->data, j->missing, j->stride, j->s0, j->s1, j->off1, j->n1, j->alt1, j->called1, &j->p1);
  fo_population_summary_range(j->data, j->missing, j->stride, j->s0, j->s1, j->off2, j->n2, j->alt2, j->called2, &j->p2);
  fo_hudson_from_summaries_range(j->alt1, j->called1, j->alt2, j->called2, j->s0, j->s1, &j->h);
  fo_dense_hudson_sites_biallelic_range(j->alt1, j->called1, j->alt2, j->called2, j->missing != NULL, j->s0, j->s1,
                                        j->fst, j->dxy, j->pi1, j->pi2, j->num, j->den, &j->h);
  return NULL;
}

/*
 * Restatement of the reference's dense Rayon algorithm for one population pair: summaries
 * (1367-1470) x2, Hudson totals from summaries (1554-1623) and the per-site records (3179-3278),
 * parallelised over site ranges like rayon's fold/reduce (partials combined in range order).
 * Per-site arrays have `variants` entries (f64 tracks may be NULL).
 */
void fo_hudson_sweep_threaded(const uint8_t* data, const uint64_t* missing, size_t variants, size_t stride,
                              const size_t* off1, size_t n1, const size_t* off2, size_t n2, uint32_t* alt1,
                              uint32_t* called1, uint32_t* alt2, uint32_t* called2, double* fst, double* dxy,
                              double* pi1, double* pi2, double* num, double* den, fo_pop_totals* pop_totals /*[2]*/,
                              fo_hudson_totals* totals, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  sweep_job* jobs = (sweep_job*)calloc(nthreads, sizeof(sweep_job));
  for (int t = 0; t < nthreads; ++t) {
    sweep_job* j = &jobs[t];
    j->data = data; j->missing = missing; j->stride = stride;
    j->s0 = variants * t / nthreads; j->s1 = variants * (t + 1) / nthreads;
    j->off1 = off1; j->off2 = off2; j->n1 = n1; j->n2 = n2;
    j->alt1 = alt1; j->called1 = called1; j->alt2 = alt2; j->called2 = called2;
    j->fst = fst; j->dxy = dxy; j->pi1 = pi1; j->pi2 = pi2; j->num = num; j->den = den;
    pthread_create(&th[t], NULL, sweep_worker, j);
  }
  memset(totals, 0, sizeof *totals);
  memset(pop_totals, 0, 2 * sizeof *pop_totals);
  pop_totals[0].haplotype_capacity = n1;
  pop_totals[1].haplotype_capacity = n2;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(th[t], NULL);
    sweep_job* j = &jobs[t];
    pop_totals[0].segregating_sites += j->p1.segregating_sites;
    pop_totals[0].uncallable_sites += j->p1.uncallable_sites;
    pop_totals[0].pi_sum += j->p1.pi_sum;
    pop_totals[1].segregating_sites += j->p2.segregating_sites;
    pop_totals[1].uncallable_sites += j->p2.uncallable_sites;
    pop_totals[1].pi_sum += j->p2.pi_sum;
    totals->numerator_sum += j->h.numerator_sum;
    totals->denominator_sum += j->h.denominator_sum;
    totals->pi1_sum += j->h.pi1_sum;
    totals->pi2_sum += j->h.pi2_sum;
    totals->dxy_sum_all += j->h.dxy_sum_all;
    totals->dxy_uncallable_sites += j->h.dxy_uncallable_sites;
    totals->site_num_sum += j->h.site_num_sum;
    totals->site_den_sum += j->h.site_den_sum;
    totals->sites_with_components += j->h.sites_with_components;
  }
  free(th);
  free(jobs);
}
