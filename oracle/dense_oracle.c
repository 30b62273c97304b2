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

/* ================================================================================================
 * Weir & Cockerham per site on a dense matrix (stats.rs:1814-2032, 2034-2127, 1781-1812) and the regional
 * sums of calculate_overall_fst_wc (2145-2374), for the full-size C3 parity gate: the Python
 * restatement (oracle/ferromic_ref.py) does ~100 sites per second, this does millions.
 * Pinned to the Python half bit for bit by tests/test_oracle_dense_c.py.
 *
 * Dense model: column h of a row is one (sample, side) entry, called unless its missing bit is set;
 * group_of_column[h] = group index or 0xFF (what SubpopulationMembership.left / right hold, 1104-1150).
 * ================================================================================================ */

/* calculate_variance_components, stats.rs:2034-2127; stats = (n_i, p_i) of the r groups with data */
static void wc_variance_components(const size_t* n, const double* p, int r_i, double global_freq, double* a_out, double* b_out) {
  const double r = (double)r_i;
  *a_out = 0.0;
  *b_out = 0.0;
  if (r < 2.0) return;
  size_t total_haplotypes = 0;
  for (int i = 0; i < r_i; ++i) total_haplotypes += n[i];
  const double n_bar = (double)total_haplotypes / r;
  if ((n_bar - 1.0) < 1e-9) return;
  const double global_p = global_freq;
  double sum_sq_diff_n = 0.0;
  for (int i = 0; i < r_i; ++i) {
    const double diff = (double)n[i] - n_bar;
    sum_sq_diff_n += diff * diff;
  }
  const double c_squared = (r > 0.0 && n_bar > 0.0) ? sum_sq_diff_n / (r * n_bar * n_bar) : 0.0;
  double numerator_s_squared = 0.0;
  for (int i = 0; i < r_i; ++i) {
    const double diff_p = p[i] - global_p;
    numerator_s_squared += (double)n[i] * diff_p * diff_p;
  }
  const double s_squared = ((r - 1.0) > 1e-9 && n_bar > 1e-9) ? numerator_s_squared / ((r - 1.0) * n_bar) : 0.0;
  const double x_wc = global_p * (1.0 - global_p) - ((r - 1.0) / r) * s_squared;
  const double a_numerator_term = s_squared - (x_wc / (n_bar - 1.0));
  const double a_denominator_factor = 1.0 - (c_squared / (r - 1.0));
  *a_out = a_numerator_term / a_denominator_factor;
  *b_out = (n_bar / (n_bar - 1.0)) * x_wc;
}

/* fst_estimate_from_components, stats.rs:1781-1812 -> 0 calculable, 1 indeterminate, 2 no variance */
static uint8_t wc_state(double a, double b) {
  const double denominator = a + b;
  if (denominator > FST_EPSILON) return 0;
  if (denominator < -FST_EPSILON) return 1;
  if (fabs(a) > FST_EPSILON) return 0;
  return 2;
}

#define FO_WC_MAX_GROUPS 16
typedef struct {
  const uint8_t* data;
  const uint64_t* missing;
  size_t stride, s0, s1, variants;
  const uint8_t* group_of_column;
  int G;
  double *a, *b; /* [slots][variants] */
  uint8_t* state;
} wc_job;

static void* wc_worker(void* pv) {
  wc_job* j = (wc_job*)pv;
  const int G = j->G;
  const int slots = 1 + G * (G - 1) / 2;
  for (size_t s = j->s0; s < j->s1; ++s) {
    const size_t base = s * j->stride;
    /* alleles present among ALL called entries (1826-1837), ascending (BTreeSet) */
    uint8_t present[256];
    memset(present, 0, sizeof present);
    for (size_t h = 0; h < j->stride; ++h)
      if (!j->missing || !dense_missing(j->missing, base + h)) present[j->data[base + h]] = 1;
    double sum_a = 0.0, sum_b = 0.0, pa[FO_WC_MAX_GROUPS * FO_WC_MAX_GROUPS], pb[FO_WC_MAX_GROUPS * FO_WC_MAX_GROUPS];
    uint8_t pseen[FO_WC_MAX_GROUPS * FO_WC_MAX_GROUPS];
    memset(pseen, 0, sizeof pseen);
    for (int k = 0; k < slots; ++k) { pa[k] = 0.0; pb[k] = 0.0; }
    int populated = 0;
    for (int target = 0; target < 256; ++target) {
      if (!present[target]) continue;
      size_t total_counts[FO_WC_MAX_GROUPS], alt_counts[FO_WC_MAX_GROUPS];
      for (int g = 0; g < G; ++g) { total_counts[g] = 0; alt_counts[g] = 0; }
      for (size_t h = 0; h < j->stride; ++h) {
        if (j->missing && dense_missing(j->missing, base + h)) continue;
        const uint8_t g = j->group_of_column[h];
        if (g == 0xFF) continue;
        total_counts[g] += 1;
        if (j->data[base + h] == (uint8_t)target) alt_counts[g] += 1;
      }
      size_t total_called = 0, total_target = 0, ns[FO_WC_MAX_GROUPS];
      double ps[FO_WC_MAX_GROUPS];
      int valid = 0;
      for (int g = 0; g < G; ++g) {
        if (total_counts[g] == 0) continue;
        ns[valid] = total_counts[g];
        ps[valid] = (double)alt_counts[g] / (double)total_counts[g];
        ++valid;
        total_called += total_counts[g];
        total_target += alt_counts[g];
      }
      populated = 1; /* pop_sizes_populated (1987): an allele was iterated */
      if (valid < 2) continue;
      const double global_freq = total_called > 0 ? (double)total_target / (double)total_called : 0.0;
      double ca, cb;
      wc_variance_components(ns, ps, valid, global_freq, &ca, &cb);
      sum_a += ca;
      sum_b += cb;
      int k = 1;
      for (int x = 0; x < G; ++x)
        for (int y = x + 1; y < G; ++y, ++k) {
          const size_t ta = total_counts[x], tb = total_counts[y];
          if (ta == 0 || tb == 0) continue;
          const size_t pn[2] = {ta, tb};
          const double pp[2] = {(double)alt_counts[x] / (double)ta, (double)alt_counts[y] / (double)tb};
          const size_t pair_total = ta + tb;
          const double pair_global = pair_total > 0 ? (double)(alt_counts[x] + alt_counts[y]) / (double)pair_total : 0.0;
          double qa, qb;
          wc_variance_components(pn, pp, 2, pair_global, &qa, &qb);
          pa[k] += qa;
          pb[k] += qb;
          pseen[k] = 1;
        }
    }
    if (!populated) { /* InsufficientData everywhere (1987-2003) */
      for (int k = 0; k < slots; ++k) { j->a[(size_t)k * j->variants + s] = 0.0; j->b[(size_t)k * j->variants + s] = 0.0; j->state[(size_t)k * j->variants + s] = 3; }
      continue;
    }
    j->a[s] = sum_a;
    j->b[s] = sum_b;
    j->state[s] = wc_state(sum_a, sum_b);
    for (int k = 1; k < slots; ++k) {
      const size_t o = (size_t)k * j->variants + s;
      if (pseen[k]) { j->a[o] = pa[k]; j->b[o] = pb[k]; j->state[o] = wc_state(pa[k], pb[k]); }
      else { j->a[o] = 0.0; j->b[o] = 0.0; j->state[o] = 3; }
    }
  }
  return NULL;
}

/* Per-site a, b, state for every slot (0 = overall, then pairs (0,1),(0,2),...) and the regional sums over the sites whose
 * state is not insufficient, accumulated SERIALLY in site order as the reference does (2222-2229). */
void fo_wc_sites_threaded(const uint8_t* data, const uint64_t* missing, size_t variants, size_t stride, const uint8_t* group_of_column,
                          int G, double* a, double* b, uint8_t* state, double* sum_a, double* sum_b, uint64_t* informative, int nthreads) {
  if (G < 2 || G > FO_WC_MAX_GROUPS) return;
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > variants) nthreads = variants ? (int)variants : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  wc_job* jobs = (wc_job*)malloc(sizeof(wc_job) * (size_t)nthreads);
  for (int t = 0; t < nthreads; ++t) {
    wc_job j = {data, missing, stride, variants * (size_t)t / (size_t)nthreads, variants * (size_t)(t + 1) / (size_t)nthreads, variants, group_of_column, G, a, b, state};
    jobs[t] = j;
    pthread_create(&th[t], NULL, wc_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  const int slots = 1 + G * (G - 1) / 2;
  for (int k = 0; k < slots; ++k) {
    double sa = 0.0, sb = 0.0;
    uint64_t n = 0;
    for (size_t s = 0; s < variants; ++s) {
      const size_t o = (size_t)k * variants + s;
      if (state[o] != 3) { sa += a[o]; sb += b[o]; ++n; }
    }
    sum_a[k] = sa;
    sum_b[k] = sb;
    informative[k] = n;
  }
}
