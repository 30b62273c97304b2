// wc_counts_kernels.hpp — Weir & Cockerham from per-group count tables (any number of groups) and the slot reductions.
// Include after sweep_kernels.hpp (WcShape, wc_shape, wc_apply).
#pragma once

namespace fmh {

// ------------------------------------------------------------------------------------------------
// W&C for any number of groups (calculate_fst_wc_at_site_with_membership, stats.rs:1814-2032), from count tables.
// The fused sweep keeps P <= 8 groups in registers; beyond that the counting is done by summary sweeps over
// batches of 8 groups (same kernels, same HBM traffic per batch) and this kernel does the per-site arithmetic:
// one site per thread, groups and pairs in loops, the same wc_shape / wc_apply operand order as the fused path.
//   called[g][site], alt[g][site] (biallelic: c1 = alt, c0 = called - alt) or acounts[a][g][site] (general),
//   n_all[site] = called entries over ALL columns (pop_sizes_populated, stats.rs:1987).
// Outputs [(1 + G(G-1)/2)][rows]: a, b, state; slot 0 = overall, pairs in (0,1),(0,2),... order.
// ------------------------------------------------------------------------------------------------
struct WcManyShape {
  WcShape sh;
  __device__ void from(const uint32_t* __restrict__ called, size_t rows, size_t site, int gi, int gj, int G) {
    if (gi >= 0) from_visit([&](auto&& fn) { fn(called[(size_t)gi * rows + site]); fn(called[(size_t)gj * rows + site]); });
    else from_visit([&](auto&& fn) { for (int g = 0; g < G; ++g) { const uint32_t v = called[(size_t)g * rows + site]; if (v != 0) fn(v); } });
  }
  __device__ void from_pair(uint32_t ni, uint32_t nj) { from_visit([&](auto&& fn) { fn(ni); fn(nj); }); }
  template <class Visit>
  __device__ void from_visit(Visit&& visit) {
    // r groups with data, visited in group order (overall: all groups; pair: gi, gj)
    sh.s2_den = 0.0; sh.rm1_over_r = 0.0; sh.nbar_m1 = 0.0; sh.a_den = 1.0; sh.b_fac = 0.0; sh.live = 0; sh.s2_ok = 0;
    int r_i = 0;
    unsigned long long total_h = 0;
    visit([&](uint32_t v) { ++r_i; total_h += v; });
    const double r = (double)r_i;
    if (r < 2.0) return;
    const double n_bar = (double)total_h / r;
    if ((n_bar - 1.0) < 1e-9) return;
    double sum_sq_diff_n = 0.0;
    visit([&](uint32_t v) { double diff = (double)v - n_bar; sum_sq_diff_n += diff * diff; });
    const double c_squared = (r > 0.0 && n_bar > 0.0) ? sum_sq_diff_n / (r * n_bar * n_bar) : 0.0;
    sh.s2_ok = ((r - 1.0) > 1e-9 && n_bar > 1e-9) ? 1 : 0;
    sh.s2_den = (r - 1.0) * n_bar;
    sh.rm1_over_r = (r - 1.0) / r;
    sh.nbar_m1 = n_bar - 1.0;
    sh.a_den = 1.0 - (c_squared / (r - 1.0));
    sh.b_fac = n_bar / (n_bar - 1.0);
    sh.live = 1;
  }
};

// the overall components of one site (stats.rs:1893-1946); count_of(a, g) = calls of allele a in group g
template <class CountOf>
__device__ __forceinline__ void wc_overall_site(int G, int n_alleles, size_t rows, size_t site, const uint32_t* __restrict__ called, CountOf&& count_of,
                                                double& wa, double& wb) {
  wa = 0.0; wb = 0.0;
  int valid = 0;
  unsigned long long total_called = 0;
  for (int g = 0; g < G; ++g) { const uint32_t v = called[(size_t)g * rows + site]; if (v != 0) { ++valid; total_called += v; } }
  if (valid < 2) return;
  WcManyShape ms;
  ms.from(called, rows, site, -1, -1, G);
  if (!ms.sh.live) return;
  for (int a = 0; a < n_alleles; ++a) {
    unsigned long long total_target = 0;
    for (int g = 0; g < G; ++g) if (called[(size_t)g * rows + site] != 0) total_target += count_of(a, g);
    const double global_freq = total_called > 0 ? (double)total_target / (double)total_called : 0.0;
    double num = 0.0;
    for (int g = 0; g < G; ++g) {
      const uint32_t v = called[(size_t)g * rows + site];
      if (v == 0) continue;
      const double nd = (double)v;
      const double diff_p = (double)count_of(a, g) / nd - global_freq;
      num += nd * diff_p * diff_p;
    }
    double ca, cb;
    wc_apply<false>(ms.sh, num, global_freq, ca, cb, nullptr);
    wa += ca;
    wb += cb;
  }
}
// one pair of groups with ni, nj > 0 called haplotypes at one site (stats.rs:1948-1985); ci(a), cj(a) = calls of allele a in either group
template <class Ci, class Cj>
__device__ __forceinline__ void wc_pair_site(uint32_t ni, uint32_t nj, int n_alleles, Ci&& ci_of, Cj&& cj_of, double& wa, double& wb) {
  wa = 0.0; wb = 0.0;
  WcManyShape ms;
  ms.from_pair(ni, nj);
  if (!ms.sh.live) return;
  const unsigned long long pair_total = (unsigned long long)ni + nj;
  const double ndi = (double)ni, ndj = (double)nj;
  for (int a = 0; a < n_alleles; ++a) {
    const uint32_t ci = ci_of(a), cj = cj_of(a);
    const double pair_global = pair_total > 0 ? (double)((unsigned long long)ci + cj) / (double)pair_total : 0.0;
    double num = 0.0;
    { const double diff_p = (double)ci / ndi - pair_global; num += ndi * diff_p * diff_p; }
    { const double diff_p = (double)cj / ndj - pair_global; num += ndj * diff_p * diff_p; }
    double pa, pb;
    wc_apply<false>(ms.sh, num, pair_global, pa, pb, nullptr);
    wa += pa;
    wb += pb;
  }
}

// ---- a matrix without missing calls: every group's called count is its size at every site ------------------------------------
// Then the shape of a slot and every denominator are LAUNCH constants, exactly as in the fused kernels (sweep_kernels.hpp, "divisions that
// share a denominator"): wc_many_pre_kernel computes them once per slot - with the very functions the per-site path would call at every
// site, so the values are those - and a division becomes div_shared's three operations.  A true f64 division is ~35 instructions and the
// per-site path makes 17 of them per pair and site: 26 groups x 2 M sites spent 10 ms in them.
struct WcSlotPre {
  double s2_den, rm1_over_r, nbar_m1, a_den, b_fac;  // WcShape
  double rcp[3];                                     // refined reciprocals of s2_den, nbar_m1, a_den (wc_apply<true>)
  double total, rcp_total;                           // called haplotypes of the slot's groups and the reciprocal
  int live, s2_ok, has_data, pad;                    // has_data: overall = at least two groups with members; pair = both have members
};
__device__ __forceinline__ void wc_pair_of_slot(size_t pair, int G, int& gi, int& gj) {  // pairs in (0,1), (0,2), ... order
  gi = 0;
  size_t rem = pair;
  while (rem >= (size_t)(G - 1 - gi)) { rem -= (size_t)(G - 1 - gi); ++gi; }
  gj = gi + 1 + (int)rem;
}
__global__ __launch_bounds__(256) void wc_many_pre_kernel(int G, const uint32_t* __restrict__ gsize, WcSlotPre* __restrict__ pre, double* __restrict__ grcp) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nslots = 1 + (size_t)G * (G - 1) / 2;
  if (k < (size_t)G) grcp[k] = gsize[k] ? refined_rcp((double)gsize[k]) : 0.0;
  if (k >= nslots) return;
  WcManyShape ms;
  WcSlotPre o;
  unsigned long long total = 0;
  if (k == 0) {
    int valid = 0;
    for (int g = 0; g < G; ++g) if (gsize[g]) { ++valid; total += gsize[g]; }
    ms.from_visit([&](auto&& fn) { for (int g = 0; g < G; ++g) if (gsize[g]) fn(gsize[g]); });
    o.has_data = valid >= 2;
  } else {
    int gi, gj;
    wc_pair_of_slot(k - 1, G, gi, gj);
    const uint32_t ni = gsize[gi], nj = gsize[gj];
    total = (unsigned long long)ni + nj;
    ms.from_pair(ni, nj);
    o.has_data = ni != 0 && nj != 0;
  }
  o.s2_den = ms.sh.s2_den; o.rm1_over_r = ms.sh.rm1_over_r; o.nbar_m1 = ms.sh.nbar_m1; o.a_den = ms.sh.a_den; o.b_fac = ms.sh.b_fac;
  o.live = ms.sh.live; o.s2_ok = ms.sh.s2_ok; o.pad = 0;
  o.rcp[0] = refined_rcp(ms.sh.s2_den); o.rcp[1] = refined_rcp(ms.sh.nbar_m1); o.rcp[2] = refined_rcp(ms.sh.a_den);
  o.total = (double)total;
  o.rcp_total = total ? refined_rcp((double)total) : 0.0;
  pre[k] = o;
}
__device__ __forceinline__ WcShape wc_shape_of_pre(const WcSlotPre& o) {
  WcShape sh;
  sh.s2_den = o.s2_den; sh.rm1_over_r = o.rm1_over_r; sh.nbar_m1 = o.nbar_m1; sh.a_den = o.a_den; sh.b_fac = o.b_fac; sh.live = o.live; sh.s2_ok = o.s2_ok;
  return sh;
}
// wc_overall_site / wc_pair_site with the launch constants: the same operands in the same order, div_shared for every division
template <class CountOf>
__device__ __forceinline__ void wc_overall_site_pre(int G, int n_alleles, const uint32_t* __restrict__ gsize, const double* __restrict__ grcp,
                                                    const WcSlotPre& o, CountOf&& count_of, double& wa, double& wb) {
  wa = 0.0; wb = 0.0;
  if (!o.has_data || !o.live) return;
  const WcShape sh = wc_shape_of_pre(o);
  for (int a = 0; a < n_alleles; ++a) {
    unsigned long long total_target = 0;
    for (int g = 0; g < G; ++g) if (gsize[g] != 0) total_target += count_of(a, g);
    const double global_freq = o.total > 0.0 ? div_shared((double)total_target, o.total, o.rcp_total) : 0.0;
    double num = 0.0;
    for (int g = 0; g < G; ++g) {
      if (gsize[g] == 0) continue;
      const double nd = (double)gsize[g];
      const double diff_p = div_shared((double)count_of(a, g), nd, grcp[g]) - global_freq;
      num += nd * diff_p * diff_p;
    }
    double ca, cb;
    wc_apply<true>(sh, num, global_freq, ca, cb, o.rcp);
    wa += ca;
    wb += cb;
  }
}
template <class Ci, class Cj>
__device__ __forceinline__ void wc_pair_site_pre(double ndi, double ndj, double rcpi, double rcpj, const WcSlotPre& o, int n_alleles, Ci&& ci_of,
                                                 Cj&& cj_of, double& wa, double& wb) {
  wa = 0.0; wb = 0.0;
  if (!o.live) return;
  const WcShape sh = wc_shape_of_pre(o);
  for (int a = 0; a < n_alleles; ++a) {
    const uint32_t ci = ci_of(a), cj = cj_of(a);
    const double pair_global = o.total > 0.0 ? div_shared((double)((unsigned long long)ci + cj), o.total, o.rcp_total) : 0.0;
    double num = 0.0;
    { const double diff_p = div_shared((double)ci, ndi, rcpi) - pair_global; num += ndi * diff_p * diff_p; }
    { const double diff_p = div_shared((double)cj, ndj, rcpj) - pair_global; num += ndj * diff_p * diff_p; }
    double pa, pb;
    wc_apply<true>(sh, num, pair_global, pa, pb, o.rcp);
    wa += pa;
    wb += pb;
  }
}

__global__ __launch_bounds__(256) void wc_from_counts_kernel(int G, int n_alleles, size_t rows, const uint32_t* __restrict__ called,
                                                             const uint32_t* __restrict__ alt, const uint32_t* __restrict__ acounts,
                                                             const uint32_t* __restrict__ n_all, double* __restrict__ out_a,
                                                             double* __restrict__ out_b, uint8_t* __restrict__ out_state,
                                                             const WcSlotPre* __restrict__ pre, const double* __restrict__ grcp,
                                                             const uint32_t* __restrict__ gsize) {
  const size_t site = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= rows) return;
  const size_t nslots = 1 + (size_t)G * (G - 1) / 2;
  if (pre) {  // nothing missing: called counts are the group sizes, every slot's shape and denominators are launch constants (wave-uniform loads)
    auto count_of = [&](int a, int g) -> uint32_t {
      if (acounts) return acounts[((size_t)a * G + g) * rows + site];
      const uint32_t c1 = alt[(size_t)g * rows + site];
      return a == 1 ? c1 : gsize[g] - c1;
    };
    {
      double wa, wb;
      wc_overall_site_pre(G, n_alleles, gsize, grcp, pre[0], count_of, wa, wb);
      out_a[site] = wa;
      out_b[site] = wb;
      out_state[site] = wc_classify(wa, wb);
    }
    size_t k = 1;
    for (int i = 0; i < G; ++i) {
      const double ndi = (double)gsize[i], rcpi = grcp[i];
      for (int j = i + 1; j < G; ++j, ++k) {
        if (!pre[k].has_data) { out_a[k * rows + site] = 0.0; out_b[k * rows + site] = 0.0; out_state[k * rows + site] = 3; continue; }
        double wa, wb;
        wc_pair_site_pre(ndi, (double)gsize[j], rcpi, grcp[j], pre[k], n_alleles, [&](int a) { return count_of(a, i); }, [&](int a) { return count_of(a, j); }, wa, wb);
        out_a[k * rows + site] = wa;
        out_b[k * rows + site] = wb;
        out_state[k * rows + site] = wc_classify(wa, wb);
      }
    }
    return;
  }
  auto count_of = [&](int a, int g) -> uint32_t {
    if (acounts) return acounts[((size_t)a * G + g) * rows + site];
    const uint32_t c1 = alt[(size_t)g * rows + site];
    return a == 1 ? c1 : called[(size_t)g * rows + site] - c1;
  };
  if (n_all[site] == 0) {  // no allele among all samples: InsufficientData everywhere (stats.rs:1987-2003)
    for (size_t k = 0; k < nslots; ++k) { out_a[k * rows + site] = 0.0; out_b[k * rows + site] = 0.0; out_state[k * rows + site] = 3; }
    return;
  }
  // ---- overall (stats.rs:1893-1946) ----
  {
    double wa, wb;
    wc_overall_site(G, n_alleles, rows, site, called, count_of, wa, wb);
    out_a[site] = wa;
    out_b[site] = wb;
    out_state[site] = wc_classify(wa, wb);
  }
  // ---- pairs (stats.rs:1948-1985) ----
  size_t k = 1;
  for (int i = 0; i < G; ++i) {
    const uint32_t ni = called[(size_t)i * rows + site];
    for (int j = i + 1; j < G; ++j, ++k) {
      const uint32_t nj = called[(size_t)j * rows + site];
      if (ni == 0 || nj == 0) { out_a[k * rows + site] = 0.0; out_b[k * rows + site] = 0.0; out_state[k * rows + site] = 3; continue; }
      double wa, wb;
      wc_pair_site(ni, nj, n_alleles, [&](int a) { return count_of(a, i); }, [&](int a) { return count_of(a, j); }, wa, wb);
      out_a[k * rows + site] = wa;
      out_b[k * rows + site] = wb;
      out_state[k * rows + site] = wc_classify(wa, wb);
    }
  }
}

// Regional sums per slot (calculate_overall_fst_wc, stats.rs:2172-2229).  Grid (slot, chunk): a workgroup sums one chunk
// of the sites of one slot (thread t takes sites t, t+256, ... of the chunk in ascending order, fixed LDS tree) into
// partial[slot][chunk]; wc_slot_finalize_kernel adds the chunks in ascending order.  Deterministic for a given launch.
__global__ __launch_bounds__(256) void wc_slot_reduce_kernel(size_t rows, const double* __restrict__ a, const double* __restrict__ b,
                                                             const uint8_t* __restrict__ state, double* __restrict__ part_a,
                                                             double* __restrict__ part_b, unsigned long long* __restrict__ part_inf) {
  const size_t k = blockIdx.x, chunks = gridDim.y, c = blockIdx.y;
  const size_t per = (rows + chunks - 1) / chunks;
  const size_t s0 = c * per, s1 = s0 + per < rows ? s0 + per : rows;
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  for (size_t s = s0 + threadIdx.x; s < s1; s += 256)
    if (state[k * rows + s] != 3) { va += a[k * rows + s]; vb += b[k * rows + s]; ++vi; }
  __shared__ double la[256], lb[256];
  __shared__ unsigned long long li[256];
  la[threadIdx.x] = va; lb[threadIdx.x] = vb; li[threadIdx.x] = vi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { la[threadIdx.x] += la[threadIdx.x + w]; lb[threadIdx.x] += lb[threadIdx.x + w]; li[threadIdx.x] += li[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_a[k * chunks + c] = la[0]; part_b[k * chunks + c] = lb[0]; part_inf[k * chunks + c] = li[0]; }
}

__global__ void wc_slot_finalize_kernel(size_t nslots, size_t chunks, const double* __restrict__ part_a, const double* __restrict__ part_b,
                                        const unsigned long long* __restrict__ part_inf, double* __restrict__ sum_a,
                                        double* __restrict__ sum_b, unsigned long long* __restrict__ informative) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nslots) return;
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  for (size_t c = 0; c < chunks; ++c) { va += part_a[k * chunks + c]; vb += part_b[k * chunks + c]; vi += part_inf[k * chunks + c]; }
  sum_a[k] = va; sum_b[k] = vb; informative[k] = vi;
}

// ------------------------------------------------------------------------------------------------
// Regional sums ONLY, from the count tables (fmh_wc_sweep_many without per-site tracks: run_vcf's CSV populations ask for nothing else).
// The per-site route above writes and re-reads (1 + G(G-1)/2) x 17 B per site - 26 populations: 5.5 KB per site, 11 GB of scratch per
// 2 M sites.  Here nothing per site and slot leaves the chip:
//  * pairs: one THREAD per pair walks the sites of a chunk in ascending order with its sums in registers; the workgroup's 256 pairs share
//    the count tables of sixteen sites at a time through LDS ([called | alt or the alleles' counts][G][16]);
//  * overall: one thread per site (G-long loops), summed like wc_slot_reduce_kernel.
// Both write partial[slot][chunk]; wc_slot_finalize_kernel adds the chunks in ascending order.  The per-site arithmetic is wc_pair_site /
// wc_overall_site, i.e. the bits of the per-site route; only the order of the regional additions differs (as it does between any two routes).
// ------------------------------------------------------------------------------------------------
constexpr int kWcTotTile = 16;
constexpr int kWcTotCellsMax = 64;  // cells of a tile per thread (256 threads) at most: tables of up to 64 KB
template <int CPT>  // cells of a tile per thread: (1 + NAr) x G x 16 <= 256 CPT
__global__ __launch_bounds__(256) void wc_pair_totals_kernel(int G, int n_alleles, size_t rows, size_t chunk_rows, size_t chunks,
                                                             const uint32_t* __restrict__ called, const uint32_t* __restrict__ alt,
                                                             const uint32_t* __restrict__ acounts, const uint32_t* __restrict__ n_all,
                                                             double* __restrict__ part_a, double* __restrict__ part_b,
                                                             unsigned long long* __restrict__ part_inf, const WcSlotPre* __restrict__ pre,
                                                             const double* __restrict__ grcp, const uint32_t* __restrict__ gsize) {
  extern __shared__ uint32_t wc_tot_lds[];  // [1 + NAr][G][16] counts, then [16] n_all
  const int NAr = acounts ? n_alleles : 1;
  const size_t npairs = (size_t)G * (G - 1) / 2;
  const size_t pair = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = pair < npairs;
  int gi = 0, gj = 1;
  if (active) wc_pair_of_slot(pair, G, gi, gj);
  // nothing missing (pre != null): this pair's launch constants in registers; the `called` plane of the tile is then the group sizes
  WcSlotPre mine{};
  double ndi = 0.0, ndj = 0.0, rcpi = 0.0, rcpj = 0.0;
  uint32_t size_i = 0, size_j = 0;
  if (pre && active) { mine = pre[1 + pair]; size_i = gsize[gi]; size_j = gsize[gj]; ndi = (double)size_i; ndj = (double)size_j; rcpi = grcp[gi]; rcpj = grcp[gj]; }
  const size_t c = blockIdx.y;
  const size_t s0 = c * chunk_rows, s1 = s0 + chunk_rows < rows ? s0 + chunk_rows : rows;
  uint32_t* tile_nall = wc_tot_lds + (size_t)(1 + NAr) * G * kWcTotTile;
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  // the tile after the current one travels in registers while the current one is worked on (a tile is sixteen sites of every table:
  // (1 + NAr) x G x 16 cells over 256 threads, CPT per thread)
  const uint32_t cells = (uint32_t)(1 + NAr) * (uint32_t)G * kWcTotTile;
  auto fetch = [&](size_t t0, uint32_t (&v)[CPT]) {
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const uint32_t idx = threadIdx.x + 256u * q;
      v[q] = 0;
      if (idx >= cells) continue;
      const uint32_t t = idx % kWcTotTile, g = (idx / kWcTotTile) % (uint32_t)G, arr = idx / (kWcTotTile * (uint32_t)G);
      const size_t site = t0 + t;
      if (site >= s1) continue;
      if (arr == 0) v[q] = pre ? 0u : called[(size_t)g * rows + site];  // (not read when the sizes are launch constants)
      else if (acounts) v[q] = acounts[((size_t)(arr - 1) * G + g) * rows + site];
      else v[q] = alt[(size_t)g * rows + site];
    }
  };
  constexpr bool kAhead = CPT <= 4;  // larger tables are staged directly (their cells' addresses alone would take the register file)
  uint32_t next[CPT];
  if (kAhead && s0 < s1) fetch(s0, next);
  for (size_t t0 = s0; t0 < s1; t0 += kWcTotTile) {
    __syncthreads();
    if constexpr (kAhead) {
#pragma unroll
      for (int q = 0; q < CPT; ++q) { const uint32_t idx = threadIdx.x + 256u * q; if (idx < cells) wc_tot_lds[idx] = next[q]; }
    } else {
      for (uint32_t idx = threadIdx.x; idx < cells; idx += 256) {
        const uint32_t t = idx % kWcTotTile, g = (idx / kWcTotTile) % (uint32_t)G, arr = idx / (kWcTotTile * (uint32_t)G);
        const size_t site = t0 + t;
        uint32_t v = 0;
        if (site < s1) {
          if (arr == 0) v = pre ? 0u : called[(size_t)g * rows + site];
          else if (acounts) v = acounts[((size_t)(arr - 1) * G + g) * rows + site];
          else v = alt[(size_t)g * rows + site];
        }
        wc_tot_lds[idx] = v;
      }
    }
    if (threadIdx.x < kWcTotTile) tile_nall[threadIdx.x] = t0 + threadIdx.x < s1 ? (pre ? 1u : n_all[t0 + threadIdx.x]) : 0;
    __syncthreads();
    if (kAhead && t0 + kWcTotTile < s1) fetch(t0 + kWcTotTile, next);
    if (!active) continue;
    if (pre) {
      if (!mine.has_data) continue;
      for (int t = 0; t < kWcTotTile; ++t) {
        if (tile_nall[t] == 0) continue;  // past the chunk
        auto count_of = [&](int a, int g, uint32_t n) -> uint32_t {
          if (acounts) return wc_tot_lds[((size_t)(1 + a) * G + g) * kWcTotTile + t];
          const uint32_t c1 = wc_tot_lds[((size_t)G + g) * kWcTotTile + t];
          return a == 1 ? c1 : n - c1;
        };
        double wa, wb;
        wc_pair_site_pre(ndi, ndj, rcpi, rcpj, mine, n_alleles, [&](int a) { return count_of(a, gi, size_i); }, [&](int a) { return count_of(a, gj, size_j); }, wa, wb);
        va += wa; vb += wb; ++vi;
      }
      continue;
    }
    for (int t = 0; t < kWcTotTile; ++t) {
      if (tile_nall[t] == 0) continue;  // no allele among all samples (or past the chunk): InsufficientData, not summed
      const uint32_t ni = wc_tot_lds[(size_t)gi * kWcTotTile + t], nj = wc_tot_lds[(size_t)gj * kWcTotTile + t];
      if (ni == 0 || nj == 0) continue;
      auto count_of = [&](int a, int g, uint32_t n) -> uint32_t {
        if (acounts) return wc_tot_lds[((size_t)(1 + a) * G + g) * kWcTotTile + t];
        const uint32_t c1 = wc_tot_lds[((size_t)G + g) * kWcTotTile + t];
        return a == 1 ? c1 : n - c1;
      };
      double wa, wb;
      wc_pair_site(ni, nj, n_alleles, [&](int a) { return count_of(a, gi, ni); }, [&](int a) { return count_of(a, gj, nj); }, wa, wb);
      va += wa; vb += wb; ++vi;  // every state but InsufficientData is summed (stats.rs:2172-2203)
    }
  }
  if (active) { part_a[(1 + pair) * chunks + c] = va; part_b[(1 + pair) * chunks + c] = vb; part_inf[(1 + pair) * chunks + c] = vi; }
}

// The pair sums of a BIALLELIC matrix with NOTHING MISSING (the 1000-Genomes-populations case: 26 groups, 325 pairs): the kernel above in
// its one shape that matters, without its run-time arms.  What a pair needs of a group at a site - the allele's frequency in the group and
// its count - does not depend on the other group, so the staging threads compute it once per (site, group) instead of the pair threads
// once per (site, pair): the tile in LDS is [site][group] of {p1 = c1 / n, p0 = (n - c1) / n, c1, n - c1} as doubles (32 bytes, two
// ds_read_b128; groups of one site lie side by side, so the 64 pairs of a wave read different banks).  Per pair, site and allele that
// leaves div_shared for the pair frequency, two differences, the five operations of the numerator and wc_apply: 28 f64 operations where
// the general kernel issues 40 and the conversions - the same operands in the same order, so the same bits per (site, pair).  The R
// replicas of a pair (threads pair + npairs * r, r < R) take the sites r, r + R, ... of every tile: 325 pairs fill 5.08 waves, 4 x 325 fill
// 20.3 of 21.  Partials land in part[(1 + pair)][chunk * R + r]; wc_slot_finalize_wave_kernel adds them in that order.
constexpr int kWcBiTile = 32;       // sites per tile: 128-byte runs of every group's count row
constexpr int kWcBiGroupsMax = 64;  // 64 KiB of LDS; beyond that the general kernel
struct alignas(16) WcBiCell { double p1, p0, c1, c0; };
typedef double wc_f64x2 __attribute__((ext_vector_type(2)));
// Both alleles of one (site, pair) in lockstep: the two chains are independent until their terms are added (allele 0 first, the order of
// the allele loop), and written side by side they stay side by side in the ISA - a dependent v_*_f64 waits ~8 cycles behind the one before
// it, which two or three waves per SIMD do not cover on their own.  Every expression is wc_pair_site_pre's / wc_apply<true>'s (a live pair
// always has s2_ok; the select keeps the flag's meaning without a branch).
// N independent (site, allele) chains of one pair in lockstep: every step is written for all chains before the next step, and stays so in
// the ISA - a dependent v_*_f64 issues ~12 cycles after the one it waits for (tools/microbench/f64_issue_rates.hip: 11.7 cycles alone on a
// SIMD, 4.4 with four chains on four waves), which the three or four resident waves per SIMD do not cover on their own.  Every expression is
// wc_pair_site_pre's / wc_apply<true>'s (a live pair always has s2_ok; the select keeps the flag's meaning without a branch).
// c = c_i + c_j (integers below 2^33: exact), pi / pj = the allele's frequency in either group; out: the allele's a term and x_wc.
template <int N>
__device__ __forceinline__ void wc_bi_chains(const double (&c)[N], const double (&pi)[N], const double (&pj)[N], double ndi, double ndj,
                                             const WcSlotPre& o, const WcShape& sh, double (&a)[N], double (&x)[N]) {
  double g[N], q[N], e[N], di[N], dj[N], num[N], s[N], t[N];
#define FMH_EACH for (int k = 0; k < N; ++k)
#define FMH_DIV(out, n, d, r) /* div_shared, step by step over the chains */ \
  _Pragma("unroll") FMH_EACH q[k] = (n) * (r);                               \
  _Pragma("unroll") FMH_EACH e[k] = __builtin_fma(-(d), q[k], (n));          \
  _Pragma("unroll") FMH_EACH out[k] = __builtin_fma(e[k], (r), q[k]);
  FMH_DIV(g, c[k], o.total, o.rcp_total)
#pragma unroll
  FMH_EACH { di[k] = pi[k] - g[k]; dj[k] = pj[k] - g[k]; }
#pragma unroll
  FMH_EACH { di[k] = ndi * di[k] * di[k]; dj[k] = ndj * dj[k] * dj[k]; }
#pragma unroll
  FMH_EACH num[k] = di[k] + dj[k];  // (0.0 + x) + y of the allele loop: x is never -0.0
  FMH_DIV(s, num[k], sh.s2_den, o.rcp[0])
#pragma unroll
  FMH_EACH s[k] = sh.s2_ok ? s[k] : 0.0;
#pragma unroll
  FMH_EACH x[k] = g[k] * (1.0 - g[k]) - sh.rm1_over_r * s[k];
  FMH_DIV(t, x[k], sh.nbar_m1, o.rcp[1])
#pragma unroll
  FMH_EACH t[k] = s[k] - t[k];
  FMH_DIV(a, t[k], sh.a_den, o.rcp[2])
#undef FMH_DIV
#undef FMH_EACH
}
// SITES sites of one pair: chains (site 0 allele 0, site 0 allele 1, site 1 allele 0, ...); the sums take the sites in order and, per site,
// allele 0 then allele 1 - the allele loop's wa = 0.0; wa += a0; wa += a1; va += wa without the 0.0: that matters only when a0 and a1 are
// both -0.0, and then va + (-0.0) and va + 0.0 are the same bits (va starts at +0.0 and never becomes -0.0)
struct WcBiPairCells { wc_f64x2 fi, ni, fj, nj; };  // {p1, p0} and {c1, c0} of the pair's two groups at one site
__device__ __forceinline__ WcBiPairCells wc_bi_load(const WcBiCell* __restrict__ tile, int G, int gi, int gj, int t) {
  const wc_f64x2* ci = reinterpret_cast<const wc_f64x2*>(tile + (size_t)t * G + gi);
  const wc_f64x2* cj = reinterpret_cast<const wc_f64x2*>(tile + (size_t)t * G + gj);
  WcBiPairCells v;
  v.fi = ci[0]; v.ni = ci[1]; v.fj = cj[0]; v.nj = cj[1];
  return v;
}
// SITES sites of one pair: chains (site 0 allele 0, site 0 allele 1, site 1 allele 0, ...); the sums take the sites in order and, per site,
// allele 0 then allele 1 - the allele loop's wa = 0.0; wa += a0; wa += a1; va += wa without the 0.0: that matters only when a0 and a1 are
// both -0.0, and then va + (-0.0) and va + 0.0 are the same bits (va starts at +0.0 and never becomes -0.0)
template <int SITES>
__device__ __forceinline__ void wc_bi_sites(const WcBiPairCells (&v)[SITES], double ndi, double ndj, const WcSlotPre& o, const WcShape& sh,
                                            double& va, double& vb) {
  double c[2 * SITES], pi[2 * SITES], pj[2 * SITES], a[2 * SITES], x[2 * SITES];
#pragma unroll
  for (int u = 0; u < SITES; ++u) {
    c[2 * u] = v[u].ni.y + v[u].nj.y; c[2 * u + 1] = v[u].ni.x + v[u].nj.x;
    pi[2 * u] = v[u].fi.y; pi[2 * u + 1] = v[u].fi.x;
    pj[2 * u] = v[u].fj.y; pj[2 * u + 1] = v[u].fj.x;
  }
  wc_bi_chains<2 * SITES>(c, pi, pj, ndi, ndj, o, sh, a, x);
#pragma unroll
  for (int u = 0; u < SITES; ++u) {
    va += a[2 * u] + a[2 * u + 1];
    vb += sh.b_fac * x[2 * u] + sh.b_fac * x[2 * u + 1];
  }
}
template <int CELLS>  // cells of a tile a thread stages: G x 32 <= CELLS x blockDim.x
__global__ __launch_bounds__(512, 5) void wc_pair_totals_biallelic_kernel(int G, int R, int blocks_per_chunk, size_t rows, size_t chunk_rows, size_t chunks,
                                                                       const uint32_t* __restrict__ alt, double* __restrict__ part_a,
                                                                       double* __restrict__ part_b, unsigned long long* __restrict__ part_inf,
                                                                       const WcSlotPre* __restrict__ pre, const double* __restrict__ grcp,
                                                                       const uint32_t* __restrict__ gsize) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wc_bi_lds[];  // WcBiCell[kWcBiTile][G]
  WcBiCell* tile = reinterpret_cast<WcBiCell*>(wc_bi_lds);
  const size_t npairs = (size_t)G * (G - 1) / 2;
  const uint32_t B = blockDim.x;  // a multiple of 64 (the host sizes it to the items of a block)
  // The workgroups of one chunk read the same rows of the count tables: they go to ONE XCD, next to each other in its dispatch order, so the
  // second and later ones find the rows in that XCD's L2 (workgroup ids go round the eight XCDs; 40 groups with seven workgroups per chunk
  // re-read 2.2 GB from HBM before this mapping).  Grid: 8 x ceil(chunks / 8) x blocks_per_chunk workgroups, the padding ones leave at once.
  const uint32_t xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u;
  const size_t c = (size_t)(slot / (uint32_t)blocks_per_chunk) * 8u + xcd;
  if (c >= chunks) return;
  const size_t item = (size_t)(slot % (uint32_t)blocks_per_chunk) * B + threadIdx.x;
  const bool active = item < npairs * (size_t)R;
  const size_t pair = active ? item % npairs : 0;
  const int rep = active ? (int)(item / npairs) : 0;
  int gi = 0, gj = 1;
  wc_pair_of_slot(pair, G, gi, gj);
  const WcSlotPre mine = pre[1 + pair];
  const WcShape sh = wc_shape_of_pre(mine);
  const double ndi = (double)gsize[gi], ndj = (double)gsize[gj];
  const bool work = active && mine.has_data && mine.live;
  const size_t s0 = c * chunk_rows, s1 = s0 + chunk_rows < rows ? s0 + chunk_rows : rows;
  // the staging side: cell q of this thread is site threadIdx.x % 32 of group threadIdx.x / 32 + q B / 32
  constexpr int kCellsMax = CELLS;
  const uint32_t st_t = threadIdx.x % kWcBiTile, st_g0 = threadIdx.x / kWcBiTile, st_gstep = B / kWcBiTile;
  uint32_t next[kCellsMax], st_n[kCellsMax];
  double st_r[kCellsMax];
#pragma unroll
  for (int q = 0; q < kCellsMax; ++q) {  // the groups of this thread's cells: sizes and reciprocals once
    const uint32_t g = st_g0 + st_gstep * q;
    st_n[q] = g < (uint32_t)G ? gsize[g] : 0;
    st_r[q] = g < (uint32_t)G ? grcp[g] : 0.0;
  }
  auto fetch = [&](size_t t0) {
    const bool in = t0 + st_t < s1;
    const uint32_t* src = alt + (size_t)st_g0 * rows + t0 + st_t;
#pragma unroll
    for (int q = 0; q < kCellsMax; ++q) {
      next[q] = 0;
      if (in && st_g0 + st_gstep * q < (uint32_t)G) next[q] = src[(size_t)(st_gstep * q) * rows];
    }
  };
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < kCellsMax; ++q) {
      const uint32_t g = st_g0 + st_gstep * q;
      if (g >= (uint32_t)G) continue;
      const uint32_t n = st_n[q], c1 = next[q];
      const double nd = (double)n, r = st_r[q];
      WcBiCell cell;  // (a group without members: its cells are never read - its pairs have no data)
      cell.c1 = (double)c1;
      cell.c0 = (double)(n - c1);
      cell.p1 = div_shared(cell.c1, nd, r);
      cell.p0 = div_shared(cell.c0, nd, r);
      tile[(size_t)st_t * G + g] = cell;
    }
  };
  if (s0 < s1) fetch(s0);
  for (size_t t0 = s0; t0 < s1; t0 += kWcBiTile) {
    __syncthreads();  // the previous tile has been read
    stage();
    __syncthreads();
    if (t0 + kWcBiTile < s1) fetch(t0 + kWcBiTile);  // the next tile's counts travel while this one is worked on
    const int tmax = s1 - t0 < (size_t)kWcBiTile ? (int)(s1 - t0) : kWcBiTile;
    if (active && mine.has_data && rep < tmax) vi += (unsigned long long)((tmax - rep + R - 1) / R);
    if (!work) continue;
    // One site = two chains per step, five waves per SIMD.  Measured and not kept (profiles/r04/wc_many_groups_pair_kernel_variants.md): two
    // sites in lockstep on four waves per SIMD (3-6 % behind), a second tile image with one barrier per tile (level), the next site's cells
    // read from LDS a site ahead (level to 8 % behind).
#pragma unroll 2
    for (int t = rep; t < tmax; t += R) {
      const WcBiPairCells cells[1] = {wc_bi_load(tile, G, gi, gj, t)};
      wc_bi_sites<1>(cells, ndi, ndj, mine, sh, va, vb);
    }
  }
  if (active) {
    const size_t k = (1 + pair) * (chunks * (size_t)R) + c * (size_t)R + (size_t)rep;
    part_a[k] = va; part_b[k] = vb; part_inf[k] = vi;
  }
}

__global__ __launch_bounds__(256) void wc_overall_totals_kernel(int G, int n_alleles, size_t rows, size_t chunk_rows, size_t chunks,
                                                                const uint32_t* __restrict__ called, const uint32_t* __restrict__ alt,
                                                                const uint32_t* __restrict__ acounts, const uint32_t* __restrict__ n_all,
                                                                double* __restrict__ part_a, double* __restrict__ part_b,
                                                                unsigned long long* __restrict__ part_inf, const WcSlotPre* __restrict__ pre,
                                                                const double* __restrict__ grcp, const uint32_t* __restrict__ gsize) {
  const size_t c = blockIdx.x;
  const size_t s0 = c * chunk_rows, s1 = s0 + chunk_rows < rows ? s0 + chunk_rows : rows;
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  for (size_t site = s0 + threadIdx.x; site < s1; site += 256) {
    if (pre) {
      auto count_of = [&](int a, int g) -> uint32_t {
        if (acounts) return acounts[((size_t)a * G + g) * rows + site];
        const uint32_t c1 = alt[(size_t)g * rows + site];
        return a == 1 ? c1 : gsize[g] - c1;
      };
      double wa, wb;
      wc_overall_site_pre(G, n_alleles, gsize, grcp, pre[0], count_of, wa, wb);
      va += wa; vb += wb; ++vi;
      continue;
    }
    if (n_all[site] == 0) continue;
    auto count_of = [&](int a, int g) -> uint32_t {
      if (acounts) return acounts[((size_t)a * G + g) * rows + site];
      const uint32_t c1 = alt[(size_t)g * rows + site];
      return a == 1 ? c1 : called[(size_t)g * rows + site] - c1;
    };
    double wa, wb;
    wc_overall_site(G, n_alleles, rows, site, called, count_of, wa, wb);
    va += wa; vb += wb; ++vi;
  }
  __shared__ double la[256], lb[256];
  __shared__ unsigned long long li[256];
  la[threadIdx.x] = va; lb[threadIdx.x] = vb; li[threadIdx.x] = vi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { la[threadIdx.x] += la[threadIdx.x + w]; lb[threadIdx.x] += lb[threadIdx.x + w]; li[threadIdx.x] += li[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_a[c] = la[0]; part_b[c] = lb[0]; part_inf[c] = li[0]; }
}

// ------------------------------------------------------------------------------------------------
// The Hudson pair of two populations from their per-site count tables (aggregate_hudson_components_from_summaries, stats.rs:1554-1623,
// and the per-site records of dense_hudson_sites_biallelic): what the reference computes from two DensePopulationSummary objects, which
// may come from DIFFERENT matrices.  Same per-site code as the sweep after its counting (finish_biallelic_site, site_epilogue), one site
// per thread, the block / grid reduction of the sweep.
// ------------------------------------------------------------------------------------------------
template <bool MISSING>
__global__ __launch_bounds__(kBlock) void hudson_from_counts_kernel(const SweepArgs A, const uint32_t* __restrict__ called1, const uint32_t* __restrict__ alt1,
                                                                    const uint32_t* __restrict__ called2, const uint32_t* __restrict__ alt2) {
  constexpr int MODE = kModeSummary | kModeHudson;
  LaneTotals<2, MODE> T;
  T.clear();
  const size_t stride = (size_t)gridDim.x * kBlock;
  const size_t rounds = (A.row_count + stride - 1) / stride;
  for (size_t r = 0; r < rounds; ++r) {
    const size_t site = r * stride + (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool row_ok = site < A.row_count;
    SiteTally<2> mine;
    mine.n[0] = row_ok ? called1[site] : 0; mine.alt[0] = row_ok ? alt1[site] : 0;
    mine.n[1] = row_ok ? called2[site] : 0; mine.alt[1] = row_ok ? alt2[site] : 0;
    mine.n_all = mine.n[0] + mine.n[1];
    double hud_dot = 0.0;
    finish_biallelic_site<2, MODE>(mine, hud_dot);
    WcSite<2> wc;
    site_epilogue<2, MODE, MISSING, false>(A, site, row_ok, mine, hud_dot, wc, T);
  }
  reduce_block_totals<2, MODE>(A, T);
}

// the chunks of one slot added by one wave: lane l takes chunks l, l + 64, ... in ascending order, then a fixed xor tree over the lanes
// (the totals route has thousands of chunks per slot; wc_slot_finalize_kernel's one thread per slot would walk them one by one)
__global__ __launch_bounds__(64) void wc_slot_finalize_wave_kernel(size_t chunks, const double* __restrict__ part_a, const double* __restrict__ part_b,
                                                                   const unsigned long long* __restrict__ part_inf, double* __restrict__ sum_a,
                                                                   double* __restrict__ sum_b, unsigned long long* __restrict__ informative) {
  const size_t k = blockIdx.x;
  double va = 0.0, vb = 0.0;
  unsigned long long vi = 0;
  for (size_t c = threadIdx.x; c < chunks; c += 64) { va += part_a[k * chunks + c]; vb += part_b[k * chunks + c]; vi += part_inf[k * chunks + c]; }
  va = wave_sum(va); vb = wave_sum(vb); vi = wave_sum(vi);
  if (threadIdx.x == 0) { sum_a[k] = va; sum_b[k] = vb; informative[k] = vi; }
}

}  // namespace fmh
