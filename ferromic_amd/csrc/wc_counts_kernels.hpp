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
    // r groups with data, visited in group order (overall: all groups; pair: gi, gj)
    sh.s2_den = 0.0; sh.rm1_over_r = 0.0; sh.nbar_m1 = 0.0; sh.a_den = 1.0; sh.b_fac = 0.0; sh.live = 0; sh.s2_ok = 0;
    int r_i = 0;
    unsigned long long total_h = 0;
    auto visit = [&](auto&& fn) {
      if (gi >= 0) { fn(called[(size_t)gi * rows + site]); fn(called[(size_t)gj * rows + site]); }
      else for (int g = 0; g < G; ++g) { const uint32_t v = called[(size_t)g * rows + site]; if (v != 0) fn(v); }
    };
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

__global__ __launch_bounds__(256) void wc_from_counts_kernel(int G, int n_alleles, size_t rows, const uint32_t* __restrict__ called,
                                                             const uint32_t* __restrict__ alt, const uint32_t* __restrict__ acounts,
                                                             const uint32_t* __restrict__ n_all, double* __restrict__ out_a,
                                                             double* __restrict__ out_b, uint8_t* __restrict__ out_state) {
  const size_t site = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= rows) return;
  const size_t nslots = 1 + (size_t)G * (G - 1) / 2;
  auto count_of = [&](int a, int g) -> uint32_t {
    if (acounts) return acounts[((size_t)a * G + g) * rows + site];
    const uint32_t c1 = alt[(size_t)g * rows + site];
    return a == 1 ? c1 : called[(size_t)g * rows + site] - c1;
  };
  if (n_all[site] == 0) {  // no allele among all samples: InsufficientData everywhere (stats.rs:1987-2003)
    for (size_t k = 0; k < nslots; ++k) { out_a[k * rows + site] = 0.0; out_b[k * rows + site] = 0.0; out_state[k * rows + site] = 3; }
    return;
  }
  WcManyShape ms;
  // ---- overall (stats.rs:1893-1946) ----
  {
    double wa = 0.0, wb = 0.0;
    int valid = 0;
    unsigned long long total_called = 0;
    for (int g = 0; g < G; ++g) { const uint32_t v = called[(size_t)g * rows + site]; if (v != 0) { ++valid; total_called += v; } }
    if (valid >= 2) {
      ms.from(called, rows, site, -1, -1, G);
      if (ms.sh.live) {
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
    }
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
      double wa = 0.0, wb = 0.0;
      ms.from(called, rows, site, i, j, G);
      if (ms.sh.live) {
        const unsigned long long pair_total = (unsigned long long)ni + nj;
        const double ndi = (double)ni, ndj = (double)nj;
        for (int a = 0; a < n_alleles; ++a) {
          const uint32_t ci = count_of(a, i), cj = count_of(a, j);
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

}  // namespace fmh
