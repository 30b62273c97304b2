// pairwise.hip — fmh_pairwise_differences: the sample-pair Gram contraction on the matrix cores (kernels in
// pairwise_kernels.hpp).  Its own translation unit so that it builds in parallel with the sweep routes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "abi_internal.hpp"
#include "pairwise_kernels.hpp"

using namespace fmh;
using namespace fmhi;

extern "C" int fmh_pairwise_differences(const fmh_matrix* m, size_t n_samples, unsigned long long* d_diff,
                                        unsigned long long* d_both, void* stream) {
  if (!m || !d_diff || !d_both) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_samples > m->samples) return fail(FMH_ERR_INVALID, "n_samples %zu exceeds the matrix's %zu samples", n_samples, m->samples);
  if (m->ploidy > 127) return fail(FMH_ERR_UNSUPPORTED, "pairwise differences support ploidy <= 127 (int8 operands), got %zu", m->ploidy);
  FMH_TRY(use_device(m->device));
  if (n_samples < 2 || m->variants == 0) return FMH_OK;
  hipStream_t st = (hipStream_t)stream;
  const int n_alleles = (int)m->max_allele + 1;
  const bool missing = m->has_missing;
  // biallelic and nothing missing: one plane (allele 1) and an all-ones row after the last sample (pairwise_kernels.hpp)
  const bool env_two_planes = options().pd_two_planes.load() != 0;  // measurements / tests: the general route
  const bool single = !missing && n_alleles == 2 && !env_two_planes;
  const int n_planes = single ? 1 : (missing ? n_alleles + 2 : n_alleles);  // + genotype length, + valid flag only when calls can be missing
  const size_t tile_edge = kPdBig;
  const size_t n_pad = round_up(n_samples + (single ? 1 : 0), kPdBig);
  // counts 0..4 are exact in FP4 (e2m1): twice the MFMA rate of int8 at half the plane bytes (pairwise_kernels.hpp)
  const bool env_int8 = options().pd_int8.load() != 0;  // measurements / tests: the int8 route
  const bool fp4 = m->ploidy <= 4 && !env_int8;
  // round 4: the phase-interleaved Gram kernel (pd_gram256p_kernel) and its planes layout (K halves contiguous); FMH_PD_PHASED=0 keeps round 3's
  const bool phased = options().pd_phased.load() != 0;
  const int layout = phased ? 1 : 0;
  const size_t spb = fp4 ? 2 : 1;            // sites per byte of a plane row
  const size_t ksites = kPdStageK * spb;     // sites per K block = per Gram stage
  // sites are processed in slabs so that the sample-major planes stay within a fixed budget of HBM
  const size_t budget = (size_t)options().pd_planes_bytes.load();
  size_t slab = std::max<size_t>(budget / ((size_t)n_planes * n_pad), kPdStageK) / kPdStageK * ksites;
  slab = std::min(round_up(m->variants, ksites), slab);
  // a matrix with a packed image feeds the planes kernel its bit rows (1/8 of the bytes); FMH_LAYOUT=bytes keeps the u8 route
  // (a three-plane image, alleles 4..7, is unpacked slab by slab into scratch bytes and takes the u8 planes kernel)
  const bool packed_only = m->p0 && !(m->data && layout_bytes_forced());
  const bool from_packed = packed_only && !m->p2;
  const bool via_unpack = packed_only && m->p2;
  Workspace* w = nullptr;
  FMH_TRY(workspace(m->device, &w));
  std::lock_guard<std::mutex> busy(w->in_use);
  const size_t planes_bytes = (size_t)n_planes * n_pad * slab / spb;
  if (w->pd_planes_bytes < planes_bytes) {
    if (w->pd_planes) (void)hipFree(w->pd_planes);
    w->pd_planes = nullptr;
    w->pd_planes_bytes = 0;
    HIP_TRY(hipMalloc((void**)&w->pd_planes, planes_bytes));
    w->pd_planes_bytes = planes_bytes;
  }
  uint8_t* planes = w->pd_planes;
  hipError_t e = hipSuccess;
  DeviceScratch scratch;
  scratch.device = m->device;
  scratch.stream = st;
  unsigned long long *d_gram = nullptr, *d_totals = nullptr;
  if (single) {
    const size_t gram_bytes = n_samples * n_samples * sizeof(unsigned long long), totals_bytes = n_samples * sizeof(unsigned long long);
    FMH_TRY(scratch.get(&d_gram, n_samples * n_samples));
    FMH_TRY(scratch.get(&d_totals, n_samples));
    HIP_TRY(hipMemsetAsync(d_gram, 0, gram_bytes, st));
    HIP_TRY(hipMemsetAsync(d_totals, 0, totals_bytes, st));
  }
  const size_t nt = n_pad / tile_edge, tiles = nt * (nt + 1) / 2;
  const size_t env_chunk = (size_t)options().pd_kchunk.load();
  uint8_t* unpacked = nullptr;
  if (via_unpack) {
    HIP_TRY(hipMalloc((void**)&unpacked, std::min(slab, m->variants) * m->pitch));
  }
  for (size_t row0 = 0; row0 < m->variants && e == hipSuccess; row0 += slab) {
    const size_t rows = std::min(slab, m->variants - row0);
    const size_t s_pad = round_up(rows, ksites);  // sites
    const size_t k_bytes = s_pad / spb;           // K bytes per sample in this slab
    if (from_packed) {
      // bit rows are tiny in LDS: 256 samples per workgroup (64-byte row pieces for diploid samples)
      const uint32_t sbp = 256;
      const size_t bitb = ((size_t)sbp * m->ploidy + 7) / 8 + 1;
      const size_t smem_p = 3 * ksites * bitb;
      const dim3 grid_p((unsigned)(s_pad / ksites), (unsigned)(n_pad / sbp));
      const uint8_t* q0 = m->p0 + row0 * m->plane_pitch;
      const uint8_t* q1 = m->p1 ? m->p1 + row0 * m->plane_pitch : nullptr;
      const uint8_t* qc = m->pc ? m->pc + row0 * m->plane_pitch : nullptr;
      const void* fn = fp4 ? (const void*)pd_planes_packed_kernel<true> : (const void*)pd_planes_packed_kernel<false>;
      if (smem_p > 64 * 1024 && (e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p)) != hipSuccess) break;
      if (fp4)
        hipLaunchKernelGGL(pd_planes_packed_kernel<true>, grid_p, dim3(256), smem_p, st, q0, q1, qc, m->plane_pitch, rows, (uint32_t)n_samples,
                           (uint32_t)m->ploidy, n_alleles, n_planes, single ? 1 : 0, single ? 1 : 0, sbp, planes, n_pad, s_pad, layout);
      else
        hipLaunchKernelGGL(pd_planes_packed_kernel<false>, grid_p, dim3(256), smem_p, st, q0, q1, qc, m->plane_pitch, rows, (uint32_t)n_samples,
                           (uint32_t)m->ploidy, n_alleles, n_planes, single ? 1 : 0, single ? 1 : 0, sbp, planes, n_pad, s_pad, layout);
    } else {
      if (!m->data && !via_unpack) { e = hipErrorInvalidValue; break; }
      MatrixView mv{};
      mv.pitch = m->pitch;
      mv.columns = m->columns;
      mv.nvec = m->nvec;
      if (via_unpack) {
        if ((e = unpack_rows(m, row0, rows, unpacked, m->pitch, st)) != hipSuccess) break;
        mv.data = unpacked;
        mv.bits = m->pc ? m->pc + row0 * m->plane_pitch : nullptr;  // a called plane row has the bit order of a called bit-row
        mv.bits_pitch = m->plane_pitch;
      } else {
        mv.data = m->data + row0 * m->pitch;
        mv.bits = m->bits ? m->bits + row0 * m->bits_pitch : nullptr;
        mv.bits_pitch = m->bits_pitch;
      }
      // samples per planes workgroup: the tile's raw bytes (one K block of sites x sb x ploidy) stay within 32 KiB of LDS, so
      // four workgroups share a CU and one's loads overlap another's packing (measured, 1 M x 2 500 FP4: 64 KiB tiles 1.94 ms,
      // 32 KiB 1.76 ms, 16 KiB 1.97 ms)
      const uint32_t env_sb = (uint32_t)options().pd_sb.load();  // measurements
      uint32_t sb = env_sb ? env_sb : kPdBlock;
      while ((size_t)sb * m->ploidy * ksites > 32 * 1024 && sb > 4) sb /= 2;
      const size_t planes_smem = ksites * ((size_t)sb * m->ploidy + ((size_t)sb * m->ploidy + 7) / 8 + 1);
      const void* planes_fn = fp4 ? (const void*)pd_planes_kernel<true> : (const void*)pd_planes_kernel<false>;
      if (planes_smem > 64 * 1024 && (e = hipFuncSetAttribute(planes_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)planes_smem)) != hipSuccess) break;
      const dim3 planes_grid((unsigned)(s_pad / ksites), (unsigned)(n_pad / sb));
      if (fp4)
        hipLaunchKernelGGL(pd_planes_kernel<true>, planes_grid, dim3(256), planes_smem, st, mv, rows, (uint32_t)n_samples, (uint32_t)m->ploidy, n_alleles,
                           n_planes, single ? 1 : 0, single ? 1 : 0, sb, planes, n_pad, s_pad, layout);
      else
        hipLaunchKernelGGL(pd_planes_kernel<false>, planes_grid, dim3(256), planes_smem, st, mv, rows, (uint32_t)n_samples, (uint32_t)m->ploidy, n_alleles,
                           n_planes, single ? 1 : 0, single ? 1 : 0, sb, planes, n_pad, s_pad, layout);
    }
    if ((e = hipGetLastError()) != hipSuccess) break;
    // persistent grid: as many workgroups per CU as are resident, dealt round-robin to the 8 XCDs; K is cut into 8 * j slices, j per XCD, sized so
    // that every XCD has several rounds of (slice, tile pair) items (balance) but a slice still spans many stages
    static thread_local int gram_occ[64][2];
    if (gram_occ[m->device][fp4] == 0) {
      int occ = 0;
      hipError_t oe = hipFuncSetAttribute(fp4 ? (const void*)pd_gram256_kernel<4, 4, true> : (const void*)pd_gram256_kernel<4, 4, false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kPdBigStageBytes);
      if (oe == hipSuccess) oe = fp4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pd_gram256_kernel<4, 4, true>, 1024, 2 * kPdBigStageBytes)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pd_gram256_kernel<4, 4, false>, 1024, 2 * kPdBigStageBytes);
      if (oe != hipSuccess || occ < 1) occ = 1;
      const int env_occ = (int)options().pd_occ.load();
      if (env_occ > 0 && occ > env_occ) occ = env_occ;
      gram_occ[m->device][fp4] = occ;
    }
    static thread_local bool phased_ready[64][2];
    if (phased && !phased_ready[m->device][fp4]) {
      e = hipFuncSetAttribute(fp4 ? (const void*)pd_gram256p_kernel<true> : (const void*)pd_gram256p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPdPhaseLdsBytes);
      if (e != hipSuccess) break;
      phased_ready[m->device][fp4] = true;
    }
    // persistent: every workgroup resident (the phased kernel's two K tiles of LDS leave one workgroup per CU)
    const unsigned grid = phased ? (unsigned)std::max(8, w->cus / 8 * 8) : (unsigned)std::max(8, w->cus * gram_occ[m->device][fp4] / 8 * 8);
    const size_t slots = grid / 8;
    size_t j = env_chunk ? std::max<size_t>(1, (k_bytes + 8 * env_chunk - 1) / (8 * env_chunk)) : std::max<size_t>(1, (slots * 8 + tiles - 1) / tiles);
    // an item's accumulators must stay exact: int32 for the int8 route, integers up to 2^24 in f32 for FP4 (counts <= ploidy)
    const size_t cap_sites = (fp4 ? ((size_t)1 << 24) : (((size_t)1 << 31) - 1)) / (m->ploidy * m->ploidy);
    const size_t k_cap = std::max<size_t>(cap_sites / spb / kPdStageK, 1) * kPdStageK;
    size_t k_chunk = round_up((k_bytes + 8 * j - 1) / (8 * j), kPdStageK);
    const size_t k_floor = std::min<size_t>(k_bytes, 4096);  // at least 32 stages per item unless the slab is shorter
    if (k_chunk < k_floor) k_chunk = k_floor;
    if (k_chunk > k_cap) k_chunk = k_cap;
    j = ((k_bytes + k_chunk - 1) / k_chunk + 7) / 8;
    // the phased kernel writes every item's partial tile as a plain slab and a second kernel sums the slabs (no atomics) when the slabs fit the budget
    int* slabs = nullptr;
    if (phased && options().pd_slabs.load() != 0) {
      const size_t slab_bytes = tiles * 8 * j * (size_t)kPdBig * kPdBig * sizeof(int);
      if (slab_bytes <= (size_t)options().pd_slab_bytes.load()) {
        if (w->pd_slab_bytes < slab_bytes) {
          if (w->pd_slabs) (void)hipFree(w->pd_slabs);
          w->pd_slabs = nullptr;
          w->pd_slab_bytes = 0;
          if ((e = hipMalloc((void**)&w->pd_slabs, slab_bytes)) != hipSuccess) break;
          w->pd_slab_bytes = slab_bytes;
        }
        slabs = w->pd_slabs;
      }
    }
    auto gram = [&](int plane_begin, int plane_count, int negate, unsigned long long* dst, unsigned long long* totals = nullptr) {
      if (phased) {
        if (fp4)
          hipLaunchKernelGGL((pd_gram256p_kernel<true>), dim3(grid), dim3(kPdPhaseThreads), kPdPhaseLdsBytes, st, planes, n_pad, k_bytes, plane_begin, plane_count,
                             k_chunk, (uint32_t)j, (uint32_t)n_samples, negate, dst, totals, slabs);
        else
          hipLaunchKernelGGL((pd_gram256p_kernel<false>), dim3(grid), dim3(kPdPhaseThreads), kPdPhaseLdsBytes, st, planes, n_pad, k_bytes, plane_begin, plane_count,
                             k_chunk, (uint32_t)j, (uint32_t)n_samples, negate, dst, totals, slabs);
        if (slabs && hipGetLastError() == hipSuccess) {
          const size_t threads = tiles * 16384;
          hipLaunchKernelGGL(pd_slab_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, slabs, (uint32_t)nt, (uint32_t)j, k_chunk, k_bytes,
                             (uint32_t)n_samples, negate, dst, totals);
        }
        return hipGetLastError();
      }
      if (fp4)
        hipLaunchKernelGGL((pd_gram256_kernel<4, 4, true>), dim3(grid), dim3(1024), 2 * kPdBigStageBytes, st, planes, n_pad, k_bytes, plane_begin, plane_count,
                           k_chunk, (uint32_t)j, (uint32_t)n_samples, negate, dst, totals);
      else
        hipLaunchKernelGGL((pd_gram256_kernel<4, 4, false>), dim3(grid), dim3(1024), 2 * kPdBigStageBytes, st, planes, n_pad, k_bytes, plane_begin, plane_count,
                           k_chunk, (uint32_t)j, (uint32_t)n_samples, negate, dst, totals);
      return hipGetLastError();
    };
    if (single) {
      if ((e = gram(0, 1, 0, d_gram, d_totals)) != hipSuccess) break;
      continue;
    }
    // diff = sum len_i len_j - sum_a cnt_i(a) cnt_j(a)
    if ((e = gram(0, n_alleles, 1, d_diff)) != hipSuccess) break;
    if (missing) {
      if ((e = gram(n_alleles, 1, 0, d_diff)) != hipSuccess) break;
      if ((e = gram(n_alleles + 1, 1, 0, d_both)) != hipSuccess) break;
    }
  }
  if (e == hipSuccess && single) {
    const size_t total = n_samples * n_samples;
    hipLaunchKernelGGL(pd_single_plane_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_diff, d_both, d_gram, d_totals,
                       (uint32_t)n_samples, (unsigned long long)m->ploidy, (unsigned long long)m->variants);
    e = hipGetLastError();
  } else if (e == hipSuccess && !missing) {
    const size_t total = n_samples * n_samples;
    hipLaunchKernelGGL(pd_constant_terms_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_diff, d_both, (uint32_t)n_samples,
                       (unsigned long long)m->variants * m->ploidy * m->ploidy, (unsigned long long)m->variants);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (unpacked) { if (e != hipSuccess) (void)hipStreamSynchronize(st); (void)hipFree(unpacked); }
  if (e != hipSuccess) return fail(FMH_ERR_HIP, "pairwise differences failed: %s", hipGetErrorString(e));
  scratch.settled = true;
  return FMH_OK;
}

