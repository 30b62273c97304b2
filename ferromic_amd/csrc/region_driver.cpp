// region_driver.cpp — run_vcf's device side and per-region driver: region matrices over the C-ABI of libferromic_hip.so, the statistics of
// process.rs:821-1188 and stats.rs on top of the sweeps, CSV-defined populations, the per-config-entry driver (process.rs:2468-3653),
// flags and main().  Every statistic over genotype data is computed on the GPU.
#include "run_vcf.hpp"

namespace {

using namespace fmv;
using std::string;
using std::vector;

// ---- device side --------------------------------------------------------------------------------------
struct GroupsHandle;
struct DeviceMatrix {
  fmh_matrix* h = nullptr;
  size_t variants = 0, samples = 0, ploidy = 0;
  // the group-mask sets already uploaded for this matrix (a region sweeps haplotype groups 0 / 1 twice: the fused pair sweep and W&C).  One
  // worker owns a region's matrices, so no lock.
  mutable vector<std::pair<vector<uint8_t>, std::shared_ptr<GroupsHandle>>> recent_groups;
  ~DeviceMatrix();
  size_t columns() const { return samples * ploidy; }
};

struct DevBuf {
  int device;
  void* p = nullptr;
  DevBuf(int dev, size_t bytes) : device(dev) { fmh_check(fmh_device_alloc(dev, bytes, &p), "device alloc"); }
  ~DevBuf() { if (p) fmh_device_free(device, p); }
  template <class T> vector<T> fetch(size_t n) const {
    vector<T> out(n);
    // (the thread's own stream: the sweep that wrote the block has completed, and the legacy default stream would wait for every other region worker)
    if (n) fmh_check(fmh_copy_to_host(device, out.data(), p, n * sizeof(T), FMH_STREAM_PER_THREAD), "copy to host");
    return out;
  }
};

// DenseGenotypeMatrix::from_variants (stats.rs:339-500), uploaded.  When no genotype is called at all the
// reference has no dense matrix; a one-allele all-missing matrix carries the same (empty) information for
// the sparse formulas, and `has_dense` records which arm the reference would take.
// One region's matrix, whole on one GPU or - a large region with --devices N - as N contiguous site slabs, one per GPU
// (SURVEY.md 8e): per-site tracks are computed by the GPU that owns the sites, the regional accumulators are summed
// through the library's communicator (RCCL; an in-process rendezvous when --devices lists a GPU twice).
struct Slab {
  std::shared_ptr<DeviceMatrix> dm;
  int device = 0;
  size_t row0 = 0;
  fmh_comm* comm = nullptr;  // null: the only slab
};
struct RegionMatrix {
  std::shared_ptr<DeviceMatrix> dm;  // slab 0 (the whole matrix when not sharded): geometry queries
  vector<Slab> slabs;
  size_t variants = 0;               // over all slabs
  bool has_dense = false;
  size_t ploidy = 0;
};

// --devices N: the communicators of the run (one per listed device) and when a region is worth splitting
struct ShardState {
  vector<int> devices;
  vector<fmh_comm*> comms;
  size_t min_bytes = (size_t)256 << 20;  // FERROMIC_SHARD_MIN_BYTES: matrices smaller than this stay whole
  std::mutex region_mutex;               // one sharded region at a time (the communicators are shared)
  std::atomic<bool> broken{false};       // a slab failed and the group was aborted: the communicators are re-created before the next sharded region
  int inject_slab = -1;                  // FERROMIC_INJECT_SLAB_FAILURE=<slab>[:<nth on_slabs call>]: tests of the failure path
  std::atomic<int> inject_countdown{0};
} g_shard;

// A slab that fails must not leave its peers inside a collective (they would wait for ever: the in-process rendezvous has no timeout, RCCL
// neither, and the region holds g_shard.region_mutex).  The failing thread aborts EVERY communicator of the group - fmh_comm_abort wakes the
// peers of the in-process transport with an error and calls ncclCommAbort, which also releases a peer blocked on its communicator's stream -
// so the region ends with an error on every slab and is logged as dropped.  The communicators are unusable afterwards: restore_shard_group()
// re-creates them before the next sharded region (or leaves later regions whole when that fails).
void abort_shard_group() {
  // once per failure: the peers that fail BECAUSE the group was aborted come through here too
  if (g_shard.broken.exchange(true)) return;
  for (fmh_comm* c : g_shard.comms) if (c) fmh_comm_abort(c);
}
void restore_shard_group() {  // called with region_mutex held
  if (!g_shard.broken) return;
  for (fmh_comm* c : g_shard.comms) fmh_comm_destroy(c);
  g_shard.comms.assign(g_shard.devices.size(), nullptr);
  const int rc = fmh_comm_init_all(g_shard.devices.data(), (int)g_shard.devices.size(), g_shard.comms.data());
  if (rc != FMH_OK) {
    logmsg("WARN", string("the communicators could not be re-created after a failed region; large regions stay whole from here on: ") + fmh_last_error());
    g_shard.comms.clear();
  } else {
    logmsg("WARN", "the communicators of --devices were re-created after a failed region");
  }
  g_shard.broken = false;
}

bool want_shard(size_t rows, size_t n_samples) {
  if (g_shard.comms.size() < 2) return false;
  return rows >= 64 * g_shard.comms.size() && rows * n_samples * 2 >= g_shard.min_bytes;
}

// runs fn(slab, k) for every slab: inline for one slab, one thread per slab otherwise (the collectives inside need every
// slab to take part at the same time); the first error is rethrown after all threads have ended
template <class F> void on_slabs(const RegionMatrix& rm, F fn) {
  if (rm.slabs.size() <= 1) { if (!rm.slabs.empty()) fn(rm.slabs[0], (size_t)0); return; }
  vector<std::exception_ptr> errs(rm.slabs.size());
  vector<std::thread> pool;
  const bool inject = g_shard.inject_slab >= 0 && g_shard.inject_countdown.fetch_sub(1) == 0;
  for (size_t k = 0; k < rm.slabs.size(); ++k)
    pool.emplace_back([&, k]() {
      try {
        if (inject && (size_t)g_shard.inject_slab == k) throw Error("injected failure of slab " + std::to_string(k) + " (FERROMIC_INJECT_SLAB_FAILURE)");
        fn(rm.slabs[k], k);
      } catch (...) {
        errs[k] = std::current_exception();
        abort_shard_group();  // the peers may be waiting for this slab inside a collective
      }
    });
  for (auto& t : pool) t.join();
  // the slab that failed FIRST carries the cause; its peers only report that the group was aborted
  std::exception_ptr first;
  for (auto& e : errs) {
    if (!e) continue;
    if (!first) first = e;
    try { std::rethrow_exception(e); } catch (const std::exception& ex) { if (!strstr(ex.what(), "aborted")) { first = e; break; } } catch (...) {}
  }
  if (first) std::rethrow_exception(first);
}

// The common shape - every line diploid over all samples (what process_variant stores for `a|b` cells), alleles up to 7 - goes from the
// parsed lines straight to BIT PLANES on the host (SSE2, 16 entries per instruction; the missing sentinel 0xFF and the rule that a
// genotype ends at its first missing allele become the called plane) and up through fmh_matrix_create_packed: no u8 intermediate (1 GB
// for 200 000 sites x 5 000 haplotypes), an eighth to three eighths of the bytes over PCIe.  Anything else (ragged ploidy, short lines,
// alleles beyond 7) returns false and takes the general u8 route below.
bool build_matrix_planes(const vector<const Variant*>& vs, size_t n_samples, size_t P, int device, bool shard, RegionMatrix& out) {
  if (P != 2 || getenv("FERROMIC_NO_HOST_PLANES")) return false;
  for (auto* v : vs) if (v->stride != 2 || v->num_samples < n_samples) return false;
  const size_t columns = n_samples * 2, row_bytes = (columns + 7) / 8, pitch = (row_bytes + 15) / 16 * 16, S = vs.size();
  std::unique_ptr<uint8_t[]> planes(new uint8_t[4 * S * pitch]);  // [p0 | p1 | p2 | called][S][pitch], not zero-filled: every byte is written below
  uint8_t* pl[4] = {planes.get(), planes.get() + S * pitch, planes.get() + 2 * S * pitch, planes.get() + 3 * S * pitch};
  const unsigned T = S * columns < ((size_t)4 << 20) ? 1u : (unsigned)std::min<size_t>(worker_threads(), S);
  vector<uint8_t> t_max(T, 0), t_missing(T, 0);
  parallel_for(T, [&](unsigned t) {
    const __m128i ff = _mm_set1_epi8((char)0xFF), lo_bytes = _mm_set1_epi16(0x00FF);
    __m128i mx = _mm_setzero_si128();
    bool any = false;
    for (size_t i = S * t / T; i < S * (t + 1) / T; ++i) {
      const uint8_t* row = vs[i]->data.data();
      const size_t o = i * pitch;
      for (int k = 0; k < 4; ++k) memset(pl[k] + o + row_bytes / 16 * 16, 0, pitch - row_bytes / 16 * 16);  // the tail vector of the row
      for (size_t c = 0; c < columns; c += 16) {
        __m128i v;
        uint32_t valid = 0xFFFFu;
        if (c + 16 <= columns) {
          v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + c));
        } else {
          alignas(16) uint8_t tmp[16];
          memset(tmp, 0xFF, 16);
          memcpy(tmp, row + c, columns - c);
          v = _mm_load_si128(reinterpret_cast<const __m128i*>(tmp));
          valid = (1u << (columns - c)) - 1u;
        }
        // missing: the sentinel, and the second allele of a sample whose first is missing (CompressedGenotypes::get, process.rs:479-496)
        __m128i miss = _mm_cmpeq_epi8(v, ff);
        miss = _mm_or_si128(miss, _mm_slli_epi16(_mm_and_si128(miss, lo_bytes), 8));
        const uint32_t called = ~(uint32_t)_mm_movemask_epi8(miss) & valid;
        const __m128i vc = _mm_andnot_si128(miss, v);  // called entries keep their allele, missing ones read 0 (as the u8 route stores them)
        mx = _mm_max_epu8(mx, vc);
        any |= called != valid;
        const uint16_t b0 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(vc, 7)), b1 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(vc, 6)),
                       b2 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(vc, 5)), bc = (uint16_t)called;
        memcpy(pl[0] + o + (c >> 3), &b0, 2);
        memcpy(pl[1] + o + (c >> 3), &b1, 2);
        memcpy(pl[2] + o + (c >> 3), &b2, 2);
        memcpy(pl[3] + o + (c >> 3), &bc, 2);
      }
    }
    alignas(16) uint8_t lanes[16];
    _mm_store_si128(reinterpret_cast<__m128i*>(lanes), mx);
    t_max[t] = *std::max_element(lanes, lanes + 16);
    t_missing[t] = any;
  });
  bool any_missing = false;
  uint8_t max_allele = 0;
  for (unsigned t = 0; t < T; ++t) { any_missing |= t_missing[t] != 0; max_allele = std::max(max_allele, t_max[t]); }
  if (max_allele > 7) return false;  // beyond three planes: u8 rows
  const size_t runs = (S + 63) / 64;
  const size_t D = shard ? g_shard.comms.size() : 1;
  for (size_t k = 0; k < D; ++k) {
    Slab sl;
    sl.row0 = D == 1 ? 0 : std::min(S, (runs * k / D) * 64);
    sl.device = D == 1 ? device : g_shard.devices[k];
    sl.comm = D == 1 ? nullptr : g_shard.comms[k];
    out.slabs.push_back(sl);
  }
  on_slabs(out, [&](const Slab& csl, size_t k) {
    Slab& sl = out.slabs[k];
    const size_t r1 = k + 1 < D ? out.slabs[k + 1].row0 : S, o = csl.row0 * pitch;
    sl.dm.reset(new DeviceMatrix());
    sl.dm->variants = r1 - csl.row0; sl.dm->samples = n_samples; sl.dm->ploidy = 2;
    // every slab with the REGION's max_allele and mask presence (see the note on formula arms in build_matrix)
    fmh_check(fmh_matrix_create_packed(pl[0] + o, max_allele >= 2 ? pl[1] + o : nullptr, max_allele >= 4 ? pl[2] + o : nullptr, any_missing ? pl[3] + o : nullptr,
                                       pitch, r1 - csl.row0, n_samples, 2, max_allele, sl.device, &sl.dm->h), "matrix upload");
  });
  out.dm = out.slabs[0].dm;
  return true;
}

RegionMatrix build_matrix(const vector<const Variant*>& vs, size_t n_samples, int device, bool shard = false) {
  RegionMatrix out;
  if (vs.empty()) return out;
  size_t max_ploidy = 0;
  for (auto* v : vs) max_ploidy = std::max(max_ploidy, v->max_len);
  out.has_dense = max_ploidy > 0;
  const size_t P = std::max<size_t>(max_ploidy, 1);
  out.ploidy = P;
  out.variants = vs.size();
  if (build_matrix_planes(vs, n_samples, P, device, shard, out)) return out;  // the common shape: straight to bit planes
  const size_t stride = n_samples * P, total = vs.size() * stride;
  vector<uint8_t> data(total, 0);
  vector<uint64_t> missing((total + 63) / 64, 0);
  // rows are dealt out in runs of 64 so that no two threads share a 64-bit word of the missing mask
  const size_t runs = (vs.size() + 63) / 64;
  // threads only when there is enough to pack: a config of many small regions must not pay thread start-up per region
  const unsigned T = total < ((size_t)4 << 20) ? 1u : (unsigned)std::min<size_t>(worker_threads(), runs);
  vector<uint8_t> t_max(T, 0), t_missing(T, 0);
  parallel_for(T, [&](unsigned t) {
    const size_t r0 = runs * t / T * 64, r1 = std::min(vs.size(), runs * (t + 1) / T * 64);
    uint8_t mx = 0;
    bool any = false;
    for (size_t i = r0; i < r1; ++i) {
      const Variant& v = *vs[i];
      uint8_t* row = &data[i * stride];
      const size_t vs_stride = v.stride, ns = std::min(n_samples, v.num_samples);
      for (size_t s = 0; s < n_samples; ++s) {
        size_t k = 0;
        if (s < ns) {
          const uint8_t* g = &v.data[s * vs_stride];
          const size_t lim = std::min(P, vs_stride);
          for (; k < lim && g[k] != 0xFF; ++k) { row[s * P + k] = g[k]; mx = std::max(mx, g[k]); }
        }
        for (; k < P; ++k) {
          const size_t idx = i * stride + s * P + k;
          missing[idx >> 6] |= 1ull << (idx & 63);
          any = true;
        }
      }
    }
    t_max[t] = mx;
    t_missing[t] = any;
  });
  bool any_missing = false;
  uint8_t max_allele = 0;
  for (unsigned t = 0; t < T; ++t) { any_missing |= t_missing[t] != 0; max_allele = std::max(max_allele, t_max[t]); }
  // Slabs start at multiples of 64 rows (a slab's slice of the missing bitset then starts on a word boundary) and every slab is
  // created with the REGION's ploidy, max_allele and mask presence: the reference picks its formula arms from the region-wide
  // matrix (stats.rs:3191 / 3218, 4454 / 4485), so a slab without missing calls must still take the masked arms if the region has any
  const size_t D = shard ? g_shard.comms.size() : 1;
  for (size_t k = 0; k < D; ++k) {
    Slab sl;
    sl.row0 = D == 1 ? 0 : std::min(vs.size(), (runs * k / D) * 64);
    sl.device = D == 1 ? device : g_shard.devices[k];
    sl.comm = D == 1 ? nullptr : g_shard.comms[k];
    out.slabs.push_back(sl);
  }
  on_slabs(out, [&](const Slab& csl, size_t k) {
    Slab& sl = out.slabs[k];
    const size_t r1 = k + 1 < D ? out.slabs[k + 1].row0 : vs.size();
    sl.dm.reset(new DeviceMatrix());
    sl.dm->variants = r1 - csl.row0; sl.dm->samples = n_samples; sl.dm->ploidy = P;
    fmh_check(fmh_matrix_create(data.data() + csl.row0 * stride, any_missing ? missing.data() + (csl.row0 * stride) / 64 : nullptr, r1 - csl.row0, n_samples, P,
                                max_allele, sl.device, &sl.dm->h), "matrix upload");
  });
  out.dm = out.slabs[0].dm;
  return out;
}

vector<uint8_t> mask_of(const HapList& haps, size_t n_samples, size_t ploidy, bool dense_rules) {
  // DenseMembership::build (stats.rs:1252-1284) / HapMembership::build (1212-1238): identical as column masks
  // except that the dense one drops Right when ploidy <= 1 (a side that does not exist is never called anyway)
  vector<uint8_t> m(n_samples * ploidy, 0);
  for (auto& h : haps) {
    if (h.first >= n_samples || (size_t)h.second >= ploidy) continue;
    (void)dense_rules;
    m[h.first * ploidy + h.second] = 1;
  }
  return m;
}
size_t mask_count(const vector<uint8_t>& m) { size_t c = 0; for (uint8_t x : m) c += x; return c; }
// HapMembership::total (stats.rs:1212-1238): distinct (sample, side) pairs inside sample_count, whatever the ploidy
size_t membership_total(const HapList& haps, size_t sample_count) {
  std::set<std::pair<size_t, int>> seen;
  for (auto& h : haps) if (h.first < sample_count) seen.insert(h);
  return seen.size();
}

struct GroupsHandle {
  fmh_groups* h = nullptr;
  ~GroupsHandle() { if (h) fmh_groups_destroy(h); }
};
DeviceMatrix::~DeviceMatrix() {
  recent_groups.clear();  // the group handles go before the matrix they were made for
  if (h) fmh_matrix_destroy(h);
}
// the device handle of `masks` over `dm`: made once per matrix and mask set (every handle is a device allocation and a blocking copy)
struct Groups {
  fmh_groups* h = nullptr;
  std::shared_ptr<GroupsHandle> keep;
  Groups(const DeviceMatrix& dm, const vector<vector<uint8_t>>& masks) {
    vector<uint8_t> flat;
    flat.push_back((uint8_t)masks.size());
    for (auto& m : masks) flat.insert(flat.end(), m.begin(), m.end());
    for (auto& e : dm.recent_groups)
      if (e.first == flat) { keep = e.second; h = keep->h; return; }
    keep = std::make_shared<GroupsHandle>();
    fmh_check(fmh_groups_create(dm.h, flat.data() + 1, (int)masks.size(), &keep->h), "groups");
    h = keep->h;
    if (dm.recent_groups.size() < 8) dm.recent_groups.emplace_back(std::move(flat), keep);
  }
};

// ---- sweeps over a (possibly sharded) region matrix -------------------------------------------------------------
// Each returns what the single-matrix C-ABI call returns: per-site tracks assembled in site order from the slabs, totals summed
// over the slabs through the communicator (fmh_*_totals_pack -> fmh_allreduce_totals -> unpack; every slab ends with the sums).
vector<fmh_pop_totals> sharded_summaries(const RegionMatrix& rm, const vector<vector<uint8_t>>& masks, int formula, vector<uint32_t>* called0 = nullptr) {
  const size_t G = masks.size();
  vector<vector<fmh_pop_totals>> per(rm.slabs.size(), vector<fmh_pop_totals>(G));
  if (called0) called0->assign(rm.variants, 0);
  on_slabs(rm, [&](const Slab& sl, size_t k) {
    const size_t rows = sl.dm->variants;
    Groups grp(*sl.dm, masks);
    std::unique_ptr<DevBuf> dcalled;
    if (called0) dcalled.reset(new DevBuf(sl.device, 4 * G * std::max<size_t>(rows, 1)));
    // everything that can fail on this slab alone (groups, buffers) is done; the sharded call validates before it enqueues, then
    // sweep -> finalise -> ncclAllReduce on the device -> one D2H (no host hop between the sweep and the reduce)
    if (sl.comm) fmh_check(fmh_population_summaries_sharded(sl.comm, sl.dm->h, grp.h, 0, rows, formula, nullptr, dcalled ? (uint32_t*)dcalled->p : nullptr, per[k].data(), nullptr), "sharded summaries");
    else fmh_check(fmh_population_summaries(sl.dm->h, grp.h, 0, rows, formula, nullptr, dcalled ? (uint32_t*)dcalled->p : nullptr, per[k].data(), nullptr), "summaries");
    if (called0 && rows) { vector<uint32_t> c = dcalled->fetch<uint32_t>(rows); std::copy(c.begin(), c.end(), called0->begin() + (ptrdiff_t)sl.row0); }
  });
  return per[0];
}

// fused W&C sweep (2..8 groups); tracks [nw][S] when wanted
fmh_wc_totals sharded_wc(const RegionMatrix& rm, const vector<vector<uint8_t>>& masks, vector<double>* a, vector<double>* b, vector<uint8_t>* st) {
  const size_t G = masks.size(), nw = 1 + G * (G - 1) / 2, S = rm.variants;
  if (a) { a->assign(nw * S, 0.0); b->assign(nw * S, 0.0); st->assign(nw * S, 0); }
  vector<fmh_wc_totals> per(rm.slabs.size());
  on_slabs(rm, [&](const Slab& sl, size_t k) {
    const size_t rows = sl.dm->variants;
    Groups grp(*sl.dm, masks);
    // the three track families in ONE device block, fetched with ONE copy: a small region is latency, and every blocking copy is ~20 us
    // (a [nw][rows] f64 | b [nw][rows] f64 | state [nw][rows] u8)
    std::unique_ptr<DevBuf> blk;
    double *da = nullptr, *db = nullptr;
    uint8_t* ds = nullptr;
    if (a && rows) {
      blk.reset(new DevBuf(sl.device, 17 * nw * rows));
      da = (double*)blk->p; db = da + nw * rows; ds = (uint8_t*)(db + nw * rows);
    }
    if (sl.comm) fmh_check(fmh_wc_sweep_sharded(sl.comm, sl.dm->h, grp.h, 0, rows, da, db, ds, nullptr, &per[k], nullptr), "sharded wc sweep");
    else fmh_check(fmh_wc_sweep(sl.dm->h, grp.h, 0, rows, da, db, ds, nullptr, &per[k], nullptr), "wc sweep");
    if (blk) {
      const vector<uint8_t> host = blk->fetch<uint8_t>(17 * nw * rows);
      const double* ha = reinterpret_cast<const double*>(host.data());
      const double* hb = ha + nw * rows;
      const uint8_t* hs = host.data() + 16 * nw * rows;
      for (size_t w = 0; w < nw; ++w) {
        std::copy(ha + w * rows, ha + (w + 1) * rows, a->begin() + (ptrdiff_t)(w * S + sl.row0));
        std::copy(hb + w * rows, hb + (w + 1) * rows, b->begin() + (ptrdiff_t)(w * S + sl.row0));
        std::copy(hs + w * rows, hs + (w + 1) * rows, st->begin() + (ptrdiff_t)(w * S + sl.row0));
      }
    }
  });
  return per[0];
}

// more than 8 groups: counting sweeps in batches + the counts kernel; regional sums only
void sharded_wc_many(const RegionMatrix& rm, const vector<uint8_t>& flat_masks, size_t G, vector<double>& sum_a, vector<double>& sum_b, vector<uint64_t>& informative) {
  const size_t nw = 1 + G * (G - 1) / 2;
  vector<vector<double>> fa(rm.slabs.size(), vector<double>(2 * nw, 0.0));
  vector<vector<uint64_t>> ui(rm.slabs.size(), vector<uint64_t>(nw, 0));
  on_slabs(rm, [&](const Slab& sl, size_t k) {
    fmh_check(fmh_wc_sweep_many(sl.dm->h, flat_masks.data(), (int)G, 0, sl.dm->variants, nullptr, nullptr, nullptr, nullptr, fa[k].data(), fa[k].data() + nw,
                                ui[k].data(), nullptr), "wc sweep (many groups)");
    if (sl.comm) {  // slot vectors beyond the communicator's per-call limit go in pieces
      for (size_t off = 0; off < 2 * nw; off += FMH_COMM_MAX_VALUES)
        fmh_check(fmh_allreduce_totals(sl.comm, fa[k].data() + off, std::min<size_t>(FMH_COMM_MAX_VALUES, 2 * nw - off), nullptr, 0), "all-reduce of the W&C sums");
      for (size_t off = 0; off < nw; off += FMH_COMM_MAX_VALUES)
        fmh_check(fmh_allreduce_totals(sl.comm, nullptr, 0, ui[k].data() + off, std::min<size_t>(FMH_COMM_MAX_VALUES, nw - off)), "all-reduce of the W&C site counts");
    }
  });
  sum_a.assign(fa[0].begin(), fa[0].begin() + (ptrdiff_t)nw);
  sum_b.assign(fa[0].begin() + (ptrdiff_t)nw, fa[0].end());
  informative = ui[0];
}

// Hudson pair sweep: the library's own sharded entry point (sweep, device-side finalise, RCCL reduce, one synchronisation)
// (want_sites = false: the regional totals only - what the CSV-population pairs need, K (K - 1) / 2 sweeps per region: no per-site tracks are
// written on the device and nothing but the totals comes back)
fmh_hudson_totals sharded_hudson(const RegionMatrix& rm, const vector<uint8_t>& m0, const vector<uint8_t>& m1, int formula, vector<double>& fst,
                                 vector<double>& num, vector<double>& den, bool want_sites = true) {
  const size_t S = rm.variants;
  if (want_sites) { fst.assign(S, 0.0); num.assign(S, 0.0); den.assign(S, 0.0); }
  vector<fmh_hudson_totals> per(rm.slabs.size());
  on_slabs(rm, [&](const Slab& sl, size_t k) {
    const size_t rows = sl.dm->variants;
    Groups grp(*sl.dm, {m0, m1});
    if (!want_sites) {
      if (sl.comm) fmh_check(fmh_hudson_sweep_sharded(sl.comm, sl.dm->h, grp.h, 0, rows, formula, nullptr, &per[k], nullptr), "sharded hudson sweep");
      else fmh_check(fmh_hudson_sweep(sl.dm->h, grp.h, 0, rows, formula, nullptr, &per[k], nullptr), "hudson sweep");
      return;
    }
    DevBuf blk(sl.device, 8 * 3 * std::max<size_t>(rows, 1));  // fst | num | den: one block, one copy back
    fmh_hudson_sites sites{};
    sites.d_fst = (double*)blk.p; sites.d_num = sites.d_fst + rows; sites.d_den = sites.d_num + rows;
    if (sl.comm) fmh_check(fmh_hudson_sweep_sharded(sl.comm, sl.dm->h, grp.h, 0, rows, formula, &sites, &per[k], nullptr), "sharded hudson sweep");
    else fmh_check(fmh_hudson_sweep(sl.dm->h, grp.h, 0, rows, formula, &sites, &per[k], nullptr), "hudson sweep");
    if (!rows) return;
    const vector<double> h = blk.fetch<double>(3 * rows);
    std::copy(h.begin(), h.begin() + (ptrdiff_t)rows, fst.begin() + (ptrdiff_t)sl.row0);
    std::copy(h.begin() + (ptrdiff_t)rows, h.begin() + (ptrdiff_t)(2 * rows), num.begin() + (ptrdiff_t)sl.row0);
    std::copy(h.begin() + (ptrdiff_t)(2 * rows), h.end(), den.begin() + (ptrdiff_t)sl.row0);
  });
  return per[0];
}

// The fused region sweep (fmh_pair_region_sweep): groups 0 and 1 of one region matrix in ONE read - both population summaries, both
// groups' per-site diversity and, when asked, the Hudson pair (per-site fst / num / den + totals by the sparse formulas).  Round 2 read the
// matrix once for the summaries, once per group for the diversity tracks and once more for Hudson.
struct PairSweep {
  fmh_hudson_totals tot{};         // pop[0..1]: the summaries; the Hudson fields when `hudson`
  vector<double> pi[2], theta[2];  // per-site diversity of group 0 / 1
  bool hudson = false;
  vector<double> fst, num, den;
};
PairSweep sharded_pair_region(const RegionMatrix& rm, const vector<uint8_t>& m0, const vector<uint8_t>& m1, int summary_formula, bool hudson) {
  PairSweep out;
  out.hudson = hudson;
  const size_t S = rm.variants;
  for (int g = 0; g < 2; ++g) { out.pi[g].assign(S, 0.0); out.theta[g].assign(S, 0.0); }
  if (hudson) { out.fst.assign(S, 0.0); out.num.assign(S, 0.0); out.den.assign(S, 0.0); }
  vector<fmh_hudson_totals> per(rm.slabs.size());
  on_slabs(rm, [&](const Slab& sl, size_t k) {
    const size_t rows = sl.dm->variants, cap = std::max<size_t>(rows, 1);
    Groups grp(*sl.dm, {m0, m1});
    // every track of the sweep in ONE device block (pi [2][rows] | theta [2][rows] | fst | num | den), fetched with ONE copy: a region of a few
    // thousand sites is latency, and each blocking copy costs ~20 us (five of them per sweep before)
    const size_t ntracks = hudson ? 7 : 4;
    DevBuf blk(sl.device, 8 * ntracks * cap);
    double* base = (double*)blk.p;
    fmh_pair_diversity_sites div{base, base + 2 * rows};
    fmh_hudson_sites sites{};
    if (hudson) { sites.d_fst = base + 4 * rows; sites.d_num = base + 5 * rows; sites.d_den = base + 6 * rows; }
    const int hf = hudson ? FMH_FORMULA_SPARSE : -1;
    if (sl.comm) fmh_check(fmh_pair_region_sweep_sharded(sl.comm, sl.dm->h, grp.h, 0, rows, summary_formula, hf, &div, hudson ? &sites : nullptr, &per[k], nullptr), "sharded region sweep");
    else fmh_check(fmh_pair_region_sweep(sl.dm->h, grp.h, 0, rows, summary_formula, hf, &div, hudson ? &sites : nullptr, &per[k], nullptr), "region sweep");
    if (!rows) return;
    const vector<double> h = blk.fetch<double>(ntracks * rows);
    auto slice = [&](size_t track, vector<double>& dst) { std::copy(h.begin() + (ptrdiff_t)(track * rows), h.begin() + (ptrdiff_t)((track + 1) * rows), dst.begin() + (ptrdiff_t)sl.row0); };
    for (int g = 0; g < 2; ++g) { slice((size_t)g, out.pi[g]); slice((size_t)(2 + g), out.theta[g]); }
    if (hudson) { slice(4, out.fst); slice(5, out.num); slice(6, out.den); }
  });
  out.tot = per[0];
  return out;
}

// ---- statistics (host scalars are literal restatements; genotype work is on the GPU) ------------------
double harmonic(size_t n) { double s = 0.0; for (size_t k = 1; k <= n; ++k) s += 1.0 / (double)k; return s; }  // stats.rs:4234
double watterson_theta(size_t S, size_t n, int64_t L) {                                                          // stats.rs:4243-4307
  if (n <= 1 || L <= 0) return S == 0 ? NAN : INFINITY;
  const double h = harmonic(n - 1);
  if (h > 0.0) return (double)S / h / (double)L;
  return S == 0 ? NAN : INFINITY;
}
int64_t sat_sub(int64_t a, int64_t b) { __int128 r = (__int128)a - b; if (r < INT64_MIN) return INT64_MIN; if (r > INT64_MAX) return INT64_MAX; return (int64_t)r; }

vector<Interval> subtract_regions(const vector<Interval>& intervals, const vector<Interval>* masks) {  // stats.rs:3739-3775 (1-based inclusive)
  if (!masks) return intervals;
  vector<Interval> out;
  for (auto& a : intervals) {
    vector<Interval> parts{a};
    for (auto& m : *masks) {
      vector<Interval> next;
      for (auto& p : parts) {
        if (m.second < p.first || m.first > p.second) { next.push_back(p); continue; }
        if (m.first > p.first && m.first - 1 >= p.first) next.push_back({p.first, m.first - 1});
        if (m.second < p.second && m.second + 1 <= p.second) next.push_back({m.second + 1, p.second});
      }
      parts.swap(next);
      if (parts.empty()) break;
    }
    out.insert(out.end(), parts.begin(), parts.end());
  }
  return out;
}

int64_t adjusted_sequence_length(int64_t start1, int64_t end1, const vector<Interval>* allow, const vector<Interval>* mask) {  // stats.rs:3644-3736
  const Interval region = from_1based_inclusive(start1, end1);
  vector<Interval> allowed;
  if (allow) {
    for (auto& a : *allow) {
      const uint64_t s = std::max((uint64_t)region.first, (uint64_t)a.first), e = std::min((uint64_t)region.second, (uint64_t)a.second);
      if (s < e) allowed.push_back({(int64_t)s + 1, (int64_t)e});
    }
  } else allowed.push_back({start1, end1});
  vector<Interval> conv;
  if (mask) for (auto& m : *mask) conv.push_back({(int64_t)((uint64_t)m.first) + 1, (int64_t)(uint64_t)m.second});
  int64_t total = 0;
  for (auto& iv : subtract_regions(allowed, mask ? &conv : nullptr)) total += hal_len(from_1based_inclusive(iv.first, iv.second));
  return total;
}

std::optional<double> inversion_allele_frequency(const SampleMap& m) {  // stats.rs:3778-3805
  size_t ones = 0, total = 0;
  for (auto& kv : m) for (uint8_t a : {kv.second.first, kv.second.second}) if (a == 0 || a == 1) { ++total; ones += a; }
  if (!total) return std::nullopt;
  return (double)ones / (double)total;
}

// calculate_pi_for_population (stats.rs:4599-4614) from the sweep totals of one population:
// calculate_pi_dense (4534-4597) on a diploid dense matrix, calculate_pi (4317-4432) otherwise.
double pi_from_totals(const RegionMatrix& rm, const HapList& haps, const vector<uint8_t>& mask, size_t N, int64_t L,
                      const fmh_pop_totals& tot) {
  const bool dense_arm = rm.has_dense && rm.ploidy == 2;
  const size_t members = dense_arm ? mask_count(mask) : haps.size();  // membership.len() vs haplotypes_in_group.len()
  if (members <= 1) return NAN;
  if (L < 0) return 0.0;
  if (L == 0) return INFINITY;
  if (!dense_arm) {  // stats.rs:4353-4374
    size_t hap_samples = 0;
    for (auto& h : haps) hap_samples = std::max(hap_samples, h.first + 1);
    if (membership_total(haps, std::max(N, hap_samples)) <= 1) return NAN;
  }
  const int64_t eff = sat_sub(L, (int64_t)tot.uncallable_sites);
  return eff == 0 ? NAN : tot.pi_sum / (double)eff;
}

struct SiteDiv { int64_t pos1; double pi, theta; };
struct GroupStats { bool present = false; size_t segsites = 0, n_hap = 0; double theta = 0.0, pi = 0.0; vector<SiteDiv> sites; };

// process_variants (process.rs:821-1188), statistics only, for groups 0 and 1 of one (variant set, sample filter)
// `hud`: when given, the same read also yields the Hudson pair of groups 0 / 1 (hudson_groups then takes it from there)
void process_variants_pair(const vector<const Variant*>& vs, const RegionMatrix& rm, const vector<string>& sample_names,
                           const SampleMap& filter, const Interval& interval, int64_t L, const vector<Interval>* mask_intervals,
                           int device, GroupStats out[2], PairSweep* hud = nullptr) {
  const auto index = map_sample_names_to_indices(sample_names);
  HapList haps[2] = {haplotypes_for_group(0, filter, index), haplotypes_for_group(1, filter, index)};
  const size_t N = sample_names.size();
  for (int g = 0; g < 2; ++g) {
    out[g] = GroupStats();
    if (haps[g].empty()) continue;
    out[g].present = true;
    out[g].n_hap = haps[g].size();
    if (vs.empty()) { out[g].theta = out[g].pi = out[g].n_hap < 2 ? NAN : 0.0; }
  }
  if (vs.empty() || (!out[0].present && !out[1].present)) return;
  const DeviceMatrix& dm = *rm.dm;
  const size_t S = rm.variants;
  // one pass: segregating sites + pi of both groups.  calculate_pi_for_population picks calculate_pi_dense
  // only for a diploid dense matrix (stats.rs:4603-4608), calculate_pi otherwise.
  const bool dense_arm = rm.has_dense && rm.ploidy == 2;
  vector<vector<uint8_t>> masks = {mask_of(haps[0], N, dm.ploidy, true), mask_of(haps[1], N, dm.ploidy, true)};
  // ONE read of the matrix: the summaries of both groups (regional pi by calculate_pi_dense's formulas on a diploid dense matrix, else the
  // sparse ones), both groups' per-site diversity and - for the caller that wants it - the Hudson pair by the sparse per-site formulas
  PairSweep local;
  PairSweep& ps = hud ? *hud : local;
  ps = sharded_pair_region(rm, masks[0], masks[1], dense_arm ? FMH_FORMULA_DENSE : FMH_FORMULA_SPARSE, hud != nullptr);
  const fmh_pop_totals tot[2] = {ps.tot.pop[0], ps.tot.pop[1]};
  for (int g = 0; g < 2; ++g) {
    if (!out[g].present) continue;
    out[g].segsites = (size_t)tot[g].segregating_sites;
    out[g].theta = watterson_theta(out[g].segsites, out[g].n_hap, L);
    out[g].pi = pi_from_totals(rm, haps[g], masks[g], N, L, tot[g]);
    // calculate_per_site_diversity (stats.rs:4628-4806): needs >= 2 listed haplotypes
    if (haps[g].size() < 2 || hal_len(interval) <= 0) continue;
    const vector<double>&pi_v = ps.pi[g], &th_v = ps.theta[g];
    for (size_t i = 0; i < S; ++i) {
      const int64_t pos0 = vs[i]->position;
      if (!hal_contains(interval, pos0)) continue;
      SiteDiv sd{pos0 + 1, pi_v[i], th_v[i]};
      if (mask_intervals && position_in_regions(pos0, *mask_intervals)) sd.pi = sd.theta = NAN;  // stats.rs:4731-4743
      out[g].sites.push_back(sd);
    }
  }
}

struct WcRegion {
  bool computed = false;
  // overall FstEstimate pieces for the CSV (extract_wc_fst_components, stats.rs:4860-4914)
  std::optional<double> value; double sum_a = 0.0, sum_b = 0.0; size_t sites = 0;
  vector<WcSite> per_site;
};

int wc_classify(double a, double b) {  // stats.rs:1781-1812 -> 0 calculable, 1 indeterminate, 2 no variance
  const double d = a + b;
  if (d > 1e-12) return 0;
  if (d < -1e-12) return 1;
  if (std::fabs(a) > 1e-12) return 0;
  return 2;
}

// calculate_fst_wc_haplotype_groups (stats.rs:675-806) on the filtered variants of the region
WcRegion wc_haplotype_groups(const vector<const Variant*>& vs, const RegionMatrix& rm, const vector<string>& sample_names,
                             const SampleMap& filter, int device) {
  WcRegion out;
  out.computed = true;
  const size_t N = sample_names.size();
  const auto index = map_sample_names_to_indices(sample_names);
  // map_samples_to_haplotype_groups + SubpopulationMembership::from_map: labels are the decimal strings, sorted
  std::map<std::pair<size_t, int>, string> hap_to_group;
  for (auto& kv : filter) {
    auto it = index.find(normalize_sample_name(kv.first));
    if (it == index.end()) continue;
    hap_to_group[{it->second, 0}] = std::to_string(kv.second.first);
    hap_to_group[{it->second, 1}] = std::to_string(kv.second.second);
  }
  std::set<string> label_set;
  for (auto& kv : hap_to_group) label_set.insert(kv.second);
  vector<string> labels(label_set.begin(), label_set.end());
  const size_t G = labels.size();
  if (vs.empty()) { out.value = std::nullopt; out.sites = 0; return out; }  // InsufficientData { sites_attempted: 0 }
  const DeviceMatrix& dm = *rm.dm;
  const size_t S = rm.variants, P = dm.ploidy;
  vector<vector<uint8_t>> masks(std::max<size_t>(G, 1), vector<uint8_t>(N * P, 0));
  for (auto& kv : hap_to_group) {
    if (kv.first.first >= N || (size_t)kv.first.second >= P) continue;
    const size_t gi = std::find(labels.begin(), labels.end(), kv.second) - labels.begin();
    masks[gi][kv.first.first * P + kv.first.second] = 1;
  }
  const bool has01 = G == 2 && labels[0] == "0" && labels[1] == "1";
  if (G < 2) {
    // fewer than two groups: a site with any called allele is NoInterPopulationVariance (0, 0), a site with none is
    // InsufficientData (stats.rs:1925-1930, 1987-2003); "any called" comes from an all-columns summary sweep
    vector<uint32_t> called;
    sharded_summaries(rm, {vector<uint8_t>(N * P, 1)}, FMH_FORMULA_SPARSE, &called);
    size_t informative = 0;
    for (size_t i = 0; i < S; ++i) {
      informative += called[i] != 0;
      out.per_site.push_back({vs[i]->position + 1, NAN, 0.0, 0.0, NAN, NAN, NAN});
    }
    out.sites = informative ? informative : S;
    out.value = std::nullopt;
    return out;
  }
  if (G > FMH_MAX_GROUPS) throw Error("more than 8 haplotype groups");
  vector<double> a, b;
  vector<uint8_t> st;
  const fmh_wc_totals tot = sharded_wc(rm, masks, &a, &b, &st);
  for (size_t i = 0; i < S; ++i) {
    WcSite w{vs[i]->position + 1, NAN, 0.0, 0.0, NAN, NAN, NAN};
    if (st[i] != FMH_WC_INSUFFICIENT) {
      w.overall_num = a[i];
      w.overall_den = a[i] + b[i];
      if (wc_classify(a[i], b[i]) == 0) w.overall_fst = a[i] / (a[i] + b[i]);
      if (has01) {  // slot 1 is the only pair
        w.pair_num = a[S + i];
        w.pair_den = a[S + i] + b[S + i];
        if (st[S + i] != FMH_WC_INSUFFICIENT && wc_classify(a[S + i], b[S + i]) == 0) w.pair_fst = a[S + i] / (a[S + i] + b[S + i]);
      }
    }
    out.per_site.push_back(w);
  }
  // calculate_overall_fst_wc (stats.rs:2145-2374)
  if (tot.informative_sites[0] == 0) { out.value = std::nullopt; out.sum_a = out.sum_b = 0.0; out.sites = S; }
  else {
    out.sum_a = tot.sum_a[0]; out.sum_b = tot.sum_b[0]; out.sites = (size_t)tot.informative_sites[0];
    if (wc_classify(out.sum_a, out.sum_b) == 0) out.value = out.sum_a / (out.sum_a + out.sum_b);
  }
  return out;
}

// ---- CSV-defined populations (stats.rs:816-1078, process.rs:3301-3392, 4054-4089) ---------------------------
typedef std::map<string, vector<string>> PopulationCsv;

PopulationCsv parse_population_csv(const string& path) {  // stats.rs:951-1007
  std::ifstream in(path);
  if (!in) throw Error("Failed to open population CSV file " + path);
  PopulationCsv out;
  string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (trim(line).empty() || starts_with(line, "#")) continue;
    vector<string> parts = split(line, ',');
    for (auto& x : parts) x = trim(x);
    if (parts.empty() || parts[0].empty()) continue;
    vector<string> samples;
    for (size_t i = 1; i < parts.size(); ++i) if (!parts[i].empty()) samples.push_back(parts[i]);
    if (!samples.empty()) out[parts[0]] = samples;
  }
  if (out.empty()) throw Error("Population CSV file '" + path + "' contains no valid population data after parsing.");
  return out;
}

struct WcEstimate { int state = 3; std::optional<double> value; double sum_a = 0.0, sum_b = 0.0; size_t sites = 0; };

WcEstimate wc_estimate(double a, double b, size_t sites) {
  WcEstimate e;
  e.state = wc_classify(a, b);
  e.sum_a = a; e.sum_b = b; e.sites = sites;
  if (e.state == 0) e.value = a / (a + b);
  return e;
}

// calculate_fst_wc_csv_populations (stats.rs:816-934): regional overall + pairwise estimates as TSV rows
vector<vector<string>> wc_csv_population_rows(const vector<const Variant*>& vs, const RegionMatrix& rm, const vector<string>& sample_names,
                                              const PopulationCsv& csv, const ConfigEntry& entry, int device) {
  const size_t N = sample_names.size();
  const auto index = map_sample_names_to_indices(sample_names);
  std::map<string, string> sample_to_pop;  // later populations overwrite earlier ones (stats.rs:1062-1067)
  for (auto& kv : csv) for (auto& sid : kv.second) sample_to_pop[sid] = kv.first;
  std::map<size_t, string> idx_to_pop;
  for (auto& kv : sample_to_pop) {
    auto it = index.find(normalize_sample_name(kv.first));
    if (it != index.end()) idx_to_pop[it->second] = kv.second;
  }
  std::set<string> label_set;
  for (auto& kv : idx_to_pop) label_set.insert(kv.second);
  vector<string> labels(label_set.begin(), label_set.end());
  const size_t G = labels.size();
  const string rs = std::to_string(entry.interval.first + 1), re = std::to_string(entry.interval.second);
  auto row = [&](const string& kind, const string& p1, const string& p2, const WcEstimate& e) {
    return vector<string>{entry.seqname, rs, re, kind, p1, p2, fmt_opt(e.value), fmt_opt(e.sum_a), fmt_opt(e.sum_a + e.sum_b), std::to_string(e.sites)};
  };
  vector<vector<string>> rows;
  WcEstimate insufficient;
  if (vs.empty()) { rows.push_back(row("overall", "ALL", "ALL", insufficient)); return rows; }  // sites_attempted 0, no pair keys
  const DeviceMatrix& dm = *rm.dm;
  const size_t S = rm.variants, P = dm.ploidy;
  if (G > FMH_MAX_GROUPS_MANY) throw Error("--fst_populations: more than 256 populations");
  if (G < 2) {
    vector<uint32_t> called;
    sharded_summaries(rm, {vector<uint8_t>(N * P, 1)}, FMH_FORMULA_SPARSE, &called);
    size_t informative = 0;
    for (uint32_t c : called) informative += c != 0;
    WcEstimate e;
    if (informative) { e = wc_estimate(0.0, 0.0, informative); } else { e.sites = S; }
    rows.push_back(row("overall", "ALL", "ALL", e));
    return rows;
  }
  vector<vector<uint8_t>> masks(G, vector<uint8_t>(N * P, 0));
  for (auto& kv : idx_to_pop) {
    const size_t gi = std::find(labels.begin(), labels.end(), kv.second) - labels.begin();
    for (size_t k = 0; k < std::min<size_t>(P, 2); ++k) if (kv.first < N) masks[gi][kv.first * P + k] = 1;
  }
  // totals per slot (0 = overall, then pairs in label order): the fused sweep up to 8 populations, beyond that the
  // counting sweeps in batches of 8 + the counts kernel (fmh_wc_sweep_many)
  struct { vector<double> sum_a, sum_b; vector<uint64_t> informative_sites; } tot;
  const size_t nslots = 1 + G * (G - 1) / 2;
  tot.sum_a.assign(nslots, 0.0); tot.sum_b.assign(nslots, 0.0); tot.informative_sites.assign(nslots, 0);
  if (G <= FMH_MAX_GROUPS) {
    const fmh_wc_totals t8 = sharded_wc(rm, masks, nullptr, nullptr, nullptr);
    for (size_t k2 = 0; k2 < nslots; ++k2) { tot.sum_a[k2] = t8.sum_a[k2]; tot.sum_b[k2] = t8.sum_b[k2]; tot.informative_sites[k2] = t8.informative_sites[k2]; }
  } else {
    vector<uint8_t> flat;
    for (auto& m2 : masks) flat.insert(flat.end(), m2.begin(), m2.end());
    sharded_wc_many(rm, flat, G, tot.sum_a, tot.sum_b, tot.informative_sites);
  }
  WcEstimate overall;
  if (tot.informative_sites[0] == 0) overall.sites = S; else overall = wc_estimate(tot.sum_a[0], tot.sum_b[0], (size_t)tot.informative_sites[0]);
  rows.push_back(row("overall", "ALL", "ALL", overall));
  if (tot.informative_sites[0] != 0) {  // pair keys exist only if some site had any called allele
    std::map<string, WcEstimate> pairs;
    size_t k = 1;
    for (size_t i = 0; i < G; ++i)
      for (size_t j = i + 1; j < G; ++j, ++k) {
        WcEstimate e;
        if (tot.informative_sites[k] != 0) e = wc_estimate(tot.sum_a[k], tot.sum_b[k], (size_t)tot.informative_sites[k]);
        else e.sites = (size_t)tot.informative_sites[0];
        pairs[labels[i] + "_vs_" + labels[j]] = e;
      }
    for (auto& kv : pairs) {
      vector<string> parts;
      size_t b = 0;
      for (;;) { size_t e = kv.first.find("_vs_", b); if (e == string::npos) { parts.push_back(kv.first.substr(b)); break; } parts.push_back(kv.first.substr(b, e - b)); b = e + 4; }
      if (parts.size() == 2) rows.push_back(row("pairwise", parts[0], parts[1], kv.second));
      else rows.push_back(row("pairwise", "unknown", "unknown", kv.second));
    }
  }
  return rows;
}

struct HudsonRegion {
  bool have_outcome = false;
  std::optional<double> fst, dxy, pi0, pi1, avg;
  vector<std::tuple<int64_t, double, double, double>> sites;  // (pos1, fst, num, den)
};

// calculate_hudson_fst_for_pair_with_sites (stats.rs:3619) for haplotype groups 0 / 1 of the filtered set
HudsonRegion hudson_groups(const vector<const Variant*>& vs, const RegionMatrix& rm, const vector<string>& sample_names,
                           const HapList& h0, const HapList& h1, int64_t L, const double pi_raw[2], int device, const PairSweep* pre = nullptr,
                           bool want_sites = true) {
  HudsonRegion out;
  if (L <= 0) return out;  // Err(InvalidRegion) -> logged, no outcome (process.rs:3261-3271)
  out.have_outcome = true;
  double num_sum = 0.0, den_sum = 0.0, dxy_sum = 0.0;
  uint64_t dxy_skipped = 0;
  const size_t N = sample_names.size();
  if (!vs.empty()) {
    const DeviceMatrix& dm = *rm.dm;
    const size_t S = rm.variants;
    vector<double> fst, num, den;
    fmh_hudson_totals tot;
    if (pre && pre->hudson && pre->fst.size() == S) {  // the filtered process_variants sweep of the same two haplotype lists already read the matrix
      tot = pre->tot; fst = pre->fst; num = pre->num; den = pre->den;
    } else {
      tot = sharded_hudson(rm, mask_of(h0, N, dm.ploidy, false), mask_of(h1, N, dm.ploidy, false), FMH_FORMULA_SPARSE, fst, num, den, want_sites);
    }
    num_sum = tot.site_num_sum; den_sum = tot.site_den_sum; dxy_sum = tot.site_dxy_sum; dxy_skipped = tot.site_dxy_skipped;
    if (want_sites) {
      size_t informative = 0;
      for (size_t i = 0; i < S; ++i) informative += (!std::isnan(den[i]) && std::isfinite(den[i]) && den[i] > 0.0);
      if (informative > 0) for (size_t i = 0; i < S; ++i) out.sites.push_back({vs[i]->position + 1, fst[i], num[i], den[i]});
    }
  }
  if (den_sum > 1e-12) out.fst = num_sum / den_sum;
  // auxiliaries (stats.rs:3562-3565): calculate_pi_for_population x2 == the filtered process_variants pi of the same
  // haplotype lists and length; calculate_d_xy_hudson: dense shared / sparse fold share the frequency-dot form
  for (int g = 0; g < 2; ++g) {
    const double raw = pi_raw[g];
    (g == 0 ? out.pi0 : out.pi1) = std::isfinite(raw) ? std::optional<double>(raw) : std::nullopt;
  }
  if (!h0.empty() && !h1.empty()) {
    const bool dense_arm = rm.has_dense && rm.ploidy == 2;
    bool members_ok = true;
    if (dense_arm && !vs.empty()) members_ok = mask_count(mask_of(h0, N, rm.ploidy, true)) && mask_count(mask_of(h1, N, rm.ploidy, true));
    const int64_t eff = sat_sub(L, (int64_t)dxy_skipped);
    if (members_ok && eff > 0) out.dxy = dxy_sum / (double)eff;
  }
  if (out.pi0 && out.pi1) out.avg = 0.5 * (*out.pi0 + *out.pi1);
  return out;
}

// ---- per-region driver (process.rs:2468-3653) ----------------------------------------------------------------
struct Args {
  string vcf_folder, chr, region, config_file, output_file = "output.csv", mask_file, allow_file, reference, gtf, fst_populations;
  vector<string> exclude;
  unsigned min_gq = 30;
  bool enable_fst = false, enable_pca = false;
  int device = 0;
  int workers_per_device = 0;  // region workers per GPU (0 = by the CPU share): host-side packing, downloads and track writers of one region overlap the sweeps of another
  bool print_formats = false;  // diagnostic: header lines + sample FALSTA records (needs no GPU, no inputs)
  bool ingest_only = false;  // diagnostic: parse the inputs, report counts, compute nothing (needs no GPU)
  vector<int> devices;  // --devices: one worker thread per entry, config regions dealt out dynamically
};

std::optional<RegionOutput> process_single_config_entry(const ConfigEntry& entry, const VcfData& vcf, const RegionMap& mask,
                                                        const RegionMap* allow, int64_t chr_length, const string& chr, const Args& args,
                                                        const PopulationCsv* csv_for_hudson) {
  const Interval ext = from_1based_inclusive(std::max<int64_t>(entry.interval.first - 3000000, 0),
                                             std::min<int64_t>(wrap_add(entry.interval.second, 3000000), chr_length));
  const vector<Interval>* allow_chr = nullptr;
  if (allow) { auto it = allow->find(chr); if (it != allow->end()) allow_chr = &it->second; }
  const vector<Interval>* mask_chr = nullptr;
  { auto it = mask.find(chr); if (it != mask.end()) mask_chr = &it->second; }
  vector<const Variant*> unf, fil;
  // the variants are sorted by position (process_vcf): only the run inside the region is visited
  const int64_t lo_pos = std::max(ext.first, entry.interval.first);
  size_t first = (size_t)(std::lower_bound(vcf.variants.begin(), vcf.variants.end(), lo_pos, [](const Variant& v, int64_t p) { return v.position < p; }) - vcf.variants.begin());
  for (size_t i = first; i < vcf.variants.size(); ++i) {
    const Variant& v = vcf.variants[i];
    if ((uint64_t)v.position >= (uint64_t)ext.second || (uint64_t)v.position >= (uint64_t)entry.interval.second) break;  // hal_contains' unsigned order; positions ascend
    if (!hal_contains(ext, v.position) || !hal_contains(entry.interval, v.position)) continue;
    if ((!allow_chr || position_in_regions(v.position, *allow_chr)) && (!mask_chr || !position_in_regions(v.position, *mask_chr))) unf.push_back(&v);
    if (vcf.flags[i] == FLAG_PASS) fil.push_back(&v);
  }
  const size_t N = vcf.sample_names.size();
  std::optional<StageTimer> tm;
  tm.emplace("  region:pack_and_upload_matrices");
  // a region large enough to be worth it is split into one site slab per --devices GPU; the communicators are shared, so only one
  // such region is in flight at a time (small regions keep flowing through the other workers)
  const bool shard = want_shard(std::max(unf.size(), fil.size()), N);
  std::unique_lock<std::mutex> shard_lock(g_shard.region_mutex, std::defer_lock);
  if (shard) { shard_lock.lock(); restore_shard_group(); }
  // (want_shard is asked again: restore_shard_group may have had to give the communicators up)
  RegionMatrix m_unf = build_matrix(unf, N, args.device, shard && want_shard(unf.size(), N));
  RegionMatrix m_fil = fil == unf ? m_unf : build_matrix(fil, N, args.device, shard && want_shard(fil.size(), N));  // same variants -> one matrix in HBM
  tm.emplace("  region:gpu_sweeps_and_host_statistics");

  WcRegion wc;
  vector<vector<string>> wc_rows;
  if (args.enable_fst) {
    wc = wc_haplotype_groups(fil, m_fil, vcf.sample_names, entry.samples_filtered, args.device);
    if (!args.fst_populations.empty()) {  // process.rs:2769-2802: the CSV is re-read here, without exclusions
      try { wc_rows = wc_csv_population_rows(fil, m_fil, vcf.sample_names, parse_population_csv(args.fst_populations), entry, args.device); }
      catch (const Error& e) { logmsg("ERROR", string("Error calculating population FST: ") + e.what()); }
    }
  }

  const int64_t sequence_length = (int64_t)((uint64_t)entry.interval.second - (uint64_t)entry.interval.first);
  const int64_t adj = adjusted_sequence_length(entry.interval.first + 1, entry.interval.second, allow_chr, mask_chr);
  const double callable = sequence_length > 0 ? (double)adj / (double)sequence_length : NAN;
  const double masked_fraction = 1.0 - callable;
  if (!std::isfinite(callable) || masked_fraction >= 0.99) {
    logmsg("WARN", "DROPPED: Region " + entry.seqname + ":" + std::to_string(entry.interval.first) + "-" + std::to_string(entry.interval.second) + " is >= 99% masked");
    return std::nullopt;
  }
  const int64_t fil_adj = adj;  // FilteringStats.filtered_positions is never filled for the slice (process.rs:2568-2571, 2677-2708)

  GroupStats sf[2], su[2];
  // the filtered pair's sweep also carries the Hudson pair of haplotype groups 0 / 1 when --fst will ask for it (same matrix, same masks)
  PairSweep fil_pair;
  bool want_hudson_pair = false;
  if (args.enable_fst && !fil.empty() && (uint64_t)entry.interval.second > (uint64_t)entry.interval.first && fil_adj > 0) {
    const auto index0 = map_sample_names_to_indices(vcf.sample_names);
    want_hudson_pair = haplotypes_for_group(0, entry.samples_filtered, index0).size() >= 2 && haplotypes_for_group(1, entry.samples_filtered, index0).size() >= 2;
  }
  process_variants_pair(fil, m_fil, vcf.sample_names, entry.samples_filtered, entry.interval, fil_adj, mask_chr, args.device, sf, want_hudson_pair ? &fil_pair : nullptr);
  process_variants_pair(unf, m_unf, vcf.sample_names, entry.samples_unfiltered, entry.interval, adj, mask_chr, args.device, su);
  if (!sf[0].present && !sf[1].present && !su[0].present && !su[1].present) return std::nullopt;

  const double inv_f = inversion_allele_frequency(entry.samples_filtered).value_or(-1.0);
  const double inv_u = inversion_allele_frequency(entry.samples_unfiltered).value_or(-1.0);

  RegionOutput out;
  out.seqname = entry.seqname;
  out.region_start1 = entry.interval.first + 1;
  out.region_end1 = entry.interval.second;
  HudsonRegion hud;
  if (args.enable_fst) {
    const auto index = map_sample_names_to_indices(vcf.sample_names);
    const HapList h0 = haplotypes_for_group(0, entry.samples_filtered, index), h1 = haplotypes_for_group(1, entry.samples_filtered, index);
    const bool region_valid = (uint64_t)entry.interval.second > (uint64_t)entry.interval.first;
    if (h0.size() >= 2 && h1.size() >= 2 && region_valid) {
      const double pis[2] = {sf[0].pi, sf[1].pi};
      hud = hudson_groups(fil, m_fil, vcf.sample_names, h0, h1, fil_adj, pis, args.device, want_hudson_pair ? &fil_pair : nullptr);
      if (hud.have_outcome) {
        out.hudson_rows.push_back({entry.seqname, std::to_string(entry.interval.first), std::to_string(entry.interval.second - 1),
                                   "HaplotypeGroup", "0", "HaplotypeGroup", "1", fmt_opt(hud.dxy), fmt_opt(hud.pi0), fmt_opt(hud.pi1),
                                   fmt_opt(hud.avg), fmt_opt(hud.fst)});
        out.hudson_sites = hud.sites;
      }
    }
  }
  if (args.enable_fst && csv_for_hudson) {  // process.rs:3301-3392
    const auto index = map_sample_names_to_indices(vcf.sample_names);
    std::map<string, HapList> pop_haps;
    for (auto& kv : *csv_for_hudson) {
      HapList hl;
      for (auto& sid : kv.second) { auto it = index.find(sid); if (it != index.end()) { hl.push_back({it->second, 0}); hl.push_back({it->second, 1}); } }
      if (!hl.empty()) pop_haps[kv.first] = hl;
    }
    // pi of every population present (calculate_pi_for_population), eight populations per sweep
    std::map<string, double> pop_pi;
    vector<string> names;
    for (auto& kv : pop_haps) names.push_back(kv.first);
    if (!fil.empty()) {
      const bool dense_arm = m_fil.has_dense && m_fil.ploidy == 2;
      for (size_t b = 0; b < names.size(); b += FMH_MAX_GROUPS) {
        const size_t cnt = std::min<size_t>(FMH_MAX_GROUPS, names.size() - b);
        vector<vector<uint8_t>> masks;
        for (size_t i = 0; i < cnt; ++i) masks.push_back(mask_of(pop_haps[names[b + i]], N, m_fil.dm->ploidy, true));
        const vector<fmh_pop_totals> tot = sharded_summaries(m_fil, masks, dense_arm ? FMH_FORMULA_DENSE : FMH_FORMULA_SPARSE);
        for (size_t i = 0; i < cnt; ++i) pop_pi[names[b + i]] = pi_from_totals(m_fil, pop_haps[names[b + i]], masks[i], N, fil_adj, tot[i]);
      }
    } else {
      for (auto& n : names) pop_pi[n] = pop_haps[n].size() <= 1 ? NAN : (fil_adj < 0 ? 0.0 : (fil_adj == 0 ? INFINITY : 0.0 / (double)fil_adj));
    }
    const bool region_valid = (uint64_t)entry.interval.second > (uint64_t)entry.interval.first;
    for (size_t i = 0; i < names.size(); ++i)
      for (size_t j = i + 1; j < names.size(); ++j) {
        const HapList &ha = pop_haps[names[i]], &hb = pop_haps[names[j]];
        if (ha.size() < 2 || hb.size() < 2 || !region_valid) continue;
        const double pis[2] = {pop_pi[names[i]], pop_pi[names[j]]};
        HudsonRegion h = hudson_groups(fil, m_fil, vcf.sample_names, ha, hb, fil_adj, pis, args.device, nullptr, /*want_sites=*/false);  // only the row is written for named pairs
        if (!h.have_outcome) continue;
        out.hudson_rows.push_back({entry.seqname, std::to_string(entry.interval.first), std::to_string(entry.interval.second - 1),
                                   "NamedPopulation", names[i], "NamedPopulation", names[j], fmt_opt(h.dxy), fmt_opt(h.pi0), fmt_opt(h.pi1),
                                   fmt_opt(h.avg), fmt_opt(h.fst)});
      }
  }
  out.wc_rows = wc_rows;
  // CsvRowData (process.rs:3429-3466)
  std::optional<double> hap_a, hap_b;
  std::optional<size_t> hap_sites;
  if (args.enable_fst) { hap_a = wc.sum_a; hap_b = wc.sum_b; hap_sites = wc.sites; }
  else { hap_a = 0.0; hap_b = 0.0; hap_sites = 0; }
  out.csv_row = {entry.seqname, std::to_string(out.region_start1), std::to_string(out.region_end1), std::to_string(sequence_length),
                 std::to_string(sequence_length), std::to_string(adj), std::to_string(adj), std::to_string(su[0].segsites),
                 std::to_string(su[1].segsites), fmt6(su[0].theta), fmt6(su[1].theta), fmt6(su[0].pi), fmt6(su[1].pi),
                 std::to_string(sf[0].segsites), std::to_string(sf[1].segsites), fmt6(sf[0].theta), fmt6(sf[1].theta), fmt6(sf[0].pi),
                 fmt6(sf[1].pi), std::to_string(su[0].n_hap), std::to_string(su[1].n_hap), std::to_string(sf[0].n_hap),
                 std::to_string(sf[1].n_hap), fmt6(inv_u), fmt6(inv_f), fmt_opt(args.enable_fst ? wc.value : std::nullopt),
                 fmt_opt(hap_a), fmt_opt(hap_b), hap_sites ? std::to_string(*hap_sites) : "NA", fmt_opt(hud.fst), fmt_opt(hud.dxy),
                 fmt_opt(hud.pi0), fmt_opt(hud.pi1), fmt_opt(hud.avg)};
  for (auto& s : su[0].sites) out.diversity.push_back({s.pos1, s.pi, s.theta, 0, false});
  for (auto& s : su[1].sites) out.diversity.push_back({s.pos1, s.pi, s.theta, 1, false});
  for (auto& s : sf[0].sites) out.diversity.push_back({s.pos1, s.pi, s.theta, 0, true});
  for (auto& s : sf[1].sites) out.diversity.push_back({s.pos1, s.pi, s.theta, 1, true});
  if (args.enable_fst) out.wc_sites = wc.per_site;
  return out;
}

// resolve_sample_exclusions, run_vcf.rs:24-187
std::set<string> resolve_exclusions(const Args& args, const string& chr, const vector<ConfigEntry>* entries) {
  std::set<string> requested(args.exclude.begin(), args.exclude.end());
  if (requested.empty()) return {};
  std::set<string> vcf_ids, cfg_ids;
  try { for (auto& n : read_sample_names_from_vcf(find_vcf_file(args.vcf_folder, chr))) vcf_ids.insert(n); } catch (const Error&) {}
  if (entries) for (auto& e : *entries) { for (auto& kv : e.samples_unfiltered) cfg_ids.insert(kv.first); for (auto& kv : e.samples_filtered) cfg_ids.insert(kv.first); }
  if (vcf_ids.empty() && cfg_ids.empty()) return requested;
  std::set<string> resolved;
  for (const string& req : requested) {
    const string t = trim(req);
    for (const std::set<string>* ids : {&vcf_ids, &cfg_ids}) {
      if (ids->count(t)) resolved.insert(t);
      else for (auto& s : *ids) if (s.find(t) != string::npos) resolved.insert(s);
    }
  }
  return resolved;
}

void erase_excluded(SampleMap& m, const std::set<string>& ex) {
  m.erase(std::remove_if(m.begin(), m.end(), [&](const auto& kv) { return ex.count(kv.first) > 0; }), m.end());
}

// --devices with more than one entry: one communicator per listed GPU (fmh_comm_init_all: RCCL over xGMI; the in-process host
// rendezvous when a GPU is listed twice), released when run() ends
struct ShardSetup {
  explicit ShardSetup(const Args& args) {
    if (args.devices.size() < 2 || args.ingest_only) return;
    if (const char* e = getenv("FERROMIC_SHARD_MIN_BYTES")) g_shard.min_bytes = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("FERROMIC_INJECT_SLAB_FAILURE")) {
      g_shard.inject_slab = atoi(e);
      if (const char* c = strchr(e, ':')) g_shard.inject_countdown = atoi(c + 1);
    }
    g_shard.devices = args.devices;
    g_shard.comms.assign(args.devices.size(), nullptr);
    const int rc = fmh_comm_init_all(g_shard.devices.data(), (int)g_shard.devices.size(), g_shard.comms.data());
    if (rc != FMH_OK) {  // no RCCL on this host: regions stay whole (the worker queue still spreads them over the GPUs)
      logmsg("WARN", string("large regions will not be split across --devices: ") + fmh_last_error());
      g_shard.comms.clear();
    }
  }
  ~ShardSetup() {
    for (fmh_comm* c : g_shard.comms) fmh_comm_destroy(c);
    g_shard.comms.clear();
  }
};

// HIP start-up (context, code objects of the library, pinned staging of the upload path, the first sweep's buffers) costs 0.15-0.2 s per
// process; a helper thread pays it with a throw-away 64-site matrix per GPU while the main thread reads the VCF text, and is joined before
// the first region touches a GPU.  Failures are left for the real calls to report.
struct DeviceWarmup {
  std::thread t;
  static bool enabled(const Args& args) { return !args.ingest_only && !getenv("FERROMIC_NO_WARMUP"); }
  static bool checks_gpu(const Args& args) { return enabled(args) && args.devices.size() < 2; }  // several GPUs: main() checks before the communicators
  explicit DeviceWarmup(const Args& args) {
    if (!enabled(args)) return;
    const bool check_first = checks_gpu(args);
    vector<int> devices = args.devices.empty() ? vector<int>{args.device} : args.devices;
    std::sort(devices.begin(), devices.end());
    devices.erase(std::unique(devices.begin(), devices.end()), devices.end());
    t = std::thread([devices, check_first] {
      StageTimer tw("  (helper thread) hip_start_up");
      if (check_first) {
        int n = 0;
        if (fmh_device_count(&n) != FMH_OK) {
          g_gpu_check.message = string("GPU required (run_vcf has no CPU fallback): ") + fmh_last_error();
          g_gpu_check.state.store(2, std::memory_order_release);
          return;
        }
        g_gpu_check.state.store(1, std::memory_order_release);
      }
      const vector<uint8_t> rows(64 * 4, 1), mask(4, 1);
      for (int d : devices) {
        fmh_matrix* m = nullptr;
        fmh_groups* g = nullptr;
        fmh_pop_totals totals;
        if (fmh_matrix_create(rows.data(), nullptr, 64, 2, 2, 1, d, &m) == FMH_OK && fmh_groups_create(m, mask.data(), 1, &g) == FMH_OK)
          (void)fmh_population_summaries(m, g, 0, 64, 0, nullptr, nullptr, &totals, nullptr);
        if (g) fmh_groups_destroy(g);
        if (m) fmh_matrix_destroy(m);
      }
    });
  }
  void join() { if (t.joinable()) t.join(); throw_if_no_gpu(); }
  ~DeviceWarmup() { if (t.joinable()) t.join(); }
};

int run(const Args& args) {
  ShardSetup shard_setup(args);
  DeviceWarmup warmup(args);
  std::optional<RegionMap> mask_regions, allow_regions;
  if (!args.mask_file.empty()) mask_regions = parse_regions_file(args.mask_file);
  if (!args.allow_file.empty()) allow_regions = parse_regions_file(args.allow_file);
  vector<ConfigEntry> entries;
  std::set<string> exclusion;
  if (!args.config_file.empty()) {
    entries = parse_config_file(args.config_file);
    if (!entries.empty()) exclusion = resolve_exclusions(args, entries[0].seqname, &entries);
    else exclusion = std::set<string>(args.exclude.begin(), args.exclude.end());
    for (auto& e : entries) { erase_excluded(e.samples_unfiltered, exclusion); erase_excluded(e.samples_filtered, exclusion); }
  } else if (!args.chr.empty()) {
    exclusion = resolve_exclusions(args, args.chr, nullptr);
    ConfigEntry e;
    e.seqname = args.chr;
    e.interval = args.region.empty() ? from_1based_inclusive(1, INT64_MAX) : parse_region(args.region);
    for (auto& n : read_sample_names_from_vcf(find_vcf_file(args.vcf_folder, args.chr)))
      if (!exclusion.count(n)) { sample_map_set(e.samples_unfiltered, n, 0, 0); sample_map_set(e.samples_filtered, n, 0, 0); }
    if (e.samples_unfiltered.empty()) throw Error("Parse(\"No samples remain after applying exclusions\")");
    entries.push_back(e);
  } else {
    throw Error("Parse(\"Either --config_file or --chr must be specified\")");
  }
  if (args.enable_pca) logmsg("WARN", "--pca is outside the accelerated path and is ignored (DESIGN.md section 8)");
  std::optional<PopulationCsv> csv_for_hudson;  // process.rs:1394-1426: parsed once, exclusions removed
  if (args.enable_fst && !args.fst_populations.empty()) {
    try {
      csv_for_hudson = parse_population_csv(args.fst_populations);
      for (auto& kv : *csv_for_hudson) kv.second.erase(std::remove_if(kv.second.begin(), kv.second.end(), [&](const string& x) { return exclusion.count(x) > 0; }), kv.second.end());
    } catch (const Error& e) { logmsg("ERROR", string("Failed to parse population CSV: ") + e.what()); }
  }

  const string out_dir = dirname_of(args.output_file);
  mkdirs(out_dir);
  const string div_path = out_dir + "/per_site_diversity_output.falsta.gz", fst_path = out_dir + "/per_site_fst_output.falsta.gz";
  const string hudson_path = out_dir + "/hudson_fst_results.tsv.gz";
  remove(div_path.c_str());
  remove(fst_path.c_str());
  std::ofstream csv(args.output_file);
  if (!csv) throw Error("cannot create " + args.output_file);
  { vector<string> h(kCsvHeader, kCsvHeader + 34); csv << join(h, ',', true) << "\n"; csv.flush(); }

  std::map<string, vector<const ConfigEntry*>> by_chr;
  for (auto& e : entries) by_chr[e.seqname].push_back(&e);
  vector<vector<string>> hudson_rows, wc_rows;
  const string wc_path = out_dir + "/wc_fst_results.tsv.gz";
  for (auto& kv : by_chr) {
    const string& chr = kv.first;
    try {
      std::optional<StageTimer> tm;
      tm.emplace("reference_fasta");
      const string ref_seq = read_reference_sequence(args.reference, chr);
      const int64_t chr_length = (int64_t)ref_seq.size();
      RegionMap final_mask = mask_regions ? *mask_regions : RegionMap();
      auto& chr_mask = final_mask[chr];
      for (auto& n : find_n_regions(ref_seq)) chr_mask.push_back(n);
      if (!file_exists(args.gtf)) throw Error("cannot open GTF " + args.gtf);
      string vcf_path;
      try { vcf_path = find_vcf_file(args.vcf_folder, chr); }
      catch (const Error& e) { logmsg("ERROR", "Error finding VCF file for chr" + chr + ": " + e.what()); continue; }
      vector<Interval> hulls;
      for (auto* e : kv.second)
        hulls.push_back({std::max<int64_t>(e->interval.first - 3000000, 0), std::min<int64_t>(wrap_add(e->interval.second, 3000000), chr_length)});
      VcfData vcf;
      tm.emplace("vcf_ingest");
      try { vcf = process_vcf(vcf_path, chr, merge_intervals(hulls), args.min_gq, &final_mask, allow_regions ? &*allow_regions : nullptr, exclusion); }
      catch (const Error& e) { logmsg("ERROR", "Error processing VCF for " + chr + ": " + e.what()); continue; }
      if (args.ingest_only) {
        tm.reset();
        // FNV-1a over (position, flags, stride, genotype bytes) of every variant in order: lets a CPU-only test pin the
        // whole text -> Variant stage against the oracle's parse
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint8_t b) { h ^= b; h *= 1099511628211ull; };
        for (size_t i = 0; i < vcf.variants.size(); ++i) {
          const Variant& v = vcf.variants[i];
          for (int k = 0; k < 8; ++k) mix((uint8_t)((uint64_t)v.position >> (8 * k)));
          mix(vcf.flags[i]);
          mix((uint8_t)v.stride);
          for (uint8_t b : v.data) mix(b);
        }
        if (getenv("FERROMIC_INGEST_DUMP"))  // one line per variant, for diffing against the oracle's parse
          for (size_t i = 0; i < vcf.variants.size(); ++i) {
            const Variant& v = vcf.variants[i];
            printf("[VARIANT] %lld %u %zu ", (long long)v.position, (unsigned)vcf.flags[i], (size_t)v.stride);
            for (uint8_t b : v.data) printf("%02x", b);
            printf("\n");
          }
        printf("[INGEST] chr %s: %zu variants x %zu samples digest %016llx\n", chr.c_str(), vcf.variants.size(), vcf.sample_names.size(),
               (unsigned long long)h);
        continue;
      }
      warmup.join();
      tm.emplace("regions_statistics_and_writers");
      // Regions are independent units: one worker per GPU pulls the next config entry; rows and
      // tracks are emitted in config order whatever the completion order, so the files match a 1-GPU run.
      const vector<const ConfigEntry*>& todo = kv.second;
      vector<std::optional<RegionOutput>> done(todo.size());
      vector<char> finished(todo.size(), 0);
      std::atomic<size_t> next{0};
      std::mutex emit_mutex;
      size_t emitted = 0;
      auto emit_ready = [&]() {  // caller holds emit_mutex
        while (emitted < todo.size() && finished[emitted]) {
          std::optional<RegionOutput>& res = done[emitted];
          if (res) {
            csv << join(res->csv_row, ',', true) << "\n";
            append_members(div_path, res->diversity_members);
            append_members(fst_path, res->fst_members);
            for (auto& r : res->hudson_rows) hudson_rows.push_back(r);
            for (auto& r : res->wc_rows) wc_rows.push_back(r);
            res.reset();
          }
          ++emitted;
        }
      };
      auto worker = [&](int device) {
        Args mine = args;
        mine.device = device;
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= todo.size()) break;
          std::optional<RegionOutput> res;
          try { res = process_single_config_entry(*todo[i], vcf, final_mask, allow_regions ? &*allow_regions : nullptr, chr_length, chr, mine,
                                                    csv_for_hudson ? &*csv_for_hudson : nullptr); }
          catch (const std::exception& err) { logmsg("ERROR", string("DROPPED: Error processing region: ") + err.what()); }
          if (res) try {  // tracks are formatted and deflated here, outside the ordered emit: only the file appends are serial
            vector<vector<string>> members = compress_tracks({diversity_tracks(*res), fst_tracks(*res)}, (size_t)std::max<int64_t>(res->region_end1 - res->region_start1 + 1, 1));
            res->diversity_members = std::move(members[0]);
            res->fst_members = std::move(members[1]);
            res->diversity.clear(); res->diversity.shrink_to_fit();
            res->wc_sites.clear(); res->wc_sites.shrink_to_fit();
            res->hudson_sites.clear(); res->hudson_sites.shrink_to_fit();
          } catch (const std::exception& err) { logmsg("ERROR", string("DROPPED: Error writing the tracks of a region: ") + err.what()); res.reset(); }
          std::lock_guard<std::mutex> lock(emit_mutex);
          done[i] = std::move(res);
          finished[i] = 1;
          emit_ready();
        }
      };
      vector<int> worker_devices;
      // Per region the host side (packing, track formatting and deflate: several ms of CPU for a few thousand sites) outweighs the GPU side
      // (~1 ms of latencies).  The tracks go through the shared pool, so a region worker mostly overlaps one region's GPU latencies with
      // another's host work: half the process's CPUs, shared between the GPUs, at least 4 and at most 8 per GPU (16 measured 10 % behind 8).
      const int n_gpus = (int)std::max<size_t>(args.devices.size(), 1);
      const int workers_per_device = args.workers_per_device > 0 ? args.workers_per_device : std::max(4, std::min(8, (int)fmh_host::usable_cpus() / 2 / n_gpus));
      for (int d : args.devices.empty() ? vector<int>{args.device} : args.devices)
        for (int k = 0; k < workers_per_device; ++k) worker_devices.push_back(d);
      if (worker_devices.size() > todo.size()) worker_devices.resize(std::max<size_t>(todo.size(), 1));
      g_region_workers.store((unsigned)std::max<size_t>(worker_devices.size(), 1));
      if (worker_devices.size() <= 1) {
        worker(worker_devices.empty() ? args.device : worker_devices[0]);
      } else {
        vector<std::thread> pool;
        for (int d : worker_devices) pool.emplace_back(worker, d);
        for (auto& t : pool) t.join();
      }
    } catch (const Error& e) {
      fprintf(stderr, "Error processing chromosome %s: %s\n", chr.c_str(), e.what());
      continue;
    }
  }
  warmup.join();  // a run whose chromosomes were all skipped still reports a missing GPU
  csv.flush();
  if (args.enable_fst) {  // final rewrite with header (process.rs:1557-1625)
    remove(hudson_path.c_str());
    string text = kHudsonTsvHeader;
    for (auto& r : hudson_rows) text += join(r, '\t') + "\n";
    gz_append(hudson_path, text);
    if (!wc_rows.empty()) {  // process.rs:1628-1726
      remove(wc_path.c_str());
      string wt = kWcTsvHeader;
      for (auto& r : wc_rows) wt += join(r, '\t') + "\n";
      gz_append(wc_path, wt);
    }
  }
  printf("Wrote FASTA-style per-site diversity data to per_site_diversity_output.falsta.gz\n");
  printf("Wrote FASTA-style per-site FST data to per_site_fst_output.falsta.gz\n");
  printf("Processing complete. Check the output file: \"%s\"\n", args.output_file.c_str());
  return 0;
}

Args parse_args(int argc, char** argv) {  // clap Args, process.rs:67-144
  Args a;
  bool have_vcf = false, have_ref = false, have_gtf = false;
  for (int i = 1; i < argc; ++i) {
    string k = argv[i], v;
    const size_t eq = k.find('=');
    bool has_inline = false;
    if (starts_with(k, "--") && eq != string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); has_inline = true; }
    auto value = [&]() -> string {
      if (has_inline) return v;
      if (i + 1 >= argc) throw Error("missing value for " + k);
      return argv[++i];
    };
    if (k == "--vcf_folder" || k == "-v") { a.vcf_folder = value(); have_vcf = true; }
    else if (k == "--chr" || k == "-c") a.chr = value();
    else if (k == "--region" || k == "-r") a.region = value();
    else if (k == "--config_file") a.config_file = value();
    else if (k == "--output_file" || k == "-o") a.output_file = value();
    else if (k == "--min_gq") { unsigned g; if (!parse_unsigned(value(), 65535, &g)) throw Error("invalid --min_gq"); a.min_gq = g; }
    else if (k == "--mask_file") a.mask_file = value();
    else if (k == "--allow_file") a.allow_file = value();
    else if (k == "--exclude") { for (auto& s : split(value(), ',')) if (!s.empty()) a.exclude.push_back(s); }
    else if (k == "--reference") { a.reference = value(); have_ref = true; }
    else if (k == "--gtf") { a.gtf = value(); have_gtf = true; }
    else if (k == "--pca") a.enable_pca = true;
    else if (k == "--pca_components" || k == "--pca_output") (void)value();
    else if (k == "--fst") a.enable_fst = true;
    else if (k == "--fst_populations") a.fst_populations = value();
    else if (k == "--ingest_only") a.ingest_only = true;
    else if (k == "--print_formats") { a.print_formats = true; return a; }
    else if (k == "--bench_tracks") {  // [variants [region length]]
      const int v = i + 1 < argc && argv[i + 1][0] != '-' ? atoi(argv[i + 1]) : 120;
      exit(bench_tracks(v, i + 2 < argc && argv[i + 1][0] != '-' && argv[i + 2][0] != '-' ? atoi(argv[i + 2]) : 15000));
    }
    else if (k == "--dump_writer_cases") { exit(dump_writer_cases(i + 1 < argc ? argv[i + 1] : "writer_cases", i + 2 < argc ? atoi(argv[i + 2]) : 60)); }
    else if (k == "--check_writers" || k == "--check_fmt6") { exit(check_fmt6(i + 1 < argc ? (size_t)atoll(argv[i + 1]) : 1000000)); }
    else if (k == "--workers_per_device") a.workers_per_device = std::max(1, atoi(value().c_str()));
    else if (k == "--device") a.device = atoi(value().c_str());
    else if (k == "--devices") {  // "4" = devices 0..3, "0,2,5" = those devices (one worker thread each)
      const string v2 = value();
      if (v2.find(',') == string::npos) { for (int d = 0; d < atoi(v2.c_str()); ++d) a.devices.push_back(d); }
      else for (auto& t : split(v2, ',')) if (!t.empty()) a.devices.push_back(atoi(t.c_str()));
      if (a.devices.empty()) throw Error("invalid --devices");
      if (v2.find(',') == string::npos && atoi(v2.c_str()) < 1) throw Error("invalid --devices");
    }
    else if (k == "--help" || k == "-h") {
      printf("run_vcf --vcf_folder DIR --reference FA --gtf GTF [--config_file TSV | --chr C [--region S-E]] [--output_file CSV]\n"
             "        [--min_gq 30] [--mask_file F] [--allow_file F] [--exclude a,b] [--fst] [--fst_populations CSV] [--device N | --devices N|a,b,c] [--workers_per_device N]\n");
      exit(0);
    } else throw Error("unexpected argument '" + k + "'");
  }
  if (!have_vcf || !have_ref || !have_gtf) throw Error("the following required arguments were not provided: --vcf_folder --reference --gtf");
  if (getenv("FERROMIC_HIP_DEVICES") && a.device == 0) a.device = atoi(getenv("FERROMIC_HIP_DEVICES"));
  return a;
}

}  // namespace

int main(int argc, char** argv) {
  // glibc's allocator hands freed heap tops back to the kernel and grows arenas in small steps; with sixteen parser threads allocating a 5-KB
  // row per VCF line that is a stream of brk / mprotect / madvise calls which serialise on the process's address-space lock against every
  // page fault of every other thread (measured on the 16-CPU GPU box: 2.4 s of system time and 0.33-0.36 s for the 3.5 GB ingest, against
  // 0.4 s and 0.18 s with the two settings below).  The process lives for seconds; it keeps what it has touched.
  mallopt(M_TOP_PAD, 256 << 20);
  mallopt(M_TRIM_THRESHOLD, INT32_MAX);
  if (getenv("FERROMIC_TIMING")) {  // how long the loader took: process start (/proc/self/stat field 22, 10-ms ticks since boot) to main()
    if (FILE* f = fopen("/proc/self/stat", "r")) {
      char buf[1024] = {0};
      const size_t got = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      const char* p = got ? strrchr(buf, ')') : nullptr;
      unsigned long long start_ticks = 0;
      if (p) {
        int field = 2;
        for (const char* q = p + 1; *q && field < 22; ++q) if (*q == ' ') { ++field; if (field == 22) start_ticks = strtoull(q + 1, nullptr, 10); }
      }
      struct timespec ts;
      clock_gettime(CLOCK_BOOTTIME, &ts);
      const double now = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, started = (double)start_ticks / (double)sysconf(_SC_CLK_TCK);
      if (start_ticks) fprintf(stderr, "[TIMING] process_start_to_main %.3f\n", now - started);
    }
  }
  try {
    const Args args = parse_args(argc, argv);
    if (args.print_formats) return print_formats();
    int n = 0;
    if (!args.ingest_only && !DeviceWarmup::checks_gpu(args)) fmh_check(fmh_device_count(&n), "GPU required (run_vcf has no CPU fallback)");
    int rc;
    {
      StageTimer t("run");
      rc = run(args);
    }
    // Every output file is closed by now.  Leaving through exit() would spend 0.1-0.2 s unloading the HIP runtime and running the static
    // destructors of a process that is about to disappear (FERROMIC_FULL_TEARDOWN=1 keeps the long way, for leak checkers).
    fflush(stdout);
    fflush(stderr);
    if (!getenv("FERROMIC_FULL_TEARDOWN")) _exit(rc);
    return rc;
  } catch (const std::exception& e) {
    fprintf(stderr, "Error: %s\n", e.what());
    return 1;
  }
}
