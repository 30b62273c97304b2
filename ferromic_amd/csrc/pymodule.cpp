// ferromic._core — the host side of the drop-in Python module, in C++ (pybind11) over the C-ABI of libferromic_hip.so.
//
// It mirrors the reference's PyO3 module `ferromic` (src/lib.rs:2227-2270) name for name: the same classes,
// functions, keyword names, defaults, read-only attributes, reprs and ValueError texts, for the per-site
// diversity / FST path.  What lives here is what lives in src/lib.rs and in the path-selection parts of
// src/stats.rs: input coercion, the choice of code path (summary / dense / sparse) exactly as the reference
// makes it, column masks, and turning device tracks into result objects.  Every statistic that touches genotype
// data is computed by the HIP kernels behind the C-ABI (include/ferromic_hip.h); there is no CPU fallback —
// without a GPU the calls raise.  The GIL is released around every device call (lib.rs `py.allow_threads`).
#include <pybind11/numpy.h>
#include <emmintrin.h>
#include <sys/mman.h>

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ferromic_hip.h"
#include "host_cpus.hpp"

namespace py = pybind11;
using std::optional;
using std::shared_ptr;
using std::string;
using std::vector;

namespace {

constexpr double kFstEpsilon = 1e-12;  // stats.rs:26
constexpr int kLeft = 0, kRight = 1;

// ---------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------
[[noreturn]] void raise(PyObject* type, const string& msg) {
  PyErr_SetString(type, msg.c_str());
  throw py::error_already_set();
}
[[noreturn]] void value_error(const string& msg) { raise(PyExc_ValueError, msg); }
// vcf_error_to_pyerr, lib.rs:1551: ValueError(f"VCF error: {err:?}")
[[noreturn]] void vcf_error(const string& kind, const string& msg) { value_error("VCF error: " + kind + "(\"" + msg + "\")"); }
[[noreturn]] void pca_unavailable() {
  raise(PyExc_NotImplementedError, "PCA (src/pca.rs) is outside the per-site diversity/FST path implemented by ferromic_amd");
}

void fmh_check(int status) {
  if (status == FMH_OK) return;
  raise(PyExc_RuntimeError, "libferromic_hip status " + std::to_string(status) + ": " + fmh_last_error());
}

int64_t sat_sub(int64_t a, int64_t b) {
  __int128 r = (__int128)a - (__int128)b;
  if (r > std::numeric_limits<int64_t>::max()) return std::numeric_limits<int64_t>::max();
  if (r < std::numeric_limits<int64_t>::min()) return std::numeric_limits<int64_t>::min();
  return (int64_t)r;
}

// ---------------------------------------------------------------------------------------------------------------
// device plumbing (RAII over the C-ABI handles)
// ---------------------------------------------------------------------------------------------------------------
// Which GPU new matrices go to: FERROMIC_HIP_DEVICES (a device index or a comma-separated list whose
// first entry is taken - one process drives one GPU on this path), else device 0; ferromic.set_device() overrides it.
// A matrix stays on the device it was uploaded to, and so do the buffers of every statistic computed from it.
std::atomic<int> g_device{-1};
int current_device() {
  int d = g_device.load();
  if (d >= 0) return d;
  d = 0;
  if (const char* env = getenv("FERROMIC_HIP_DEVICES")) {
    char* end = nullptr;
    const long v = strtol(env, &end, 10);
    if (end == env || v < 0 || v > 63 || (*end != '\0' && *end != ',')) value_error(string("FERROMIC_HIP_DEVICES: cannot read a device index from \"") + env + "\"");
    d = (int)v;
  }
  g_device.store(d);
  // Once per process, with the start-up: the first pageable copy of more than a few KB (and the first device block of its size class) costs
  // 7-8 ms on the GPU box - otherwise paid by the first statistic of whichever cohort is the first large enough.
  for (size_t bytes : {(size_t)64 << 10, (size_t)1 << 20}) {
    void* p = nullptr;
    if (fmh_device_alloc(d, bytes, &p) != FMH_OK) break;
    vector<uint8_t> h(bytes, 0);
    (void)fmh_copy_to_device(d, p, h.data(), bytes, nullptr);
    (void)fmh_copy_to_host(d, h.data(), p, bytes, nullptr);
    (void)fmh_device_free(d, p);
  }
  return d;
}

struct Groups;
struct DevMatrix {
  fmh_matrix* h = nullptr;
  int device = 0;
  size_t variants = 0, samples = 0, ploidy = 0;
  // the last few group-mask sets swept over this matrix, with their device handles: a Population asks for the same masks at every call, and
  // making a handle is a device allocation and a blocking copy (15 us of a 45-us hudson_fst on a small cohort).  Callers hold the GIL.
  mutable vector<std::pair<vector<uint8_t>, shared_ptr<Groups>>> recent_groups;
  ~DevMatrix();
  size_t columns() const { return samples * ploidy; }
};

shared_ptr<DevMatrix> upload_matrix(const uint8_t* data, const uint64_t* missing_words, size_t variants, size_t samples, size_t ploidy,
                                    uint8_t max_allele) {
  auto m = std::make_shared<DevMatrix>();
  m->variants = variants; m->samples = samples; m->ploidy = ploidy;
  m->device = current_device();
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fmh_matrix_create(data, missing_words, variants, samples, ploidy, max_allele, m->device, &m->h);
  }
  fmh_check(rc);
  return m;
}

// missing bitset in the reference host layout (stats.rs:1298-1302): bit (site * stride + column), LSB first
vector<uint64_t> missing_words_from_flags(const uint8_t* missing_flag, size_t total, bool* any) {
  vector<uint64_t> words((total + 63) / 64, 0);
  bool a = false;
  for (size_t i = 0; i < total; ++i)
    if (missing_flag[i]) { words[i >> 6] |= 1ull << (i & 63); a = true; }
  *any = a;
  return words;
}

shared_ptr<DevMatrix> upload_planes(const uint8_t* plane0, const uint8_t* called_or_null, size_t pitch, size_t variants, size_t samples, size_t ploidy,
                                    uint8_t max_allele) {
  auto m = std::make_shared<DevMatrix>();
  m->variants = variants; m->samples = samples; m->ploidy = ploidy;
  m->device = current_device();
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fmh_matrix_create_packed(plane0, nullptr, nullptr, called_or_null, pitch, variants, samples, ploidy, max_allele, m->device, &m->h);
  }
  fmh_check(rc);
  return m;
}

struct Groups {
  fmh_groups* h = nullptr;
  int n = 0;
  Groups(const DevMatrix& m, const vector<vector<uint8_t>>& masks) : n((int)masks.size()) {
    vector<uint8_t> flat;
    flat.reserve(masks.size() * m.columns());
    for (auto& k : masks) flat.insert(flat.end(), k.begin(), k.end());
    fmh_check(fmh_groups_create(m.h, flat.data(), n, &h));
  }
  Groups(const Groups&) = delete;
  ~Groups() { if (h) fmh_groups_destroy(h); }
};
DevMatrix::~DevMatrix() {
  recent_groups.clear();  // the handles go before the matrix they were made for
  if (h) fmh_matrix_destroy(h);
}
// the device handle of `masks` over `m`: one of the last eight asked for, or a new one
shared_ptr<Groups> groups_for(const DevMatrix& m, const vector<vector<uint8_t>>& masks) {
  size_t total = 0;
  for (auto& k : masks) total += k.size();
  vector<uint8_t> flat;
  flat.reserve(total + 1);
  flat.push_back((uint8_t)masks.size());
  for (auto& k : masks) flat.insert(flat.end(), k.begin(), k.end());
  auto& cache = m.recent_groups;
  for (size_t i = 0; i < cache.size(); ++i) {
    if (cache[i].first.size() == flat.size() && memcmp(cache[i].first.data(), flat.data(), flat.size()) == 0) {
      if (i != 0) std::rotate(cache.begin(), cache.begin() + (std::ptrdiff_t)i, cache.begin() + (std::ptrdiff_t)i + 1);  // most recent first
      return cache[0].second;
    }
  }
  auto g = std::make_shared<Groups>(m, masks);
  if (cache.size() >= 8) cache.pop_back();
  cache.insert(cache.begin(), {std::move(flat), g});
  return g;
}

struct DevBuf {
  void* p = nullptr;
  int device = 0;
  DevBuf(int dev, size_t bytes) : device(dev) { fmh_check(fmh_device_alloc(device, std::max<size_t>(bytes, 1), &p)); }
  DevBuf(const DevBuf&) = delete;
  ~DevBuf() { if (p) fmh_device_free(device, p); }
  template <class T> vector<T> fetch(size_t n) const {
    vector<T> out(n);
    if (n) {
      int rc;
      { py::gil_scoped_release nogil; rc = fmh_copy_to_host(device, out.data(), p, n * sizeof(T), nullptr); }
      fmh_check(rc);
    }
    return out;
  }
};

size_t mask_count(const vector<uint8_t>& m) { size_t c = 0; for (uint8_t x : m) c += x; return c; }

// ---------------------------------------------------------------------------------------------------------------
// input coercion (lib.rs:825-1080, 1301-1367)
// ---------------------------------------------------------------------------------------------------------------
bool is_int(const py::handle& o) { return PyLong_Check(o.ptr()) && !PyBool_Check(o.ptr()); }
bool is_np_integer(const py::handle& o) {
  static py::object np_integer = py::module_::import("numpy").attr("integer");
  return py::isinstance(o, np_integer);
}
bool is_intlike(const py::handle& o) { return is_int(o) || is_np_integer(o); }

int64_t to_i64(const py::handle& o) { return py::cast<int64_t>(py::int_(py::reinterpret_borrow<py::object>(o))); }

int extract_u8(const py::handle& o) {
  py::object v = py::reinterpret_borrow<py::object>(o);
  if (PyBool_Check(o.ptr())) v = py::int_(v);
  if (!PyLong_Check(v.ptr()) && !is_np_integer(v)) raise(PyExc_TypeError, "expected an integer allele");
  py::int_ as_int(v);
  int overflow = 0;
  long long x = PyLong_AsLongLongAndOverflow(as_int.ptr(), &overflow);
  if (overflow || x < 0 || x > 255) raise(PyExc_OverflowError, "out of range integral type conversion attempted");
  return (int)x;
}

// extract_optional_field, lib.rs:1369-1379: item access first, then attribute
py::object field(const py::handle& obj, std::initializer_list<const char*> names) {
  for (const char* name : names) {
    PyObject* item = PyObject_GetItem(obj.ptr(), py::str(name).ptr());
    if (item) return py::reinterpret_steal<py::object>(item);
    PyErr_Clear();
    if (PyObject_HasAttrString(obj.ptr(), name)) return obj.attr(name);
  }
  return py::none();
}
py::object mapping_field(const py::dict& d, std::initializer_list<const char*> names) {
  string joined;
  for (const char* name : names) {
    if (d.contains(name)) return d[name];
    joined += (joined.empty() ? "" : " / ") + string(name);
  }
  value_error("mapping missing required field: " + joined);
}

// The genotypes of one variant, flat: sample i is None (len -1) or alleles[off[i] .. off[i] + len[i])
// (a haploid int is a one-allele genotype)
struct ParsedVariant {
  int64_t position = 0;
  vector<int32_t> len;
  vector<uint32_t> off;
  vector<uint8_t> alleles;
};

inline uint8_t allele_from(PyObject* a) {
  if (PyLong_CheckExact(a)) {  // the overwhelmingly common case: a plain int
    int overflow = 0;
    const long v = PyLong_AsLongAndOverflow(a, &overflow);
    if (overflow || v < 0 || v > 255) raise(PyExc_OverflowError, "out of range integral type conversion attempted");
    return (uint8_t)v;
  }
  return (uint8_t)extract_u8(py::handle(a));
}

// parse_genotypes, lib.rs:1301-1332: None | int (haploid) | iterable of ints per sample
void parse_genotypes(const py::handle& obj, ParsedVariant* out) {
  auto add_entry = [&](PyObject* entry) {
    out->off.push_back((uint32_t)out->alleles.size());
    if (entry == Py_None) { out->len.push_back(-1); return; }
    if (PyList_CheckExact(entry) || PyTuple_CheckExact(entry)) {  // fast path: no iterator object per sample
      const Py_ssize_t n = PySequence_Fast_GET_SIZE(entry);
      PyObject** items = PySequence_Fast_ITEMS(entry);
      for (Py_ssize_t k = 0; k < n; ++k) out->alleles.push_back(allele_from(items[k]));
      out->len.push_back((int32_t)n);
      return;
    }
    if (is_intlike(entry)) {
      bool ok = true;
      int v = 0;
      try { v = extract_u8(entry); } catch (py::error_already_set& e) { if (e.matches(PyExc_OverflowError)) ok = false; else throw; }
      if (ok) { out->alleles.push_back((uint8_t)v); out->len.push_back(1); return; }
    }
    PyObject* it = PyObject_GetIter(entry);
    if (!it) { PyErr_Clear(); value_error("genotypes must be sequences of allele integers or None"); }
    py::object iter = py::reinterpret_steal<py::object>(it);
    int32_t n = 0;
    for (;;) {
      PyObject* a = PyIter_Next(iter.ptr());
      if (!a) { if (PyErr_Occurred()) throw py::error_already_set(); break; }
      py::object allele = py::reinterpret_steal<py::object>(a);
      out->alleles.push_back(allele_from(allele.ptr()));
      ++n;
    }
    out->len.push_back(n);
  };
  // Buffer fast path (round 4): a record whose genotypes are ONE numpy integer array - (samples, ploidy), or (samples,) for haploid calls - is read
  // through its buffer instead of one Python object per allele (per_site_diversity fed 100 000 records of 500 genotypes spent 1.2 s in the
  // object-by-object walk).  Same result as that walk: every row a genotype of `ploidy` alleles, a value outside 0..255 the same OverflowError.
  if (py::isinstance<py::array>(obj)) {
    py::array arr = py::reinterpret_borrow<py::array>(obj);
    const char kind = arr.dtype().kind();
    const py::ssize_t item = arr.itemsize();
    if ((kind == 'i' || kind == 'u') && (arr.ndim() == 1 || arr.ndim() == 2) && (item == 1 || item == 2 || item == 4 || item == 8)) {
      const py::ssize_t n = arr.shape(0), ploidy = arr.ndim() == 2 ? arr.shape(1) : 1;
      const py::ssize_t s0 = arr.strides(0), s1 = arr.ndim() == 2 ? arr.strides(1) : 0;
      const char* base = static_cast<const char*>(arr.data());
      out->len.reserve((size_t)n);
      out->off.reserve((size_t)n);
      out->alleles.reserve(out->alleles.size() + (size_t)(n * ploidy));
      for (py::ssize_t i = 0; i < n; ++i) {
        out->off.push_back((uint32_t)out->alleles.size());
        for (py::ssize_t k = 0; k < ploidy; ++k) {
          const char* q = base + i * s0 + k * s1;
          long long v;
          if (kind == 'i') {
            if (item == 1) { int8_t x; memcpy(&x, q, 1); v = x; } else if (item == 2) { int16_t x; memcpy(&x, q, 2); v = x; }
            else if (item == 4) { int32_t x; memcpy(&x, q, 4); v = x; } else { int64_t x; memcpy(&x, q, 8); v = x; }
          } else {
            unsigned long long u;
            if (item == 1) { uint8_t x; memcpy(&x, q, 1); u = x; } else if (item == 2) { uint16_t x; memcpy(&x, q, 2); u = x; }
            else if (item == 4) { uint32_t x; memcpy(&x, q, 4); u = x; } else { uint64_t x; memcpy(&x, q, 8); u = x; }
            v = u > 255 ? 256 : (long long)u;
          }
          if (v < 0 || v > 255) raise(PyExc_OverflowError, "out of range integral type conversion attempted");
          out->alleles.push_back((uint8_t)v);
        }
        out->len.push_back((int32_t)ploidy);
      }
      return;
    }
  }
  if (PyList_CheckExact(obj.ptr()) || PyTuple_CheckExact(obj.ptr())) {
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(obj.ptr());
    PyObject** items = PySequence_Fast_ITEMS(obj.ptr());
    out->len.reserve((size_t)n);
    out->off.reserve((size_t)n);
    for (Py_ssize_t i = 0; i < n; ++i) add_entry(items[i]);
    return;
  }
  for (py::handle entry : py::reinterpret_borrow<py::object>(obj)) add_entry(entry.ptr());
}

// VariantInput::extract, lib.rs:834-873
ParsedVariant parse_variant(const py::handle& obj) {
  ParsedVariant pv;
  if (py::isinstance<py::tuple>(obj)) {
    py::tuple t = py::reinterpret_borrow<py::tuple>(obj);
    if (t.size() != 2) value_error("variant tuples must have length 2: (position, genotypes)");
    pv.position = to_i64(t[0]);
    parse_genotypes(t[1], &pv);
    return pv;
  }
  if (py::isinstance<py::dict>(obj)) {
    py::dict d = py::reinterpret_borrow<py::dict>(obj);
    pv.position = to_i64(mapping_field(d, {"position", "pos", "site"}));
    parse_genotypes(mapping_field(d, {"genotypes", "calls"}), &pv);
    return pv;
  }
  py::object position = field(obj, {"position", "pos", "site"});
  if (position.is_none()) value_error("variant is missing a position");
  py::object genotypes = field(obj, {"genotypes", "calls"});
  if (genotypes.is_none()) value_error("variant is missing genotypes");
  pv.position = to_i64(position);
  parse_genotypes(genotypes, &pv);
  return pv;
}

// parse_side, lib.rs:1334-1367
int parse_side(const py::handle& obj) {
  if (is_intlike(obj)) {
    const int64_t v = to_i64(obj);
    if (v == 0) return kLeft;
    if (v == 1) return kRight;
    value_error("haplotype side must be 0 or 1");
  }
  if (py::isinstance<py::str>(obj)) {
    string lower = py::cast<string>(obj.attr("lower")());
    if (lower == "l" || lower == "left" || lower == "0") return kLeft;
    if (lower == "r" || lower == "right" || lower == "1") return kRight;
    value_error("haplotype side must be one of 0, 1, 'L', 'R', 'left', 'right'");
  }
  value_error("haplotype side must be 0/1 or a left/right string");
}

typedef std::pair<int64_t, int> Hap;  // (sample index, side)

// HaplotypeInput::extract, lib.rs:887-923
Hap parse_haplotype(const py::handle& obj) {
  if (py::isinstance<py::tuple>(obj) || py::isinstance<py::list>(obj)) {
    py::sequence s = py::reinterpret_borrow<py::sequence>(obj);
    if (s.size() < 2) value_error("haplotypes must contain (sample_index, side)");
    const int64_t idx = to_i64(s[0]);
    if (idx < 0) raise(PyExc_OverflowError, "can't convert negative int to unsigned");
    return {idx, parse_side(s[1])};
  }
  py::object index_obj = field(obj, {"sample_index", "sample", "index"});
  if (index_obj.is_none()) value_error("haplotype missing sample index");
  py::object side_obj = field(obj, {"side", "haplotype", "haplotype_side"});
  if (side_obj.is_none()) value_error("haplotype missing side");
  return {to_i64(index_obj), parse_side(side_obj)};
}
vector<Hap> parse_haplotypes(const py::handle& obj) {
  vector<Hap> out;
  for (py::handle h : py::reinterpret_borrow<py::object>(obj)) out.push_back(parse_haplotype(h));
  return out;
}

// PopulationIdInput::extract, lib.rs:928-965
struct PopId { bool is_group = false; int group = 0; string name; };
PopId parse_population_id(const py::handle& obj) {
  PopId id;
  if (py::isinstance<py::dict>(obj)) {
    py::dict d = py::reinterpret_borrow<py::dict>(obj);
    if (d.contains("haplotype_group")) { id.is_group = true; id.group = extract_u8(d["haplotype_group"]); return id; }
    if (d.contains("named")) { id.name = py::cast<string>(py::str(d["named"])); return id; }
    value_error("population id dictionaries must provide 'haplotype_group' or 'named'");
  }
  if (is_intlike(obj)) {
    int overflow = 0;
    long long v = PyLong_AsLongLongAndOverflow(py::int_(py::reinterpret_borrow<py::object>(obj)).ptr(), &overflow);
    if (!overflow && v >= 0 && v <= 255) { id.is_group = true; id.group = (int)v; return id; }
    if (overflow > 0 || v > 255) value_error("haplotype_group ids must be <= 255");
  }
  if (py::isinstance<py::str>(obj)) { id.name = py::cast<string>(obj); return id; }
  value_error("could not interpret population id; pass an int, string, or mapping");
}

// ---------------------------------------------------------------------------------------------------------------
// variant store: the reference's SPARSE model (process.rs:431-536) kept as arrays.  Entry (site, sample, k) is called
// iff the sample's genotype is Some and has more than k alleles (CompressedGenotypes::get, process.rs:479-496: a
// leading 0xFF byte means None, a later 0xFF truncates).
// ---------------------------------------------------------------------------------------------------------------
struct Dense;

struct Store {
  int64_t S = 0, N = 0, P = 1;          // N, P are the padded array extents (N >= 1)
  vector<int64_t> positions;            // [S]
  vector<int64_t> num_samples;          // [S] genotypes.len() of every variant
  // data/called [S][N][P]; built on first use when the store came from from_numpy (the reference builds this
  // sparse copy eagerly, lib.rs:1165-1206; the statistics only touch it off the dense paths)
  mutable vector<uint8_t> data, called;
  mutable std::function<void(vector<uint8_t>&, vector<uint8_t>&)> lazy;
  shared_ptr<Dense> twin;               // dense matrix whose device image equals this store's (nothing missing)
  shared_ptr<const Store> root;         // contiguous row view: the rows live in root's device matrix from row0 on
  int64_t row0 = 0;
  mutable shared_ptr<DevMatrix> device;

  void materialise() const {
    if (lazy) { lazy(data, called); lazy = nullptr; }
  }
  int64_t first_sample_count() const { return S ? num_samples[0] : 0; }
  shared_ptr<DevMatrix> device_matrix() const;
  // (device matrix, first row, row count) holding this store's variants
  std::tuple<shared_ptr<DevMatrix>, size_t, size_t> device_rows() const {
    if (root) return {root->device_matrix(), (size_t)row0, (size_t)S};
    return {device_matrix(), 0, (size_t)S};
  }
  // HapMembership::build (stats.rs:1212-1238) as a column mask; sample_count < 0 = no bound beyond the data
  vector<uint8_t> mask_for(const vector<Hap>& haps, int64_t sample_count) const {
    vector<uint8_t> mask((size_t)(N * P), 0);
    const int64_t limit = sample_count < 0 ? N : std::min(sample_count, N);
    for (auto& h : haps) {
      if (h.first >= limit || h.second >= P) continue;
      mask[(size_t)(h.first * P + h.second)] = 1;
    }
    return mask;
  }
};

// DenseGenotypeMatrix built by from_numpy (lib.rs:1208-1224): per-allele missing flags
// byte array without vector's zero fill (a 1 GB matrix is written exactly once, by several threads)
// The narrowed copy of a from_numpy array (contents unspecified after resize).  Buffers of 2 MiB and more are anonymous mappings advised to
// transparent huge pages: first-touching a fresh 33 MB heap block in 4-KiB pages was most of from_numpy's time (8 192 page faults contending
// on the address-space lock under the copy threads; 65 536 x 256 diploid samples: 23 ms for a plain memcpy's worth of work).
struct ByteBuf {
  uint8_t* p = nullptr;
  size_t n = 0, mapped = 0;
  ByteBuf() = default;
  ByteBuf(const ByteBuf&) = delete;
  ByteBuf& operator=(const ByteBuf&) = delete;
  ~ByteBuf() { release(); }
  void release() {
    if (mapped) munmap(p, mapped); else delete[] p;
    p = nullptr; n = 0; mapped = 0;
  }
  void resize(size_t k) {
    release();
    if (k >= ((size_t)2 << 20)) {
      const size_t len = (k + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
      void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, len, MADV_HUGEPAGE);
        p = static_cast<uint8_t*>(m); mapped = len; n = k;
        return;
      }
    }
    p = k ? new uint8_t[k] : nullptr;
    n = k;
  }
  uint8_t* data() { return p; }
  const uint8_t* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void assign(size_t k, uint8_t v) { resize(k); if (k) memset(p, v, k); }
  uint8_t& operator[](size_t i) { return p[i]; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
};

// fn(begin, end, worker) over [0, total) on up to 16 host threads (callers hold the GIL; fn touches no Python object)
template <class F>
void parallel_ranges(size_t total, F fn) {
  const size_t grain = (size_t)4 << 20;
  const unsigned hw = fmh_host::usable_cpus();
  size_t workers = std::min<size_t>(std::min<size_t>(hw, 16), (total + grain - 1) / grain);
  if (workers <= 1) { fn((size_t)0, total, (size_t)0); return; }
  vector<std::thread> pool;
  const size_t step = (total + workers - 1) / workers;
  for (size_t w = 0; w < workers; ++w) {
    const size_t b = std::min(total, w * step), e = std::min(total, b + step);
    pool.emplace_back([=, &fn] { fn(b, e, w); });
  }
  for (auto& t : pool) t.join();
}

struct Dense {
  int64_t variants = 0, samples = 0, ploidy = 0;
  ByteBuf g;             // [S][N][P], negatives stored as 0
  ByteBuf neg;           // empty = nothing missing, else one flag per entry
  // A biallelic int8 / uint8 array is packed to bit planes WHILE it is read (convert_planes below): plane0 = the allele bit, called = the
  // entries that are not negative, rows of `plane_pitch` bytes, column c = bit (c & 7) of byte (c >> 3) - what fmh_matrix_create_packed takes.
  // Then g / neg stay empty until someone asks for bytes (materialize): a 1 GB array is read once and 125 MB are written, instead of a 1 GB
  // narrowed copy that the first statistic packs again.
  ByteBuf plane0, called;
  size_t plane_pitch = 0;
  bool planes_only = false, any_missing = false;
  bool has_missing() const { return planes_only ? any_missing : !neg.empty(); }
  int max_allele = 0;
  shared_ptr<DevMatrix> device;
  shared_ptr<DevMatrix> device_matrix() {
    if (!device) {
      if (planes_only) {
        device = upload_planes(plane0.data(), any_missing ? called.data() : nullptr, plane_pitch, (size_t)variants, (size_t)samples, (size_t)ploidy, (uint8_t)max_allele);
      } else {
        bool any = false;
        vector<uint64_t> words;
        if (!neg.empty()) words = missing_words_from_flags(neg.data(), neg.size(), &any);
        device = upload_matrix(g.data(), any ? words.data() : nullptr, (size_t)variants, (size_t)samples, (size_t)ploidy, (uint8_t)max_allele);
      }
    }
    return device;
  }
  // the byte form (g, neg) of a planes-only matrix, for the rare caller that needs entries on the host (the sparse copy of a population with
  // missing calls)
  void materialize() {
    if (!planes_only || !g.empty() || variants == 0) return;
    const size_t cols = (size_t)(samples * ploidy), total = (size_t)variants * cols;
    g.resize(total);
    if (any_missing) neg.resize(total);
    for (size_t r = 0; r < (size_t)variants; ++r) {
      const uint8_t* p = plane0.data() + r * plane_pitch;
      const uint8_t* c = any_missing ? called.data() + r * plane_pitch : nullptr;
      for (size_t k = 0; k < cols; ++k) {
        g[r * cols + k] = (uint8_t)((p[k >> 3] >> (k & 7)) & 1);
        if (c) neg[r * cols + k] = (uint8_t)(((c[k >> 3] >> (k & 7)) & 1) ^ 1);
      }
    }
  }
  // DenseMembership::build, stats.rs:1252-1284
  vector<uint8_t> mask_for(const vector<Hap>& haps) const {
    vector<uint8_t> mask((size_t)(samples * ploidy), 0);
    for (auto& h : haps) {
      if (h.first >= samples) continue;
      if (h.second == kLeft) mask[(size_t)(h.first * ploidy)] = 1;
      else if (ploidy > 1) mask[(size_t)(h.first * ploidy + 1)] = 1;
    }
    return mask;
  }
};

shared_ptr<DevMatrix> Store::device_matrix() const {
  if (twin) return twin->device_matrix();  // same bytes, same (absent) missing mask: one copy in HBM
  if (!device) {
    materialise();
    const size_t total = (size_t)(S * N * P);
    vector<uint8_t> missing(total);
    bool all = true;
    uint8_t max_allele = 0;
    for (size_t i = 0; i < total; ++i) {
      missing[i] = called[i] ? 0 : 1;
      all = all && called[i];
      max_allele = std::max(max_allele, data[i]);
    }
    bool any = false;
    vector<uint64_t> words;
    if (!all) words = missing_words_from_flags(missing.data(), total, &any);
    device = upload_matrix(data.data(), any ? words.data() : nullptr, (size_t)S, (size_t)N, (size_t)P, max_allele);
  }
  return device;
}

// _Store.from_python: variants as Python records
shared_ptr<Store> store_from_python(const py::handle& variants) {
  vector<ParsedVariant> parsed;
  for (py::handle v : py::reinterpret_borrow<py::object>(variants)) parsed.push_back(parse_variant(v));
  auto st = std::make_shared<Store>();
  st->S = (int64_t)parsed.size();
  int64_t N = 0, P = 1;
  for (auto& pv : parsed) {
    N = std::max<int64_t>(N, (int64_t)pv.len.size());
    for (int32_t l : pv.len) P = std::max<int64_t>(P, l);
  }
  st->N = std::max<int64_t>(N, 1);
  st->P = P;
  st->data.assign((size_t)(st->S * st->N * st->P), 0);
  st->called.assign(st->data.size(), 0);
  st->positions.resize(parsed.size());
  st->num_samples.resize(parsed.size());
  for (size_t s = 0; s < parsed.size(); ++s) {
    const ParsedVariant& pv = parsed[s];
    st->positions[s] = pv.position;
    st->num_samples[s] = (int64_t)pv.len.size();
    for (size_t i = 0; i < pv.len.size(); ++i) {
      if (pv.len[i] < 0) continue;
      // CompressedGenotypes::new + get: 0xFF is the missing sentinel: leading -> None, later -> truncation
      const uint8_t* al = pv.alleles.data() + pv.off[i];
      const size_t base = (s * (size_t)st->N + i) * (size_t)st->P;
      for (int32_t k = 0; k < pv.len[i]; ++k) {
        if (al[k] == 0xFF) break;
        st->data[base + (size_t)k] = al[k];
        st->called[base + (size_t)k] = 1;
      }
    }
  }
  return st;
}

// rows `idx` (ascending).  A contiguous run becomes a VIEW: no host copy, and its sweeps run over a row range of the
// root's resident device matrix instead of uploading the rows again.
shared_ptr<const Store> store_subset(const shared_ptr<const Store>& st, const vector<int64_t>& idx) {
  auto out = std::make_shared<Store>();
  out->N = st->N; out->P = st->P;
  out->S = (int64_t)idx.size();
  out->positions.reserve(idx.size());
  out->num_samples.reserve(idx.size());
  for (int64_t r : idx) { out->positions.push_back(st->positions[(size_t)r]); out->num_samples.push_back(st->num_samples[(size_t)r]); }
  const size_t row_bytes = (size_t)(st->N * st->P);
  if (!idx.empty() && idx.back() - idx.front() + 1 == (int64_t)idx.size()) {
    const int64_t a = idx.front();
    out->root = st->root ? st->root : st;
    out->row0 = st->row0 + a;
    shared_ptr<const Store> src = st;
    const size_t count = idx.size();
    out->lazy = [src, a, count, row_bytes](vector<uint8_t>& d, vector<uint8_t>& c) {
      src->materialise();
      d.assign(src->data.begin() + (size_t)a * row_bytes, src->data.begin() + ((size_t)a + count) * row_bytes);
      c.assign(src->called.begin() + (size_t)a * row_bytes, src->called.begin() + ((size_t)a + count) * row_bytes);
    };
    return out;
  }
  st->materialise();
  out->data.resize(idx.size() * row_bytes);
  out->called.resize(idx.size() * row_bytes);
  for (size_t i = 0; i < idx.size(); ++i) {
    memcpy(&out->data[i * row_bytes], &st->data[(size_t)idx[i] * row_bytes], row_bytes);
    memcpy(&out->called[i * row_bytes], &st->called[(size_t)idx[i] * row_bytes], row_bytes);
  }
  return out;
}

// extract_positions, lib.rs:1229-1299
vector<int64_t> extract_positions(const py::handle& obj, int64_t expected_len) {
  static const char* kMsg = "positions must be a sequence of integers (NumPy array with dtype int64/int32/uint32/uint64 or an iterable of ints)";
  vector<int64_t> out;
  if (py::isinstance<py::array>(obj)) {
    py::array arr = py::reinterpret_borrow<py::array>(obj);
    const py::dtype dt = arr.dtype();
    const bool ok = arr.ndim() == 1 && (dt.is(py::dtype::of<int64_t>()) || dt.is(py::dtype::of<int32_t>()) || dt.is(py::dtype::of<uint32_t>()) ||
                                       dt.is(py::dtype::of<uint64_t>()));
    if (!ok) value_error(kMsg);
    out.resize((size_t)arr.shape(0));
    if (dt.is(py::dtype::of<uint64_t>())) {
      auto a = py::array_t<uint64_t, py::array::c_style | py::array::forcecast>(arr);
      for (size_t i = 0; i < out.size(); ++i) {
        if (a.data()[i] > (uint64_t)std::numeric_limits<int64_t>::max()) value_error("positions must fit into signed 64-bit integers");
        out[i] = (int64_t)a.data()[i];
      }
    } else {
      auto a = py::array_t<int64_t, py::array::c_style | py::array::forcecast>(arr);
      memcpy(out.data(), a.data(), out.size() * sizeof(int64_t));
    }
  } else {
    try {
      for (py::handle x : py::reinterpret_borrow<py::object>(obj)) out.push_back(to_i64(x));
    } catch (py::error_already_set&) {
      value_error(kMsg);
    }
  }
  if ((int64_t)out.size() != expected_len)
    value_error("positions length " + std::to_string(out.size()) + " does not match variant dimension " + std::to_string(expected_len));
  return out;
}

// build_variants_from_numpy + convert_numeric_array, lib.rs:1082-1227
template <class T>
void convert_block(const py::array& arr, ByteBuf& g, ByteBuf& neg, bool* any_neg, int* max_allele) {
  auto a = py::array_t<T, py::array::c_style | py::array::forcecast>(arr);
  const T* src = a.data();
  const size_t total = (size_t)a.size();
  g.resize(total);
  uint8_t* dst = g.data();
  uint8_t w_max[16] = {0};
  uint8_t w_neg[16] = {0}, w_big[16] = {0};
  // one pass per thread: copy / narrow, byte maximum, and whether anything is negative (uint8 can never be missing,
  // lib.rs:1086-1090) or beyond u8
  parallel_ranges(total, [&](size_t b, size_t e, size_t w) {
    uint8_t mx = 0;
    bool negs = false, big = false;
    if constexpr (std::is_same<T, uint8_t>::value) {
      memcpy(dst + b, src + b, e - b);
      for (size_t i = b; i < e; ++i) mx = src[i] > mx ? src[i] : mx;
    } else if constexpr (std::is_same<T, int8_t>::value) {
      // sixteen entries per step: a negative entry (sign bit) is missing and stored as 0, the rest is copied; byte maximum over the kept values
      size_t i = b;
      __m128i vmax = _mm_setzero_si128(), vneg = _mm_setzero_si128();
      const __m128i zero = _mm_setzero_si128();
      for (; i + 16 <= e; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i isneg = _mm_cmplt_epi8(v, zero);
        const __m128i kept = _mm_andnot_si128(isneg, v);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), kept);
        vmax = _mm_max_epu8(vmax, kept);
        vneg = _mm_or_si128(vneg, isneg);
      }
      alignas(16) uint8_t lanes[16];
      _mm_store_si128(reinterpret_cast<__m128i*>(lanes), vmax);
      for (int k = 0; k < 16; ++k) mx = lanes[k] > mx ? lanes[k] : mx;
      negs = _mm_movemask_epi8(vneg) != 0;
      for (; i < e; ++i) {
        const int8_t v = src[i];
        if (v < 0) { negs = true; dst[i] = 0; } else { dst[i] = (uint8_t)v; mx = dst[i] > mx ? dst[i] : mx; }
      }
    } else {
      size_t i = b;
      if constexpr (sizeof(T) == 2) {
        // sixteen 16-bit entries per step: beyond u8 -> error; negative (int16) -> missing, stored as 0; the rest narrowed (saturating pack)
        __m128i vmax = _mm_setzero_si128(), vneg = _mm_setzero_si128(), vbig = _mm_setzero_si128();
        const __m128i zero = _mm_setzero_si128(), hi = _mm_set1_epi16((short)0xFF00);
        for (; i + 16 <= e; i += 16) {
          const __m128i a0 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), a1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 8));
          __m128i n0 = zero, n1 = zero;
          if constexpr (std::is_signed<T>::value) { n0 = _mm_cmplt_epi16(a0, zero); n1 = _mm_cmplt_epi16(a1, zero); }
          // a high byte on a non-negative entry: the value exceeds 255
          vbig = _mm_or_si128(vbig, _mm_or_si128(_mm_andnot_si128(n0, _mm_and_si128(a0, hi)), _mm_andnot_si128(n1, _mm_and_si128(a1, hi))));
          const __m128i k0 = _mm_andnot_si128(n0, a0), k1 = _mm_andnot_si128(n1, a1);
          const __m128i kept = _mm_packus_epi16(_mm_and_si128(k0, _mm_set1_epi16(0x00FF)), _mm_and_si128(k1, _mm_set1_epi16(0x00FF)));
          _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), kept);
          vmax = _mm_max_epu8(vmax, kept);
          vneg = _mm_or_si128(vneg, _mm_or_si128(n0, n1));
        }
        alignas(16) uint8_t lanes[16];
        _mm_store_si128(reinterpret_cast<__m128i*>(lanes), vmax);
        for (int k = 0; k < 16; ++k) mx = lanes[k] > mx ? lanes[k] : mx;
        negs = _mm_movemask_epi8(vneg) != 0;
        big = _mm_movemask_epi8(_mm_cmpeq_epi8(vbig, zero)) != 0xFFFF;
      }
      for (; i < e; ++i) {
        const T v = src[i];
        if constexpr (sizeof(T) > 1) {
          if ((std::is_signed<T>::value ? (int64_t)v : (int64_t)(uint64_t)v) > 255) { big = true; dst[i] = 0; continue; }
        }
        if (std::is_signed<T>::value && v < 0) { negs = true; dst[i] = 0; }
        else { dst[i] = (uint8_t)v; mx = dst[i] > mx ? dst[i] : mx; }
      }
    }
    w_max[w] = mx; w_neg[w] = negs; w_big[w] = big;
  });
  bool negs = false;
  int mx = 0;
  for (int w = 0; w < 16; ++w) {
    if (w_big[w]) value_error("allele values must be <= 255");
    negs = negs || w_neg[w];
    mx = std::max<int>(mx, w_max[w]);
  }
  if (negs) {
    neg.resize(total);  // every byte is written below
    uint8_t* flags = neg.data();
    parallel_ranges(total, [&](size_t b, size_t e, size_t) {
      size_t i = b;
      if constexpr (std::is_same<T, int8_t>::value) {
        const __m128i zero = _mm_setzero_si128(), one = _mm_set1_epi8(1);
        for (; i + 16 <= e; i += 16)
          _mm_storeu_si128(reinterpret_cast<__m128i*>(flags + i), _mm_and_si128(_mm_cmplt_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), zero), one));
      }
      for (; i < e; ++i) flags[i] = std::is_signed<T>::value && src[i] < 0 ? 1 : 0;
    });
  }
  *any_neg = negs;
  *max_allele = mx;
}

// int8 / uint8 input whose values are all 0 / 1 (or negative = missing, int8): bit planes straight from the array, sixteen entries per
// step - the narrowing pass and the library's host packer (host_pack.hpp: the same movemask bit order) in one read.  Returns false, having
// touched nothing that matters, as soon as any called entry is above 1: the caller then takes the byte route (convert_block).
template <class T>
bool convert_planes(const py::array& arr, Dense& d, size_t rows, size_t cols) {
  static_assert(sizeof(T) == 1, "one-byte entries");
  auto a = py::array_t<T, py::array::c_style | py::array::forcecast>(arr);
  const T* src = a.data();
  const size_t written = ((cols + 15) / 16) * 2;             // two bytes per sixteen columns
  const size_t pitch = ((cols + 7) / 8 + 15) / 16 * 16;      // the device's plane pitch: each plane then goes up in one copy (fmh_matrix_create_packed)
  if (rows == 0 || cols == 0) return false;
  // a look at 64 rows spread over the array first: a multi-allelic cohort shows an allele above 1 there and is spared the wasted pass
  for (size_t k = 0; k < 64; ++k) {
    const T* row = src + ((rows - 1) * k / 63) * cols;
    for (size_t c = 0; c < cols; ++c) if (row[c] > 1) return false;
  }
  d.plane0.resize(rows * pitch);
  d.called.resize(rows * pitch);
  uint8_t* p0 = d.plane0.data();
  uint8_t* pc = d.called.data();
  std::atomic<bool> big{false}, negs{false}, ones{false};
  parallel_ranges(rows * cols, [&](size_t b, size_t e, size_t) {
    // whole rows per worker: [b, e) is an entry range, rounded to rows here (workers get disjoint row ranges)
    const size_t r0 = (b + cols - 1) / cols, r1 = e == rows * cols ? rows : (e + cols - 1) / cols;
    const __m128i zero = _mm_setzero_si128(), hi = _mm_set1_epi8((char)0xFE);
    __m128i vbig = zero, vneg = zero;
    uint32_t any_one = 0;
    for (size_t r = r0; r < r1 && !big.load(std::memory_order_relaxed); ++r) {
      const T* row = src + r * cols;
      uint8_t* o0 = p0 + r * pitch;
      uint8_t* oc = pc + r * pitch;
      for (size_t c = 0; c < cols; c += 16) {
        __m128i v;
        uint32_t valid = 0xFFFFu;
        if (c + 16 <= cols) v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + c));
        else {
          alignas(16) uint8_t tmp[16] = {0};
          memcpy(tmp, row + c, cols - c);
          v = _mm_load_si128(reinterpret_cast<const __m128i*>(tmp));
          valid = (1u << (cols - c)) - 1u;
        }
        __m128i isneg = zero;
        if constexpr (std::is_signed<T>::value) isneg = _mm_cmplt_epi8(v, zero);
        const __m128i kept = _mm_andnot_si128(isneg, v);
        vbig = _mm_or_si128(vbig, _mm_and_si128(kept, hi));
        vneg = _mm_or_si128(vneg, isneg);
        const uint16_t b0 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(kept, 7));
        const uint16_t bc = (uint16_t)(~(uint32_t)_mm_movemask_epi8(isneg) & valid);
        any_one |= b0;
        memcpy(o0 + (c >> 3), &b0, 2);
        memcpy(oc + (c >> 3), &bc, 2);
      }
      if (written < pitch) { memset(o0 + written, 0, pitch - written); memset(oc + written, 0, pitch - written); }  // the row's padding
    }
    if (_mm_movemask_epi8(_mm_cmpeq_epi8(vbig, zero)) != 0xFFFF) big = true;
    if (_mm_movemask_epi8(vneg) != 0) negs = true;
    if (any_one) ones = true;
  });
  if (big) { d.plane0.release(); d.called.release(); return false; }
  d.plane_pitch = pitch;
  d.planes_only = true;
  d.any_missing = negs;
  if (!negs) d.called.release();
  d.max_allele = ones ? 1 : 0;
  return true;
}

std::pair<shared_ptr<Store>, shared_ptr<Dense>> convert_numeric_array(const py::handle& genotypes, const py::handle& positions) {
  static const char* kMsg = "genotypes must be a numpy.ndarray with dtype uint8/int8/uint16/int16 and shape (variants, samples, ploidy)";
  if (!py::isinstance<py::array>(genotypes)) value_error(kMsg);
  py::array arr = py::reinterpret_borrow<py::array>(genotypes);
  const py::dtype dt = arr.dtype();
  const bool u8 = dt.is(py::dtype::of<uint8_t>()), i8 = dt.is(py::dtype::of<int8_t>()), u16 = dt.is(py::dtype::of<uint16_t>()),
             i16 = dt.is(py::dtype::of<int16_t>());
  if (arr.ndim() != 3 || !(u8 || i8 || u16 || i16)) value_error(kMsg);
  const int64_t S = arr.shape(0), N = arr.shape(1), P = arr.shape(2);
  vector<int64_t> pos = extract_positions(positions, S);
  auto dense = std::make_shared<Dense>();
  bool any_neg = false;
  int mx = 0;
  // diploid biallelic one-byte input: straight to bit planes (FERROMIC_NUMPY_BYTES=1, or the library's u8-row kernels forced with
  // FMH_LAYOUT=bytes - the u8 soak -, keep the byte route)
  static const bool bytes_route = [] {
    long long forced = 0;
    (void)fmh_get_option("FMH_LAYOUT", &forced);
    return forced != 0 || getenv("FERROMIC_NUMPY_BYTES") != nullptr;
  }();
  bool planes = false;
  if (!bytes_route && P == 2 && N > 0 && S > 0 && (u8 || i8))
    planes = u8 ? convert_planes<uint8_t>(arr, *dense, (size_t)S, (size_t)(N * P)) : convert_planes<int8_t>(arr, *dense, (size_t)S, (size_t)(N * P));
  if (planes) { any_neg = dense->any_missing; mx = dense->max_allele; }
  else if (u8) convert_block<uint8_t>(arr, dense->g, dense->neg, &any_neg, &mx);
  else if (i8) convert_block<int8_t>(arr, dense->g, dense->neg, &any_neg, &mx);
  else if (u16) convert_block<uint16_t>(arr, dense->g, dense->neg, &any_neg, &mx);
  else convert_block<int16_t>(arr, dense->g, dense->neg, &any_neg, &mx);
  dense->variants = S; dense->samples = N; dense->ploidy = P;
  dense->max_allele = mx;

  auto st = std::make_shared<Store>();
  st->S = S;
  st->positions = std::move(pos);
  st->num_samples.assign((size_t)S, N);
  if (N == 0) {
    st->N = 1; st->P = std::max<int64_t>(P, 1);
    st->data.assign((size_t)(S * st->N * st->P), 0);
    st->called.assign(st->data.size(), 0);
  } else {
    st->N = N; st->P = P;
    shared_ptr<Dense> src = dense;
    // the sparse half of convert_numeric_array (lib.rs:1165-1206): a sample with ANY missing allele is None as a whole;
    // 0xFF keeps its sentinel meaning (allele 255 is indistinguishable from missing)
    st->lazy = [src, S, N, P](vector<uint8_t>& d, vector<uint8_t>& c) {
      src->materialize();  // (a planes-only matrix: its bytes are rebuilt here, once)
      const size_t total = (size_t)(S * N * P);
      d.assign(total, 0);
      c.assign(total, 0);
      for (size_t sn = 0; sn < (size_t)(S * N); ++sn) {
        bool sample_ok = true;
        if (!src->neg.empty()) for (int64_t k = 0; k < P; ++k) if (src->neg[sn * (size_t)P + (size_t)k]) sample_ok = false;
        if (!sample_ok) continue;
        for (int64_t k = 0; k < P; ++k) {
          const uint8_t v = src->g[sn * (size_t)P + (size_t)k];
          if (v == 0xFF) break;
          d[sn * (size_t)P + (size_t)k] = v;
          c[sn * (size_t)P + (size_t)k] = 1;
        }
      }
    };
  }
  shared_ptr<Dense> dense_out = P == 2 ? dense : nullptr;
  if (dense_out && !any_neg && dense->max_allele != 0xFF && N > 0) st->twin = dense_out;  // sparse semantics == dense semantics
  if (!dense_out && N > 0) {
    // no dense matrix in the reference for ploidy != 2 (lib.rs:1208): the store keeps the converted bytes alive through its lazy builder
  }
  return {st, dense_out};
}

}  // namespace

#include "pymodule_stats.inc"
