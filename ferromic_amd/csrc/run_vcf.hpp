// run_vcf.hpp — what the translation units of the run_vcf CLI share: errors, the host thread pool, string / interval / number-format helpers,
// the data model of process.rs:397-536, and the interfaces of the text ingest (vcf_ingest.cpp) and the writers (writers.cpp) as the
// per-region driver (region_driver.cpp) calls them.
#pragma once
#include <malloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <emmintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>
#include <fstream>
#include <functional>
#include <map>
#include <array>
#include <memory>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <sys/stat.h>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/ferromic_hip.h"
#include "host_cpus.hpp"
#include "deflate_runs.hpp"

namespace fmv {

using std::string;
using std::vector;

typedef std::pair<int64_t, int64_t> Interval;  // 0-based half-open unless said otherwise

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// The GPU check (HIP start-up: 0.07-0.2 s) runs on a helper thread under the text ingest; its verdict is polled once per block of VCF text and
// when the helper is joined.  Not an Error: the per-chromosome handlers must not swallow it.
struct NoGpuError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct GpuCheck {
  std::atomic<int> state{0};  // 0 = pending or checked up front, 1 = present, 2 = absent
  string message;
};
inline GpuCheck g_gpu_check;
inline void throw_if_no_gpu() {
  if (g_gpu_check.state.load(std::memory_order_acquire) == 2) throw NoGpuError(g_gpu_check.message);
}

inline void logmsg(const char* level, const string& m) {
  static const bool quiet = getenv("FERROMIC_PROGRESS") && string(getenv("FERROMIC_PROGRESS")) == "0";
  if (!quiet || string(level) != "INFO") fprintf(stderr, "[%s] %s\n", level, m.c_str());
}

// FERROMIC_TIMING=1: stage wall times on stderr as "[TIMING] stage seconds"
struct StageTimer {
  const char* stage;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit StageTimer(const char* s) : stage(s) {}
  ~StageTimer() {
    static const bool on = getenv("FERROMIC_TIMING") && string(getenv("FERROMIC_TIMING")) == "1";
    if (on) fprintf(stderr, "[TIMING] %s %.3f\n", stage, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
};

inline void fmh_check(int status, const char* what) {
  if (status != FMH_OK) throw Error(string(what) + ": " + fmh_last_error());
}

// ---- host thread pool in its simplest form: T short-lived threads per parallel stage -----------------------
inline unsigned worker_threads() {
  static const unsigned n = [] {
    if (const char* e = getenv("FERROMIC_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)std::min(v, 256); }
    return std::max(1u, std::min(fmh_host::usable_cpus(), 64u));  // the process's CPU share, not the machine (host_cpus.hpp)
  }();
  return n;
}
// Persistent workers: parallel stages are entered once per block of VCF text, per matrix and per region's tracks, so
// thread start-up per stage would dominate configs with many small regions.  Callers are the main thread or the
// per-GPU region workers; pool workers never enter parallel_for themselves.
class ThreadPool {
 public:
  explicit ThreadPool(unsigned n) {
    for (unsigned i = 0; i < n; ++i)
      workers_.emplace_back([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lock(m_);
            cv_.wait(lock, [this] { return stop_ || !jobs_.empty(); });
            if (stop_ && jobs_.empty()) return;
            job = std::move(jobs_.front());
            jobs_.pop_front();
          }
          job();
        }
      });
  }
  ~ThreadPool() {
    { std::lock_guard<std::mutex> lock(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void submit(std::function<void()> job) {
    { std::lock_guard<std::mutex> lock(m_); jobs_.push_back(std::move(job)); }
    cv_.notify_one();
  }
 private:
  vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> jobs_;
  bool stop_ = false;
};
inline ThreadPool& thread_pool() {
  static ThreadPool pool(worker_threads());
  return pool;
}
template <class F> void parallel_for(unsigned tasks, F&& fn) {
  if (tasks <= 1) { if (tasks) fn(0u); return; }
  struct Sync { std::mutex m; std::condition_variable cv; unsigned left; std::exception_ptr failure; } sync;
  sync.left = tasks;
  auto run = [&](unsigned t) {
    try { fn(t); } catch (...) { std::lock_guard<std::mutex> lock(sync.m); if (!sync.failure) sync.failure = std::current_exception(); }
    std::lock_guard<std::mutex> lock(sync.m);
    if (--sync.left == 0) sync.cv.notify_all();
  };
  for (unsigned t = 1; t < tasks; ++t) thread_pool().submit([&run, t] { run(t); });
  run(0u);  // the caller takes a share
  std::unique_lock<std::mutex> lock(sync.m);
  sync.cv.wait(lock, [&] { return sync.left == 0; });
  if (sync.failure) std::rethrow_exception(sync.failure);
}

// ---- small string helpers ------------------------------------------------------------------------
inline vector<string> split(const string& s, char d) {
  vector<string> out;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(d, b);
    if (e == string::npos) { out.push_back(s.substr(b)); break; }
    out.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}
inline vector<string> split_ws(const string& s) {
  vector<string> out;
  std::istringstream is(s);
  string t;
  while (is >> t) out.push_back(t);
  return out;
}
inline string trim(const string& s) {
  size_t b = 0, e = s.size();
  while (b < e && isspace((unsigned char)s[b])) ++b;
  while (e > b && isspace((unsigned char)s[e - 1])) --e;
  return s.substr(b, e - b);
}
inline string trim_start_matches(string s, const string& p) {
  while (s.compare(0, p.size(), p) == 0 && !p.empty()) s = s.substr(p.size());
  return s;
}
inline bool ends_with(const string& s, const string& x) { return s.size() >= x.size() && s.compare(s.size() - x.size(), x.size(), x) == 0; }
inline bool starts_with(const string& s, const string& x) { return s.compare(0, x.size(), x) == 0; }
// Rust str::parse::<i64>: optional sign, ASCII digits only (strtoll alone would also take leading whitespace)
inline bool parse_i64(const string& s, int64_t* out) {
  size_t i = (!s.empty() && (s[0] == '+' || s[0] == '-')) ? 1 : 0;
  if (i >= s.size()) return false;
  for (size_t k = i; k < s.size(); ++k)
    if (s[k] < '0' || s[k] > '9') return false;
  char* end = nullptr;
  errno = 0;
  long long v = strtoll(s.c_str(), &end, 10);
  if (errno || *end) return false;
  *out = v;
  return true;
}
// Rust str::parse::<u8/u16>: optional '+', ASCII digits only
inline bool parse_unsigned(const string& s, unsigned max, unsigned* out) {
  size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
  if (i >= s.size()) return false;
  unsigned long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
    if (v > max) return false;
  }
  *out = (unsigned)v;
  return true;
}
inline bool file_exists(const string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
inline bool is_dir(const string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
inline string dirname_of(const string& p) { size_t s = p.find_last_of('/'); return s == string::npos ? "." : (s == 0 ? "/" : p.substr(0, s)); }
inline void mkdirs(const string& p) {
  string cur;
  for (const string& part : split(p, '/')) {
    cur += part + "/";
    if (!part.empty()) mkdir(cur.c_str(), 0777);
  }
}

// ---- interval newtypes (process.rs:146-352) --------------------------------------------------------
inline Interval from_1based_inclusive(int64_t s, int64_t e) {  // -> 0-based half-open (process.rs:193-206)
  int64_t a = s < 1 ? 1 : s;
  int64_t b = e < a ? a : e;
  return {a - 1, b};
}
inline int64_t hal_len(const Interval& iv) { return (uint64_t)iv.second > (uint64_t)iv.first ? (int64_t)((uint64_t)iv.second - (uint64_t)iv.first) : 0; }
inline bool hal_contains(const Interval& iv, int64_t pos) { return (uint64_t)pos >= (uint64_t)iv.first && (uint64_t)pos < (uint64_t)iv.second; }
inline bool position_in_regions(int64_t pos, const vector<Interval>& r) {  // process.rs:738-744
  for (auto& iv : r) if (pos >= iv.first && pos < iv.second) return true;
  return false;
}
inline int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }  // release-build i64 wrap

// ---- number formatting: Rust `{:.6}` ---------------------------------------------------------------
inline string fmt6_printf(double x) {
  if (std::isnan(x)) return "NaN";
  if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
  char buf[64];
  snprintf(buf, sizeof buf, "%.6f", x);
  return buf;
}
// The same text without printf: a region's tracks are tens of thousands of these (a dense 15-kb region: 65 000, 13 ms of snprintf).
// |x| = m * 2^e exactly (m < 2^53), so |x| * 10^6 = (m * 10^6) / 2^-e is a 73-bit integer over a power of two: quotient and remainder are
// exact, the quotient is rounded half to even on the remainder - the decimal expansion of the binary value, correctly rounded, which is
// what both printf's %.6f and Rust's {:.6} print.  Values of 10^15 and beyond, NaN and infinities take the printf path.
inline void fmt6_append(string& out, double x) {
  if (!(std::fabs(x) < 1e15)) { out += fmt6_printf(x); return; }  // also NaN
  uint64_t bits;
  memcpy(&bits, &x, 8);
  const bool neg = (bits >> 63) != 0;
  const int be = (int)((bits >> 52) & 0x7FF);
  uint64_t m = bits & ((1ull << 52) - 1);
  int e;  // |x| = m * 2^e
  if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }
  unsigned __int128 q;
  if (e >= 0) {
    q = ((unsigned __int128)m << e) * 1000000u;  // |x| < 10^15 < 2^50: m << e < 2^50, the product < 2^70
  } else {
    const unsigned __int128 prod = (unsigned __int128)m * 1000000u;  // < 2^73
    const int s = -e;
    if (s > 80) q = 0;  // |x| * 10^6 < 2^73 / 2^81: far below one half
    else {
      q = prod >> s;
      const unsigned __int128 rem = prod & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
      if (rem > half || (rem == half && (q & 1))) ++q;
    }
  }
  const uint64_t ip = (uint64_t)(q / 1000000u), fp = (uint64_t)(q % 1000000u);
  char buf[32];
  int n = 31;
  buf[n] = 0;
  uint64_t f = fp;
  for (int k = 0; k < 6; ++k) { buf[--n] = (char)('0' + f % 10); f /= 10; }
  buf[--n] = '.';
  uint64_t i = ip;
  do { buf[--n] = (char)('0' + i % 10); i /= 10; } while (i);
  if (neg) buf[--n] = '-';
  out.append(buf + n, (size_t)(31 - n));
}
inline string fmt6(double x) { string o; fmt6_append(o, x); return o; }
inline string fmt_opt(const std::optional<double>& v) { return (!v || std::isnan(*v)) ? "NA" : fmt6(*v); }  // process.rs:3702-3713
inline void falsta_div_value(string& out, double v) {  // process.rs:3786-3792
  if (std::isnan(v)) out += "NA"; else if (v == 0.0) out += '0'; else fmt6_append(out, v);
}
inline void falsta_fst_value(string& out, double v) {  // process.rs:3842-3856
  if (std::isnan(v)) out += "NA";
  else if (std::isinf(v)) out += v > 0 ? "Infinity" : "-Infinity";
  else if (v == 0.0) out += '0';
  else fmt6_append(out, v);
}

// ---- data model (process.rs:397-536) --------------------------------------------------------------
struct Variant {
  int64_t position = 0;
  vector<uint8_t> data;  // CompressedGenotypes: 0xFF sentinel
  size_t stride = 0, num_samples = 0;
  size_t max_len = 0;  // longest genotype in the row before the sentinel (0 = every sample None)
  // genotype length of sample i (0 = None)
  size_t glen(size_t i) const {
    if (i >= num_samples || stride == 0) return 0;
    size_t n = 0;
    while (n < stride && data[i * stride + n] != 0xFF) ++n;
    return n;
  }
};

typedef vector<std::pair<string, std::pair<uint8_t, uint8_t>>> SampleMap;  // insertion-ordered, unique keys

struct ConfigEntry {
  string seqname;
  Interval interval;  // 0-based half-open
  SampleMap samples_unfiltered, samples_filtered;
};

enum : uint8_t { FLAG_PASS = 0, FLAG_MASK = 1, FLAG_ALLOW = 2, FLAG_LOW_GQ = 4, FLAG_MISSING = 8 };

typedef std::map<string, vector<Interval>> RegionMap;


// one per-site Weir & Cockerham record of a region (stats.rs:1996-2112), as the FALSTA / TSV writers take it
struct WcSite { int64_t pos1; double overall_fst, overall_num, overall_den, pair_fst, pair_num, pair_den; };

// ---- vcf_ingest.cpp: parse.rs and the VCF side of process.rs ------------------------------------------------------------
RegionMap parse_regions_file(const string& path);           // parse.rs:15-88
void sample_map_set(SampleMap& m, const string& k, uint8_t l, uint8_t r);  // insert or overwrite, first-seen order kept
vector<ConfigEntry> parse_config_file(const string& path);  // parse.rs:91-239
Interval parse_region(const string& r);                     // parse.rs:241-261
string find_vcf_file(const string& folder, const string& chr);  // parse.rs:263-515
string read_reference_sequence(const string& reference, const string& chr);  // process.rs:1915-1952
vector<Interval> find_n_regions(const string& seq);        // process.rs:1849-1874
vector<Interval> merge_intervals(vector<Interval> v);      // process.rs:762-783
string normalize_chr_prefix(const string& c);
struct VcfData {
  vector<Variant> variants;
  vector<uint8_t> flags;
  vector<string> sample_names;
};
vector<string> read_sample_names_from_vcf(const string& path);  // run_vcf.rs:190-214
VcfData process_vcf(const string& path, const string& chr, const vector<Interval>& regions, unsigned min_gq, const RegionMap* mask,
                    const RegionMap* allow, const std::set<string>& exclusion);  // process.rs:4092-4469
string normalize_sample_name(const string& n);
std::map<string, size_t> map_sample_names_to_indices(const vector<string>& names);  // process.rs:1192-1333
typedef vector<std::pair<size_t, int>> HapList;  // (sample index, side)
HapList haplotypes_for_group(uint8_t group, const SampleMap& filter, const std::map<string, size_t>& index);

// ---- writers.cpp: output.csv fields, the FALSTA tracks, the TSV headers, the writer self-checks ------------------------------
extern const char* kCsvHeader[34];
extern const char* kHudsonTsvHeader;
extern const char* kWcTsvHeader;
string csv_field(const string& f);
string join(const vector<string>& v, char d, bool csv_quote = false);
void gz_append(const string& path, const string& text);
// One track: write(sink) puts header + line into the sink and returns false (having written nothing) when the track is not to appear;
// `records` / `tokens` say how dense it is (values against positions), which picks the writer.
struct TrackFn {
  std::function<bool(TrackSink&)> write;
  size_t records = 0, tokens = 0;
  string text() const { TextSink t; return write(t) ? std::move(t.out) : string(); }  // tests, --print_formats
};
extern std::atomic<unsigned> g_region_workers;  // region workers running side by side (set by run())
vector<vector<string>> compress_tracks(const vector<vector<TrackFn>>& files, size_t approx_tokens);
void append_members(const string& path, const vector<string>& members);
struct RegionOutput {
  vector<string> csv_row;
  string seqname;
  int64_t region_start1 = 0, region_end1 = 0;
  vector<std::tuple<int64_t, double, double, int, bool>> diversity;  // (pos1, pi, theta, group, filtered)
  vector<WcSite> wc_sites;
  vector<std::tuple<int64_t, double, double, double>> hudson_sites;
  vector<vector<string>> hudson_rows;
  vector<vector<string>> wc_rows;
  vector<string> diversity_members, fst_members;  // the region's FALSTA tracks as gzip members, made by the region's worker
};
vector<TrackFn> diversity_tracks(const RegionOutput& r);  // append_diversity_falsta, process.rs:3740-3806
vector<TrackFn> fst_tracks(const RegionOutput& r);        // append_fst_falsta, process.rs:3809-4003
int print_formats();
int dump_writer_cases(const string& dir, int count);
int check_fmt6(size_t n);
int bench_tracks(int variants, int length);

}  // namespace fmv
