// deflate_runs.cpp — see deflate_runs.hpp.  No GPU, no ferromic types: zlib and the standard library only, so that the run-aware writer can be
// tested on its own (run_vcf --check_writers, --dump_writer_cases; tests/test_output_formats_cpu.py inflates its members with zlib AND gzip).
#include "deflate_runs.hpp"

#include <zlib.h>

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

namespace fmv {

using std::string;
using std::vector;
typedef std::runtime_error Error;

// CRC-32 of the gzip trailer, eight bytes per step (slicing by 8).  zlib 1.2.11's crc32 - what deflate() runs over every input byte when it
// writes the gzip wrapper itself - does about 1 GB/s, and a region's tracks are a megabyte of text that deflates at several GB/s because
// it is mostly runs of one token: the checksum was half of a sparse region's track time.  gzip_member therefore deflates RAW and frames the
// member itself.  `run_vcf --check_writers` also checks this against zlib's crc32 on random buffers.
uint32_t crc32_slice8(const uint8_t* p, size_t n, uint32_t crc) {
  static const auto table = [] {
    auto t = std::make_unique<std::array<std::array<uint32_t, 256>, 8>>();
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      (*t)[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) (*t)[(size_t)s][i] = ((*t)[(size_t)s - 1][i] >> 8) ^ (*t)[0][(*t)[(size_t)s - 1][i] & 0xFF];
    return t;
  }();
  const auto& T = *table;
  crc = ~crc;
  while (n && ((uintptr_t)p & 7)) { crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);  // little-endian host (x86-64)
    w ^= crc;
    crc = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^ T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^
          T[1][(w >> 48) & 0xFF] ^ T[0][w >> 56];
    p += 8; n -= 8;
  }
  while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

// A complete gzip member holding `text` (what one open_append_compressed + write + finish produces).
string gzip_member(const string& text) {
  // Level 1: the tracks are long runs of "0," / "NA," around sparse values; the default level spends ~1 ms per 40 kB of such
  // text searching for longer matches and gains a few hundred bytes.  Readers see the same text either way.
  static const int level = getenv("FERROMIC_GZIP_LEVEL") ? atoi(getenv("FERROMIC_GZIP_LEVEL")) : 1;
  // One deflate state per thread, reset per member: deflateInit2 allocates and clears ~270 kB (one block of it above the allocator's
  // mmap threshold), which was half the cost of a 30-kB track and a map / unmap per track on the process's address space.
  struct State {
    z_stream z;
    bool ready = false;
    ~State() { if (ready) deflateEnd(&z); }
  };
  thread_local State st;
  z_stream& z = st.z;
  if (!st.ready) {
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15 /* raw: header, CRC-32 and length are written here */, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Error("deflateInit2 failed");
    st.ready = true;
  } else if (deflateReset(&z) != Z_OK) {
    throw Error("deflateReset failed");
  }
  string out;
  out.resize(deflateBound(&z, (uLong)std::min<size_t>(text.size(), (size_t)1 << 30)) + 64);
  static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};  // no name, no time stamp, written on Unix: zlib's own default header
  memcpy(&out[0], header, 10);
  size_t in_off = 0, out_off = 10;
  for (;;) {
    const size_t n = std::min<size_t>(text.size() - in_off, (size_t)1 << 30);
    z.next_in = (Bytef*)text.data() + in_off;
    z.avail_in = (uInt)n;
    in_off += n;
    const int flush = in_off == text.size() ? Z_FINISH : Z_NO_FLUSH;
    int rc;
    do {
      if (out.size() - out_off < (size_t)1 << 16) out.resize(out.size() * 2 + ((size_t)1 << 16));
      z.next_out = (Bytef*)&out[out_off];
      const size_t room = std::min<size_t>(out.size() - out_off, (size_t)1 << 30);
      z.avail_out = (uInt)room;
      rc = deflate(&z, flush);
      if (rc == Z_STREAM_ERROR) throw Error("deflate failed");
      out_off += room - z.avail_out;
    } while (z.avail_out == 0 || (flush == Z_FINISH && rc != Z_STREAM_END));
    if (flush == Z_FINISH) break;
  }
  out.resize(out_off + 8);
  const uint32_t crc = crc32_slice8(reinterpret_cast<const uint8_t*>(text.data()), text.size()), isize = (uint32_t)text.size();
  for (int k = 0; k < 4; ++k) { out[out_off + (size_t)k] = (char)(crc >> (8 * k)); out[out_off + 4 + (size_t)k] = (char)(isize >> (8 * k)); }
  return out;
}

void TextSink::run(const char* pattern, size_t period, size_t count) {
    thread_local string pat, pat_of;
    if (pat.empty() || pat_of.size() != period || memcmp(pat_of.data(), pattern, period) != 0) {
      pat.clear();
      while (pat.size() < 65536) pat.append(pattern, period);
      pat_of.assign(pattern, period);
    }
    const size_t per_block = pat.size() / period;
    while (count) {
      const size_t t = std::min(count, per_block);
      out.append(pat.data(), t * period);
      count -= t;
    }
}

// CRC-32 register update without the pre / post inversion (crc32_slice8's inner loop): affine in the register, which is what lets a run of a
// repeated block be folded in a few table look-ups per block instead of a pass over its bytes.
uint32_t crc32_raw(const uint8_t* p, size_t n, uint32_t state) { return ~crc32_slice8(p, n, ~state); }
// The register after `len` bytes of a fixed block, for any register before: state' = M * state ^ c.  M (32 x 32 over GF(2)) is held as four
// 256-entry tables, c = the block's own contribution.  Levels: the pattern repeated 16, 256, 4 096 and 65 536 times; twice a block is
// (M^2, M c ^ c), so each level comes from the previous one by four doublings and only the smallest touches bytes.
struct RunCrc {
  struct Level { size_t bytes; uint32_t t[4][256]; uint32_t c; };
  Level level[4];
  string pattern;
  uint8_t tail[64 * 16];  // the pattern repeated: the < 16 repetitions below the smallest level
  static void tables_of(const uint32_t (&col)[32], uint32_t (&t)[4][256]) {
    for (int b = 0; b < 4; ++b)
      for (int v = 0; v < 256; ++v) { uint32_t x = 0; for (int k = 0; k < 8; ++k) if (v >> k & 1) x ^= col[8 * b + k]; t[b][v] = x; }
  }
  static uint32_t apply(const uint32_t (&col)[32], uint32_t v) { uint32_t x = 0; for (int k = 0; k < 32; ++k) if (v >> k & 1) x ^= col[k]; return x; }
  explicit RunCrc(const string& pat) : pattern(pat) {
    const size_t d = pat.size();
    if (d == 0 || d > 64) throw Error("run pattern of 1..64 bytes expected");
    for (size_t k = 0; k < 16; ++k) memcpy(tail + k * d, pat.data(), d);
    uint32_t col[32], c;
    const vector<uint8_t> zeros(16 * d, 0);
    for (int k = 0; k < 32; ++k) col[k] = crc32_raw(zeros.data(), zeros.size(), 1u << k);
    c = crc32_raw(tail, 16 * d, 0);
    size_t bytes = 16 * d;
    for (int lv = 0; lv < 4; ++lv) {
      if (lv) for (int dbl = 0; dbl < 4; ++dbl) {  // sixteen times the block
        c = apply(col, c) ^ c;
        uint32_t sq[32];
        for (int k = 0; k < 32; ++k) sq[k] = apply(col, col[k]);
        memcpy(col, sq, sizeof col);
        bytes *= 2;
      }
      level[lv].bytes = bytes;
      level[lv].c = c;
      tables_of(col, level[lv].t);
    }
  }
  uint32_t advance(uint32_t state, size_t count) const {  // the register after `count` more repetitions of the pattern
    size_t bytes = count * pattern.size();
    for (int lv = 3; lv >= 0; --lv) {
      const Level& L = level[lv];
      while (bytes >= L.bytes) {
        state = L.t[0][state & 0xFF] ^ L.t[1][(state >> 8) & 0xFF] ^ L.t[2][(state >> 16) & 0xFF] ^ L.t[3][state >> 24] ^ L.c;
        bytes -= L.bytes;
      }
    }
    return bytes ? crc32_raw(tail, bytes, state) : state;
  }
};

void RunDeflateSink::canonical(const uint8_t* len, int n, uint16_t* code) {
    int count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    for (int b = 1, c = 0; b < 16; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)rev((uint32_t)next[len[i]]++, len[i]) : 0;
  }
const RunDeflateSink::Codes& RunDeflateSink::fixed_codes() {  // RFC 1951 3.2.6
    static const Codes k = [] {
      Codes c{};
      for (int sym = 0; sym < 288; ++sym) c.lit_len[sym] = sym < 144 ? 8 : sym < 256 ? 9 : sym < 280 ? 7 : 8;
      canonical(c.lit_len, 288, c.lit);
      for (int d = 0; d < 30; ++d) { c.dist_len[d] = 5; c.dist[d] = (uint16_t)rev((uint32_t)d, 5); }
      return c;
    }();
    return k;
  }
const RunDeflateSink::Codes& RunDeflateSink::tuned_codes() {
    static const Codes k = [] {
      Codes c{};
      for (int sym = 0; sym < 286; ++sym) c.lit_len[sym] = sym <= 20 ? 12 : 11;  // twenty control bytes in 12 bits (0..20 without the newline, set below)
      for (int ch = '0'; ch <= '9'; ++ch) c.lit_len[ch] = 4;
      c.lit_len[(int)','] = 4; c.lit_len[(int)'.'] = 4;
      for (int sym : {(int)'N', (int)'A', (int)'-', (int)'\n', 256, 283, 284, 285}) c.lit_len[sym] = 6;
      canonical(c.lit_len, 286, c.lit);
      for (int d = 0; d < 30; ++d) c.dist_len[d] = d == 1 || d == 2 ? 2 : d >= 26 ? 5 : 6;
      canonical(c.dist_len, 30, c.dist);
      return c;
    }();
    return k;
  }
RunDeflateSink::RunDeflateSink(bool tuned) : code_set(tuned ? &tuned_codes() : &fixed_codes()) {
    static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    out.assign(reinterpret_cast<const char*>(header), 10);
    put(1, 1);  // BFINAL
    if (!tuned) { put(1, 2); return; }  // BTYPE = 01: fixed Huffman codes
    put(2, 2);                          // BTYPE = 10: the code lengths follow
    put(286 - 257, 5);                  // HLIT
    put(30 - 1, 5);                     // HDIST
    // the code that spells the lengths: only 2, 4, 5, 6, 11 and 12 occur (no repeat symbols): 11 in one bit, 4 / 6 / 12 in three, 2 / 5 in four
    uint8_t cl_len[19] = {0};
    cl_len[11] = 1; cl_len[4] = 3; cl_len[6] = 3; cl_len[12] = 3; cl_len[2] = 4; cl_len[5] = 4;
    uint16_t cl_code[19];
    canonical(cl_len, 19, cl_code);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    put(16 - 4, 4);                     // HCLEN: the first sixteen of `order` (up to symbol 2)
    for (int i = 0; i < 16; ++i) put(cl_len[order[i]], 3);
    const Codes& c = codes();
    for (int sym = 0; sym < 286; ++sym) put(cl_code[c.lit_len[sym]], cl_len[c.lit_len[sym]]);
    for (int d = 0; d < 30; ++d) put(cl_code[c.dist_len[d]], cl_len[c.dist_len[d]]);
  }
void RunDeflateSink::match(size_t len, size_t dist) {  // 3 <= len <= 258, 1 <= dist <= 32 768
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int k = 28;
    while (base[k] > len) --k;
    const Codes& c = codes();
    put(c.lit[257 + k], c.lit_len[257 + k]);
    if (extra[k]) put((uint32_t)(len - base[k]), extra[k]);
    int dk = 29;
    while (dbase[dk] > dist) --dk;
    put(c.dist[dk], c.dist_len[dk]);
    if (dextra[dk]) put((uint32_t)(dist - dbase[dk]), dextra[dk]);
  }
void RunDeflateSink::text(const char* p, size_t n) {
    bool as_match = false;
    if (n >= 4 && n <= sizeof Seen::text) {
      uint64_t h = 1469598103934665603ull;
      for (size_t i = 0; i < n; ++i) h = (h ^ (uint8_t)p[i]) * 1099511628211ull;
      Seen& e = seen[(h >> 20) % kSeenSlots];
      if (e.len == n && memcmp(e.text, p, n) == 0 && total - e.at <= 32768 && total > e.at) { match(n, (size_t)(total - e.at)); as_match = true; }
      e.at = total; e.len = (uint8_t)n; memcpy(e.text, p, n);
    }
    if (!as_match) for (size_t i = 0; i < n; ++i) literal((uint8_t)p[i]);
    crc_state = crc32_raw(reinterpret_cast<const uint8_t*>(p), n, crc_state);
    total += n;
  }
void RunDeflateSink::run(const char* pattern, size_t period, size_t count) {
    if (!count) return;
    if (period > 64 || period == 0) {  // (RunCrc's tail buffer holds patterns of up to 64 bytes)
      for (size_t k = 0; k < count; ++k) text(pattern, period);
      return;
    }
    thread_local std::map<string, std::unique_ptr<RunCrc>> tables;  // per pattern, built on first use
    const string key(pattern, period);
    auto it = tables.find(key);
    if (it == tables.end()) it = tables.emplace(key, std::make_unique<RunCrc>(key)).first;
    for (size_t i = 0; i < period; ++i) literal((uint8_t)pattern[i]);
    size_t rest = (count - 1) * period;
    while (rest >= 258 + 3 || rest == 258) { match(258, period); rest -= 258; }  // never leave a tail of 1 or 2 bytes behind a full match
    if (rest > 258) { const size_t half = rest / 2; match(half, period); rest -= half; }
    if (rest >= 3) { match(rest, period); rest = 0; }
    for (size_t i = 0; i < rest; ++i) literal((uint8_t)pattern[i % period]);  // rest < 3 only when the whole run is shorter than one match
    crc_state = it->second->advance(crc_state, count);
    total += count * period;
  }
string RunDeflateSink::finish() {  // end-of-block, trailer; the object is spent
    const Codes& c = codes();
    put(c.lit[256], c.lit_len[256]);
    if (nbits) put(0, 8 - nbits);
    const uint32_t crc = ~crc_state, isize = (uint32_t)total;
    for (int k = 0; k < 4; ++k) out.push_back((char)(crc >> (8 * k)));
    for (int k = 0; k < 4; ++k) out.push_back((char)(isize >> (8 * k)));
    return std::move(out);
  }

}  // namespace fmv
