// deflate_runs.hpp — the gzip side of run_vcf's writers (reference: open_append_compressed + the FALSTA writers of process.rs:3655-4052, which
// hand every track to flate2): a CRC-32 that keeps up with run-heavy text, one-member-per-call zlib framing, and the sinks a track is
// written into - plain text, or a gzip member produced directly from literal text and RUNS of a repeated token.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

namespace fmv {

// CRC-32 (the gzip trailer's), eight bytes per step; crc32_raw is the register update without the pre / post inversion
uint32_t crc32_slice8(const uint8_t* p, size_t n, uint32_t crc = 0);
uint32_t crc32_raw(const uint8_t* p, size_t n, uint32_t state);

// A complete gzip member holding `text`, deflated by zlib (level FERROMIC_GZIP_LEVEL, default 1), framed here.  Throws std::runtime_error.
std::string gzip_member(const std::string& text);

// ---- where a track's text goes ------------------------------------------------------------------------------------------------------
// A FALSTA track is runs of one default token around sparse values: a 1-Mb region is 17 tracks of a million tokens, 45 MB of text of which a
// few hundred kB are values.  Writers therefore hand their output to a sink as literal text and as RUNS ("pattern" repeated `count` times):
// TextSink materialises the text (tests, --print_formats, dense tracks that go through zlib), RunDeflateSink writes the gzip member directly
// - a run is a handful of length-258 back references and a table-driven CRC step, whatever its length.
struct TrackSink {
  virtual ~TrackSink() {}
  virtual void text(const char* p, size_t n) = 0;
  virtual void run(const char* pattern, size_t period, size_t count) = 0;  // pattern[0 .. period) written `count` times
  void text(const std::string& t) { text(t.data(), t.size()); }
};
struct TextSink : TrackSink {
  std::string out;
  void text(const char* p, size_t n) override { out.append(p, n); }
  void run(const char* pattern, size_t period, size_t count) override;
};


struct RunCrc;  // folds a run of a repeated block into the CRC register in a few table look-ups (deflate_runs.cpp)

// A gzip member written directly: one deflate block with the FIXED Huffman codes (RFC 1951 3.2.6).  Literal text costs 8-9 bits a byte (no
// entropy coding: this writer is for tracks that are mostly runs - dense ones go through zlib, compress_tracks decides); a run is its
// pattern once as literals, then back references of up to 258 bytes at distance = the period, 13 bits each.
struct RunDeflateSink : TrackSink {
  std::string out;
  uint64_t acc = 0;
  int nbits = 0;
  uint32_t crc_state = 0xFFFFFFFFu;  // raw register (crc32's ~crc)
  uint64_t total = 0;
  struct Codes { uint16_t lit[288]; uint8_t lit_len[288]; uint16_t dist[30]; uint8_t dist_len[30]; };
  static uint32_t rev(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i); return r; }
  // canonical Huffman codes of a set of lengths (RFC 1951 3.2.2), bit-reversed for the LSB-first stream
  static void canonical(const uint8_t* len, int n, uint16_t* code);
  static const Codes& fixed_codes();
  // A code made for this text instead of the fixed one: digits, ',' and '.' in 4 bits, 'N' 'A' '-' newline, end-of-block and the three longest
  // length symbols in 6, every other byte and length in 11 or 12 (complete: 12/16 + 8/64 + 246/2048 + 20/4096 = 1); the two run distances
  // (2 and 3: ",0" and ",NA") in 2 bits, the rest in 5 or 6 (2/4 + 4/32 + 24/64 = 1).  Sent once per member as a dynamic block's header.
  static const Codes& tuned_codes();
  const Codes* code_set;
  const Codes& codes() const { return *code_set; }
  // tuned = true: a dynamic block carrying tuned_codes() (about 60 bytes of header: worth it from a few dozen values on)
  explicit RunDeflateSink(bool tuned = false);
  void put(uint32_t v, int n) {
    acc |= (uint64_t)v << nbits;
    nbits += n;
    while (nbits >= 8) { out.push_back((char)(acc & 0xFF)); acc >>= 8; nbits -= 8; }
  }
  void literal(uint8_t b) { const Codes& c = codes(); put(c.lit[b], c.lit_len[b]); }
  void match(size_t len, size_t dist);
  // The values of a track are few distinct texts (theta is one number at every segregating site, pi and the F_ST components functions of a
  // handful of allele counts): a value seen in the last 32 KiB of the text goes as ONE back reference to its previous occurrence instead of
  // eight-bit literals - what zlib's hash chains find, at one table probe per token.
  struct Seen { uint64_t at; uint8_t len; char text[23]; };
  static constexpr size_t kSeenSlots = 1024;
  std::unique_ptr<Seen[]> seen{new Seen[kSeenSlots]()};
  using TrackSink::text;
  void text(const char* p, size_t n) override;
  void run(const char* pattern, size_t period, size_t count) override;
  std::string finish();
};

}  // namespace fmv
