// vcf_ingest.cpp — the text side of run_vcf: config TSV, BED / TSV regions, FASTA index, VCF (plain, gzip, BGZF) into the Variant rows of
// process.rs, and the sample-name mapping.  Restates parse.rs and process.rs:4092-4768; nothing here touches the GPU.
#include "run_vcf.hpp"

namespace fmv {

// ---- parse.rs ---------------------------------------------------------------------------------------
RegionMap parse_regions_file(const string& path) {  // parse.rs:15-88
  std::ifstream in(path);
  if (!in) throw Error("cannot open regions file " + path);
  const bool is_bed = ends_with(path, ".bed");
  RegionMap regions;
  string line;
  while (std::getline(in, line)) {
    vector<string> f = split_ws(line);
    if (f.size() < 3) continue;
    int64_t s, e;
    if (!parse_i64(f[1], &s) || !parse_i64(f[2], &e)) continue;
    regions[trim_start_matches(f[0], "chr")].push_back(is_bed ? Interval{s, e} : from_1based_inclusive(s, e));
  }
  for (auto& kv : regions) std::stable_sort(kv.second.begin(), kv.second.end(), [](const Interval& a, const Interval& b) { return (uint64_t)a.first < (uint64_t)b.first; });
  return regions;
}

void sample_map_set(SampleMap& m, const string& k, uint8_t l, uint8_t r) {
  for (auto& kv : m) if (kv.first == k) { kv.second = {l, r}; return; }
  m.push_back({k, {l, r}});
}

vector<ConfigEntry> parse_config_file(const string& path) {  // parse.rs:91-239
  std::ifstream in(path);
  if (!in) throw Error("cannot open config file " + path);
  string line;
  vector<string> headers;
  vector<ConfigEntry> entries;
  size_t line_no = 0;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    ++line_no;
    vector<string> rec = split(line, '\t');
    if (headers.empty()) {
      headers = rec;
      if (headers.size() <= 7) throw Error("Parse(\"No sample names found in config file header.\")");
      continue;
    }
    if (rec.size() != headers.size()) throw Error("Parse(\"Mismatched number of fields in record on line " + std::to_string(line_no) + "\")");
    ConfigEntry e;
    e.seqname = trim_start_matches(trim(rec[0]), "chr");
    int64_t s, en;
    if (!parse_i64(rec[1], &s)) throw Error("Parse(\"Invalid start\")");
    if (!parse_i64(rec[2], &en)) throw Error("Parse(\"Invalid end\")");
    e.interval = from_1based_inclusive(s, en);
    for (size_t i = 7; i < rec.size(); ++i) {
      const string& field = rec[i];
      const string& name = headers[i];
      const string g = split(field, '_')[0];
      if (g.size() >= 3 && g[1] == '|' && isdigit((unsigned char)g[0]) && isdigit((unsigned char)g[2])) {
        const int l = g[0] - '0', r = g[2] - '0';
        if (l <= 1 && r <= 1) sample_map_set(e.samples_unfiltered, name, (uint8_t)l, (uint8_t)r);
      }
      if (field == "0|0" || field == "0|1" || field == "1|0" || field == "1|1")
        sample_map_set(e.samples_filtered, name, (uint8_t)(field[0] - '0'), (uint8_t)(field[2] - '0'));
    }
    if (e.samples_unfiltered.empty()) continue;
    entries.push_back(std::move(e));
  }
  if (headers.empty()) throw Error("empty config file");
  return entries;
}

Interval parse_region(const string& r) {  // parse.rs:241-261
  vector<string> p = split(r, '-');
  int64_t s, e;
  if (p.size() != 2) throw Error("InvalidRegion(\"Invalid region format. Use start-end\")");
  if (!parse_i64(p[0], &s)) throw Error("InvalidRegion(\"Invalid start position\")");
  if (!parse_i64(p[1], &e)) throw Error("InvalidRegion(\"Invalid end position\")");
  if (s >= e) throw Error("InvalidRegion(\"Start position must be less than end position\")");
  return from_1based_inclusive(s, e);
}

string find_vcf_file(const string& folder, const string& chr) {  // parse.rs:263-515
  if (!is_dir(folder)) throw Error("VCF folder does not exist: " + folder);
  for (const string& pat : {"chr" + chr + ".vcf.gz", "chr" + chr + ".vcf", chr + ".vcf.gz", chr + ".vcf"})
    if (file_exists(folder + "/" + pat)) return folder + "/" + pat;
  auto boundary_match = [&](const string& name) {
    for (const string& pat : {"chr" + chr, chr}) {
      size_t from = 0;
      for (;;) {
        size_t idx = name.find(pat, from);
        if (idx == string::npos) break;
        const bool after_ok = idx + pat.size() >= name.size() || !isdigit((unsigned char)name[idx + pat.size()]);
        const bool before_ok = idx == 0 || !isdigit((unsigned char)name[idx - 1]);
        if (after_ok && before_ok) return true;
        from = idx + 1;
      }
    }
    return false;
  };
  auto prefix_boundary = [](const string& name, const string& prefix) {
    if (!starts_with(name, prefix)) return false;
    return name.size() == prefix.size() || !isdigit((unsigned char)name[prefix.size()]);
  };
  vector<std::pair<int, string>> cands;
  DIR* d = opendir(folder.c_str());
  if (!d) throw Error("cannot read directory " + folder);
  while (dirent* ent = readdir(d)) {
    const string name = ent->d_name;
    if (!(ends_with(name, ".vcf") || ends_with(name, ".vcf.gz"))) continue;
    bool aux = false;
    for (const char* x : {".csi", ".tbi", ".idx", ".md5", ".bai"}) aux |= ends_with(name, x);
    if (aux || !boundary_match(name)) continue;
    int score = 0;
    if (name == "chr" + chr + ".vcf.gz") score += 100;
    else if (name == "chr" + chr + ".vcf") score += 90;
    else if (name == chr + ".vcf.gz") score += 80;
    else if (name == chr + ".vcf") score += 70;
    if (ends_with(name, ".vcf.gz")) score += 15;
    if (prefix_boundary(name, "chr" + chr)) score += 10;
    else if (prefix_boundary(name, chr)) score += 5;
    score -= (int)(name.size() / 5);
    cands.push_back({-score, folder + "/" + name});
  }
  closedir(d);
  if (cands.empty()) throw Error("NoVcfFiles");
  std::sort(cands.begin(), cands.end());
  return cands[0].second;
}

// line reader over plain or (multi-member) gzip files
// BGZF (bgzip / htslib) is a series of independent gzip members of <= 64 KiB, each announcing its compressed size in a
// 'BC' extra subfield: the blocks of a batch are inflated in parallel, so a .vcf.gz from bgzip is ingested about as fast
// as plain text.  Any other gzip stream stays on zlib's serial reader.
struct BgzfSource {
  int fd = -1;
  int64_t file_off = 0;
  bool file_eof = false;
  vector<unsigned char> raw;  // compressed bytes not yet consumed
  string out;                 // inflated text of the current batch
  size_t out_pos = 0;

  static bool is_bgzf(const string& path) {
    unsigned char h[18];
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    const ssize_t got = pread(fd, h, sizeof h, 0);
    close(fd);
    return got == (ssize_t)sizeof h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
  }
  explicit BgzfSource(const string& path) : fd(open(path.c_str(), O_RDONLY)) {
    if (fd < 0) throw Error("cannot open " + path);
  }
  ~BgzfSource() { if (fd >= 0) close(fd); }

  bool refill() {  // inflate the next batch of blocks into `out`; false at end of file
    out.clear();
    out_pos = 0;
    for (;;) {
      if (!file_eof) {
        const size_t want = (size_t)32 << 20, had = raw.size();
        raw.resize(had + want);
        size_t got_total = 0;
        while (got_total < want) {
          const ssize_t got = pread(fd, raw.data() + had + got_total, want - got_total, file_off);
          if (got <= 0) { file_eof = true; break; }
          got_total += (size_t)got;
          file_off += got;
        }
        raw.resize(had + got_total);
      }
      struct Block { size_t data, clen, isize, out_off; };
      vector<Block> blocks;
      size_t pos = 0, total = 0;
      while (pos + 18 <= raw.size()) {
        const unsigned char* h = raw.data() + pos;
        if (!(h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4))) throw Error("corrupt BGZF block header");
        const size_t xlen = h[10] | (h[11] << 8);
        if (pos + 12 + xlen > raw.size()) break;
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + xlen;) {  // extra subfields: SI1 SI2 SLEN(2) data
          const size_t slen = h[x + 2] | (h[x + 3] << 8);
          if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
          x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) throw Error("BGZF block without a BC size field");
        if (pos + bsize > raw.size()) break;  // block not complete yet
        const unsigned char* tail = raw.data() + pos + bsize - 4;
        const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        blocks.push_back({pos + 12 + xlen, bsize - 12 - xlen - 8, isize, total});
        total += isize;
        pos += bsize;
      }
      if (blocks.empty()) {
        if (file_eof) { if (!raw.empty() && pos < raw.size()) throw Error("truncated BGZF file"); return false; }
        continue;  // need more bytes for one whole block
      }
      out.resize(total);
      std::atomic<size_t> next{0};
      std::atomic<bool> bad{false};
      parallel_for((unsigned)std::min<size_t>(worker_threads(), blocks.size()), [&](unsigned) {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= blocks.size()) break;
          const Block& b = blocks[i];
          if (b.isize == 0) continue;
          z_stream z;
          memset(&z, 0, sizeof z);
          if (inflateInit2(&z, -15) != Z_OK) { bad = true; continue; }
          z.next_in = raw.data() + b.data;
          z.avail_in = (uInt)b.clen;
          z.next_out = (Bytef*)&out[b.out_off];
          z.avail_out = (uInt)b.isize;
          const int rc = inflate(&z, Z_FINISH);
          if (rc != Z_STREAM_END || z.avail_out != 0) bad = true;
          inflateEnd(&z);
        }
      });
      if (bad) throw Error("BGZF block failed to inflate");
      raw.erase(raw.begin(), raw.begin() + (ptrdiff_t)pos);
      if (total == 0) { if (file_eof && raw.empty()) return false; continue; }  // only empty (EOF marker) blocks in this batch
      return true;
    }
  }
  size_t read(char* dst, size_t n) {
    size_t done = 0;
    while (done < n) {
      if (out_pos == out.size() && !refill()) break;
      const size_t take = std::min(n - done, out.size() - out_pos);
      memcpy(dst + done, out.data() + out_pos, take);
      out_pos += take;
      done += take;
    }
    return done;
  }
  bool next_line(string& line) {
    line.clear();
    for (;;) {
      if (out_pos == out.size() && !refill()) return !line.empty();
      const char* base = out.data() + out_pos;
      const void* nl = memchr(base, '\n', out.size() - out_pos);
      const size_t take = nl ? (size_t)((const char*)nl - base) + 1 : out.size() - out_pos;
      line.append(base, take);
      out_pos += take;
      if (nl) return true;
    }
  }
};

struct LineReader {
  gzFile f;
  std::unique_ptr<BgzfSource> bgzf;
  explicit LineReader(const string& path) : f(nullptr), path_(path) {
    if (BgzfSource::is_bgzf(path)) { bgzf.reset(new BgzfSource(path)); return; }
    f = gzopen(path.c_str(), "rb");
    if (!f) throw Error("cannot open " + path);
    gzbuffer(f, 1 << 20);
  }
  ~LineReader() { if (f) gzclose(f); if (raw_fd >= 0) close(raw_fd); if (map_base) munmap((void*)map_base, map_len); }
  int raw_fd = -1;
  int64_t raw_off = 0;
  string path_;
  size_t read(char* dst, size_t n) {  // raw bytes following whatever next() consumed
    if (bgzf) return bgzf->read(dst, n);
    if (raw_fd < 0 && gzdirect(f)) {  // plain text: skip zlib's copy and read the file itself from here on
      raw_off = (int64_t)gztell(f);
      raw_fd = open(path_.c_str(), O_RDONLY);
    }
    if (raw_fd >= 0) {
      size_t total = 0;
      while (total < n) {
        const ssize_t got = pread(raw_fd, dst + total, n - total, raw_off);
        if (got <= 0) break;
        total += (size_t)got;
        raw_off += got;
      }
      return total;
    }
    size_t total = 0;
    while (total < n) {
      const int got = gzread(f, dst + total, (unsigned)std::min<size_t>(n - total, 1u << 30));
      if (got <= 0) break;
      total += (size_t)got;
    }
    return total;
  }
  // Plain text only: the rest of the file (whatever follows the lines next() consumed) as one read-only mapping, so the
  // body is parsed where the page cache holds it - no read() copy, no reader thread.  False for gzip / BGZF input.
  const char* map_base = nullptr;
  size_t map_len = 0;
  bool map_rest(const char** base, size_t* len) {
    if (bgzf || !gzdirect(f)) return false;
    const int64_t off = (int64_t)gztell(f);
    const int fd = open(path_.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0 || off > st.st_size) { close(fd); return false; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    map_base = (const char*)m;
    map_len = (size_t)st.st_size;
    *base = map_base + off;
    *len = map_len - (size_t)off;
    return true;
  }
  bool next(string& out) {
    if (bgzf) return bgzf->next_line(out);
    out.clear();
    char buf[1 << 16];
    for (;;) {
      if (!gzgets(f, buf, sizeof buf)) return !out.empty();
      out += buf;
      if (!out.empty() && out.back() == '\n') return true;
    }
  }
};

struct FaiEntry { int64_t len, offset, line_bases, line_width; };

std::map<string, FaiEntry> read_fai(const string& reference) {
  std::ifstream in(reference + ".fai");
  if (!in) throw Error("Failed to open reference index " + reference + ".fai");
  std::map<string, FaiEntry> out;
  string line;
  while (std::getline(in, line)) {
    vector<string> f = split(line, '\t');
    if (f.size() < 5) continue;
    FaiEntry e;
    if (parse_i64(f[1], &e.len) && parse_i64(f[2], &e.offset) && parse_i64(f[3], &e.line_bases) && parse_i64(f[4], &e.line_width)) out[f[0]] = e;
  }
  return out;
}

string read_reference_sequence(const string& reference, const string& chr) {  // process.rs:1915-1952, parse.rs:545-650
  auto fai = read_fai(reference);
  auto it = fai.find(chr);
  if (it == fai.end()) it = fai.find("chr" + chr);
  if (it == fai.end()) throw Error("Chromosome " + chr + " not found in reference");
  const FaiEntry& e = it->second;
  std::ifstream in(reference, std::ios::binary);
  if (!in) throw Error("Failed to open reference file " + reference);
  string seq;
  seq.reserve((size_t)e.len);
  int64_t pos = 0;
  vector<char> buf((size_t)std::max<int64_t>(e.line_bases, 1));
  while (pos < e.len) {
    const int64_t line_idx = pos / e.line_bases, col = pos % e.line_bases;
    const int64_t take = std::min(e.line_bases - col, e.len - pos);
    in.seekg(e.offset + line_idx * e.line_width + col);
    in.read(buf.data(), take);
    if (in.gcount() != take) throw Error("Failed to read sequence for " + chr);
    seq.append(buf.data(), (size_t)take);
    pos += take;
  }
  return seq;
}

vector<Interval> find_n_regions(const string& seq) {  // process.rs:1849-1874
  vector<Interval> out;
  bool in_n = false;
  size_t start = 0;
  for (size_t i = 0; i < seq.size(); ++i) {
    const bool is_n = seq[i] == 'N' || seq[i] == 'n';
    if (is_n && !in_n) { in_n = true; start = i; }
    else if (!is_n && in_n) { in_n = false; out.push_back({(int64_t)start, (int64_t)i}); }
  }
  if (in_n) out.push_back({(int64_t)start, (int64_t)seq.size()});
  return out;
}

vector<Interval> merge_intervals(vector<Interval> v) {  // process.rs:762-783
  if (v.empty()) return v;
  std::stable_sort(v.begin(), v.end(), [](const Interval& a, const Interval& b) { return (uint64_t)a.first < (uint64_t)b.first; });
  vector<Interval> out;
  Interval cur = v[0];
  for (size_t i = 1; i < v.size(); ++i) {
    if ((uint64_t)v[i].first <= (uint64_t)cur.second) cur.second = (int64_t)std::max((uint64_t)cur.second, (uint64_t)v[i].second);
    else { out.push_back(cur); cur = v[i]; }
  }
  out.push_back(cur);
  return out;
}

string normalize_chr_prefix(const string& c) {
  for (const char* p : {"chr", "Chr", "CHR"}) if (starts_with(c, p)) return c.substr(3);
  return c;
}

// ---- string_view twins of the helpers above for the per-cell hot loop (no allocation per genotype cell) ------
typedef std::string_view sv;
sv trim_sv(sv s) {
  size_t b = 0, e = s.size();
  while (b < e && isspace((unsigned char)s[b])) ++b;
  while (e > b && isspace((unsigned char)s[e - 1])) --e;
  return s.substr(b, e - b);
}
bool parse_unsigned_sv(sv s, unsigned max, unsigned* out) {  // Rust str::parse::<u8/u16>: optional '+', ASCII digits only
  size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
  if (i >= s.size()) return false;
  unsigned long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (unsigned long)(s[i] - '0');
    if (v > max) return false;
  }
  *out = (unsigned)v;
  return true;
}
// the k-th ':'-separated part of a cell; false when the cell has fewer parts
bool colon_part(sv cell, size_t k, sv* out) {
  size_t b = 0;
  for (size_t i = 0;; ++i) {
    const size_t e = cell.find(':', b);
    if (i == k) { *out = cell.substr(b, e == sv::npos ? sv::npos : e - b); return true; }
    if (e == sv::npos) return false;
    b = e + 1;
  }
}

// per-thread scratch reused across lines
struct VariantScratch {
  vector<sv> cells;       // the kept sample columns of the line
  vector<uint32_t> off;   // start of sample s in vals
  vector<uint16_t> len;   // parsed alleles of sample s
  vector<uint8_t> none;   // 1 = genotype is None
  vector<uint8_t> vals;
};

// process_variant, process.rs:4471-4768.  Returns false when the line yields no variant.  `line` keeps its
// trailing newline exactly as the reference's read_line buffer does.
bool process_variant(sv line, const string& chr, const vector<Interval>& regions, const vector<size_t>& kept,
                     unsigned min_gq, const RegionMap* allow, const RegionMap* mask, VariantScratch& scr, Variant* out, uint8_t* out_flags) {
  // The nine fixed fields are cut first; the sample columns are then walked in place, one pass: a typical line is
  // thousands of 7-byte cells, and tokenising it into a vector before looking at any cell cost more than parsing it.
  // Every failure below makes the reference skip the line (an Err is printed and dropped, None is dropped), so only the
  // set of conditions matters, not the order they are found in.
  const char* const lbeg = line.data();
  const char* const lend = lbeg + line.size();
  sv fields[9];
  const char* cur = lbeg;
  bool more = true;  // a tab followed the last field cut so far
  for (int f = 0; f < 9; ++f) {
    if (!more) throw Error("Invalid VCF line format");
    const char* t = (const char*)memchr(cur, '\t', (size_t)(lend - cur));
    fields[f] = sv(cur, (size_t)((t ? t : lend) - cur));
    more = t != nullptr;
    cur = t ? t + 1 : lend;
  }
  const string vcf_chr = normalize_chr_prefix(string(trim_sv(fields[0])));
  const bool chr_ok = vcf_chr == normalize_chr_prefix(trim(chr));
  int64_t pos1 = 0;
  const bool pos_ok = parse_i64(string(fields[1]), &pos1);
  // the reference checks the column count before anything else; a short line is an error even on another chromosome,
  // which is the same outcome (skipped) as the None of a foreign chromosome - so the cheap exits come first
  if (!chr_ok) return false;
  if (!pos_ok) throw Error("Invalid position");
  if (pos1 < 1) throw Error("Invalid 1-based pos");
  const int64_t pos0 = pos1 - 1;
  bool in_regions = false;
  for (auto& r : regions) if (hal_contains(r, pos0)) { in_regions = true; break; }
  if (!in_regions) return false;
  uint8_t flags = FLAG_PASS;
  if (allow) {
    auto it = allow->find(vcf_chr);
    if (it != allow->end()) { if (!position_in_regions(pos0, it->second)) flags |= FLAG_ALLOW; }
    else flags |= FLAG_ALLOW;
  }
  if (mask) {
    auto it = mask->find(vcf_chr);
    if (it != mask->end())
      for (auto& m : it->second)
        if (std::max<uint64_t>((uint64_t)pos0, (uint64_t)m.first) < std::min<uint64_t>((uint64_t)pos0 + 1, (uint64_t)m.second)) { flags |= FLAG_MASK; break; }
  }
  bool indel = fields[3].size() != 1;
  if (!indel) {
    const sv alts = fields[4];
    for (size_t b = 0;;) {
      const size_t e = alts.find(',', b);
      if ((e == sv::npos ? alts.size() : e) - b != 1) indel = true;
      if (e == sv::npos) break;
      b = e + 1;
    }
  }
  size_t gq_index = SIZE_MAX;
  {
    const sv fmt = fields[8];
    size_t b = 0;
    for (size_t i = 0;; ++i) {
      const size_t e = fmt.find(':', b);
      if (fmt.substr(b, e == sv::npos ? sv::npos : e - b) == "GQ") { gq_index = i; break; }
      if (e == sv::npos) break;
      b = e + 1;
    }
  }
  const size_t n = kept.size();
  scr.off.resize(n); scr.len.resize(n); scr.none.resize(n); scr.cells.resize(n);
  scr.vals.clear();
  bool low_gq = false, missing = false;
  size_t max_len = 0;
  // Whole-line fast path: every kept cell is "a|b:GQ[:...]" with one-digit alleles, GQ second, and no sample column is skipped.  The
  // alleles go straight into the packed row, nothing else is recorded per cell; the first cell that looks different sends the whole
  // line through the general walk below (same acceptance rules as its per-cell shortcut, so the outcome is the same).
  if (gq_index == 1 && n && more && kept[0] == 9 && kept[n - 1] == 8 + n) {
    scr.vals.resize(2 * n);
    uint8_t* dst = scr.vals.data();
    const char* p = cur;
    size_t i = 0;
    bool low = false;
    for (; i < n; ++i) {
      if (lend - p < 5) break;
      const unsigned a = (unsigned)(p[0] - '0'), b = (unsigned)(p[2] - '0');
      if (a > 9u || b > 9u || p[3] != ':' || (p[1] != '|' && p[1] != '/')) break;
      const char* q = p + 4;
      unsigned v = 0;
      while (q < lend && (unsigned)(*q - '0') < 10u && v < 100000u) { v = v * 10 + (unsigned)(*q - '0'); ++q; }
      if (q == p + 4 || v > 65535u) break;
      if (q < lend && *q == ':') { const char* t = (const char*)memchr(q, '\t', (size_t)(lend - q)); q = t ? t : lend; }
      else if (q < lend && *q == '\n' && q + 1 == lend) q = lend;
      else if (q < lend && *q != '\t') break;
      dst[2 * i] = (uint8_t)a;
      dst[2 * i + 1] = (uint8_t)b;
      low |= v < min_gq;
      if (q == lend) { ++i; break; }  // the line ends with this cell
      p = q + 1;
    }
    if (i == n) {
      // all n cells were taken (a line that ends early leaves i < n and is reported by the general walk)
      if (low) flags |= FLAG_LOW_GQ;
      if (indel) return false;
      out->position = pos0;
      out->num_samples = n;
      out->stride = 2;
      out->max_len = 2;
      out->data.swap(scr.vals);
      scr.vals.clear();
      *out_flags = flags;
      return true;
    }
    scr.vals.clear();
  }
  size_t col = 9;  // column index of the field that starts at `cur` (valid while `more`)
  for (size_t i = 0; i < n; ++i) {
    // skip to column kept[i] (ascending: the header is read left to right)
    while (col < kept[i]) {
      if (!more) throw Error("Invalid VCF line format: missing genotype column");
      const char* t = (const char*)memchr(cur, '\t', (size_t)(lend - cur));
      more = t != nullptr;
      cur = t ? t + 1 : lend;
      ++col;
    }
    if (!more) throw Error("Invalid VCF line format: missing genotype column");
    const char* c = cur;
    const size_t room = (size_t)(lend - c);
    scr.off[i] = (uint32_t)scr.vals.size();
    const char* cell_end = nullptr;
    bool fast = false;
    // the overwhelmingly common cell "a|b:GQ..." with one-digit alleles and GQ second: parsed where it stands
    if (gq_index == 1 && room >= 5 && c[3] == ':' && (c[1] == '|' || c[1] == '/') && (unsigned)(c[0] - '0') < 10u && (unsigned)(c[2] - '0') < 10u) {
      size_t j = 4;
      unsigned v = 0;
      while (j < room && (unsigned)(c[j] - '0') < 10u && v < 100000u) { v = v * 10 + (unsigned)(c[j] - '0'); ++j; }
      if (j > 4 && v <= 65535u) {
        if (j == room || c[j] == '\t') { cell_end = c + j; fast = true; }
        else if (c[j] == '\n' && j + 1 == room) { cell_end = lend; fast = true; }
        else if (c[j] == ':') { const char* t = (const char*)memchr(c + j, '\t', room - j); cell_end = t ? t : lend; fast = true; }
      }
      if (fast) {
        scr.vals.push_back((uint8_t)(c[0] - '0'));
        scr.vals.push_back((uint8_t)(c[2] - '0'));
        scr.len[i] = 2;
        scr.none[i] = 2;  // called, and its GQ is already judged
        max_len = std::max<size_t>(max_len, 2);
        if (v < min_gq) low_gq = true;
      }
    }
    if (!cell_end) { const char* t = (const char*)memchr(c, '\t', room); cell_end = t ? t : lend; }
    const sv cell(c, (size_t)(cell_end - c));
    scr.cells[i] = cell;
    more = cell_end != lend;
    cur = more ? cell_end + 1 : lend;
    ++col;
    if (fast) continue;
    const sv alleles = cell.substr(0, cell.find(':'));
    scr.len[i] = 0;
    scr.none[i] = 1;
    if (alleles == "." || alleles == "./." || alleles == ".|.") continue;
    bool ok = true;
    size_t b = 0, cnt = 0;
    for (size_t j = 0; j <= alleles.size(); ++j) {
      if (j == alleles.size() || alleles[j] == '|' || alleles[j] == '/') {
        unsigned v;
        if (!parse_unsigned_sv(alleles.substr(b, j - b), 255, &v)) { ok = false; break; }
        scr.vals.push_back((uint8_t)v);
        ++cnt;
        b = j + 1;
      }
    }
    if (!ok) { scr.vals.resize(scr.off[i]); continue; }
    scr.none[i] = 0;
    scr.len[i] = (uint16_t)std::min<size_t>(cnt, 65535);
    max_len = std::max(max_len, cnt);
  }
  if (gq_index == SIZE_MAX) throw Error("GQ field not found in FORMAT");
  for (size_t i = 0; i < n; ++i) {
    if (scr.none[i] == 2) continue;
    if (scr.none[i]) { missing = true; continue; }
    sv part;
    if (!colon_part(scr.cells[i], gq_index, &part)) throw Error("GQ value missing in sample genotype field");
    const sv gq_str = trim_sv(part);
    unsigned gq = 0;
    if (!(gq_str == "." || gq_str.empty())) { if (!parse_unsigned_sv(gq_str, 65535, &gq)) gq = 0; }
    if (gq < min_gq) low_gq = true;
  }
  if (low_gq) flags |= FLAG_LOW_GQ;
  if (missing) flags |= FLAG_MISSING;
  if (indel) return false;
  // CompressedGenotypes::new, process.rs:440-477
  size_t max_ploidy = max_len;
  if (n) max_ploidy = std::max<size_t>(max_ploidy, 1);
  out->position = pos0;
  out->num_samples = n;
  out->stride = max_ploidy;
  out->max_len = max_len;
  if (scr.vals.size() == n * max_ploidy && max_ploidy == 2 && !missing) {
    // nothing is None, no genotype is longer than two alleles and there are 2 n alleles in all: every genotype is a called
    // diploid one (the usual line) and the parsed alleles already ARE the packed row
    out->data.swap(scr.vals);
    scr.vals.clear();
    *out_flags = flags;
    return true;
  }
  out->data.assign(n * max_ploidy, 0xFF);
  for (size_t s2 = 0; s2 < n; ++s2)
    if (scr.none[s2] != 1) {
      const uint8_t* src = &scr.vals[scr.off[s2]];
      uint8_t* dst = &out->data[s2 * max_ploidy];
      for (size_t k = 0, m = std::min<size_t>(scr.len[s2], max_ploidy); k < m; ++k) dst[k] = src[k];
    }
  *out_flags = flags;
  return true;
}

vector<string> read_sample_names_from_vcf(const string& path) {  // run_vcf.rs:190-214
  LineReader r(path);
  string line;
  while (r.next(line)) {
    if (starts_with(line, "#CHROM")) {
      vector<string> f = split_ws(line);
      if (f.size() <= 9) throw Error("VCF header found, but no sample columns");
      return vector<string>(f.begin() + 9, f.end());
    }
  }
  throw Error("No #CHROM line found in VCF header");
}

VcfData process_vcf(const string& path, const string& chr, const vector<Interval>& regions, unsigned min_gq,
                    const RegionMap* mask, const RegionMap* allow, const std::set<string>& exclusion) {  // process.rs:4092-4469
  VcfData d;
  vector<size_t> kept;
  LineReader r(path);
  string line;
  bool header = false;
  while (r.next(line)) {
    if (starts_with(line, "##")) continue;
    if (starts_with(line, "#CHROM")) {
      string h = line;
      vector<string> tabs = split(h, '\t');
      static const char* req[9] = {"#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"};
      bool ok = tabs.size() >= 9;
      for (int i = 0; ok && i < 9; ++i) ok = tabs[i] == req[i];
      if (!ok) throw Error("InvalidVcfFormat(\"Invalid VCF header format\")");
      vector<string> f = split_ws(h);
      for (size_t i = 9; i < f.size(); ++i) if (!exclusion.count(f[i])) { d.sample_names.push_back(f[i]); kept.push_back(i); }
      header = true;
      break;
    }
  }
  if (!header || d.sample_names.empty()) throw Error("Parse(\"No samples remain after applying exclusions\")");
  // Body: blocks of whole lines are cut from the (gunzipped) stream and parsed by a pool of threads, each on a
  // contiguous run of lines; results are concatenated in file order, so the outcome equals a serial read
  // (the reference runs the same stage as a reader thread + rayon consumers, process.rs:4274-4392).
  vector<std::pair<Variant, uint8_t>> items;
  const unsigned T = worker_threads();
  // one block of whole lines: cut into T line-aligned runs, each parsed by one pool thread; results keep the file order
  auto parse_block = [&](const char* bdata, size_t usable) {
    struct { const char* d; const char* data() const { return d; } } block{bdata};
    // line-aligned cut points
    vector<size_t> cut(T + 1, usable);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
      size_t p = std::max(cut[t - 1], usable * t / T);
      if (p < usable) { const void* q = memchr(block.data() + p, '\n', usable - p); p = q ? (size_t)((const char*)q - block.data()) + 1 : usable; }
      cut[t] = p;
    }
    vector<vector<std::pair<Variant, uint8_t>>> parts(T);
    vector<string> complaints(T);
    throw_if_no_gpu();
    StageTimer tparse("    ingest:parse_block");
    parallel_for(T, [&](unsigned t) {
      VariantScratch scr;
      size_t b = cut[t];
      const size_t end = cut[t + 1];
      while (b < end) {
        const void* q = memchr(block.data() + b, '\n', end - b);
        const size_t e = q ? (size_t)((const char*)q - block.data()) + 1 : end;
        Variant v;
        uint8_t fl;
        try {
          if (process_variant(sv(block.data() + b, e - b), chr, regions, kept, min_gq, allow, mask, scr, &v, &fl)) parts[t].push_back({std::move(v), fl});
        } catch (const Error& err) {
          complaints[t] += string(err.what()) + "\n";  // the collector prints and carries on (process.rs:4358-4360)
        }
        b = e;
      }
    });
    for (unsigned t = 0; t < T; ++t) {
      if (!complaints[t].empty()) fputs(complaints[t].c_str(), stderr);
      for (auto& it : parts[t]) items.push_back(std::move(it));
    }
  };
  const char* mapped = nullptr;
  size_t mapped_len = 0;
  static const bool env_no_mmap = getenv("FERROMIC_NO_MMAP") != nullptr;
  if (!env_no_mmap && r.map_rest(&mapped, &mapped_len)) {
    // plain text: blocks are windows of the mapping, each ending at a line end (a line longer than a block is one block)
    const size_t kWindow = getenv("FERROMIC_INGEST_BLOCK") ? std::max<size_t>(16, (size_t)atoll(getenv("FERROMIC_INGEST_BLOCK"))) : ((size_t)64 << 20);
    size_t off = 0;
    while (off < mapped_len) {
      size_t bsize = std::min(kWindow, mapped_len - off);
      if (off + bsize < mapped_len) {
        const void* nl = memrchr(mapped + off, '\n', bsize);
        if (nl) {
          bsize = (size_t)((const char*)nl - (mapped + off)) + 1;
        } else {
          const void* nl2 = memchr(mapped + off + bsize, '\n', mapped_len - off - bsize);
          bsize = nl2 ? (size_t)((const char*)nl2 - (mapped + off)) + 1 : mapped_len - off;
        }
      }
      parse_block(mapped + off, bsize);
      off += bsize;
    }
  } else {
    string cur, next, carry, spill;
    // (FERROMIC_INGEST_BLOCK / FERROMIC_INGEST_HEAD shrink the two sizes so that tests cross block borders on small files)
    const size_t kBlock = getenv("FERROMIC_INGEST_BLOCK") ? std::max<size_t>(16, (size_t)atoll(getenv("FERROMIC_INGEST_BLOCK"))) : ((size_t)64 << 20);
    const size_t kHead = getenv("FERROMIC_INGEST_HEAD") ? (size_t)atoll(getenv("FERROMIC_INGEST_HEAD")) : ((size_t)4 << 20);  // room in front of every block for the unfinished line of the previous one
    bool eof = false;
    // the next block is read (and inflated) by a helper thread while this one is parsed; the two buffers are allocated
    // once and a block is never copied: the carried-over partial line is written into the headroom in front of it
    auto fetch = [&r, kBlock, kHead](string* dst) -> size_t { if (dst->size() != kHead + kBlock) dst->resize(kHead + kBlock); return r.read(&(*dst)[kHead], kBlock); };
    size_t next_got = fetch(&next);
    while (!eof) {
      cur.swap(next);
      const size_t got = next_got;
      if (got < kBlock) eof = true;
      const char* bdata;
      size_t bsize;
      if (carry.size() <= kHead) {
        memcpy(&cur[kHead - carry.size()], carry.data(), carry.size());
        bdata = cur.data() + (kHead - carry.size());
        bsize = carry.size() + got;
      } else {  // a line longer than the headroom: the slow way, once
        spill.assign(carry);
        spill.append(cur.data() + kHead, got);
        bdata = spill.data();
        bsize = spill.size();
      }
      std::thread reader;
      if (!eof) reader = std::thread([&] { next_got = fetch(&next); });
      struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) { StageTimer w("    ingest:wait_for_reader"); t.join(); } } } joiner{reader};
      size_t usable = bsize;
      if (!eof) {
        const void* nlp = memrchr(bdata, '\n', bsize);
        if (!nlp) { carry.assign(bdata, bsize); continue; }
        usable = (size_t)((const char*)nlp - bdata) + 1;
        carry.assign(bdata + usable, bsize - usable);
      } else {
        carry.clear();
      }
      if (usable) parse_block(bdata, usable);
    }
  }
  StageTimer tsort("    ingest:sort_and_store");
  std::stable_sort(items.begin(), items.end(), [](const auto& a, const auto& b) {
    if (a.first.position != b.first.position) return a.first.position < b.first.position;
    return a.first.data < b.first.data;  // lexicographic on the flat genotype bytes (process.rs:4397-4405)
  });
  for (auto& it : items) { d.variants.push_back(std::move(it.first)); d.flags.push_back(it.second); }
  return d;
}

// ---- sample-name mapping (process.rs:1192-1333) -----------------------------------------------------
string normalize_sample_name(const string& n) { return (ends_with(n, "_L") || ends_with(n, "_R")) ? n.substr(0, n.size() - 2) : n; }

std::map<string, size_t> map_sample_names_to_indices(const vector<string>& names) {
  std::map<string, size_t> exact;
  std::map<string, std::optional<size_t>> alias;
  for (size_t i = 0; i < names.size(); ++i) {
    exact[names[i]] = i;
    const size_t us = names[i].rfind('_');
    if (us != string::npos) {
      const string suffix = names[i].substr(us + 1);
      auto it = alias.find(suffix);
      if (it == alias.end()) alias[suffix] = i;
      else if (!(it->second && *it->second == i)) it->second = std::nullopt;
    }
  }
  for (auto& kv : alias) if (kv.second && !exact.count(kv.first)) exact[kv.first] = *kv.second;
  return exact;
}


HapList haplotypes_for_group(uint8_t group, const SampleMap& filter, const std::map<string, size_t>& index) {
  HapList out;
  for (auto& kv : filter) {
    auto it = index.find(normalize_sample_name(kv.first));
    if (it == index.end()) continue;
    if (kv.second.first == group) out.push_back({it->second, 0});
    if (kv.second.second == group) out.push_back({it->second, 1});
  }
  return out;
}

}  // namespace fmv
