// writers.cpp — the output surface of run_vcf: output.csv quoting, the per-site FALSTA tracks (process.rs:3740-4003) through the
// run-aware DEFLATE writer or zlib, the TSV headers, and the writer self-checks (--print_formats, --dump_writer_cases, --check_fmt6,
// --bench_tracks).
#include "run_vcf.hpp"

namespace fmv {

// ---- writers --------------------------------------------------------------------------------------------
const char* kCsvHeader[34] = {
    "chr", "region_start", "region_end", "0_sequence_length", "1_sequence_length", "0_sequence_length_adjusted",
    "1_sequence_length_adjusted", "0_segregating_sites", "1_segregating_sites", "0_w_theta", "1_w_theta", "0_pi", "1_pi",
    "0_segregating_sites_filtered", "1_segregating_sites_filtered", "0_w_theta_filtered", "1_w_theta_filtered",
    "0_pi_filtered", "1_pi_filtered", "0_num_hap_no_filter", "1_num_hap_no_filter", "0_num_hap_filter", "1_num_hap_filter",
    "inversion_freq_no_filter", "inversion_freq_filter", "haplotype_overall_fst_wc", "haplotype_between_pop_variance_wc",
    "haplotype_within_pop_variance_wc", "haplotype_num_informative_sites_wc", "hudson_fst_hap_group_0v1",
    "hudson_dxy_hap_group_0v1", "hudson_pi_hap_group_0", "hudson_pi_hap_group_1", "hudson_pi_avg_hap_group_0v1"};

string csv_field(const string& f) {  // csv crate default quoting: only when needed
  if (f.find_first_of(",\"\n\r") == string::npos) return f;
  string o = "\"";
  for (char c : f) { if (c == '"') o += '"'; o += c; }
  return o + "\"";
}
string join(const vector<string>& v, char d, bool csv_quote) {
  string o;
  for (size_t i = 0; i < v.size(); ++i) { if (i) o += d; o += csv_quote ? csv_field(v[i]) : v[i]; }
  return o;
}

void gz_append(const string& path, const string& text) {  // open_append_compressed: one gzip member per call
  gzFile f = gzopen(path.c_str(), "ab");
  if (!f) throw Error("cannot open " + path);
  for (size_t off = 0; off < text.size();) {
    const size_t n = std::min<size_t>(text.size() - off, (size_t)1 << 30);
    if (gzwrite(f, text.data() + off, (unsigned)n) <= 0) { gzclose(f); throw Error("write failed: " + path); }
    off += n;
  }
  gzclose(f);
}

// (the CRC-32, the zlib member writer, the track sinks and the run-aware DEFLATE writer live in deflate_runs.cpp)

// FALSTA tracks of one region: every track is formatted and deflated on its own thread and lands in the file as
// its own gzip member, in track order (the files are multi-member already: one member per region in the
// reference; readers see the same decompressed text).
// `files` = one list of tracks per output file; returns, per file, the gzip members of its tracks in order (empty tracks dropped).
// All tracks form one batch: formatted and deflated on the pool when they are large, inline when the whole region is small (hundreds of
// small regions are compressed by their region workers side by side; waking the pool for 40-kB tracks cost more than deflating them).
std::atomic<unsigned> g_region_workers{1};
vector<vector<string>> compress_tracks(const vector<vector<TrackFn>>& files, size_t approx_tokens) {
  vector<std::pair<size_t, size_t>> jobs;  // (file, track)
  for (size_t f = 0; f < files.size(); ++f) for (size_t t = 0; t < files[f].size(); ++t) jobs.push_back({f, t});
  vector<string> members(jobs.size());
  std::atomic<size_t> next{0};
  auto work = [&](unsigned) {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= jobs.size()) break;
      const TrackFn& track = files[jobs[i].first][jobs[i].second];
      // mostly runs (fewer than one value per eight positions): the run-aware writer; dense: the text through zlib (entropy-coded values).
      // FERROMIC_TRACK_WRITER=zlib | runs forces one of them (tests run both).
      static const char* forced = getenv("FERROMIC_TRACK_WRITER");
      const bool runs = forced ? strcmp(forced, "runs") == 0 : track.records * 8 < track.tokens;
      if (runs) {
        RunDeflateSink sink(track.records >= 48);  // a member with a few dozen values repays the tuned code's 60-byte header
        if (track.write(sink)) members[i] = sink.finish();
      } else {
        const string text = track.text();
        if (!text.empty()) members[i] = gzip_member(text);
      }
    }
  };
  // inline when the tracks are tiny; else the shared pool, however many region workers there are (500 regions of 2-25 kb, 16 CPUs:
  // 2 / 4 / 8 / 16 workers 3.9 / 2.0 / 1.1 / 0.73 ms per region inline, 1.1-1.2 / 0.65-0.69 / 0.63-0.68 / 0.74 through the pool,
  // profiles/r03/run_vcf_tracks_pool_or_inline.jsonl).  FERROMIC_TRACKS_POOL=0 | 1 forces one of them (measurement).
  static const int force = getenv("FERROMIC_TRACKS_POOL") ? atoi(getenv("FERROMIC_TRACKS_POOL")) : -1;
  if (force == 0 || (force < 0 && approx_tokens * jobs.size() < ((size_t)1 << 16))) work(0u);
  else parallel_for((unsigned)std::min<size_t>(worker_threads(), jobs.size()), work);
  vector<vector<string>> out(files.size());
  for (size_t i = 0; i < jobs.size(); ++i) if (!members[i].empty()) out[jobs[i].first].push_back(std::move(members[i]));
  return out;
}
void append_members(const string& path, const vector<string>& members) {
  if (members.empty()) return;
  FILE* f = fopen(path.c_str(), "ab");
  if (!f) throw Error("cannot open " + path);
  for (const string& m : members)
    if (fwrite(m.data(), 1, m.size(), f) != m.size()) { fclose(f); throw Error("write failed: " + path); }
  fclose(f);
}


// One dense FALSTA line: `n` comma-joined tokens, `dflt` everywhere except at the positions present, where the
// LAST record of a position wins (the reference assigns into a Vec in record order).
// Written into a TrackSink: the gaps between records are runs of (comma + default token).
template <class PosAt, class TokenAt>
bool falsta_line(TrackSink& out, const Interval& region, int64_t n, size_t count, PosAt pos_at, TokenAt token_at, const char* dflt,
                 vector<int32_t>& slot) {
  const size_t dl = strlen(dflt);
  string unit(1, ',');
  unit.append(dflt, dl);  // ",0" / ",NA": what a default position adds to a line that has begun
  // `tokens` default positions starting at position `at` of the line
  auto default_run = [&](int64_t at, size_t tokens) {
    if (!tokens) return;
    if (at == 0) { out.text(dflt, dl); --tokens; }
    out.run(unit.data(), unit.size(), tokens);
  };
  string tok;
  // Records in ascending position order (the usual case: variants are sorted): the gaps between them are runs of the default token;
  // equal positions are neighbours, so "the last record wins" is "skip a record whose successor has its position".
  bool ascending = true;
  for (size_t i = 1; i < count && ascending; ++i) ascending = pos_at(i - 1) <= pos_at(i);
  if (ascending) {
    bool any_rec = false;
    int64_t next_k = 0;  // first position of the line not written yet
    for (size_t i = 0; i < count; ++i) {
      const int64_t p = pos_at(i) - 1;
      if (!hal_contains(region, p)) continue;
      any_rec = true;
      if (i + 1 < count && pos_at(i + 1) - 1 == p) continue;
      const int64_t k = p - region.first;
      default_run(next_k, (size_t)(k - next_k));
      tok.clear();
      if (k) tok.push_back(',');
      token_at(tok, i);
      out.text(tok);
      next_k = k + 1;
    }
    if (n > next_k) default_run(next_k, (size_t)(n - next_k));
    out.text("\n", 1);
    return any_rec;
  }
  slot.assign((size_t)n, -1);
  bool any = false;
  for (size_t i = 0; i < count; ++i) {
    const int64_t p = pos_at(i) - 1;
    if (!hal_contains(region, p)) continue;
    slot[(size_t)(p - region.first)] = (int32_t)i;
    any = true;
  }
  for (int64_t k = 0; k < n; ++k) {
    tok.clear();
    if (k) tok.push_back(',');
    if (slot[(size_t)k] < 0) tok.append(dflt, dl);
    else token_at(tok, (size_t)slot[(size_t)k]);
    out.text(tok);
  }
  out.text("\n", 1);
  return any;
}
// is any record of the track inside the region? (a diversity track without one is not written at all)
template <class PosAt>
bool falsta_any(const Interval& region, size_t count, PosAt pos_at) {
  for (size_t i = 0; i < count; ++i) if (hal_contains(region, pos_at(i) - 1)) return true;
  return false;
}

vector<TrackFn> diversity_tracks(const RegionOutput& r) {  // append_diversity_falsta, process.rs:3740-3806
  vector<TrackFn> tracks;
  if (r.diversity.empty()) return tracks;
  const Interval region = from_1based_inclusive(r.region_start1, r.region_end1);
  const int64_t n = hal_len(region);
  if (n > (int64_t)1 << 31 || r.diversity.size() >= (size_t)1 << 31) throw Error("region too long for a dense FALSTA track");
  std::set<int> gids;
  for (auto& d : r.diversity) gids.insert(std::get<3>(d));
  struct Spec { bool filtered; bool is_pi; const char* prefix; };
  static const Spec specs[4] = {{false, true, "unfiltered_pi_"}, {false, false, "unfiltered_theta_"}, {true, true, "filtered_pi_"}, {true, false, "filtered_theta_"}};
  for (int g : gids)
    for (const Spec& sp : specs)
    {
      TrackFn track;
      track.tokens = (size_t)n;
      for (auto& d : r.diversity) track.records += std::get<3>(d) == g && std::get<4>(d) == sp.filtered;
      track.write = [&r, region, n, g, sp](TrackSink& out) -> bool {
        vector<size_t> sel;  // records of this (group, filter) in record order
        for (size_t i = 0; i < r.diversity.size(); ++i)
          if (std::get<3>(r.diversity[i]) == g && std::get<4>(r.diversity[i]) == sp.filtered) sel.push_back(i);
        auto pos_at = [&](size_t i) { return std::get<0>(r.diversity[sel[i]]); };
        if (!falsta_any(region, sel.size(), pos_at)) return false;
        out.text(">" + string(sp.prefix) + "chr_" + r.seqname + "_start_" + std::to_string(r.region_start1) + "_end_" +
                 std::to_string(r.region_end1) + "_group_" + std::to_string(g) + "\n");
        vector<int32_t> slot;
        falsta_line(out, region, n, sel.size(), pos_at,
                    [&](string& o, size_t i) { const auto& d = r.diversity[sel[i]]; falsta_div_value(o, sp.is_pi ? std::get<1>(d) : std::get<2>(d)); }, "0", slot);
        return true;
      };
      tracks.push_back(std::move(track));
    }
  return tracks;
}

vector<TrackFn> fst_tracks(const RegionOutput& r) {  // append_fst_falsta, process.rs:3809-4003
  vector<TrackFn> tracks;
  if (r.wc_sites.empty() && r.hudson_sites.empty()) return tracks;
  const Interval region = from_1based_inclusive(r.region_start1, r.region_end1);
  const int64_t n = hal_len(region);
  if (n > (int64_t)1 << 31 || r.wc_sites.size() >= (size_t)1 << 31 || r.hudson_sites.size() >= (size_t)1 << 31)
    throw Error("region too long for a dense FALSTA track");
  const string suffix = "chr_" + r.seqname + "_start_" + std::to_string(r.region_start1) + "_end_" + std::to_string(r.region_end1);
  auto add = [&](const string& header, size_t count, auto getter) {  // getter(i) -> (position, value); no std::function on the per-record path
    TrackFn track;
    track.tokens = (size_t)n;
    track.records = count;
    track.write = [=](TrackSink& out) -> bool {
      out.text(">" + header + "_" + suffix + "\n");
      vector<int32_t> slot;
      falsta_line(out, region, n, count, [&](size_t i) { return getter(i).first; }, [&](string& o, size_t i) { falsta_fst_value(o, getter(i).second); }, "NA", slot);
      return true;
    };
    tracks.push_back(std::move(track));
  };
  if (!r.wc_sites.empty()) {
    const vector<WcSite>* w = &r.wc_sites;
    add("haplotype_overall_fst_summary", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].overall_fst); });
    add("haplotype_overall_fst_numerator", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].overall_num); });
    add("haplotype_overall_fst_denominator", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].overall_den); });
    add("haplotype_0v1_pairwise_fst_summary", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].pair_fst); });
    add("haplotype_0v1_pairwise_fst_numerator", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].pair_num); });
    add("haplotype_0v1_pairwise_fst_denominator", w->size(), [w](size_t i) { return std::make_pair((*w)[i].pos1, (*w)[i].pair_den); });
  }
  if (!r.hudson_sites.empty()) {
    const auto* h = &r.hudson_sites;
    add("hudson_pairwise_fst_hap_0v1", h->size(), [h](size_t i) { return std::make_pair(std::get<0>((*h)[i]), std::get<1>((*h)[i])); });
    add("hudson_pairwise_fst_hap_0v1_numerator", h->size(), [h](size_t i) { return std::make_pair(std::get<0>((*h)[i]), std::get<2>((*h)[i])); });
    add("hudson_pairwise_fst_hap_0v1_denominator", h->size(), [h](size_t i) { return std::make_pair(std::get<0>((*h)[i]), std::get<3>((*h)[i])); });
  }
  return tracks;
}

const char* kHudsonTsvHeader = "chr\tregion_start_0based\tregion_end_0based\tpop1_id_type\tpop1_id_name\tpop2_id_type\tpop2_id_name\tDxy\tpi_pop1\tpi_pop2\tpi_xy_avg\tFST\n";  // process.rs:1576-1590
const char* kWcTsvHeader = "chr\tregion_start_1based\tregion_end_1based\tcomparison_type\tpop1\tpop2\tfst\tnumerator_a\tdenominator_a_plus_b\tinformative_sites\n";  // process.rs:1628-1650

// --print_formats (no GPU): the header lines of every output file and the FALSTA records the writers produce for one tiny
// made-up region, so that the output surface can be pinned against the reference's committed exemplars
// (data/output.csv, data/FST_data.tsv, data/per_site_diversity_output.falsta.gz) on a machine without a GPU.
int print_formats() {
  vector<string> header(kCsvHeader, kCsvHeader + 34);
  printf("output.csv\t%s\n", join(header, ',', true).c_str());
  printf("hudson_fst_results.tsv\t%s", kHudsonTsvHeader);
  printf("wc_fst_results.tsv\t%s", kWcTsvHeader);
  RegionOutput r;
  r.seqname = "1";
  r.region_start1 = 5;
  r.region_end1 = 12;
  for (int g = 0; g < 2; ++g)
    for (int f = 0; f < 2; ++f) {
      r.diversity.push_back({6, 0.289855, 0.267788, g, f != 0});
      r.diversity.push_back({9, NAN, NAN, g, f != 0});
      r.diversity.push_back({11, 0.0, 0.0, g, f != 0});
    }
  r.wc_sites.push_back({6, 0.5, 0.25, 0.5, 0.5, 0.25, 0.5});
  r.wc_sites.push_back({9, NAN, 0.0, 0.0, INFINITY, 1.0, 0.0});
  r.hudson_sites.push_back({6, 1.0, 1.0, 1.0});
  r.hudson_sites.push_back({7, -0.5, -0.5, 1.0});
  r.hudson_sites.push_back({11, NAN, 0.0, 0.0});
  for (auto& t : diversity_tracks(r)) printf("per_site_diversity_output.falsta\t%s", t.text().c_str());
  for (auto& t : fst_tracks(r)) printf("per_site_fst_output.falsta\t%s", t.text().c_str());
  return 0;
}

// The adversarial tracks of --check_writers / --dump_writer_cases: both default tokens, runs of every length around 258 and its multiples, dense
// and sparse, records at the first and the last position, unsorted records, empty lines.  `next` is the caller's xorshift stream; the SAME
// track is written twice (text sink, then run-aware sink) by replaying the record list drawn the first time.
struct AdversarialTrack { const char* dflt; int64_t n; vector<std::pair<int64_t, double>> recs; };
template <class Next> void adversarial_track(int rep, Next& next, TrackSink& out, bool replay = false) {
  static thread_local AdversarialTrack t;
  if (!replay) {
    t.dflt = (rep & 1) ? "NA" : "0";
    t.n = rep < 8 ? rep : (int64_t)(next() % (rep % 7 == 0 ? 700000 : 3000));
    t.recs.clear();
    const uint64_t gap = 1 + next() % (rep % 5 == 0 ? 3 : 600);
    for (int64_t p = (int64_t)(next() % 3); p < t.n; p += 1 + (int64_t)(next() % gap)) t.recs.push_back({1001 + p, (double)(next() >> 11) / 9007199254740992.0});
    if (rep % 11 == 0 && t.n > 0) { t.recs.insert(t.recs.begin(), {1001, 0.5}); t.recs.push_back({1000 + t.n, -0.25}); }
    for (size_t special : {(size_t)258, (size_t)259, (size_t)260, (size_t)261, (size_t)516, (size_t)517, (size_t)130})  // gaps that hit the match-length edges
      if (rep % 13 == 0 && (int64_t)(special * 3) < t.n) t.recs.push_back({t.recs.empty() ? 1001 + (int64_t)special : t.recs.back().first + (int64_t)special, 1.0});
    if (rep % 17 != 0) std::sort(t.recs.begin(), t.recs.end()); else if (t.recs.size() > 2) std::swap(t.recs[0], t.recs[t.recs.size() / 2]);  // (the unsorted path too)
  }
  const Interval region{1000, 1000 + t.n};
  out.text(string(">header_") + t.dflt + "\n");
  vector<int32_t> slot;
  falsta_line(out, region, t.n, t.recs.size(), [&](size_t i) { return t.recs[i].first; }, [&](string& o, size_t i) { falsta_fst_value(o, t.recs[i].second); }, t.dflt, slot);
}

// --dump_writer_cases DIR [N] (no GPU): N adversarial tracks (default 60), each as DIR/case_<k>.txt (the text), .runs.gz (the run-aware writer's
// member, fixed and tuned code sets alternating) and .zlib.gz (the same text through zlib level 1): tests/test_output_formats_cpu.py inflates
// both members with zlib AND Python's gzip module and compares all of them with the text.
int dump_writer_cases(const string& dir, int count) {
  mkdirs(dir);
  uint64_t state = 0x243F6A8885A308D3ull;
  auto next = [&] { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
  auto put = [&](const string& path, const string& bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size()) throw Error("cannot write " + path);
    fclose(f);
  };
  for (int k = 0; k < count; ++k) {
    const int rep = k < 20 ? k : 7 * k + 3;  // the tiny regions, then a spread over the generator's cases (multiples of 7: up to 700 000 positions)
    TextSink text;
    RunDeflateSink runs(k % 2 == 0);
    adversarial_track(rep, next, text);
    adversarial_track(rep, next, runs, true);
    const string base = dir + "/case_" + std::to_string(k);
    put(base + ".txt", text.out);
    put(base + ".runs.gz", runs.finish());
    put(base + ".zlib.gz", gzip_member(text.out));
  }
  printf("%d writer cases in %s\n", count, dir.c_str());
  return 0;
}

// --check_writers N (no GPU; self-checks of the output writers): the CRC-32 against zlib's, the run-aware gzip writer against the text writer
// through zlib's inflate, and fmt6 against printf's %.6f on N pseudo-random doubles of every magnitude a statistic can take, exact ties
// (k / 128 and their neighbours one ulp away), values next to a carry (0.9999995, 9.9999995, ...), zeros and subnormals.
int check_fmt6(size_t n) {
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto next = [&] { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
  size_t bad = 0, checked = 0;
  auto check = [&](double v) {
    ++checked;
    const string a = fmt6(v), b = fmt6_printf(v);
    if (a != b && bad++ < 10) fprintf(stderr, "fmt6 mismatch: %.17g -> '%s' vs printf '%s'\n", v, a.c_str(), b.c_str());
  };
  const double specials[] = {0.0, -0.0, 5e-7, 4.9999999999999998e-7, 5.0000000000000004e-7, 1.5e-6, 2.5e-6, 0.9999995, 0.99999949999999994, 9.9999995,
                             99.9999995, 1e-300, -1e-300, 4.9e-324, 1e15, 999999999999999.9, 123456789012345.67, 0.1, 0.2, 0.3, 1.0 / 3.0, 2.0 / 3.0,
                             1e-6, 1e-7, 0.000001499999999, 0.0078125, -0.0078125, 0.0234375, NAN, INFINITY, -INFINITY, 1e16, -1e22};
  for (double v : specials) { check(v); check(-v); check(std::nextafter(v, 1e300)); check(std::nextafter(v, -1e300)); }
  for (int k = 0; k < 200000; ++k) {  // exact ties of the sixth decimal and their neighbours
    const double t = (double)k / 128.0;
    check(t); check(std::nextafter(t, 1e300)); check(std::nextafter(t, -1e300)); check(-t);
  }
  for (size_t i = 0; i < n; ++i) {
    const uint64_t r = next();
    double v;
    switch (r % 5) {
      case 0: v = (double)(next() >> 11) / 9007199254740992.0; break;                                  // [0, 1)
      case 1: v = ((double)(next() >> 11) / 9007199254740992.0) * std::pow(10.0, (double)((int)(r >> 8 & 31) - 12)); break;  // 1e-12 .. 1e19
      case 2: { uint64_t b = next(); b = (b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (r >> 8) % 80) << 52); memcpy(&v, &b, 8); break; }  // 2^-40 .. 2^40
      case 3: v = (double)((int64_t)(next() % 2000000000) - 1000000000) / 1000000.0 + ((r >> 8 & 1) ? 5e-7 : 0.0); break;            // near six-decimal grid points / half-way points
      default: { uint64_t b = next(); memcpy(&v, &b, 8); break; }                                                                        // any bit pattern
    }
    check(v);
  }
  printf("fmt6: %zu values checked against printf, %zu differ\n", checked, bad);
  size_t crc_bad = 0, crc_checked = 0;
  vector<uint8_t> buf(70000);
  for (int rep = 0; rep < 400; ++rep) {
    for (auto& b : buf) b = (uint8_t)(next() >> 56);
    const size_t off = (size_t)(next() % 9), len = (size_t)(next() % (buf.size() - 8));
    ++crc_checked;
    if (crc32_slice8(buf.data() + off, len) != (uint32_t)crc32(0L, buf.data() + off, (uInt)len)) ++crc_bad;
  }
  const string member = gzip_member(string(100000, 'x') + "tail");
  printf("crc32: %zu buffers checked against zlib, %zu differ; a 100 004-byte member is %zu bytes\n", crc_checked, crc_bad, member.size());
  // the run-aware gzip writer against the text writer: random tracks (both default tokens, runs of every length around 258 and its multiples,
  // dense and sparse, records at the first and the last position, empty lines), each member inflated by zlib - which also verifies the
  // CRC-32 and the length of the trailer - and compared with the text
  auto inflate_member = [](const string& m, string& out) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return false;
    out.clear();
    out.resize(1 << 16);
    z.next_in = (Bytef*)m.data();
    z.avail_in = (uInt)m.size();
    size_t off = 0;
    int rc;
    do {
      if (out.size() - off < (1 << 15)) out.resize(out.size() * 2);
      z.next_out = (Bytef*)&out[off];
      z.avail_out = (uInt)(out.size() - off);
      rc = inflate(&z, Z_NO_FLUSH);
      off = out.size() - z.avail_out;
    } while (rc == Z_OK);
    inflateEnd(&z);
    out.resize(off);
    return rc == Z_STREAM_END && z.avail_in == 0;
  };
  size_t tracks_checked = 0, tracks_bad = 0;
  for (int rep = 0; rep < 600; ++rep) {
    TextSink text;
    RunDeflateSink runs(rep % 2 == 0);  // both code sets
    adversarial_track(rep, next, text);
    adversarial_track(rep, next, runs, /*replay=*/true);
    const string member2 = runs.finish();
    string back;
    ++tracks_checked;
    if (!inflate_member(member2, back) || back != text.out) {
      if (tracks_bad++ < 5) fprintf(stderr, "run-aware member differs: rep %d, text %zu bytes, inflated %zu\n", rep, text.out.size(), back.size());
    }
    string back2;
    if (!inflate_member(gzip_member(text.out), back2) || back2 != text.out) ++tracks_bad;
  }
  printf("run-aware gzip writer: %zu tracks inflated by zlib and compared with the text, %zu differ\n", tracks_checked, tracks_bad);
  return bad || crc_bad || tracks_bad ? 1 : 0;
}

// --bench_tracks [variants [length]] (no GPU): formats and deflates the tracks of a made-up 15-kb region (or `length` bp) with 120 variants (or `variants`), 500 times on one thread;
// what the writers cost per small region.
int bench_tracks(int variants, int length) {  // 120 = a variant every 125 bp; 3 750 = every 4 bp (tools/run_vcf_many_regions.py's cohort)
  RegionOutput r;
  r.seqname = "1";
  r.region_start1 = 1000;
  r.region_end1 = 1000 + length - 1;
  const int step = std::max(1, length / std::max(variants, 1));
  for (int g = 0; g < 2; ++g)
    for (int f = 0; f < 2; ++f)
      for (int i = 0; i < variants; ++i) r.diversity.push_back({1000 + step * i, 0.289855 + i * 1e-5, 0.267788 + i * 1e-6, g, f != 0});
  for (int i = 0; i < variants; ++i) {
    r.wc_sites.push_back({1000 + step * i, 0.5 + i * 1e-6, 0.25, 0.5, 0.5, 0.25 + i * 1e-6, 0.5});
    r.hudson_sites.push_back({1000 + step * i, 0.25 + i * 1e-6, 0.125, 0.5});
  }
  const auto t0 = std::chrono::steady_clock::now();
  size_t bytes = 0, members = 0;
  const int reps = length > 100000 ? 20 : 500;
  for (int rep = 0; rep < reps; ++rep)
    for (auto& file : compress_tracks({diversity_tracks(r), fst_tracks(r)}, (size_t)length)) for (auto& m : file) { bytes += m.size(); ++members; }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("tracks of one region: %.3f ms, %zu members, %zu bytes\n", ms, members / (size_t)reps, bytes / (size_t)reps);
  return 0;
}

}  // namespace fmv
