// host_pack_check.cpp — sanitizer check of the host-side packer (host_pack.hpp, what fmh_matrix_create runs on every upload):
// random u8 matrices of ragged widths, with and without a missing bitset, alleles inside and above the plane range, packed by
// pack_rows_host in row ranges (as upload.hip's threads do) and compared bit for bit with a column-by-column restatement.
// Built by `make asan` with -fsanitize=address,undefined -fno-sanitize-recover=all; exits 0 and prints one line when every case agrees.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "host_pack.hpp"

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 400;
  std::mt19937_64 rng(20251004);
  size_t checked = 0;
  for (int c = 0; c < cases; ++c) {
    const size_t rows = 1 + rng() % 40, columns = 1 + rng() % 300;
    const int nplanes = 1 + (int)(rng() % 3);
    const bool with_missing = rng() % 2 == 0;
    const unsigned max_allele = (1u << nplanes) - 1;
    const bool overflow_wanted = rng() % 5 == 0;
    const size_t pitch = ((columns + 7) / 8 + 15) / 16 * 16, total = rows * columns;
    // exact-size heap blocks: a read or write one byte past the matrix, the bitset or a plane is an ASan report
    std::vector<uint8_t> data(total);
    for (auto& v : data) v = (uint8_t)(rng() % (max_allele + 1));
    std::vector<uint64_t> missing((total + 63) / 64, 0);
    if (with_missing)
      for (size_t i = 0; i < total; ++i)
        if (rng() % 7 == 0) missing[i >> 6] |= 1ull << (i & 63);
    bool expect_overflow = false;
    if (overflow_wanted) {
      const size_t at = rng() % total;
      data[at] = (uint8_t)(max_allele + 1 + rng() % 3);
      expect_overflow = !(with_missing && ((missing[at >> 6] >> (at & 63)) & 1));  // a missing entry's value is never looked at
    }
    const int planes_total = nplanes + (with_missing ? 1 : 0);
    std::vector<uint8_t> dst((size_t)planes_total * rows * pitch, 0xAB);
    bool overflow = false;
    const size_t cut = rng() % (rows + 1);  // two row ranges, like two packer threads
    overflow |= fmh_host::pack_rows_host(data.data(), with_missing ? missing.data() : nullptr, columns, total, 0, cut, nplanes, with_missing, dst.data(), 0, rows, pitch);
    overflow |= fmh_host::pack_rows_host(data.data(), with_missing ? missing.data() : nullptr, columns, total, cut, rows, nplanes, with_missing, dst.data(), 0, rows, pitch);
    if (overflow != expect_overflow) { fprintf(stderr, "case %d: overflow flag %d, expected %d\n", c, (int)overflow, (int)expect_overflow); return 1; }
    for (size_t r = 0; r < rows; ++r)
      for (size_t b = 0; b < pitch * 8; ++b) {
        const bool inside = b < columns;
        const size_t idx = r * columns + b;
        const bool miss = inside && with_missing && ((missing[idx >> 6] >> (idx & 63)) & 1);
        for (int k = 0; k < planes_total; ++k) {
          const int got = (dst[((size_t)k * rows + r) * pitch + (b >> 3)] >> (b & 7)) & 1;
          int want = 0;
          if (inside) want = k < nplanes ? ((data[idx] >> k) & 1) : (miss ? 0 : 1);
          // value bits of a missing entry are whatever the row holds (the sweeps AND them with the called plane; unpack masks them)
          if (inside && k < nplanes && miss) continue;
          if (got != want) { fprintf(stderr, "case %d: row %zu column %zu plane %d: %d, expected %d\n", c, r, b, k, got, want); return 1; }
          ++checked;
        }
      }
  }
  printf("host_pack_check: %d cases, %zu bits ok\n", cases, checked);
  return 0;
}
