// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (e2m1) operands as an exact small-integer Gram engine.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
// rows[r][k] packed fp4: byte (k/2), low nibble = even k.  A = rowsA (16 x K), B = rowsB (16 x K); out[i][j] = sum_k A[i][k]*B[j][k]
__global__ void gram(const unsigned char* a, const unsigned char* b, int K, float* out) {
  const int lane = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 128) {
    const v4i x = *reinterpret_cast<const v4i*>(a + (size_t)(lane & 15) * (K / 2) + k0 / 2 + (lane >> 4) * 16);
    const v4i y = *reinterpret_cast<const v4i*>(b + (size_t)(lane & 15) * (K / 2) + k0 / 2 + (lane >> 4) * 16);
    v8i A = {x[0], x[1], x[2], x[3], 0, 0, 0, 0};
    v8i B = {y[0], y[1], y[2], y[3], 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}
int main() {
  const int K = 1 << 20;  // 1 M: sums up to 16 * 2^20 = 2^24
  static const unsigned char code[5] = {0, 2, 4, 5, 6};  // e2m1 codes of 0, 1, 2, 3, 4
  std::vector<unsigned char> va(16 * K), vb(16 * K), pa(16 * K / 2), pb(16 * K / 2);
  srand(7);
  for (int maxv = 2; maxv <= 4; maxv += 2) {
    for (size_t i = 0; i < va.size(); ++i) { va[i] = rand() % (maxv + 1); vb[i] = rand() % (maxv + 1); }
    for (int r = 0; r < 16; ++r) for (int k = 0; k < K; ++k) { va[(size_t)r * K + k] = (r == 3) ? maxv : va[(size_t)r * K + k]; }  // a worst-case row
    for (int r = 0; r < 16; ++r) for (int k = 0; k < K; ++k) { vb[(size_t)r * K + k] = (r == 5) ? maxv : vb[(size_t)r * K + k]; }
    for (size_t i = 0; i < pa.size(); ++i) { pa[i] = code[va[2 * i]] | (code[va[2 * i + 1]] << 4); pb[i] = code[vb[2 * i]] | (code[vb[2 * i + 1]] << 4); }
    unsigned char *da, *db; float* dout;
    hipMalloc(&da, pa.size()); hipMalloc(&db, pb.size()); hipMalloc(&dout, 256 * 4);
    hipMemcpy(da, pa.data(), pa.size(), hipMemcpyHostToDevice); hipMemcpy(db, pb.data(), pb.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gram, dim3(1), dim3(64), 0, 0, da, db, K, dout);
    float out[256];
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    int bad = 0; long long worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      long long e = 0;
      for (int k = 0; k < K; ++k) e += (long long)va[(size_t)i * K + k] * vb[(size_t)j * K + k];
      if ((long long)out[i * 16 + j] != e) { if (bad < 5) printf("maxv %d mismatch (%d,%d): got %.1f exp %lld\n", maxv, i, j, out[i * 16 + j], e); ++bad; }
      if (e > worst) worst = e;
    }
    printf("maxv %d: %d mismatches of 256, largest sum %lld (2^24 = 16777216)\n", maxv, bad, worst);
    hipFree(da); hipFree(db); hipFree(dout);
  }
  return 0;
}
