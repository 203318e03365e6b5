// mfma_f64_order.hip -- is v_mfma_f64_16x16x4_f64 chained through C the sequential fma chain
//   acc = fma(A[i][k], B[k][j], acc), k ascending,
// bit for bit?  (What an MFMA formulation of X1's Gram matrix would need)
// Random operands with exponents spread over +-20 binades so that every rounding matters; K = 64 steps = 16 chained
// instructions; also checked: the four k of one instruction in descending order, and a pairwise tree.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void mfma_chain(const double* A, const double* B, double* D, int K) {  // A[16][K], B[K][16], D[16][16]
  const int l = threadIdx.x;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const double a = A[(l % 16) * K + k0 + l / 16];
    const double b = B[(k0 + l / 16) * 16 + l % 16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  // (output register r of lane l is row 4 r + l / 16, column l % 16 -- the other guess, 4 (l / 16) + r, matches the
  // sequential chain on exactly a quarter of the outputs)
  for (int r = 0; r < 4; ++r) D[(4 * r + l / 16) * 16 + l % 16] = acc[r];
}

int main() {
  const int K = 64, TRIALS = 2000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> mant(-1.0, 1.0);
  std::uniform_int_distribution<int> ex(-20, 20);
  double *dA, *dB, *dD;
  hipMalloc(&dA, 16 * K * 8);
  hipMalloc(&dB, 16 * K * 8);
  hipMalloc(&dD, 256 * 8);
  long seq_ok = 0, desc_ok = 0, tree_ok = 0, total = 0;
  for (int t = 0; t < TRIALS; ++t) {
    std::vector<double> A(16 * K), B(16 * K), D(256);
    for (auto& v : A) v = std::ldexp(mant(rng), ex(rng));
    for (auto& v : B) v = std::ldexp(mant(rng), ex(rng));
    if (t % 3 == 0)  // integers of 24 bits (the flat part of a window): sums stay exact for a while, then round
      for (int i = 0; i < 16 * K; ++i) { A[i] = std::floor(A[i] * 8388608.0); B[i] = std::floor(B[i] * 8388608.0); }
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0.0, d = 0.0, tr = 0.0;
        for (int k = 0; k < K; ++k) s = std::fma(A[i * K + k], B[k * 16 + j], s);
        for (int k0 = 0; k0 < K; k0 += 4)
          for (int k = 3; k >= 0; --k) d = std::fma(A[i * K + k0 + k], B[(k0 + k) * 16 + j], d);
        for (int k0 = 0; k0 < K; k0 += 4) {
          const double p0 = std::fma(A[i * K + k0], B[k0 * 16 + j], A[i * K + k0 + 1] * B[(k0 + 1) * 16 + j]);
          const double p1 = std::fma(A[i * K + k0 + 2], B[(k0 + 2) * 16 + j], A[i * K + k0 + 3] * B[(k0 + 3) * 16 + j]);
          tr = tr + (p0 + p1);
        }
        const double g = D[i * 16 + j];
        seq_ok += std::memcmp(&g, &s, 8) == 0;
        desc_ok += std::memcmp(&g, &d, 8) == 0;
        tree_ok += std::memcmp(&g, &tr, 8) == 0;
        ++total;
      }
  }
  std::printf("v_mfma_f64_16x16x4_f64, K = %d chained through C, %ld outputs:\n", K, total);
  std::printf("  == sequential fma chain, k ascending : %ld (%.4f %%)\n", seq_ok, 100.0 * seq_ok / total);
  std::printf("  == fma chain, k descending inside an instruction: %ld (%.4f %%)\n", desc_ok, 100.0 * desc_ok / total);
  std::printf("  == pairwise tree inside an instruction: %ld (%.4f %%)\n", tree_ok, 100.0 * tree_ok / total);
  return 0;
}
