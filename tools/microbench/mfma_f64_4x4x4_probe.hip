// mfma_f64_4x4x4_probe.hip -- v_mfma_f64_4x4x4_4b_f64: which lane holds which element of A, B and D of which
// block, is the K accumulation the sequential fma chain, and what does a dependent chain of them cost?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ void one(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
__global__ void chain(const double* A, const double* B, double* D, int K4) {  // A, B: [K4][64] per-lane operands
  const int l = threadIdx.x;
  double acc = 0.0;
  for (int m = 0; m < K4; ++m) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[m * 64 + l], B[m * 64 + l], acc, 0, 0, 0);
  D[l] = acc;
}
__global__ void __launch_bounds__(256) timing(double* out, int iters, double x, double y) {
  double acc = threadIdx.x;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc, 0, 0, 0);
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 12345.678) out[1] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (double)(r1 - r0);
}

// one step of the autocorrelation kernel without its LDS reads: two f32 -> f64 conversions feeding an MFMA
__global__ void __launch_bounds__(256) mixed(double* out, int iters, float x, float y) {
  double acc = threadIdx.x;
  float xa = x + threadIdx.x, ya = y;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      double a, b;
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(xa));
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(ya));
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 12345.678) out[1] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (double)(r1 - r0);
}

int main() {
  double *dA, *dB, *dD;
  hipMalloc(&dA, 64 * 64 * 8);
  hipMalloc(&dB, 64 * 64 * 8);
  hipMalloc(&dD, 64 * 8);
  // 1. layout: A one-hot at lane la (value 1), B = lane index + 1 everywhere -> D lanes that become non-zero and
  //    their values tell which B lanes pair with A lane la
  std::printf("A lane -> (D lane : B lane) pairs\n");
  for (int la = 0; la < 64; ++la) {
    std::vector<double> A(64, 0.0), B(64), D(64);
    A[la] = 1.0;
    for (int l = 0; l < 64; ++l) B[l] = l + 1;
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    std::printf("A%2d:", la);
    for (int l = 0; l < 64; ++l)
      if (D[l] != 0.0) std::printf(" D%d:B%d", l, (int)D[l] - 1);
    std::printf("\n");
  }
  // 2. accumulation order with the layout read off part 1 -- A_b[i][k]: lane 16 k + 4 b + i, B_b[k][j]: lane 16 k + 4 b + j,
  //    D_b[i][j]: lane 16 i + 4 b + j -- a 64-instruction chain against the sequential fma chain
  {
    const int K4 = 64;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> mant(-1.0, 1.0);
    std::uniform_int_distribution<int> ex(-20, 20);
    long ok = 0, tot = 0;
    for (int t = 0; t < 500; ++t) {
      std::vector<double> A(K4 * 64), B(K4 * 64), D(64);
      for (auto& v : A) v = std::ldexp(mant(rng), ex(rng));
      for (auto& v : B) v = std::ldexp(mant(rng), ex(rng));
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, K4);
      hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
      for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int m = 0; m < K4; ++m)
              for (int k = 0; k < 4; ++k) s = std::fma(A[m * 64 + 16 * k + 4 * b + i], B[m * 64 + 16 * k + 4 * b + j], s);
            const double g = D[16 * i + 4 * b + j];
            ok += std::memcmp(&g, &s, 8) == 0;
            ++tot;
          }
    }
    std::printf("chain of %d instructions == sequential fma (k ascending) under the layout hypothesis: %ld of %ld\n", K4, ok, tot);
  }
  // 3. cost of a dependent chain, 1 / 2 / 4 waves per SIMD
  for (int w : {1, 2, 4}) {
    const int iters = 20000;
    double h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(timing, dim3(256 * w), dim3(256), 0, 0, dD, iters, 1.0000001, 0.5);
      hipDeviceSynchronize();
      hipMemcpy(&h, dD, 8, hipMemcpyDeviceToHost);
    }
    const double ns = h * 10.0;  // 100 MHz ticks
    std::printf("%d wave(s)/SIMD: %.2f ns per dependent v_mfma_f64_4x4x4 per wave, %.2f ns per instruction per SIMD\n", w,
                ns / (iters * 16.0), ns / (iters * 16.0 * w));
  }
  for (int w : {1, 2, 4, 7, 8}) {
    const int iters = 20000;
    double h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(mixed, dim3(256 * w), dim3(256), 0, 0, dD, iters, 1.5f, 0.5f);
      hipDeviceSynchronize();
      hipMemcpy(&h, dD, 8, hipMemcpyDeviceToHost);
    }
    const double ns = h * 10.0;
    std::printf("%d wave(s)/SIMD: 2 x v_cvt_f64_f32 + v_mfma_f64_4x4x4: %.2f ns per step per wave, %.2f ns per step per SIMD\n", w,
                ns / (iters * 16.0), ns / (iters * 16.0 * w));
  }
  return 0;
}
