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
// the autocorrelation kernel's step without its LDS reads -- two f32 -> f64 conversions feeding an MFMA (CVT) or
// the MFMA alone -- as a grid of workgroups of four waves, timed by events around the whole launch.  (Timing one
// workgroup from inside, as a first version of this probe did, reports that workgroup's share of an unevenly filled
// chip: 3.4 ns per instruction where the launch as a whole says 8.)
template <bool CVT>
__global__ void __launch_bounds__(256) steps(double* out, int iters, float x, float y) {
  double acc = threadIdx.x;
  float xa = x + threadIdx.x, ya = y;
  double a = xa, b = ya;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (CVT) {
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(xa));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(ya));
      }
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
  }
  if (acc == 12345.678) out[1] = acc;
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
  // 3. cost: grids of 256 .. 6144 workgroups (1 .. 8 waves per SIMD resident, then several rounds), 4352 steps per wave
  for (int cvt = 0; cvt < 2; ++cvt)
    for (int grid : {256, 512, 1024, 2048, 6144}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (cvt) hipLaunchKernelGGL((steps<true>), dim3(grid), dim3(256), 0, 0, dD, 272, 1.5f, 0.5f);
        else hipLaunchKernelGGL((steps<false>), dim3(grid), dim3(256), 0, 0, dD, 272, 1.5f, 0.5f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::printf("%s, %4d workgroups x 4 waves x 4352 steps: %7.1f us = %.2f ns per step per SIMD\n",
                  cvt ? "2 x v_cvt_f64_f32 + v_mfma_f64_4x4x4" : "v_mfma_f64_4x4x4 alone", grid, ms * 1e3,
                  ms * 1e6 / (grid * 4.0 * 4352 / 1024));
    }
  return 0;
}
