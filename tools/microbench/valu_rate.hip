// valu_rate.hip -- how many wave64 VALU instructions per cycle does one SIMD of gfx950 retire with
// 1..8 resident waves?  f64 fma, f32 fma, v_mad_i32_i24, v_bitop3; 9 independent chains per wave.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) chains(double* out, int iters, double seed) {
  double a[9];
  float f[9];
  int m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    a[i] = seed + i + threadIdx.x;
    f[i] = (float)a[i];
    m[i] = (int)a[i];
  }
  const double x = seed * 1.0000001;
  const float xf = (float)x;
  const int xi = (int)seed | 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (KIND == 0) a[i] = __builtin_fma(a[i], x, a[i]);
        if (KIND == 1) f[i] = __builtin_fmaf(f[i], xf, f[i]);
        if (KIND == 2) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(m[i]) : "v"(xi), "v"(m[i]));
        if (KIND == 3) m[i] = __builtin_amdgcn_bitop3_b32(m[i], xi, m[(i + 1) % 9], 0x96);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += a[i] + f[i] + m[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name) {
  double* out;
  hipMalloc(&out, sizeof(double) * 64 * 256 * 4 * 8);
  const int iters = 4000;
  for (int wps = 1; wps <= 8; ++wps) {
    const int blocks = 256 * 4 * wps;  // one 64-thread block per wave slot
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    chains<KIND><<<blocks, 64>>>(out, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<KIND><<<blocks, 64>>>(out, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)wps * iters * 72;
    const double cycles = ms * 1e-3 * 2.4e9;
    std::printf("%-12s waves/SIMD %d: %.3f ms, %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", name, wps, ms,
                cycles / insts_per_simd);
  }
  hipFree(out);
}

int main() {
  run<0>("v_fma_f64");
  run<1>("v_fma_f32");
  run<2>("v_mad_i32_i24");
  run<3>("v_bitop3_b32");
  return 0;
}
