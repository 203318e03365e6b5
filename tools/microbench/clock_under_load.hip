// clock_under_load.hip -- the shader clock the chip actually sustains while every SIMD runs one kind of
// instruction: s_memtime (shader cycles) against s_memrealtime (100 MHz) over a few milliseconds of
// (a) v_fma_f64, (b) v_mad_i64_i32, (c) v_add_u32, with 1 / 2 / 4 waves per SIMD.  The peak rates of
// MI355X_MICROARCH.md are quoted at 2.4 GHz; a kernel made of f64 fma is priced against what this prints.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(256) load(unsigned long long* out, int iters, int seed) {
  double d[8];
  int m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = seed * (i + 3) + threadIdx.x;
    d[i] = (double)m[i] * 1.25;
  }
  const double xd = 1.0000001, yd = 0.5;
  const int x = seed | 0x10003;
  const unsigned long long c0 = __builtin_readcyclecounter();        // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();    // 100 MHz
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(xd), "v"(yd));
        if (KIND == 1) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(m[i]) : "vcc");
        if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(m[i]) : "v"(x));
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i] + m[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = (unsigned long long)s;
  }
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  unsigned long long* d;
  hipMalloc(&d, 64);
  const int blocks = 256 * waves_per_simd;  // 4 waves per block: one per SIMD
  const int iters = 60000;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((load<KIND>), dim3(blocks), dim3(256), 0, 0, d, iters, 3 + rep);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    const double insts = (double)iters * 64.0;
    std::printf("%-14s %d wave(s)/SIMD: %.2f ms, s_memtime/s_memrealtime -> %.0f MHz; %.2f shader cycles and %.2f ns per wave instruction per SIMD\n",
                name, waves_per_simd, ms, mhz, (double)h[0] / (insts * waves_per_simd), ms * 1e6 / (insts * waves_per_simd));
  }
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) run<0>("v_fma_f64", w);
  for (int w : {1, 2, 4}) run<1>("v_mad_i64_i32", w);
  for (int w : {1, 2, 4}) run<2>("v_add_u32", w);
  return 0;
}
