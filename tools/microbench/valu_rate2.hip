// valu_rate2.hip -- SIMD cycles per wave64 instruction on gfx950 for the opcodes the QLPC kernels are
// built from (8 independent chains per wave, 1 / 2 / 4 resident waves per SIMD).  Inline asm so that the
// compiler cannot substitute or fold anything.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(ASM)                                                                                 \
  for (int it = 0; it < iters; ++it) {                                                              \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                 \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM; }                                        \
    }                                                                                               \
  }

template <int KIND>
__global__ void __launch_bounds__(64) chains(int* out, int iters, int seed) {
  int m[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = seed * (i + 3) + threadIdx.x;
    d[i] = (double)m[i] * 1.25;
  }
  int x = seed | 0x10003;
  int y = seed * 7 + 1;
  double xd = 1.0000001;
  if (KIND == 0) CHAIN8(asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 1) CHAIN8(asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 2) CHAIN8(asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 3) CHAIN8(asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 4) CHAIN8(asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(m[i]) : "v"(x)))
  if (KIND == 5) CHAIN8(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 6) CHAIN8(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 7) CHAIN8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 8) CHAIN8(asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 9) CHAIN8(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(m[i])))
  if (KIND == 10) CHAIN8(asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(m[i])))
  if (KIND == 11) CHAIN8(asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(m[i])))
  if (KIND == 12) CHAIN8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(m[i])))
  if (KIND == 13) CHAIN8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(xd)))
  if (KIND == 14) CHAIN8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 15) CHAIN8(asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])))
  if (KIND == 16) CHAIN8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(xd)))
  if (KIND == 17) CHAIN8(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y) : "vcc"))
  if (KIND == 18) CHAIN8(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(m[i])))
  if (KIND == 19) CHAIN8(asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 20) CHAIN8(asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(m[i])))
  if (KIND == 21) CHAIN8(asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 22) CHAIN8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 23) CHAIN8(asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(m[i]) : "v"(x)))
  if (KIND == 24) CHAIN8(asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(xd)))
  if (KIND == 25) CHAIN8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 26) CHAIN8(asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 27) CHAIN8(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(m[i])))
  if (KIND == 28) CHAIN8(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 29) CHAIN8(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 30) CHAIN8(asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 31) CHAIN8(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += m[i] + (int)d[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

static const char* kNames[] = {"v_dot2_i32_i16", "v_dot2c_i32_i16", "v_mad_i32_i16", "v_mad_i32_i24", "v_mul_lo_u32",
                               "v_lshl_add_u32", "v_add3_u32", "v_add_u32", "v_min3_u32", "v_add_u32_dpp",
                               "v_mov_b32_dpp", "v_cvt_f32_i32", "v_cvt_f64_f32", "v_pk_mul_f32", "v_mul_f32",
                               "v_mov_b64", "v_add_f64", "v_mad_i64_i32", "v_ashrrev_i32", "v_sad_u32", "v_bfe_u32",
                               "v_lshl_or_b32", "v_perm_b32", "v_alignbit_b32", "v_fma_f64", "v_max_i32",
                               "v_max3_i32", "v_cvt_f64_i32", "v_mul_i32_i24", "v_pk_add_u16", "v_pk_mad_i16",
                               "v_sub_u32"};

template <int KIND>
void run(int* out) {
  const int iters = 2000;
  std::printf("%-16s", kNames[KIND]);
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("  %dw: %5.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64));
  }
  std::printf("   cycles (at 2.4 GHz) per wave-instruction per SIMD\n");
}

template <int K>
void run_all(int* out) {
  run<K>(out);
  if constexpr (K + 1 < 32) run_all<K + 1>(out);
}

int main() {
  int* out;
  if (hipMalloc(&out, sizeof(int) * 64 * 256 * 4 * 8) != hipSuccess) return 1;
  run_all<0>(out);
  return 0;
}
