// mfma_f64_coissue.hip -- does a dependent chain of v_mfma_f64_4x4x4_4b_f64 (the stable-order autocorrelation of
// acorr_reference.cpp) run BESIDE vector-pipe work of other waves of the same SIMD, or does it hold the pipe?
//
// Workgroups of 8 waves: waves 0..3 ("M waves", one per SIMD) run a chain of MFMAs, waves 4..7 ("V waves", their SIMD
// siblings) run 8 independent chains of one VALU opcode.  Three launches per opcode -- M alone (V waves return at
// once), V alone, both -- timed by events around the whole launch:
//     hidden = (t_M + t_V - t_both) / min(t_M, t_V)        1 = perfect overlap, 0 = the two serialise.
// Part 2 puts both in ONE instruction stream (an MFMA followed by k independent VALU instructions), which is what a
// kernel that interleaves the chains with its own integer phases would issue.
#include <hip/hip_runtime.h>
#include <cstdio>

#define V8(ASM) _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM; }

template <int KIND>
__device__ __forceinline__ void valu_block(int (&m)[8], double (&d)[8], int x, int y, double xd) {
  if (KIND == 0) V8(asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 1) V8(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 2) V8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
  if (KIND == 3) V8(asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(m[i])))
  if (KIND == 4) V8(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y) : "vcc"))
  if (KIND == 5) V8(asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(xd)))
  if (KIND == 6) V8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(m[i])))
  if (KIND == 7) V8(asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(m[i]) : "v"(x), "v"(y)))
  if (KIND == 8) V8(asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(m[i])))
  if (KIND == 9) V8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(m[i]) : "v"(x)))
}

template <int KIND>
__global__ void __launch_bounds__(512) split(int* out, int iters_m, int iters_v, int seed) {
  const int wave = threadIdx.x >> 6;
  int m[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = seed * (i + 3) + threadIdx.x;
    d[i] = (double)m[i] * 1.25;
  }
  int x = seed | 0x10003, y = seed * 7 + 1;
  double xd = 1.0000001;
  if (wave < 4) {
    double acc = threadIdx.x, a = 1.0 + threadIdx.x, b = 0.5;
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
    d[0] = acc;
  } else {
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) valu_block<KIND>(m, d, x, y, xd);
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += m[i] + (int)d[i];
  if (s == 0x7eadbeef) out[0] = s;
}

// one stream: an MFMA, then K independent VALU instructions of KIND (K a multiple of 8)
template <int KIND, int K>
__global__ void __launch_bounds__(256) mixed(int* out, int iters, int seed) {
  int m[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = seed * (i + 3) + threadIdx.x;
    d[i] = (double)m[i] * 1.25;
  }
  int x = seed | 0x10003, y = seed * 7 + 1;
  double xd = 1.0000001;
  double acc = threadIdx.x, a = 1.0 + threadIdx.x, b = 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < K / 8; ++r) valu_block<KIND>(m, d, x, y, xd);
    }
  }
  d[0] += acc;
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += m[i] + (int)d[i];
  if (s == 0x7eadbeef) out[0] = s;
}

static float time_launch(void (*k)(int*, int, int, int), int grid, int block, int* out, int a, int b) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0, best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, a, b, 12345);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}
static float time_launch3(void (*k)(int*, int, int), int grid, int block, int* out, int a) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0, best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, a, 12345);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

template <int KIND>
static void run_split(const char* name, int* out, int wg_per_cu) {
  // M: 256 x 16 = 4096 MFMAs per wave (65.5 k cycles of matrix pipe); V: iters_v x 32 instructions per wave, sized to about
  // the same pipe time at ~4.4 cycles per instruction
  const int grid = 256 * wg_per_cu;
  const int im = 256, iv = 470;
  const float tm = time_launch(split<KIND>, grid, 512, out, im, 0);
  const float tv = time_launch(split<KIND>, grid, 512, out, 0, iv);
  const float tb = time_launch(split<KIND>, grid, 512, out, im, iv);
  const float lo = tm < tv ? tm : tv;
  std::printf("split  %-18s %d WG/CU: M alone %7.1f us, V alone %7.1f us, both %7.1f us -> hidden %.2f\n", name, wg_per_cu,
              tm * 1e3, tv * 1e3, tb * 1e3, (tm + tv - tb) / lo);
}

template <int KIND>
static void run_mixed(const char* name, int* out) {
  const int grid = 256 * 3, it = 256;  // three workgroups of four waves per CU: the headline kernel's occupancy
  const float t0 = time_launch3(mixed<KIND, 0>, grid, 256, out, it);
  const float t8 = time_launch3(mixed<KIND, 8>, grid, 256, out, it);
  const float t16 = time_launch3(mixed<KIND, 16>, grid, 256, out, it);
  const float t32 = time_launch3(mixed<KIND, 32>, grid, 256, out, it);
  const double steps = 3.0 * it * 8;  // MFMAs per SIMD
  std::printf("mixed  %-18s ns per MFMA step and SIMD with 0 / 8 / 16 / 32 VALU behind each MFMA: %6.2f %6.2f %6.2f %6.2f"
              "  (per extra VALU: %5.2f %5.2f %5.2f ns)\n",
              name, t0 * 1e6 / steps, t8 * 1e6 / steps, t16 * 1e6 / steps, t32 * 1e6 / steps, (t8 - t0) * 1e6 / steps / 8,
              (t16 - t0) * 1e6 / steps / 16, (t32 - t0) * 1e6 / steps / 32);
}

int main() {
  int* out;
  hipMalloc(&out, 64);
  // spin the clock up
  for (int i = 0; i < 20; ++i) time_launch(split<2>, 1024, 512, out, 64, 64);
  for (int wg = 1; wg <= 2; ++wg) {
    run_split<0>("v_dot2_i32_i16", out, wg);
    run_split<1>("v_bitop3_b32", out, wg);
    run_split<2>("v_add_u32", out, wg);
    run_split<3>("v_mov_b32_dpp", out, wg);
    run_split<4>("v_mad_i64_i32", out, wg);
    run_split<7>("v_min3_u32", out, wg);
    run_split<8>("v_cvt_f32_i32", out, wg);
    run_split<9>("v_mul_f32", out, wg);
    run_split<5>("v_fma_f64", out, wg);
    run_split<6>("v_cvt_f64_f32", out, wg);
  }
  run_mixed<0>("v_dot2_i32_i16", out);
  run_mixed<1>("v_bitop3_b32", out);
  run_mixed<2>("v_add_u32", out);
  run_mixed<3>("v_mov_b32_dpp", out);
  run_mixed<4>("v_mad_i64_i32", out);
  run_mixed<5>("v_fma_f64", out);
  run_mixed<6>("v_cvt_f64_f32", out);
  return 0;
}
