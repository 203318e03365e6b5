// sumabs_chain.h -- one estimator partition's five sums of |e_k| in the summation orders of the reference
// (find_sum_abs_f32::<16>, src/arrayutils.rs:496-506, under estimate_entropy, src/coding.rs:200-227), as a
// serial walk over the partition's samples.  Shared by sumabs_reference_kernel (one partition per lane, from
// global memory) and by the fused kernel's rare path (partitions whose sums reach 2^24, from its LDS images).
//
//   stable (slice_as_simd = (data, [], []), arrayutils.rs:435-438; simd_map_and_reduce :459-493):
//     ONE sequential chain  acc = |e[t]| as f32 + acc  over the partition, then acc + 0.0;
//   simd-nightly (`as_simd`: 64-byte aligned body of 16-lane vectors, scalar head and foot):
//     head and foot elements on the scalar chain, body element i on lane chain (i mod 16), result
//     acc + (ordered sum of the 16 lane accumulators).
//
// The order-k error signal is the k-th wrapping difference of the zero-extended signal
// (reset_fixed_lpc_errors, coding.rs:182-197; the first k entries are partial differences, not zeros, and ARE
// summed), so one pass with four values of state produces all five orders.
// |e| as f32: v_cvt_f32_i32 rounds to nearest even symmetrically, so |cvt(e)| == cvt(|e|) for every e but
// i32::MIN, which 25-bit inputs cannot produce at order <= 4 (|e_4| <= 16 * 2^24).
#ifndef FLACENC_HIP_SUMABS_CHAIN_H_
#define FLACENC_HIP_SUMABS_CHAIN_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flacenc_hip {

template <bool NIGHTLY>
struct SumAbsChains {
  // state of the differencing at t - 1
  uint32_t sp, e1p, e2p, e3p;
  // stable: acc[k] is the one chain of order k.  nightly: vl[k][j] are the 16 vector-lane chains of the
  // aligned body and acc[k] the scalar chain of head and foot (SimdVec storage is 64-byte aligned, so the
  // body starts at the first multiple of 16 elements at or after `begin`, arrayutils.rs:459-493)
  float acc[5];
  float vl[NIGHTLY ? 5 : 1][NIGHTLY ? 16 : 1];
  int begin, end, body_lo, body_hi;

  __device__ __forceinline__ void init(int begin_, int end_) {
    begin = begin_;
    end = end_;
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = 0.0f;
    if (NIGHTLY) {
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) vl[k][j] = 0.0f;
    }
    body_lo = body_hi = 0;
    if (NIGHTLY) {
      body_lo = (begin + 15) & ~15;
      if (body_lo > end) body_lo = end;
      body_hi = body_lo + ((end - body_lo) & ~15);
    }
  }
  // the four samples in front of the walk's first one (zeros in front of the block)
  __device__ __forceinline__ void seed(uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3) {
    sp = h3;
    e1p = h3 - h2;
    e2p = h3 - 2u * h2 + h1;
    e3p = h3 - 3u * h2 + 3u * h1 - h0;
  }
  // sample x at position t; j16 == (t - a multiple of 16) for NIGHTLY (the walk is laid on a grid of 16)
  template <bool MASKED>
  __device__ __forceinline__ void step(uint32_t x, int t, int j16) {
    const uint32_t e1 = x - sp, e2 = e1 - e1p, e3 = e2 - e2p, e4 = e3 - e3p;
    sp = x;
    e1p = e1;
    e2p = e2;
    e3p = e3;
    float f[5];
    f[0] = __builtin_fabsf((float)(int32_t)x);
    f[1] = __builtin_fabsf((float)(int32_t)e1);
    f[2] = __builtin_fabsf((float)(int32_t)e2);
    f[3] = __builtin_fabsf((float)(int32_t)e3);
    f[4] = __builtin_fabsf((float)(int32_t)e4);
    const bool in = !MASKED || (t >= begin && t < end);
    if (!NIGHTLY) {
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k] = (in ? f[k] : 0.0f) + acc[k];  // acc >= +0: adding +0 changes nothing
    } else {
      const bool body = t >= body_lo && t < body_hi;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float fk = in ? f[k] : 0.0f;
        acc[k] = (body ? 0.0f : fk) + acc[k];
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j == j16) vl[k][j] = (body ? fk : 0.0f) + vl[k][j];
      }
    }
  }
  __device__ __forceinline__ float result(int k) const {
    if (!NIGHTLY) return acc[k] + 0.0f;  // scalar_reduce_fn(acc, reduce_sum(zero vector)), arrayutils.rs:492
    float lanes = 0.0f;  // SimdFloat::reduce_sum: ordered sum of the lanes (as the oracle restates it)
#pragma unroll
    for (int j = 0; j < 16; ++j) lanes += vl[k][j];
    return acc[k] + lanes;
  }
};

}  // namespace flacenc_hip
#endif
