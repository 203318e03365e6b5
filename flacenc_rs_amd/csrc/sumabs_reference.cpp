// sumabs_reference.cpp -- the fixed-LPC order selector's sums of |e| in the summation orders of the reference.
//
// estimate_entropy (src/coding.rs:200-227) feeds every estimator partition's errors to
// find_sum_abs_f32::<16> (src/arrayutils.rs:496-506), an f32 reduction whose order depends on the build:
//
//   stable (slice_as_simd = (data, [], []), arrayutils.rs:435-438; simd_map_and_reduce :459-493):
//     ONE sequential chain  acc = |e[t]| as f32 + acc  over the partition, then acc + 0.0;
//   simd-nightly (`as_simd`: 64-byte aligned body of 16-lane vectors, scalar head and foot):
//     head and foot elements on the scalar chain, body element i on lane chain (i mod 16), result
//     acc + (ordered sum of the 16 lane accumulators).
//
// The kernels' own definition (exact integer sum, rounded once) coincides with both while a partition's
// sum stays below 2^24 and differs by f32 roundings above -- 24-bit material at the default 16 partitions.
// A chain of dependent f32 adds cannot be split over lanes without changing the roundings, so here a LANE
// owns one (subframe, partition) pair and walks it serially, carrying the five orders' chains side by side:
// the order-k error signal is the k-th wrapping difference of the zero-extended signal
// (reset_fixed_lpc_errors, coding.rs:182-197; the first k entries are partial differences, not zeros, and
// ARE summed), so one pass over the samples with four values of state produces all five.
// With the default 16 partitions a wave covers 4 subframes = the roles L, R, M, S of one stereo frame
// (M = (l + r) >> 1 and S = l - r formed on the fly, coding.rs:483).
//
// |e| as f32: v_cvt_f32_i32 rounds to nearest even symmetrically, so |cvt(e)| == cvt(|e|) for every e but
// i32::MIN, which 25-bit inputs cannot produce at order <= 4 (|e_4| <= 16 * 2^24).
#include "sumabs_reference.h"

#include "sumabs_chain.h"

namespace flacenc_hip {
namespace {

struct Chains {
  float acc[5];
};

template <bool STEREO, bool NIGHTLY>
__global__ void __launch_bounds__(256) sumabs_reference_kernel(SumAbsRefArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const int n = (int)a.block_size;
  const int parts = (int)a.partitions;
  const int psz = (n + parts - 1) / parts;  // block_size.div_ceil(partitions), coding.rs:209
  const int rpw = 64 / parts;               // subframes per wave
  const int used = rpw * parts;
  // lanes beyond the last whole subframe of the wave, and waves' lanes beyond the batch, shadow a valid
  // lane (same loads, nothing stored) so that the wave-uniform "every lane is inside its partition" test
  // below is not spoiled by idle lanes
  const int el = lane < used ? lane : lane % used;
  const int r = el / parts, p = el - r * parts;
  uint32_t sf = wave * (uint32_t)rpw + (uint32_t)r;
  const bool store = lane < used && sf < a.n_subframes;
  if (wave * (uint32_t)rpw >= a.n_subframes) return;  // whole wave idle (uniform)
  if (sf >= a.n_subframes) sf = a.n_subframes - 1u;

  int kind = 0;
  const int32_t* rowA;
  const int32_t* rowB;
  if (STEREO) {
    const uint32_t frame = sf >> 2;
    kind = (int)(sf & 3u);
    rowA = a.samples + (size_t)(2u * frame + (kind == 1 ? 1u : 0u)) * a.stride;
    rowB = kind >= 2 ? a.samples + (size_t)(2u * frame + 1u) * a.stride : rowA;
  } else {
    rowA = a.samples + (size_t)sf * a.stride;
    rowB = rowA;
  }
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0) && ((a.stride & 3) == 0);
  auto ld = [&](const int32_t* row, int t) -> int4 {  // samples [t, t + 4), zero outside [0, n); t % 4 == 0
    if (vec_ok && t >= 0 && t + 4 <= n) return *reinterpret_cast<const int4*>(row + t);
    int4 v;
    v.x = (t + 0 >= 0 && t + 0 < n) ? row[t + 0] : 0;
    v.y = (t + 1 >= 0 && t + 1 < n) ? row[t + 1] : 0;
    v.z = (t + 2 >= 0 && t + 2 < n) ? row[t + 2] : 0;
    v.w = (t + 3 >= 0 && t + 3 < n) ? row[t + 3] : 0;
    return v;
  };
  auto role_quad = [&](int t) -> int4 {
    int4 v = ld(rowA, t);
    if (STEREO) {
      const int4 q = ld(rowB, t);
      const int4 m = make_int4((v.x + q.x) >> 1, (v.y + q.y) >> 1, (v.z + q.z) >> 1, (v.w + q.w) >> 1);
      const int4 s = make_int4(v.x - q.x, v.y - q.y, v.z - q.z, v.w - q.w);
      v = kind == 2 ? m : (kind == 3 ? s : v);
    }
    return v;
  };

  const long long b0 = (long long)p * psz;
  const int begin = b0 < n ? (int)b0 : n;
  const int end = begin + psz < n ? begin + psz : n;
  const int b4 = begin & ~3;

  // state of the differencing at t = b4 - 1 (zero-extended signal in front of the block); the chains themselves
  // are SumAbsChains (sumabs_chain.h)
  SumAbsChains<NIGHTLY> ch;
  ch.init(begin, end);
  {
    const int4 h = role_quad(b4 - 4);
    ch.seed((uint32_t)h.x, (uint32_t)h.y, (uint32_t)h.z, (uint32_t)h.w);
  }
  auto step = [&](auto masked_tag, uint32_t x, int t, int j16) { ch.template step<decltype(masked_tag)::value>(x, t, j16); };

  // the walk: 16 samples per iteration from b4 (a multiple of 4; for NIGHTLY the iteration grid is aligned
  // to 16 so that the vector lane index is a compile-time constant)
  const int w0 = NIGHTLY ? (begin & ~15) : b4;
  if (NIGHTLY && w0 != b4) {
    // state at w0 - 1 instead of b4 - 1
    const int4 h = role_quad(w0 - 4);
    ch.seed((uint32_t)h.x, (uint32_t)h.y, (uint32_t)h.z, (uint32_t)h.w);
  }
  int my_iters = (end - w0 + 15) >> 4;
  if (my_iters < 0) my_iters = 0;
  int iters = my_iters;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int o = __shfl_xor(iters, m, 64);
    iters = o > iters ? o : iters;
  }
  // Default geometry -- 16 partitions (a wave = the four roles of one stereo frame, or four plain subframes),
  // 16-byte aligned rows, partitions that are whole multiples of 16 samples: the rows are read ONCE, coalesced
  // (four lanes fetch the 64 contiguous bytes of a partition's next 16 samples; 16 cache lines per instruction
  // where a lane-per-partition walk touches 64, per role), parked in a double-buffered per-wave LDS tile
  // [row][partition][16 + 4 pad] and picked up from there by the lanes of every role.
  __shared__ __attribute__((aligned(16))) int32_t tiles[4][2][4][16 * 20];
  const bool fast = vec_ok && parts == 16 && (psz & 15) == 0 && n == 16 * psz &&
                    (wave + 1u) * 4u <= a.n_subframes;  // wave-uniform
  if (fast) {
    constexpr int NROW = STEREO ? 2 : 4;
    int32_t(*const tile)[4][16 * 20] = tiles[threadIdx.x >> 6];
    const int lp = lane >> 2, lq = lane & 3;
    const int32_t* lrow[NROW];
#pragma unroll
    for (int rr = 0; rr < NROW; ++rr)
      lrow[rr] = STEREO ? a.samples + (size_t)(2u * (sf >> 2) + (uint32_t)rr) * a.stride
                        : a.samples + (size_t)(wave * 4u + (uint32_t)rr) * a.stride;
    int4 pre[NROW];
    auto issue = [&](int j) {
#pragma unroll
      for (int rr = 0; rr < NROW; ++rr) pre[rr] = *reinterpret_cast<const int4*>(lrow[rr] + lp * psz + 16 * j + 4 * lq);
    };
    auto land = [&](int b) {
#pragma unroll
      for (int rr = 0; rr < NROW; ++rr) *reinterpret_cast<int4*>(&tile[b][rr][lp * 20 + 4 * lq]) = pre[rr];
    };
    const int ra = STEREO ? (kind == 1 ? 1 : 0) : r;
    const int ntile = psz >> 4;
    issue(0);
    land(0);
    for (int j = 0; j < ntile; ++j) {
      if (j + 1 < ntile) issue(j + 1);
      const int32_t* ta = &tile[j & 1][ra][p * 20];
      const int32_t* tb = &tile[j & 1][STEREO ? 1 : ra][p * 20];
      const int t0 = begin + 16 * j;
      // (component by component: a select between whole int4 values went through private memory)
      auto role_of = [&](int x, int y) -> uint32_t {
        if (!STEREO) return (uint32_t)x;
        const int mid = (x + y) >> 1, side = x - y;
        return (uint32_t)(kind == 2 ? mid : (kind == 3 ? side : x));
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 v = *reinterpret_cast<const int4*>(ta + 4 * q);
        int4 w = v;
        if (STEREO) w = *reinterpret_cast<const int4*>(tb + 4 * q);
        step(std::false_type{}, role_of(v.x, w.x), t0 + 4 * q + 0, 4 * q + 0);
        step(std::false_type{}, role_of(v.y, w.y), t0 + 4 * q + 1, 4 * q + 1);
        step(std::false_type{}, role_of(v.z, w.z), t0 + 4 * q + 2, 4 * q + 2);
        step(std::false_type{}, role_of(v.w, w.w), t0 + 4 * q + 3, 4 * q + 3);
      }
      // (one wave per tile set: its LDS operations complete in order, no barrier)
      if (j + 1 < ntile) land((j + 1) & 1);
    }
  } else {
  int4 cur[4], nxt[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cur[q] = role_quad(w0 + 4 * q);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int t0 = w0 + 16 * it;
#pragma unroll
    for (int q = 0; q < 4; ++q) nxt[q] = role_quad(t0 + 16 + 4 * q);
    const bool full = t0 >= begin && t0 + 16 <= end;
    auto run = [&](auto masked_tag) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        step(masked_tag, (uint32_t)cur[q].x, t0 + 4 * q + 0, 4 * q + 0);
        step(masked_tag, (uint32_t)cur[q].y, t0 + 4 * q + 1, 4 * q + 1);
        step(masked_tag, (uint32_t)cur[q].z, t0 + 4 * q + 2, 4 * q + 2);
        step(masked_tag, (uint32_t)cur[q].w, t0 + 4 * q + 3, 4 * q + 3);
      }
    };
    if (__builtin_amdgcn_ballot_w64(!full) == 0) run(std::false_type{});
    else run(std::true_type{});
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
  }

  if (store) {
    float* __restrict__ o = a.out + (size_t)sf * (5 * 64) + p;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = ch.result(k);
      o[k * 64] = v;
    }
  }
}

}  // namespace

hipError_t launch_sumabs_reference(const SumAbsRefArgs& a, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  if (a.partitions == 0 || a.partitions > 64) return hipErrorInvalidValue;
  if (a.stereo && (a.n_subframes & 3u)) return hipErrorInvalidValue;
  const uint32_t rpw = 64u / a.partitions;
  const uint32_t waves = (a.n_subframes + rpw - 1u) / rpw;
  const uint32_t blocks = (waves + 3u) / 4u;
  if (a.nightly) {
    if (a.stereo) hipLaunchKernelGGL((sumabs_reference_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((sumabs_reference_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, a);
  } else {
    if (a.stereo) hipLaunchKernelGGL((sumabs_reference_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((sumabs_reference_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace flacenc_hip
