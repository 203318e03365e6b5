// acorr_reference.cpp -- autocorrelation in the summation order of the reference's stable build.
//
// weighted_auto_correlation_nosimd (src/lpc.rs:533-548) keeps ONE accumulator per lag and walks the
// block once:  for t in P..n { for tau in 0..=P { R[tau] = fma(x_w[t - tau], x_w[t], R[tau]) } }.
// A chain of n dependent fma per lag cannot be split over lanes without changing the roundings, so
// here a LANE owns a subframe and walks it serially with P + 1 independent chains (the layout of
// levinson_batch_kernel); a wave covers 64 subframes = 16 stereo frames x {L, R, M, S}.  Global loads
// stay coalesced by going through LDS tiles: the wave reads 64 consecutive samples of each of its rows
// (16 lanes x 16 bytes per row), windows them in f32 exactly as fill_windowed_signal does
// (src/lpc.rs:739-756; M = (l + r) >> 1 and S = l - r formed here, src/coding.rs:483), and stores
// them transposed-friendly (68-float rows) so that lane r then streams row r with ds_read_b128.
// The next tile's loads are in flight while the current one is summed.
//
// Steps with t < P are masked to x_w[t] = 0 instead of skipped: fma(y, 0, acc) == acc for every
// accumulator value that can occur (accumulators start at +0 and can never become -0), so the chain
// is bit-identical to the reference's, which starts at t = P.  The same holds for the zero padding of
// the last tile of a block that is not a multiple of 64.
#include "acorr_reference.h"

#include <type_traits>

#include "lds_opt_in.h"

namespace flacenc_hip {
namespace {

constexpr int kTile = 64;  // samples per tile
constexpr int kRow = 68;   // floats per LDS row (64 + 4: lane r reading its row with b128 hits its own 4 banks)

template <int MAXP, bool STEREO>
__global__ void __launch_bounds__(64) acorr_reference_kernel(AcorrRefArgs a) {
  constexpr int HP = MAXP;  // lagged values kept in registers in front of the current 8
  constexpr int NLOAD = STEREO ? 4 : 16;  // row groups of 4 rows (16 lanes x 16 B each) per tile
  __shared__ __attribute__((aligned(16))) float tile[2][64 * kRow];
  const int lane = threadIdx.x;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf0 = blockIdx.x * 64u;
  const int n_tiles = (n + kTile - 1) / kTile;
  const int sub = lane >> 4;           // which of the 4 rows of a load group
  const int col = (lane & 15) << 2;    // first of this lane's 4 samples inside the tile
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0) && ((a.stride & 3) == 0);
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;

  int4 rawA[NLOAD], rawB[STEREO ? NLOAD : 1];
  float4 wv;
  // global -> registers for tile k: rows of this wave's subframes, samples [64 k, 64 k + 64)
  auto issue = [&](int k) {
    const int t = k * kTile + col;
    const bool full = vec_ok && (k + 1) * kTile <= n;
    auto ld = [&](const int32_t* row) -> int4 {
      if (full) return *reinterpret_cast<const int4*>(row + t);
      int4 v;
      v.x = t + 0 < n ? row[t + 0] : 0;
      v.y = t + 1 < n ? row[t + 1] : 0;
      v.z = t + 2 < n ? row[t + 2] : 0;
      v.w = t + 3 < n ? row[t + 3] : 0;
      return v;
    };
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const uint32_t rr = (uint32_t)(j * 4 + sub);
      if (STEREO) {
        const uint32_t frame = (sf0 >> 2) + rr;
        if (frame * 4u < a.n_subframes) {
          rawA[j] = ld(a.samples + (size_t)(2u * frame) * a.stride);
          rawB[j] = ld(a.samples + (size_t)(2u * frame + 1u) * a.stride);
        } else {
          rawA[j] = make_int4(0, 0, 0, 0);
          rawB[j] = make_int4(0, 0, 0, 0);
        }
      } else {
        const uint32_t sf = sf0 + rr;
        rawA[j] = sf < a.n_subframes ? ld(a.samples + (size_t)sf * a.stride) : make_int4(0, 0, 0, 0);
      }
    }
    wv = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab) {
      wv.x = t + 0 < n ? wtab[t + 0] : 0.0f;
      wv.y = t + 1 < n ? wtab[t + 1] : 0.0f;
      wv.z = t + 2 < n ? wtab[t + 2] : 0.0f;
      wv.w = t + 3 < n ? wtab[t + 3] : 0.0f;
    }
  };
  // registers -> LDS: x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754)
  auto land = [&](float* buf) {
    auto put = [&](int row, const int4& s) {
      float4 x;
      x.x = (float)s.x * wv.x;
      x.y = (float)s.y * wv.y;
      x.z = (float)s.z * wv.z;
      x.w = (float)s.w * wv.w;
      *reinterpret_cast<float4*>(&buf[row * kRow + col]) = x;
    };
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const int rr = j * 4 + sub;
      if (STEREO) {
        const int4 l = rawA[j], r = rawB[j];
        put(4 * rr + 0, l);
        put(4 * rr + 1, r);
        put(4 * rr + 2, make_int4((l.x + r.x) >> 1, (l.y + r.y) >> 1, (l.z + r.z) >> 1, (l.w + r.w) >> 1));
        put(4 * rr + 3, make_int4(l.x - r.x, l.y - r.y, l.z - r.z, l.w - r.w));
      } else {
        put(rr, rawA[j]);
      }
    }
  };

  double dw[HP + 8];
  double acc[MAXP + 1];
#pragma unroll
  for (int i = 0; i < HP + 8; ++i) dw[i] = 0.0;
#pragma unroll
  for (int i = 0; i <= MAXP; ++i) acc[i] = 0.0;

  auto sum_tile = [&](auto masked_tag, const float* buf, int t_tile) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const float* row = buf + lane * kRow;
#pragma unroll 1
    for (int step = 0; step < 8; ++step) {
#pragma unroll
      for (int k = 0; k < HP; ++k) dw[k] = dw[k + 8];
      const float4 x0 = *reinterpret_cast<const float4*>(row + 8 * step);
      const float4 x1 = *reinterpret_cast<const float4*>(row + 8 * step + 4);
      dw[HP + 0] = (double)x0.x;
      dw[HP + 1] = (double)x0.y;
      dw[HP + 2] = (double)x0.z;
      dw[HP + 3] = (double)x0.w;
      dw[HP + 4] = (double)x1.x;
      dw[HP + 5] = (double)x1.y;
      dw[HP + 6] = (double)x1.z;
      dw[HP + 7] = (double)x1.w;
      if (MASKED) {
        // t < P: no step of the reference's loop; as a zero it stays in the window for later lags,
        // where the reference reads the real x_w[t - tau] -- so only the multiplier is masked
        const int tb = t_tile + 8 * step;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double cur = (tb + k >= P) ? dw[HP + k] : 0.0;
#pragma unroll
          for (int tau = 0; tau <= MAXP; ++tau) acc[tau] = __builtin_fma(dw[HP + k - tau], cur, acc[tau]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double cur = dw[HP + k];
#pragma unroll
          for (int tau = 0; tau <= MAXP; ++tau) acc[tau] = __builtin_fma(dw[HP + k - tau], cur, acc[tau]);
        }
      }
    }
  };

  issue(0);
  land(tile[0]);
  for (int k = 0; k < n_tiles; ++k) {
    if (k + 1 < n_tiles) issue(k + 1);
    // (one wave per workgroup: LDS accesses of a wave are ordered, no barrier needed)
    if (k * kTile < P) sum_tile(std::true_type{}, tile[k & 1], k * kTile);
    else sum_tile(std::false_type{}, tile[k & 1], k * kTile);
    if (k + 1 < n_tiles) land(tile[(k + 1) & 1]);
  }

  const uint32_t sf = sf0 + (uint32_t)lane;
  if (sf < a.n_subframes) {
    double* __restrict__ o = a.out + (size_t)sf * 33;
#pragma unroll
    for (int tau = 0; tau <= MAXP; ++tau) o[tau] = tau <= P ? acc[tau] : 0.0;
    for (int tau = MAXP + 1; tau < 33; ++tau) o[tau] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// The simd-nightly build's order (weighted_auto_correlation_simd, src/lpc.rs:510-531, over
// weighted_delay_prod_sum_impl :439-500).  Per lag d the windowed signal from t = P on is split by
// `as_simd::<LANES>()` with LANES = 8 for d < 8 and 16 for d = 8..15 into a scalar head (up to the next
// LANES-aligned element; the buffer is a SimdVec<f32, 16>, 64-byte aligned, lpc.rs:710), a body of whole
// vectors and a scalar foot:
//   acc = 0 + chain(head);  acc_v[l] = fma(x[t + l], x[t + l - d], acc_v[l]) over the body's vectors;
//   acc += chain(foot);  R[d] = acc + reduce_sum(acc_v)   (ordered: ((0 + v0) + v1) + ...).
// For d >= 16 LANES is 32 or 64 and the split depends on where the allocator put the buffer, so the mode
// stops at order 15.  A vector lane's chain takes every LANES-th sample, so here a GPU LANE is a vector lane:
// threads 0..63 are the 8 x 8 lane chains of lags 0..7, threads 64..191 the 8 x 16 of lags 8..15; one
// workgroup per subframe, the windowed block staged once in LDS as f32.
template <bool STEREO>
__global__ void __launch_bounds__(256) acorr_nightly_kernel(AcorrRefArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xw[];
  const int tid = threadIdx.x;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
  {
    // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754); M = (l + r) >> 1, S = l - r (coding.rs:483)
    const int32_t* rowA;
    const int32_t* rowB = nullptr;
    int kind = 0;
    if (STEREO) {
      const uint32_t frame = sf >> 2;
      kind = (int)(sf & 3u);
      rowA = a.samples + (size_t)(2u * frame + (kind == 1 ? 1u : 0u)) * a.stride;
      rowB = a.samples + (size_t)(2u * frame + 1u) * a.stride;
    } else {
      rowA = a.samples + (size_t)sf * a.stride;
    }
    for (int t = tid; t < n; t += 256) {
      int32_t s = rowA[t];
      if (STEREO && kind >= 2) {
        const int32_t r = rowB[t];
        s = kind == 2 ? (s + r) >> 1 : s - r;
      }
      xw[t] = (float)s * (wtab ? wtab[t] : 1.0f);
    }
  }
  __syncthreads();
  int d, l, L;
  if (tid < 64) {
    d = tid >> 3;
    l = tid & 7;
    L = 8;
  } else {
    d = 8 + ((tid - 64) >> 4);
    l = (tid - 64) & 15;
    L = 16;
  }
  // the split of signal[P..]: element i of the buffer is aligned iff i % L == 0
  const int len = n - P;  // >= 1: blocks are at least 64 samples, P <= 15
  const int mis = P % L;
  int head = mis ? L - mis : 0;
  if (head > len) head = len;
  const int nbody = (len - head) / L;
  const int t0 = P + head;
  const bool lag_used = d <= P && d <= 15;
  double acc_v = 0.0;
  if (lag_used) {
    const float* __restrict__ cur = xw + t0 + l;
    int b = 0;
    for (; b + 4 <= nbody; b += 4) {
      const float c0 = cur[0], c1 = cur[L], c2 = cur[2 * L], c3 = cur[3 * L];
      const float g0 = cur[0 - d], g1 = cur[L - d], g2 = cur[2 * L - d], g3 = cur[3 * L - d];
      acc_v = __builtin_fma((double)c0, (double)g0, acc_v);
      acc_v = __builtin_fma((double)c1, (double)g1, acc_v);
      acc_v = __builtin_fma((double)c2, (double)g2, acc_v);
      acc_v = __builtin_fma((double)c3, (double)g3, acc_v);
      cur += 4 * L;
    }
    for (; b < nbody; ++b) {
      acc_v = __builtin_fma((double)cur[0], (double)cur[0 - d], acc_v);
      cur += L;
    }
  }
  // ordered lane sum, gathered by the group's first lane (groups are aligned runs of L lanes of one wave)
  double lanesum = 0.0;
  const int base = (tid & 63) & ~(L - 1);
  for (int j = 0; j < L; ++j) {
    const double v = __shfl(acc_v, base + j, 64);
    lanesum += v;
  }
  if (l == 0 && d <= 32) {
    double r = 0.0;
    if (lag_used) {
      double acc = 0.0, h = 0.0, f = 0.0;
      for (int t = P; t < t0; ++t) h = __builtin_fma((double)xw[t - d], (double)xw[t], h);
      acc += h;
      for (int t = t0 + nbody * L; t < n; ++t) f = __builtin_fma((double)xw[t - d], (double)xw[t], f);
      acc += f;
      r = acc + lanesum;
    }
    a.out[(size_t)sf * 33 + d] = r;
  }
  // lags 16..32 are not produced in this mode: written as 0 like every lag above P
  if (tid >= 192 && tid < 192 + 17) a.out[(size_t)sf * 33 + 16 + (tid - 192)] = 0.0;
}

hipError_t launch_nightly(const AcorrRefArgs& a, hipStream_t stream) {
  if (a.lpc_order > 15) return hipErrorNotSupported;
  const size_t smem = (((size_t)a.block_size + 3) & ~(size_t)3) * sizeof(float);
  static DynamicLdsOptIn opt_s, opt_p;
  if (a.stereo) {
    if (hipError_t e = opt_s.ensure(reinterpret_cast<const void*>(acorr_nightly_kernel<true>), smem); e != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_nightly_kernel<true>), dim3(a.n_subframes), dim3(256), smem, stream, a);
  } else {
    if (hipError_t e = opt_p.ensure(reinterpret_cast<const void*>(acorr_nightly_kernel<false>), smem); e != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_nightly_kernel<false>), dim3(a.n_subframes), dim3(256), smem, stream, a);
  }
  return hipGetLastError();
}

template <int MAXP>
hipError_t launch_bucket(const AcorrRefArgs& a, hipStream_t stream) {
  const uint32_t blocks = (a.n_subframes + 63u) / 64u;
  if (a.stereo) hipLaunchKernelGGL((acorr_reference_kernel<MAXP, true>), dim3(blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((acorr_reference_kernel<MAXP, false>), dim3(blocks), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_acorr_reference(const AcorrRefArgs& a, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  if (a.stereo && (a.n_subframes & 3u)) return hipErrorInvalidValue;
  if (a.nightly) return launch_nightly(a, stream);
  const uint32_t P = a.lpc_order;
  if (P <= 8) return launch_bucket<8>(a, stream);
  if (P <= 12) return launch_bucket<12>(a, stream);
  if (P <= 16) return launch_bucket<16>(a, stream);
  if (P <= 24) return launch_bucket<24>(a, stream);
  if (P <= 32) return launch_bucket<32>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace flacenc_hip
