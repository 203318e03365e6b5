// qlpc_bigblock.cpp -- blocks of 4096 / 8192 / 16384 samples with LPC order 13..32 (BASELINE configs[2] and
// configs[4]: 96 kHz / 192 kHz 24-bit material; 4096-sample blocks at the orders the fused 4096 kernel does
// not carry), as two kernels either side of levinson_batch_kernel:
//
//   bigblock_acorr_kernel      window + f64 autocorrelation -> R[]            (src/lpc.rs:739-756, 533-548)
//   levinson_batch_kernel      one subframe per lane (qlpc_kernel_impl.h)     (src/lpc.rs:633-705, 234-302)
//   bigblock_residual_kernel   residual + exhaustive partitioned-Rice search  (src/lpc.rs:306-390,
//                                                                              src/rice.rs:65-298)
//
// Both walk a block in PASSES of 4096 samples with the layout of the fused 4096 kernel
// (qlpc_wave_kernel_impl.h): a workgroup is 4 waves = the roles L, R, M, S of one stereo frame sharing
// two channel images in LDS (or four plain subframes), lane l of a wave owns the 64 samples
// [4096 k + 64 l, +64) of pass k = four 16-sample chunks = one finest Rice partition.  Only the pass's
// 4096 samples (+ 64 of halo) are staged, so LDS stays at 35 KB per workgroup whatever the block size.
// The autocorrelation sums are the canonical ones: 16-sample chunk chains, then a
// balanced tree over the chunk index c = 256 k + 4 l + i -- chunk bits in the lane, lane bits by a wave
// butterfly, pass bits last.  One launch for all lags would need three 25- or 33-entry f64 accumulator
// sets per lane next to the window; the lags are therefore worked off in groups of at most 13 (two groups
// at order <= 24, three above), each group re-reading the lane's samples from LDS.
// The Rice search keeps the seven bit-planes of every pass in registers, runs levels 0..6 per pass with
// the 4096 kernel's level code and adds the levels that merge whole passes (orders 7 - level, 8 - level).
#include <type_traits>

#include "qlpc_kernel.h"
#include "qlpc_wave_kernel_impl.h"

namespace flacenc_hip {
namespace {

constexpr int kPass = 4096;
#ifndef FLACENC_BIG_ACORR_OCC
#define FLACENC_BIG_ACORR_OCC 2
#endif
#ifndef FLACENC_BIG_NG24
#define FLACENC_BIG_NG24 1
#endif
#ifndef FLACENC_BIG_NG32
#define FLACENC_BIG_NG32 2
#endif
#ifndef FLACENC_BIG_RESID_PREFETCH
#define FLACENC_BIG_RESID_PREFETCH 1
#endif
#ifndef FLACENC_BIG_RESID_OCC
#define FLACENC_BIG_RESID_OCC 3
#endif

// cooperative load of pass k of the workgroup's rows into the LDS images: segment 0 of an image holds the
// 64 samples in front of the pass (zeros in front of the block), the pass follows (widx layout)
template <bool STEREO>
__device__ __forceinline__ void bigblock_load_pass(const QlpcKernelArgs& a, int32_t* sm, uint32_t blk, int k, int tid,
                                                   int wave, int lane, uint32_t sf) {
  const size_t t0 = (size_t)k * kPass;
  if (STEREO) {
    const int32_t* __restrict__ src = a.samples + (size_t)(2u * blk) * a.stride + t0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = tid + it * 256;  // 0..2047
      const int ch = q >> 10;
      const int t = (q & 1023) << 2;
      const int4 v = *reinterpret_cast<const int4*>(src + (size_t)ch * a.stride + t);
      *reinterpret_cast<int4*>(&sm[ch * kBufDwords + widx(t)]) = v;
    }
    if (tid < 32) {
      const int ch = tid >> 4;
      const int t = ((tid & 15) << 2) - 64;
      int4 v = make_int4(0, 0, 0, 0);
      if (k > 0) v = *reinterpret_cast<const int4*>(src + (size_t)ch * a.stride + t);
      *reinterpret_cast<int4*>(&sm[ch * kBufDwords + widx(t)]) = v;
    }
  } else {
    const int32_t* __restrict__ src = a.samples + (size_t)sf * a.stride + t0;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int t = (lane + it * 64) << 2;
      const int4 v = *reinterpret_cast<const int4*>(src + t);
      *reinterpret_cast<int4*>(&sm[wave * kBufDwords + widx(t)]) = v;
    }
    if (lane < 16) {
      const int t = (lane << 2) - 64;
      int4 v = make_int4(0, 0, 0, 0);
      if (k > 0) v = *reinterpret_cast<const int4*>(src + t);
      *reinterpret_cast<int4*>(&sm[wave * kBufDwords + widx(t)]) = v;
    }
  }
}

// four samples of a role at pass-relative t (multiple of 4, >= -64): own image, mid or side
template <int KIND>
__device__ __forceinline__ int4 bigblock_ld4(const int32_t* bufA, const int32_t* bufB, int t) {
  int4 v = *reinterpret_cast<const int4*>(&bufA[widx(t)]);
  if (KIND >= 2) {
    const int4 r = *reinterpret_cast<const int4*>(&bufB[widx(t)]);
    if (KIND == 2) {  // mid = (l + r) >> 1, coding.rs:483
      v.x = (v.x + r.x) >> 1;
      v.y = (v.y + r.y) >> 1;
      v.z = (v.z + r.z) >> 1;
      v.w = (v.w + r.w) >> 1;
    } else {  // side = l - r
      v.x -= r.x;
      v.y -= r.y;
      v.z -= r.z;
      v.w -= r.w;
    }
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// Autocorrelation in HALF passes of 2048 samples: a lane owns 32 samples = two 16-sample chunks of a half
// pass, so the in-lane part of the tree is one level (c0 + c1) and a lane carries two accumulator sets
// instead of three -- all 25 lags of an order-24 analysis (or 17 + 16 of an order-32 one) next to the
// window of HP lagged values + one chunk, in 256 registers, with every sample converted once per group.
constexpr int kHalf = 2048;
constexpr int kHSeg = 36;                    // dwords per lane segment: 32 samples + 4 pad (conflict-free b128)
constexpr int kHBufDwords = 65 * kHSeg + 4;  // one leading segment: the 32 samples in front of the half pass
__device__ __forceinline__ int hidx(int t) { return ((t >> 5) + 1) * kHSeg + (t & 31); }  // t >= -32

// A half pass travels global -> registers -> LDS in two steps so that the loads of half pass k + 1 are in
// flight while half pass k is being summed: fetch issues them (4 int4 per thread, plain variables -- a
// struct or array here stays in scratch memory and the loads are waited for at once), store parks them in
// the images once every wave is done with the previous contents.  Plain (non-stereo) mode: a wave loads its
// own row, 8 int4 per lane; the second four are loaded and stored in the store step.  The 32 samples in front
// of a half pass are the tail of the previous one: copied inside LDS (read before the barrier, while the old
// contents are intact).
#define FLACENC_HALF_FETCH(K_)                                                                                        \
  {                                                                                                                    \
    const size_t t0_ = (size_t)(K_) * kHalf;                                                                           \
    if (STEREO) {                                                                                                      \
      const int32_t* __restrict__ src_ = a.samples + (size_t)(2u * blk) * a.stride + t0_;                              \
      pf0 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 0) >> 9) * a.stride + (((tid + 0) & 511) << 2));     \
      pf1 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 256) >> 9) * a.stride + (((tid + 256) & 511) << 2)); \
      pf2 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 512) >> 9) * a.stride + (((tid + 512) & 511) << 2)); \
      pf3 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 768) >> 9) * a.stride + (((tid + 768) & 511) << 2)); \
    } else {                                                                                                           \
      const int32_t* __restrict__ src_ = a.samples + (size_t)sf * a.stride + t0_;                                      \
      pf0 = *reinterpret_cast<const int4*>(src_ + ((lane + 0) << 2));                                                  \
      pf1 = *reinterpret_cast<const int4*>(src_ + ((lane + 64) << 2));                                                 \
      pf2 = *reinterpret_cast<const int4*>(src_ + ((lane + 128) << 2));                                                \
      pf3 = *reinterpret_cast<const int4*>(src_ + ((lane + 192) << 2));                                                \
    }                                                                                                                  \
  }
#define FLACENC_HALF_STORE(K_)                                                                                         \
  {                                                                                                                    \
    if (STEREO) {                                                                                                      \
      *reinterpret_cast<int4*>(&sm[((tid + 0) >> 9) * kHBufDwords + hidx(((tid + 0) & 511) << 2)]) = pf0;              \
      *reinterpret_cast<int4*>(&sm[((tid + 256) >> 9) * kHBufDwords + hidx(((tid + 256) & 511) << 2)]) = pf1;          \
      *reinterpret_cast<int4*>(&sm[((tid + 512) >> 9) * kHBufDwords + hidx(((tid + 512) & 511) << 2)]) = pf2;          \
      *reinterpret_cast<int4*>(&sm[((tid + 768) >> 9) * kHBufDwords + hidx(((tid + 768) & 511) << 2)]) = pf3;          \
      if (tid < 16) *reinterpret_cast<int4*>(&sm[(tid >> 3) * kHBufDwords + hidx(((tid & 7) << 2) - 32)]) = pfh;       \
    } else {                                                                                                           \
      const int32_t* __restrict__ src_ = a.samples + (size_t)sf * a.stride + (size_t)(K_) * kHalf;                     \
      *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane + 0) << 2)]) = pf0;                                 \
      *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane + 64) << 2)]) = pf1;                                \
      *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane + 128) << 2)]) = pf2;                               \
      *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane + 192) << 2)]) = pf3;                               \
      _Pragma("unroll") for (int it_ = 4; it_ < 8; ++it_)                                                              \
        *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane + it_ * 64) << 2)]) =                             \
            *reinterpret_cast<const int4*>(src_ + ((lane + it_ * 64) << 2));                                           \
      if (lane < 8) *reinterpret_cast<int4*>(&sm[wave * kHBufDwords + hidx((lane << 2) - 32)]) = pfh;                  \
    }                                                                                                                  \
  }

template <int KIND>
__device__ __forceinline__ int4 bigblock_hld4(const int32_t* bufA, const int32_t* bufB, int t) {
  int4 v = *reinterpret_cast<const int4*>(&bufA[hidx(t)]);
  if (KIND >= 2) {
    const int4 r = *reinterpret_cast<const int4*>(&bufB[hidx(t)]);
    if (KIND == 2) {  // mid = (l + r) >> 1, coding.rs:483
      v.x = (v.x + r.x) >> 1;
      v.y = (v.y + r.y) >> 1;
      v.z = (v.z + r.z) >> 1;
      v.w = (v.w + r.w) >> 1;
    } else {  // side = l - r
      v.x -= r.x;
      v.y -= r.y;
      v.z -= r.z;
      v.w -= r.w;
    }
  }
  return v;
}

template <int HP, int NG, bool STEREO, int NLAGS = HP + 1>
__global__ void __launch_bounds__(256, FLACENC_BIG_ACORR_OCC) bigblock_acorr_kernel(QlpcKernelArgs a) {
  // HP = window depth = the order bucket (8, 16, 24 or 32); lags 0..NLAGS-1 (<= HP) in NG groups of at most NL
  constexpr int NLAG = NLAGS;
  constexpr int NL = (NLAG + NG - 1) / NG;
  constexpr int NBATCH = 13;  // lags per LDS tree round (4 lanes per lag, <= 16)
  constexpr int LVMAX = 3;    // half passes per block <= 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  constexpr int NBUF = STEREO ? 2 : 4;
  double* const part = reinterpret_cast<double*>(sm + NBUF * kHBufDwords);    // [4 waves][LVMAX][NLAG]
  double* const cross = part + 4 * LVMAX * NLAG;                               // [4 waves][NBATCH][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const uint32_t blk = blockIdx.x;
  uint32_t sf = blk * 4u + (uint32_t)wave;
  const bool active = sf < a.n_subframes;
  if (!active) sf = a.n_subframes - 1u;  // (plain mode tail: redo the last subframe, write nothing)
  const int role = STEREO ? wave : 0;
  const int P = (int)a.lpc_order;
  const int K2 = (int)(a.block_size / kHalf);
  const int32_t* const bufA = sm + (STEREO ? (role == 1 ? 1 : 0) : wave) * kHBufDwords;
  const int32_t* const bufB = sm + kHBufDwords;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
  const int tl = lane << 5;
  double* const mine = part + wave * LVMAX * NLAG;
  double* const mycross = cross + wave * NBATCH * 64;

  auto stamp = [&](int slot) {
    if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + slot] = (unsigned long long)clock64();
  };
  stamp(0);
  int4 pf0, pf1, pf2, pf3;
  FLACENC_HALF_FETCH(0)
  for (int k = 0; k < K2; ++k) {
    int4 pfh = make_int4(0, 0, 0, 0);
    if (k > 0) {
      if (STEREO) {
        if (tid < 16) pfh = *reinterpret_cast<const int4*>(&sm[(tid >> 3) * kHBufDwords + hidx(kHalf - 32 + ((tid & 7) << 2))]);
      } else {
        if (lane < 8) pfh = *reinterpret_cast<const int4*>(&sm[wave * kHBufDwords + hidx(kHalf - 32 + (lane << 2))]);
      }
    }
    __syncthreads();  // every wave is done with the previous half pass
    if (k == 0) stamp(1);
    FLACENC_HALF_STORE(k)
    if (k + 1 < K2) FLACENC_HALF_FETCH(k + 1)  // lands during the sums below
    // a half pass (and the 32 samples in front of it) inside the window's run of exact ones needs no
    // weights: (f32)s * 1.0f == (f32)s.  Otherwise they are read where they are used, from the table every
    // workgroup shares (L1 / L2 hits; the samples in front of the block are 0, whatever their weight).
    const int g0 = k * kHalf;
    const bool tapered = wtab != nullptr && !(g0 - 32 >= a.flat_lo && g0 + kHalf <= a.flat_hi);
    if (k == 0) stamp(2);
    __syncthreads();
    if (k == 0) stamp(3);
    auto run = [&](auto kind_tag, auto tapered_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      constexpr bool TAPERED = decltype(tapered_tag)::value;
      // x_w[t] = (f32)s[t] * w[t], one f32 rounding, then widened (lpc.rs:751-754)
      auto conv8 = [&](double* dst, int t) {  // t relative to the half pass, multiple of 8, >= -HP
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int4 v = bigblock_hld4<KIND>(bufA, bufB, t + 4 * q);
          if (TAPERED) {
            const int g = g0 + t + 4 * q;  // position in the block; negative only in front of the block (samples 0)
            const float4 w = *reinterpret_cast<const float4*>(wtab + (g < 0 ? 0 : g));
            dst[4 * q + 0] = (double)((float)v.x * w.x);
            dst[4 * q + 1] = (double)((float)v.y * w.y);
            dst[4 * q + 2] = (double)((float)v.z * w.z);
            dst[4 * q + 3] = (double)((float)v.w * w.w);
          } else {
            // weight exactly 1: (f64)((f32)s * 1.0f) == (f64)s, because samples of at most 25 bits (|s| <= 2^24,
            // include/flacenc_hip.h) are exact in f32 -- one conversion instead of two
            dst[4 * q + 0] = (double)v.x;
            dst[4 * q + 1] = (double)v.y;
            dst[4 * q + 2] = (double)v.z;
            dst[4 * q + 3] = (double)v.w;
          }
        }
      };
      auto group = [&](auto g_tag) {
        constexpr int G = decltype(g_tag)::value;
        constexpr int lag0 = G * NL;
        constexpr int NLG = (lag0 + NL <= NLAG) ? NL : NLAG - lag0;
        static_assert(NLG <= 2 * NBATCH, "a lag group is reduced in at most two tree rounds");
        // window of HP lagged values + one 16-sample chunk, slid by 16 between the lane's two chunks
        double dw[HP + 16];
#pragma unroll
        for (int b = 0; b < HP / 8; ++b) conv8(&dw[8 * b], tl - HP + 8 * b);
        double acc[NLG], s0[NLG];
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          conv8(&dw[HP], tl + 16 * i);
          conv8(&dw[HP + 8], tl + 16 * i + 8);
          // common lower bound t = P for every lag (lpc.rs:542): only the block's first 32 samples -- lane 0
          // of half pass 0 -- can lie below it; every other half pass runs the body without the selects
          auto body = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              double cur = dw[HP + kk];
              if (MASKED) cur = (tl + 16 * i + kk < P) ? 0.0 : cur;
#pragma unroll
              for (int j = 0; j < NLG; ++j) {
                const double lagged = dw[HP + kk - (lag0 + j)];  // lag <= HP: index >= 0
                acc[j] = kk == 0 ? __builtin_fma(cur, lagged, 0.0) : __builtin_fma(cur, lagged, acc[j]);
              }
            }
          };
          if (k == 0) body(std::true_type{});
          else body(std::false_type{});
          // in-lane level of the balanced tree over the chunk index: c0 + c1
          if (i == 0) {
#pragma unroll
            for (int j = 0; j < NLG; ++j) s0[j] = acc[j];
#pragma unroll
            for (int q = 0; q < HP; ++q) dw[q] = dw[q + 16];
          } else {
#pragma unroll
            for (int j = 0; j < NLG; ++j) s0[j] = s0[j] + acc[j];
          }
        }
        // lane levels (up to NBATCH lags at once through LDS, the total of lag j lands in lane 4 j + 3), then
        // the half-pass levels as a binary counter: a finished half pass absorbs the waiting partial of every
        // level whose bit is set in k (earlier + later, as the tree pairs them) and parks at the first clear one
        auto batch = [&](auto b_tag) {
          constexpr int B0 = decltype(b_tag)::value;
          constexpr int CNT = (NLG - B0 < NBATCH) ? NLG - B0 : NBATCH;
          double r = wave_tree_sums_lds_n<CNT>(&s0[B0], mycross, lane);
          const int lag = lag0 + B0 + (lane >> 2);
          if ((lane & 3) == 3 && (lane >> 2) < CNT) {
            int lv = 0;
            for (; lv < LVMAX && ((k >> lv) & 1); ++lv) r = mine[lv * NLAG + lag] + r;
            if (k != K2 - 1) mine[lv * NLAG + lag] = r;
            else if (active) a.autocorr[(size_t)sf * 33 + lag] = (lag <= P) ? r : 0.0;
          }
        };
        batch(std::integral_constant<int, 0>{});
        if (NLG > NBATCH) batch(std::integral_constant<int, (NLG > NBATCH ? NBATCH : 0)>{});
      };
      group(std::integral_constant<int, 0>{});
      if (NG > 1) group(std::integral_constant<int, (NG > 1 ? 1 : 0)>{});
      if (NG > 2) group(std::integral_constant<int, (NG > 2 ? 2 : 0)>{});
    };
    auto run_role = [&](auto tapered_tag) {
      if (STEREO && role == 2) run(std::integral_constant<int, 2>{}, tapered_tag);
      else if (STEREO && role == 3) run(std::integral_constant<int, 3>{}, tapered_tag);
      else run(std::integral_constant<int, 0>{}, tapered_tag);
    };
    if (tapered) run_role(std::true_type{});
    else run_role(std::false_type{});
    if (k == 0) stamp(4);
    if (k == K2 - 1) stamp(5);
  }
  if (lane == 0 && active)
    for (int j = NLAG; j < 33; ++j) a.autocorr[(size_t)sf * 33 + j] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// FIXD: the predictor is one of fixed_lpc's (order `warm` <= 4, FIXED_LPC_COEFS, shift 0): the error signal by
// repeated differencing, no multiply-adds (an instantiation of its own: as a run-time branch next to the
// multiply-add path it cost the <8, stereo, 2> kernel 122 spilled registers)
template <int MAXP, bool STEREO, int K, bool FIXD = false>
__global__ void __launch_bounds__(256, FLACENC_BIG_RESID_OCC) bigblock_residual_kernel(QlpcKernelArgs a) {
  static_assert(!FIXD || MAXP == 8, "the differencing path lives in the order-8 bucket");
  constexpr int HP = MAXP;  // multiple of 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const uint32_t blk = blockIdx.x;
  uint32_t sf = blk * 4u + (uint32_t)wave;
  const bool active = sf < a.n_subframes;
  if (!active) sf = a.n_subframes - 1u;
  const int role = STEREO ? wave : 0;
  const int n = (int)a.block_size;
  const int32_t* const bufA = sm + (STEREO ? (role == 1 ? 1 : 0) : wave) * kBufDwords;
  const int32_t* const bufB = sm + kBufDwords;
  const int tl = lane << 6;
  const unsigned long long bps_role = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + ((STEREO && role == 3) ? 1u : 0u));
  // the quantised predictor levinson_batch_kernel left: qc[32], order, shift, status
  const int32_t* __restrict__ pr = a.pred + (size_t)sf * 36;
  int32_t cq[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) cq[i] = uni(pr[i]);
  const int warm = uni(pr[32]);
  const int shift = uni(pr[33]);
  const int status = uni(pr[34]);
  int32_t* __restrict__ rrow = a.residual + (size_t)sf * a.residual_stride;
  if (STEREO && a.residual_lr != nullptr && role < 2)  // L / R candidates in place of the output channel they can fill
    rrow = a.residual_lr + (size_t)(2u * blk + (uint32_t)role) * a.residual_lr_stride;
  int vmax = INT32_MIN, vmin = INT32_MAX;  // (only kept when minmax_out is set)

  uint32_t pl[K][7];
  // Stereo blocks of two passes: the second pass's loads are issued before the first pass is worked off and
  // parked in the images afterwards (8 int4 per thread in plain variables, see FLACENC_HALF_FETCH); the 64
  // samples in front of a pass are the tail of the previous one, copied inside LDS.
  // (order bucket 24 only: at 32 the eight extra registers per int4 spill, and a spilled load is waited for at once)
  constexpr bool PREFETCH = STEREO && K == 2 && MAXP <= 24 && FLACENC_BIG_RESID_PREFETCH;
  int4 pq0, pq1, pq2, pq3, pq4, pq5, pq6, pq7;
#define FLACENC_PASS_FETCH(K_)                                                                                \
  {                                                                                                            \
    const int32_t* __restrict__ src_ = a.samples + (size_t)(2u * blk) * a.stride + (size_t)(K_) * kPass;       \
    pq0 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 0) >> 10) * a.stride + (((tid + 0) & 1023) << 2));       \
    pq1 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 256) >> 10) * a.stride + (((tid + 256) & 1023) << 2));   \
    pq2 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 512) >> 10) * a.stride + (((tid + 512) & 1023) << 2));   \
    pq3 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 768) >> 10) * a.stride + (((tid + 768) & 1023) << 2));   \
    pq4 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 1024) >> 10) * a.stride + (((tid + 1024) & 1023) << 2)); \
    pq5 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 1280) >> 10) * a.stride + (((tid + 1280) & 1023) << 2)); \
    pq6 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 1536) >> 10) * a.stride + (((tid + 1536) & 1023) << 2)); \
    pq7 = *reinterpret_cast<const int4*>(src_ + (size_t)((tid + 1792) >> 10) * a.stride + (((tid + 1792) & 1023) << 2)); \
  }
#define FLACENC_PASS_PUT(I_, V_) \
  *reinterpret_cast<int4*>(&sm[((tid + 256 * (I_)) >> 10) * kBufDwords + widx(((tid + 256 * (I_)) & 1023) << 2)]) = V_;
  if (PREFETCH) FLACENC_PASS_FETCH(0)
  for (int k = 0; k < K; ++k) {
    if (PREFETCH) {
      int4 halo = make_int4(0, 0, 0, 0);
      if (k > 0 && tid < 32) halo = *reinterpret_cast<const int4*>(&sm[(tid >> 4) * kBufDwords + widx(kPass - 64 + ((tid & 15) << 2))]);
      __syncthreads();
      FLACENC_PASS_PUT(0, pq0) FLACENC_PASS_PUT(1, pq1) FLACENC_PASS_PUT(2, pq2) FLACENC_PASS_PUT(3, pq3)
      FLACENC_PASS_PUT(4, pq4) FLACENC_PASS_PUT(5, pq5) FLACENC_PASS_PUT(6, pq6) FLACENC_PASS_PUT(7, pq7)
      if (tid < 32) *reinterpret_cast<int4*>(&sm[(tid >> 4) * kBufDwords + widx(((tid & 15) << 2) - 64)]) = halo;
      if (k + 1 < K) FLACENC_PASS_FETCH(k + 1)
    } else {
      __syncthreads();
      bigblock_load_pass<STEREO>(a, sm, blk, k, tid, wave, lane, sf);
    }
    __syncthreads();
    auto run = [&](auto kind_tag, uint32_t (&planes)[7]) {
      constexpr int KIND = decltype(kind_tag)::value;
      // e[t] = s[t] - ((sum_j c_j s[t-1-j]) >> shift), exact in 64 bits, truncated to i32
      // (lpc.rs:306-350, the i64 branch of :379-388 -- both branches give the same value)
      int sw[HP + 16];
      uint32_t pc[6];
      // the HP samples in front of the lane, then a rolled loop over its four 16-sample chunks (every
      // register array index below is a compile-time constant; the window slides at the loop's end)
#pragma unroll
      for (int q = 0; q < HP; q += 4) {
        const int4 v = bigblock_ld4<KIND>(bufA, bufB, tl - HP + q);
        sw[q + 0] = v.x;
        sw[q + 1] = v.y;
        sw[q + 2] = v.z;
        sw[q + 3] = v.w;
      }
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int t0 = tl + 16 * i;
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const int4 v = bigblock_ld4<KIND>(bufA, bufB, t0 + q);
          sw[HP + q + 0] = v.x;
          sw[HP + q + 1] = v.y;
          sw[HP + q + 2] = v.z;
          sw[HP + q + 3] = v.w;
        }
        if (a.minmax_out != nullptr) {
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            vmax = max(vmax, max(sw[HP + q], sw[HP + q + 1]));
            vmin = min(vmin, min(sw[HP + q], sw[HP + q + 1]));
          }
        }
        int32_t e[16];
        if (FIXD) {
          // fixed_lpc's predictors are repeated differences (reset_fixed_lpc_errors, coding.rs:182-197): `warm`
          // wrapping subtractions per sample instead of eight 64-bit multiply-adds, a shift and a subtraction
          // (the same value: FIXED_LPC_COEFS[k] with shift 0 is the k-th difference, decode.rs:179-201)
          // (one straight-line body per order, chosen by a wave-uniform switch: a rolled loop over the levels kept
          // the whole window live across its back edge and spilled)
          auto diff = [&](auto order_tag) {
            constexpr int ORD = decltype(order_tag)::value;
            int32_t d[16 + ORD];
#pragma unroll
            for (int q = 0; q < 16 + ORD; ++q) d[q] = sw[HP - ORD + q];
#pragma unroll
            for (int lvl = 0; lvl < ORD; ++lvl) {
#pragma unroll
              for (int q = 15 + ORD; q >= 1 + lvl; --q) d[q] = (int32_t)((uint32_t)d[q] - (uint32_t)d[q - 1]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) e[q] = ((k == 0 && t0 + q < ORD) || status != 0) ? 0 : d[ORD + q];
          };
          switch (warm) {
            case 0: diff(std::integral_constant<int, 0>{}); break;
            case 1: diff(std::integral_constant<int, 1>{}); break;
            case 2: diff(std::integral_constant<int, 2>{}); break;
            case 3: diff(std::integral_constant<int, 3>{}); break;
            default: diff(std::integral_constant<int, 4>{}); break;
          }
        } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          int64_t pred = 0;
#pragma unroll
          for (int j = 0; j < MAXP; ++j) pred += (int64_t)cq[j] * (int64_t)sw[HP + q - 1 - j];
          e[q] = (int32_t)(uint32_t)(uint64_t)((int64_t)sw[HP + q] - (pred >> shift));
          // e[0 .. order') = 0 (lpc.rs:349): the block's first samples, i.e. pass 0, lane 0
          // (as a wave-uniform branch around the first two chunks instead of selects on every sample: measured
          // 3-7 % slower -- more registers live across the branch, 30 spilled)
          if ((k == 0 && t0 + q < warm) || status != 0) e[q] = 0;
        }
        }
        if (active) {
#pragma unroll
          for (int q = 0; q < 16; q += 4)
            *reinterpret_cast<int4*>(rrow + (size_t)k * kPass + t0 + q) = make_int4(e[q], e[q + 1], e[q + 2], e[q + 3]);
        }
        // bit-sliced population counts of the chunk, accumulated into the lane's 7 planes for this pass
        uint32_t pb[5];
        popcount_planes16(e, pb);
        if (i == 0) {
#pragma unroll
          for (int q = 0; q < 5; ++q) planes[q] = pb[q];
        } else if (i == 1) {
          planes_add<5>(planes, pb);
        } else if (i == 2) {
#pragma unroll
          for (int q = 0; q < 5; ++q) pc[q] = pb[q];
        } else {
          planes_add<5>(pc, pb);
          planes_add<6>(planes, pc);
        }
#pragma unroll
        for (int q = 0; q < HP; ++q) sw[q] = sw[q + 16];
      }
    };
    // (the pass loop is rolled; pl[k] is selected by a compare chain so that the planes stay in registers)
    uint32_t now[7];
    if (STEREO && role == 2) run(std::integral_constant<int, 2>{}, now);
    else if (STEREO && role == 3) run(std::integral_constant<int, 3>{}, now);
    else run(std::integral_constant<int, 0>{}, now);
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
      if (kk == k) {
#pragma unroll
        for (int q = 0; q < 7; ++q) pl[kk][q] = now[q];
      }
  }

  // ======================= partitioned-Rice search over 64 K partitions =======================
  // finest order FO = 6 + log2 K (rice.rs:157-165); level L = order FO - L
  constexpr int LK = K == 1 ? 0 : (K == 2 ? 1 : 2);
  constexpr int NLEV = 7 + LK;
  constexpr uint32_t kWMax = kMaxPToBits - 4u;
  uint32_t orp = 0;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int q = 0; q < 7; ++q) orp |= pl[k][q];
  const uint32_t orw = wave_or_dpp(orp);
  const uint32_t maxu = (orw << 1) | (orw >> 31);
  const uint32_t bitlen = maxu ? (uint32_t)(32 - __builtin_clz(maxu)) : 0u;
  const uint32_t max_p = a.max_rice_parameter < bitlen ? a.max_rice_parameter : bitlen;
  const bool finest_only = a.rice_finest_only != 0;
  const bool small_bits = a.max_rice_parameter >= bitlen;
  PlaneSums ps[K];
#pragma unroll
  for (int k = 0; k < K; ++k) ps[k] = make_plane_sums(pl[k]);
  uint32_t len0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) len0[k] = 64u - ((k == 0 && lane == 0) ? (uint32_t)warm : 0u);
  // residuals of 2^26 and more (the reference's wrapping chunk sums, rice.rs:88-93, then differ from the
  // exact ones) are left to the generic kernel: this launch reports it and the dispatcher reruns it
  const bool literal = !(maxu < (1u << 26));

  uint32_t pk[K][7], pk7[K >= 2 ? K / 2 : 1], pk8 = 0xFFFFFFFFu;
  auto search = [&](uint32_t p_lo, uint32_t p_hi) {  // parameters p_lo..p_hi in groups of 4 (see rice_search)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int q = 0; q < 7; ++q) pk[k][q] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < K / 2; ++j) pk7[j] = 0xFFFFFFFFu;
    pk8 = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t p_base = p_lo; p_base <= p_hi; p_base += 4u) {
      uint32_t top[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        rice_build_tables<true>(ps[k], nullptr, len0[k], p_base, max_p, (k == 0) ? lane : 1, warm, top[k]);
        rice_group_levels(top[k], pk[k], p_base, finest_only);
      }
      if (!finest_only) {
        // levels that merge whole passes: lane 0 of the wave holds every pass's merged table
#pragma unroll
        for (int j = 0; j < K / 2; ++j) {
          uint32_t packed = pk7[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t v = top[2 * j][q] + top[2 * j + 1][q];
            v = v < kWMax ? v : kWMax;
            top[2 * j][q] = v;
            const uint32_t c = (v << 5) | (p_base + (uint32_t)q);
            packed = c < packed ? c : packed;
          }
          pk7[j] = packed;
        }
        if (K == 4) {
          uint32_t packed = pk8;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t v = top[0][q] + top[2][q];
            v = v < kWMax ? v : kWMax;
            const uint32_t c = (v << 5) | (p_base + (uint32_t)q);
            packed = c < packed ? c : packed;
          }
          pk8 = packed;
        }
      }
    }
  };
  // rice_window (see the 4096 kernel): the wave-minimum of floor(log2(mean + 1)) over all partitions
  uint32_t p0l = 31u, p0h = 0u;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t s0 = 2u * ps[k].sum_m + ps[k].negs;
    const uint32_t q0 = (s0 >> 6) + 1u;
    const uint32_t c = 31u - (uint32_t)__builtin_clz(q0);
    p0l = c < p0l ? c : p0l;
    // upper end (see the 4096 kernel): q bounds the partition's mean from above; the block's first partition
    // has only 64 - warm >= 32 coded samples: twice the 64-sample mean covers it
    const uint32_t qh = (k == 0 && lane == 0) ? 2u * q0 : q0;
    const uint32_t ch = 31u - (uint32_t)__builtin_clz(qh);
    p0h = ch > p0h ? ch : p0h;
  }
  const uint32_t p0min = wave_min_dpp(p0l);
  uint32_t p_lo = p0min > 2u ? p0min - 2u : 0u;
  p_lo = p_lo < max_p ? p_lo : max_p;
  uint32_t p_hi = wave_max_dpp(p0h) + 1u;
  p_hi = p_hi < max_p ? p_hi : max_p;
  if (literal) p_lo = 0u;

  // level totals; strict < keeps the finer order on ties (rice.rs:285)
  int bestl = 0;
  unsigned long long best_bits = 0;
  uint32_t sat_levels = 0;
  auto totals = [&]() {
    sat_levels = 0;
#pragma unroll
    for (int L = 0; L < NLEV; ++L) {
      if (L > 0 && finest_only) break;
      unsigned long long tot = 0;
      uint32_t sat = 0;
      if (L < 7) {
        const bool lead = (lane & ((1 << L) - 1)) == 0;
        uint32_t lbsum = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t bits = (pk[k][L] >> 5) + 4u;
          const uint32_t lb = lead ? bits : 0u;
          sat |= (lead && bits >= kMaxPToBits) ? 1u : 0u;
          // with the search not cut short by the configuration every minimum is <= 4 + 64 (bitlen + 1) < 2^12:
          // the passes' values are added in the lane and summed over the wave once
          if (small_bits) lbsum += lb;
          else tot += ((unsigned long long)wave_sum_dpp(lb >> 16) << 16) + wave_sum_dpp(lb & 0xFFFFu);
        }
        if (small_bits) tot = wave_sum_dpp(lbsum);
        sat = wave_or_dpp(sat);
      } else if (L == 7) {
#pragma unroll
        for (int j = 0; j < K / 2; ++j) {
          const uint32_t bits = (uint32_t)uni((int)((pk7[j] >> 5) + 4u));
          sat |= bits >= kMaxPToBits ? 1u : 0u;
          tot += bits;
        }
      } else {
        const uint32_t bits = (uint32_t)uni((int)((pk8 >> 5) + 4u));
        sat |= bits >= kMaxPToBits ? 1u : 0u;
        tot = bits;
      }
      sat_levels |= sat << L;
      if (L == 0 || tot < best_bits) {
        best_bits = tot;
        bestl = L;
      }
    }
  };
  if (!literal) {
    search(p_lo, p_hi);
    totals();
    // a saturated minimum could tie with clamped entries outside the window: search the whole range
    if (sat_levels != 0 && p_lo != 0) {
      search(0u, max_p);
      totals();
    }
  }
  const bool saturated = (sat_levels >> bestl) & 1u;
  const int rice_order = (6 + LK) - bestl;
  const uint32_t best_parts = 1u << rice_order;

  // the parameter of the chosen-order partition each (pass, lane) leads
  uint32_t myp[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    uint32_t v = 0;
#pragma unroll
    for (int L = 0; L < 7; ++L) v = (L == bestl) ? (pk[k][L] & 31u) : v;
    if (bestl == 7) v = (uint32_t)uni((int)(pk7[k >> 1] & 31u));
    if (bestl == 8) v = (uint32_t)uni((int)(pk8 & 31u));
    myp[k] = v;
  }
  // Residual::sum_quotients / count_bits (datatype.rs:2325-2331, bitrepr.rs:533-544)
  const int lanebits = bestl < 6 ? bestl : 6;
  const bool lane_leader = (lane & ((1 << lanebits) - 1)) == 0;
  uint32_t sum_p = 0, rice2 = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool pass_leader = bestl <= 6 || (bestl == 7 && (k & 1) == 0) || (bestl == 8 && k == 0);
    const bool leader = lane_leader && pass_leader;
    sum_p += wave_sum_dpp(leader ? myp[k] : 0u);
    rice2 |= wave_or_dpp((leader && myp[k] > 14) ? 1u : 0u);
  }
  const uint32_t p_first = (uint32_t)uni((int)myp[0]);
  const unsigned long long rem_bits = (unsigned long long)sum_p * (unsigned long long)(n >> rice_order) -
                                      (unsigned long long)warm * p_first;
  unsigned long long sum_q;
  if (saturated) {
    // exact quotient sum from the planes under each partition's parameter
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t gp = (uint32_t)__shfl((int)myp[k], lane & ~((1 << lanebits) - 1), 64);
      if (bestl == 7) gp = (uint32_t)uni((int)(pk7[k >> 1] & 31u));
      if (bestl == 8) gp = (uint32_t)uni((int)(pk8 & 31u));
      const unsigned long long mine = plane_sum_any64(ps[k], gp);
      acc += ((unsigned long long)wave_sum_dpp((uint32_t)(mine >> 16)) << 16) +
             (unsigned long long)wave_sum_dpp((uint32_t)(mine & 0xFFFFu));
    }
    sum_q = acc;
  } else {
    sum_q = best_bits - 4ull * best_parts - (unsigned long long)(n - warm) - rem_bits;
  }
  const unsigned long long residual_bits = 2ull + 4ull + (unsigned long long)best_parts * (rice2 ? 5ull : 4ull) +
                                           (sum_q + (unsigned long long)(n - warm)) + rem_bits;
  // Lpc::count_bits (bitrepr.rs:492-499); as fixed_lpc's coder: FixedLpc::count_bits (no precision / shift /
  // coefficient fields)
  const unsigned long long sub_bits = a.fixed_mode != 0
      ? 8ull + bps_role * (unsigned long long)warm + residual_bits
      : 8ull + bps_role * (unsigned long long)warm + 4ull + 5ull +
            (unsigned long long)a.precision * (unsigned long long)warm + residual_bits;

  if (a.minmax_out != nullptr) {
    const int mx = (int)(wave_max_dpp((uint32_t)vmax ^ 0x80000000u) ^ 0x80000000u);
    const int mn = (int)(wave_min_dpp((uint32_t)vmin ^ 0x80000000u) ^ 0x80000000u);
    if (active && lane == 0) {
      a.minmax_out[(size_t)sf * 2 + 0] = mn;
      a.minmax_out[(size_t)sf * 2 + 1] = mx;
    }
  }
  if (!active) return;
  flacenc_hip_subframe_params* rec = a.params + sf;
  if (literal) {
    // marker for the dispatcher: this subframe has to go through the generic kernel's literal tables
    if (lane == 0) rec->status = -1;
    return;
  }
  // partition j of the chosen order: pass (j << bestl) >> 6, lane (j << bestl) & 63
  {
    const uint32_t ok = status == 0 ? 1u : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t j = (uint32_t)(lane + 64 * r);
      const uint32_t first = (j << bestl) & (uint32_t)(64 * K - 1);  // finest-partition index of the first member
      const uint32_t src_lane = first & 63u, src_pass = first >> 6;
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {  // (every lane takes part in every shuffle)
        const uint32_t got = (uint32_t)__shfl((int)myp[k], (int)src_lane, 64);
        v = (src_pass == (uint32_t)k) ? got : v;
      }
      if (j >= best_parts) v = 0;
      rec->rice_params[j] = (uint8_t)(ok ? v : 0u);
    }
  }
  if (lane < 32) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (i == lane) c = cq[i];
    rec->coefs[lane] = (status == 0) ? (int16_t)c : (int16_t)0;
  }
  if (lane == 0) {
    rec->order = (uint8_t)warm;
    rec->shift = (int8_t)shift;
    rec->precision = (uint8_t)(a.fixed_mode != 0 ? 0u : a.precision);
    rec->rice_order = (uint8_t)(status == 0 ? rice_order : 0);
    rec->status = status;
    rec->code_bits = status == 0 ? best_bits : 0ull;
    rec->subframe_bits = status == 0 ? sub_bits : 0ull;
    rec->sum_quotients = status == 0 ? sum_q : 0ull;
  }
}

// ---------------------------------------------------------------------------------------------
// fixed_lpc's order selection (OrderSel::ApproxEnt, coding.rs:265-287) for the big-block shapes: per pass
// the exact sums of |e_k| over every estimator partition (v_sad_u32 on biased values, as in the fused
// 4096 kernel), estimate_entropy per partition, the estimates added up over partitions and passes, the
// first minimum of estimate + bps * order.  Writes the chosen order as a predictor record -- FIXED_LPC_COEFS
// (decode.rs:179-185), shift 0 -- that bigblock_residual_kernel<8, ..> turns into residual, Rice partition and
// bit counts exactly as the generic kernel's fixed mode does with its QLPC machinery.
template <bool STEREO>
__global__ void __launch_bounds__(256, 2) bigblock_fixed_select_kernel(QlpcKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const uint32_t blk = blockIdx.x;
  uint32_t sf = blk * 4u + (uint32_t)wave;
  const bool active = sf < a.n_subframes;
  if (!active) sf = a.n_subframes - 1u;
  const int role = STEREO ? wave : 0;
  const int n = (int)a.block_size;
  const int K = n / kPass;
  const int32_t* const bufA = sm + (STEREO ? (role == 1 ? 1 : 0) : wave) * kBufDwords;
  const int32_t* const bufB = sm + kBufDwords;
  const int tl = lane << 6;
  const unsigned long long bps_role = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + ((STEREO && role == 3) ? 1u : 0u));
  // estimator partitions of n / parts samples = groups of 2^g lanes of a pass (host: a power of two, 64..4096)
  const int psize = n / (int)a.fixed_partitions;
  const int g = 31 - __builtin_clz((unsigned)(psize >> 6));
  const int G = 1 << g;
  const int jsub = lane & (G - 1);
  const int max_order = (int)a.fixed_max_order;
  uint32_t acc_pb[5] = {0u, 0u, 0u, 0u, 0u};  // lane j of a group takes order r G + j in round r
  const bool ref_sums = a.sumabs_in != nullptr;  // FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: sumabs_reference_kernel's chains
  for (int k = 0; k < K; ++k) {
    double ls[5];
    if (ref_sums) {
      const float* __restrict__ sref = a.sumabs_in + (size_t)sf * (5 * 64) + (k * (64 >> g) + (lane >> g));
#pragma unroll
      for (int ord = 0; ord < 5; ++ord) ls[ord] = (double)sref[ord * 64];
    } else {
    __syncthreads();
    bigblock_load_pass<STEREO>(a, sm, blk, k, tid, wave, lane, sf);
    __syncthreads();
    auto sums = [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      uint32_t b[68];  // the lane's 64 samples + 4 in front of them (zeros in front of the block), biased by 2^31
#pragma unroll
      for (int q = 0; q < 17; ++q) {
        const int4 v = bigblock_ld4<KIND>(bufA, bufB, tl - 4 + 4 * q);
        b[4 * q + 0] = (uint32_t)v.x ^ 0x80000000u;
        b[4 * q + 1] = (uint32_t)v.y ^ 0x80000000u;
        b[4 * q + 2] = (uint32_t)v.z ^ 0x80000000u;
        b[4 * q + 3] = (uint32_t)v.w ^ 0x80000000u;
      }
#pragma unroll
      for (int ord = 0; ord < 5; ++ord) {
        uint32_t c[4] = {0u, 0u, 0u, 0u};
        if (ord == 0) {
#pragma unroll
          for (int j = 0; j < 64; ++j) c[j >> 4] = sad_u32(b[4 + j], 0x80000000u, c[j >> 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 64; ++j) c[j >> 4] = sad_u32(b[4 + j], b[3 + j], c[j >> 4]);
          if (ord < 4) {
#pragma unroll
            for (int i = 67; i >= ord; --i) b[i] = xad_u32(b[i - 1], 0x7FFFFFFFu, b[i]);  // (bias 2^31 - 1 from here on)
          }
        }
        ls[ord] = ((double)c[0] + (double)c[1]) + ((double)c[2] + (double)c[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (STEREO && role == 2) sums(std::integral_constant<int, 2>{});
    else if (STEREO && role == 3) sums(std::integral_constant<int, 3>{});
    else sums(std::integral_constant<int, 0>{});
#pragma unroll 1
    for (int lvl = 0; lvl < g; ++lvl) {
#pragma unroll
      for (int ord = 0; ord < 5; ++ord) ls[ord] += __shfl_xor(ls[ord], 1 << lvl, 64);
    }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      if (r * G <= max_order) {
        const int ord = r * G + jsub;
        double sv = ls[0];
#pragma unroll
        for (int q = 1; q < 5; ++q) sv = (q == ord) ? ls[q] : sv;
        // sample_count = min(end - warmup, partition_len): only the block's first partition loses the warm-up
        const uint32_t cnt = (uint32_t)psize - ((k == 0 && (lane >> g) == 0) ? (uint32_t)ord : 0u);
        acc_pb[r] += ord <= max_order ? approx_ent_bits(sv, cnt) : 0u;
      }
    }
  }
  uint32_t best_packed = 0xFFFFFFFFu;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    if (r * G <= max_order) {
      const int ord = r * G + jsub;
      const bool valid = ord <= max_order;
      uint32_t pb = acc_pb[r];  // every lane of a group holds its partitions' estimates: one copy per group
#pragma unroll 1
      for (int lvl = g; lvl < 6; ++lvl) pb += (uint32_t)__shfl_xor((int)pb, 1 << lvl, 64);
      const unsigned long long key = (unsigned long long)pb + bps_role * (unsigned long long)ord;
      if (a.fixed_keys && active && valid && lane < G) a.fixed_keys[(size_t)sf * 8 + ord] = key;
      const uint32_t packed = valid ? (((uint32_t)key << 3) | (uint32_t)ord) : 0xFFFFFFFFu;  // key < 2^29
      const uint32_t m = wave_min_dpp(packed);
      best_packed = m < best_packed ? m : best_packed;
    }
  }
  if (!active || lane != 0) return;
  const int kord = (int)(best_packed & 7u);
  if (a.selector_keys) a.selector_keys[sf] = (unsigned long long)(best_packed >> 3);
  int32_t* pr = a.pred_out + (size_t)sf * 36;
  for (int i = 0; i < 36; ++i) pr[i] = 0;
  pr[0] = kord;  // FIXED_LPC_COEFS[k]: 0 / 1 / 2,-1 / 3,-3,1 / 4,-6,4,-1
  pr[1] = kord == 2 ? -1 : (kord == 3 ? -3 : (kord == 4 ? -6 : 0));
  pr[2] = kord == 3 ? 1 : (kord == 4 ? 4 : 0);
  pr[3] = kord == 4 ? -1 : 0;
  pr[32] = kord;
}

template <typename KernelT>
hipError_t launch_big(KernelT kern, DynamicLdsOptIn& opt_in, const QlpcKernelArgs& a, size_t smem, hipStream_t stream) {
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  const uint32_t blocks = a.stereo ? a.n_subframes / 4u : (a.n_subframes + 3u) / 4u;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, stream, a);
  return hipGetLastError();
}

template <int HP, int NG, int NLAGS = HP + 1>
hipError_t launch_acorr(const QlpcKernelArgs& a, hipStream_t stream) {
  static DynamicLdsOptIn opt_s, opt_p;
  const size_t part = 4 * 3 * (HP + 1) * sizeof(double);
  const size_t cross = 4 * 13 * 64 * sizeof(double);  // wave_tree_sums_lds_n
  if (a.stereo) return launch_big(bigblock_acorr_kernel<HP, NG, true, NLAGS>, opt_s, a, 2 * kHBufDwords * 4 + part + cross, stream);
  return launch_big(bigblock_acorr_kernel<HP, NG, false, NLAGS>, opt_p, a, 4 * kHBufDwords * 4 + part + cross, stream);
}

template <int MAXP, int K, bool FIXD = false>
hipError_t launch_residual(const QlpcKernelArgs& a, hipStream_t stream) {
  static DynamicLdsOptIn opt_s, opt_p;
  if (a.stereo) return launch_big(bigblock_residual_kernel<MAXP, true, K, FIXD>, opt_s, a, 2 * kBufDwords * 4, stream);
  return launch_big(bigblock_residual_kernel<MAXP, false, K, FIXD>, opt_p, a, 4 * kBufDwords * 4, stream);
}

}  // namespace

// fixed_lpc with OrderSel::ApproxEnt on the same shapes: estimator partitions that are whole groups of lanes
bool bigblock_fixed_eligible(const QlpcKernelArgs& a) {
  if (a.fixed_mode != 1u || a.lpc_stage != 0 || a.force_generic) return false;
  if (a.block_size != 4096 && a.block_size != 8192 && a.block_size != 16384) return false;
  const uint32_t parts = a.fixed_partitions;
  if (parts == 0 || (parts & (parts - 1)) != 0) return false;
  const uint32_t psize = a.block_size / parts;
  if (psize < 64 || psize > 4096) return false;
  if (a.fixed_max_order > 4) return false;
  if (a.frame_results || a.chan_results || a.pack_out) return false;
  if ((reinterpret_cast<uintptr_t>(a.samples) & 15) || (a.stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(a.residual) & 15) || (a.residual_stride & 3)) return false;
  if (a.stereo && (a.n_subframes & 3)) return false;
  if (a.split_scratch == nullptr) return false;
  return true;
}

hipError_t launch_bigblock_fixed_select(const QlpcKernelArgs& a, hipStream_t stream) {
  static DynamicLdsOptIn opt_s, opt_p;
  if (a.stereo) return launch_big(bigblock_fixed_select_kernel<true>, opt_s, a, 2 * kBufDwords * 4, stream);
  return launch_big(bigblock_fixed_select_kernel<false>, opt_p, a, 4 * kBufDwords * 4, stream);
}

hipError_t launch_bigblock_fixed_residual(const QlpcKernelArgs& a, hipStream_t stream) {
  const int k = (int)(a.block_size / 4096u);
  return k == 1 ? launch_residual<8, 1, true>(a, stream)
                : (k == 2 ? launch_residual<8, 2, true>(a, stream) : launch_residual<8, 4, true>(a, stream));
}

bool bigblock_eligible(const QlpcKernelArgs& a) {
  // (4096 at orders 13..32 too: below 13 the fused 4096 kernel has it, above it would fall to the generic one;
  // 8192 / 16384 at every order -- round 2 left orders up to 12 on those blocks to the generic kernel, 88 G
  // samples/s where order 24 ran at 146)
  if (a.lpc_order < 1 || a.lpc_order > 32) return false;
  if (a.lpc_order < 13 && a.block_size == 4096) return false;
  return bigblock_shape_eligible(a);
}

bool bigblock_shape_eligible(const QlpcKernelArgs& a) {
  if (a.block_size != 4096 && a.block_size != 8192 && a.block_size != 16384) return false;
  if (a.lpc_order < 1 || a.lpc_order > 32) return false;
  if (a.fixed_mode != 0 || a.lpc_stage != 0 || a.force_generic) return false;
  if (a.frame_results || a.chan_results || a.pack_out) return false;
  if ((reinterpret_cast<uintptr_t>(a.samples) & 15) || (a.stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(a.residual) & 15) || (a.residual_stride & 3)) return false;
  if (a.stereo && (a.n_subframes & 3)) return false;
  if (a.split_scratch == nullptr) return false;
  return true;
}

// R[] (unless `have_r`: already in `racc`, e.g. from the reference-order kernel) into racc
hipError_t launch_bigblock_acorr(const QlpcKernelArgs& a, hipStream_t stream) {
  // order <= 24: all 25 lags in one group; up to 32: 17 + 16 (window of 48 doubles + two accumulator sets
  // must fit 256 VGPRs: all 33 in one group compile to 256 registers + 5 spilled, and run 2.4 x slower)
  if (a.lpc_order <= 8) return launch_acorr<8, 1>(a, stream);    // 9 lags, window of 8
  if (a.lpc_order <= 12) return launch_acorr<16, 1, 13>(a, stream);  // 13 lags, window of 16
  if (a.lpc_order <= 16) return launch_acorr<16, 1>(a, stream);  // 17 lags, window of 16
  if (a.lpc_order <= 24) return launch_acorr<24, FLACENC_BIG_NG24>(a, stream);
  return launch_acorr<32, FLACENC_BIG_NG32>(a, stream);
}

hipError_t launch_bigblock_residual(const QlpcKernelArgs& a, hipStream_t stream) {
  const int k = (int)(a.block_size / 4096u);
  if (a.lpc_order <= 8)
    return k == 1 ? launch_residual<8, 1>(a, stream) : (k == 2 ? launch_residual<8, 2>(a, stream) : launch_residual<8, 4>(a, stream));
  if (a.lpc_order <= 10)  // (the reference's default order)
    return k == 1 ? launch_residual<10, 1>(a, stream) : (k == 2 ? launch_residual<10, 2>(a, stream) : launch_residual<10, 4>(a, stream));
  if (a.lpc_order <= 12)
    return k == 1 ? launch_residual<12, 1>(a, stream) : (k == 2 ? launch_residual<12, 2>(a, stream) : launch_residual<12, 4>(a, stream));
  if (a.lpc_order <= 16)
    return k == 1 ? launch_residual<16, 1>(a, stream) : (k == 2 ? launch_residual<16, 2>(a, stream) : launch_residual<16, 4>(a, stream));
  if (a.lpc_order <= 24)
    return k == 1 ? launch_residual<24, 1>(a, stream) : (k == 2 ? launch_residual<24, 2>(a, stream) : launch_residual<24, 4>(a, stream));
  return k == 1 ? launch_residual<32, 1>(a, stream) : (k == 2 ? launch_residual<32, 2>(a, stream) : launch_residual<32, 4>(a, stream));
}

}  // namespace flacenc_hip
