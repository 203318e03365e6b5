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
  double* const cross = part + 4 * LVMAX * NLAG;                               // [4 waves][NBATCH][kTreeRow]
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
  double* const mycross = cross + wave * NBATCH * kTreeRow;

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
  const size_t cross = 4 * 13 * kTreeRow * sizeof(double);  // wave_tree_sums_lds_n
  if (a.stereo) return launch_big(bigblock_acorr_kernel<HP, NG, true, NLAGS>, opt_s, a, 2 * kHBufDwords * 4 + part + cross, stream);
  return launch_big(bigblock_acorr_kernel<HP, NG, false, NLAGS>, opt_p, a, 4 * kHBufDwords * 4 + part + cross, stream);
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
  return launch_bigblock_residual(a, stream);  // (FIXED_LPC_COEFS with shift 0 is a predictor like any other)
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

// byte limbs of the rows L, R, M (or of every plain row): one width for the whole batch is known here, per-subframe
// widths (a device array) get four limbs, which carry any i32
hipError_t launch_bigblock_residual(const QlpcKernelArgs& a, hipStream_t stream) {
  const int k = (int)(a.block_size / 4096u);
  int nlb = 4;
  if (a.bps == nullptr) nlb = a.bps_uniform <= 16u ? 2 : (a.bps_uniform <= 24u ? 3 : 4);
#define FLACENC_HIP_BIGRES(K_, NLB_) \
  if (k == K_ && nlb == NLB_) return launch_bigblock_residual_##K_##_##NLB_(a, stream);
  FLACENC_HIP_FOR_EACH_BIGRES_INSTANCE(FLACENC_HIP_BIGRES)
#undef FLACENC_HIP_BIGRES
  return hipErrorInvalidValue;
}

}  // namespace flacenc_hip
