// qlpc_bigblock_residual_impl.h -- bigblock_residual_kernel: residual + exhaustive partitioned-Rice search for blocks
// of 4096 / 8192 / 16384 samples behind a predictor record (src/lpc.rs:306-390, src/rice.rs:65-298), with
// compute_error's FIR on the matrix cores.
//
// The residual
//   e[t] = s[t] - ((sum_{j < P} c_j s[t-1-j]) >> shift)  =  -(X[t] >> shift),
//   X[t] = sum_{j = -1}^{P-1} c'_j s[t-1-j],   c'_{-1} = -2^shift   (the "s[t] -" rides along as one more tap)
// is a banded-Toeplitz product: a TILE is 16 columns, each one 16-sample chunk c(n), and
//   D[i][n] = X[16 c(n) + i] = sum_m A[i][m] B[m][n],   A[i][m] = c'[i - 1 - m],   B[m][n] = s[16 c(n) + m],
// m = -32..31 -- one v_mfma_i32_16x16x64_i8 per pair of byte limbs, A the same for the whole subframe (built once from
// the predictor record), whatever the order up to 32.  Everything is exact integer arithmetic:
//   * samples are staged in LDS as byte PLANES (plane a = byte a of every sample; 16 consecutive samples = one
//     16-byte B fragment), complemented (~s = -s - 1) and with the lower bytes biased by -128 so that they read as
//     i8: s~ = sum_a 256^a l_a + bias, l_a in [-128, 127]; the bias times sum c' is a per-subframe constant that
//     enters through the accumulators' initial values (the C operand of each weight's first MFMA);
//   * coefficients (|c| <= 2^14, the extra tap -2^shift >= -2^15) are two signed digits c = c0 + 256 c1;
//   * limb products of equal weight 256^(a+b) chain through C: NL sample limbs cost 2 NL MFMAs and leave NL + 1
//     accumulators a_w; with the seeds they hold Y = -X + 2^shift - 1 = hi 2^16 + lo, lo = a0 + (a1 << 8),
//     hi = a2 + (a3 << 8) + (a4 << 16) (mod 2^32), and e = -(X >> shift) = Y >> shift
//       = (hi << (16 - shift)) + (lo >> shift)   -- four 32-bit instructions per sample, no 64-bit arithmetic
//     (hi 2^16 is a multiple of 2^shift, so the floor splits; |lo| < 2^29; e is the exact value mod 2^32, i.e. the
//     truncation to i32 of lpc.rs:379-388).
// NLB = limbs of the rows L, R, M (or of every plain row) = ceil(bits / 8); the 17- / 25-bit side channel of 16- /
// 24-bit material takes one more.  tools/microbench/mfma_i8_fir.hip measured the form against the v_mad_i64_i32
// window walk this kernel used until round 4: 4.5 x at order 24, 7.6 x at order 32, 3.6 x with four limbs
// (profiles/r04_mfma_i8_fir.txt) -- the FIR's arithmetic is no longer what the kernel's time is made of.
//
// Lane map of a tile (C/D layout of the instruction): lane l = 16 kb + n holds rows 4 kb .. 4 kb + 3 of column n.
// Tile T = 4 Q + r of a pass takes column n from chunk 64 Q + 4 n + r, so over the four tiles of one Q the lanes
// {n, n + 16, n + 32, n + 48} see exactly the 64 samples of finest Rice partition 16 Q + n.  Each lane counts its 16
// residuals per Q into 5 bit planes; after the pass a two-level exchange (v_permlane16_swap, v_permlane32_swap: a
// reduce-scatter over the four lanes) leaves lane 16 Q + n with the 7-plane sum of partition 16 Q + n -- lane l holds
// partition l, the layout the Rice search below was written for.
#ifndef FLACENC_HIP_QLPC_BIGBLOCK_RESIDUAL_IMPL_H_
#define FLACENC_HIP_QLPC_BIGBLOCK_RESIDUAL_IMPL_H_

#include <type_traits>

#include "frame_decide_device.h"
#include "qlpc_kernel.h"
#include "qlpc_wave_kernel_impl.h"

namespace flacenc_hip {
namespace {

#ifndef FLACENC_BIG_RESID_OCC
#define FLACENC_BIG_RESID_OCC 2  // (two passes and more: 212 registers unconstrained; at 168 the pass-ahead loads spill)
#endif
#ifndef FLACENC_BIG_OCC3_MODE
#define FLACENC_BIG_OCC3_MODE 2  // (the deciding store pass needs 156 registers)
#endif
constexpr int kBigPass = 4096;
constexpr int kLimbPlane = 32 + kBigPass;  // bytes: the 32 samples in front of the pass, then the pass
typedef int v4i __attribute__((ext_vector_type(4)));

// byte planes of four consecutive samples (a 4 x 4 byte transpose by v_perm_b32; selector byte k picks byte k of
// the result from {src0 : src1} = bytes 7..4 : 3..0), stored through the xor masks that complement every byte and
// bias all but the top one
template <int NL>
__device__ __forceinline__ void store_limbs4(unsigned char* plane0, int byte, const int4 v) {
  const uint32_t u0 = (uint32_t)v.x, u1 = (uint32_t)v.y, u2 = (uint32_t)v.z, u3 = (uint32_t)v.w;
  const uint32_t a_lo = __builtin_amdgcn_perm(u1, u0, 0x05010400u);  // u0.b0 u1.b0 u0.b1 u1.b1
  const uint32_t b_lo = __builtin_amdgcn_perm(u3, u2, 0x05010400u);
  constexpr uint32_t kLow = 0x7F7F7F7Fu, kTop = 0xFFFFFFFFu;
  *reinterpret_cast<uint32_t*>(plane0 + byte) = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u) ^ (NL == 1 ? kTop : kLow);
  if (NL > 1)
    *reinterpret_cast<uint32_t*>(plane0 + kLimbPlane + byte) = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u) ^ (NL == 2 ? kTop : kLow);
  if (NL > 2) {
    const uint32_t a_hi = __builtin_amdgcn_perm(u1, u0, 0x07030602u);  // u0.b2 u1.b2 u0.b3 u1.b3
    const uint32_t b_hi = __builtin_amdgcn_perm(u3, u2, 0x07030602u);
    *reinterpret_cast<uint32_t*>(plane0 + 2 * kLimbPlane + byte) = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u) ^ (NL == 3 ? kTop : kLow);
    if (NL > 3) *reinterpret_cast<uint32_t*>(plane0 + 3 * kLimbPlane + byte) = __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u) ^ kTop;
  }
}

__device__ __forceinline__ void minmax4(const int4 v, int& mx, int& mn) {
  mx = max(max(mx, v.x), max(v.y, max(v.z, v.w)));
  mn = min(min(mn, v.x), min(v.y, min(v.z, v.w)));
}

// The lanes l, l ^ 16, l ^ 32, l ^ 48 (kb = l >> 4) each hold 5-plane counts c[Q] of their share of partitions
// 16 Q + n, Q = 0..3; lane kb is to end up with the 7-plane sum of partition Q = kb over the four lanes.
// v_permlane16_swap(X, Y) exchanges the odd rows of X with the even rows of Y: with X = c[0], Y = c[1] the even rows
// then hold (own c[0], the odd neighbour's c[0]) and the odd rows (the even neighbour's c[1], own c[1]) -- X + Y is the
// pair's sum of c[0] on even rows and of c[1] on odd rows.  v_permlane32_swap does the same with the two halves.
__device__ __forceinline__ void planes_reduce_scatter(uint32_t (&c)[4][5], uint32_t (&out)[7]) {
  uint32_t s01[6], s23[6];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t x[6], y[6];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const auto r = __builtin_amdgcn_permlane16_swap(c[2 * h][j], c[2 * h + 1][j], false, false);
      x[j] = r[0];
      y[j] = r[1];
    }
    planes_add<5>(x, y);
#pragma unroll
    for (int j = 0; j < 6; ++j) (h == 0 ? s01 : s23)[j] = x[j];
  }
  uint32_t x[7], y[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const auto r = __builtin_amdgcn_permlane32_swap(s01[j], s23[j], false, false);
    x[j] = r[0];
    y[j] = r[1];
  }
  planes_add<6>(x, y);
#pragma unroll
  for (int j = 0; j < 7; ++j) out[j] = x[j];
}

// NLB: byte limbs of the rows L, R, M (stereo) or of every row (plain); the side channel carries one bit more.
// MODE 0: analyse and store (the candidate-level batches: four residual rows per frame leave the CU).
// MODE 1: analyse only -- records, no residual rows (the frame-level pipeline's bit-count passes: one launch behind the
//         QLPC predictors, one behind fixed_lpc's).
// MODE 2: decide and store (stereo): encode_subframe / try_stereo_coding over the records the MODE-1 launches left
//         (frame_decide_device.h), then the TWO chosen (role, predictor) pairs are run through the FIR again and only
//         their rows are written -- eight candidate rows per frame never reach HBM.
template <bool STEREO, int K, int NLB, int MODE>
__global__ void __launch_bounds__(256, ((STEREO && K == 1) || MODE == FLACENC_BIG_OCC3_MODE) ? 3 : FLACENC_BIG_RESID_OCC) bigblock_residual_kernel(QlpcKernelArgs a) {
  static_assert(MODE == 0 || STEREO, "the frame-level modes are stereo");
  constexpr bool ANALYSE = MODE != 2, STORE = MODE != 1;
  constexpr int NLS = STEREO ? (NLB < 4 ? NLB + 1 : 4) : NLB;
  constexpr int kPlanes = 3 * NLB + NLS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const uint32_t blk = blockIdx.x;
  uint32_t sf = blk * 4u + (uint32_t)wave;
  const bool active = sf < a.n_subframes;
  if (!active) sf = a.n_subframes - 1u;
  const int n = (int)a.block_size;
  int* const xch = reinterpret_cast<int*>(smem_raw + kPlanes * kLimbPlane);  // [4 roles][max, min]: LDS atomics meet here
  if (STEREO && tid < 8) xch[tid] = (tid & 1) ? INT32_MAX : INT32_MIN;  // (ordered before their use by the pass loop's barriers)
  // profiling hook (flacenc_hip_debug_set_stamps): slots 0 / 7 wall clock (100 MHz) at entry / exit, 1..6 shader clock
  auto stamp = [&](int slot) {
    if (a.stamps) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      const unsigned long long t = (slot == 0 || slot == 7) ? (unsigned long long)__builtin_amdgcn_s_memrealtime()
                                                            : (unsigned long long)__builtin_amdgcn_s_memtime();
      if (lane == 0) a.stamps[(size_t)sf * 8 + slot] = t;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  stamp(1);
  // which row of the frame this wave filters, behind which predictor, and where its residual goes
  int role = STEREO ? wave : 0;
  int warm, shift, status;
  const int32_t* __restrict__ pr = nullptr;                        // MODE 0 / 1: qc[32], order, shift, status
  const flacenc_hip_subframe_params* __restrict__ prec = nullptr;  // MODE 2: the chosen candidate's record
  int32_t* __restrict__ rrow = nullptr;
  uint32_t chosen_kind = FLACENC_HIP_KIND_LPC;
  if (MODE == 2) {
    FrameDecision* const sdec = reinterpret_cast<FrameDecision*>(smem_raw + kPlanes * kLimbPlane + 32);
    FrameCandidates cand;
    cand.block_size = a.block_size;
    cand.bits_per_sample = a.bps_uniform;
    cand.use_constant = a.use_constant;
    cand.use_fixed = a.use_fixed;
    cand.use_lpc = a.use_lpc;
    cand.use_leftside = a.use_leftside;
    cand.use_rightside = a.use_rightside;
    cand.use_midside = a.use_midside;
    cand.lpc_params = a.cand_lpc_params;
    cand.fixed_params = a.cand_fixed_params;
    cand.fixed_keys = a.cand_fixed_keys;
    int lo = 0, hi = 0;
    if (tid < 4) {
      lo = a.cand_minmax[((size_t)blk * 4 + tid) * 2 + 0];
      hi = a.cand_minmax[((size_t)blk * 4 + tid) * 2 + 1];
    }
    decide_frame(cand, blk, tid, 256, lo, hi, *sdec, a.frame_results);
    // waves 0 / 1 = output channels 0 / 1 (waves 2, 3 only help staging the planes)
    const int ch = wave & 1;
    role = uni((int)sdec->choice[1 + ch]);
    chosen_kind = (uint32_t)uni((int)sdec->kind[role]);
    const size_t csf = (size_t)blk * 4 + (size_t)role;
    prec = chosen_kind == FLACENC_HIP_KIND_LPC ? a.cand_lpc_params + csf
                                               : (chosen_kind == FLACENC_HIP_KIND_FIXED ? a.cand_fixed_params + csf : nullptr);
    warm = prec ? uni((int)prec->order) : 0;
    shift = prec ? uni((int)prec->shift) : 0;
    status = prec ? 0 : 1;  // Constant / Verbatim: a zero predictor, zero rows
    rrow = a.residual + (size_t)(2u * blk + (uint32_t)ch) * a.residual_stride;
  } else {
    pr = a.pred + (size_t)sf * 36;
    warm = uni(pr[32]);
    shift = uni(pr[33]);
    status = uni(pr[34]);
    rrow = a.residual + (size_t)sf * a.residual_stride;
    if (STEREO && a.residual_lr != nullptr && role < 2)  // L / R candidates in place of the output channel they can fill
      rrow = a.residual_lr + (size_t)(2u * blk + (uint32_t)role) * a.residual_lr_stride;
  }
  const unsigned long long bps_role = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + ((STEREO && role == 3) ? 1u : 0u));
  const int NL = (STEREO && role == 3) ? NLS : NLB;
  unsigned char* const myplanes = smem_raw + (STEREO ? role : wave) * (NLB * kLimbPlane);

  // ---- A operand: row i = lane & 15, k = 16 kb + j in byte j; tap(i, k) = i - 1 - (k - 32).  The taps come from a
  // zero-padded table in LDS (it overlays the planes: they are filled behind the first barrier of the pass loop)
  const int fi = lane & 15, kb = lane >> 4;
  v4i A0, A1;
  int csum;
  {
    int32_t* const ctab = reinterpret_cast<int32_t*>(smem_raw) + wave * 128;  // tap j at [48 + j], j = -1 .. 31
    ctab[lane] = 0;
    ctab[lane + 64] = 0;
    if (status == 0) {
      if (lane < 32) ctab[48 + lane] = MODE == 2 ? (int32_t)prec->coefs[lane] : pr[lane];
      if (lane == 32) ctab[47] = -(1 << shift);
    }
    uint32_t w0[4], w1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t p0 = 0, p1 = 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int c = ctab[79 + fi - 16 * kb - (4 * q + jj)];
        const int c0 = (int)(int8_t)(c & 0xFF);
        const int c1 = (c - c0) >> 8;
        p0 |= (uint32_t)(c0 & 0xFF) << (8 * jj);
        p1 |= (uint32_t)(c1 & 0xFF) << (8 * jj);
      }
      w0[q] = p0;
      w1[q] = p1;
    }
    A0 = v4i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3]};
    A1 = v4i{(int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
    csum = (int)wave_sum_dpp(lane <= 32 ? (uint32_t)ctab[47 + lane] : 0u);  // sum of c' over the 33 taps
  }
  // ---- accumulator seeds: the planes hold ~s = -s - 1 with the lower bytes biased, so the limb sums give
  // -X - (bias + 1) csum; the seeds add that back, and 2^shift - 1 for the ceiling (file comment):
  //   NL = 1: bias 0;  2: 128;  3: 128 + 2^15 -> a1 += (csum & 1) << 7, a2 += csum >> 1;  4: + 2^23 -> a2, a3 likewise
  v4i seed0, seed1, seed2, seed3;
  {
    const int half = csum >> 1, odd = (csum & 1) << 7;
    const int s0 = (NL >= 2 ? 129 : 1) * csum + (1 << shift) - 1;
    const int s1 = NL >= 3 ? odd : 0;
    const int s2 = NL == 3 ? half : (NL == 4 ? half + odd : 0);
    const int s3 = NL == 4 ? half : 0;
    seed0 = v4i{s0, s0, s0, s0};
    seed1 = v4i{s1, s1, s1, s1};
    seed2 = v4i{s2, s2, s2, s2};
    seed3 = v4i{s3, s3, s3, s3};
    // (opaque: as wave-uniform values the compiler rebuilds the quads from SGPRs in front of every MFMA)
    asm volatile("" : "+v"(seed0), "+v"(seed1), "+v"(seed2), "+v"(seed3));
  }
  const int lsh = 16 - shift;

  // min / max of the rows.  The byte planes carry samples of the declared width only: a row outside it sends the
  // subframe to the generic kernel (below).  mid lies between l and r and side within their spread, so that check
  // needs L and R only; is_constant's input at frame level (minmax_out) wants every role's own extremes.
  int mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN}, mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
  const bool track_ms = MODE == 2 || a.minmax_out != nullptr;  // (MODE 2: the same criterion as the MODE-1 launches that marked)

  uint32_t pl[K][7];
  // Stereo: a pass's loads (the thread's four quads of L and of R: 8 int4 in plain variables -- a struct or array of
  // them stays in scratch and is waited for at once) are issued one pass ahead and split into planes behind the
  // barrier that ends the previous pass's tile loop.  Plain: every wave stages its own row, no barriers between waves.
  int4 pq0, pq1, pq2, pq3, pq4, pq5, pq6, pq7;
#define FLACENC_PASS_FETCH(K_)                                                                                \
  {                                                                                                            \
    const int32_t* __restrict__ src_ = a.samples + (size_t)(2u * blk) * a.stride + (size_t)(K_) * kBigPass;    \
    pq0 = *reinterpret_cast<const int4*>(src_ + ((tid + 0) << 2));                                             \
    pq1 = *reinterpret_cast<const int4*>(src_ + ((tid + 256) << 2));                                           \
    pq2 = *reinterpret_cast<const int4*>(src_ + ((tid + 512) << 2));                                           \
    pq3 = *reinterpret_cast<const int4*>(src_ + ((tid + 768) << 2));                                           \
    pq4 = *reinterpret_cast<const int4*>(src_ + a.stride + ((tid + 0) << 2));                                  \
    pq5 = *reinterpret_cast<const int4*>(src_ + a.stride + ((tid + 256) << 2));                                \
    pq6 = *reinterpret_cast<const int4*>(src_ + a.stride + ((tid + 512) << 2));                                \
    pq7 = *reinterpret_cast<const int4*>(src_ + a.stride + ((tid + 768) << 2));                                \
  }
#define FLACENC_PASS_SPLIT(I_, L_, R_)                                                                         \
  {                                                                                                            \
    const int byte_ = 32 + ((tid + 256 * (I_)) << 2);                                                          \
    const int4 l_ = L_, r_ = R_;                                                                               \
    const int4 m_ = make_int4((l_.x + r_.x) >> 1, (l_.y + r_.y) >> 1, (l_.z + r_.z) >> 1, (l_.w + r_.w) >> 1); \
    const int4 s_ = make_int4(l_.x - r_.x, l_.y - r_.y, l_.z - r_.z, l_.w - r_.w);  /* coding.rs:483 */         \
    store_limbs4<NLB>(smem_raw, byte_, l_);                                                                    \
    store_limbs4<NLB>(smem_raw + NLB * kLimbPlane, byte_, r_);                                                 \
    store_limbs4<NLB>(smem_raw + 2 * NLB * kLimbPlane, byte_, m_);                                             \
    store_limbs4<NLS>(smem_raw + 3 * NLB * kLimbPlane, byte_, s_);                                             \
    minmax4(l_, mx[0], mn[0]);                                                                                 \
    minmax4(r_, mx[1], mn[1]);                                                                                 \
    if (track_ms) {                                                                                            \
      minmax4(m_, mx[2], mn[2]);                                                                               \
      minmax4(s_, mx[3], mn[3]);                                                                               \
    }                                                                                                          \
  }
  if (STEREO) FLACENC_PASS_FETCH(0)
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    if (STEREO) {
      // the 32 samples in front of the pass: the tail of the previous one (read before the barrier, while it is
      // intact), or zeros in front of the block (a zero sample is stored as its plane's xor mask)
      uint32_t halo = 0;
      if (k > 0 && tid < 8 * kPlanes) halo = *reinterpret_cast<const uint32_t*>(smem_raw + (tid >> 3) * kLimbPlane + kBigPass + ((tid & 7) << 2));
      __syncthreads();  // every wave is done with the previous pass's planes (first pass: with the tap tables)
      if (k > 0) {
        if (tid < 8 * kPlanes) *reinterpret_cast<uint32_t*>(smem_raw + (tid >> 3) * kLimbPlane + ((tid & 7) << 2)) = halo;
      } else if (lane < 8 * NL) {
        const int q = lane >> 3;
        *reinterpret_cast<uint32_t*>(myplanes + q * kLimbPlane + ((lane & 7) << 2)) = (q == NL - 1) ? 0xFFFFFFFFu : 0x7F7F7F7Fu;
      }
      FLACENC_PASS_SPLIT(0, pq0, pq4)
      FLACENC_PASS_SPLIT(1, pq1, pq5)
      FLACENC_PASS_SPLIT(2, pq2, pq6)
      FLACENC_PASS_SPLIT(3, pq3, pq7)
      if (k + 1 < K) FLACENC_PASS_FETCH(k + 1)
      if (k == K - 1) {
        // the workgroup's minima / maxima per role
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r >= 2 && !track_ms) break;
          const int wmx = (int)(wave_max_dpp((uint32_t)mx[r] ^ 0x80000000u) ^ 0x80000000u);
          const int wmn = (int)(wave_min_dpp((uint32_t)mn[r] ^ 0x80000000u) ^ 0x80000000u);
          if (lane == 0) {
            atomicMax(&xch[2 * r], wmx);
            atomicMin(&xch[2 * r + 1], wmn);
          }
        }
      }
      __syncthreads();
    } else {
      uint32_t halo = 0;
      if (k > 0 && lane < 8 * NLB) halo = *reinterpret_cast<const uint32_t*>(myplanes + (lane >> 3) * kLimbPlane + kBigPass + ((lane & 7) << 2));
      if (k == 0) __syncthreads();  // (the tap tables overlay other waves' planes)
      if (lane < 8 * NLB) {
        const int q = lane >> 3;
        *reinterpret_cast<uint32_t*>(myplanes + q * kLimbPlane + ((lane & 7) << 2)) = k > 0 ? halo : ((q == NLB - 1) ? 0xFFFFFFFFu : 0x7F7F7F7Fu);
      }
      const int32_t* __restrict__ src = a.samples + (size_t)sf * a.stride + (size_t)k * kBigPass;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int4 v0 = *reinterpret_cast<const int4*>(src + ((lane + (4 * g + 0) * 64) << 2));
        int4 v1 = *reinterpret_cast<const int4*>(src + ((lane + (4 * g + 1) * 64) << 2));
        int4 v2 = *reinterpret_cast<const int4*>(src + ((lane + (4 * g + 2) * 64) << 2));
        int4 v3 = *reinterpret_cast<const int4*>(src + ((lane + (4 * g + 3) * 64) << 2));
        store_limbs4<NLB>(myplanes, 32 + ((lane + (4 * g + 0) * 64) << 2), v0);
        store_limbs4<NLB>(myplanes, 32 + ((lane + (4 * g + 1) * 64) << 2), v1);
        store_limbs4<NLB>(myplanes, 32 + ((lane + (4 * g + 2) * 64) << 2), v2);
        store_limbs4<NLB>(myplanes, 32 + ((lane + (4 * g + 3) * 64) << 2), v3);
        minmax4(v0, mx[0], mn[0]);
        minmax4(v1, mx[0], mn[0]);
        minmax4(v2, mx[0], mn[0]);
        minmax4(v3, mx[0], mn[0]);
      }
      __builtin_amdgcn_wave_barrier();
    }

    if (k == 0) stamp(2);  // first pass split into planes, barrier passed
    // ---- the pass's 16 tiles
    uint32_t now[7];
    auto tiles = [&](auto nl_tag) {
      constexpr int NLT = decltype(nl_tag)::value;
      uint32_t cnt[4][5];
      const unsigned char* const bcol = myplanes + 64 * fi + 16 * kb;  // chunk 4 n + kb - 2, behind the 32 bytes of halo
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) {
        int32_t e[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // column n of tile 4 Q + r is chunk 64 Q + 4 n + r of the pass
          const unsigned char* bsrc = bcol + 1024 * Q + 16 * r;
          v4i B[NLT];
#pragma unroll
          for (int q = 0; q < NLT; ++q) B[q] = *reinterpret_cast<const v4i*>(bsrc + q * kLimbPlane);
          const v4i zero = {0, 0, 0, 0};
          v4i a0, a1, a2 = zero, a3 = zero, a4 = zero;
          a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[0], seed0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[0], seed1, 0, 0, 0);
          if (NLT >= 2) {
            a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[NLT >= 2 ? 1 : 0], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[NLT >= 2 ? 1 : 0], seed2, 0, 0, 0);
          }
          if (NLT >= 3) {
            a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[NLT >= 3 ? 2 : 0], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[NLT >= 3 ? 2 : 0], seed3, 0, 0, 0);
          }
          if (NLT >= 4) {
            a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[NLT >= 4 ? 3 : 0], a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[NLT >= 4 ? 3 : 0], zero, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int32_t lo = (int32_t)((uint32_t)a0[q] + ((uint32_t)a1[q] << 8));
            uint32_t hi = (uint32_t)a2[q];
            if (NLT >= 3) hi += (uint32_t)a3[q] << 8;
            if (NLT >= 4) hi += (uint32_t)a4[q] << 16;
            e[4 * r + q] = NLT >= 2 ? (int32_t)((hi << lsh) + (uint32_t)(lo >> shift)) : (lo >> shift);
          }
          const int t0 = 1024 * Q + 64 * fi + 16 * r + 4 * kb;  // rows 4 kb .. 4 kb + 3 of the column
          // e[0 .. order') = 0 (lpc.rs:349): chunks 0 and 1 of the block, i.e. column 0 of the first two tiles
          if (Q == 0 && r < 2 && k == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (t0 + q < warm) e[4 * r + q] = 0;
          }
          // (an inactive wave -- plain mode's last workgroup -- recomputes the batch's last subframe and stores the same
          // values to the same row: no branch around the stores, the tiles stay one basic block)
          if (STORE)
            *reinterpret_cast<int4*>(rrow + (size_t)k * kBigPass + t0) = make_int4(e[4 * r], e[4 * r + 1], e[4 * r + 2], e[4 * r + 3]);
          // (the stores pace the tiles; without them the scheduler runs all 96 MFMAs of a pass ahead of the first
          // recombination and keeps every result alive -- 58 spilled registers at 256: an empty volatile asm that "uses"
          // the tile's residuals stands in for the store)
          else asm volatile("" ::"v"(e[4 * r]), "v"(e[4 * r + 1]), "v"(e[4 * r + 2]), "v"(e[4 * r + 3]));
        }
        // the lane's 16 residuals of partition 16 Q + n -> bit-plane counts
        if (ANALYSE) popcount_planes16(e, cnt[Q]);
      }
      if (ANALYSE) planes_reduce_scatter(cnt, now);
    };
    if (MODE != 2 || wave < 2) {
      if (STEREO && role == 3) tiles(std::integral_constant<int, NLS>{});
      else tiles(std::integral_constant<int, NLB>{});
    }
    if (k == 0) stamp(3);  // first pass's tiles
    // (the pass loop is rolled; pl[k] is selected by a compare chain so that the planes stay in registers)
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
      if (kk == k) {
#pragma unroll
        for (int q = 0; q < 7; ++q) pl[kk][q] = now[q];
      }
  }
#undef FLACENC_PASS_FETCH
#undef FLACENC_PASS_SPLIT
  stamp(4);  // all passes
  // the role's minimum / maximum over the whole block; samples outside the declared width (which the byte planes
  // do not carry) send the subframe to the generic kernel like a literal-range residual does
  int vmax, vmin;
  if (STEREO) {
    if (role < 2 || track_ms) {
      vmax = xch[2 * role];
      vmin = xch[2 * role + 1];
    } else if (role == 2) {  // (l + r) >> 1 lies between l and r
      vmax = max(xch[0], xch[2]);
      vmin = min(xch[1], xch[3]);
    } else {  // l - r
      vmax = xch[0] - xch[3];
      vmin = xch[1] - xch[2];
    }
  } else {
    vmax = (int)(wave_max_dpp((uint32_t)mx[0] ^ 0x80000000u) ^ 0x80000000u);
    vmin = (int)(wave_min_dpp((uint32_t)mn[0] ^ 0x80000000u) ^ 0x80000000u);
  }
  vmax = uni(vmax);
  vmin = uni(vmin);
  const int lim = NL >= 4 ? INT32_MAX : (1 << (8 * NL - 1)) - 1;
  const bool out_of_width = vmax > lim || vmin < -lim - 1;
  if (MODE == 2) {
    // a row outside its declared width was marked by the analysing launches and redone by the generic kernel, whose
    // candidate row is the residual: copy it over what the byte planes produced
    if (wave < 2 && out_of_width && prec != nullptr) {
      const int32_t* __restrict__ src = (chosen_kind == FLACENC_HIP_KIND_LPC ? a.cand_lpc_rows : a.cand_fixed_rows) +
                                        ((size_t)blk * 4 + (size_t)role) * a.cand_stride;
      for (int t = lane; t < n; t += 64) rrow[t] = src[t];
    }
    stamp(7);
    return;
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
  const bool literal = !(maxu < (1u << 26)) || out_of_width;

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

  stamp(5);  // Rice search
  if (a.minmax_out != nullptr && active && lane == 0) {
    a.minmax_out[(size_t)sf * 2 + 0] = vmin;
    a.minmax_out[(size_t)sf * 2 + 1] = vmax;
  }
  stamp(7);
  if (!active) return;
  flacenc_hip_subframe_params* rec = a.params + sf;
  if (literal) {
    // marker for the dispatcher: this subframe has to go through the generic kernel's literal tables
    if (lane == 0) {
      rec->status = -1;
      count_marked(a, sf);
    }
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
  if (lane < 32) rec->coefs[lane] = (status == 0 && MODE != 2) ? (int16_t)pr[lane] : (int16_t)0;
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

// LDS: the byte planes of the workgroup's four rows + 32 bytes of min / max exchange (24-bit stereo: 13 planes =
// 53 696 bytes, 42 of the 1280-byte granules: three workgroups per CU); the tap tables, 2 KB, overlay the planes
template <int K, int NLB, int MODE>
hipError_t launch_bigblock_residual_mode(const QlpcKernelArgs& a, hipStream_t stream) {
  static DynamicLdsOptIn opt_s;
  const uint32_t blocks = a.n_subframes / 4u;
  const size_t smem = (size_t)(3 * NLB + (NLB < 4 ? NLB + 1 : 4)) * kLimbPlane + 32 + (MODE == 2 ? 128 : 0);
  if (hipError_t err = opt_s.ensure(reinterpret_cast<const void*>(bigblock_residual_kernel<true, K, NLB, MODE>), smem); err != hipSuccess) return err;
  hipLaunchKernelGGL((bigblock_residual_kernel<true, K, NLB, MODE>), dim3(blocks), dim3(256), smem, stream, a);
  return hipGetLastError();
}

template <int K, int NLB>
hipError_t launch_bigblock_residual_inst(const QlpcKernelArgs& a, hipStream_t stream) {
  if (a.stereo && a.residual_mode == 1u) return launch_bigblock_residual_mode<K, NLB, 1>(a, stream);
  if (a.stereo && a.residual_mode == 2u) return launch_bigblock_residual_mode<K, NLB, 2>(a, stream);
  if (a.residual_mode != 0u) return hipErrorInvalidValue;
  if (a.stereo) return launch_bigblock_residual_mode<K, NLB, 0>(a, stream);
  static DynamicLdsOptIn opt_p;
  const uint32_t blocks = (a.n_subframes + 3u) / 4u;
  const size_t smem = (size_t)(4 * NLB) * kLimbPlane + 32;
  if (hipError_t err = opt_p.ensure(reinterpret_cast<const void*>(bigblock_residual_kernel<false, K, NLB, 0>), smem); err != hipSuccess) return err;
  hipLaunchKernelGGL((bigblock_residual_kernel<false, K, NLB, 0>), dim3(blocks), dim3(256), smem, stream, a);
  return hipGetLastError();
}

}  // namespace
}  // namespace flacenc_hip
#endif
