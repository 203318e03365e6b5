// qlpc_subwave_kernel_impl.h -- the fused QLPC kernel for blocks smaller than one wave's worth of finest Rice
// partitions: 256 / 512 / 1024 / 2048 (64 samples per lane) and 288 / 576 / 1152 / 2304 (72 per lane, the CD-style sizes).
//
// qlpc_wave_kernel_impl.h gives a subframe of 4096 (4608) samples one wave: lane l holds one finest Rice partition
// (rice.rs:157-165).  A block of 1152 samples has sixteen such partitions, so here a wave carries 64 / LPS subframes
// side by side, LPS = 4 / 8 / 16 / 32 lanes each ("segments" of the wave): every per-wave cost of the generic kernel on
// these sizes -- the lag trees, the serial Levinson recursion, the Rice level search, each the same number of wave
// instructions whatever the block size -- is paid once per 2 / 4 / 8 subframes.  The phases are those of the wave
// kernel with the wave-wide reductions cut at the segment:
//   phase 0  coalesced loads HBM -> LDS images, one all-zero segment in front of every image (halo of lane 0)
//   phase 1  window (lpc.rs:739-756) + autocorrelation (lpc.rs:533-548) in the canonical order of DESIGN.md 3.1:
//            16-sample fma chains, balanced tree over the chunk index padded to a power of two.  Lane sl of a
//            segment takes chunks 4 sl .. 4 sl + 3 (two in-lane tree levels), log2(LPS) DPP levels finish
//            node(0 .. 4 LPS); 72-sample lanes: the LPS / 2 chunks behind 64 LPS are the other half of the padded
//            tree, one chunk per lane of the lower half-segment, same lane tree.
//   phase 2  Levinson + quantisation (lpc.rs:633-705, 234-302) of ALL the workgroup's subframes on the first lanes
//            of wave 0 (one instruction stream for up to 32 systems)
//   phase 3  residual (lpc.rs:306-350), coefficients per lane (they differ between segments), 64-bit sums: the
//            reference's i32 path (lpc.rs:373-377) is taken only where it cannot overflow, i.e. where it agrees
//   phase 4  partitioned-Rice search (rice.rs:65-165, 246-298): bit-plane counts per lane, levels 0 .. log2(LPS)
//            inside the segment, level totals by a segment all-reduce
//   phase 5  residual rows straight from the registers (16 bytes per lane and store), one record per segment
// Rare subframes -- residuals of 2^25 (72-sample lanes; 2^26 for 64) and more, or a saturated table minimum
// (rice.rs:51) -- are marked (record status -1, QlpcKernelArgs::marked_count) and redone by the generic kernel's
// clean-up launch, as bigblock_residual_kernel does.
// Round 6 -- the unflagged order on these shapes emits the reference's integers by TWO PASSES: acorr_reference_mfma_kernel
// runs the reference's chains (lpc.rs:533-548) for every subframe in front, and this kernel takes that R[] (acorr_in) in
// place of its phase 1; what it marks is then marked -2 and redone from the same R[].  (An order certificate inside this
// kernel -- chunk-tree sums kept where a perturbation bound cleared them, the rest marked for a clean-up launch -- was built
// first: 6-13 % faster on noise-like material, 20 to 140 x slower on music at orders 10-12, where a third to two thirds of the
// subframes took the clean-up; profiles/r06_subwave_two_pass.txt.)
// STEREO: wave w of the workgroup is role w (L, R, M, S) of the workgroup's 64 / LPS frames, whose two channel
// images are shared in LDS -- the role stays wave-uniform and each channel is read from HBM once.
// Plain: two waves per workgroup, every segment an independent subframe with its own image.
//
// All `file:line` citations are relative to the flacenc-rs v0.5.1 tree.
#ifndef FLACENC_HIP_QLPC_SUBWAVE_KERNEL_IMPL_H_
#define FLACENC_HIP_QLPC_SUBWAVE_KERNEL_IMPL_H_

#include <type_traits>

#include "qlpc_wave_kernel_impl.h"

namespace flacenc_hip {
namespace {

template <int SPL, int LPS>
struct SubGeom {
  static constexpr int N = SPL * LPS;
  static constexpr int Seg = SPL + 4;             // a lane's samples + 4 dwords: conflict-free 16-byte reads (see WaveGeom)
  static constexpr int Img = (LPS + 1) * Seg + 4; // one all-zero segment in front
  static constexpr int S = 64 / LPS;              // subframes per wave
  static constexpr int LOGL = LPS == 4 ? 2 : (LPS == 8 ? 3 : (LPS == 16 ? 4 : 5));
  static constexpr int QuadsPerRow = N / 4;
  static constexpr int QuadsPerSeg = SPL / 4;
  // index inside an image of the 16-byte piece qq of a row
  static __device__ __forceinline__ int qidx(int qq) {
    const int sg = qq / QuadsPerSeg;
    return (sg + 1) * Seg + ((qq - sg * QuadsPerSeg) << 2);
  }
  // ... of sample t, -SPL <= t < N (t a multiple of 4 keeps the piece inside one segment)
  static __device__ __forceinline__ int idx(int t) {
    const int u = t + SPL;  // >= 0
    const int sg = SPL == 64 ? (u >> 6) : (int)(((uint32_t)u * 58255u) >> 22);  // u / 72, exact below 73727
    return sg * Seg + (u - sg * SPL);
  }
  // ... of sample SPL sl + off relative to the lane's segment base lb = (sl + 1) Seg, -SPL <= off < 2 SPL
  static __device__ __forceinline__ int rel(int lb, int off) {
    return lb + off + (off >= SPL ? (Seg - SPL) : 0) - (off < 0 ? (Seg - SPL) : 0);
  }
};

#define FLACENC_SUB_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, false))

// all-reduce over the LPS lanes of a segment (segments are aligned: 4 lanes = a quad, 8 = half a DPP row, 16 = a row, 32 = two):
// quad butterfly, half-row mirror, row mirror, lane ^ 16.  `op` must be commutative and associative (integers).
template <int LPS, class Op>
__device__ __forceinline__ uint32_t seg_allreduce(uint32_t v, Op op) {
  v = op(v, FLACENC_SUB_DPP(v, 0xB1));   // quad_perm [1, 0, 3, 2]
  v = op(v, FLACENC_SUB_DPP(v, 0x4E));   // quad_perm [2, 3, 0, 1]
  if (LPS >= 8) v = op(v, FLACENC_SUB_DPP(v, 0x141));  // row_half_mirror: lane i <- 7 - i of its half row
  if (LPS >= 16) v = op(v, FLACENC_SUB_DPP(v, 0x140));  // row_mirror
  if (LPS >= 32) v = op(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));  // lane ^ 16
  return v;
}
template <int LPS>
__device__ __forceinline__ uint32_t seg_sum(uint32_t v) {
  return seg_allreduce<LPS>(v, [](uint32_t x, uint32_t y) { return x + y; });
}
template <int LPS>
__device__ __forceinline__ uint32_t seg_or(uint32_t v) {
  return seg_allreduce<LPS>(v, [](uint32_t x, uint32_t y) { return x | y; });
}

// The canonical lane tree of an f64 inside a segment: level k adds lanes i and i ^ (1 << k) -- by row shifts the
// total of the segment arrives in its LAST lane exactly as the butterfly pairs it (wave_tree_sum_dpp, cut at LPS).
// Other lanes end up with partial sums nobody reads.
template <int LPS>
__device__ __forceinline__ double seg_tree_sum_last(double v) {
#define FLACENC_F64_DPP_STEP(CTRL, ROWMASK)                                                                        \
  {                                                                                                                \
    const unsigned long long b_ = (unsigned long long)__double_as_longlong(v);                                    \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b_, CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF);        \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b_ >> 32), CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF); \
    v = v + __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                              \
  }
  FLACENC_F64_DPP_STEP(0x111, 0xF)
  FLACENC_F64_DPP_STEP(0x112, 0xF)
  if (LPS >= 8) FLACENC_F64_DPP_STEP(0x114, 0xF)
  if (LPS >= 16) FLACENC_F64_DPP_STEP(0x118, 0xF)
  if (LPS >= 32) FLACENC_F64_DPP_STEP(0x142, 0xA)  // row_bcast15 into rows 1 and 3
#undef FLACENC_F64_DPP_STEP
  return v;
}

// VARIANT 0: the QLPC candidate of every subframe (records + residual rows); 1: fixed_lpc with the ApproxEnt order
// selector (coding.rs:298-331) as a batch of its own -- what the generic kernel's fixed_mode 1 produces; 2 (STEREO):
// encode_frame for 2-channel frames -- both candidates of the four roles, encode_subframe's choice (coding.rs:384-418),
// try_stereo_coding's assignment (:493-522), one flacenc_hip_stereo_frame_result and the TWO chosen rows per frame;
// 3 (plain): encode_frame for Independent(n) frames (coding.rs:537-541) -- both candidates and encode_subframe's choice
// per channel, one flacenc_hip_channel_result and the chosen row per subframe.
// S16 (plain, material declared at most 16 bits wide): the images hold int16 -- half the LDS, six workgroups per CU
// instead of three (a plain workgroup's 8 .. 32 images, not its registers, bound its occupancy).  A sample outside int16
// (the caller's declared width was wrong: flacenc_hip.h's precondition) marks its subframe for the generic kernel.
template <int MAXP, bool STEREO, int SPL, int LPS, int VARIANT, bool S16 = false>
__global__ void __launch_bounds__(STEREO ? 256 : 128, 3) qlpc_subwave_kernel(QlpcKernelArgs a) {
  static_assert(!S16 || !STEREO, "mid / side need 17 bits");
  static_assert(VARIANT != 2 || STEREO, "the frame decision is the 2-channel one");
  static_assert(VARIANT != 3 || !STEREO, "independent channels are plain subframes");
  using G = SubGeom<SPL, LPS>;
  constexpr int WAVES = STEREO ? 4 : 2;
  constexpr int THREADS = 64 * WAVES;
  constexpr int S = G::S;
  constexpr int SUBS = WAVES * S;           // subframes per workgroup
  constexpr int NIMG = STEREO ? 2 * S : SUBS;
  constexpr int HP = (MAXP + 3) & ~3;
  constexpr int NLAG = MAXP + 1;
  constexpr int n = G::N;
  constexpr bool LPC = VARIANT != 1;
  constexpr bool FIXED = VARIANT != 0;
  constexpr bool DECIDE = VARIANT >= 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  constexpr int IMGD = S16 ? G::Img / 2 : G::Img;  // dwords per image
  const int16_t* const sm16 = reinterpret_cast<const int16_t*>(smem_raw);
  float* const wlds = reinterpret_cast<float*>(sm + NIMG * IMGD);
  double* const xr = reinterpret_cast<double*>(sm + NIMG * IMGD + G::Img);  // [SUBS][NLAG]
  int32_t* const xq = reinterpret_cast<int32_t*>(xr + SUBS * NLAG);        // [SUBS][16]
  // DECIDE: per subframe slot {bits lo, bits hi, kind, dc, status, redo, 0, 0}
  uint32_t* const xd = reinterpret_cast<uint32_t*>(xq + SUBS * 16);        // [SUBS][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int sl = lane & (LPS - 1);   // lane inside the segment
  const int sg = lane / LPS;         // segment of the wave
  const int seg0 = lane & ~(LPS - 1);
  const int P = (int)a.lpc_order;
  const uint32_t blk = blockIdx.x;

  // ---- which subframe does this segment own ----
  const int g = STEREO ? (sg * 4 + wave) : (wave * S + sg);  // slot in the exchange areas (STEREO: frame-major, like sf)
  uint32_t sf, frame = 0;
  bool active;
  int img_a, img_b = 0;
  if (STEREO) {
    const uint32_t frames = a.n_subframes >> 2;
    frame = blk * (uint32_t)S + (uint32_t)sg;
    active = frame < frames;
    if (!active) frame = frames - 1u;
    sf = frame * 4u + (uint32_t)wave;
    img_a = (2 * sg + (wave == 1 ? 1 : 0)) * G::Img;
    img_b = (2 * sg + 1) * G::Img;
  } else {
    sf = blk * (uint32_t)SUBS + (uint32_t)g;
    active = sf < a.n_subframes;
    if (!active) sf = a.n_subframes - 1u;
    img_a = g * G::Img;
  }
  const int role = STEREO ? wave : 0;

  // ======================= phase 0: HBM -> LDS ==============================
  {
    constexpr int ZD = S16 ? G::Seg / 2 : G::Seg;  // dwords of an image's leading zero segment
    for (int i = tid; i < NIMG * ZD; i += THREADS) sm[(i / ZD) * IMGD + (i % ZD)] = 0;
    if (S16) {
      for (int i = tid; i < SUBS; i += THREADS) xd[i * 8 + 7] = 0u;  // "a sample did not fit int16"
      __syncthreads();
    }
  }
  if (LPC && a.acorr_in == nullptr) {
    const bool has_window = a.window != nullptr;
    const float* __restrict__ wsrc = a.window + 32;
    for (int i = tid; i < G::Seg; i += THREADS) wlds[i] = 0.0f;
    for (int q = tid; q < G::QuadsPerRow; q += THREADS) {
      const float4 w = has_window ? *reinterpret_cast<const float4*>(wsrc + (q << 2)) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
      *reinterpret_cast<float4*>(&wlds[G::qidx(q)]) = w;
    }
  }
  {
    constexpr int ROWS = NIMG;
    constexpr int NQ = ROWS * G::QuadsPerRow;
#pragma unroll
    for (int it = 0; it < (NQ + THREADS - 1) / THREADS; ++it) {
      const int q = tid + it * THREADS;
      if ((NQ % THREADS) == 0 || q < NQ) {
        const int row = q / G::QuadsPerRow;
        const int qq = q - row * G::QuadsPerRow;
        size_t src_row;
        if (STEREO) {
          const uint32_t frames = a.n_subframes >> 2;
          uint32_t f = blk * (uint32_t)S + (uint32_t)(row >> 1);
          if (f >= frames) f = frames - 1u;
          src_row = (size_t)(2u * f + (uint32_t)(row & 1));
        } else {
          uint32_t r = blk * (uint32_t)SUBS + (uint32_t)row;
          if (r >= a.n_subframes) r = a.n_subframes - 1u;
          src_row = r;
        }
        const int4 v = *reinterpret_cast<const int4*>(a.samples + src_row * a.stride + (qq << 2));
        if (S16) {
          const uint32_t over = ((uint32_t)(v.x + 32768) | (uint32_t)(v.y + 32768) | (uint32_t)(v.z + 32768) | (uint32_t)(v.w + 32768)) >> 16;
          if (over) atomicOr(&xd[row * 8 + 7], 1u);
          int2 pk;
          pk.x = (int)(((uint32_t)v.x & 0xFFFFu) | ((uint32_t)v.y << 16));
          pk.y = (int)(((uint32_t)v.z & 0xFFFFu) | ((uint32_t)v.w << 16));
          *reinterpret_cast<int2*>(&sm[row * IMGD + (G::qidx(qq) >> 1)]) = pk;
        } else {
          *reinterpret_cast<int4*>(&sm[row * G::Img + G::qidx(qq)]) = v;
        }
      }
    }
  }
  __syncthreads();

  const int32_t* const bufA = sm + img_a;
  const int32_t* const bufB = sm + img_b;
  auto ld4_at = [&](auto kind_tag, int ix) -> int4 {
    constexpr int KIND = decltype(kind_tag)::value;  // 0 = own image, 2 = mid, 3 = side
    if (S16) {
      const int2 pk = *reinterpret_cast<const int2*>(&sm16[img_a + ix]);
      return make_int4((pk.x << 16) >> 16, pk.x >> 16, (pk.y << 16) >> 16, pk.y >> 16);
    }
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    v4i_t va = *reinterpret_cast<const v4i_t*>(&bufA[ix]);
    asm("" : "+v"(va));
    int4 v = make_int4(va.x, va.y, va.z, va.w);
    if (KIND >= 2) {
      v4i_t vb = *reinterpret_cast<const v4i_t*>(&bufB[ix]);
      asm("" : "+v"(vb));
      const int4 r = make_int4(vb.x, vb.y, vb.z, vb.w);
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
  };
  auto with_role = [&](auto&& f) {
    if (STEREO && role == 2) f(std::integral_constant<int, 2>{});
    else if (STEREO && role == 3) f(std::integral_constant<int, 3>{});
    else f(std::integral_constant<int, 0>{});
  };
  const int lb = (sl + 1) * G::Seg;  // the lane's segment inside an image

  const unsigned long long bps_role = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + ((STEREO && role == 3) ? 1u : 0u));

  // ======================= phase 1: window + autocorrelation ==============
  if (LPC) {
    double R[NLAG];
    // (acorr_in: R[] comes from the launch in front -- the reference's chains by acorr_reference_mfma_kernel, or a flagged
    // order -- and the phase is a load; QlpcKernelArgs::acorr_in is a kernel argument: the branch is uniform)
    if (a.acorr_in != nullptr) {
      if (sl == LPS - 1) {
#pragma unroll
        for (int k = 0; k < NLAG; ++k) R[k] = a.acorr_in[(size_t)sf * 33 + k];
      }
    } else
    with_role([&](auto kind) {
      double dw[HP + 16];
      double acc[NLAG], s01[NLAG], p2[NLAG];
      // windowed samples [t, t + 4) of the subframe -> dw[at .. at + 4): x_w = (f32)s * w, one f32 rounding, then
      // widened (lpc.rs:751-754); t < 0 lands in the zero segment
      auto conv4 = [&](int t, int at) {
        const int ix = G::idx(t);
        const int4 v = ld4_at(kind, ix);
        const float4 w = *reinterpret_cast<const float4*>(&wlds[ix]);
        dw[at + 0] = (double)((float)v.x * w.x);
        dw[at + 1] = (double)((float)v.y * w.y);
        dw[at + 2] = (double)((float)v.z * w.z);
        dw[at + 3] = (double)((float)v.w * w.w);
      };
      // one 16-sample chunk starting at sample t0: fma chains from the literal +0.0.  MASKED: the chunk may hold
      // t < P, which contributes to no lag (common lower bound, lpc.rs:542) -- only a segment's first chunk can.
      // FRESH: the HP values in front of the chunk are converted here; otherwise they are the previous chunk's last.
      auto chunk = [&](int t0, auto masked_tag, auto fresh_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FRESH = decltype(fresh_tag)::value;
        if (FRESH) {
#pragma unroll
          for (int k = 0; k < HP; k += 4) conv4(t0 - HP + k, k);
        } else {
#pragma unroll
          for (int k = 0; k < HP; ++k) dw[k] = dw[k + 16];
        }
#pragma unroll
        for (int k = 0; k < 16; k += 4) conv4(t0 + k, HP + k);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          double cur = dw[HP + k];
          if (MASKED) cur = (t0 + k >= P) ? cur : 0.0;
#pragma unroll
          for (int tau = 0; tau <= MAXP; ++tau)
            acc[tau] = (k == 0) ? __builtin_fma(cur, dw[HP + k - tau], 0.0) : __builtin_fma(cur, dw[HP + k - tau], acc[tau]);
        }
      };
      const int t_first = 64 * sl;
      chunk(t_first, std::true_type{}, std::true_type{});
#pragma unroll
      for (int k = 0; k < NLAG; ++k) s01[k] = acc[k];
#pragma unroll 1
      for (int i = 1; i < 4; ++i) {
        chunk(t_first + 16 * i, std::false_type{}, std::false_type{});
        if (i == 1) {
#pragma unroll
          for (int k = 0; k < NLAG; ++k) s01[k] = s01[k] + acc[k];
        } else if (i == 2) {
#pragma unroll
          for (int k = 0; k < NLAG; ++k) p2[k] = acc[k];
        } else {
#pragma unroll
          for (int k = 0; k < NLAG; ++k) p2[k] = s01[k] + (p2[k] + acc[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < NLAG; ++k) R[k] = seg_tree_sum_last<LPS>(p2[k]);
      if (SPL != 64) {
        // chunks 4 LPS .. 4 LPS + LPS / 2: the other half of the padded tree (zeros behind them add nothing)
        chunk(64 * LPS + 16 * (sl & (LPS / 2 - 1)), std::false_type{}, std::true_type{});
#pragma unroll
        for (int k = 0; k < NLAG; ++k) R[k] = R[k] + seg_tree_sum_last<LPS>(sl < LPS / 2 ? acc[k] : 0.0);
      }
    });
    if (sl == LPS - 1) {
#pragma unroll
      for (int k = 0; k < NLAG; ++k) xr[g * NLAG + k] = R[k];
      if (a.autocorr && active && a.autocorr != a.acorr_in) {
#pragma unroll
        for (int k = 0; k < NLAG; ++k) a.autocorr[(size_t)sf * 33 + k] = k <= P ? R[k] : 0.0;
        for (int k = NLAG; k < 33; ++k) a.autocorr[(size_t)sf * 33 + k] = 0.0;
      }
    }
    __syncthreads();

    // ======================= phase 2: Levinson + quantisation ================
    if (wave == 0 && lane < SUBS) {
      double Rl[NLAG];
#pragma unroll
      for (int k = 0; k < NLAG; ++k) Rl[k] = xr[lane * NLAG + k];
      double coef[MAXP];
      int32_t cqv[MAXP];
      int warm_v, shift_v;
      const int st = levinson_quantize<MAXP>(Rl, P, (int)a.precision, coef, cqv, &warm_v, &shift_v);
#pragma unroll
      for (int i = 0; i < MAXP; ++i) xq[lane * 16 + i] = cqv[i];
      xq[lane * 16 + 12] = warm_v;
      xq[lane * 16 + 13] = shift_v;
      xq[lane * 16 + 14] = st;
      if (a.lpc_coefs) {
        // slot -> subframe (STEREO slots are frame-major: slot = 4 frame + role)
        uint32_t sfl = STEREO ? (blk * (uint32_t)S + (uint32_t)(lane >> 2)) * 4u + (uint32_t)(lane & 3)
                              : blk * (uint32_t)SUBS + (uint32_t)lane;
        if (sfl < a.n_subframes) {
#pragma unroll
          for (int i = 0; i < MAXP; ++i) a.lpc_coefs[(size_t)sfl * 32 + i] = (i < P && st == 0) ? coef[i] : 0.0;
          for (int i = MAXP; i < 32; ++i) a.lpc_coefs[(size_t)sfl * 32 + i] = 0.0;
        }
      }
    }
  }

  // ---- the Rice search of whatever e[] holds (phase 4) ----
  int32_t ebuf[SPL + 4];
  int32_t* const e = ebuf + 4;
  struct Coded {
    int bestk;
    uint32_t best_bits, my_p;
    unsigned long long sum_q, residual_bits;
    bool redo;
  };
  const bool finest_only = a.rice_finest_only != 0;
  auto rice_phase = [&](int warm) -> Coded {
    uint32_t pl[7];
    {
      uint32_t pb[5];
      popcount_planes16(e, pb);
#pragma unroll
      for (int k = 0; k < 5; ++k) pl[k] = pb[k];
      __builtin_amdgcn_sched_barrier(0);
      popcount_planes16(e + 16, pb);
      planes_add<5>(pl, pb);
      __builtin_amdgcn_sched_barrier(0);
      uint32_t pc[6], pd[5];
      popcount_planes16(e + 32, pd);
#pragma unroll
      for (int k = 0; k < 5; ++k) pc[k] = pd[k];
      __builtin_amdgcn_sched_barrier(0);
      popcount_planes16(e + 48, pd);
      planes_add<5>(pc, pd);
      planes_add<6>(pl, pc);
      __builtin_amdgcn_sched_barrier(0);
      if (SPL == 72) {
        int32_t t8[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t8[k] = k < 8 ? e[(SPL == 72 ? 64 : 0) + k] : 0;
        popcount_planes16(t8, pd);
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const uint32_t x_ = pl[k], y_ = k < 5 ? pd[k] : 0u;
          pl[k] = __builtin_amdgcn_bitop3_b32(x_, y_, carry, 0x96);
          carry = __builtin_amdgcn_bitop3_b32(x_, y_, carry, 0xE8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const uint32_t orw = seg_or<LPS>(pl[0] | pl[1] | pl[2] | pl[3] | pl[4] | pl[5] | pl[6]);
    const uint32_t maxu = (orw << 1) | (orw >> 31);
    const PlaneSums ps = make_plane_sums(pl);
    const uint32_t bitlen = maxu ? (uint32_t)(32 - __builtin_clz(maxu)) : 0u;
    const uint32_t max_p = a.max_rice_parameter < bitlen ? a.max_rice_parameter : bitlen;  // per segment
    const uint32_t len0 = (uint32_t)SPL - (sl == 0 ? (uint32_t)warm : 0u);
    Coded r;
    // the exact partition sums must fit 32 bits (64 codes below 2^26, 72 below 2^25): otherwise the generic kernel
    r.redo = maxu >= (1u << (SPL == 64 ? 26 : 25));

    // rice_window (see the wave kernel): per-lane bounds p0 of the partition means; the wave-wide minimum and maximum
    // bound every group of every segment, and a window wider than a segment's own only adds parameters that provably
    // lose (entries above the segment's max_p are set to the saturation value)
    const uint32_t s0 = 2u * ps.sum_m + ps.negs;
    const uint32_t q0 = (s0 >> 6) + 1u;
    const uint32_t q0lo = SPL == 64 ? q0 : (s0 >> 7) + 1u;
    const uint32_t p0min = wave_min_dpp(r.redo ? 31u : 31u - (uint32_t)__builtin_clz(q0lo));
    const uint32_t maxp_lo = wave_min_dpp(max_p), maxp_hi = wave_max_dpp(max_p);
    uint32_t p_lo = p0min > 2u ? p0min - 2u : 0u;
    p_lo = p_lo < maxp_lo ? p_lo : maxp_lo;
    const uint32_t q0hi = q0 + (sl == 0 ? (s0 >> 8) + 1u : 0u);
    const uint32_t p0max = wave_max_dpp(r.redo ? 0u : 31u - (uint32_t)__builtin_clz(q0hi));
    uint32_t p_hi = p0max + 1u;
    p_hi = p_hi < maxp_hi ? p_hi : maxp_hi;
    if (p_hi < p_lo) p_hi = p_lo;

    constexpr uint32_t kWMax = kMaxPToBits - 4u;
    uint32_t pk[G::LOGL + 1];
#pragma unroll
    for (int k = 0; k <= G::LOGL; ++k) pk[k] = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t p_base = p_lo; p_base <= p_hi; p_base += 4u) {
      uint32_t Wp[4];
      rice_build_tables<true, false, SPL>(ps, nullptr, len0, p_base, max_p, lane, warm, Wp);
#define FLACENC_SUB_RICE_LEVEL(K, SH)                                                         \
  if (K <= G::LOGL) {                                                                         \
    if (K > 0) {                                                                              \
      uint32_t part[4];                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) part[j] = from_upper_half<SH>(Wp[j]);     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
        const uint32_t v = Wp[j] + part[j];                                                   \
        Wp[j] = v < kWMax ? v : kWMax;                                                        \
      }                                                                                       \
    }                                                                                         \
    uint32_t packed = pk[K <= G::LOGL ? K : 0];                                               \
    _Pragma("unroll") for (int j = 0; j + 1 < 4; j += 2) {                                    \
      const uint32_t c0 = (Wp[j] << 5) | (p_base + (uint32_t)j);                              \
      const uint32_t c1 = (Wp[j + 1] << 5) | (p_base + (uint32_t)j + 1u);                     \
      packed = umin3(packed, c0, c1);                                                         \
    }                                                                                         \
    pk[K <= G::LOGL ? K : 0] = packed;                                                        \
  }
      FLACENC_SUB_RICE_LEVEL(0, 1)
      if (!finest_only) {
        FLACENC_SUB_RICE_LEVEL(1, 1)
        FLACENC_SUB_RICE_LEVEL(2, 2)
        FLACENC_SUB_RICE_LEVEL(3, 4)
        FLACENC_SUB_RICE_LEVEL(4, 8)
        FLACENC_SUB_RICE_LEVEL(5, 16)
      }
#undef FLACENC_SUB_RICE_LEVEL
    }
    // level totals inside the segment; strict < keeps the finer order on ties (rice.rs:285)
    r.bestk = 0;
    r.best_bits = 0;
    r.my_p = 0;
    uint32_t sat_any = 0;
#pragma unroll
    for (int K = 0; K <= G::LOGL; ++K) {
      if (K > 0 && finest_only) break;
      const uint32_t bits = (pk[K] >> 5) + 4u;
      const bool lead = (sl & ((1 << K) - 1)) == 0;
      sat_any |= (lead && bits >= kMaxPToBits) ? 1u : 0u;
      // (a segment's level total is at most 32 minima below 2^27)
      const uint32_t tot = seg_sum<LPS>(lead ? bits : 0u);
      if (K == 0 || tot < r.best_bits) {
        r.best_bits = tot;
        r.bestk = K;
        r.my_p = pk[K] & 31u;
      }
    }
    r.redo = r.redo || seg_or<LPS>(sat_any) != 0u;  // a saturated minimum: clamped entries may tie outside the window
    // Residual::sum_quotients / count_bits (datatype.rs:2325-2331, bitrepr.rs:533-544)
    const int rice_order = G::LOGL - r.bestk;
    const uint32_t best_parts = 1u << rice_order;
    const bool leader = (sl & ((1 << r.bestk) - 1)) == 0;
    const uint32_t sum_p = seg_sum<LPS>(leader ? r.my_p : 0u);
    const uint32_t p0 = (uint32_t)__shfl((int)r.my_p, seg0, 64);
    const uint32_t rice2 = seg_or<LPS>((leader && r.my_p > 14) ? 1u : 0u);
    const unsigned long long rem_bits = (unsigned long long)sum_p * (unsigned long long)(n >> rice_order) -
                                        (unsigned long long)warm * p0;
    r.sum_q = (unsigned long long)r.best_bits - 4ull * best_parts - (unsigned long long)(n - warm) - rem_bits;
    r.residual_bits = 2ull + 4ull + (unsigned long long)best_parts * (rice2 ? 5ull : 4ull) +
                      (r.sum_q + (unsigned long long)(n - warm)) + rem_bits;
    return r;
  };
  // one predictor record (all LPS lanes of the segment take part); `c4`: coefs[0..3] of a FixedLpc record
  auto write_record = [&](flacenc_hip_subframe_params* rec, const Coded& c, int status, int warm, int shift, uint32_t precision,
                          unsigned long long sub_bits, const int32_t* cq, auto ncq_tag) {
    constexpr int ncq = decltype(ncq_tag)::value;
    const int rice_order = G::LOGL - c.bestk;
    const uint32_t best_parts = 1u << rice_order;
    {
      // partition j of the chosen order lives on lane j << bestk of the segment
      const int srcl = seg0 | ((sl << c.bestk) & (LPS - 1));
      const uint32_t pv = (uint32_t)__shfl((int)c.my_p, srcl, 64);
      rec->rice_params[sl] = (uint8_t)((sl < (int)best_parts && status == 0) ? pv : 0u);
      uint32_t* words = reinterpret_cast<uint32_t*>(rec->rice_params);
      for (int w = LPS / 4 + sl; w < FLACENC_HIP_MAX_RICE_PARTITIONS / 4; w += LPS) words[w] = 0u;
    }
    if (sl == 0) {
      uint32_t* cw = reinterpret_cast<uint32_t*>(rec->coefs);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c0 = (2 * i < ncq && status == 0) ? cq[2 * i < ncq ? 2 * i : 0] : 0;
        const int c1 = (2 * i + 1 < ncq && status == 0) ? cq[2 * i + 1 < ncq ? 2 * i + 1 : 0] : 0;
        cw[i] = ((uint32_t)c0 & 0xFFFFu) | ((uint32_t)c1 << 16);
      }
      rec->order = (uint8_t)warm;
      rec->shift = (int8_t)shift;
      rec->precision = (uint8_t)precision;
      rec->rice_order = (uint8_t)(status == 0 ? rice_order : 0);
      rec->status = status;
      rec->code_bits = status == 0 ? (unsigned long long)c.best_bits : 0ull;
      rec->subframe_bits = status == 0 ? sub_bits : 0ull;
      rec->sum_quotients = status == 0 ? c.sum_q : 0ull;
    }
  };
  auto store_row = [&](int32_t* __restrict__ row) {
    int32_t* __restrict__ dst = row + sl * SPL;
#pragma unroll
    for (int k = 0; k < SPL; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(e[k], e[k + 1], e[k + 2], e[k + 3]);
  };

  // ======================= fixed_lpc: order selection + coding =============
  int fx_order = 0;
  unsigned long long fx_key = ~0ull, fx_sub_bits = 0;
  Coded fx{};
  int role_min = 0, role_max = 0;
  // the lane's samples + the 4 in front of them, differenced `ord` times in place (reset_fixed_lpc_errors,
  // coding.rs:182-197: zero history in front of the block); valid from index ord on
  auto fixed_error_signal = [&](int ord) {
    with_role([&](auto kind) {
#pragma unroll
      for (int k = 0; k < SPL / 4 + 1; ++k) {
        const int4 q = ld4_at(kind, G::rel(lb, -4 + 4 * k));
        ebuf[4 * k + 0] = q.x;
        ebuf[4 * k + 1] = q.y;
        ebuf[4 * k + 2] = q.z;
        ebuf[4 * k + 3] = q.w;
      }
    });
#pragma unroll 1
    for (int lvl = 1; lvl <= 4; ++lvl) {
      if (lvl <= ord) {  // (per segment: lanes of lower orders sit out the later passes)
#pragma unroll
        for (int i = SPL + 3; i >= 1; --i) ebuf[i] = (int32_t)((uint32_t)ebuf[i] - (uint32_t)ebuf[i - 1]);
      }
    }
    // the first `order` errors are never coded (Residual keeps zeros there, coding.rs:151-160)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sl == 0 && k < ord) e[k] = 0;
  };
  if (FIXED && (!DECIDE || a.use_fixed)) {
    // ---- estimate_entropy (coding.rs:200-227) for orders 0..max_order: exact integer sums of |e_k| per estimator
    // partition -- v_sad_u32 on biased values gives the next order's magnitudes from this order's values, sub-sums of
    // SPL / 8 samples stay below 2^32 for inputs of up to 25 bits -- then one f32 evaluation per (order, partition)
    constexpr int GS = SPL / 8;
    const int psz = n / (int)a.fixed_partitions;            // launch_qlpc: a whole number of sub-sums, see subwave_fixed_ok
    const int qpl = psz >= SPL ? 1 : SPL / psz;             // partitions per lane: 1, 2 or 4
    const int glog = psz > SPL ? (31 - __builtin_clz(psz / SPL)) : 0;  // ... or 2^glog lanes per partition
    uint32_t tot[5] = {0u, 0u, 0u, 0u, 0u};
    {
      uint32_t b[SPL + 4];
      uint32_t bias0 = 0x80000000u;
      uint32_t bmin = 0xFFFFFFFFu, bmax = 0u;
      with_role([&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
        typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
        bias0 = KIND == 2 ? 0x40000000u : KIND == 3 ? 0x7FFFFFFFu : 0x80000000u;
#pragma unroll
        for (int k = 0; k < SPL / 4 + 1; ++k) {
          const int ix = G::rel(lb, -4 + 4 * k);
          v4u_t va;
          if (S16) {
            const int4 q = ld4_at(kind, ix);
            va = v4u_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w};
          } else {
            va = *reinterpret_cast<const v4u_t*>(&bufA[ix]);
            asm("" : "+v"(va));
          }
          if (KIND >= 2) {
            v4u_t vb = *reinterpret_cast<const v4u_t*>(&bufB[ix]);
            asm("" : "+v"(vb));
#pragma unroll
            for (int q = 0; q < 4; ++q)
              b[4 * k + q] = KIND == 2 ? (va[q] + vb[q] + 0x80000000u) >> 1 : xad_u32(vb[q], 0x7FFFFFFFu, va[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) b[4 * k + q] = va[q] ^ 0x80000000u;
          }
        }
      });
      // (a segment's first lane sees the zero segment in front of it: the biased zero, as the differences need it)
      if (DECIDE) {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
          bmin = b[4 + j] < bmin ? b[4 + j] : bmin;
          bmax = b[4 + j] > bmax ? b[4 + j] : bmax;
        }
        bmin = seg_allreduce<LPS>(bmin, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
        bmax = seg_allreduce<LPS>(bmax, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
        role_min = (int)(bmin - bias0);
        role_max = (int)(bmax - bias0);
      }
#pragma unroll
      for (int ord = 0; ord < 5; ++ord) {
        uint32_t c[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        if (ord == 0) {
#pragma unroll
          for (int j = 0; j < SPL; ++j) c[j / GS] = sad_u32(b[4 + j], bias0, c[j / GS]);
        } else {
#pragma unroll
          for (int j = 0; j < SPL; ++j) c[j / GS] = sad_u32(b[4 + j], b[3 + j], c[j / GS]);
          if (ord < 4) {
#pragma unroll
            for (int i = SPL + 3; i >= ord; --i) b[i] = xad_u32(b[i - 1], 0x7FFFFFFFu, b[i]);
          }
        }
        // partition sums (exact integers below 2^53: the order of these adds is immaterial)
        double ls[4];
        if (qpl == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) ls[q] = (double)c[2 * q] + (double)c[2 * q + 1];
        } else if (qpl == 2) {
          ls[0] = ((double)c[0] + (double)c[1]) + ((double)c[2] + (double)c[3]);
          ls[1] = ((double)c[4] + (double)c[5]) + ((double)c[6] + (double)c[7]);
          ls[2] = ls[3] = 0.0;
        } else {
          ls[0] = (((double)c[0] + (double)c[1]) + ((double)c[2] + (double)c[3])) +
                  (((double)c[4] + (double)c[5]) + ((double)c[6] + (double)c[7]));
          ls[1] = ls[2] = ls[3] = 0.0;
#pragma unroll 1
          for (int lvl = 0; lvl < glog; ++lvl) ls[0] += __shfl_xor(ls[0], 1 << lvl, 64);
        }
        if (ord <= (int)a.fixed_max_order) {
          uint32_t pb = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q < qpl) {
              // sample_count = min(end - warmup, partition_len): only the block's first partition loses the warm-up
              const uint32_t cnt = (uint32_t)psz - ((sl >> glog) == 0 && q == 0 ? (uint32_t)ord : 0u);
              pb += approx_ent_bits(ls[q], cnt);
            }
          }
          tot[ord] = seg_sum<LPS>((sl & ((1 << glog) - 1)) == 0 ? pb : 0u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    fx_key = ~0ull;
#pragma unroll
    for (int ord = 0; ord < 5; ++ord) {
      if (ord <= (int)a.fixed_max_order) {
        const unsigned long long kk = (unsigned long long)tot[ord] + bps_role * (unsigned long long)ord;
        if (a.fixed_keys && active && sl == 0) a.fixed_keys[(size_t)sf * 8 + ord] = kk;
        if (kk < fx_key) {  // min_by_key keeps the first minimum
          fx_key = kk;
          fx_order = ord;
        }
      }
    }
    if (!DECIDE && a.selector_keys && active && sl == 0) a.selector_keys[sf] = fx_key;
    // fixed_lpc returns None when the estimate does not beat verbatim_bits (coding.rs:284): then nothing is coded
    // (the stand-alone batch codes the order regardless, as the generic kernel's fixed_mode 1 does)
    const bool have = !DECIDE || fx_key < 8ull + (unsigned long long)n * bps_role;
    if (__builtin_amdgcn_ballot_w64(have) != 0ull) {
      fixed_error_signal(fx_order);
      fx = rice_phase(fx_order);
      fx_sub_bits = 8ull + bps_role * (unsigned long long)fx_order + fx.residual_bits;  // bitrepr.rs:473-477
    }
  } else if (DECIDE) {
    // no fixed candidate: the role's min / max alone (is_constant, arrayutils.rs:382)
    int vmin = INT32_MAX, vmax = INT32_MIN;
    with_role([&](auto kind) {
#pragma unroll
      for (int k = 0; k < SPL / 4; ++k) {
        const int4 v = ld4_at(kind, lb + 4 * k);
        vmax = max(max(vmax, v.x), max(v.y, max(v.z, v.w)));
        vmin = min(min(vmin, v.x), min(v.y, min(v.z, v.w)));
      }
    });
    role_min = (int)(seg_allreduce<LPS>((uint32_t)vmin ^ 0x80000000u, [](uint32_t x, uint32_t y) { return x < y ? x : y; }) ^ 0x80000000u);
    role_max = (int)(seg_allreduce<LPS>((uint32_t)vmax ^ 0x80000000u, [](uint32_t x, uint32_t y) { return x > y ? x : y; }) ^ 0x80000000u);
  }
  if (VARIANT == 1) {
    // ---- the stand-alone fixed_lpc batch: row + record, as the generic kernel's fixed_mode 1 ----
    if (!active) return;
    store_row(a.residual + (size_t)sf * a.residual_stride);
    if (a.params == nullptr) return;
    flacenc_hip_subframe_params* rec = a.params + sf;
    if (fx.redo || (S16 && xd[g * 8 + 7] != 0u)) {
      if (sl == 0) {
        rec->status = -1;
        count_marked(a, sf);
      }
      return;
    }
    int32_t c4[4];
    c4[0] = fx_order;  // FIXED_LPC_COEFS[order]: 0 / 1 / 2,-1 / 3,-3,1 / 4,-6,4,-1 (decode.rs:179-185)
    c4[1] = fx_order == 2 ? -1 : (fx_order == 3 ? -3 : (fx_order == 4 ? -6 : 0));
    c4[2] = fx_order == 3 ? 1 : (fx_order == 4 ? 4 : 0);
    c4[3] = fx_order == 4 ? -1 : 0;
    write_record(rec, fx, 0, fx_order, 0, 0u, fx_sub_bits, c4, std::integral_constant<int, 4>{});
    return;
  }

  // ======================= phase 3: the QLPC residual -> registers =========
  __syncthreads();  // wave 0's recursions are done
  int32_t cq[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) cq[i] = xq[g * 16 + i];
  const int warm = xq[g * 16 + 12];
  const int shift = xq[g * 16 + 13];
  const int status = xq[g * 16 + 14];
  with_role([&](auto kind) {
    int sw[HP + 16];
    constexpr int NCH = (SPL + 15) / 16;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int t0 = 16 * i;
      const int cn = (SPL - 16 * i) < 16 ? (SPL - 16 * i) : 16;
      asm volatile("" ::: "memory");
      if (i > 0) {
#pragma unroll
        for (int k = 0; k < HP; ++k) sw[k] = sw[k + 16];
      }
      const int first = (i == 0) ? 0 : HP;
#pragma unroll
      for (int k = first; k < HP + 16; k += 4) {
        if (k >= HP + cn) continue;
        const int4 v = ld4_at(kind, G::rel(lb, t0 - HP + k));
        sw[k + 0] = v.x;
        sw[k + 1] = v.y;
        sw[k + 2] = v.z;
        sw[k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k >= cn) continue;
        int64_t pred = 0;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) pred += (int64_t)cq[j] * (int64_t)sw[HP + k - 1 - j];
        e[16 * i + k] = (int32_t)(uint32_t)(uint64_t)((int64_t)sw[HP + k] - (pred >> shift));
      }
    }
  });
  // e[0 .. order') = 0 (lpc.rs:349): the segment's first lane; a failed analysis leaves an all-zero row
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if ((sl == 0 && k < warm) || status != 0) e[k] = 0;
  if (status != 0) {
#pragma unroll
    for (int k = 16; k < SPL; ++k) e[k] = 0;
  }
  if (!DECIDE && active) store_row(a.residual + (size_t)sf * a.residual_stride);

  const Coded lp = rice_phase(warm);
  const unsigned long long sub_bits = 8ull + bps_role * (unsigned long long)warm + 4ull + 5ull +
                                      (unsigned long long)a.precision * (unsigned long long)warm + lp.residual_bits;
  if (!DECIDE) {
    // ======================= phase 5: the record ============================
    if (!active || a.params == nullptr) return;
    flacenc_hip_subframe_params* rec = a.params + sf;
    if ((lp.redo && status == 0) || (S16 && xd[g * 8 + 7] != 0u)) {
      if (sl == 0) {
        rec->status = a.acorr_in != nullptr ? -2 : -1;  // (-2: redone from the R[] this launch was given)
        count_marked(a, sf);
      }
      return;
    }
    write_record(rec, lp, status, warm, shift, a.precision, sub_bits, cq, std::integral_constant<int, MAXP>{});
    return;
  }

  // ======================= encode_subframe + try_stereo_coding =============
  if (DECIDE) {
    const unsigned long long verbatim_bits = 8ull + (unsigned long long)n * bps_role;  // datatype.rs:1944
    const bool have_fixed = a.use_fixed && fx_key < verbatim_bits;  // coding.rs:262, :284
    const unsigned long long fixed_bits = have_fixed ? fx_sub_bits : ~0ull;
    const unsigned long long baseline = fixed_bits < verbatim_bits ? fixed_bits : verbatim_bits;  // coding.rs:403-405
    uint32_t kind;
    unsigned long long bits;
    if (a.use_constant && role_min == role_max) {
      kind = FLACENC_HIP_KIND_CONSTANT;
      bits = 8ull + bps_role;  // bitrepr.rs:445
    } else if (a.use_lpc && status == 0 && sub_bits < baseline) {
      kind = FLACENC_HIP_KIND_LPC;
      bits = sub_bits;
    } else if (have_fixed && fixed_bits < verbatim_bits) {
      kind = FLACENC_HIP_KIND_FIXED;
      bits = fixed_bits;
    } else {
      kind = FLACENC_HIP_KIND_VERBATIM;
      bits = verbatim_bits;
    }
    // a candidate the exact sums could not carry: the whole frame goes to the general path (launch_qlpc)
    const bool redo = (a.use_lpc && status == 0 && lp.redo) || (have_fixed && fx.redo) || (S16 && xd[g * 8 + 7] != 0u);
    if (VARIANT == 3) {
      // ---- Independent(n) frames: the segment's subframe is one output channel ----
      if (!active) return;
      flacenc_hip_channel_result* out = a.chan_results + sf;
      if (redo) {
        if (sl == 0) {
          if (a.cand_lpc_params) const_cast<flacenc_hip_subframe_params*>(a.cand_lpc_params)[sf].status = a.acorr_in != nullptr ? -2 : -1;
          if (a.cand_fixed_params) const_cast<flacenc_hip_subframe_params*>(a.cand_fixed_params)[sf].status = -1;
          out->kind = 0xFF;
          count_marked(a, sf);
        }
        return;
      }
      if (sl == 0) {
        out->kind = (uint8_t)kind;
        out->analysis_status = (uint8_t)(a.use_lpc ? status : 0);
        out->pad[0] = out->pad[1] = 0;
        out->dc_offset = kind == FLACENC_HIP_KIND_CONSTANT ? role_min : 0;
        out->bits = bits;
      }
      int32_t* row = a.residual + (size_t)sf * a.residual_stride;
      flacenc_hip_subframe_params* rec = &out->params;
      if (kind == FLACENC_HIP_KIND_LPC) {
        store_row(row);
        write_record(rec, lp, 0, warm, shift, a.precision, sub_bits, cq, std::integral_constant<int, MAXP>{});
      } else if (kind == FLACENC_HIP_KIND_FIXED) {
        fixed_error_signal(fx_order);
        store_row(row);
        int32_t c4[4];
        c4[0] = fx_order;
        c4[1] = fx_order == 2 ? -1 : (fx_order == 3 ? -3 : (fx_order == 4 ? -6 : 0));
        c4[2] = fx_order == 3 ? 1 : (fx_order == 4 ? 4 : 0);
        c4[3] = fx_order == 4 ? -1 : 0;
        write_record(rec, fx, 0, fx_order, 0, 0u, fx_sub_bits, c4, std::integral_constant<int, 4>{});
      } else {
#pragma unroll
        for (int k = 0; k < SPL; ++k) e[k] = 0;
        store_row(row);
        uint32_t* w = reinterpret_cast<uint32_t*>(rec);
        for (int i = sl; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += LPS) w[i] = 0u;
      }
      return;
    }
    if (sl == 0) {
      xd[g * 8 + 0] = (uint32_t)bits;
      xd[g * 8 + 1] = (uint32_t)(bits >> 32);
      xd[g * 8 + 2] = kind;
      xd[g * 8 + 3] = (uint32_t)role_min;
      xd[g * 8 + 4] = a.use_lpc ? (uint32_t)status : 0u;
      xd[g * 8 + 5] = redo ? 1u : 0u;
    }
    __syncthreads();
    unsigned long long rb[4];
    uint32_t any_redo = 0, any_status = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t* x = xd + (sg * 4 + r) * 8;
      rb[r] = (unsigned long long)x[0] | ((unsigned long long)x[1] << 32);
      any_redo |= x[5];
      any_status |= x[4];
    }
    unsigned long long min_bits = rb[0] + rb[1];
    uint32_t assignment = 0;  // Independent(2)
    if (a.use_leftside && rb[0] + rb[3] < min_bits) {
      min_bits = rb[0] + rb[3];
      assignment = 1;
    }
    if (a.use_rightside && rb[1] + rb[3] < min_bits) {
      min_bits = rb[1] + rb[3];
      assignment = 2;
    }
    if (a.use_midside && rb[2] + rb[3] < min_bits) {
      min_bits = rb[2] + rb[3];
      assignment = 3;
    }
    // ChannelAssignment::select_channels, datatype.rs:1173-1185
    const int role0 = assignment == 2 ? 3 : (assignment == 3 ? 2 : 0);
    const int role1 = (assignment == 0 || assignment == 2) ? 1 : 3;
    if (!active) return;
    flacenc_hip_stereo_frame_result* fr = a.frame_results + frame;
    if (any_redo) {
      // marked for the general path: its candidate batches redo the frame's four roles, frame_decide_kernel the frame
      if (sl == 0) {
        if (a.cand_lpc_params) const_cast<flacenc_hip_subframe_params*>(a.cand_lpc_params)[sf].status = a.acorr_in != nullptr ? -2 : -1;
        if (a.cand_fixed_params) const_cast<flacenc_hip_subframe_params*>(a.cand_fixed_params)[sf].status = -1;
        if (role == 0) {
          fr->channel_assignment = 0xFF;
          count_marked(a, frame);  // (marked_unit 4: the frame's four roles)
        }
      }
      return;
    }
    if (role == 0 && sl == 0) {
      fr->channel_assignment = (uint8_t)assignment;
      fr->analysis_status = (uint8_t)any_status;
      fr->pad[0] = fr->pad[1] = 0;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const uint32_t* x = xd + (sg * 4 + (ch == 0 ? role0 : role1)) * 8;
        fr->role[ch] = (uint8_t)(ch == 0 ? role0 : role1);
        fr->kind[ch] = (uint8_t)x[2];
        fr->dc_offset[ch] = x[2] == FLACENC_HIP_KIND_CONSTANT ? (int32_t)x[3] : 0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) fr->bits[r] = rb[r];
    }
    if (role != role0 && role != role1) return;
    const int ch = role == role0 ? 0 : 1;
    flacenc_hip_subframe_params* rec = &fr->lpc[ch];
    int32_t* row = a.residual + (size_t)(2u * frame + (uint32_t)ch) * a.residual_stride;
    if (kind == FLACENC_HIP_KIND_LPC) {
      store_row(row);
      write_record(rec, lp, 0, warm, shift, a.precision, sub_bits, cq, std::integral_constant<int, MAXP>{});
    } else if (kind == FLACENC_HIP_KIND_FIXED) {
      fixed_error_signal(fx_order);
      store_row(row);
      int32_t c4[4];
      c4[0] = fx_order;
      c4[1] = fx_order == 2 ? -1 : (fx_order == 3 ? -3 : (fx_order == 4 ? -6 : 0));
      c4[2] = fx_order == 3 ? 1 : (fx_order == 4 ? 4 : 0);
      c4[3] = fx_order == 4 ? -1 : 0;
      write_record(rec, fx, 0, fx_order, 0, 0u, fx_sub_bits, c4, std::integral_constant<int, 4>{});
    } else {
      // Constant / Verbatim: an all-zero row and a blank record
#pragma unroll
      for (int k = 0; k < SPL; ++k) e[k] = 0;
      store_row(row);
      uint32_t* w = reinterpret_cast<uint32_t*>(rec);
      for (int i = sl; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += LPS) w[i] = 0u;
    }
  }
}

template <int MAXP, bool STEREO, int SPL, int LPS, int VARIANT, bool S16 = false>
hipError_t launch_subwave_geom(const QlpcKernelArgs& a, hipStream_t stream) {
  using G = SubGeom<SPL, LPS>;
  constexpr int WAVES = STEREO ? 4 : 2;
  constexpr int SUBS = WAVES * G::S;
  constexpr int NIMG = STEREO ? 2 * G::S : SUBS;
  constexpr size_t smem = ((size_t)NIMG * (S16 ? G::Img / 2 : G::Img) + G::Img) * 4 + (size_t)SUBS * ((MAXP + 1) * 8 + 64 + 32);
  auto kern = qlpc_subwave_kernel<MAXP, STEREO, SPL, LPS, VARIANT, S16>;
  static DynamicLdsOptIn opt_in;
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  const uint32_t blocks = STEREO ? ((a.n_subframes >> 2) + (uint32_t)G::S - 1u) / (uint32_t)G::S
                                 : (a.n_subframes + (uint32_t)SUBS - 1u) / (uint32_t)SUBS;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), smem, stream, a);
  return hipGetLastError();
}

template <int MAXP, bool STEREO, int SPL, int VARIANT>
hipError_t launch_subwave(const QlpcKernelArgs& a, hipStream_t stream) {
  const uint32_t lps = a.block_size / (uint32_t)SPL;
  if (a.block_size != lps * (uint32_t)SPL) return hipErrorInvalidValue;
  if constexpr (!STEREO) {
    if (a.bps == nullptr && a.bps_uniform <= 16u) {  // int16 images
      if (lps == 4) return launch_subwave_geom<MAXP, STEREO, SPL, 4, VARIANT, true>(a, stream);
      if (lps == 8) return launch_subwave_geom<MAXP, STEREO, SPL, 8, VARIANT, true>(a, stream);
      if (lps == 16) return launch_subwave_geom<MAXP, STEREO, SPL, 16, VARIANT, true>(a, stream);
      if (lps == 32) return launch_subwave_geom<MAXP, STEREO, SPL, 32, VARIANT, true>(a, stream);
      return hipErrorInvalidValue;
    }
  }
  if (lps == 4) return launch_subwave_geom<MAXP, STEREO, SPL, 4, VARIANT>(a, stream);
  if (lps == 8) return launch_subwave_geom<MAXP, STEREO, SPL, 8, VARIANT>(a, stream);
  if (lps == 16) return launch_subwave_geom<MAXP, STEREO, SPL, 16, VARIANT>(a, stream);
  if (lps == 32) return launch_subwave_geom<MAXP, STEREO, SPL, 32, VARIANT>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace
}  // namespace flacenc_hip
#endif
