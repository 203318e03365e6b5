// qlpc_subwave_kernel_impl.h -- the fused QLPC kernel for blocks smaller than one wave's worth of finest Rice
// partitions: 512 / 1024 / 2048 (64 samples per lane) and 576 / 1152 / 2304 (72 per lane, the CD-style sizes).
//
// qlpc_wave_kernel_impl.h gives a subframe of 4096 (4608) samples one wave: lane l holds one finest Rice partition
// (rice.rs:157-165).  A block of 1152 samples has sixteen such partitions, so here a wave carries 64 / LPS subframes
// side by side, LPS = 8 / 16 / 32 lanes each ("segments" of the wave): every per-wave cost of the generic kernel on
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
  static constexpr int LOGL = LPS == 8 ? 3 : (LPS == 16 ? 4 : 5);
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

// all-reduce over the LPS lanes of a segment (segments are aligned: 8 lanes = half a DPP row, 16 = a row, 32 = two):
// quad butterfly, half-row mirror, row mirror, lane ^ 16.  `op` must be commutative and associative (integers).
template <int LPS, class Op>
__device__ __forceinline__ uint32_t seg_allreduce(uint32_t v, Op op) {
  v = op(v, FLACENC_SUB_DPP(v, 0xB1));   // quad_perm [1, 0, 3, 2]
  v = op(v, FLACENC_SUB_DPP(v, 0x4E));   // quad_perm [2, 3, 0, 1]
  v = op(v, FLACENC_SUB_DPP(v, 0x141));  // row_half_mirror: lane i <- 7 - i of its half row
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
  FLACENC_F64_DPP_STEP(0x114, 0xF)
  if (LPS >= 16) FLACENC_F64_DPP_STEP(0x118, 0xF)
  if (LPS >= 32) FLACENC_F64_DPP_STEP(0x142, 0xA)  // row_bcast15 into rows 1 and 3
#undef FLACENC_F64_DPP_STEP
  return v;
}

template <int MAXP, bool STEREO, int SPL, int LPS>
__global__ void __launch_bounds__(STEREO ? 256 : 128, 3) qlpc_subwave_kernel(QlpcKernelArgs a) {
  using G = SubGeom<SPL, LPS>;
  constexpr int WAVES = STEREO ? 4 : 2;
  constexpr int THREADS = 64 * WAVES;
  constexpr int S = G::S;
  constexpr int SUBS = WAVES * S;           // subframes per workgroup
  constexpr int NIMG = STEREO ? 2 * S : SUBS;
  constexpr int HP = (MAXP + 3) & ~3;
  constexpr int NLAG = MAXP + 1;
  constexpr int n = G::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  float* const wlds = reinterpret_cast<float*>(sm + NIMG * G::Img);
  double* const xr = reinterpret_cast<double*>(sm + (NIMG + 1) * G::Img);  // [SUBS][NLAG]
  int32_t* const xq = reinterpret_cast<int32_t*>(xr + SUBS * NLAG);        // [SUBS][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int sl = lane & (LPS - 1);   // lane inside the segment
  const int sg = lane / LPS;         // segment of the wave
  const int P = (int)a.lpc_order;
  const uint32_t blk = blockIdx.x;

  // ---- which subframe does this segment own ----
  const int g = STEREO ? (sg * 4 + wave) : (wave * S + sg);  // slot in the exchange areas (STEREO: frame-major, like sf)
  uint32_t sf;
  bool active;
  int img_a, img_b = 0;
  if (STEREO) {
    const uint32_t frames = a.n_subframes >> 2;
    uint32_t f = blk * (uint32_t)S + (uint32_t)sg;
    active = f < frames;
    if (!active) f = frames - 1u;
    sf = f * 4u + (uint32_t)wave;
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
  for (int i = tid; i < NIMG * G::Seg; i += THREADS) sm[(i / G::Seg) * G::Img + (i % G::Seg)] = 0;
  {
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
        *reinterpret_cast<int4*>(&sm[row * G::Img + G::qidx(qq)]) = v;
      }
    }
  }
  __syncthreads();

  const int32_t* const bufA = sm + img_a;
  const int32_t* const bufB = sm + img_b;
  auto ld4_at = [&](auto kind_tag, int ix) -> int4 {
    constexpr int KIND = decltype(kind_tag)::value;  // 0 = own image, 2 = mid, 3 = side
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
  {
    double R[NLAG];
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
      if (a.autocorr && active) {
#pragma unroll
        for (int k = 0; k < NLAG; ++k) a.autocorr[(size_t)sf * 33 + k] = k <= P ? R[k] : 0.0;
        for (int k = NLAG; k < 33; ++k) a.autocorr[(size_t)sf * 33 + k] = 0.0;
      }
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
  __syncthreads();
  int32_t cq[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) cq[i] = xq[g * 16 + i];
  const int warm = xq[g * 16 + 12];
  const int shift = xq[g * 16 + 13];
  int status = xq[g * 16 + 14];

  // ======================= phase 3: residual -> registers ==================
  int32_t e[SPL];
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

  // ======================= residual store: registers -> HBM ================
  if (active) {
    int32_t* __restrict__ dst = a.residual + (size_t)sf * a.residual_stride + sl * SPL;
#pragma unroll
    for (int k = 0; k < SPL; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(e[k], e[k + 1], e[k + 2], e[k + 3]);
  }

  // ======================= phase 4: partitioned-Rice search ================
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
  const bool finest_only = a.rice_finest_only != 0;
  // the exact partition sums must fit 32 bits (64 codes below 2^26, 72 below 2^25): otherwise the clean-up launch
  bool redo = maxu >= (1u << (SPL == 64 ? 26 : 25));

  // rice_window (see the wave kernel): per-lane bounds p0 of the partition means; the wave-wide minimum and maximum
  // bound every group of every segment, and a window wider than a segment's own only adds parameters that provably
  // lose (entries above the segment's max_p are set to the saturation value)
  const uint32_t s0 = 2u * ps.sum_m + ps.negs;
  const uint32_t q0 = (s0 >> 6) + 1u;
  const uint32_t q0lo = SPL == 64 ? q0 : (s0 >> 7) + 1u;
  const uint32_t p0min = wave_min_dpp(redo ? 31u : 31u - (uint32_t)__builtin_clz(q0lo));
  const uint32_t maxp_lo = wave_min_dpp(max_p), maxp_hi = wave_max_dpp(max_p);
  uint32_t p_lo = p0min > 2u ? p0min - 2u : 0u;
  p_lo = p_lo < maxp_lo ? p_lo : maxp_lo;
  const uint32_t q0hi = q0 + (sl == 0 ? (s0 >> 8) + 1u : 0u);
  const uint32_t p0max = wave_max_dpp(redo ? 0u : 31u - (uint32_t)__builtin_clz(q0hi));
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
  int bestk = 0;
  uint32_t best_bits = 0, my_p = 0, sat_any = 0;
#pragma unroll
  for (int K = 0; K <= G::LOGL; ++K) {
    if (K > 0 && finest_only) break;
    const uint32_t bits = (pk[K] >> 5) + 4u;
    const bool lead = (sl & ((1 << K) - 1)) == 0;
    sat_any |= (lead && bits >= kMaxPToBits) ? 1u : 0u;
    // (a segment's level total is at most 32 minima below 2^27)
    const uint32_t tot = seg_sum<LPS>(lead ? bits : 0u);
    if (K == 0 || tot < best_bits) {
      best_bits = tot;
      bestk = K;
      my_p = pk[K] & 31u;
    }
  }
  redo = redo || seg_or<LPS>(sat_any) != 0u;  // a saturated minimum at any level: clamped entries may tie outside the window

  const int rice_order = G::LOGL - bestk;
  const uint32_t best_parts = 1u << rice_order;
  // Residual::sum_quotients / count_bits (datatype.rs:2325-2331, bitrepr.rs:533-544)
  const bool leader = (sl & ((1 << bestk) - 1)) == 0;
  const uint32_t sum_p = seg_sum<LPS>(leader ? my_p : 0u);
  const uint32_t p0 = (uint32_t)__shfl((int)my_p, lane & ~(LPS - 1), 64);
  const uint32_t rice2 = seg_or<LPS>((leader && my_p > 14) ? 1u : 0u);
  const unsigned long long rem_bits = (unsigned long long)sum_p * (unsigned long long)(n >> rice_order) -
                                      (unsigned long long)warm * p0;
  const unsigned long long sum_q = (unsigned long long)best_bits - 4ull * best_parts - (unsigned long long)(n - warm) - rem_bits;
  const unsigned long long residual_bits = 2ull + 4ull + (unsigned long long)best_parts * (rice2 ? 5ull : 4ull) +
                                           (sum_q + (unsigned long long)(n - warm)) + rem_bits;
  const unsigned long long sub_bits = 8ull + bps_role * (unsigned long long)warm + 4ull + 5ull +
                                      (unsigned long long)a.precision * (unsigned long long)warm + residual_bits;

  // ======================= phase 5: the record ==============================
  if (!active || a.params == nullptr) return;
  flacenc_hip_subframe_params* rec = a.params + sf;
  if (redo && status == 0) {
    if (sl == 0) {
      rec->status = -1;
      if (a.marked_count != nullptr) atomicAdd(a.marked_count, 1u);
    }
    return;
  }
  {
    // partition j of the chosen order lives on lane j << bestk of the segment
    const int srcl = (lane & ~(LPS - 1)) | ((sl << bestk) & (LPS - 1));
    const uint32_t pv = (uint32_t)__shfl((int)my_p, srcl, 64);
    rec->rice_params[sl] = (uint8_t)((sl < (int)best_parts && status == 0) ? pv : 0u);
    uint32_t* words = reinterpret_cast<uint32_t*>(rec->rice_params);
    for (int w = LPS / 4 + sl; w < FLACENC_HIP_MAX_RICE_PARTITIONS / 4; w += LPS) words[w] = 0u;
  }
  if (sl == 0) {
    uint32_t* cw = reinterpret_cast<uint32_t*>(rec->coefs);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c0 = (2 * i < MAXP && status == 0) ? cq[2 * i < MAXP ? 2 * i : 0] : 0;
      const int c1 = (2 * i + 1 < MAXP && status == 0) ? cq[2 * i + 1 < MAXP ? 2 * i + 1 : 0] : 0;
      cw[i] = ((uint32_t)c0 & 0xFFFFu) | ((uint32_t)c1 << 16);
    }
    rec->order = (uint8_t)warm;
    rec->shift = (int8_t)shift;
    rec->precision = (uint8_t)a.precision;
    rec->rice_order = (uint8_t)(status == 0 ? rice_order : 0);
    rec->status = status;
    rec->code_bits = status == 0 ? (unsigned long long)best_bits : 0ull;
    rec->subframe_bits = status == 0 ? sub_bits : 0ull;
    rec->sum_quotients = status == 0 ? sum_q : 0ull;
  }
}

template <int MAXP, bool STEREO, int SPL, int LPS>
hipError_t launch_subwave_geom(const QlpcKernelArgs& a, hipStream_t stream) {
  using G = SubGeom<SPL, LPS>;
  constexpr int WAVES = STEREO ? 4 : 2;
  constexpr int SUBS = WAVES * G::S;
  constexpr int NIMG = STEREO ? 2 * G::S : SUBS;
  constexpr size_t smem = (size_t)(NIMG + 1) * G::Img * 4 + (size_t)SUBS * ((MAXP + 1) * 8 + 64);
  auto kern = qlpc_subwave_kernel<MAXP, STEREO, SPL, LPS>;
  static DynamicLdsOptIn opt_in;
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  const uint32_t blocks = STEREO ? ((a.n_subframes >> 2) + (uint32_t)G::S - 1u) / (uint32_t)G::S
                                 : (a.n_subframes + (uint32_t)SUBS - 1u) / (uint32_t)SUBS;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), smem, stream, a);
  return hipGetLastError();
}

template <int MAXP, bool STEREO, int SPL>
hipError_t launch_subwave(const QlpcKernelArgs& a, hipStream_t stream) {
  const uint32_t lps = a.block_size / (uint32_t)SPL;
  if (a.block_size != lps * (uint32_t)SPL) return hipErrorInvalidValue;
  if (lps == 8) return launch_subwave_geom<MAXP, STEREO, SPL, 8>(a, stream);
  if (lps == 16) return launch_subwave_geom<MAXP, STEREO, SPL, 16>(a, stream);
  if (lps == 32) return launch_subwave_geom<MAXP, STEREO, SPL, 32>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace
}  // namespace flacenc_hip
#endif
