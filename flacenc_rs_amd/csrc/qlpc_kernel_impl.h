#ifndef FLACENC_HIP_QLPC_KERNEL_IMPL_H_
#define FLACENC_HIP_QLPC_KERNEL_IMPL_H_
// qlpc_kernel_impl.h -- the QLPC analysis hot path as one fused HIP kernel for gfx950.
// (template; instantiated once per (order bucket, big) pair by qlpc_inst.hip)
//
// One workgroup analyses one subframe (one channel of one block), start to
// finish, with the samples staged once in LDS:
//
//   phase 0  coalesced int32 loads HBM -> LDS (padded 16-sample rows), max|s|
//   phase 1  Tukey windowing in f32 + autocorrelation R[0..=P] in f64
//            (lpc.rs:739-756, 533-548): every thread owns 16-sample chunks, keeps
//            the windowed samples of its chunk (+P halo) in registers and runs
//            P+1 independent fma chains; chunk partials are combined by a
//            balanced tree (wave butterfly, then LDS) -- the build's canonical
//            summation order, reproduced bit-for-bit by the CPU oracle
//   phase 2  Levinson-Durbin + coefficient quantisation (lpc.rs:633-705, 234-302),
//            serial, one lane, registers only
//   phase 3  integer residual (lpc.rs:306-390) from the LDS-resident samples,
//            written back over them in LDS
//   phase 4  partitioned-Rice parameter search (rice.rs:65-165, 246-298): per
//            finest partition a 32-entry bit table, then log2 merges
//   phase 5  coalesced residual store LDS -> HBM, one parameter record
//
// Algorithmic HBM traffic: 4 B read + 4 B written per sample.
// All `file:line` citations are relative to the flacenc-rs v0.5.1 tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flacenc_hip.h"
#include "qlpc_kernel.h"

namespace flacenc_hip {
namespace {

constexpr uint32_t kMaxPToBits = (1u << 27) - 1u;  // rice.rs:51
constexpr int kLeadRows = 2;                       // zero rows in front of the samples (halo of chunk 0/1)

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zigzag(int32_t e) {  // rice::encode_signbit, rice.rs:169-171
  return ((uint32_t)e << 1) ^ (uint32_t)(e >> 31);
}

__device__ __forceinline__ double wave_butterfly_sum(double v) {
  // balanced pairwise tree over the lane index: level k adds lanes i and i^(1<<k)
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v |= (uint32_t)__shfl_xor((int)v, m, 64);
  return v;
}

// exact ceil(log2(m)) for finite m > 0 from the exponent/mantissa fields
__device__ __forceinline__ int ceil_log2_pos(double m) {
  uint64_t b = (uint64_t)__double_as_longlong(m);
  int e = (int)((b >> 52) & 0x7FF);
  uint64_t frac = b & 0xFFFFFFFFFFFFFull;
  if (e == 0) return -32752;  // zero / subnormal: far below the clamp of lpc.rs:247-250
  return (e - 1023) + (frac != 0 ? 1 : 0);
}

// f32::log2 = libm log2f.  glibc's algorithm (sysdeps/ieee754/flt-32/e_log2f.c, 2.27+): 16-entry
// {1/c, log2 c} table around OFF = 0x3f330000, degree-4 polynomial in double, one rounding to
// float.  Same restatement as oracle/flacenc_oracle.c orc_log2f (checked there against the
// host libm over every positive float).
__device__ const double kLog2fTab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
    0x1.49539f0f010bp+0,  -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
    0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0,  -0x1.97c1d1b3b7afp-3,
    0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
    0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,  0x1.ca4b31f026aap-1,  0x1.476a9543891bap-3,
    0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,  0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
    0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,  0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
};

__device__ __forceinline__ float dev_log2f(float x) {
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0u) return -__builtin_inff();
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
    ix = __float_as_uint(x * 0x1p23f) - (23u << 23);
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const uint32_t top = tmp & 0xff800000u;
  const int k = (int)tmp >> 23;
  const double z = (double)__uint_as_float(ix - top);
  const double r = z * kLog2fTab[2 * i] - 1.0;
  const double y0 = kLog2fTab[2 * i + 1] + (double)k;
  const double r2 = r * r;
  double y = 0x1.ecabf496832ep-2 * r + -0x1.715479ffae3dep-1;
  y = -0x1.712b6f70a7e4dp-2 * r2 + y;
  const double p = 0x1.715475f35c8b8p0 * r + y0;
  y = y * r2 + p;
  return (float)y;
}

// one partition of estimate_entropy (coding.rs:215-222): `sum` is the exact integer sum of
// |e| over the partition (rounded to f32 once -- the canonical definition of DESIGN.md; equal to
// find_sum_abs_f32 in either reference build while the sum stays below 2^24)
__device__ __forceinline__ uint32_t approx_ent_bits(double sum, uint32_t count) {
  const float sum_errors = (float)sum;
  const float cnt = (float)count;
  const float avg_errors = sum_errors * 2.0f / (cnt + 0.00001f);
  const float geom_p = 1.0f / (avg_errors + 1.0f);
  const float xent = __builtin_fmaf(avg_errors, -dev_log2f(1.0f - geom_p), -dev_log2f(geom_p));
  const float v = xent * cnt;
  return v > 0.0f ? (uint32_t)v : 0u;  // `as usize`: NaN and negatives -> 0
}

// Optional per-phase timestamps (s_memtime, shader clock) for the profiling build of
// tools/phase_profile.py: workgroup leader only, never read by the kernel itself.
#define FLACENC_STAMP(slot)                                                             \
  do {                                                                                  \
    if (a.stamps && tid == 0) a.stamps[(size_t)sf * 8 + (slot)] = (unsigned long long)clock64(); \
  } while (0)

// LDS layout ------------------------------------------------------------------
// Samples live in rows of 16 int32 with a row stride of ROWSTRIDE dwords
// (20 = 16 + 4 pad: a thread reading its own row with ds_read_b128 then lands
// on a distinct 4-bank group for any 16 consecutive lanes; 16 = unpadded, for
// blocks too large for the padded image).  Two all-zero rows precede row 0.
template <int ROWSTRIDE>
__device__ __forceinline__ int sidx(int t) {
  return ((t >> 4) + kLeadRows) * ROWSTRIDE + (t & 15);
}

struct SmemLayout {
  int32_t* sbuf;
  uint32_t* tables;
  double* red;
  double* racc;    // R[33]
  int32_t* qc;     // 32 quantised coefficients as int32
  uint32_t* misc;  // see kMisc*
  unsigned long long* level_bits;  // [9]
  uint8_t* ps;     // per-level rice parameters, 2*nparts bytes
  unsigned long long* fsum;  // fixed-LPC order selection: [5][64] partition sums + [8] totals
};
constexpr int kFixedSumWords = 5 * 64 + 16;

enum {
  kMiscMaxAbs = 0,
  kMiscOrder,
  kMiscShift,
  kMiscStatus,
  kMiscWide,
  kMiscOrBits,
  kMiscSat,
  kMiscBestOrder,
  kMiscCount = 16
};

// ---------------------------------------------------------------------------
// phase 2: Levinson-Durbin + quantisation, one lane, fully unrolled
// ---------------------------------------------------------------------------
// symmetric_levinson_recursion::<f64, _>, lpc.rs:633-705, with
// coefs = R[0..P], ys = R[1..P+1] (lpc.rs:792-796).  NOTE lpc.rs:679-682: the
// unlabeled `continue` targets `for n in 1..order`, so a zero denominator
// skips iteration n (it does not restart).  Operation order and every fma
// follow the reference; forward[] is updated in place pairwise (d, n-d) which
// reads exactly the old values forward_next[] would be computed from.
template <int MAXP>
__device__ int levinson_quantize(const double* __restrict__ Rl, int P, int precision,
                                 double (&a)[MAXP], int32_t* qc_out, int* order_out,
                                 int* shift_out) {
  double R[MAXP + 1];
#pragma unroll
  for (int i = 0; i <= MAXP; ++i) R[i] = (i <= P) ? Rl[i] : 0.0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) a[i] = 0.0;

  int status = FLACENC_HIP_SUBFRAME_OK;
#pragma unroll
  for (int i = 0; i <= MAXP; ++i) {
    if (i <= P) {
      uint64_t b = (uint64_t)__double_as_longlong(R[i]);
      if (((b >> 52) & 0x7FF) == 0x7FF) status |= FLACENC_HIP_SUBFRAME_NONFINITE;  // lpc.rs:786-791
    }
  }
  if (status == 0 && !(R[0] >= 0.0)) status |= FLACENC_HIP_SUBFRAME_NEG_ENERGY;  // lpc.rs:646
  if (status == 0 && R[0] == 0.0) {
    bool allzero = true;
#pragma unroll
    for (int i = 0; i <= MAXP; ++i)
      if (i <= P && R[i] != 0.0) allzero = false;
    if (!allzero) status |= FLACENC_HIP_SUBFRAME_NEG_ENERGY;  // lpc.rs:652-655
  }

  if (status == 0 && R[0] != 0.0) {
    double fwd[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) fwd[i] = 0.0;
    fwd[0] = 1.0 / R[0];   // Float::recip(coefs[0] + diagonal_loading), loading = 0
    a[0] = R[1] / R[0];    // ys[0] / (coefs[0] + diagonal_loading)
#pragma unroll
    for (int n = 1; n < MAXP; ++n) {
      if (n < P) {
        double err = 0.0;
#pragma unroll
        for (int d = 0; d < n; ++d) err = __builtin_fma(R[n - d], fwd[d], err);
        double denom = __builtin_fma(err, -err, 1.0);
        if (denom != 0.0) {
          double alpha = 1.0 / denom;
          double beta = -alpha * err;
          // forward_next[d] = fma(alpha, forward[d], beta * forward[n - d]), d <= n
#pragma unroll
          for (int d = 0; 2 * d <= n; ++d) {
            double fd = fwd[d], fe = fwd[n - d];
            double nd = __builtin_fma(alpha, fd, beta * fe);
            double ne = __builtin_fma(alpha, fe, beta * fd);
            fwd[d] = nd;
            fwd[n - d] = ne;
          }
          double delta = 0.0;
#pragma unroll
          for (int d = 0; d < n; ++d) delta = __builtin_fma(R[n - d], a[d], delta);
          double resid = R[n + 1] - delta;  // ys[n] - delta
#pragma unroll
          for (int d = 0; d <= n; ++d) a[d] = __builtin_fma(resid, fwd[n - d], a[d]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i < P) {
        uint64_t b = (uint64_t)__double_as_longlong(a[i]);
        if (((b >> 52) & 0x7FF) == 0x7FF) status |= FLACENC_HIP_SUBFRAME_NONFINITE;  // lpc.rs:797-799
      }
    }
  }

  // quantize_parameters, lpc.rs:273-302 (find_shift :234-254, quantize_parameter :258-270)
  int shift = 0, order = 0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) qc_out[i] = 0;
  if (status == 0) {
    double max_abs = 0.0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (i < P) max_abs = fmax(max_abs, fabs(a[i]));
    int abs_log2 = ceil_log2_pos(max_abs);
    if (abs_log2 < -32752) abs_log2 = -32752;
    shift = (precision - 1) - abs_log2;
    shift = shift < 0 ? 0 : (shift > 15 ? 15 : shift);
    double scalefac = (double)(1 << shift);
    int lo = -(1 << (precision - 1)), hi = (1 << (precision - 1)) - 1;
    order = 1;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i < P) {
        double s = round(a[i] * scalefac);  // half away from zero
        s = s < -32768.0 ? -32768.0 : (s > 32767.0 ? 32767.0 : s);
        int q = (int)s;
        q = q < lo ? lo : (q > hi ? hi : q);
        qc_out[i] = q;
        if (q != 0) order = i + 1;  // tail-zero truncation, min 1 (lpc.rs:295-299)
      }
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (i >= order) qc_out[i] = 0;
  }
  *order_out = order;
  *shift_out = shift;
  return status;
}

// ---------------------------------------------------------------------------
// phase 1 inner block: 16 samples x (MAXP+1) lags, registers only
// ---------------------------------------------------------------------------
template <int MAXP, int HP, bool MASK>
__device__ __forceinline__ void acorr_chunk(const double (&dw)[HP + 16], double (&acc)[MAXP + 1],
                                            int t0, int P) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double cur = dw[HP + k];
    if (MASK) cur = (t0 + k >= P) ? cur : 0.0;  // common lower bound t = P for every lag (lpc.rs:542)
    if (k == 0) {
      // chain start: fma(x, y, +0.0) -- written with the literal so no zero-filled accumulators
#pragma unroll
      for (int tau = 0; tau <= MAXP; ++tau) acc[tau] = __builtin_fma(cur, dw[HP + k - tau], 0.0);
    } else {
#pragma unroll
      for (int tau = 0; tau <= MAXP; ++tau) acc[tau] = __builtin_fma(cur, dw[HP + k - tau], acc[tau]);
    }
  }
}

// ---------------------------------------------------------------------------
// the fused kernel
// ---------------------------------------------------------------------------
template <int MAXP, bool BIG>
__global__ void __launch_bounds__(MAXP <= 12 ? 1024 : (MAXP <= 16 ? 512 : 256))
qlpc_subframe_kernel(QlpcKernelArgs a) {
  constexpr int ROWSTRIDE = BIG ? 16 : 20;
  constexpr int HP = (MAXP + 3) & ~3;  // halo samples loaded, multiple of 4
  constexpr int NLAG = MAXP + 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int T = blockDim.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int W = T >> 6;
  const int n = (int)a.block_size;
  const int rows = (n + 15) >> 4;
  const int J = (rows + T - 1) / T;
  int Jp = 1;
  while (Jp < J) Jp <<= 1;
  const int P = (int)a.lpc_order;
  // Stereo mode: the four workgroups of a frame should share an XCD (and its L2) so that the
  // left/right channels are fetched from HBM once: workgroups are dealt round-robin over the
  // 8 XCDs, so block b and b+8 share one.  Blocks are grouped by 32: block 32g + 8r + x
  // analyses role r of frame 8g + x.  (Placement is a speed matter only.)
  uint32_t sf = blockIdx.x;
  if (a.stereo && (a.n_subframes & 31u) == 0) {
    const uint32_t g = sf >> 5, r = (sf >> 3) & 3u, x = sf & 7u;
    sf = ((g << 3) + x) * 4u + r;
  }
  // clean-up launch behind the big-block kernels: only the subframes they marked (residuals of 2^26
  // and more, whose bit tables need the literal chunk-clamped sums) are redone here
  if (a.only_marked && a.params[sf].status != -1) return;

  // ---- carve LDS (every offset a multiple of 16 bytes) ----
  SmemLayout L;
  {
    unsigned char* p = smem_raw;
    L.sbuf = reinterpret_cast<int32_t*>(p);
    p += (size_t)(rows + kLeadRows) * ROWSTRIDE * 4;
    L.red = reinterpret_cast<double*>(p);
    p += (size_t)Jp * W * NLAG * 8;
    p = smem_raw + (((size_t)(p - smem_raw) + 15) & ~(size_t)15);
    L.racc = reinterpret_cast<double*>(p);
    p += 40 * 8;
    L.level_bits = reinterpret_cast<unsigned long long*>(p);
    p += 16 * 8;
    L.qc = reinterpret_cast<int32_t*>(p);
    p += 32 * 4;
    L.misc = reinterpret_cast<uint32_t*>(p);
    p += kMiscCount * 4;
    L.ps = reinterpret_cast<uint8_t*>(p);
    p += 2 * FLACENC_HIP_MAX_RICE_PARTITIONS;
    L.fsum = reinterpret_cast<unsigned long long*>(p);
    p += kFixedSumWords * 8;
    L.tables = BIG ? (a.table_scratch + (size_t)sf * FLACENC_HIP_MAX_RICE_PARTITIONS * 32)
                   : reinterpret_cast<uint32_t*>(p);
  }

  FLACENC_STAMP(0);
  // ======================= phase 0: load ===================================
  // Plain mode: subframe sf is the block at samples + sf*stride.  Stereo mode
  // (try_stereo_coding, coding.rs:476-484): workgroups 4f..4f+3 analyse L, R,
  // M = (l + r) >> 1 and S = l - r of frame f, formed here from the two channels.
  const int role = a.stereo ? (int)(sf & 3u) : 0;
  const int32_t* __restrict__ src =
      a.stereo ? a.samples + (size_t)(2u * (sf >> 2) + (role == 1 ? 1u : 0u)) * a.stride
               : a.samples + (size_t)sf * a.stride;
  const int32_t* __restrict__ src2 = src + a.stride;  // right channel, roles 2 and 3 only
  if (tid < kMiscCount) L.misc[tid] = 0;
  if (tid < 16) L.level_bits[tid] = 0ull;
  for (int i = tid; i < kLeadRows * ROWSTRIDE; i += T) L.sbuf[i] = 0;
  uint32_t my_maxabs = 0;
  {
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((a.stride & 3) == 0);
    for (int q = tid; q < rows * 4; q += T) {
      int t = q * 4;
      int4 v;
      if (vec_ok && t + 3 < n) {
        v = *reinterpret_cast<const int4*>(src + t);
      } else {
        v.x = (t + 0 < n) ? src[t + 0] : 0;
        v.y = (t + 1 < n) ? src[t + 1] : 0;
        v.z = (t + 2 < n) ? src[t + 2] : 0;
        v.w = (t + 3 < n) ? src[t + 3] : 0;
      }
      if (role >= 2) {
        int4 r;
        if (vec_ok && t + 3 < n) {
          r = *reinterpret_cast<const int4*>(src2 + t);
        } else {
          r.x = (t + 0 < n) ? src2[t + 0] : 0;
          r.y = (t + 1 < n) ? src2[t + 1] : 0;
          r.z = (t + 2 < n) ? src2[t + 2] : 0;
          r.w = (t + 3 < n) ? src2[t + 3] : 0;
        }
        if (role == 2) {
          v.x = (v.x + r.x) >> 1;
          v.y = (v.y + r.y) >> 1;
          v.z = (v.z + r.z) >> 1;
          v.w = (v.w + r.w) >> 1;
        } else {
          v.x -= r.x;
          v.y -= r.y;
          v.z -= r.z;
          v.w -= r.w;
        }
      }
      *reinterpret_cast<int4*>(&L.sbuf[sidx<ROWSTRIDE>(t)]) = v;
      uint32_t ax = (uint32_t)(v.x < 0 ? -(int64_t)v.x : (int64_t)v.x);
      uint32_t ay = (uint32_t)(v.y < 0 ? -(int64_t)v.y : (int64_t)v.y);
      uint32_t az = (uint32_t)(v.z < 0 ? -(int64_t)v.z : (int64_t)v.z);
      uint32_t aw = (uint32_t)(v.w < 0 ? -(int64_t)v.w : (int64_t)v.w);
      ax = ax > ay ? ax : ay;
      az = az > aw ? az : aw;
      ax = ax > az ? ax : az;
      my_maxabs = my_maxabs > ax ? my_maxabs : ax;
    }
  }
  __syncthreads();
  my_maxabs = wave_max_u32(my_maxabs);
  if (lane == 0) atomicMax(&L.misc[kMiscMaxAbs], my_maxabs);

  FLACENC_STAMP(1);
  if (a.fixed_mode != 0) {
    // ======================= fixed-LPC candidate (coding.rs:298-331) =========
    // The predictor is one of FIXED_LPC_COEFS (decode.rs:179-185) with shift 0 instead of the
    // Levinson solution; everything downstream (error signal, Rice search, bit counts) is the
    // QLPC machinery.  fixed_mode 1: pick the order with estimate_entropy (OrderSel::ApproxEnt,
    // coding.rs:265-287) first; 2: order = forced_uniform (one pass of OrderSel::BitCount,
    // coding.rs:243-264); 3: order = forced_orders[sf] (BitCount's final pass).
    const unsigned long long bps_sf = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + (role == 3 ? 1u : 0u));
    if (a.fixed_mode == 1u) {
      unsigned long long* const fsum = L.fsum;
      unsigned long long* const ftot = L.fsum + 5 * 64;
      for (int i = tid; i < kFixedSumWords; i += T) fsum[i] = 0ull;
      __syncthreads();
      const int parts = (int)a.fixed_partitions;
      const int psz = (n + parts - 1) / parts;  // block_size.div_ceil(partitions), coding.rs:209
      // exact integer sums of |e_k[t]| per (order, partition); e_k by the binomial form of k
      // wrapping differences with zero history (reset_fixed_lpc_errors, coding.rs:182-197)
      for (int j = 0; j < J; ++j) {
        const int c = tid + j * T;
        const int t0 = c << 4;
        if (c < rows) {
          uint32_t x[20];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int4 v = *reinterpret_cast<const int4*>(&L.sbuf[sidx<ROWSTRIDE>(t0 - 4 + 4 * i)]);
            x[4 * i + 0] = (uint32_t)v.x;
            x[4 * i + 1] = (uint32_t)v.y;
            x[4 * i + 2] = (uint32_t)v.z;
            x[4 * i + 3] = (uint32_t)v.w;
          }
          int cur = t0 / psz;
          unsigned long long acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int t = t0 + k;
            if (t < n) {
              const int pidx = t / psz;
              if (pidx != cur) {
#pragma unroll
                for (int o = 0; o < 5; ++o) {
                  if (acc[o]) atomicAdd(&fsum[o * 64 + cur], acc[o]);
                  acc[o] = 0ull;
                }
                cur = pidx;
              }
              const uint32_t x0 = x[4 + k], x1 = x[3 + k], x2 = x[2 + k], x3 = x[1 + k], x4 = x[k];
              const int32_t e0 = (int32_t)x0;
              const int32_t e1 = (int32_t)(x0 - x1);
              const int32_t e2 = (int32_t)(x0 - 2u * x1 + x2);
              const int32_t e3 = (int32_t)(x0 - 3u * x1 + 3u * x2 - x3);
              const int32_t e4 = (int32_t)(x0 - 4u * x1 + 6u * x2 - 4u * x3 + x4);
              acc[0] += (unsigned long long)(e0 < 0 ? -(int64_t)e0 : (int64_t)e0);
              acc[1] += (unsigned long long)(e1 < 0 ? -(int64_t)e1 : (int64_t)e1);
              acc[2] += (unsigned long long)(e2 < 0 ? -(int64_t)e2 : (int64_t)e2);
              acc[3] += (unsigned long long)(e3 < 0 ? -(int64_t)e3 : (int64_t)e3);
              acc[4] += (unsigned long long)(e4 < 0 ? -(int64_t)e4 : (int64_t)e4);
            }
          }
#pragma unroll
          for (int o = 0; o < 5; ++o)
            if (acc[o]) atomicAdd(&fsum[o * 64 + cur], acc[o]);
        }
      }
      __syncthreads();
      // estimate_entropy, coding.rs:200-227: one (order, partition) pair per thread
      for (int i = tid; i < 5 * parts; i += T) {
        const int ord = i / parts, pidx = i - ord * parts;
        if (ord <= (int)a.fixed_max_order) {
          const long long off0 = (long long)pidx * psz;
          const int offset = off0 < n ? (int)off0 : n;
          const int end = offset + psz < n ? offset + psz : n;
          const int plen = end - offset;
          if (end >= ord) {
            const int cnt = end - ord < plen ? end - ord : plen;
            const uint32_t pb = approx_ent_bits((double)fsum[ord * 64 + pidx], (uint32_t)cnt);
            atomicAdd(&ftot[ord], (unsigned long long)pb);
          }
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      int k;
      unsigned long long key = 0ull;
      if (a.fixed_mode == 1u) {
        k = 0;
        key = ~0ull;
        for (int ord = 0; ord <= (int)a.fixed_max_order; ++ord) {
          const unsigned long long kk = L.fsum[5 * 64 + ord] + bps_sf * (unsigned long long)ord;
          if (a.fixed_keys) a.fixed_keys[(size_t)sf * 8 + ord] = kk;
          if (kk < key) {  // min_by_key keeps the first minimum
            key = kk;
            k = ord;
          }
        }
        if (a.selector_keys) a.selector_keys[sf] = key;
      } else if (a.fixed_mode == 2u) {
        k = (int)a.forced_uniform;
      } else {
        k = (int)a.forced_orders[sf];
      }
      for (int i = 0; i < 32; ++i) L.qc[i] = 0;
      L.qc[0] = k;  // FIXED_LPC_COEFS[k]: 0 / 1 / 2,-1 / 3,-3,1 / 4,-6,4,-1
      L.qc[1] = k == 2 ? -1 : (k == 3 ? -3 : (k == 4 ? -6 : 0));
      L.qc[2] = k == 3 ? 1 : (k == 4 ? 4 : 0);
      L.qc[3] = k == 4 ? -1 : 0;
      const uint64_t maxabs = (uint64_t)L.misc[kMiscMaxAbs];
      const uint64_t sumabs = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 3 : (k == 3 ? 7 : 15)));
      const bool narrow = (maxabs * sumabs < 0x7FFFFFFFull) && (maxabs < (1u << 23));
      L.misc[kMiscOrder] = (uint32_t)k;
      L.misc[kMiscShift] = 0u;
      L.misc[kMiscStatus] = 0u;
      L.misc[kMiscWide] = narrow ? 0u : 1u;
    }
    __syncthreads();
  } else if (a.lpc_stage == 3u) {
    // ======================= predictor solved by levinson_batch_kernel ========
    // (three-launch split for large orders: the serial recursion runs one subframe per LANE in a
    // kernel of its own instead of one per workgroup here; see launch_qlpc)
    if (tid == 0) {
      const int32_t* pr = a.pred + (size_t)sf * 36;
      int64_t sumabs = 0;
      for (int i = 0; i < 32; ++i) {
        L.qc[i] = pr[i];
        sumabs += pr[i] < 0 ? -pr[i] : pr[i];
      }
      const uint64_t maxabs = (uint64_t)L.misc[kMiscMaxAbs];
      const bool narrow = (maxabs * (uint64_t)sumabs < 0x7FFFFFFFull) && (maxabs < (1u << 23));
      L.misc[kMiscOrder] = (uint32_t)pr[32];
      L.misc[kMiscShift] = (uint32_t)pr[33];
      L.misc[kMiscStatus] = (uint32_t)pr[34];
      L.misc[kMiscWide] = narrow ? 0u : 1u;
    }
    __syncthreads();
  } else {
  // ======================= phase 1: window + autocorrelation ==============
  // window table has 32 floats of zero padding in front and is padded to whole rows
  const float* __restrict__ wtab = a.window ? (a.window + 32) : nullptr;
  for (int j = 0; j < J; ++j) {
    const int c = tid + j * T;  // chunk index
    const int t0 = c << 4;
    double acc[NLAG];
#pragma unroll
    for (int i = 0; i < NLAG; ++i) acc[i] = 0.0;
    if (c < rows) {
      // samples t0-HP .. t0+15 from LDS
      int sw[HP + 16];
#pragma unroll
      for (int i = 0; i < (HP + 16) / 4; ++i) {
        int t = t0 - HP + 4 * i;
        int4 v = *reinterpret_cast<const int4*>(&L.sbuf[sidx<ROWSTRIDE>(t)]);
        sw[4 * i + 0] = v.x;
        sw[4 * i + 1] = v.y;
        sw[4 * i + 2] = v.z;
        sw[4 * i + 3] = v.w;
      }
      double dw[HP + 16];
      const bool flat = (wtab == nullptr) || (t0 - HP >= a.flat_lo && t0 + 16 <= a.flat_hi);
      if (flat) {
        // w == 1.0f: (f32)s * 1.0f == (f32)s exactly (lpc.rs:751-754)
#pragma unroll
        for (int i = 0; i < HP + 16; ++i) dw[i] = (double)(float)sw[i];
      } else {
#pragma unroll
        for (int i = 0; i < (HP + 16) / 4; ++i) {
          float4 wv = *reinterpret_cast<const float4*>(wtab + (t0 - HP + 4 * i));
          dw[4 * i + 0] = (double)((float)sw[4 * i + 0] * wv.x);
          dw[4 * i + 1] = (double)((float)sw[4 * i + 1] * wv.y);
          dw[4 * i + 2] = (double)((float)sw[4 * i + 2] * wv.z);
          dw[4 * i + 3] = (double)((float)sw[4 * i + 3] * wv.w);
        }
      }
      if (t0 < P) acorr_chunk<MAXP, HP, true>(dw, acc, t0, P);
      else acorr_chunk<MAXP, HP, false>(dw, acc, t0, P);
    }
    // balanced tree over the chunk index: lanes first ...
#pragma unroll
    for (int tau = 0; tau < NLAG; ++tau) {
      double v = wave_butterfly_sum(acc[tau]);
      if (lane == 0) L.red[((size_t)j * W + wave) * NLAG + tau] = v;
    }
  }
  for (int i = tid; i < (Jp - J) * W * NLAG; i += T) L.red[(size_t)J * W * NLAG + i] = 0.0;
  __syncthreads();
  // ... then waves and chunk rounds (entry e = j*W + w are the chunk index's high bits)
  if (tid < NLAG) {
    const int E = Jp * W;
    for (int s = 1; s < E; s <<= 1)
      for (int e = 0; e < E; e += 2 * s)
        L.red[(size_t)e * NLAG + tid] += L.red[(size_t)(e + s) * NLAG + tid];
    double r = L.red[tid];
    L.racc[tid] = r;
    if (a.autocorr) a.autocorr[(size_t)sf * 33 + tid] = (tid <= P) ? r : 0.0;
  }
  if (a.autocorr && tid >= NLAG && tid < 33) a.autocorr[(size_t)sf * 33 + tid] = 0.0;
  if (a.lpc_stage == 1u) return;  // first launch of the split: R[] is all this one produces
  __syncthreads();

  FLACENC_STAMP(2);
  // ======================= phase 2: Levinson + quantisation ================
  if (tid == 0) {
    double coef[MAXP];
    int32_t qc[MAXP];
    int order, shift;
    int status = levinson_quantize<MAXP>(L.racc, P, (int)a.precision, coef, qc, &order, &shift);
    int64_t sumabs = 0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      L.qc[i] = qc[i];
      sumabs += qc[i] < 0 ? -qc[i] : qc[i];
    }
    for (int i = MAXP; i < 32; ++i) L.qc[i] = 0;
    // compute_error's path choice (lpc.rs:361-377): i32 lanes iff max|s| * sum|c| < i32::MAX.
    // The narrow path below additionally needs |s| < 2^23 for the 24-bit multiplier.
    uint64_t maxabs = (uint64_t)L.misc[kMiscMaxAbs];
    bool narrow = (maxabs * (uint64_t)sumabs < 0x7FFFFFFFull) && (maxabs < (1u << 23));
    L.misc[kMiscOrder] = (uint32_t)order;
    L.misc[kMiscShift] = (uint32_t)shift;
    L.misc[kMiscStatus] = (uint32_t)status;
    L.misc[kMiscWide] = narrow ? 0u : 1u;
    if (a.lpc_coefs) {
#pragma unroll
      for (int i = 0; i < MAXP; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = (i < P && status == 0) ? coef[i] : 0.0;
      for (int i = MAXP; i < 32; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = 0.0;
    }
  }
  __syncthreads();
  }  // QLPC predictor

  FLACENC_STAMP(3);
  const int warm = (int)L.misc[kMiscOrder];
  const int shift = (int)L.misc[kMiscShift];
  const int status = (int)L.misc[kMiscStatus];
  const bool wide = L.misc[kMiscWide] != 0;

  // ======================= phase 3: residual ===============================
  // e[t] = s[t] - ((sum_j c_j * s[t-1-j]) >> shift), e[0..order') = 0 (lpc.rs:306-350).
  // Chunk rounds run downwards so a round only overwrites rows no later round reads.
  uint32_t my_or = 0;
  {
    int32_t cq[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) cq[i] = L.qc[i];
    for (int j = J - 1; j >= 0; --j) {
      const int c = tid + j * T;
      const int t0 = c << 4;
      int32_t e[16];
      if (c < rows) {
        int sw[HP + 16];
#pragma unroll
        for (int i = 0; i < (HP + 16) / 4; ++i) {
          int t = t0 - HP + 4 * i;
          int4 v = *reinterpret_cast<const int4*>(&L.sbuf[sidx<ROWSTRIDE>(t)]);
          sw[4 * i + 0] = v.x;
          sw[4 * i + 1] = v.y;
          sw[4 * i + 2] = v.z;
          sw[4 * i + 3] = v.w;
        }
        if (!wide) {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            int32_t pred = 0;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) pred += __mul24(cq[i], sw[HP + k - 1 - i]);
            e[k] = sw[HP + k] - (pred >> shift);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            int64_t pred = 0;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) pred += (int64_t)cq[i] * (int64_t)sw[HP + k - 1 - i];
            e[k] = (int32_t)(uint32_t)(uint64_t)((int64_t)sw[HP + k] - (pred >> shift));
          }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          int t = t0 + k;
          if (t < warm || t >= n || status != 0) e[k] = 0;
          my_or |= zigzag(e[k]);
        }
      }
      __syncthreads();
      if (c < rows) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int4 v;
          v.x = e[4 * i + 0];
          v.y = e[4 * i + 1];
          v.z = e[4 * i + 2];
          v.w = e[4 * i + 3];
          *reinterpret_cast<int4*>(&L.sbuf[sidx<ROWSTRIDE>(t0 + 4 * i)]) = v;
        }
      }
    }
  }
  my_or = wave_or_u32(my_or);
  if (lane == 0) atomicOr(&L.misc[kMiscOrBits], my_or);
  __syncthreads();

  FLACENC_STAMP(4);
  // ======================= phase 4: partitioned-Rice search ================
  // finest_partition_order(n, max(64, warm)), rice.rs:157-165, 247-250 (warm <= 32 < 64)
  int fo;
  {
    uint32_t max_splits = (uint32_t)n / 64u;
    int lg = 31 - __clz((int)max_splits);
    int tz = __ffs(n) - 1;
    fo = lg < tz ? lg : tz;
    fo = fo < 15 ? fo : 15;
    if (fo > 8) fo = 8;  // unreachable for n <= 32767
  }
  const int nparts = 1 << fo;
  const int psize = n >> fo;
  const int hw = tid >> 5;      // half-wave id: one bit table (32 lanes = 32 parameters) each
  const int NHW = T >> 5;
  const uint32_t p = (uint32_t)(tid & 31);

  // Parameters beyond the residual's bit length can never win: for p >= bitlen every
  // u >> p is 0, so a table entry is 4 + len*(p+1), strictly increasing in p, in every
  // partition and therefore in every merged table.  Capping the search at
  // min(max_p, bitlen) returns the same minimiser and the same bits as rice.rs:115-141.
  const bool finest_only = a.rice_finest_only != 0;  // FLACENC_HIP_FLAG_FINEST_RICE_ORDER
  const uint32_t maxu = L.misc[kMiscOrBits];
  const uint32_t bitlen = maxu ? (uint32_t)(32 - __clz((int)maxu)) : 0u;
  const uint32_t max_p = a.max_rice_parameter < bitlen ? a.max_rice_parameter : bitlen;

  // Fast path: finest partitions of exactly 64 samples (every power-of-two block) are the
  // 4 x 16 samples of 4 adjacent lanes.  Sum_i (u_i >> p) = Sum_b C_b 2^(b-p) with C_b the
  // number of samples whose bit b is set; C_b is counted for all 32 b at once, bit-sliced:
  // a carry-save adder tree over the thread's 16 words gives 5 bit-planes, two cross-lane
  // adds (DPP quad permutes) give the 7 planes of the partition's counts, and then
  // Sum_i (u_i >> p) = Sum_k (plane_k >> p) << k.  Exact totals equal the reference's
  // chunk-clamped u32 sums (rice.rs:75-98) as long as no 16-sample chunk can wrap or
  // exceed the clamp before the final min, i.e. u < 2^26; otherwise the literal path runs.
  int first_generic_level = 0;
  const bool fast = (psize == 64) && (maxu < (1u << 26));
  if (fast) {
    int kw = fo < 4 ? fo : 4;  // merge levels that stay inside one wave (16 partitions)
    for (int j = 0; j < J; ++j) {
      const int c = tid + j * T;
      uint32_t u[16];
      if (c < rows) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int4 v = *reinterpret_cast<const int4*>(&L.sbuf[sidx<ROWSTRIDE>((c << 4) + 4 * i)]);
          u[4 * i + 0] = zigzag(v.x);
          u[4 * i + 1] = zigzag(v.y);
          u[4 * i + 2] = zigzag(v.z);
          u[4 * i + 3] = zigzag(v.w);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = 0;
      }
      // Harley-Seal carry-save adder tree: 16 words -> planes {1, 2, 4, 8, 16}
      uint32_t pl[7];
      {
        uint32_t ones = 0, twos = 0, fours = 0, eights = 0, sixteens;
        uint32_t twosA, twosB, foursA, foursB, eightsA, eightsB;
#define FLACENC_CSA(h, l, a_, b_, c_)                  \
  {                                                    \
    uint32_t t_ = (a_) ^ (b_);                         \
    uint32_t h_ = (t_ & (c_)) | (~t_ & (a_));          \
    l = t_ ^ (c_);                                     \
    h = h_;                                            \
  }
        FLACENC_CSA(twosA, ones, ones, u[0], u[1])
        FLACENC_CSA(twosB, ones, ones, u[2], u[3])
        FLACENC_CSA(foursA, twos, twos, twosA, twosB)
        FLACENC_CSA(twosA, ones, ones, u[4], u[5])
        FLACENC_CSA(twosB, ones, ones, u[6], u[7])
        FLACENC_CSA(foursB, twos, twos, twosA, twosB)
        FLACENC_CSA(eightsA, fours, fours, foursA, foursB)
        FLACENC_CSA(twosA, ones, ones, u[8], u[9])
        FLACENC_CSA(twosB, ones, ones, u[10], u[11])
        FLACENC_CSA(foursA, twos, twos, twosA, twosB)
        FLACENC_CSA(twosA, ones, ones, u[12], u[13])
        FLACENC_CSA(twosB, ones, ones, u[14], u[15])
        FLACENC_CSA(foursB, twos, twos, twosA, twosB)
        FLACENC_CSA(eightsB, fours, fours, foursA, foursB)
        FLACENC_CSA(sixteens, eights, eights, eightsA, eightsB)
#undef FLACENC_CSA
        pl[0] = ones;
        pl[1] = twos;
        pl[2] = fours;
        pl[3] = eights;
        pl[4] = sixteens;
      }
      // bit-sliced ripple adds with lane^1 (quad_perm 1,0,3,2) then lane^2 (quad_perm 2,3,0,1)
      {
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          uint32_t b = (uint32_t)__builtin_amdgcn_mov_dpp((int)pl[k], 0xB1, 0xF, 0xF, false);
          uint32_t t_ = pl[k] ^ b;
          uint32_t s_ = t_ ^ carry;
          carry = (t_ & carry) | (~t_ & pl[k]);
          pl[k] = s_;
        }
        pl[5] = carry;
        carry = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          uint32_t b = (uint32_t)__builtin_amdgcn_mov_dpp((int)pl[k], 0x4E, 0xF, 0xF, false);
          uint32_t t_ = pl[k] ^ b;
          uint32_t s_ = t_ ^ carry;
          carry = (t_ & carry) | (~t_ & pl[k]);
          pl[k] = s_;
        }
        pl[6] = carry;
      }
      // this lane's share of the partition's table: parameters p = (lane & 3) + 4 i
      const int q0 = c >> 2;  // finest partition index
      const uint32_t len = 64u - (q0 == 0 ? (uint32_t)warm : 0u);
      const uint32_t pq = (uint32_t)(tid & 3);
      uint32_t tv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t pp = pq + 4u * (uint32_t)i;
        uint32_t sum = 0;
        if (4u * (uint32_t)i <= max_p) {  // block-uniform trip count
#pragma unroll
          for (int k = 0; k < 7; ++k) sum += (pl[k] >> pp) << k;
        }
        sum = sum < kMaxPToBits ? sum : kMaxPToBits;
        uint32_t v = sum + (4u + len * (pp + 1u));  // rice.rs:69-71, 95-98
        tv[i] = v < kMaxPToBits ? v : kMaxPToBits;
      }
      // orders fo .. fo-kw inside the wave: minimiser per partition (quad, then wider groups),
      // merge with the neighbouring group by a lane butterfly (rice.rs:144-152, 193-216)
      for (int k = 0; k <= (finest_only ? 0 : kw); ++k) {
        if (k > 0) {
          const int xm = 2 << k;  // lane xor 4, 8, 16, 32
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            uint32_t o = (uint32_t)__shfl_xor((int)tv[i], xm, 64);
            uint32_t v = tv[i] + o - 4u;
            tv[i] = v < kMaxPToBits ? v : kMaxPToBits;
          }
        }
        uint32_t packed = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t pp = pq + 4u * (uint32_t)i;
          uint32_t cand = (((pp <= max_p) ? tv[i] : 0xFFFFFFFFu) << 5) | pp;
          packed = cand < packed ? cand : packed;
        }
        {
          uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)packed, 0xB1, 0xF, 0xF, false);
          packed = o < packed ? o : packed;
          o = (uint32_t)__builtin_amdgcn_mov_dpp((int)packed, 0x4E, 0xF, 0xF, false);
          packed = o < packed ? o : packed;
        }
        const int qk = q0 >> k;
        const int m = nparts >> k;
        if ((lane & ((4 << k) - 1)) == 0 && qk < m) {
          uint32_t bits = packed >> 5;
          uint8_t* ps_k = L.ps + (2 * nparts - 2 * m);
          ps_k[qk] = (uint8_t)(packed & 31u);
          atomicAdd(&L.level_bits[k], (unsigned long long)bits);
          if (bits >= kMaxPToBits) atomicOr(&L.misc[kMiscSat], 1u << k);
        }
      }
      // hand the wave's order-(fo-kw) table to the cross-wave levels
      if (kw < fo && lane < 4 && c < rows) {
#pragma unroll
        for (int i = 0; i < 8; ++i) L.tables[(size_t)q0 * 32 + pq + 4u * (uint32_t)i] = tv[i];
      }
    }
    first_generic_level = kw + 1;
  } else {
    // PrcBitTable::from_errors(errs, 4), rice.rs:65-103, literally: u32 wrapping adds,
    // clamp after every 16 samples of the partition's slice and after the offset.
    for (int q = hw; q < nparts; q += NHW) {
      int start = q * psize;
      if (start < warm) start = warm;
      const int end = (q + 1) * psize;
      const int len = end - start;
      uint32_t accb = 0;
      // (16 at a time: the LDS reads of a run are independent of the running sum, so unrolled they
      // overlap instead of paying the LDS latency once per sample)
      int i = 0;
      for (; i + 16 <= len; i += 16) {
        uint32_t uu[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) uu[j] = zigzag(L.sbuf[sidx<ROWSTRIDE>(start + i + j)]);
#pragma unroll
        for (int j = 0; j < 16; ++j) accb += uu[j] >> p;
        accb = accb < kMaxPToBits ? accb : kMaxPToBits;  // i + 15 is the run's 16th sample
      }
      for (; i < len; ++i) {
        uint32_t u = zigzag(L.sbuf[sidx<ROWSTRIDE>(start + i)]);
        accb += u >> p;
      }
      accb = accb < kMaxPToBits ? accb : kMaxPToBits;
      uint32_t v = accb + (4u + (uint32_t)len * (p + 1u));
      v = v < kMaxPToBits ? v : kMaxPToBits;
      L.tables[(size_t)q * 32 + p] = v;
    }
  }
  if (BIG) __threadfence_block();
  __syncthreads();

  FLACENC_STAMP(5);
  // eval_partitions / merge_partitions over the remaining orders (rice.rs:193-216, 277-291).
  // Level k keeps its tables at indices q << k.
  for (int k = first_generic_level; k <= (finest_only ? 0 : fo); ++k) {
    const int m = nparts >> k;
    const int stride = 1 << k;
    uint8_t* ps_k = L.ps + (2 * nparts - 2 * m);  // level k owns m bytes at this offset
    for (int q = hw; q < m; q += NHW) {
      const size_t idx = (size_t)q * stride;
      uint32_t v = L.tables[idx * 32 + p];
      if (k > 0) {
        // PrcBitTable::merge(other, 4), rice.rs:144-152
        uint32_t o = L.tables[(idx + (stride >> 1)) * 32 + p];
        v = v + o - 4u;
        v = v < kMaxPToBits ? v : kMaxPToBits;
        L.tables[idx * 32 + p] = v;
      }
      // PrcBitTable::minimizer(max_p), rice.rs:115-141: min of (bits << 5) | p, ties -> smallest p
      uint32_t packed = (((p <= max_p) ? v : 0xFFFFFFFFu) << 5) | p;
#pragma unroll
      for (int msk = 1; msk < 32; msk <<= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)packed, msk, 64);
        packed = o < packed ? o : packed;
      }
      if (p == 0) {
        uint32_t bits = packed >> 5;
        ps_k[q] = (uint8_t)(packed & 31u);
        atomicAdd(&L.level_bits[k], (unsigned long long)bits);
        if (bits >= kMaxPToBits) atomicOr(&L.misc[kMiscSat], 1u << k);
      }
    }
    if (BIG) __threadfence_block();
    __syncthreads();
  }

  // pick the order: start at the finest, move to a coarser one only on strictly
  // fewer bits (rice.rs:285) -> among equal totals the finest wins.
  if (tid == 0) {
    int best = 0;
    unsigned long long best_bits = L.level_bits[0];
    for (int k = 1; k <= (finest_only ? 0 : fo); ++k) {
      if (L.level_bits[k] < best_bits) {
        best_bits = L.level_bits[k];
        best = k;
      }
    }
    L.misc[kMiscBestOrder] = (uint32_t)best;
  }
  __syncthreads();
  const int bestk = (int)L.misc[kMiscBestOrder];
  const int rice_order = fo - bestk;
  const int best_parts = nparts >> bestk;
  const uint8_t* best_ps = L.ps + (2 * nparts - 2 * best_parts);
  const int best_psize = n >> rice_order;
  const unsigned long long code_bits = L.level_bits[bestk];

  FLACENC_STAMP(6);
  // ======================= phase 5: outputs ================================
  int32_t* __restrict__ dst = a.residual + (size_t)sf * a.residual_stride;
  {
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) && ((a.residual_stride & 3) == 0);
    for (int q = tid; q < rows * 4; q += T) {
      int t = q * 4;
      int4 v = *reinterpret_cast<const int4*>(&L.sbuf[sidx<ROWSTRIDE>(t)]);
      if (vec_ok && t + 3 < n) {
        *reinterpret_cast<int4*>(dst + t) = v;
      } else {
        if (t + 0 < n) dst[t + 0] = v.x;
        if (t + 1 < n) dst[t + 1] = v.y;
        if (t + 2 < n) dst[t + 2] = v.z;
        if (t + 3 < n) dst[t + 3] = v.w;
      }
    }
  }

  // Residual::sum_quotients (datatype.rs:2325-2331).  When no selected table entry
  // saturated it follows from code_bits: code_bits = sum_q + 4*parts + (n - warm)
  // + sum_k p_k * len_k (rice.rs:69-71, 95-98).  Otherwise count it from the samples.
  unsigned long long sum_q = 0;
  // (the literal table sums are u32 wrapping adds, rice.rs:88-93: once a 16-sample run of quotients
  // can wrap -- zig-zag codes of 2^26 and more -- code_bits no longer determines the true sum either)
  const bool saturated = ((L.misc[kMiscSat] >> bestk) & 1u) != 0 || maxu >= (1u << 26);
  if (saturated) {
    unsigned long long mine = 0;
    for (int t = tid; t < n; t += T) {
      if (t >= warm) {
        uint32_t u = zigzag(L.sbuf[sidx<ROWSTRIDE>(t)]);
        mine += (unsigned long long)(u >> best_ps[t / best_psize]);
      }
    }
    __syncthreads();
    if (tid == 0) L.level_bits[15] = 0ull;
    __syncthreads();
    atomicAdd(&L.level_bits[15], mine);
    __syncthreads();
    sum_q = L.level_bits[15];
  }

  flacenc_hip_subframe_params* rec = a.params + sf;
  for (int i = tid; i < FLACENC_HIP_MAX_RICE_PARTITIONS; i += T)
    rec->rice_params[i] = (i < best_parts && status == 0) ? best_ps[i] : (uint8_t)0;
  if (tid < 32) rec->coefs[tid] = (status == 0) ? (int16_t)L.qc[tid] : (int16_t)0;
  if (tid == 0) {
    unsigned long long sum_p = 0;
    for (int i = 0; i < best_parts; ++i) sum_p += best_ps[i];
    if (!saturated) {
      sum_q = code_bits - 4ull * (unsigned long long)best_parts - (unsigned long long)(n - warm) -
              (sum_p * (unsigned long long)best_psize - (unsigned long long)warm * best_ps[0]);
    }
    // BitRepr for Residual::count_bits, bitrepr.rs:533-544
    bool rice2 = false;
    for (int i = 0; i < best_parts; ++i) rice2 |= best_ps[i] > 14;
    unsigned long long residual_bits = 2ull + 4ull + (unsigned long long)best_parts * (rice2 ? 5ull : 4ull) +
                                       (sum_q + (unsigned long long)(n - warm)) +
                                       (sum_p * (unsigned long long)best_psize -
                                        (unsigned long long)warm * best_ps[0]);
    // BitRepr for Lpc::count_bits, bitrepr.rs:492-499
    // side channel carries one extra bit (ChannelAssignment::bits_per_sample_offset, coding.rs:444)
    unsigned long long bps = a.bps ? (unsigned long long)a.bps[sf]
                                   : (unsigned long long)(a.bps_uniform + (role == 3 ? 1u : 0u));
    unsigned long long sub_bits = 8ull + bps * (unsigned long long)warm + 4ull + 5ull +
                                  (unsigned long long)a.precision * (unsigned long long)warm + residual_bits;
    // BitRepr for FixedLpc::count_bits, bitrepr.rs:473-477
    if (a.fixed_mode != 0) sub_bits = 8ull + bps * (unsigned long long)warm + residual_bits;
    // OrderSel::BitCount's key: bits_per_sample * order + code_bits (coding.rs:249)
    if (a.fixed_mode >= 2u && a.selector_keys) a.selector_keys[sf] = bps * (unsigned long long)warm + code_bits;
    rec->order = (uint8_t)warm;
    rec->shift = (int8_t)shift;
    rec->precision = (uint8_t)(a.fixed_mode != 0 ? 0u : a.precision);
    rec->rice_order = (uint8_t)(status == 0 ? rice_order : 0);
    rec->status = status;
    rec->code_bits = status == 0 ? code_bits : 0ull;
    rec->subframe_bits = status == 0 ? sub_bits : 0ull;
    rec->sum_quotients = status == 0 ? sum_q : 0ull;
  }
  FLACENC_STAMP(7);
}

// Levinson-Durbin + quantisation for a batch, one subframe per lane (second launch of the split):
// pred[sf] = {qc[32], order, shift, status, 0}.
template <int MAXP>
__global__ void __launch_bounds__(64) levinson_batch_kernel(QlpcKernelArgs a) {
  const uint32_t sf = blockIdx.x * 64u + threadIdx.x;
  if (sf >= a.n_subframes) return;
  double coef[MAXP];
  int32_t qc[MAXP];
  int order, shift;
  const int status = levinson_quantize<MAXP>(a.autocorr + (size_t)sf * 33, (int)a.lpc_order, (int)a.precision, coef, qc,
                                             &order, &shift);
  int32_t* pr = a.pred_out + (size_t)sf * 36;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) pr[i] = qc[i];
  for (int i = MAXP; i < 32; ++i) pr[i] = 0;
  pr[32] = order;
  pr[33] = shift;
  pr[34] = status;
  pr[35] = 0;
  if (a.lpc_coefs) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = (i < (int)a.lpc_order && status == 0) ? coef[i] : 0.0;
    for (int i = MAXP; i < 32; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = 0.0;
  }
}

template <int MAXP>
hipError_t launch_levinson_batch(const QlpcKernelArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(levinson_batch_kernel<MAXP>, dim3((a.n_subframes + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError();
}

template <int MAXP, bool BIG>
hipError_t launch_one(const QlpcKernelArgs& a, int threads, size_t smem, hipStream_t stream) {
  auto kern = qlpc_subframe_kernel<MAXP, BIG>;
  static DynamicLdsOptIn opt_in;  // per instantiation, per device inside; the attribute only ever grows
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(a.n_subframes), dim3(threads), smem, stream, a);
  return hipGetLastError();
}

}  // namespace
}  // namespace flacenc_hip
#endif
