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

// Sum of an f64 over the 64 lanes in the canonical order -- the balanced pairwise tree over the lane
// index -- by DPP row shifts and row broadcasts instead of six dependent LDS round trips
// (wave_butterfly_sum): after row_shr 1, 2, 4, 8 lane 15 of a row holds ((..)+(..)) of its row exactly as
// the butterfly pairs them (own + shifted = right + left, one IEEE add per tree node either way), then
// row_bcast15 folds rows 0/1 and 2/3, row_bcast31 the two halves; the total sits in lane 63 and is
// returned wave-uniform.  Lanes whose shifted source does not exist add +0.0 (row shifts: bound_ctrl) or
// whatever the destination held (row broadcasts into masked-off rows: no preset, the 14 moves per tree that
// an `old` operand of 0 costs are not spent) to a value nobody reads.
__device__ __forceinline__ double wave_tree_sum_dpp(double v) {
#define FLACENC_F64_DPP_STEP(CTRL, ROWMASK)                                                              \
  {                                                                                                      \
    const unsigned long long b_ = (unsigned long long)__double_as_longlong(v);                          \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b_, CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF);        \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b_ >> 32), CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF); \
    v = v + __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                    \
  }
  FLACENC_F64_DPP_STEP(0x111, 0xF)
  FLACENC_F64_DPP_STEP(0x112, 0xF)
  FLACENC_F64_DPP_STEP(0x114, 0xF)
  FLACENC_F64_DPP_STEP(0x118, 0xF)
  FLACENC_F64_DPP_STEP(0x142, 0xA)
  FLACENC_F64_DPP_STEP(0x143, 0xC)
#undef FLACENC_F64_DPP_STEP
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
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

// (x ^ m) + y (v_xad_u32).  With m = 0x7FFFFFFF it is y - x + (2^31 - 1) mod 2^32: the difference of two values
// that carry a common bias, itself biased by 2^31 - 1 -- one instruction per differencing step where
// (y - x) ^ 0x80000000 takes two.
__device__ __forceinline__ uint32_t xad_u32(uint32_t x, uint32_t m, uint32_t y) {
  uint32_t r;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(m), "v"(y));
  return r;
}
// |x - y| + acc on unsigned operands (v_sad_u32)
__device__ __forceinline__ uint32_t sad_u32(uint32_t x, uint32_t y, uint32_t acc) {
  uint32_t r;
  asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(acc));
  return r;
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
// ---- the order certificate ("default order") -------------------------------------------------------
// The kernels sum the autocorrelation in their own order, the reference in one sequential chain per lag (lpc.rs:533-548).
// With u = 2^-53, what the certificate rests on (oracle/flacenc_oracle.c, orc_quant_certified, states the same arithmetic
// operation for operation and says which steps are shown and which assumed):
//   eps   = (n + 96) u S >= |R^ - R~|: n - P roundings of the chain + at most 71 of the kernels' own orders, each at most u
//           times a partial sum of |products| <= S = R0 + P max|s|^2 / 2 (Cauchy-Schwarz);
//   F_i   = rowsum_i(|T^-1|) eps (1 + |a|_1): for T a = r, (T + E)(a + da) = r + g with |E_ij|, |g_i| <= eps the first-order
//           |da_i| (attained by low-pass material: T^-1 a sign checkerboard under an alternating a);
//   2 F_i : the factor 2 covers the second-order term (<= 0.23 F_i once the boundary test passes, quant_precision >= 6) and
//           the rounding of the two floating-point recursions (<= 0.77 F_i if their residuals obey c_L <= 11, measured
//           <= 0.39) -- ON SYSTEMS THE RECURSION FINDS POSITIVE DEFINITE.  The sums start at t = P for every lag, so R[]
//           need not be an autocorrelation (a block that opens on a clipped plateau is enough); on such systems the
//           recursion is unstable -- round 6's attack, tools/certificate_attack.py, found the two computed solutions 68 x
//           further apart than 2 F_i on a subframe the rule of round 5 certified -- and they are excluded (`nonpd`).
// The quantiser (lpc.rs:234-302) is a step function of a: if no a_i 2^shift comes within 2 F_i 2^shift of a rounding
// boundary k + 1/2, and max |a| -+ 2 F_i does not straddle a power of two (find_shift), the reference's own R[] quantises
// to the SAME QuantizedParameters -- the subframe is certified, and every integer output downstream is the reference's.  A
// subframe that is not certified is redone from the reference's chains.
// Tier 1 bounds every row sum of |T^-1| by the Gohberg-Semencul norm bound 2 |f|_1^2 / |f_0|, f = T^-1 e_0 = the
// recursion's `forward` vector -- O(P), enough for material that is not strongly tonal; tier 2 evaluates the rows
// themselves from f (T^-1_ij = T^-1_(i-1)(j-1) + (f_i f_j - f_(P-i) f_(P-j)) / f_0), O(P^2), ~70 x tighter.
constexpr double kCertSafety = 2.0;
constexpr int kCertOwnRoundings = 96;  // >= P + the roundings of the kernels' own sums (71: a 64-sample chain, six tree levels, the 4608 tail)

template <int MAXP>
__device__ __forceinline__ bool quant_stable(const double (&a)[MAXP], int P, int shift, const double (&da)[MAXP]) {
  double amax = 0.0, dmax = 0.0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i)
    if (i < P) {
      amax = fmax(amax, fabs(a[i]));
      dmax = fmax(dmax, da[i]);
    }
  const double lo = amax - dmax, hi = amax + dmax;
  bool ok = lo > 0.0 && hi < 1.0e300;  // (false for a NaN bound)
  if (ok) ok = ceil_log2_pos(lo) == ceil_log2_pos(hi);
  const double scalefac = (double)(1 << shift);
#pragma unroll
  for (int i = 0; i < MAXP; ++i)
    if (i < P) {
      const double v = fabs(a[i]) * scalefac;
      const double d = fabs((v - floor(v)) - 0.5);
      if (!(d > da[i] * scalefac)) ok = false;
    }
  return ok;
}

// tier 2: row sums of |T^-1| from the forward vector; out of line -- only strongly tonal material gets here
// The recursion proper (lpc.rs:657-703), shared by levinson_quantize and the certificate's second tier: a[] = the
// solution, fwd[] = the final `forward` vector, *skipped = a zero denominator skipped a step (lpc.rs:679-682).
template <int MAXP>
__device__ __forceinline__ void levinson_core(const double (&R)[MAXP + 1], int P, double (&a)[MAXP], double (&fwd)[MAXP],
                                              bool* skipped, bool* nonpd = nullptr) {
  fwd[0] = 1.0 / R[0];   // Float::recip(coefs[0] + diagonal_loading), loading = 0
  a[0] = R[1] / R[0];    // ys[0] / (coefs[0] + diagonal_loading)
#pragma unroll
  for (int n = 1; n < MAXP; ++n) {
    if (n < P) {
      double err = 0.0;
#pragma unroll
      for (int d = 0; d < n; ++d) err = __builtin_fma(R[n - d], fwd[d], err);
      double denom = __builtin_fma(err, -err, 1.0);
      if (denom == 0.0) *skipped = true;
      // (the certificate: a denominator that is not positive = the Toeplitz matrix is not positive definite -- the sums
      // start at t = P for every lag, R[] need not be an autocorrelation -- and the recursion is outside its stable domain)
      if (nonpd != nullptr && !(denom > 0.0)) *nonpd = true;
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
}

// tier 2: row sums of |T^-1| from the forward vector.  Out of line and self-contained -- only strongly tonal material gets
// here, and nothing of it may lengthen a live range on the common path: it reloads R[] from memory (`r_mem`: where the
// caller took it from) and runs the recursion again for a[] and the forward vector (identical bits: the same code).
template <int MAXP>
__device__ __attribute__((noinline)) bool quant_certified_rows(const double* r_mem, int P, int precision, uint32_t max_abs_s,
                                                               int n_sum) {
  double R[MAXP + 1], a[MAXP], fwd[MAXP];
#pragma unroll
  for (int i = 0; i <= MAXP; ++i) R[i] = (i <= P) ? r_mem[i] : 0.0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) a[i] = fwd[i] = 0.0;
  bool skipped = false;
  levinson_core<MAXP>(R, P, a, fwd, &skipped);
  // shift and eps_a as the first tier has them (find_shift, lpc.rs:234-254)
  double a1 = 0.0, amax = 0.0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    a1 += fabs(a[i]);
    amax = fmax(amax, fabs(a[i]));
  }
  int abs_log2 = ceil_log2_pos(amax);
  if (abs_log2 < -32752) abs_log2 = -32752;
  int shift = (precision - 1) - abs_log2;
  shift = shift < 0 ? 0 : (shift > 15 ? 15 : shift);
  const double m = (double)max_abs_s;
  const double S = R[0] + (0.5 * (double)P) * (m * m);
  const double eps = ((double)(n_sum + kCertOwnRoundings) * 0x1p-53) * S;
  const double eps_a = eps * (1.0 + a1);
  double z[MAXP];  // z[i] = fwd[P - i], i >= 1
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) v = (k == P - i) ? fwd[k] : v;
    z[i] = v;
  }
  const double inv_f0 = 1.0 / fwd[0];
  double row[MAXP], da[MAXP];
  {
    double rs = 0.0;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      row[j] = j < P ? fwd[j] : 0.0;
      rs += fabs(row[j]);
    }
    da[0] = (kCertSafety * rs) * eps_a;
  }
#pragma unroll
  for (int i = 1; i < MAXP; ++i) {
    double rs = 0.0;
    if (i < P) {
#pragma unroll
      for (int j = MAXP - 1; j >= 1; --j) {
        const double t = (fwd[i] * fwd[j] - z[i] * z[j]) * inv_f0;
        row[j] = j < P ? row[j - 1] + t : 0.0;
      }
      row[0] = fwd[i];
#pragma unroll
      for (int j = 0; j < MAXP; ++j) rs += fabs(row[j]);
    }
    da[i] = (kCertSafety * rs) * eps_a;
  }
  return quant_stable<MAXP>(a, P, shift, da);
}

// The certificate proper, out of line: inlined behind the recursion its mere presence -- skipped or not -- cost the
// kernel 6 % (47 more register copies and a different schedule of the recursion in front of it: the recursion is wave 0's
// serial path, every cycle of it is the workgroup's).  Operands by value: a[] and a handful of scalars.
// Returns bit 0: certified, bit 1: the rows of T^-1 were needed.
template <int MAXP>
struct CertArgs {
  double a[MAXP];
  double f1, f0, r0;    // |forward|_1, |forward[0]|, R[0]
  uint32_t max_abs_s;
  int n_sum, P, shift;
};
#ifdef FLACENC_CERT_T1_OUTLINE
#define FLACENC_CERT_T1_ATTR __attribute__((noinline))
#else
#define FLACENC_CERT_T1_ATTR __forceinline__
#endif
template <int MAXP>
__device__ FLACENC_CERT_T1_ATTR int quant_certified(const CertArgs<MAXP>& in) {
  // (entries above P are +0.0 in a[]: no masks -- they add nothing, and a zero coefficient sits 0.5 from its boundary,
  // further than any bound under which a real coefficient could pass)
  double a1 = 0.0, amax = 0.0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    a1 += fabs(in.a[i]);
    amax = fmax(amax, fabs(in.a[i]));
  }
  const double m = (double)in.max_abs_s;
  const double S = in.r0 + (0.5 * (double)in.P) * (m * m);
  const double eps = ((double)(in.n_sum + kCertOwnRoundings) * 0x1p-53) * S;
  const double eps_a = eps * (1.0 + a1);
  // tier 1: |da_i| <= kCertSafety (2 f1^2 / |f0|) eps_a for every i; everything is compared multiplied through by |f0|
  // (no division): num = |da| |f0|
  const double f1 = in.f1, f0 = in.f0;
  const double num = ((kCertSafety * 2.0) * (f1 * f1)) * eps_a;
  // find_shift: max |a| -+ |da| must stay inside (2^(e-1), 2^e], e = ceil(log2(max |a|)) (both gaps are exact)
  const int e = ceil_log2_pos(amax);
  const double g_lo = amax - ldexp(1.0, e - 1), g_hi = ldexp(1.0, e) - amax;
  bool ok = amax > 0.0 && num < g_lo * f0 && num < g_hi * f0;
  const double scalefac = (double)(1 << in.shift);
  const double nums = num * scalefac;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const double v = fabs(in.a[i]) * scalefac;
    const double d = fabs(__builtin_amdgcn_fract(v) - 0.5);  // (v - floor(v): v_fract_f64; v < 2^15, no clamp case)
    if (!(d * f0 > nums)) ok = false;
  }
  return ok ? 1 : 2;  // 2: not certified by this tier -- the caller asks quant_certified_rows
}

// CERT: *certified_out = the certificate above holds: the quantised parameters are those of any R[] within the summation bound of Rl
// (max_abs_s = the subframe's max |s|, n = samples summed per lag); tier2_out counts evaluations of the rows (statistics)
template <int MAXP, bool CERT = false>
__device__ int levinson_quantize(const double* __restrict__ Rl, int P, int precision,
                                 double (&a)[MAXP], int32_t* qc_out, int* order_out,
                                 int* shift_out, uint32_t max_abs_s = 0, int n_sum = 0, bool* certified_out = nullptr,
                                 bool* tier2_out = nullptr, bool do_cert = true) {
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

  bool skipped = false;  // a zero denominator skipped a step: forward[] is not T^-1 e_0 any more
  bool nonpd = false;    // a denominator was not positive: no certificate for this system
  double cert_f1 = 0.0, cert_f0 = 0.0;  // |forward|_1 and |forward[0]|: all the certificate's first tier needs of it
  if (status == 0 && R[0] != 0.0) {
    double fwd[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) fwd[i] = 0.0;
    levinson_core<MAXP>(R, P, a, fwd, &skipped, CERT ? &nonpd : nullptr);
    if (CERT) {
      // (entries above P are +0.0: no masks)
#pragma unroll
      for (int i = 0; i < MAXP; ++i) cert_f1 += fabs(fwd[i]);
      cert_f0 = fabs(fwd[0]);
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
  if (CERT && !do_cert) {
    *certified_out = true;
    if (tier2_out) *tier2_out = false;
  }
#ifdef FLACENC_CERT_NOBLOCK
  if (CERT) *certified_out = true;
#else
  if (CERT && do_cert) {
    // digital silence (R[0] == 0: every product is an exact zero in either order) is certified as it is; a status the
    // reference would panic on, a skipped step or all-zero coefficients are left to the reference's own chains
    bool certified = status == 0 && R[0] == 0.0;
    bool tier2 = false;
    if (status == 0 && R[0] != 0.0 && !skipped && !nonpd) {
      CertArgs<MAXP> ca;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) ca.a[i] = a[i];
      ca.f1 = cert_f1;
      ca.f0 = cert_f0;
      ca.r0 = R[0];
      ca.max_abs_s = max_abs_s;
      ca.n_sum = n_sum;
      ca.P = P;
      ca.shift = shift;
      const int fl = quant_certified<MAXP>(ca);
      certified = (fl & 1) != 0;
      tier2 = (fl & 2) != 0;  // the first tier could not decide: the caller runs quant_certified_rows (out of line,
                              // NOT from here: a call site inside this function costs the common path 1.3 %)
    }
    *certified_out = certified;
    if (tier2_out) *tier2_out = tier2;
  }
#endif
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
#define FLACENC_BODY_BLOCK blockIdx.x
#define FLACENC_BODY_DIRECT false
#include "qlpc_kernel_body.inc"
#undef FLACENC_BODY_BLOCK
#undef FLACENC_BODY_DIRECT
}

// Clean-up launch behind the big-block kernels: only the subframes they marked (status -1: residuals of 2^26
// and more, whose bit tables need the literal chunk-clamped sums) are redone.  A handful of workgroups, each
// looking at blockDim.x consecutive records at once -- normally none is marked and the launch costs what an
// empty kernel costs instead of one workgroup, with its LDS allocation, per subframe.  The body is a real
// call here (inlined into a loop it does not finish compiling).
template <int MAXP, bool BIG>
__device__ __attribute__((noinline)) void qlpc_subframe_call(const QlpcKernelArgs& a, const uint32_t sf_direct) {
#define FLACENC_BODY_BLOCK sf_direct
#define FLACENC_BODY_DIRECT true
#include "qlpc_kernel_body.inc"
#undef FLACENC_BODY_BLOCK
#undef FLACENC_BODY_DIRECT
}
// (launched with the generic kernel's own workgroup size for the bucket: up to 1024 threads at orders <= 12)
template <int MAXP, bool BIG>
__global__ void __launch_bounds__(MAXP <= 12 ? 1024 : (MAXP <= 16 ? 512 : 256)) qlpc_marked_kernel(QlpcKernelArgs a) {
  if (a.marked_count != nullptr) {
    // nothing marked (the usual case): no scan of the records.  Two counters take turns from pipeline to pipeline
    // (QlpcKernelArgs::marked_count / marked_next): this launch clears the one the NEXT pipeline's residual kernel will
    // count into.  (Tickets on one counter -- the last workgroup to arrive clears it -- cost 96 same-address
    // device-scope atomics from all eight XCDs: 21 us for a launch that has nothing to do.)
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.marked_next = 0u;
    const uint32_t count = __hip_atomic_load(a.marked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count == 0u) return;
    if (a.marked_list != nullptr && count <= a.marked_cap) {
      // the marks' own list (QlpcKernelArgs::marked_list): no scan.  Everything here is workgroup-uniform.
      const uint32_t unit = a.marked_unit;
      for (uint32_t i = blockIdx.x; i < count * unit; i += gridDim.x) {
        const uint32_t sf = a.marked_list[i / unit] * unit + i % unit;
        if (sf >= a.n_subframes) continue;
        const int st = a.params[sf].status;
        if (st != -1 && st != -2) continue;
        qlpc_subframe_call<MAXP, BIG>(a, sf);
        __syncthreads();
      }
      return;
    }
  }
  // (a small grid walks the records in strides: the usual launch finds the count at 0 and is over at once)
  // Round 6: the marked records of a window are found by wave ballots and visited one by one -- the window used to be
  // walked record by record with a barrier each (1024 barriers for one marked record: 40 us of a launch that, with the
  // order certificate on the sub-wave shapes, now normally has a handful of records to redo).
  __shared__ unsigned long long marked_ballot[16];
  const int mw = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  for (uint32_t base = blockIdx.x * blockDim.x; base < a.n_subframes; base += gridDim.x * blockDim.x) {
    const uint32_t mine = base + threadIdx.x;
    // (-1: beyond the marking kernel's exact sums; -2: not passed by the sub-wave kernel's order certificate)
    const bool marked = mine < a.n_subframes && (a.params[mine].status == -1 || a.params[mine].status == -2);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(marked);
    if ((threadIdx.x & 63) == 0) marked_ballot[mw] = bal;
    if (!__syncthreads_or(marked ? 1 : 0)) continue;
    for (int w = 0; w < nw; ++w) {
      unsigned long long b = marked_ballot[w];  // (workgroup-uniform: every thread walks the same bits)
      while (b != 0ull) {
        const int bit = __builtin_ctzll(b);
        b &= b - 1ull;
        qlpc_subframe_call<MAXP, BIG>(a, base + (uint32_t)(64 * w + bit));
        __syncthreads();
      }
    }
    __syncthreads();  // (the ballots are rewritten by the next window)
  }
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
  // the order selector's sums are carved only for the launch that selects (plan_qlpc_launch counts them always): a
  // 1152-sample block's workgroup drops from 10 LDS granules to 8, 12 -> 16 workgroups per CU
  if (a.fixed_mode != 1u) smem -= (size_t)kFixedSumWords * 8;
  if constexpr (MAXP >= 24 || MAXP <= 12) {  // (the order-16 bucket's body does not finish compiling out of line: there the
                                              // marked subframes are found by one workgroup per subframe, below)
    if (a.only_marked) {
      auto mk = qlpc_marked_kernel<MAXP, BIG>;
      static DynamicLdsOptIn opt_in_marked;
      if (hipError_t err = opt_in_marked.ensure(reinterpret_cast<const void*>(mk), smem); err != hipSuccess) return err;
      uint32_t grid = (a.n_subframes + (uint32_t)threads - 1u) / (uint32_t)threads;
      if (grid > 96u) grid = 96u;  // (the kernel walks the records in grid strides)
      hipLaunchKernelGGL(mk, dim3(grid), dim3(threads), smem, stream, a);
      return hipGetLastError();
    }
  }
  auto kern = qlpc_subframe_kernel<MAXP, BIG>;
  static DynamicLdsOptIn opt_in;  // per instantiation, per device inside; the attribute only ever grows
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(a.n_subframes), dim3(threads), smem, stream, a);
  if (hipError_t err = hipGetLastError(); err != hipSuccess) return err;
  // (a clean-up launch that is not qlpc_marked_kernel: the count of marked subframes is cleared behind it)
  if (a.only_marked && a.marked_count != nullptr) {
    if (hipError_t err = hipMemsetAsync(a.marked_count, 0, 4, stream); err != hipSuccess) return err;
    return hipMemsetAsync(a.marked_next, 0, 4, stream);
  }
  return hipSuccess;
}

}  // namespace
}  // namespace flacenc_hip
#endif
