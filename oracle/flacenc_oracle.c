/*
 * flacenc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See flacenc_oracle.h.
 *
 * CPU restatement (plain C99) of flacenc-rs v0.5.1's quantised-LPC analysis
 * path.  Every function cites the reference file:line it follows.  Floating
 * point follows the reference operation-for-operation (explicit fma() where
 * the reference calls mul_add; build with -ffp-contract=off so nothing else
 * is fused; glibc cosf/log2/round are the libm Rust-on-Linux calls too).
 *
 * Parity pin: reference KATs only (tests/test_oracle_kat.py); the Rust
 * reference cannot be built in this image (no rustc/cargo) -> coefficient-level
 * parity on arbitrary audio is "parity unpinned".
 */
#define _GNU_SOURCE
#include "flacenc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* per-thread scratch (the role of the reference's `reusable!` thread-locals, */
/* src/lib.rs:92-116: WINDOW_CACHE lpc.rs:219, LPC_ESTIMATOR :916,           */
/* QLPC_ERROR_BUFFER coding.rs:353, PRC_FINDER rice.rs:301)                  */
/* ------------------------------------------------------------------------ */
typedef struct {
  float* window;  /* cached window_weights(type, alpha, n) */
  size_t window_n;
  uint32_t window_type;
  float window_alpha;
  float* xw;
  size_t xw_cap;
  uint32_t* uerr;
  size_t uerr_cap;
  uint32_t (*tables)[32];
  size_t tables_cap;
  uint8_t* ps;
  uint8_t* min_ps;
  size_t ps_cap;
  int64_t* acc64;
  size_t acc64_cap;
  double* partial;
  size_t partial_cap;
} orc_scratch;

static __thread orc_scratch orc_tls;

#define ORC_GROW(ptr, cap, need, type)                          \
  do {                                                          \
    if ((cap) < (size_t)(need)) {                               \
      free(ptr);                                                \
      (cap) = (size_t)(need) + 64;                              \
      (ptr) = (type*)malloc(sizeof(type) * (cap));              \
    }                                                           \
  } while (0)

/* ------------------------------------------------------------------------ */
/* src/lpc.rs                                                               */
/* ------------------------------------------------------------------------ */

/* window_weights, src/lpc.rs:96-120.  All arithmetic in f32, expression order
 * exactly as written there; `Tukey{alpha: 0.0}` short-circuits to all-ones. */
void orc_window_weights(uint32_t window_type, float alpha, size_t len, float* out) {
  if (window_type == ORC_WINDOW_RECTANGLE || alpha == 0.0f) {
    for (size_t t = 0; t < len; ++t) out[t] = 1.0f;
    return;
  }
  const float pi = 3.14159265358979323846f; /* std::f32::consts::PI */
  float max_t = (float)len - 1.0f;
  float alpha_len = alpha * max_t;
  for (size_t ti = 0; ti < len; ++ti) {
    float t = (float)ti;
    float w;
    if (t < alpha_len / 2.0f) {
      float arg = 2.0f * pi * t / alpha_len;
      w = 0.5f * (1.0f - cosf(arg));
    } else if (t < max_t - alpha_len / 2.0f) {
      w = 1.0f;
    } else {
      float arg = 2.0f * pi * (max_t - t) / alpha_len;
      w = 0.5f * (1.0f - cosf(arg));
    }
    out[ti] = w;
  }
}

/* LpcEstimator::fill_windowed_signal, src/lpc.rs:739-756: i32 -> f32 cast
 * (round-to-nearest-even), one f32 multiply. */
void orc_fill_windowed_signal(const int32_t* signal, const float* window, size_t n, float* out) {
  for (size_t t = 0; t < n; ++t) out[t] = (float)signal[t] * window[t];
}

/* weighted_auto_correlation (src/lpc.rs:551-564) -> _nosimd (src/lpc.rs:533-548)
 * with NoWeight, T = f64: for t in (order-1)..n, for tau < order:
 * dest[tau] = mul_add(signal[t - tau] as f64, signal[t] as f64, dest[tau]). */
void orc_auto_correlation_f64(size_t order, const float* signal, size_t n, double* dest) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0;
  if (order == 0) return;
  for (size_t t = order - 1; t < n; ++t) {
    double wy = (double)signal[t];
    for (size_t tau = 0; tau < order; ++tau) {
      dest[tau] = fma((double)signal[t - tau], wy, dest[tau]);
    }
  }
}

/* same with T = f32 (used by the reference's tests, src/lpc.rs:998-1022) */
void orc_auto_correlation_f32(size_t order, const float* signal, size_t n, float* dest) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0f;
  if (order == 0) return;
  for (size_t t = order - 1; t < n; ++t) {
    float wy = signal[t];
    for (size_t tau = 0; tau < order; ++tau) {
      dest[tau] = fmaf(signal[t - tau], wy, dest[tau]);
    }
  }
}

/* The build's canonical summation order (NOT a reference function): the same
 * sum as src/lpc.rs:533-548 (common lower bound t = order-1 for every lag),
 * re-associated so that a GPU can evaluate it in parallel and a CPU can
 * reproduce it bit-for-bit:
 *   1. samples are cut into 16-sample chunks [16c, 16c+16);
 *   2. each chunk's partial sum is a sequential fma chain in t order starting
 *      from +0.0 (only t with order-1 <= t < n contribute);
 *   3. chunk partials are combined by a perfectly balanced pairwise tree over
 *      the chunk index, zero-padded to a power of two.
 * Every f32*f32 product is exact in f64, so only the additions round. */
static double orc_tree_sum(const double* v, size_t lo, size_t hi, size_t count) {
  if (hi - lo == 1) return lo < count ? v[lo] : 0.0;
  size_t mid = lo + (hi - lo) / 2;
  return orc_tree_sum(v, lo, mid, count) + orc_tree_sum(v, mid, hi, count);
}

/* Where the build's UNFLAGGED order is the reference's own: on blocks of 4096 / 8192 / 16384 samples at LPC orders from
 * 16 the product sums as weighted_auto_correlation_nosimd does (src/lpc.rs:533-548: one sequential fma chain per
 * lag; on the GPU, v_mfma_f64_4x4x4 chains) -- there the stable build's order costs no more than the chunk tree, so
 * ORC_ACORR_CANONICAL, "what the product computes without a summation-order flag", is ORC_ACORR_REFERENCE. */
int orc_default_order_is_stable(size_t n, size_t lpc_order) {
  return (n == 4096 || n == 8192 || n == 16384) && lpc_order >= 16;
}

/* ... and on blocks of 4096 / 4608 samples at LPC orders up to 12 (the fused kernel's shapes) the unflagged product
 * CERTIFIES its own sums: it keeps them where a perturbation bound on the Toeplitz solve (orc_quant_certified below:
 * what of it is shown and what is assumed) says the quantised parameters equal those of the reference's own chains, and
 * recomputes the subframe from those chains (ORC_ACORR_REFERENCE) where it does not -- so that every integer output of
 * the default mode is the stable build's. */
/* statistics of the certified mode since the last reset (not thread-safe: tests read them after single-threaded runs):
 * [0] subframes analysed, [1] certificates that needed the rows of T^-1, [2] subframes recomputed in the reference's order */
unsigned long orc_cert_stats[3] = {0, 0, 0};

int orc_default_order_is_certified(size_t n, size_t lpc_order) {
  return (n == 4096 || n == 4608) && lpc_order >= 1 && lpc_order <= 12;
}

/* ... and (round 6) on EVERY other shape the unflagged product runs the reference's chains for every subframe in a pass of
 * their own in front of the kernel that takes the shape (ORC_ACORR_CANONICAL is ORC_ACORR_REFERENCE there); only
 * ORC_ACORR_CHUNK_TREE, the product's FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER, keeps the one-pass chunk tree on them. */
int orc_default_order_is_two_pass(size_t n, size_t lpc_order) {
  return lpc_order >= 1 && !orc_default_order_is_certified(n, lpc_order) && !orc_default_order_is_stable(n, lpc_order);
}

/* The fused kernel's own summation order on blocks of 4096 / 4608 samples (flacenc_rs_amd/csrc/qlpc_wave_kernel_impl.h,
 * lane_order_reduce; NOT reference code): lane l = samples [64 l, 64 l + 64) is one fma chain per lag (started with the
 * literal +0.0; terms below t = order - 1 are fma(0, ., acc) = acc), the 64 lane sums v[l] meet as
 *   w[i] = (v[i] + v[i + 32]) + (v[i + 16] + v[i + 48]),  i = 0 .. 15,
 *   R = (((w15 + w14) + (w13 + w12)) + ((w11 + w10) + (w9 + w8))) + (((w7 + w6) + (w5 + w4)) + ((w3 + w2) + (w1 + w0))),
 * and a 4608-sample block adds the same tree over its last 512 samples, 16 per lane on lanes 0..31 (+0.0 above). */
static double orc_lane_tree(const double* v) {
  double w[16];
  for (int i = 0; i < 16; ++i) w[i] = (v[i] + v[i + 32]) + (v[i + 16] + v[i + 48]);
  for (int d = 1; d < 16; d <<= 1)
    for (int i = 15; i >= d; i -= 2 * d) w[i] = w[i] + w[i - d];
  return w[15];
}

void orc_auto_correlation_lane_order_f64(size_t order, const float* signal, size_t n, double* dest) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0;
  if (order == 0 || (n != 4096 && n != 4608)) return;
  size_t P = order - 1;
  for (size_t tau = 0; tau < order; ++tau) {
    double v[64];
    for (size_t l = 0; l < 64; ++l) {
      double acc = 0.0;
      for (size_t t = 64 * l; t < 64 * l + 64; ++t) {
        double cur = t >= P ? (double)signal[t] : 0.0;
        double lagged = t >= tau ? (double)signal[t - tau] : 0.0;
        acc = fma(cur, lagged, acc);
      }
      v[l] = acc;
    }
    double r = orc_lane_tree(v);
    if (n == 4608) {
      for (size_t l = 0; l < 64; ++l) {
        double acc = 0.0;
        if (l < 32)
          for (size_t t = 4096 + 16 * l; t < 4096 + 16 * l + 16; ++t) acc = fma((double)signal[t], (double)signal[t - tau], acc);
        v[l] = acc;
      }
      r = r + orc_lane_tree(v);
    }
    dest[tau] = r;
  }
}

void orc_auto_correlation_canonical_f64(size_t order, const float* signal, size_t n,
                                        double* dest) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0;
  if (order == 0 || n == 0) return;
  size_t nchunks = (n + 15) / 16;
  size_t pow2 = 1;
  while (pow2 < nchunks) pow2 <<= 1;
  ORC_GROW(orc_tls.partial, orc_tls.partial_cap, nchunks, double);
  double* partial = orc_tls.partial;
  for (size_t tau = 0; tau < order; ++tau) {
    for (size_t c = 0; c < nchunks; ++c) {
      double acc = 0.0;
      for (size_t t = 16 * c; t < 16 * c + 16 && t < n; ++t) {
        if (t + 1 < order) continue; /* t < order - 1 */
        acc = fma((double)signal[t - tau], (double)signal[t], acc);
      }
      partial[c] = acc;
    }
    dest[tau] = orc_tree_sum(partial, 0, pow2, nchunks);
  }
}

/* weighted_auto_correlation_simd (src/lpc.rs:510-531) -> weighted_delay_prod_sum_impl
 * (src/lpc.rs:439-500), the `simd-nightly` build's summation order, with NoWeight, T = f64.
 * Per lag DELAY: LANES = next_power_of_two((DELAY | 7) - 1) = 8 / 16 / 32 / 32 / 64 f64 lanes;
 * signal[warm_up..] is split by `as_simd` into an unaligned scalar head, LANES-wide body
 * vectors and a scalar foot; head and foot are sequential mul_add chains from 0, each body
 * lane is its own mul_add chain, and the lanes are added left to right (reduce_sum is
 * simd_reduce_add_ordered) before  acc(head) + acc(foot) + lanes.
 * The head length depends on the address of signal[warm_up] modulo the vector alignment:
 * `base_mod` is the element offset of signal[0] from a LANES*4-byte boundary (0 for the
 * 64-byte aligned SimdVec when LANES <= 16; allocator-dependent beyond that -- which is why the
 * reference's own simd/nosimd parity test only asserts rtol 1e-5, src/lpc.rs:1392-1413). */
void orc_auto_correlation_nightly_f64(size_t order, const float* signal, size_t n, double* dest,
                                      size_t base_mod) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0;
  if (order == 0 || n + 1 <= order) return;
  size_t warm_up = order - 1;
  for (size_t delay = 0; delay < order; ++delay) {
    size_t x = (delay | 7) - 1, lanes = 1;
    while (lanes < x) lanes <<= 1;
    if (lanes > 64) lanes = 64;
    size_t len = n - warm_up;
    size_t mis = (base_mod + warm_up) % lanes;
    size_t head = mis ? lanes - mis : 0;
    if (head > len) head = len;
    size_t nbody = (len - head) / lanes;
    size_t t = warm_up;
    double acc = 0.0;
    { /* head */
      double a = 0.0;
      for (size_t i = 0; i < head; ++i, ++t) a = fma((double)signal[t - delay], (double)signal[t], a);
      acc += a;
    }
    double acc_v[64];
    for (size_t l = 0; l < lanes; ++l) acc_v[l] = 0.0;
    for (size_t b = 0; b < nbody; ++b, t += lanes)
      for (size_t l = 0; l < lanes; ++l)
        acc_v[l] = fma((double)signal[t + l], (double)signal[t + l - delay], acc_v[l]);
    { /* foot */
      double a = 0.0;
      for (; t < n; ++t) a = fma((double)signal[t - delay], (double)signal[t], a);
      acc += a;
    }
    double lanesum = 0.0;
    for (size_t l = 0; l < lanes; ++l) lanesum += acc_v[l];
    dest[delay] = acc + lanesum;
  }
}

/* symmetric_levinson_recursion, src/lpc.rs:633-705.
 * NOTE on `continue` (src/lpc.rs:679-682): it sits inside `for n in 1..order`
 * nested in `loop { .. break; }`; an unlabeled `continue` continues the
 * innermost loop, i.e. the `for`.  So a zero denominator updates
 * `diagonal_loading` (never read again, forward[0]/dest[0] were computed
 * before the `for`) and SKIPS the rest of iteration n -- it does not restart.
 * This restatement follows the code as compiled, not the comment. */
#define ORC_DEFINE_LEVINSON(NAME, T, FMA, ONE, ZERO)                                \
  int NAME(const T* coefs, const T* ys, size_t order, T* dest) {                    \
    T forward[ORC_MAX_LPC_ORDER + 1];                                               \
    T forward_next[ORC_MAX_LPC_ORDER + 1];                                          \
    for (size_t i = 0; i < order; ++i) dest[i] = ZERO;                              \
    if (order == 0) return ORC_STATUS_OK;                                           \
    if (!(coefs[0] >= ZERO)) return ORC_STATUS_NEG_ENERGY; /* assert, :646 */       \
    if (coefs[0] == ZERO) {                                                         \
      int allzero = 1;                                                              \
      for (size_t i = 0; i < order; ++i) allzero &= (ys[i] == ZERO) & (coefs[i] == ZERO); \
      return allzero ? ORC_STATUS_OK : ORC_STATUS_NEG_ENERGY; /* assert, :652 */    \
    }                                                                               \
    for (size_t i = 0; i <= ORC_MAX_LPC_ORDER; ++i) forward[i] = forward_next[i] = ZERO; \
    T diagonal_loading = ZERO;                                                      \
    forward[0] = ONE / (coefs[0] + diagonal_loading); /* Float::recip */            \
    dest[0] = ys[0] / (coefs[0] + diagonal_loading);                                \
    for (size_t n = 1; n < order; ++n) {                                            \
      T error = ZERO;                                                               \
      for (size_t d = 0; d < n; ++d) error = FMA(coefs[n - d], forward[d], error);  \
      T denom = FMA(error, -error, ONE);                                            \
      if (denom == ZERO) {                                                          \
        T dd = diagonal_loading + diagonal_loading;                                 \
        diagonal_loading = (ONE > dd) ? ONE : dd;                                   \
        continue;                                                                   \
      }                                                                             \
      T alpha = ONE / denom;                                                        \
      T beta = -alpha * error;                                                      \
      for (size_t d = 0; d <= n; ++d) {                                             \
        T bf = beta * forward[n - d];                                               \
        forward_next[d] = FMA(alpha, forward[d], bf);                               \
      }                                                                             \
      for (size_t d = 0; d <= n; ++d) forward[d] = forward_next[d];                 \
      T delta = ZERO;                                                               \
      for (size_t d = 0; d < n; ++d) delta = FMA(coefs[n - d], dest[d], delta);     \
      T resid = ys[n] - delta;                                                      \
      for (size_t d = 0; d <= n; ++d) dest[d] = FMA(resid, forward[n - d], dest[d]);\
    }                                                                               \
    (void)diagonal_loading;                                                         \
    return ORC_STATUS_OK;                                                           \
  }

ORC_DEFINE_LEVINSON(orc_symmetric_levinson_f64, double, fma, 1.0, 0.0)
ORC_DEFINE_LEVINSON(orc_symmetric_levinson_f32, float, fmaf, 1.0f, 0.0f)

/* The same recursion (f64) handing out what the order certificate needs: the final `forward` vector (T^-1 e_0 of the
 * order x order Toeplitz system) and whether a zero denominator skipped a step (then it is not).  NOT reference code:
 * the product's certificate (flacenc_rs_amd/csrc/qlpc_kernel_impl.h, levinson_quantize<.., CERT>) restated. */
static int orc_levinson_f64_forward(const double* coefs, const double* ys, size_t order, double* dest, double* forward,
                                    int* skipped) {
  /* *skipped: bit 0 = a zero denominator skipped a step (lpc.rs:679-682), bit 1 = a denominator was not positive -- the
   * Toeplitz matrix is not positive definite (the sums start at t = P for every lag: R[] need not be an autocorrelation)
   * and the recursion is outside the domain where it is stable.  Either way the certificate does not apply. */
  double forward_next[ORC_MAX_LPC_ORDER + 1];
  *skipped = 0;
  for (size_t i = 0; i < order; ++i) dest[i] = 0.0;
  for (size_t i = 0; i <= ORC_MAX_LPC_ORDER; ++i) forward[i] = forward_next[i] = 0.0;
  if (order == 0) return ORC_STATUS_OK;
  if (!(coefs[0] >= 0.0)) return ORC_STATUS_NEG_ENERGY;
  if (coefs[0] == 0.0) {
    int allzero = 1;
    for (size_t i = 0; i < order; ++i) allzero &= (ys[i] == 0.0) & (coefs[i] == 0.0);
    return allzero ? ORC_STATUS_OK : ORC_STATUS_NEG_ENERGY;
  }
  forward[0] = 1.0 / coefs[0];
  dest[0] = ys[0] / coefs[0];
  for (size_t n = 1; n < order; ++n) {
    double error = 0.0;
    for (size_t d = 0; d < n; ++d) error = fma(coefs[n - d], forward[d], error);
    double denom = fma(error, -error, 1.0);
    if (!(denom > 0.0)) *skipped |= 2;
    if (denom == 0.0) {
      *skipped |= 1;
      continue;
    }
    double alpha = 1.0 / denom;
    double beta = -alpha * error;
    for (size_t d = 0; d <= n; ++d) forward_next[d] = fma(alpha, forward[d], beta * forward[n - d]);
    for (size_t d = 0; d <= n; ++d) forward[d] = forward_next[d];
    double delta = 0.0;
    for (size_t d = 0; d < n; ++d) delta = fma(coefs[n - d], dest[d], delta);
    double resid = ys[n] - delta;
    for (size_t d = 0; d <= n; ++d) dest[d] = fma(resid, forward[n - d], dest[d]);
  }
  return ORC_STATUS_OK;
}

/* exact ceil(log2(m)) for finite m > 0 (the product's ceil_log2_pos) */
static int orc_ceil_log2_pos(double m) {
  uint64_t b;
  memcpy(&b, &m, 8);
  int e = (int)((b >> 52) & 0x7FF);
  uint64_t frac = b & 0xFFFFFFFFFFFFFull;
  if (e == 0) return -32752;
  return (e - 1023) + (frac != 0 ? 1 : 0);
}

static int orc_quant_stable(const double* a, size_t P, int32_t shift, const double* da) {
  double amax = 0.0, dmax = 0.0;
  for (size_t i = 0; i < P; ++i) {
    amax = fmax(amax, fabs(a[i]));
    dmax = fmax(dmax, da[i]);
  }
  double lo = amax - dmax, hi = amax + dmax;
  int ok = lo > 0.0 && hi < 1.0e300;
  if (ok) ok = orc_ceil_log2_pos(lo) == orc_ceil_log2_pos(hi);
  double scalefac = (double)(1 << shift);
  for (size_t i = 0; i < P; ++i) {
    double v = fabs(a[i]) * scalefac;
    double d = fabs((v - floor(v)) - 0.5);
    if (!(d > da[i] * scalefac)) ok = 0;
  }
  return ok;
}

/* The product's order certificate (levinson_quantize<.., CERT>, qlpc_kernel_impl.h), operation for operation: are the
 * quantised parameters of `a` (solution for R with R[0] = r0, forward vector `fwd`) those of ANY autocorrelation within
 * the summation bound eps of R?  max_abs_s = max |s| of the subframe, n = its length.  *tier2 = the row sums of |T^-1|
 * had to be evaluated.
 *
 * What is shown and what is assumed (DESIGN.md section 2 has the derivation).  With u = 2^-53:
 *   eps      = (n + 96) u S >= |R^ - R~|_inf: n - P roundings of the reference's chain + <= 71 of the kernel's order, each
 *              <= u x a partial sum of |products| <= u S, S = R0 + P max|s|^2 / 2 (Cauchy-Schwarz)            [shown]
 *   F_i      = rowsum_i(|T^-1|) eps (1 + |a|_1) >= |first-order change of a_i|; attained by low-pass material, whose
 *              T^-1 is a sign checkerboard under an alternating a (profiles/r06_fallback_ablation.txt, 6)     [shown]
 *   2nd order: |da_i| <= F_i (1 + rho' / (1 - rho)), rho, rho' < P 2^-precision once the subframe passes the boundary
 *              test below, i.e. <= 1.23 F_i for quant_precision >= 6 and P <= 12                               [shown]
 *   recursion: both recursions (the reference's on R^, the kernel's on R~) are floating point.  ASSUMED: on a system the
 *              recursion itself finds positive definite (every denominator 1 - err^2 > 0) the computed solution's
 *              residual obeys |T a^ - r|_inf <= c_L P^2 u R0 (1 + |a^|_1) with c_L <= 11; then the two runs' errors add at
 *              most (2 c_L P^2 / (n + 96)) F_i <= 0.77 F_i for P <= 12 and n >= 4096, the certified shapes (the allowance
 *              shrinks with n -- 0.94 at (256, 12) -- which is one reason the sub-wave shapes are not certified but given
 *              the reference's chains outright).  Measured (tools/certificate_attack.py, exact rational residuals,
 *              870 000 adversarial subframes): c_L <= 0.39.  Systems that are NOT positive definite -- the sums
 *              start at t = P for every lag, R[] need not be an autocorrelation: a block that opens on a clipped plateau
 *              is enough -- have no such bound (c_L up to 45 000 found, the two computed solutions 68 x further apart than
 *              2 F_i on a subframe the round-5 rule certified) and are excluded: round 6, `skipped` bit 1 below.
 *   safety   = 2.0 >= 1.23 + 2 c_L P^2 / (n + 96) for P <= 12, n >= 4096, c_L <= 11.
 * Not a theorem about the floating-point recursion -- c_L is evidence, and below quant_precision 6 so is the second-order
 * factor -- which is why tests/test_certificate_cpu.py soaks it and attacks it: the worst |a^_ref - a^_kernel|_i / (2 F_i)
 * a hill-climber finds among certifiable subframes is 0.05 (1.5 ... 68 before the exclusion). */
static void orc_cert_bounds(const double* R, const double* a, const double* fwd, size_t P, uint32_t max_abs_s, size_t n,
                            double* num_out, double* f0_out, double* eps_a_out) {
  const double safety = 2.0;
  double f1 = 0.0, a1 = 0.0;
  for (size_t i = 0; i < P; ++i) {
    f1 += fabs(fwd[i]);
    a1 += fabs(a[i]);
  }
  double m = (double)max_abs_s;
  double S = R[0] + (0.5 * (double)P) * (m * m);
  double eps = ((double)(n + 96) * 0x1p-53) * S;
  double eps_a = eps * (1.0 + a1);
  *f0_out = fabs(fwd[0]);
  *num_out = ((safety * 2.0) * (f1 * f1)) * eps_a; /* tier 1: |da_i| |f0| <= num for every i */
  *eps_a_out = eps_a;
}

/* tier 2: da[i] = safety x rowsum_i(|T^-1|) x eps_a; rows of T^-1 from its first column:
 * T^-1[i][j] = T^-1[i-1][j-1] + (f_i f_j - f_(P-i) f_(P-j)) / f_0 */
static void orc_cert_rows(const double* fwd, size_t P, double eps_a, double* da) {
  const double safety = 2.0;
  double row[ORC_MAX_LPC_ORDER], inv_f0 = 1.0 / fwd[0];
  double rs = 0.0;
  for (size_t j = 0; j < P; ++j) {
    row[j] = fwd[j];
    rs += fabs(row[j]);
  }
  da[0] = (safety * rs) * eps_a;
  for (size_t i = 1; i < P; ++i) {
    for (size_t j = P - 1; j >= 1; --j) {
      double t = (fwd[i] * fwd[j] - fwd[P - i] * fwd[P - j]) * inv_f0;
      row[j] = row[j - 1] + t;
    }
    row[0] = fwd[i];
    rs = 0.0;
    for (size_t j = 0; j < P; ++j) rs += fabs(row[j]);
    da[i] = (safety * rs) * eps_a;
  }
}

int orc_quant_certified(const double* R, const double* a, const double* fwd, size_t P, uint32_t max_abs_s, size_t n,
                        uint32_t precision, int* tier2) {
  const int allow_tier2 = 1;
  if (tier2) *tier2 = 0;
  int32_t shift = orc_find_shift(a, P, precision);
  double num, f0, eps_a;
  orc_cert_bounds(R, a, fwd, P, max_abs_s, n, &num, &f0, &eps_a);
  /* tier 1, compared multiplied through by |f0| (the kernel's form: no division on the common path) */
  double amax = 0.0;
  for (size_t i = 0; i < P; ++i) amax = fmax(amax, fabs(a[i]));
  int e = orc_ceil_log2_pos(amax);
  double g_lo = amax - ldexp(1.0, e - 1), g_hi = ldexp(1.0, e) - amax;
  int ok = amax > 0.0 && num < g_lo * f0 && num < g_hi * f0;
  double scalefac = (double)(1 << shift);
  double nums = num * scalefac;
  for (size_t i = 0; i < P; ++i) {
    double v = fabs(a[i]) * scalefac;
    double d = fabs((v - floor(v)) - 0.5);
    if (!(d * f0 > nums)) ok = 0;
  }
  if (ok) return 1;
  if (!allow_tier2) return 0;
  double da[ORC_MAX_LPC_ORDER];
  if (tier2) *tier2 = 1;
  orc_cert_rows(fwd, P, eps_a, da);
  return orc_quant_stable(a, P, shift, da);
}

/* find_shift, src/lpc.rs:234-254 */
int32_t orc_find_shift(const double* coefs, size_t n, uint32_t precision) {
  double max_abs = fabs(coefs[0]);
  for (size_t i = 1; i < n; ++i) max_abs = fmax(max_abs, fabs(coefs[i])); /* f64::max */
  double l = fmax(ceil(log2(max_abs)), (double)(INT16_MIN + 16));
  /* `.as_()` float -> i16 is a saturating cast, NaN -> 0 */
  int32_t abs_log2;
  if (l != l) abs_log2 = 0;
  else if (l >= 32767.0) abs_log2 = 32767;
  else if (l <= -32768.0) abs_log2 = -32768;
  else abs_log2 = (int32_t)l;
  int32_t shift = ((int32_t)precision - 1) - abs_log2;
  if (shift < ORC_QLPC_MIN_SHIFT) shift = ORC_QLPC_MIN_SHIFT;
  if (shift > ORC_QLPC_MAX_SHIFT) shift = ORC_QLPC_MAX_SHIFT;
  return shift;
}

/* quantize_parameter, src/lpc.rs:258-270 */
static int16_t orc_quantize_parameter(double p, int32_t shift) {
  double scalefac = ldexp(1.0, shift); /* powi(2, shift): exact */
  double scaled_int = round(p * scalefac); /* Float::round: half away from zero */
  if (scaled_int < -32768.0) scaled_int = -32768.0;
  if (scaled_int > 32767.0) scaled_int = 32767.0;
  if (scaled_int != scaled_int) return 0;
  return (int16_t)scaled_int;
}

/* quantize_parameters, src/lpc.rs:273-302 */
void orc_quantize_parameters(const double* coefs, size_t n, uint32_t precision, orc_qparams* out) {
  memset(out, 0, sizeof(*out));
  out->precision = precision;
  if (n == 0) return; /* from_parts(&[], 0, 0, precision) */
  int32_t shift = orc_find_shift(coefs, n, precision);
  int32_t lo = -(1 << (precision - 1));
  int32_t hi = (1 << (precision - 1)) - 1;
  for (size_t i = 0; i < n; ++i) {
    int32_t q = orc_quantize_parameter(coefs[i], shift);
    if (q < lo) q = lo;
    if (q > hi) q = hi;
    out->coefs[i] = (int16_t)q;
  }
  size_t order = 1;
  for (size_t i = 0; i < n; ++i)
    if (out->coefs[i] != 0) order = i + 1; /* tail-zero truncation, min 1 (:295-299) */
  for (size_t i = order; i < ORC_MAX_LPC_ORDER; ++i) out->coefs[i] = 0;
  out->order = (uint32_t)order;
  out->shift = shift;
}

/* compute_error_impl::<i32, _>, src/lpc.rs:306-350 (wrapping lanes) */
static void orc_compute_error_i32(const orc_qparams* qp, const int32_t* signal, size_t n,
                                  int32_t* errors) {
  uint32_t* acc = (uint32_t*)errors;
  for (size_t t = 0; t < n; ++t) acc[t] = 0;
  for (size_t j = 0; j < qp->order; ++j) {
    uint32_t w = (uint32_t)(int32_t)qp->coefs[j];
    for (size_t i = 0; i + j + 1 < n; ++i) acc[i + j + 1] += w * (uint32_t)signal[i];
  }
  for (size_t t = 0; t < n; ++t) {
    int32_t px = (int32_t)acc[t];
    errors[t] = (int32_t)((uint32_t)signal[t] - (uint32_t)(px >> qp->shift));
  }
  for (size_t t = 0; t < qp->order && t < n; ++t) errors[t] = 0;
}

/* compute_error_impl::<i64, _> followed by `as i32`, src/lpc.rs:379-388 */
static void orc_compute_error_i64(const orc_qparams* qp, const int32_t* signal, size_t n,
                                  int32_t* errors) {
  ORC_GROW(orc_tls.acc64, orc_tls.acc64_cap, n + 1, int64_t);
  int64_t* acc = orc_tls.acc64;
  for (size_t t = 0; t < n; ++t) acc[t] = 0;
  for (size_t j = 0; j < qp->order; ++j) {
    int64_t w = qp->coefs[j];
    for (size_t i = 0; i + j + 1 < n; ++i) acc[i + j + 1] += w * (int64_t)signal[i];
  }
  for (size_t t = 0; t < n; ++t) {
    int64_t e = (int64_t)signal[t] - (acc[t] >> qp->shift);
    errors[t] = (int32_t)(uint32_t)(uint64_t)e; /* `as i32` truncation */
  }
  for (size_t t = 0; t < qp->order && t < n; ++t) errors[t] = 0;
}

/* compute_error, src/lpc.rs:359-390 */
void orc_compute_error(const orc_qparams* qp, const int32_t* signal, size_t n, int32_t* errors) {
  uint64_t maxabs_signal = 0;
  for (size_t t = 0; t < n; ++t) {
    uint64_t a = signal[t] < 0 ? (uint64_t)(-(int64_t)signal[t]) : (uint64_t)signal[t];
    if (a > maxabs_signal) maxabs_signal = a;
  }
  int64_t sumabs = 0;
  for (size_t j = 0; j < ORC_MAX_LPC_ORDER; ++j) sumabs += qp->coefs[j] < 0 ? -qp->coefs[j] : qp->coefs[j];
  uint64_t maxabs = maxabs_signal * (uint64_t)sumabs;
  if (maxabs < (uint64_t)INT32_MAX) orc_compute_error_i32(qp, signal, n, errors);
  else orc_compute_error_i64(qp, signal, n, errors);
}

/* LpcEstimator::weighted_lpc_from_auto_corr with NoWeight, src/lpc.rs:760-801,
 * reached through lpc_from_autocorr (src/lpc.rs:920-930, LpcEstimator<f64>). */
int orc_lpc_from_autocorr(const int32_t* signal, size_t n, const orc_qlpc_config* cfg,
                          double* autocorr_out, double* coefs_out) {
  size_t lpc_order = cfg->lpc_order;
  for (size_t i = 0; i < lpc_order; ++i) coefs_out[i] = 0.0;
  if (lpc_order == 0) return ORC_STATUS_OK;
  /* get_window, src/lpc.rs:222-231: the table is computed once per (size, window) per thread */
  if (orc_tls.window == NULL || orc_tls.window_n != n || orc_tls.window_type != cfg->window_type ||
      orc_tls.window_alpha != cfg->tukey_alpha) {
    free(orc_tls.window);
    orc_tls.window = (float*)malloc(sizeof(float) * (n ? n : 1));
    orc_window_weights(cfg->window_type, cfg->tukey_alpha, n, orc_tls.window);
    orc_tls.window_n = n;
    orc_tls.window_type = cfg->window_type;
    orc_tls.window_alpha = cfg->tukey_alpha;
  }
  ORC_GROW(orc_tls.xw, orc_tls.xw_cap, n + 1, float);
  float* window = orc_tls.window;
  float* xw = orc_tls.xw;
  orc_fill_windowed_signal(signal, window, n, xw);
  double corr[ORC_MAX_LPC_ORDER + 1];
  if (cfg->acorr_order == ORC_ACORR_CANONICAL && orc_default_order_is_certified(n, lpc_order)) {
    /* the unflagged product on these shapes: the fused kernel's lane-order sums where their quantised parameters are
     * certified to be the reference's, the reference's own chains where not */
    orc_auto_correlation_lane_order_f64(lpc_order + 1, xw, n, corr);
    int st = ORC_STATUS_OK;
    for (size_t i = 0; i <= lpc_order; ++i)
      if (isnan(corr[i]) || isinf(corr[i])) st = ORC_STATUS_NONFINITE;
    double fwd[ORC_MAX_LPC_ORDER + 1];
    int skipped = 0, certified = 0;
    if (st == ORC_STATUS_OK) st = orc_levinson_f64_forward(corr, corr + 1, lpc_order, coefs_out, fwd, &skipped);
    if (st == ORC_STATUS_OK)
      for (size_t i = 0; i < lpc_order; ++i)
        if (isnan(coefs_out[i]) || isinf(coefs_out[i])) st = ORC_STATUS_NONFINITE;
    if (st == ORC_STATUS_OK) {
      if (corr[0] == 0.0) {
        certified = 1; /* digital silence: exact zeros in either order */
      } else if (!skipped) {
        uint32_t maxabs = 0;
        for (size_t t = 0; t < n; ++t) {
          uint32_t m = signal[t] < 0 ? (uint32_t)0 - (uint32_t)signal[t] : (uint32_t)signal[t];
          if (m > maxabs) maxabs = m;
        }
        int tier2 = 0;
        certified = orc_quant_certified(corr, coefs_out, fwd, lpc_order, maxabs, n, cfg->quant_precision, &tier2);
        orc_cert_stats[1] += (unsigned long)tier2;
      }
    }
    orc_cert_stats[0] += 1;
    if (certified) {
      if (autocorr_out)
        for (size_t i = 0; i <= lpc_order; ++i) autocorr_out[i] = corr[i];
      return ORC_STATUS_OK;
    }
    orc_cert_stats[2] += 1;
    for (size_t i = 0; i < lpc_order; ++i) coefs_out[i] = 0.0;
    orc_auto_correlation_f64(lpc_order + 1, xw, n, corr);
  } else if (cfg->acorr_order == ORC_ACORR_CHUNK_TREE && orc_default_order_is_certified(n, lpc_order))
    orc_auto_correlation_lane_order_f64(lpc_order + 1, xw, n, corr); /* the fused kernel's order, uncertified */
  else if (cfg->acorr_order == ORC_ACORR_GENERIC_TREE ||
           (cfg->acorr_order == ORC_ACORR_CHUNK_TREE && !orc_default_order_is_stable(n, lpc_order)) ||
           (cfg->acorr_order == ORC_ACORR_CANONICAL && !orc_default_order_is_stable(n, lpc_order) &&
            !orc_default_order_is_two_pass(n, lpc_order)))
    orc_auto_correlation_canonical_f64(lpc_order + 1, xw, n, corr);
  else if (cfg->acorr_order == ORC_ACORR_NIGHTLY)
    orc_auto_correlation_nightly_f64(lpc_order + 1, xw, n, corr, 0);
  else
    orc_auto_correlation_f64(lpc_order + 1, xw, n, corr);
  int status = ORC_STATUS_OK;
  for (size_t i = 0; i <= lpc_order; ++i) {
    if (autocorr_out) autocorr_out[i] = corr[i];
    if (isnan(corr[i]) || isinf(corr[i])) status = ORC_STATUS_NONFINITE; /* assert :786-791 */
  }
  if (status != ORC_STATUS_OK) return status;
  status = orc_symmetric_levinson_f64(corr, corr + 1, lpc_order, coefs_out);
  if (status != ORC_STATUS_OK) return status;
  for (size_t i = 0; i < lpc_order; ++i)
    if (isnan(coefs_out[i]) || isinf(coefs_out[i])) return ORC_STATUS_NONFINITE; /* :797-799 */
  return ORC_STATUS_OK;
}

/* Tool / test hook (tools/certificate_attack.py, tests/test_certificate_cpu.py): the certificate's quantities for one
 * subframe of a certified shape -- R[] in the fused kernel's lane order, the recursion's solution and, per coefficient,
 * the second tier's bound da[i] (safety included); *tier1 = the first tier's uniform bound num / |f0|.  Returns the
 * recursion's status, or -1 where the certificate does not apply (shape, digital silence, a skipped step). */
int orc_certificate_bounds(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, double* corr_out, double* coefs_out,
                           double* da_out, double* tier1_out) {
  size_t P = cfg->lpc_order;
  if (!orc_default_order_is_certified(n, P)) return -1;
  float* window = (float*)malloc(sizeof(float) * n);
  float* xw = (float*)malloc(sizeof(float) * (n + 1));
  orc_window_weights(cfg->window_type, cfg->tukey_alpha, n, window);
  orc_fill_windowed_signal(signal, window, n, xw);
  double corr[ORC_MAX_LPC_ORDER + 1], fwd[ORC_MAX_LPC_ORDER + 1];
  orc_auto_correlation_lane_order_f64(P + 1, xw, n, corr);
  free(window);
  free(xw);
  int skipped = 0;
  int st = orc_levinson_f64_forward(corr, corr + 1, P, coefs_out, fwd, &skipped);
  for (size_t i = 0; i <= P; ++i) corr_out[i] = corr[i];
  if (st != ORC_STATUS_OK) return st;
  if (corr[0] == 0.0 || skipped) return -1;
  uint32_t maxabs = 0;
  for (size_t t = 0; t < n; ++t) {
    uint32_t m = signal[t] < 0 ? (uint32_t)0 - (uint32_t)signal[t] : (uint32_t)signal[t];
    if (m > maxabs) maxabs = m;
  }
  double num, f0, eps_a;
  orc_cert_bounds(corr, coefs_out, fwd, P, maxabs, n, &num, &f0, &eps_a);
  *tier1_out = num / f0;
  orc_cert_rows(fwd, P, eps_a, da_out);
  return ORC_STATUS_OK;
}

/* ------------------------------------------------------------------------ */
/* experimental: covariance-method LPC ("direct MSE") and IRLS  */
/*                                                                            */
/* PARITY UNPINNED beyond the reference's own (mostly qualitative) tests: the */
/* linear solver is nalgebra 0.32 (Cargo.toml:45, optional dependency, not    */
/* under /root/reference).  Its published algorithm is restated below from    */
/* nalgebra 0.32.x src/linalg/cholesky.rs (Cholesky::new_internal, solve_mut),*/
/* src/linalg/solve.rs (solve_lower_triangular_vector_unchecked_mut,          */
/* xx_solve_lower_triangular_vector_unchecked_mut) and src/base/blas.rs       */
/* (axcpy / array_axcpy, dotx's 8-accumulator loop).                          */
/* ------------------------------------------------------------------------ */

/* weighted_auto_correlation_nosimd, src/lpc.rs:533-548, with an optional VecWeight (:194-198):
 * wy = weight[t] * signal[t] in f32, then widened. */
void orc_weighted_auto_correlation_nosimd_f64(size_t order, const float* signal, size_t n, const float* weight,
                                              double* dest) {
  for (size_t tau = 0; tau < order; ++tau) dest[tau] = 0.0;
  if (order == 0) return;
  for (size_t t = order - 1; t < n; ++t) {
    float wyf = weight ? weight[t] * signal[t] : signal[t];
    double wy = (double)wyf;
    for (size_t tau = 0; tau < order; ++tau) dest[tau] = fma((double)signal[t - tau], wy, dest[tau]);
  }
}

/* weighted_lagged_outer_prod_sum, src/lpc.rs:573-600: dest is order x order, column-major
 * (nalgebra::DMatrix); `weight` is indexed t + wshift (ShiftedWeight<M>, src/lpc.rs:205-215). */
void orc_weighted_lagged_outer_prod_sum_f64(size_t order, const float* signal, size_t len, const float* weight,
                                            size_t wshift, double* dest) {
  for (size_t k = 0; k < order * order; ++k) dest[k] = 0.0;
  if (order == 0) return;
  for (size_t t = order - 1; t < len; ++t) {
    for (size_t i = 0; i < order; ++i) {
      for (size_t j = i; j < order; ++j) {
        float wxf = weight ? weight[t + wshift] * signal[t - j] : signal[t - j];
        dest[i + j * order] = fma((double)signal[t - i], (double)wxf, dest[i + j * order]);
      }
    }
  }
  for (size_t i = 0; i < order; ++i)
    for (size_t j = i + 1; j < order; ++j) dest[j + i * order] = dest[i + j * order];
}

/* LpcFloat::solve_sym_mut, src/lpc.rs:79-87: mat.clone().cholesky() then decompose.solve_mut(v).
 * Returns 1 and overwrites v with the solution, or 0 (v untouched) when the factorisation fails.
 * nalgebra: column j -= L[j][k] * column k for k < j as  y = (a * x) * 1 + 1 * y  (two roundings, no fma:
 * array_axcpy); pivot must be non-zero with a real square root (v >= 0); column /= pivot.  Forward solve:
 * b[i] /= L[i][i]; b[i+1..] += (-b[i]) * L[i+1.., i].  Backward (adjoint) solve: b[i] = (b[i] - dot(L[i+1.., i],
 * b[i+1..])) / L[i][i] with dotx's eight partial accumulators. */
int orc_cholesky_solve(const double* mat, size_t n, double* v) {
  double m[ORC_MAX_LPC_ORDER * ORC_MAX_LPC_ORDER];
  for (size_t k = 0; k < n * n; ++k) m[k] = mat[k];
#define M(r, c) m[(r) + (c) * n]
  for (size_t j = 0; j < n; ++j) {
    for (size_t k = 0; k < j; ++k) {
      double factor = -M(j, k);
      for (size_t r = j; r < n; ++r) {
        double ax = factor * M(r, k);
        M(r, j) = (ax * 1.0) + (1.0 * M(r, j));
      }
    }
    double diag = M(j, j);
    if (diag == 0.0 || !(diag >= 0.0)) return 0; /* is_zero() / try_sqrt() == None (NaN too) */
    double denom = sqrt(diag);
    M(j, j) = denom;
    for (size_t r = j + 1; r < n; ++r) M(r, j) = M(r, j) / denom;
  }
  /* solve_lower_triangular_unchecked_mut */
  for (size_t i = 0; i < n; ++i) {
    double coeff = v[i] / M(i, i);
    v[i] = coeff;
    double a = -coeff;
    for (size_t r = i + 1; r < n; ++r) v[r] = ((a * M(r, i)) * 1.0) + (1.0 * v[r]);
  }
  /* ad_solve_lower_triangular_unchecked_mut */
  for (size_t ii = n; ii-- > 0;) {
    size_t rows = n - (ii + 1);
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double res = 0.0;
    size_t i = 0;
    while (rows - i >= 8) {
      for (int q = 0; q < 8; ++q) acc[q] += M(ii + 1 + i + q, ii) * v[ii + 1 + i + q];
      i += 8;
    }
    res += acc[0] + acc[4];
    res += acc[1] + acc[5];
    res += acc[2] + acc[6];
    res += acc[3] + acc[7];
    for (size_t k = i; k < rows; ++k) res += M(ii + 1 + k, ii) * v[ii + 1 + k];
    v[ii] = (v[ii] - res) / M(ii, ii);
  }
#undef M
  return 1;
}

static float* orc_windowed(const int32_t* signal, size_t n, const orc_qlpc_config* cfg) {
  if (orc_tls.window == NULL || orc_tls.window_n != n || orc_tls.window_type != cfg->window_type ||
      orc_tls.window_alpha != cfg->tukey_alpha) {
    free(orc_tls.window);
    orc_tls.window = (float*)malloc(sizeof(float) * (n ? n : 1));
    orc_window_weights(cfg->window_type, cfg->tukey_alpha, n, orc_tls.window);
    orc_tls.window_n = n;
    orc_tls.window_type = cfg->window_type;
    orc_tls.window_alpha = cfg->tukey_alpha;
  }
  ORC_GROW(orc_tls.xw, orc_tls.xw_cap, n + 1, float);
  orc_fill_windowed_signal(signal, orc_tls.window, n, orc_tls.xw);
  return orc_tls.xw;
}

/* LpcEstimator::weighted_lpc_with_direct_mse, src/lpc.rs:853-903.  `weight` NULL = NoWeight.
 * gram_out (optional) receives the order x order matrix before regularisation, column-major. */
int orc_weighted_lpc_with_direct_mse(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, const float* weight,
                                     double* autocorr_out, double* gram_out, double* coefs_out) {
  size_t order = cfg->lpc_order;
  for (size_t i = 0; i < order; ++i) coefs_out[i] = 0.0;
  if (order == 0 || n < order + 1) return ORC_STATUS_OK;
  const float* xw = orc_windowed(signal, n, cfg);
  double corr[ORC_MAX_LPC_ORDER + 1];
  double gram[ORC_MAX_LPC_ORDER * ORC_MAX_LPC_ORDER];
  orc_weighted_auto_correlation_nosimd_f64(order + 1, xw, n, weight, corr);
  /* the signal without its last sample, the weight shifted by one (ShiftedWeight::<1, _>) */
  orc_weighted_lagged_outer_prod_sum_f64(order, xw, n - 1, weight, 1, gram);
  if (autocorr_out)
    for (size_t i = 0; i <= order; ++i) autocorr_out[i] = corr[i];
  if (gram_out)
    for (size_t k = 0; k < order * order; ++k) gram_out[k] = gram[k];
  double xy[ORC_MAX_LPC_ORDER];
  for (size_t i = 0; i < order; ++i) xy[i] = corr[i + 1];
  double regularizer = 0.0;
  int tries = 0;
  while (!orc_cholesky_solve(gram, order, xy)) {
    double old = regularizer;
    double twice = regularizer + regularizer;
    regularizer = 1.0 > twice ? 1.0 : twice; /* T::one().max(regularizer + regularizer) */
    for (size_t i = 0; i < order; ++i) gram[i + i * order] += regularizer - old;
    if (++tries > 2000) return ORC_STATUS_NONFINITE; /* (the reference would loop forever on NaN input) */
  }
  for (size_t i = 0; i < order; ++i) coefs_out[i] = xy[i];
  return ORC_STATUS_OK;
}

/* compute_raw_errors, src/lpc.rs:602-618 (f32, mul_add) */
void orc_compute_raw_errors(const int32_t* signal, size_t n, const double* coefs, size_t order, float* errors) {
  for (size_t t = order; t < n; ++t) {
    float e = (float)(int32_t)(0u - (uint32_t)signal[t]);
    for (size_t j = 0; j < order; ++j) e = fmaf((float)coefs[j], (float)signal[t - 1 - j], e);
    errors[t] = e;
  }
}

/* LpcEstimator::lpc_with_irls_mae, src/lpc.rs:814-850.  f32::powf is libm's powf. */
int orc_lpc_with_irls_mae(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, size_t steps,
                          double* autocorr_out, double* coefs_out) {
  size_t order = cfg->lpc_order;
  float* weights = (float*)malloc(sizeof(float) * (n ? n : 1));
  float* raw = (float*)calloc(n ? n : 1, sizeof(float));
  for (size_t t = 0; t < n; ++t) weights[t] = 1.0f;
  double best[ORC_MAX_LPC_ORDER], coefs[ORC_MAX_LPC_ORDER], corr[ORC_MAX_LPC_ORDER + 1];
  float best_error = FLT_MAX;
  int have = 0, status = ORC_STATUS_OK;
  int32_t maxabs = 0;
  for (size_t t = 0; t < n; ++t) {
    int32_t a = signal[t] < 0 ? (int32_t)(0u - (uint32_t)signal[t]) : signal[t];
    if (a > maxabs) maxabs = a;
  }
  float normalizer = (float)maxabs;
  for (size_t it = 0; it <= steps; ++it) {
    status = orc_weighted_lpc_with_direct_mse(signal, n, cfg, weights, corr, NULL, coefs);
    if (status != ORC_STATUS_OK) break;
    orc_compute_raw_errors(signal, n, coefs, order, raw);
    float sum_abs_err = 0.0f;
    for (size_t t = 0; t < n; ++t) sum_abs_err += fabsf(raw[t]);
    if (sum_abs_err < best_error) {
      best_error = sum_abs_err;
      for (size_t i = 0; i < order; ++i) best[i] = coefs[i];
      if (autocorr_out)
        for (size_t i = 0; i <= order; ++i) autocorr_out[i] = corr[i];
      have = 1;
    }
    for (size_t t = order; t < n; ++t) {
      float a = fabsf(raw[t]);
      a = a > 1.0f ? a : 1.0f;   /* f32::max(1.0) */
      float x = a / normalizer;
      x = x > 0.01f ? x : 0.01f; /* .max(0.01) */
      weights[t] = powf(x, -1.2f);
    }
  }
  for (size_t i = 0; i < order; ++i) coefs_out[i] = have ? best[i] : 0.0;
  free(weights);
  free(raw);
  return status; /* (best_coefs.unwrap() panics if no step improved on f32::MAX: NaN errors) */
}

/* perform_qlpc, src/coding.rs:333-351: acorr_order's low byte ORC_ACORR_DIRECT_MSE selects
 * config.qlpc.use_direct_mse, bits 8.. = mae_optimization_steps */
int orc_perform_qlpc(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, double* autocorr_out,
                     double* coefs_out) {
  if ((cfg->acorr_order & 0xFFu) == ORC_ACORR_DIRECT_MSE) {
    size_t steps = cfg->acorr_order >> 8;
    int st = steps > 0 ? orc_lpc_with_irls_mae(signal, n, cfg, steps, autocorr_out, coefs_out)
                       : orc_weighted_lpc_with_direct_mse(signal, n, cfg, NULL, autocorr_out, NULL, coefs_out);
    if (st != ORC_STATUS_OK) return st;
    for (size_t i = 0; i < cfg->lpc_order; ++i)
      if (isnan(coefs_out[i]) || isinf(coefs_out[i])) return ORC_STATUS_NONFINITE;
    return ORC_STATUS_OK;
  }
  return orc_lpc_from_autocorr(signal, n, cfg, autocorr_out, coefs_out);
}

/* ------------------------------------------------------------------------ */
/* src/rice.rs                                                              */
/* ------------------------------------------------------------------------ */

/* encode_signbit, src/rice.rs:169-171 */
uint32_t orc_encode_signbit(int32_t v) {
  uint32_t a = v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v; /* unsigned_abs */
  return (a << 1) - (uint32_t)(v < 0);
}

/* decode_signbit, src/rice.rs:180-187 */
int32_t orc_decode_signbit(uint32_t v) {
  if (v & 1u) return -(int32_t)((v >> 1) + 1u);
  return (int32_t)(v >> 1);
}

/* PrcBitTable::from_errors, src/rice.rs:65-103: u32 lanes with wrapping adds,
 * clamp to MAX_P_TO_BITS after every 16-sample chunk and after the offset. */
void orc_prc_bit_table_from_errors(const uint32_t* errors, size_t len, uint32_t offset,
                                   uint32_t table[32]) {
  uint32_t acc[32];
  for (int p = 0; p < 32; ++p) acc[p] = 0;
  for (size_t base = 0; base < len; base += 16) {
    size_t end = base + 16 < len ? base + 16 : len;
    for (size_t i = base; i < end; ++i) {
      uint32_t v = errors[i];
      for (int p = 0; p < 32; ++p) acc[p] += v >> p;
    }
    for (int p = 0; p < 32; ++p)
      if (acc[p] > ORC_MAX_P_TO_BITS) acc[p] = ORC_MAX_P_TO_BITS;
  }
  for (int p = 0; p < 32; ++p) {
    uint32_t off = offset + (uint32_t)len * (uint32_t)(p + 1);
    uint32_t v = acc[p] + off;
    table[p] = v > ORC_MAX_P_TO_BITS ? ORC_MAX_P_TO_BITS : v;
  }
}

/* PrcBitTable::minimizer, src/rice.rs:115-141: min over (bits << 5) | p */
void orc_prc_minimizer(const uint32_t table[32], uint32_t max_p, uint32_t* p_out,
                       uint32_t* bits_out) {
  uint32_t lo_max = max_p < 15 ? max_p : 15;
  uint32_t minim = 0xFFFFFFFFu;
  for (uint32_t p = 0; p < 16; ++p) {
    uint32_t v = p <= lo_max ? table[p] : 0xFFFFFFFFu;
    uint32_t packed = (v << 5) | p;
    if (packed < minim) minim = packed;
  }
  if (max_p > 15) {
    for (uint32_t p = 16; p < 32; ++p) {
      uint32_t v = p <= max_p ? table[p] : 0xFFFFFFFFu;
      uint32_t packed = (v << 5) | p;
      if (packed < minim) minim = packed;
    }
  }
  *bits_out = minim >> 5;
  *p_out = minim & 0x1Fu;
}

/* PrcBitTable::merge, src/rice.rs:144-152 */
void orc_prc_merge(const uint32_t a[32], const uint32_t b[32], uint32_t offset, uint32_t out[32]) {
  for (int p = 0; p < 32; ++p) {
    uint32_t v = a[p] + b[p] - offset;
    out[p] = v > ORC_MAX_P_TO_BITS ? ORC_MAX_P_TO_BITS : v;
  }
}

/* finest_partition_order, src/rice.rs:157-165 */
uint32_t orc_finest_partition_order(size_t size, size_t min_part_size) {
  uint32_t max_splits = (uint32_t)(size / min_part_size);
  if (max_splits == 0) return 0; /* unreachable on the path: blocks < 64 never arrive (coding.rs:396) */
  uint32_t max_order_for_min_part = 31u - (uint32_t)__builtin_clz(max_splits);
  uint32_t tz = size ? (uint32_t)__builtin_ctzl(size) : 64u;
  uint32_t m = max_order_for_min_part < tz ? max_order_for_min_part : tz;
  return m < ORC_MAX_RICE_PARTITION_ORDER ? m : ORC_MAX_RICE_PARTITION_ORDER;
}

/* PrcParameterFinder::find, src/rice.rs:246-298 (eval_partitions :193-202,
 * merge_partitions :208-216). */
void orc_find_partitioned_rice_parameter(const int32_t* signal, size_t n, size_t warmup_length,
                                         uint32_t max_p, orc_prc_parameter* out) {
  /* ORC_RICE_FINEST_ONLY (not a reference mode; BASELINE config 2's "fixed Rice partition order"):
   * the search stops at the finest order instead of merging down to order 0 */
  const int finest_only = (max_p & ORC_RICE_FINEST_ONLY) != 0;
  max_p &= 0xFFu;
  size_t min_part = warmup_length > ORC_MIN_RICE_PARTITION_SIZE ? warmup_length
                                                                : ORC_MIN_RICE_PARTITION_SIZE;
  uint32_t partition_order = orc_finest_partition_order(n, min_part);
  size_t nparts = (size_t)1 << partition_order;
  ORC_GROW(orc_tls.uerr, orc_tls.uerr_cap, n + 1, uint32_t);
  if (orc_tls.tables_cap < nparts) {
    free(orc_tls.tables);
    orc_tls.tables_cap = nparts;
    orc_tls.tables = (uint32_t(*)[32])malloc(sizeof(uint32_t[32]) * nparts);
  }
  if (orc_tls.ps_cap < nparts) {
    free(orc_tls.ps);
    free(orc_tls.min_ps);
    orc_tls.ps_cap = nparts;
    orc_tls.ps = (uint8_t*)malloc(nparts);
    orc_tls.min_ps = (uint8_t*)malloc(nparts);
  }
  uint32_t* errors = orc_tls.uerr;
  uint32_t(*tables)[32] = orc_tls.tables;
  uint8_t* ps = orc_tls.ps;
  uint8_t* min_ps = orc_tls.min_ps;
  for (size_t t = 0; t < n; ++t) errors[t] = orc_encode_signbit(signal[t]);
  size_t part_size = n / nparts;
  for (size_t p = 0; p < nparts; ++p) {
    size_t start = p * part_size > warmup_length ? p * part_size : warmup_length;
    size_t end = (p + 1) * part_size;
    orc_prc_bit_table_from_errors(errors + start, end - start, 4, tables[p]);
  }
  uint64_t min_bits = 0;
  for (size_t p = 0; p < nparts; ++p) {
    uint32_t pp, bits;
    orc_prc_minimizer(tables[p], max_p, &pp, &bits);
    min_bits += bits;
    min_ps[p] = (uint8_t)pp;
  }
  uint32_t min_order = partition_order;
  while (nparts > 1 && !finest_only) {
    size_t merged = nparts / 2;
    for (size_t q = 0; q < merged; ++q) {
      uint32_t tmp[32];
      orc_prc_merge(tables[2 * q], tables[2 * q + 1], 4, tmp);
      memcpy(tables[q], tmp, sizeof(tmp));
    }
    nparts = merged;
    partition_order -= 1;
    uint64_t next_bits = 0;
    for (size_t q = 0; q < nparts; ++q) {
      uint32_t pp, bits;
      orc_prc_minimizer(tables[q], max_p, &pp, &bits);
      next_bits += bits;
      ps[q] = (uint8_t)pp;
    }
    if (next_bits < min_bits) { /* strict: ties keep the finer order (:285) */
      min_bits = next_bits;
      memcpy(min_ps, ps, nparts);
      min_order = partition_order;
    }
  }
  out->order = min_order;
  out->code_bits = min_bits;
  memcpy(out->ps, min_ps, (size_t)1 << min_order);
}

/* ------------------------------------------------------------------------ */
/* src/coding.rs, src/component/bitrepr.rs, src/component/decode.rs         */
/* ------------------------------------------------------------------------ */

/* encode_residual_with_prc_parameter (src/coding.rs:140-170) with
 * quotients_and_remainders (:58-62); sums as Residual::from_parts
 * (src/component/datatype.rs:2325-2332). */
void orc_encode_residual_with_prc_parameter(const int32_t* errors, size_t n, size_t warmup_length,
                                            const orc_prc_parameter* prc, uint32_t* quotients,
                                            uint32_t* remainders, uint64_t* sum_quotients,
                                            uint64_t* sum_rice_params) {
  size_t nparts = (size_t)1 << prc->order;
  size_t part_size = n >> prc->order;
  if (quotients) memset(quotients, 0, sizeof(uint32_t) * n);
  if (remainders) memset(remainders, 0, sizeof(uint32_t) * n);
  uint64_t sq = 0, sp = 0;
  size_t offset = 0;
  for (size_t q = 0; q < nparts; ++q) {
    uint32_t rice_p = prc->ps[q];
    size_t start = offset > warmup_length ? offset : warmup_length;
    offset += part_size;
    size_t end = offset;
    uint32_t mask = (1u << rice_p) - 1u;
    for (size_t t = start; t < end; ++t) {
      uint32_t u = orc_encode_signbit(errors[t]);
      if (quotients) quotients[t] = u >> rice_p;
      if (remainders) remainders[t] = u & mask;
      sq += u >> rice_p;
    }
    sp += rice_p;
  }
  *sum_quotients = sq;
  *sum_rice_params = sp;
}

/* BitRepr for Residual::count_bits, src/component/bitrepr.rs:533-544 */
uint64_t orc_residual_count_bits(size_t block_size, size_t warmup_length, uint32_t partition_order,
                                 const uint8_t* rice_params, uint64_t sum_quotients,
                                 uint64_t sum_rice_params) {
  uint64_t nparts = (uint64_t)1 << partition_order;
  uint64_t quotient_bits = sum_quotients + block_size - warmup_length;
  uint64_t remainder_bits = sum_rice_params * (uint64_t)(block_size >> partition_order);
  remainder_bits -= (uint64_t)warmup_length * rice_params[0];
  int use_rice2 = 0;
  for (uint64_t q = 0; q < nparts; ++q) use_rice2 |= rice_params[q] > 14;
  uint64_t param_bits = use_rice2 ? 5 : 4;
  return 2 + 4 + nparts * param_bits + quotient_bits + remainder_bits;
}

/* BitRepr for Lpc::count_bits, src/component/bitrepr.rs:492-499 */
uint64_t orc_lpc_count_bits(uint32_t bits_per_sample, uint32_t order, uint32_t precision,
                            uint64_t residual_bits) {
  uint64_t warm_up_bits = (uint64_t)bits_per_sample * order;
  return 8 + warm_up_bits + 4 + 5 + (uint64_t)precision * order + residual_bits;
}

/* Verbatim::count_bits_from_metadata, src/component/datatype.rs:1944-1949 */
uint64_t orc_verbatim_count_bits(size_t n, uint32_t bits_per_sample) {
  return 8 + (uint64_t)n * bits_per_sample;
}

/* Decode for Residual::copy_signal, src/component/decode.rs:226-237 */
void orc_decode_residual(size_t block_size, uint32_t partition_order, const uint8_t* rice_params,
                         const uint32_t* quotients, const uint32_t* remainders, int32_t* dest) {
  size_t part_len = block_size >> partition_order;
  for (size_t t = 0; t < block_size; ++t)
    dest[t] = orc_decode_signbit((quotients[t] << rice_params[t / part_len]) + remainders[t]);
}

/* decode_lpc, src/component/decode.rs:159-177 (i64 prediction, wrapping i32 add) */
void orc_decode_lpc(const int32_t* warm_up, size_t order, const int16_t* coefs, uint32_t shift,
                    const int32_t* residual, size_t n, int32_t* dest) {
  for (size_t t = 0; t < n; ++t) dest[t] = residual[t];
  for (size_t t = 0; t < order && t < n; ++t) dest[t] = warm_up[t];
  for (size_t t = order; t < n; ++t) {
    int64_t pred = 0;
    for (size_t tau = 0; tau < order; ++tau) pred += (int64_t)coefs[tau] * (int64_t)dest[t - 1 - tau];
    dest[t] = (int32_t)((uint32_t)dest[t] + (uint32_t)(int32_t)(pred >> shift));
  }
}

/* estimated_qlpc, src/coding.rs:360-381 (perform_qlpc :333-351 without the
 * experimental branches; encode_residual :173-176). */
void orc_estimated_qlpc(const int32_t* signal, size_t n, uint32_t bits_per_sample,
                        const orc_qlpc_config* cfg, orc_qlpc_result* res, uint8_t* rice_params,
                        int32_t* errors, uint32_t* quotients, uint32_t* remainders) {
  memset(res, 0, sizeof(*res));
  res->status = orc_perform_qlpc(signal, n, cfg, res->autocorr, res->lpc_coefs);
  if (res->status != ORC_STATUS_OK) {
    res->qp.precision = cfg->quant_precision;
    for (size_t t = 0; t < n; ++t) errors[t] = 0;
    return;
  }
  orc_quantize_parameters(res->lpc_coefs, cfg->lpc_order, cfg->quant_precision, &res->qp);
  orc_compute_error(&res->qp, signal, n, errors);
  static __thread orc_prc_parameter prc_storage;
  orc_prc_parameter* prc = &prc_storage;
  orc_find_partitioned_rice_parameter(errors, n, res->qp.order, cfg->max_rice_parameter, prc);
  res->rice_order = prc->order;
  res->code_bits = prc->code_bits;
  memcpy(rice_params, prc->ps, (size_t)1 << prc->order);
  orc_encode_residual_with_prc_parameter(errors, n, res->qp.order, prc, quotients, remainders,
                                         &res->sum_quotients, &res->sum_rice_params);
  res->residual_bits = orc_residual_count_bits(n, res->qp.order, prc->order, prc->ps,
                                               res->sum_quotients, res->sum_rice_params);
  res->subframe_bits =
      orc_lpc_count_bits(bits_per_sample, res->qp.order, res->qp.precision, res->residual_bits);
}

/* src/coding.rs:476-484: mid = (l + r) >> 1, side = l - r */
void orc_stereo_to_midside(const int32_t* l, const int32_t* r, size_t n, int32_t* m, int32_t* s) {
  for (size_t t = 0; t < n; ++t) {
    m[t] = (int32_t)((uint32_t)l[t] + (uint32_t)r[t]) >> 1;
    s[t] = (int32_t)((uint32_t)l[t] - (uint32_t)r[t]);
  }
}

/* src/component/decode.rs:91-103 */
void orc_midside_to_stereo(const int32_t* m, const int32_t* s, size_t n, int32_t* l, int32_t* r) {
  for (size_t t = 0; t < n; ++t) {
    int32_t sv = s[t];
    int32_t mv = (int32_t)((uint32_t)m[t] << 1) + (sv & 1);
    l[t] = (mv + sv) >> 1;
    r[t] = (mv - sv) >> 1;
  }
}

/* arrayutils::is_constant, src/arrayutils.rs:382 */
int orc_is_constant(const int32_t* samples, size_t n) {
  for (size_t t = 1; t < n; ++t)
    if (samples[t] != samples[0]) return 0;
  return 1;
}

/* ------------------------------------------------------------------------ */
/* batch driver (threads = frame-parallel pool like src/par.rs:355-449)      */
/* ------------------------------------------------------------------------ */

typedef struct {
  const int32_t* samples;
  size_t begin, end, n, stride;
  const uint8_t* bps;
  uint32_t bps_const;
  const orc_qlpc_config* cfg;
  orc_subframe_record* recs;
  int32_t* residual;
  size_t residual_stride;
  double* autocorr;
  double* lpc_coefs;
} orc_batch_job;

static void* orc_batch_worker(void* arg) {
  orc_batch_job* job = (orc_batch_job*)arg;
  size_t n = job->n;
  int32_t* errors = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  uint8_t* rp = (uint8_t*)malloc(ORC_MAX_RICE_PARTITIONS);
  orc_qlpc_result res;
  for (size_t k = job->begin; k < job->end; ++k) {
    const int32_t* sig = job->samples + k * job->stride;
    uint32_t bps = job->bps ? job->bps[k] : job->bps_const;
    orc_estimated_qlpc(sig, n, bps, job->cfg, &res, rp, errors, NULL, NULL);
    if (job->recs) {
      orc_subframe_record* r = &job->recs[k];
      memset(r, 0, sizeof(*r));
      for (int i = 0; i < 32; ++i) r->coefs[i] = res.qp.coefs[i];
      r->order = (uint8_t)res.qp.order;
      r->shift = (int8_t)res.qp.shift;
      r->precision = (uint8_t)res.qp.precision;
      r->rice_order = (uint8_t)res.rice_order;
      r->status = res.status;
      r->code_bits = res.code_bits;
      r->subframe_bits = res.subframe_bits;
      r->sum_quotients = res.sum_quotients;
      size_t np = (size_t)1 << res.rice_order;
      memcpy(r->rice_params, rp, np < 256 ? np : 256);
    }
    if (job->residual) memcpy(job->residual + k * job->residual_stride, errors, sizeof(int32_t) * n);
    if (job->autocorr) memcpy(job->autocorr + k * 33, res.autocorr, sizeof(double) * 33);
    if (job->lpc_coefs) memcpy(job->lpc_coefs + k * 32, res.lpc_coefs, sizeof(double) * 32);
  }
  free(errors);
  free(rp);
  return NULL;
}

static void orc_run_batch(orc_batch_job* proto, size_t n_subframes, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n_subframes) nthreads = n_subframes ? (int)n_subframes : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  orc_batch_job* jobs = (orc_batch_job*)malloc(sizeof(orc_batch_job) * (size_t)nthreads);
  for (int i = 0; i < nthreads; ++i) {
    jobs[i] = *proto;
    jobs[i].begin = n_subframes * (size_t)i / (size_t)nthreads;
    jobs[i].end = n_subframes * (size_t)(i + 1) / (size_t)nthreads;
    if (nthreads == 1) orc_batch_worker(&jobs[i]);
    else pthread_create(&th[i], NULL, orc_batch_worker, &jobs[i]);
  }
  if (nthreads > 1)
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
}

void orc_qlpc_batch(const int32_t* samples, size_t n_subframes, size_t n, size_t stride,
                    const uint8_t* bps, const orc_qlpc_config* cfg, orc_subframe_record* recs,
                    int32_t* residual, size_t residual_stride, double* autocorr, double* lpc_coefs,
                    int nthreads) {
  orc_batch_job job;
  memset(&job, 0, sizeof(job));
  job.samples = samples;
  job.n = n;
  job.stride = stride;
  job.bps = bps;
  job.bps_const = 16;
  job.cfg = cfg;
  job.recs = recs;
  job.residual = residual;
  job.residual_stride = residual_stride;
  job.autocorr = autocorr;
  job.lpc_coefs = lpc_coefs;
  orc_run_batch(&job, n_subframes, nthreads);
}

/* cpu_baseline timing: `repeats` passes of estimated_qlpc over the batch with
 * `nthreads` workers; returns wall seconds.  Results are discarded except for
 * a checksum kept alive through a volatile sink. */
double orc_bench_qlpc(const int32_t* samples, size_t n_subframes, size_t n, size_t stride,
                      uint32_t bits_per_sample, const orc_qlpc_config* cfg, int nthreads,
                      int repeats) {
  orc_subframe_record* recs =
      (orc_subframe_record*)malloc(sizeof(orc_subframe_record) * (n_subframes ? n_subframes : 1));
  orc_batch_job job;
  memset(&job, 0, sizeof(job));
  job.samples = samples;
  job.n = n;
  job.stride = stride;
  job.bps = NULL;
  job.bps_const = bits_per_sample;
  job.cfg = cfg;
  job.recs = recs;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < repeats; ++r) orc_run_batch(&job, n_subframes, nthreads);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  static volatile uint64_t sink;
  uint64_t s = 0;
  for (size_t k = 0; k < n_subframes; ++k) s += recs[k].subframe_bits;
  sink = s;
  (void)sink;
  free(recs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* cpu_baseline for the stereo workload: per 2-channel frame the reference's
 * encode_frame runs the path four times -- L, R (encode_frame_impl, coding.rs:538)
 * and M, S formed by try_stereo_coding (coding.rs:476-491).  `frames` is batched
 * FrameBuf layout: channel c of frame f at frames + (2f + c)*stride. */
typedef struct {
  const int32_t* frames;
  size_t begin, end, n, stride;
  uint32_t bps;
  int repeats;
  const orc_qlpc_config* cfg;
  uint64_t checksum;
} orc_stereo_job;

static void* orc_stereo_worker(void* arg) {
  orc_stereo_job* job = (orc_stereo_job*)arg;
  size_t n = job->n;
  int32_t* errors = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* m = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* s = (int32_t*)malloc(sizeof(int32_t) * n);
  uint8_t* rp = (uint8_t*)malloc(ORC_MAX_RICE_PARTITIONS);
  orc_qlpc_result res;
  uint64_t sum = 0;
  for (int rep = 0; rep < job->repeats; ++rep)
  for (size_t f = job->begin; f < job->end; ++f) {
    const int32_t* l = job->frames + (2 * f) * job->stride;
    const int32_t* r = l + job->stride;
    orc_stereo_to_midside(l, r, n, m, s);
    orc_estimated_qlpc(l, n, job->bps, job->cfg, &res, rp, errors, NULL, NULL);
    sum += res.subframe_bits;
    orc_estimated_qlpc(r, n, job->bps, job->cfg, &res, rp, errors, NULL, NULL);
    sum += res.subframe_bits;
    orc_estimated_qlpc(m, n, job->bps, job->cfg, &res, rp, errors, NULL, NULL);
    sum += res.subframe_bits;
    orc_estimated_qlpc(s, n, job->bps + 1, job->cfg, &res, rp, errors, NULL, NULL);
    sum += res.subframe_bits;
  }
  job->checksum = sum;
  free(errors);
  free(m);
  free(s);
  free(rp);
  return NULL;
}

double orc_bench_stereo_qlpc(const int32_t* frames, size_t n_frames, size_t n, size_t stride,
                             uint32_t bits_per_sample, const orc_qlpc_config* cfg, int nthreads,
                             int repeats, uint64_t* checksum_out) {
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n_frames) nthreads = n_frames ? (int)n_frames : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  orc_stereo_job* jobs = (orc_stereo_job*)malloc(sizeof(orc_stereo_job) * (size_t)nthreads);
  struct timespec t0, t1;
  uint64_t total = 0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  { /* threads are spawned once; each makes `repeats` passes over its own frames */
    for (int i = 0; i < nthreads; ++i) {
      jobs[i].frames = frames;
      jobs[i].repeats = repeats;
      jobs[i].n = n;
      jobs[i].stride = stride;
      jobs[i].bps = bits_per_sample;
      jobs[i].cfg = cfg;
      jobs[i].begin = n_frames * (size_t)i / (size_t)nthreads;
      jobs[i].end = n_frames * (size_t)(i + 1) / (size_t)nthreads;
      jobs[i].checksum = 0;
      pthread_create(&th[i], NULL, orc_stereo_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i) {
      pthread_join(th[i], NULL);
      total += jobs[i].checksum;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (checksum_out) *checksum_out = total;
  free(th);
  free(jobs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------ */
/* fixed-LPC candidate, src/coding.rs:178-331                                 */
/* ------------------------------------------------------------------------ */

/* f32::log2 lowers to libm's log2f.  On the reference's Linux targets that is glibc
 * (>= 2.27: sysdeps/ieee754/flt-32/e_log2f.c + e_log2f_data.c, S. Nagy's table method; the
 * image has 2.35): 16-entry {1/c, log2 c} table, centre OFF = 0x3f330000, degree-4 polynomial
 * in double, one rounding to float at the end.  Restated here so that oracle and kernel share
 * one definition; tests/test_oracle_kat.py checks it against the host's log2f (exhaustively over
 * all 2^31 positive floats with --full-log2f: 0 mismatches with and without fma contraction). */
static const double orc_log2f_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static const double orc_log2f_poly[4] = {-0x1.712b6f70a7e4dp-2, 0x1.ecabf496832ep-2,
                                         -0x1.715479ffae3dep-1, 0x1.715475f35c8b8p0};

float orc_log2f(float x) {
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0) return -INFINITY;           /* log2(+-0) */
    if (ix == 0x7f800000u) return x;              /* log2(inf) */
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN; /* x < 0 or NaN */
    float xs = x * 0x1p23f;                       /* subnormal: normalise */
    memcpy(&ix, &xs, 4);
    ix -= 23u << 23;
  }
  uint32_t tmp = ix - 0x3f330000u;
  int i = (int)((tmp >> 19) & 15u);
  uint32_t top = tmp & 0xff800000u;
  uint32_t iz = ix - top;
  int k = (int32_t)tmp >> 23;
  float zf;
  memcpy(&zf, &iz, 4);
  double z = (double)zf;
  double r = z * orc_log2f_tab[i][0] - 1.0;
  double y0 = orc_log2f_tab[i][1] + (double)k;
  double r2 = r * r;
  double y = orc_log2f_poly[1] * r + orc_log2f_poly[2];
  y = orc_log2f_poly[0] * r2 + y;
  double p = orc_log2f_poly[3] * r + y0;
  y = y * r2 + p;
  return (float)y;
}

/* reset_fixed_lpc_errors, src/coding.rs:182-197: errors[0] = signal, errors[k+1][t] =
 * errors[k][t] - errors[k][t-1] with errors[k][-1] = 0 (carry starts at 0), wrapping i32 lanes.
 * The first k entries of errors[k] are therefore partial differences, not zeros. */
void orc_reset_fixed_lpc_errors(const int32_t* signal, size_t n, int32_t* errors) {
  memcpy(errors, signal, sizeof(int32_t) * n);
  for (int order = 0; order < 4; ++order) {
    const int32_t* cur = errors + (size_t)order * n;
    int32_t* next = errors + (size_t)(order + 1) * n;
    int32_t carry = 0;
    for (size_t t = 0; t < n; ++t) {
      next[t] = (int32_t)((uint32_t)cur[t] - (uint32_t)carry);
      carry = cur[t];
    }
  }
}

static inline float orc_abs_as_f32(int32_t x) {
  int32_t a = (int32_t)(x < 0 ? 0u - (uint32_t)x : (uint32_t)x); /* i32::abs, wrapping in release */
  return (float)a;
}

/* arrayutils::find_sum_abs_f32::<16>, src/arrayutils.rs:496-506 via simd_map_and_reduce
 * (:459-493).  ORC_SUMABS_STABLE: slice_as_simd = (data, [], []) (:435-438), one sequential f32
 * chain.  ORC_SUMABS_NIGHTLY: `as_simd` splits at 64-byte boundaries (`base_mod` = element
 * offset of data[0] from one; SimdVec storage is 64-byte aligned), 16 f32 lane sums + head/foot
 * scalar chain, acc + ordered lane sum.  ORC_SUMABS_CANONICAL (the build's definition): the
 * exact integer sum rounded to f32 once -- equal to both reference orders whenever every
 * partial sum is < 2^24 (all 16-bit material at the default 16 partitions for orders 0-1,
 * and any quiet signal), and inside their mutual spread otherwise. */
float orc_find_sum_abs_f32(const int32_t* data, size_t len, int mode, size_t base_mod) {
  if (mode == ORC_SUMABS_CANONICAL) {
    uint64_t acc = 0;
    for (size_t t = 0; t < len; ++t) acc += data[t] < 0 ? (uint64_t)(-(int64_t)data[t]) : (uint64_t)data[t];
    return (float)acc; /* round to nearest even, once */
  }
  if (mode == ORC_SUMABS_STABLE) {
    float acc = 0.0f;
    for (size_t t = 0; t < len; ++t) acc = orc_abs_as_f32(data[t]) + acc;
    return acc + 0.0f;
  }
  size_t mis = base_mod % 16;
  size_t head = mis ? 16 - mis : 0;
  if (head > len) head = len;
  size_t nbody = (len - head) / 16;
  float acc = 0.0f, acc_v[16];
  for (int l = 0; l < 16; ++l) acc_v[l] = 0.0f;
  size_t t = 0;
  for (; t < head; ++t) acc = orc_abs_as_f32(data[t]) + acc;
  for (size_t b = 0; b < nbody; ++b, t += 16)
    for (int l = 0; l < 16; ++l) acc_v[l] = orc_abs_as_f32(data[t + l]) + acc_v[l];
  for (; t < len; ++t) acc = orc_abs_as_f32(data[t]) + acc;
  float lanes = 0.0f;
  for (int l = 0; l < 16; ++l) lanes += acc_v[l];
  return acc + lanes;
}

/* Rust `f32 as usize`: saturating, NaN -> 0 */
static inline uint64_t orc_f32_as_usize(float v) {
  if (!(v > 0.0f)) return 0;
  if (v >= 18446744073709551616.0f) return UINT64_MAX;
  return (uint64_t)v;
}

/* estimate_entropy, src/coding.rs:200-227 */
uint64_t orc_estimate_entropy(const int32_t* errors, size_t n, size_t warmup_len, size_t partitions,
                              int mode) {
  size_t partition_size = (n + partitions - 1) / partitions;
  size_t offset = 0;
  uint64_t acc = 0;
  for (size_t p = 0; p < partitions; ++p) {
    size_t end = offset + partition_size < n ? offset + partition_size : n;
    size_t partition_len = end - offset;
    if (end >= warmup_len) {
      size_t sample_count = end - warmup_len < partition_len ? end - warmup_len : partition_len;
      float sum_errors = orc_find_sum_abs_f32(errors + offset, partition_len, mode, offset);
      float avg_errors = sum_errors * 2.0f / ((float)sample_count + 0.00001f);
      float geom_p = 1.0f / (avg_errors + 1.0f);
      float xent = fmaf(avg_errors, -orc_log2f(1.0f - geom_p), -orc_log2f(geom_p));
      acc += orc_f32_as_usize(xent * (float)sample_count);
    }
    offset = end;
  }
  return acc;
}

/* FIXED_LPC_COEFS, src/component/decode.rs:179-185 */
const int16_t orc_fixed_lpc_coefs[5][4] = {
    {0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};

/* fixed_lpc (src/coding.rs:298-331) = reset_fixed_lpc_errors + select_order_and_encode_residual
 * (:230-288).  Returns 1 and fills `res` / `rice_params` / `errors_out` (the chosen order's error
 * signal with the warm-up slots zeroed, as Residual stores them: coding.rs:151-160) when a
 * candidate is produced (selector key < baseline_bits), 0 otherwise.  `res->estimate[k]` is the
 * selector's key for order k either way.  min_by_key keeps the FIRST minimum. */
int orc_fixed_lpc(const int32_t* signal, size_t n, uint32_t bps, uint64_t baseline_bits,
                  const orc_fixed_config* fc, uint32_t max_rice_p, orc_fixed_result* res,
                  uint8_t* rice_params, int32_t* errors_out) {
  memset(res, 0, sizeof(*res));
  int32_t* errors = (int32_t*)malloc(sizeof(int32_t) * n * 5);
  orc_reset_fixed_lpc_errors(signal, n, errors);
  static __thread orc_prc_parameter prc_storage, best_prc_storage;
  orc_prc_parameter* prc = &prc_storage;
  orc_prc_parameter* best_prc = &best_prc_storage;
  uint32_t best_order = 0;
  uint64_t best_bits = UINT64_MAX;
  int have = 0;
  for (uint32_t order = 0; order <= fc->max_order && order <= 4; ++order) {
    const int32_t* err = errors + (size_t)order * n;
    uint64_t bits;
    if (fc->order_sel == ORC_ORDERSEL_BITCOUNT) {
      orc_find_partitioned_rice_parameter(err, n, order, max_rice_p, prc);
      bits = (uint64_t)bps * order + prc->code_bits;
    } else {
      bits = orc_estimate_entropy(err, n, order, fc->partitions, (int)fc->sum_mode) + (uint64_t)bps * order;
    }
    res->estimate[order] = bits;
    if (!have || bits < best_bits) {
      have = 1;
      best_bits = bits;
      best_order = order;
      if (fc->order_sel == ORC_ORDERSEL_BITCOUNT) {
        best_prc->order = prc->order;
        best_prc->code_bits = prc->code_bits;
        memcpy(best_prc->ps, prc->ps, (size_t)1 << prc->order);
      }
    }
  }
  res->order = best_order;
  int selected = have && best_bits < baseline_bits;
  if (selected) {
    const int32_t* err = errors + (size_t)best_order * n;
    if (fc->order_sel != ORC_ORDERSEL_BITCOUNT)
      orc_find_partitioned_rice_parameter(err, n, best_order, max_rice_p, best_prc);
    res->rice_order = best_prc->order;
    res->code_bits = best_prc->code_bits;
    memcpy(rice_params, best_prc->ps, (size_t)1 << best_prc->order);
    orc_encode_residual_with_prc_parameter(err, n, best_order, best_prc, NULL, NULL,
                                           &res->sum_quotients, &res->sum_rice_params);
    res->residual_bits = orc_residual_count_bits(n, best_order, best_prc->order, best_prc->ps,
                                                 res->sum_quotients, res->sum_rice_params);
    /* BitRepr for FixedLpc::count_bits, src/component/bitrepr.rs:473-477 */
    res->subframe_bits = 8 + (uint64_t)bps * best_order + res->residual_bits;
    if (errors_out) {
      memcpy(errors_out, err, sizeof(int32_t) * n);
      for (size_t t = 0; t < best_order && t < n; ++t) errors_out[t] = 0;
    }
  }
  res->selected = selected;
  free(errors);
  return selected;
}

/* ------------------------------------------------------------------------ */
/* encode_subframe / try_stereo_coding restricted to the candidates the GPU  */
/* path produces (use_fixed = false)                                         */
/* ------------------------------------------------------------------------ */

/* encode_subframe, src/coding.rs:384-418 with config.use_fixed == false:
 * Constant if use_constant && is_constant; else the LPC candidate iff
 * use_lpc && !too_short && count_bits < verbatim_bits; else Verbatim.
 * kind: 0 Constant, 1 Verbatim, 3 Lpc.  `lpc` receives estimated_qlpc's result when
 * it was evaluated (errors = its residual). */
int orc_encode_subframe_nofixed(const int32_t* samples, size_t n, uint32_t bps, int use_constant,
                                int use_lpc, const orc_qlpc_config* cfg, uint64_t* bits_out,
                                orc_qlpc_result* lpc, uint8_t* rice_params, int32_t* errors) {
  orc_frame_config fc;
  memset(&fc, 0, sizeof(fc));
  fc.qlpc = *cfg;
  fc.use_constant = (uint32_t)use_constant;
  fc.use_lpc = (uint32_t)use_lpc;
  orc_fixed_result fixed;
  return orc_encode_subframe(samples, n, bps, &fc, bits_out, lpc, &fixed, rice_params, errors);
}

/* encode_subframe, src/coding.rs:384-418, all candidates.  kind: 0 Constant, 1 Verbatim,
 * 2 FixedLpc, 3 Lpc.  `rice_params` / `errors` receive the chosen candidate's Rice parameters and
 * error signal (kind 2 or 3). */
int orc_encode_subframe(const int32_t* samples, size_t n, uint32_t bps, const orc_frame_config* fc,
                        uint64_t* bits_out, orc_qlpc_result* lpc, orc_fixed_result* fixed,
                        uint8_t* rice_params, int32_t* errors) {
  memset(fixed, 0, sizeof(*fixed));
  if (fc->use_constant && orc_is_constant(samples, n)) {
    *bits_out = 8 + bps; /* Constant::count_bits, bitrepr.rs:445 */
    return 0;
  }
  uint64_t verbatim_bits = orc_verbatim_count_bits(n, bps);
  int too_short = n < 64; /* MIN_BLOCK_SIZE_FOR_PREDICTION, constant.rs:51 */
  int have_fixed = 0;
  uint8_t* fixed_rp = NULL;
  int32_t* fixed_err = NULL;
  if (!too_short && fc->use_fixed) {
    fixed_rp = (uint8_t*)calloc(ORC_MAX_RICE_PARTITIONS, 1);
    fixed_err = (int32_t*)calloc(n, sizeof(int32_t));
    have_fixed = orc_fixed_lpc(samples, n, bps, verbatim_bits, &fc->fixed, fc->qlpc.max_rice_parameter,
                               fixed, fixed_rp, fixed_err);
  }
  uint64_t baseline_bits = verbatim_bits;
  if (have_fixed && fixed->subframe_bits < baseline_bits) baseline_bits = fixed->subframe_bits;
  int kind = 1;
  uint64_t bits = verbatim_bits;
  int have_lpc = 0;
  if (!too_short && fc->use_lpc) {
    orc_estimated_qlpc(samples, n, bps, &fc->qlpc, lpc, rice_params, errors, NULL, NULL);
    have_lpc = lpc->status == ORC_STATUS_OK && lpc->subframe_bits < baseline_bits;
  }
  /* est_lpc.or(fixed).filter(|sf| sf.count_bits() < verbatim_bits) */
  if (have_lpc) {
    if (lpc->subframe_bits < verbatim_bits) { kind = 3; bits = lpc->subframe_bits; }
  } else if (have_fixed) {
    if (fixed->subframe_bits < verbatim_bits) {
      kind = 2;
      bits = fixed->subframe_bits;
      memcpy(rice_params, fixed_rp, (size_t)1 << fixed->rice_order);
      memcpy(errors, fixed_err, sizeof(int32_t) * n);
    }
  }
  free(fixed_rp);
  free(fixed_err);
  *bits_out = bits;
  return kind;
}

/* encode_frame for 2 channels (src/coding.rs:530-544) = encode_frame_impl(Independent(2)) +
 * try_stereo_coding (:469-527).  Outputs mirror flacenc_hip_stereo_frame_result. */
void orc_encode_stereo_frame(const int32_t* l, const int32_t* r, size_t n, uint32_t bps,
                             const orc_qlpc_config* cfg, int use_constant, int use_lpc,
                             int use_leftside, int use_rightside, int use_midside,
                             orc_stereo_frame_result* out, int32_t* residual0, int32_t* residual1) {
  orc_frame_config fc;
  memset(&fc, 0, sizeof(fc));
  fc.qlpc = *cfg;
  fc.use_constant = (uint32_t)use_constant;
  fc.use_lpc = (uint32_t)use_lpc;
  fc.use_leftside = (uint32_t)use_leftside;
  fc.use_rightside = (uint32_t)use_rightside;
  fc.use_midside = (uint32_t)use_midside;
  orc_encode_stereo_frame_cfg(l, r, n, bps, &fc, out, residual0, residual1);
}

void orc_encode_stereo_frame_cfg(const int32_t* l, const int32_t* r, size_t n, uint32_t bps,
                                 const orc_frame_config* fc, orc_stereo_frame_result* out,
                                 int32_t* residual0, int32_t* residual1) {
  int32_t* m = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* s = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* err[4];
  uint8_t* rp[4];
  orc_qlpc_result res[4];
  orc_fixed_result fres[4];
  int kind[4];
  uint64_t bits[4];
  const int32_t* sig[4];
  orc_stereo_to_midside(l, r, n, m, s);
  sig[0] = l; sig[1] = r; sig[2] = m; sig[3] = s;
  for (int k = 0; k < 4; ++k) {
    err[k] = (int32_t*)calloc(n, sizeof(int32_t));
    rp[k] = (uint8_t*)calloc(ORC_MAX_RICE_PARTITIONS, 1);
    memset(&res[k], 0, sizeof(res[k]));
    kind[k] = orc_encode_subframe(sig[k], n, bps + (k == 3 ? 1 : 0), fc, &bits[k], &res[k], &fres[k],
                                  rp[k], err[k]);
  }
  /* src/coding.rs:493-522 */
  uint64_t min_bits = bits[0] + bits[1];
  int assignment = 0; /* Independent(2) */
  if (fc->use_leftside && bits[0] + bits[3] < min_bits) { min_bits = bits[0] + bits[3]; assignment = 1; }
  if (fc->use_rightside && bits[1] + bits[3] < min_bits) { min_bits = bits[1] + bits[3]; assignment = 2; }
  if (fc->use_midside && bits[2] + bits[3] < min_bits) { min_bits = bits[2] + bits[3]; assignment = 3; }
  /* ChannelAssignment::select_channels, datatype.rs:1173-1185 */
  int role0 = assignment == 2 ? 3 : (assignment == 3 ? 2 : 0);
  int role1 = assignment == 0 ? 1 : (assignment == 2 ? 1 : 3);
  memset(out, 0, sizeof(*out));
  out->channel_assignment = (uint8_t)assignment;
  int roles[2] = {role0, role1};
  int32_t* resid_out[2] = {residual0, residual1};
  for (int c = 0; c < 2; ++c) {
    int k = roles[c];
    out->role[c] = (uint8_t)k;
    out->kind[c] = (uint8_t)kind[k];
    out->dc_offset[c] = kind[k] == 0 ? sig[k][0] : 0;
    orc_subframe_record* rec = &out->lpc[c];
    if (kind[k] == 3) {
      for (int i = 0; i < 32; ++i) rec->coefs[i] = res[k].qp.coefs[i];
      rec->order = (uint8_t)res[k].qp.order;
      rec->shift = (int8_t)res[k].qp.shift;
      rec->precision = (uint8_t)res[k].qp.precision;
      rec->rice_order = (uint8_t)res[k].rice_order;
      rec->status = res[k].status;
      rec->code_bits = res[k].code_bits;
      rec->subframe_bits = res[k].subframe_bits;
      rec->sum_quotients = res[k].sum_quotients;
      memcpy(rec->rice_params, rp[k], (size_t)1 << res[k].rice_order);
      if (resid_out[c]) memcpy(resid_out[c], err[k], sizeof(int32_t) * n);
    } else if (kind[k] == 2) {
      /* FixedLpc: the record carries FIXED_LPC_COEFS[order] with shift 0 (decode.rs:187-201) */
      for (int i = 0; i < 4; ++i) rec->coefs[i] = orc_fixed_lpc_coefs[fres[k].order][i];
      rec->order = (uint8_t)fres[k].order;
      rec->shift = 0;
      rec->precision = 0;
      rec->rice_order = (uint8_t)fres[k].rice_order;
      rec->status = 0;
      rec->code_bits = fres[k].code_bits;
      rec->subframe_bits = fres[k].subframe_bits;
      rec->sum_quotients = fres[k].sum_quotients;
      memcpy(rec->rice_params, rp[k], (size_t)1 << fres[k].rice_order);
      if (resid_out[c]) memcpy(resid_out[c], err[k], sizeof(int32_t) * n);
    } else if (resid_out[c]) {
      memset(resid_out[c], 0, sizeof(int32_t) * n);
    }
  }
  for (int k = 0; k < 4; ++k) out->bits[k] = bits[k];
  for (int k = 0; k < 4; ++k) { free(err[k]); free(rp[k]); }
  free(m);
  free(s);
}

/* ------------------------------------------------------------------------ */
/* bit writer: src/component/bitrepr.rs (Frame, FrameHeader, SubFrame,        */
/* Residual), MSB-first like bitsink::MemSink                                 */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint8_t* buf;
  size_t cap;    /* bytes */
  size_t bitpos; /* bits written; bits beyond are zero */
} orc_bitsink;

static void orc_sink_put(orc_bitsink* s, uint64_t val, unsigned nbits) {
  for (unsigned i = 0; i < nbits; ++i) {
    size_t b = s->bitpos++;
    if ((b >> 3) >= s->cap) continue; /* counted, not stored */
    if ((val >> (nbits - 1 - i)) & 1u) s->buf[b >> 3] |= (uint8_t)(0x80u >> (b & 7));
  }
}
static void orc_sink_zeros(orc_bitsink* s, size_t n) { s->bitpos += n; }
/* BitSink::write_twoc: the low `bits` bits of the two's complement value */
static void orc_sink_twoc(orc_bitsink* s, int32_t v, unsigned bits) {
  orc_sink_put(s, (uint64_t)(uint32_t)v & ((bits >= 32) ? 0xFFFFFFFFull : ((1ull << bits) - 1)), bits);
}

/* crc::CRC_8_SMBUS (poly 0x07, init 0, no reflection, xorout 0), bitrepr.rs:39 */
uint8_t orc_crc8(const uint8_t* data, size_t len) {
  uint8_t crc = 0;
  for (size_t i = 0; i < len; ++i) {
    crc ^= data[i];
    for (int b = 0; b < 8; ++b) crc = (uint8_t)((crc & 0x80) ? (crc << 1) ^ 0x07 : (crc << 1));
  }
  return crc;
}
/* crc::CRC_16_UMTS (poly 0x8005, init 0, no reflection, xorout 0), bitrepr.rs:40 */
uint16_t orc_crc16(const uint8_t* data, size_t len) {
  uint16_t crc = 0;
  for (size_t i = 0; i < len; ++i) {
    crc ^= (uint16_t)((uint16_t)data[i] << 8);
    for (int b = 0; b < 8; ++b) crc = (uint16_t)((crc & 0x8000) ? (crc << 1) ^ 0x8005 : (crc << 1));
  }
  return crc;
}

/* encode_to_utf8like, bitrepr.rs:108-154; returns the byte count (0 if val needs > 36 bits) */
size_t orc_encode_to_utf8like(uint64_t val, uint8_t out[7]) {
  static const uint8_t heads[7] = {0x80, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC, 0xFE};
  unsigned code_bits = 0;
  while (code_bits < 64 && (val >> code_bits) != 0) ++code_bits;
  if (code_bits <= 7) {
    out[0] = (uint8_t)val;
    return 1;
  }
  if (code_bits > 36) return 0;
  unsigned trailing = (code_bits - 2) / 5;
  unsigned capacity = trailing * 6 + 6 - trailing;
  unsigned first_bits = 6 - trailing;
  uint64_t v = val << (64 - capacity);
  out[0] = trailing == 6 ? 0xFE : (uint8_t)(heads[trailing] | (first_bits ? ((v >> (64 - first_bits)) & 0xFF) : 0));
  v = first_bits ? (v << first_bits) : v;
  for (unsigned i = 0; i < trailing; ++i) {
    out[1 + i] = (uint8_t)(0x80u | (uint8_t)(v >> 58));
    v <<= 6;
  }
  return 1 + trailing;
}

/* BlockSizeSpec::from_size + tag + extra bits, datatype.rs:1239-1294 */
static void orc_block_size_spec(uint32_t size, uint32_t* tag, uint32_t* extra_bits, uint32_t* extra) {
  *extra_bits = 0;
  *extra = 0;
  if (size == 192) { *tag = 1; return; }
  for (uint32_t x = 0; x < 4; ++x)
    if (size == (576u << x)) { *tag = 2 + x; return; }
  for (uint32_t x = 0; x < 8; ++x)
    if (size == (256u << x)) { *tag = 8 + x; return; }
  if (size <= 256) { *tag = 6; *extra_bits = 8; *extra = (size - 1) & 0xFF; return; }
  *tag = 7;
  *extra_bits = 16;
  *extra = (size - 1) & 0xFFFF;
}

/* SampleRateSpec::from_freq + tag + extra bits, datatype.rs:1427-1453, 1503-1543;
 * encode_frame_impl falls back to Unspecified (coding.rs:434-435) */
static void orc_sample_rate_spec(uint32_t freq, uint32_t* tag, uint32_t* extra_bits, uint32_t* extra) {
  static const uint32_t known[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
  *extra_bits = 0;
  *extra = 0;
  for (uint32_t t = 1; t < 12; ++t)
    if (freq == known[t]) { *tag = t; return; }
  if (freq % 1000 == 0 && freq / 1000 <= 255) { *tag = 12; *extra_bits = 8; *extra = freq / 1000; return; }
  if (freq % 10 == 0 && freq / 10 <= 65535) { *tag = 14; *extra_bits = 16; *extra = freq / 10; return; }
  if (freq <= 65535) { *tag = 13; *extra_bits = 16; *extra = freq; return; }
  *tag = 0;
}

/* SampleSizeSpec::from_bits / into_tag, datatype.rs:1304-1360 */
static uint32_t orc_sample_size_tag(uint32_t bits) {
  switch (bits) {
    case 8: return 1;
    case 12: return 2;
    case 16: return 4;
    case 20: return 5;
    case 24: return 6;
    case 32: return 7;
    default: return 0;
  }
}

/* BitRepr for FrameHeader::write, bitrepr.rs:373-419.  `variable` selects the blocking strategy
 * and what `offset` means (FrameOffset::StartSample / ::Frame); `sample_rate` = 0 and
 * `bits_per_sample` = 0 give the Unspecified specs.  Returns the header length in bytes. */
size_t orc_write_frame_header(uint32_t block_size, uint32_t channel_tag, uint32_t bits_per_sample,
                              uint32_t sample_rate, int variable, uint64_t offset, uint8_t* out) {
  uint8_t hdr[24];
  size_t n = 0;
  uint32_t bs_tag, bs_bits, bs_extra, sr_tag, sr_bits, sr_extra;
  orc_block_size_spec(block_size, &bs_tag, &bs_bits, &bs_extra);
  if (sample_rate) orc_sample_rate_spec(sample_rate, &sr_tag, &sr_bits, &sr_extra);
  else { sr_tag = 0; sr_bits = 0; sr_extra = 0; }
  uint32_t header_word = 0xFFF8u + (variable ? 1u : 0u);
  hdr[n++] = (uint8_t)(header_word >> 8);
  hdr[n++] = (uint8_t)header_word;
  hdr[n++] = (uint8_t)((bs_tag << 4) | sr_tag);
  hdr[n++] = (uint8_t)((channel_tag << 4) | (orc_sample_size_tag(bits_per_sample) << 1));
  n += orc_encode_to_utf8like(offset, hdr + n);
  if (bs_bits == 8) hdr[n++] = (uint8_t)bs_extra;
  if (bs_bits == 16) { hdr[n++] = (uint8_t)(bs_extra >> 8); hdr[n++] = (uint8_t)bs_extra; }
  if (sr_bits == 8) hdr[n++] = (uint8_t)sr_extra;
  if (sr_bits == 16) { hdr[n++] = (uint8_t)(sr_extra >> 8); hdr[n++] = (uint8_t)sr_extra; }
  hdr[n] = orc_crc8(hdr, n);
  ++n;
  memcpy(out, hdr, n);
  return n;
}

/* BitRepr for Residual::write, bitrepr.rs:550-597 */
static void orc_write_residual(orc_bitsink* s, const int32_t* errors, size_t n, size_t warmup,
                               uint32_t partition_order, const uint8_t* rice_params) {
  size_t nparts = (size_t)1 << partition_order;
  int rice2 = 0;
  for (size_t q = 0; q < nparts; ++q) rice2 |= rice_params[q] > 14;
  orc_sink_put(s, ((uint64_t)(rice2 ? 1 : 0) << 4) | partition_order, 6);
  size_t part_len = n >> partition_order, offset = 0;
  for (size_t q = 0; q < nparts; ++q) {
    uint32_t p = rice_params[q];
    orc_sink_put(s, p, rice2 ? 5 : 4);
    size_t start = warmup > offset ? warmup : offset;
    offset += part_len;
    for (size_t t = start; t < offset; ++t) {
      uint32_t u = orc_encode_signbit(errors[t]);
      orc_sink_zeros(s, u >> p);                                              /* unary quotient */
      orc_sink_put(s, (uint64_t)((u & ((1u << p) - 1u)) | (1u << p)), p + 1); /* stop bit + remainder */
    }
  }
}

/* BitRepr for SubFrame::write: Constant bitrepr.rs:449-454, Verbatim :463-470, FixedLpc :479-487,
 * Lpc :501-527.  Returns the bits written. */
static size_t orc_write_subframe_sink(orc_bitsink* s, const orc_subframe_desc* d, size_t n) {
  size_t start = s->bitpos;
  if (d->kind == 0) {
    orc_sink_put(s, 0x00, 8);
    orc_sink_twoc(s, d->dc_offset, d->bps);
  } else if (d->kind == 1) {
    orc_sink_put(s, 0x02, 8);
    for (size_t t = 0; t < n; ++t) orc_sink_twoc(s, d->samples[t], d->bps);
  } else if (d->kind == 2) {
    orc_sink_put(s, 0x10u | (d->order << 1), 8);
    for (uint32_t t = 0; t < d->order; ++t) orc_sink_twoc(s, d->samples[t], d->bps);
    orc_write_residual(s, d->residual, n, d->order, d->rice_order, d->rice_params);
  } else {
    orc_sink_put(s, 0x40u | ((d->order - 1) << 1), 8);
    for (uint32_t t = 0; t < d->order; ++t) orc_sink_twoc(s, d->samples[t], d->bps);
    orc_sink_put(s, d->precision - 1, 4);
    orc_sink_twoc(s, d->shift, 5);
    for (uint32_t t = 0; t < d->order; ++t) orc_sink_twoc(s, d->coefs[t], d->precision);
    orc_write_residual(s, d->residual, n, d->order, d->rice_order, d->rice_params);
  }
  return s->bitpos - start;
}

size_t orc_write_subframe(const orc_subframe_desc* d, size_t n, uint8_t* out, size_t cap) {
  memset(out, 0, cap);
  orc_bitsink s = {out, cap, 0};
  return orc_write_subframe_sink(&s, d, n);
}

/* BitRepr for Frame::write, bitrepr.rs:289-319, for a frame made by encode_fixed_size_frame
 * (coding.rs:581-606: fixed blocking, FrameOffset::Frame).  channel_assignment: 0 Independent
 * (nch), 1 LeftSide, 2 RightSide, 3 MidSide (ChannelAssignment::write, bitrepr.rs:329-356).
 * Returns the frame length in bytes (the bytes beyond `cap` are counted but not stored). */
size_t orc_write_frame(uint32_t block_size, uint32_t channel_assignment, uint32_t nch,
                       uint32_t bits_per_sample, uint32_t sample_rate, uint32_t frame_number,
                       const orc_subframe_desc* subframes, uint8_t* out, size_t cap) {
  memset(out, 0, cap);
  uint32_t channel_tag = channel_assignment == 0 ? nch - 1 : 7 + channel_assignment;
  uint8_t hdr[24];
  size_t hn = orc_write_frame_header(block_size, channel_tag, bits_per_sample, sample_rate ? sample_rate : 0, 0,
                                     frame_number, hdr);
  memcpy(out, hdr, hn < cap ? hn : cap);
  orc_bitsink s = {out, cap, hn * 8};
  for (uint32_t c = 0; c < nch; ++c) orc_write_subframe_sink(&s, &subframes[c], block_size);
  size_t nbytes = (s.bitpos + 7) >> 3; /* align_to_byte */
  if (nbytes + 2 <= cap) {
    uint16_t crc = orc_crc16(out, nbytes);
    out[nbytes] = (uint8_t)(crc >> 8);
    out[nbytes + 1] = (uint8_t)crc;
  }
  return nbytes + 2;
}

/* the frame encode_frame would write for one orc_stereo_frame_result (2 channels) */
size_t orc_write_stereo_frame(const orc_stereo_frame_result* fr, const int32_t* l, const int32_t* r, size_t n,
                              uint32_t bits_per_sample, uint32_t sample_rate, uint32_t frame_number,
                              const int32_t* residual0, const int32_t* residual1, uint8_t* out, size_t cap) {
  int32_t* m = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* sd = (int32_t*)malloc(sizeof(int32_t) * n);
  orc_stereo_to_midside(l, r, n, m, sd);
  const int32_t* sig[4] = {l, r, m, sd};
  const int32_t* resid[2] = {residual0, residual1};
  orc_subframe_desc d[2];
  for (int c = 0; c < 2; ++c) {
    const orc_subframe_record* rec = &fr->lpc[c];
    d[c].kind = fr->kind[c];
    d[c].bps = bits_per_sample + (fr->role[c] == 3 ? 1u : 0u);
    d[c].dc_offset = fr->dc_offset[c];
    d[c].samples = sig[fr->role[c]];
    d[c].order = rec->order;
    d[c].shift = rec->shift;
    d[c].precision = rec->precision;
    d[c].coefs = rec->coefs;
    d[c].rice_order = rec->rice_order;
    d[c].rice_params = rec->rice_params;
    d[c].residual = resid[c];
  }
  size_t len = orc_write_frame((uint32_t)n, fr->channel_assignment, 2, bits_per_sample, sample_rate, frame_number,
                               d, out, cap);
  free(m);
  free(sd);
  return len;
}

/* ------------------------------------------------------------------------ */
/* input side: src/arrayutils.rs                                              */
/* ------------------------------------------------------------------------ */

/* le_bytes_to_i32s_impl, src/arrayutils.rs:273-290 */
void orc_le_bytes_to_i32s(const uint8_t* bytes, size_t nbytes, int32_t* dest, uint32_t bytes_per_sample) {
  size_t n = 0;
  for (size_t t = 0; t + bytes_per_sample <= nbytes; t += bytes_per_sample) {
    uint32_t u = 0;
    for (uint32_t i = 0; i < bytes_per_sample; ++i) u |= (uint32_t)bytes[t + i] << (8u * (i + 4u - bytes_per_sample));
    dest[n++] = (int32_t)u >> ((4u - bytes_per_sample) * 8u);
  }
}

/* deinterleave_gen, src/arrayutils.rs:229-245 (all channel counts agree with it, :248-264) */
void orc_deinterleave(const int32_t* interleaved, size_t len, size_t channels, size_t channel_stride,
                      int32_t* dest) {
  size_t samples = len / channels;
  for (size_t c = 0; c < channels; ++c) {
    for (size_t t = 0; t < samples; ++t) dest[c * channel_stride + t] = interleaved[t * channels + c];
    for (size_t t = samples; t < channel_stride; ++t) dest[c * channel_stride + t] = 0;
  }
}
