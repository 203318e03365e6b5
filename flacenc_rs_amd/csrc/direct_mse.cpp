// direct_mse.cpp -- the reference's EXPERIMENTAL estimators of perform_qlpc (src/coding.rs:333-351):
//
//   LpcEstimator::weighted_lpc_with_direct_mse  src/lpc.rs:853-903  covariance-method LPC
//     weighted_auto_correlation_nosimd          src/lpc.rs:533-548  right-hand side R[1..=P]
//     weighted_lagged_outer_prod_sum            src/lpc.rs:573-600  P x P Gram matrix of lagged vectors
//     LpcFloat::solve_sym_mut                   src/lpc.rs:79-87    Cholesky (nalgebra) + regulariser doubling
//   LpcEstimator::lpc_with_irls_mae             src/lpc.rs:814-850  IRLS towards the mean absolute error
//     compute_raw_errors                        src/lpc.rs:602-618
//
// One workgroup per subframe.  The windowed block is staged in LDS as f32; every entry the solve needs is one
// SEQUENTIAL fma chain over time in the reference -- R[tau] over t = P..n-1, G[i][j] (i <= j) over t = P-1..n-2
// of the block without its last sample -- and with y_a(t') = x_w[t' + 1 - a] both are entries of one
// (P+1) x (P+1) matrix H[a][b] = sum_{t' = P-1}^{n-2} fma(y_a, f32(w[t' + 1] * y_b), .): R[tau] = H[tau][0],
// G[i][j] = H[i + 1][j + 1].  A chain cannot be split without changing its roundings -- but H is a dense contraction
// over time with every output distinct, and v_mfma_f64_16x16x4_f64 chained through its C operand performs exactly
// the chain's operations in the chain's order (measured, see the kernel): a WAVE owns a 16 x 16 tile of H.
// The factorisation then runs on wave 0 with a lane per matrix row (the column updates of nalgebra's
// left-looking Cholesky are independent across rows, so lanes change nothing in any element's operation
// sequence), the triangular solves follow nalgebra's loops, including dotx's eight partial accumulators.
// Quantisation (lpc.rs:234-302) is done here as well: the kernel's output is the predictor record the
// residual kernels of the split pipeline take (`pred`), as levinson_batch_kernel's is.
//
// f32::powf in the IRLS weight (lpc.rs:828) is libm's powf: glibc's algorithm (sysdeps/ieee754/flt-32/e_powf.c:
// log2 by a 16-entry table + degree-5 polynomial and exp2 by a 32-entry table + cubic, both in double, one
// rounding to float) restated; checked against the host libm for y = -1.2 over every float in [0.0099, 1e6]
// (223 M arguments, no mismatch, with and without fma contraction).
//
// PARITY: bit-equal to the oracle's restatement of the same algorithms (tests/test_gpu_direct_mse.py).  The
// solver is a third-party crate outside the reference tree, the reference's tests of this path are qualitative:
// beyond those tests (restated in tests/test_oracle_kat.py) this row is "parity unpinned".
#include "direct_mse.h"

#include "flacenc_hip.h"
#include "lds_opt_in.h"

namespace flacenc_hip {
namespace {

// exact ceil(log2(m)) for finite m > 0 from the exponent/mantissa fields (as in qlpc_kernel_impl.h)
__device__ __forceinline__ int dm_ceil_log2_pos(double m) {
  const uint64_t b = (uint64_t)__double_as_longlong(m);
  const int e = (int)((b >> 52) & 0x7FF);
  const uint64_t frac = b & 0xFFFFFFFFFFFFFull;
  if (e == 0) return -32752;
  return (e - 1023) + (frac != 0 ? 1 : 0);
}

__device__ const double kPowLog2Tab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
    0x1.49539f0f010bp+0,  -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
    0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0,  -0x1.97c1d1b3b7afp-3,
    0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
    0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,  0x1.ca4b31f026aap-1,  0x1.476a9543891bap-3,
    0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,  0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
    0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,  0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
};
__device__ const unsigned long long kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

// powf(x, y) for finite normal x > 0 and moderate y (|y log2 x| < 126); +inf -> 0 for y < 0 (the weight of a
// block whose samples are all zero: normalizer 0, lpc.rs:827-828)
__device__ __forceinline__ float dev_powf_pos(float x, float y) {
  uint32_t ix = __float_as_uint(x);
  if (ix >= 0x7f800000u) return y < 0.0f ? 0.0f : x;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const uint32_t top = tmp & 0xff800000u;
  const int k = (int)top >> 23;
  const double z = (double)__uint_as_float(ix - top);
  const double r = z * kPowLog2Tab[2 * i] - 1.0;
  const double y0 = kPowLog2Tab[2 * i + 1] + (double)k;
  const double r2 = r * r;
  double yy = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
  const double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
  const double r4 = r2 * r2;
  double q = 0x1.71547652ab82bp0 * r + y0;
  q = p * r2 + q;
  yy = yy * r4 + q;
  const double ylogx = (double)y * yy;
  const double kShift = 0x1.8p+47;
  double kd = ylogx + kShift;
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double rr = ylogx - kd;
  unsigned long long t = kExp2Tab[ki & 31ull];
  t += ki << 47;
  const double s = __longlong_as_double((long long)t);
  const double zz = 0x1.c6af84b912394p-5 * rr + 0x1.ebfce50fac4f3p-3;
  const double rr2 = rr * rr;
  double o = 0x1.62e42ff0c52d6p-1 * rr + 1.0;
  o = zz * rr2 + o;
  o = o * s;
  return (float)o;
}

constexpr int kErrChunk = 1024;

// LDS: xw[n4] f32 | w[n4] f32 (IRLS only; WG: in HBM scratch instead -- blocks above 16384 samples, whose two f32 arrays
//      do not fit 160 KB) | echunk[kErrChunk] f32 (IRLS only) | gram[32*32] f64 | m[32*32] f64 |
//      corr[33] f64 | v[32] f64 | coefs[32] f64 | best[32] f64 | misc
template <bool STEREO, bool IRLS, bool WG = false>
__global__ void __launch_bounds__(576) direct_mse_kernel(DirectMseArgs a) {
  static_assert(!WG || IRLS, "only the IRLS weights ever leave the LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const int n4 = (n + 3) & ~3;
  float* const xw = reinterpret_cast<float*>(smem_raw);
  float* const wgt = WG ? a.weight_scratch + (size_t)blockIdx.x * n4 : xw + n4;
  float* const echunk = xw + n4 + ((IRLS && !WG) ? n4 : 0);
  double* const gram = reinterpret_cast<double*>(echunk + (IRLS ? kErrChunk : 0));  // column-major P x P
  double* const gram_ = gram;
  const int PP = (P * P + 1) & ~1;  // (the two P x P matrices are sized by the order: 2 waves per SIMD at order 8)
  double* const m_ = gram + PP;
  double* const corr = m_ + PP;
  double* const v_ = corr + 33;
  double* const coefs = v_ + 32;
  double* const best = coefs + 32;
  int* const misc = reinterpret_cast<int*>(best + 32);  // [0] solve ok, [1] status
  float* const fmisc = reinterpret_cast<float*>(misc + 4);  // [0] sum_abs_err, [1] best_error

  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
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
  auto sample = [&](int t) -> int32_t {
    int32_t s = rowA[t];
    if (STEREO && kind >= 2) {
      const int32_t r = rowB[t];
      s = kind == 2 ? (s + r) >> 1 : s - r;  // coding.rs:483
    }
    return s;
  };
  // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754); IRLS: weights start at 1 (lpc.rs:821-822)
  int32_t my_maxabs = 0;
  // 16-byte pieces, four in flight per thread (a thread that fetched one sample per trip spent 64 dependent round
  // trips to HBM on a 4096-sample block: two thirds of the kernel at order 8)
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0) && ((a.stride & 3) == 0);
  const int nq = vec_ok ? (n >> 2) : 0;  // whole quads
  auto stage4 = [&](int q, const int4 va, const int4 vb) {
    int4 v = va;
    if (STEREO && kind == 2) v = make_int4((va.x + vb.x) >> 1, (va.y + vb.y) >> 1, (va.z + vb.z) >> 1, (va.w + vb.w) >> 1);
    if (STEREO && kind == 3) v = make_int4(va.x - vb.x, va.y - vb.y, va.z - vb.z, va.w - vb.w);  // coding.rs:483
    float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab) w4 = *reinterpret_cast<const float4*>(wtab + 4 * q);  // (the table starts 16-byte aligned behind its 32 pad floats)
    *reinterpret_cast<float4*>(xw + 4 * q) = make_float4((float)v.x * w4.x, (float)v.y * w4.y, (float)v.z * w4.z, (float)v.w * w4.w);
    if (IRLS) {
      *reinterpret_cast<float4*>(wgt + 4 * q) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
      const int32_t sv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t ab = sv[u] < 0 ? (int32_t)(0u - (uint32_t)sv[u]) : sv[u];  // i32::abs (wrapping)
        my_maxabs = ab > my_maxabs ? ab : my_maxabs;
      }
    }
  };
  for (int q0 = tid; q0 < nq; q0 += 4 * nthr) {
    int4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * nthr;
      va[u] = vb[u] = make_int4(0, 0, 0, 0);
      if (q < nq) {
        va[u] = *reinterpret_cast<const int4*>(rowA + 4 * q);
        if (STEREO && kind >= 2) vb[u] = *reinterpret_cast<const int4*>(rowB + 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + u * nthr < nq) stage4(q0 + u * nthr, va[u], vb[u]);
  }
  for (int t = 4 * nq + tid; t < n; t += nthr) {  // unaligned rows, and the last samples of a ragged block
    const int32_t s = sample(t);
    xw[t] = (float)s * (wtab ? wtab[t] : 1.0f);
    if (IRLS) {
      wgt[t] = 1.0f;
      const int32_t ab = s < 0 ? (int32_t)(0u - (uint32_t)s) : s;  // i32::abs (wrapping)
      my_maxabs = ab > my_maxabs ? ab : my_maxabs;
    }
  }
  float normalizer = 0.0f;
  if (IRLS) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_xor(my_maxabs, d, 64);
      my_maxabs = o > my_maxabs ? o : my_maxabs;
    }
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&misc[2], my_maxabs);
    __syncthreads();
    normalizer = (float)misc[2];
    if (tid == 0) {
      fmisc[1] = 3.40282347e+38f;  // f32::MAX
      misc[3] = 0;                 // have a best
    }
  }
  if (tid == 0) misc[1] = 0;
  __syncthreads();

  // H on the matrix cores: D[a][b] += sum over four time steps of A[a][k] B[k][b] with A[a][k] = y_a(t' + k) and
  // B[k][b] = f32(w y_b)(t' + k) is v_mfma_f64_16x16x4_f64, and chained through its C operand that instruction IS the
  // sequential chain acc = fma(A[a][k], B[k][b], acc), k ascending, bit for bit (tools/microbench/mfma_f64_order.hip:
  // 512 000 outputs over +-20 binades of operands, no mismatch; the descending order and a pairwise tree match on
  // 53 % and 24 %).  A wave owns a 16 x 16 tile of H and walks the block once: 16 x 16 x (n - P) fma in (n - P) / 4
  // instructions where round 3's first version had a thread walk one entry's chain (P + 1 + P (P + 1) / 2 threads,
  // each reading both of its operands from LDS per step).  Needed tiles: column block 0 (R[tau] = H[tau][0]) and
  // the upper triangle a <= b; rows / columns beyond P shadow row / column P and are never stored.
  typedef double v4d_t __attribute__((ext_vector_type(4)));
  const int wave = tid >> 6, nwaves = nthr >> 6;
  const int NT = (P + 1 + 15) >> 4;  // tiles per side: 1 up to order 15, 2 up to 31, 3 at order 32
  const int steps = IRLS ? (int)a.mae_steps : 0;
  for (int it = 0; it <= steps; ++it) {
    // ---- chains ----
    if (n >= P + 1 && P <= 11) {
      // Orders up to 11: H is at most 12 x 12 = 3 x 3 blocks of 4 x 4, and v_mfma_f64_4x4x4_4b_f64 carries four such
      // blocks per instruction (lane 16 k + 4 b + i: A_b[i][k], + j: B_b[k][j]; D_b[i][j] in lane 16 i + 4 b + j; chained
      // through C it is the sequential chain, tools/microbench/mfma_f64_4x4x4_probe.hip).  Needed: column block 0 and
      // the upper triangle -- four blocks up to order 7 (one instruction per four time steps, 16 cycles where the
      // 16 x 16 tile takes 64), eight up to order 11 (two).
      if (wave == 0) {
        const int len = n - P;
        const int NB = (P + 1 + 3) >> 2;  // 1..3
        const int kq = lane >> 4, bs = (lane >> 2) & 3, r = lane & 3;
        // block (I, J) of instruction q, slot bs
        auto blk = [&](int q, int slot, int& I, int& J) {
          if (NB <= 2) {  // (0,0) (0,1) (1,1) (1,0)
            I = slot == 2 || slot == 3 ? 1 : 0;
            J = slot == 1 || slot == 2 ? 1 : 0;
          } else if (q == 0) {  // (0,0) (0,1) (0,2) (1,1)
            I = slot == 3 ? 1 : 0;
            J = slot == 3 ? 1 : slot;
          } else {  // (1,2) (2,2) (1,0) (2,0)
            I = slot == 0 || slot == 2 ? 1 : 2;
            J = slot < 2 ? 2 : 0;
          }
        };
        const int NI = NB <= 2 ? 1 : 2;
        int I0, J0, I1 = 0, J1 = 0;
        blk(0, bs, I0, J0);
        if (NI > 1) blk(1, bs, I1, J1);
        auto rowcol = [&](int I, int J, const float*& pa_, const float*& pb_) {
          int arow = 4 * I + r, bcol = 4 * J + r;
          arow = arow > P ? P : arow;
          bcol = bcol > P ? P : bcol;
          pa_ = xw + (P - arow) + kq;
          pb_ = xw + (P - bcol) + kq;
        };
        const float *pa0, *pb0, *pa1, *pb1;
        rowcol(I0, J0, pa0, pb0);
        rowcol(I1, J1, pa1, pb1);
        const float* __restrict__ pw = wgt + P + kq;
        double acc0 = 0.0, acc1 = 0.0;
        int k = 0;
        // Eight steps per trip of an inner loop with a constant count: their LDS reads are issued together -- one wave
        // per workgroup and two per SIMD cannot hide an LDS round trip per step.  (Left to `#pragma unroll` the loop
        // stayed rolled: 47 % of the wave's cycles in s_waitcnt, 160 us per subframe for 15 us of matrix-core time.)
        auto step = [&](int kk) __attribute__((always_inline)) {
          float b0 = pb0[kk], b1 = NI > 1 ? pb1[kk] : 0.0f;
          if (IRLS) {
            const float wk = pw[kk];
            b0 = wk * b0;
            b1 = wk * b1;
          }
          acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa0[kk], (double)b0, acc0, 0, 0, 0);
          if (NI > 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa1[kk], (double)b1, acc1, 0, 0, 0);
        };
        if (NI > 1) {
          for (; k + 32 <= len; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) step(k + 4 * u);
          }
        } else {
          for (; k + 32 <= len; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) step(k + 4 * u);
          }
        }
        for (; k + 4 <= len; k += 4) step(k);
        if (k < len) {  // the last one to three steps: the missing ones multiply by 0 (x + 0 * y == x)
          const bool in = k + kq < len;
          const float a0 = in ? pa0[k] : 0.0f, a1 = in ? pa1[k] : 0.0f;
          float b0 = in ? pb0[k] : 0.0f, b1 = in ? pb1[k] : 0.0f;
          if (IRLS && in) {
            const float wk = pw[k];
            b0 = wk * b0;
            b1 = wk * b1;
          }
          acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a0, (double)b0, acc0, 0, 0, 0);
          if (NI > 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a1, (double)b1, acc1, 0, 0, 0);
        }
        // D_b[i][j] sits in lane 16 i + 4 b + j
        const int oi = lane >> 4, oj = lane & 3;
        auto put = [&](int I, int J, double v) {
          const int ra = 4 * I + oi, cb = 4 * J + oj;
          if (cb == 0 && ra <= P) {
            corr[ra] = v;
          } else if (ra >= 1 && ra <= cb && cb <= P) {
            gram[(ra - 1) + (cb - 1) * P] = v;
            gram[(cb - 1) + (ra - 1) * P] = v;
          }
        };
        put(I0, J0, acc0);
        if (NI > 1) put(I1, J1, acc1);
      }
    } else if (n >= P + 1) {
      const int len = n - P;  // t' = P - 1 .. n - 2
      for (int tile = wave; tile < NT * NT; tile += nwaves) {  // (wave-uniform)
        const int I = tile / NT, J = tile - I * NT;
        if (J < I && J != 0) continue;
        const int kq = lane >> 4;
        int arow = 16 * I + (lane & 15), bcol = 16 * J + (lane & 15);
        arow = arow > P ? P : arow;
        bcol = bcol > P ? P : bcol;
        const float* __restrict__ pa = xw + (P - arow) + kq;  // y_a(t') = xw[t' + 1 - a], t' = P - 1 ..
        const float* __restrict__ pb = xw + (P - bcol) + kq;
        const float* __restrict__ pw = wgt + P + kq;          // w[t' + 1]
        v4d_t acc = {0.0, 0.0, 0.0, 0.0};
        int k = 0;
        auto tstep = [&](int kk) __attribute__((always_inline)) {
          const float av = pa[kk];
          float bv = pb[kk];
          if (IRLS) bv = pw[kk] * bv;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av, (double)bv, acc, 0, 0, 0);
        };
        for (; k + 32 <= len; k += 32) {  // (eight steps' LDS reads in flight: see the block form above)
#pragma unroll
          for (int u = 0; u < 8; ++u) tstep(k + 4 * u);
        }
        for (; k + 4 <= len; k += 4) tstep(k);
        if (k < len) {  // the last one to three steps: the missing ones multiply by 0 (x + 0 * y == x)
          const bool in = k + kq < len;
          const float av = in ? pa[k] : 0.0f;
          float bv = in ? pb[k] : 0.0f;
          if (IRLS && in) bv = pw[k] * bv;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av, (double)bv, acc, 0, 0, 0);
        }
        // output register r of lane l: row 4 r + l / 16, column l % 16
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ra = 16 * I + 4 * r + kq, cb = 16 * J + (lane & 15);
          if (cb == 0 && ra <= P) {
            corr[ra] = acc[r];
          } else if (ra >= 1 && ra <= cb && cb <= P) {
            gram[(ra - 1) + (cb - 1) * P] = acc[r];
            gram[(cb - 1) + (ra - 1) * P] = acc[r];
          }
        }
      }
    } else {
      for (int c = tid; c < P + 1; c += nthr) corr[c] = 0.0;
      for (int c = tid; c < P * P; c += nthr) gram[c] = 0.0;
    }
    __syncthreads();
    if (it == steps && a.autocorr && tid <= 32 && !IRLS) a.autocorr[(size_t)sf * 33 + tid] = tid <= P ? corr[tid] : 0.0;
    if (!IRLS && a.gram_scratch != nullptr) {
      // the chains' results go to HBM; direct_mse_solve_kernel takes it from here, a lane per subframe
      double* __restrict__ out = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
      for (int c = tid; c < 33; c += nthr) out[c] = c <= P ? corr[c] : 0.0;
      for (int c = tid; c < P * P; c += nthr) out[33 + c] = gram[c];
      return;
    }

    // ---- solve_sym_mut with the regulariser loop (lpc.rs:887-896), wave 0, lane = matrix row ----
    if (tid < 64) {
      // (volatile: other lanes of the wave write what this lane reads next; LDS operations of one wave complete
      // in order, the compiler just must not keep any of it in registers)
      volatile double* const m = m_;
      volatile double* const v = v_;
      volatile double* const gram = gram_;
      const int r = lane;
      double regularizer = 0.0;
      int tries = 0;
      for (;;) {
        // mat.clone(); xy = corr[1..]
        if (r < P) {
          for (int c = 0; c < P; ++c) m[r + c * P] = gram[r + c * P];
          v[r] = corr[r + 1];
        }
        __builtin_amdgcn_wave_barrier();
        bool ok = true;
        for (int j = 0; j < P && ok; ++j) {
          for (int k = 0; k < j; ++k) {
            const double factor = -m[j + k * P];
            if (r >= j && r < P) {
              const double ax = factor * m[r + k * P];
              m[r + j * P] = ax + m[r + j * P];  // array_axcpy: (a * x) * 1 + 1 * y, no fma
            }
            __builtin_amdgcn_wave_barrier();
          }
          const double diag = m[j + j * P];
          if (diag == 0.0 || !(diag >= 0.0)) {  // is_zero() / try_sqrt() == None
            ok = false;
            break;
          }
          const double denom = __builtin_sqrt(diag);
          __builtin_amdgcn_wave_barrier();
          if (r == j) m[j + j * P] = denom;
          if (r > j && r < P) m[r + j * P] = m[r + j * P] / denom;
          __builtin_amdgcn_wave_barrier();
        }
        if (ok) {
          // solve_lower_triangular_vector_unchecked_mut
          for (int i = 0; i < P; ++i) {
            const double coeff = v[i] / m[i + i * P];
            __builtin_amdgcn_wave_barrier();
            if (r == i) v[i] = coeff;
            if (r > i && r < P) v[r] = ((-coeff) * m[r + i * P]) + v[r];
            __builtin_amdgcn_wave_barrier();
          }
          // ad_solve_lower_triangular: b[i] = (b[i] - dot(L[i+1.., i], b[i+1..])) / L[i][i], dotx's accumulators
          if (r == 0) {
            for (int i = P - 1; i >= 0; --i) {
              const int rows = P - (i + 1);
              double acc8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
              double res = 0.0;
              int q = 0;
              while (rows - q >= 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc8[u] += m[(i + 1 + q + u) + i * P] * v[i + 1 + q + u];
                q += 8;
              }
              res += acc8[0] + acc8[4];
              res += acc8[1] + acc8[5];
              res += acc8[2] + acc8[6];
              res += acc8[3] + acc8[7];
              for (int k = q; k < rows; ++k) res += m[(i + 1 + k) + i * P] * v[i + 1 + k];
              v[i] = (v[i] - res) / m[i + i * P];
            }
          }
          __builtin_amdgcn_wave_barrier();
          break;
        }
        // regularizer = max(1, 2 regularizer); diag += regularizer - old (lpc.rs:889-895)
        const double old = regularizer;
        const double twice = regularizer + regularizer;
        regularizer = 1.0 > twice ? 1.0 : twice;
        if (r < P) gram[r + r * P] += regularizer - old;
        __builtin_amdgcn_wave_barrier();
        if (++tries > 2000) {  // (NaN input: the reference would not terminate)
          if (r == 0) misc[1] = FLACENC_HIP_SUBFRAME_NONFINITE;
          if (r < P) v[r] = 0.0;
          break;
        }
      }
      if (r < P) coefs[r] = v[r];
    }
    __syncthreads();
    if (!IRLS) break;

    // ---- compute_raw_errors (f32 fma chain over the taps), sum of |err| as ONE sequential f32 chain
    //      (Iterator::sum, lpc.rs:839), new weights (lpc.rs:845-847) ----
    if (tid == 0) fmisc[0] = 0.0f;
    for (int base = 0; base < n; base += kErrChunk) {
      __syncthreads();
      for (int o = tid; o < kErrChunk && base + o < n; o += nthr) {
        const int t = base + o;
        float e = 0.0f;  // raw_errors[t] for t < order: never written, 0
        if (t >= P) {
          if (wtab == nullptr) {
            // the rectangular window (the reference's experimental configuration): x_w IS (f32)s -- the taps come
            // from LDS instead of P + 1 (stereo: twice that) loads from HBM per error
            e = -xw[t];
            for (int j = 0; j < P; ++j) e = __builtin_fmaf((float)coefs[j], xw[t - 1 - j], e);
          } else {
            e = (float)(int32_t)(0u - (uint32_t)sample(t));
            for (int j = 0; j < P; ++j) e = __builtin_fmaf((float)coefs[j], (float)sample(t - 1 - j), e);
          }
          float x = __builtin_fabsf(e);
          x = x > 1.0f ? x : 1.0f;
          x = x / normalizer;
          x = x > 0.01f ? x : 0.01f;
          wgt[t] = dev_powf_pos(x, -1.2f);
        }
        echunk[o] = __builtin_fabsf(e);
      }
      __syncthreads();
      if (tid == 0) {
        // (Iterator::sum is one sequential chain: sixteen values per trip come as four 16-byte reads, the additions
        // stay in order)
        float sacc = fmisc[0];
        const int cnt = n - base < kErrChunk ? n - base : kErrChunk;
        int o = 0;
        for (; o + 16 <= cnt; o += 16) {
          const float4 q0 = *reinterpret_cast<const float4*>(&echunk[o]);
          const float4 q1 = *reinterpret_cast<const float4*>(&echunk[o + 4]);
          const float4 q2 = *reinterpret_cast<const float4*>(&echunk[o + 8]);
          const float4 q3 = *reinterpret_cast<const float4*>(&echunk[o + 12]);
          sacc += q0.x; sacc += q0.y; sacc += q0.z; sacc += q0.w;
          sacc += q1.x; sacc += q1.y; sacc += q1.z; sacc += q1.w;
          sacc += q2.x; sacc += q2.y; sacc += q2.z; sacc += q2.w;
          sacc += q3.x; sacc += q3.y; sacc += q3.z; sacc += q3.w;
        }
        for (; o < cnt; ++o) sacc += echunk[o];
        fmisc[0] = sacc;
      }
    }
    __syncthreads();
    if (fmisc[0] < fmisc[1]) {  // uniform: same LDS value for every thread
      __syncthreads();
      if (tid < P) best[tid] = coefs[tid];
      if (tid <= 32 && a.autocorr) a.autocorr[(size_t)sf * 33 + tid] = tid <= P ? corr[tid] : 0.0;
      if (tid == 0) {
        fmisc[1] = fmisc[0];
        misc[3] = 1;
      }
    }
    __syncthreads();
  }
  if (IRLS) {
    __syncthreads();
    if (tid < P) coefs[tid] = misc[3] ? best[tid] : 0.0;
    if (tid == 0 && !misc[3]) misc[1] = FLACENC_HIP_SUBFRAME_NONFINITE;  // best_coefs.unwrap() would panic
    __syncthreads();
  }

  // ---- quantize_parameters, lpc.rs:273-302 (find_shift :234-254, quantize_parameter :258-270) ----
  if (tid == 0) {
    int status = misc[1];
    for (int i = 0; i < P; ++i) {
      const uint64_t b = (uint64_t)__double_as_longlong(coefs[i]);
      if (((b >> 52) & 0x7FF) == 0x7FF) status |= FLACENC_HIP_SUBFRAME_NONFINITE;
    }
    int32_t* pr = a.pred_out + (size_t)sf * 36;
    for (int i = 0; i < 36; ++i) pr[i] = 0;
    int shift = 0, order = 0;
    if (status == 0) {
      double max_abs = 0.0;
      for (int i = 0; i < P; ++i) max_abs = fmax(max_abs, fabs(coefs[i]));
      int abs_log2 = dm_ceil_log2_pos(max_abs);
      if (abs_log2 < -32752) abs_log2 = -32752;
      const int precision = (int)a.precision;
      shift = (precision - 1) - abs_log2;
      shift = shift < 0 ? 0 : (shift > 15 ? 15 : shift);
      const double scalefac = (double)(1 << shift);
      const int lo = -(1 << (precision - 1)), hi = (1 << (precision - 1)) - 1;
      order = 1;
      for (int i = 0; i < P; ++i) {
        double s = round(coefs[i] * scalefac);  // half away from zero
        s = s < -32768.0 ? -32768.0 : (s > 32767.0 ? 32767.0 : s);
        int q = (int)s;
        q = q < lo ? lo : (q > hi ? hi : q);
        pr[i] = q;
        if (q != 0) order = i + 1;  // tail-zero truncation, min 1
      }
      for (int i = order; i < P; ++i) pr[i] = 0;
    }
    pr[32] = order;
    pr[33] = shift;
    pr[34] = status;
    if (a.lpc_coefs)
      for (int i = 0; i < 32; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = (i < P && status == 0) ? coefs[i] : 0.0;
  }
}

// The chains from a SLIDING f64 window instead of the whole block in LDS, on v_mfma_f64_4x4x4_4b_f64 for every order.
// H is cut into 4 x 4 blocks; needed are column block 0 (R[tau] = H[tau][0]) and the upper triangle: NB + NB (NB + 1) / 2 - 1
// blocks for NB = ceil((P + 1) / 4) block rows -- 4 at order 7, 8 at 11, 34 at 24, 53 at 32 -- four per instruction.  The
// form above (the whole block as f32 in LDS, a conversion for every operand of every step, one 16 x 16 tile per wave from
// order 12 with 81 .. 1089 of 1024 .. 9216 outputs wanted) spends its time waiting and multiplying padding; this kernel keeps
// 4.5 KB (560 doubles: 512 steps' samples + the P + 3 ahead of them), converts each sample once when it enters the
// window, fetches the next 512 samples into registers before it walks the current ones, and deals the instructions to
// up to four waves (NIW accumulators each): 24 and more workgroups fit a CU.  The chain of every entry --
// the MFMA through its C operand, k ascending -- is the reference's sequential fma chain, operand for operand
// (profiles/r03_mfma_f64_4x4x4_probe.txt).
// WEIGHTED (IRLS steps after the first): B[k][b] = f32(w * y_b) (lpc.rs:463-470) -- the weights' window and an f32 copy
// of the samples' ride along, the product and its widening stay in the loop; a subframe whose estimate already failed
// is skipped.
constexpr int kStreamPiece = 512;
constexpr int kStreamHalo = 48;  // >= P + 3, whole quads
template <bool STEREO, bool WEIGHTED, int NIW>
__global__ void __launch_bounds__(256) direct_mse_stream_kernel(DirectMseArgs a) {
  __shared__ __attribute__((aligned(16))) double win[kStreamPiece + kStreamHalo];
  __shared__ __attribute__((aligned(16))) float xf[WEIGHTED ? kStreamPiece + kStreamHalo : 4];
  __shared__ __attribute__((aligned(16))) float ww[WEIGHTED ? kStreamPiece + kStreamHalo : 4];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wave = tid >> 6, nwaves = nthr >> 6;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
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
  double* __restrict__ out = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
  const bool irls = a.irls_state != nullptr;  // (then R[] reaches a.autocorr through the error pass: the best step's)
  if (irls && a.irls_step > 0 && a.irls_state[(size_t)sf * kIrlsStateDoubles + 66] != 0.0) return;  // the estimate failed in an earlier step
  const float* __restrict__ wsrc = WEIGHTED ? a.irls_weights + (size_t)sf * (((size_t)n + 3) & ~(size_t)3) : nullptr;
  if (n < P + 1) {  // (lpc.rs:860-862: nothing to estimate from)
    for (int c = tid; c < 33 + P * P; c += nthr) out[c] = 0.0;
    if (a.autocorr && !irls && tid <= 32) a.autocorr[(size_t)sf * 33 + tid] = 0.0;
    return;
  }
  // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754), widened once; samples [4 q, 4 q + 4), zeros behind the block
  struct Quad {
    int4 va, vb;
    float4 wt;
  };
  auto fetch = [&](int q) -> Quad {  // (rows are 16-byte aligned with a stride of whole quads: the launcher checks)
    Quad r;
    r.va = r.vb = make_int4(0, 0, 0, 0);
    r.wt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (WEIGHTED && 4 * q < n) r.wt = *reinterpret_cast<const float4*>(wsrc + 4 * q);  // (rows of whole quads)
    if (4 * q + 3 < n) {
      r.va = *reinterpret_cast<const int4*>(rowA + 4 * q);
      if (STEREO && kind >= 2) r.vb = *reinterpret_cast<const int4*>(rowB + 4 * q);
    } else if (4 * q < n) {
      int32_t ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) {
        ta[u] = rowA[4 * q + u];
        if (STEREO && kind >= 2) tb[u] = rowB[4 * q + u];
      }
      r.va = make_int4(ta[0], ta[1], ta[2], ta[3]);
      r.vb = make_int4(tb[0], tb[1], tb[2], tb[3]);
    }
    return r;
  };
  auto place = [&](int q, const Quad& r, int at) {  // -> win[at .. at + 4)
    int4 v = r.va;
    if (STEREO && kind == 2) v = make_int4((r.va.x + r.vb.x) >> 1, (r.va.y + r.vb.y) >> 1, (r.va.z + r.vb.z) >> 1, (r.va.w + r.vb.w) >> 1);
    if (STEREO && kind == 3) v = make_int4(r.va.x - r.vb.x, r.va.y - r.vb.y, r.va.z - r.vb.z, r.va.w - r.vb.w);  // coding.rs:483
    float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab && 4 * q + 3 < n) w4 = *reinterpret_cast<const float4*>(wtab + 4 * q);
    else if (wtab) {
      float tw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) tw[u] = wtab[4 * q + u];
      w4 = make_float4(tw[0], tw[1], tw[2], tw[3]);
    }
    const float4 xv = make_float4((float)v.x * w4.x, (float)v.y * w4.y, (float)v.z * w4.z, (float)v.w * w4.w);
    *reinterpret_cast<double2*>(&win[at]) = make_double2((double)xv.x, (double)xv.y);
    *reinterpret_cast<double2*>(&win[at + 2]) = make_double2((double)xv.z, (double)xv.w);
    if (WEIGHTED) {
      *reinterpret_cast<float4*>(&xf[at]) = xv;
      *reinterpret_cast<float4*>(&ww[at]) = r.wt;
    }
  };
  // lane -> operand elements: lane 16 k + 4 b + i holds A_b[i][k], lane 16 k + 4 b + j holds B_b[k][j].  Block list:
  // (0,0) (1,0) .. (NB-1,0), then column by column (0,1) (1,1), (0,2) (1,2) (2,2), ...; instruction q takes blocks
  // 4 q .. 4 q + 3, wave w the instructions w, w + nwaves, ...; slots behind the list repeat block (0,0).
  const int len = n - P;
  const int NB = (P + 1 + 3) >> 2;  // 1..9
  const int nblocks = NB + NB * (NB + 1) / 2 - 1;
  const int kq = lane >> 4, bs = (lane >> 2) & 3, r = lane & 3;
  auto block_of = [&](int idx, int& I, int& J) {
    if (idx >= nblocks) idx = 0;
    if (idx < NB) {
      I = idx;
      J = 0;
      return;
    }
    idx -= NB;
    int col = 1;
    while (idx >= col + 1) {
      idx -= col + 1;
      ++col;
    }
    I = idx;
    J = col;
  };
  int oa[NIW], ob[NIW], bi[NIW], bj[NIW];
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const int q = wave + nwaves * i;
    block_of(4 * q + bs, bi[i], bj[i]);
    int arow = 4 * bi[i] + r, bcol = 4 * bj[i] + r;
    arow = arow > P ? P : arow;
    bcol = bcol > P ? P : bcol;
    oa[i] = (P - arow) + kq;  // step k reads x_w[(P - row) + k + kq]
    ob[i] = (P - bcol) + kq;
  }
  double acc[NIW];
#pragma unroll
  for (int i = 0; i < NIW; ++i) acc[i] = 0.0;
  // window = samples [base, base + 512 + halo)
  constexpr int kWinQuads = (kStreamPiece + kStreamHalo) / 4;  // 140
  for (int q = tid; q < kWinQuads; q += nthr) place(q, fetch(q), 4 * q);
  __syncthreads();
  for (int base = 0; base < len; base += kStreamPiece) {
    // the next 512 samples (behind the halo already here) on their way while this piece is walked
    const bool more = base + kStreamPiece < len;
    Quad nx[2];
    const int q_next = (base + kStreamPiece + kStreamHalo) / 4 + tid;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < kStreamPiece / 4) nx[u] = fetch(q_next + u * nthr);
    }
    const int steps_here = (len - base) < kStreamPiece ? (len - base) : kStreamPiece;
    const float* __restrict__ pw = ww + P + kq;  // w[t' + 1]
    auto step = [&](int kk) __attribute__((always_inline)) {
      float wk = 0.0f;
      if (WEIGHTED) wk = pw[kk];
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const double bv = WEIGHTED ? (double)(wk * xf[ob[i] + kk]) : win[ob[i] + kk];
        acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(win[oa[i] + kk], bv, acc[i], 0, 0, 0);
      }
    };
    int k = 0;
    for (; k + 32 <= steps_here; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) step(k + 4 * u);
    }
    for (; k + 4 <= steps_here; k += 4) step(k);
    if (k < steps_here) {  // the block's last one to three steps: the missing ones multiply by 0 (x + 0 * y == x)
      const bool in = k + kq < steps_here;
      const float wk = (WEIGHTED && in) ? pw[k] : 0.0f;
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const double av = in ? win[oa[i] + k] : 0.0;
        const double bv = !in ? 0.0 : (WEIGHTED ? (double)(wk * xf[ob[i] + k]) : win[ob[i] + k]);
        acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();  // every wave is done with this piece
      if (tid < kStreamHalo / 4) {
        const double2 t0 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * tid]);
        const double2 t1 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * tid + 2]);
        float4 f0, f1;
        if (WEIGHTED) {
          f0 = *reinterpret_cast<const float4*>(&xf[kStreamPiece + 4 * tid]);
          f1 = *reinterpret_cast<const float4*>(&ww[kStreamPiece + 4 * tid]);
        }
        // (the halo's first quads land where its last ones are read from only if halo > piece: not so)
        *reinterpret_cast<double2*>(&win[4 * tid]) = t0;
        *reinterpret_cast<double2*>(&win[4 * tid + 2]) = t1;
        if (WEIGHTED) {
          *reinterpret_cast<float4*>(&xf[4 * tid]) = f0;
          *reinterpret_cast<float4*>(&ww[4 * tid]) = f1;
        }
      }
      __syncthreads();  // (the refill below overwrites the old halo's place)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < kStreamPiece / 4) place(q_next + u * nthr, nx[u], kStreamHalo + 4 * (tid + u * nthr));
      __syncthreads();
    }
  }
  // D_b[i][j] sits in lane 16 i + 4 b + j
  const int oi = lane >> 4, oj = lane & 3;
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const int ra = 4 * bi[i] + oi, cb = 4 * bj[i] + oj;
    const double v = acc[i];
    if (cb == 0 && ra <= P) {
      out[ra] = v;
      if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + ra] = v;
    } else if (ra >= 1 && ra <= cb && cb <= P) {
      out[33 + (ra - 1) + (cb - 1) * P] = v;
      out[33 + (cb - 1) + (ra - 1) * P] = v;
    }
  }
  if (tid > P && tid <= 32) {
    out[tid] = 0.0;
    if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + tid] = 0.0;
  }
}

// Orders 12..31 from the same window on v_mfma_f64_16x16x4_f64: four waves = the four needed 16 x 16 tiles around one
// window (two 512-byte operand reads per 64-cycle instruction: the LDS is idle where the 4 x 4 blocks above saturate it),
// 4.5 KB of LDS per subframe where the one-kernel form stages the whole block.
template <bool STEREO, bool WEIGHTED>
__global__ void __launch_bounds__(256) direct_mse_stream_tiles_kernel(DirectMseArgs a) {
  __shared__ __attribute__((aligned(16))) double win[kStreamPiece + kStreamHalo];
  __shared__ __attribute__((aligned(16))) float xf[WEIGHTED ? kStreamPiece + kStreamHalo : 4];
  __shared__ __attribute__((aligned(16))) float ww[WEIGHTED ? kStreamPiece + kStreamHalo : 4];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wave = tid >> 6, nwaves = nthr >> 6;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
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
  double* __restrict__ out = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
  const bool irls = a.irls_state != nullptr;  // (then R[] reaches a.autocorr through the error pass: the best step's)
  if (irls && a.irls_step > 0 && a.irls_state[(size_t)sf * kIrlsStateDoubles + 66] != 0.0) return;  // the estimate failed in an earlier step
  const float* __restrict__ wsrc = WEIGHTED ? a.irls_weights + (size_t)sf * (((size_t)n + 3) & ~(size_t)3) : nullptr;
  if (n < P + 1) {  // (lpc.rs:860-862: nothing to estimate from)
    for (int c = tid; c < 33 + P * P; c += nthr) out[c] = 0.0;
    if (a.autocorr && !irls && tid <= 32) a.autocorr[(size_t)sf * 33 + tid] = 0.0;
    return;
  }
  // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754), widened once; samples [4 q, 4 q + 4), zeros behind the block
  struct Quad {
    int4 va, vb;
    float4 wt;
  };
  auto fetch = [&](int q) -> Quad {  // (rows are 16-byte aligned with a stride of whole quads: the launcher checks)
    Quad r;
    r.va = r.vb = make_int4(0, 0, 0, 0);
    r.wt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (WEIGHTED && 4 * q < n) r.wt = *reinterpret_cast<const float4*>(wsrc + 4 * q);  // (rows of whole quads)
    if (4 * q + 3 < n) {
      r.va = *reinterpret_cast<const int4*>(rowA + 4 * q);
      if (STEREO && kind >= 2) r.vb = *reinterpret_cast<const int4*>(rowB + 4 * q);
    } else if (4 * q < n) {
      int32_t ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) {
        ta[u] = rowA[4 * q + u];
        if (STEREO && kind >= 2) tb[u] = rowB[4 * q + u];
      }
      r.va = make_int4(ta[0], ta[1], ta[2], ta[3]);
      r.vb = make_int4(tb[0], tb[1], tb[2], tb[3]);
    }
    return r;
  };
  auto place = [&](int q, const Quad& r, int at) {  // -> win[at .. at + 4)
    int4 v = r.va;
    if (STEREO && kind == 2) v = make_int4((r.va.x + r.vb.x) >> 1, (r.va.y + r.vb.y) >> 1, (r.va.z + r.vb.z) >> 1, (r.va.w + r.vb.w) >> 1);
    if (STEREO && kind == 3) v = make_int4(r.va.x - r.vb.x, r.va.y - r.vb.y, r.va.z - r.vb.z, r.va.w - r.vb.w);  // coding.rs:483
    float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab && 4 * q + 3 < n) w4 = *reinterpret_cast<const float4*>(wtab + 4 * q);
    else if (wtab) {
      float tw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) tw[u] = wtab[4 * q + u];
      w4 = make_float4(tw[0], tw[1], tw[2], tw[3]);
    }
    const float4 xv = make_float4((float)v.x * w4.x, (float)v.y * w4.y, (float)v.z * w4.z, (float)v.w * w4.w);
    *reinterpret_cast<double2*>(&win[at]) = make_double2((double)xv.x, (double)xv.y);
    *reinterpret_cast<double2*>(&win[at + 2]) = make_double2((double)xv.z, (double)xv.w);
    if (WEIGHTED) {
      *reinterpret_cast<float4*>(&xf[at]) = xv;
      *reinterpret_cast<float4*>(&ww[at]) = r.wt;
    }
  };
  // wave -> tile (I, J) of 16 x 16: (0,0) (0,1) (1,1) (1,0) -- column block 0 and the upper triangle of the 2 x 2 tiles of
  // orders 12..31; lane l: A row / B column 16 I + l % 16 (rows and columns beyond P shadow P and are never stored),
  // k = l / 16
  const int len = n - P;
  const int kq = lane >> 4;
  const int TI = (wave == 2 || wave == 3) ? 1 : 0, TJ = (wave == 1 || wave == 2) ? 1 : 0;
  int arow = 16 * TI + (lane & 15), bcol = 16 * TJ + (lane & 15);
  arow = arow > P ? P : arow;
  bcol = bcol > P ? P : bcol;
  const int oa = (P - arow) + kq, ob = (P - bcol) + kq;  // step k reads x_w[(P - row) + k + kq]
  typedef double v4d_t __attribute__((ext_vector_type(4)));
  v4d_t acc = {0.0, 0.0, 0.0, 0.0};
  // window = samples [base, base + 512 + halo)
  constexpr int kWinQuads = (kStreamPiece + kStreamHalo) / 4;  // 140
  for (int q = tid; q < kWinQuads; q += nthr) place(q, fetch(q), 4 * q);
  __syncthreads();
  for (int base = 0; base < len; base += kStreamPiece) {
    // the next 512 samples (behind the halo already here) on their way while this piece is walked
    const bool more = base + kStreamPiece < len;
    Quad nx[2];
    const int q_next = (base + kStreamPiece + kStreamHalo) / 4 + tid;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < kStreamPiece / 4) nx[u] = fetch(q_next + u * nthr);
    }
    const int steps_here = (len - base) < kStreamPiece ? (len - base) : kStreamPiece;
    const float* __restrict__ pw = ww + P + kq;  // w[t' + 1]
    auto step = [&](int kk) __attribute__((always_inline)) {
      const double bv = WEIGHTED ? (double)(pw[kk] * xf[ob + kk]) : win[ob + kk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(win[oa + kk], bv, acc, 0, 0, 0);
    };
    int k = 0;
    for (; k + 32 <= steps_here; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) step(k + 4 * u);
    }
    for (; k + 4 <= steps_here; k += 4) step(k);
    if (k < steps_here) {  // the block's last one to three steps: the missing ones multiply by 0 (x + 0 * y == x)
      const bool in = k + kq < steps_here;
      const double av = in ? win[oa + k] : 0.0;
      const double bv = !in ? 0.0 : (WEIGHTED ? (double)(pw[k] * xf[ob + k]) : win[ob + k]);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    if (more) {
      __syncthreads();  // every wave is done with this piece
      if (tid < kStreamHalo / 4) {
        const double2 t0 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * tid]);
        const double2 t1 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * tid + 2]);
        float4 f0, f1;
        if (WEIGHTED) {
          f0 = *reinterpret_cast<const float4*>(&xf[kStreamPiece + 4 * tid]);
          f1 = *reinterpret_cast<const float4*>(&ww[kStreamPiece + 4 * tid]);
        }
        // (the halo's first quads land where its last ones are read from only if halo > piece: not so)
        *reinterpret_cast<double2*>(&win[4 * tid]) = t0;
        *reinterpret_cast<double2*>(&win[4 * tid + 2]) = t1;
        if (WEIGHTED) {
          *reinterpret_cast<float4*>(&xf[4 * tid]) = f0;
          *reinterpret_cast<float4*>(&ww[4 * tid]) = f1;
        }
      }
      __syncthreads();  // (the refill below overwrites the old halo's place)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < kStreamPiece / 4) place(q_next + u * nthr, nx[u], kStreamHalo + 4 * (tid + u * nthr));
      __syncthreads();
    }
  }
  // output register r of lane l: row 4 r + l / 16, column l % 16
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ra = 16 * TI + 4 * r + kq, cb = 16 * TJ + (lane & 15);
    const double v = acc[r];
    if (cb == 0 && ra <= P) {
      out[ra] = v;
      if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + ra] = v;
    } else if (ra >= 1 && ra <= cb && cb <= P) {
      out[33 + (ra - 1) + (cb - 1) * P] = v;
      out[33 + (cb - 1) + (ra - 1) * P] = v;
    }
  }
  if (tid > P && tid <= 32) {
    out[tid] = 0.0;
    if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + tid] = 0.0;
  }
}

// ---- the same for orders up to 11 alone: one wave, at most two instructions, pointers instead of tables (measured: order 8
// 0.356 ms per 12288 subframes against 0.479 in the general form above) ----
// The chains of orders up to 11 without IRLS steps, ONE WAVE per subframe and a SLIDING f64 window instead of the whole
// block in LDS: the 4 x 4 x 4 block form above spends its time waiting -- one wave per workgroup, 16 KB of image per
// subframe (eight workgroups per CU), a conversion for every operand of every step -- where this kernel keeps 4.2 KB
// (528 doubles: 512 steps' samples + the P + 3 ahead of them), converts each sample once when it enters the window and
// fetches the next 512 samples into registers before it walks the current ones; 24 and more workgroups fit a CU.
// The chain itself -- v_mfma_f64_4x4x4_4b_f64 through its C operand, k ascending -- is the same, operand for operand.
// WEIGHTED (IRLS steps after the first): B[k][b] = f32(w * y_b) (lpc.rs:463-470) -- the weights' window and an f32 copy
// of the samples' ride along, the product and its widening stay in the loop; a subframe whose estimate already failed
// is skipped.
template <bool STEREO, bool WEIGHTED = false>
__global__ void __launch_bounds__(64) direct_mse_stream_small_kernel(DirectMseArgs a) {
  __shared__ __attribute__((aligned(16))) double win[kStreamPiece + 16];
  __shared__ __attribute__((aligned(16))) float xf[WEIGHTED ? kStreamPiece + 16 : 4];
  __shared__ __attribute__((aligned(16))) float ww[WEIGHTED ? kStreamPiece + 16 : 4];
  const int lane = threadIdx.x;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
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
  double* __restrict__ out = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
  const bool irls = a.irls_state != nullptr;  // (then R[] reaches a.autocorr through the error pass: the best step's)
  if (irls && a.irls_step > 0 && a.irls_state[(size_t)sf * kIrlsStateDoubles + 66] != 0.0) return;  // the estimate failed in an earlier step
  const float* __restrict__ wsrc = WEIGHTED ? a.irls_weights + (size_t)sf * (((size_t)n + 3) & ~(size_t)3) : nullptr;
  if (n < P + 1) {  // (lpc.rs:860-862: nothing to estimate from)
    for (int c = lane; c < 33 + P * P; c += 64) out[c] = 0.0;
    if (a.autocorr && !irls && lane <= 32) a.autocorr[(size_t)sf * 33 + lane] = 0.0;
    return;
  }
  // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754), widened once; samples [4 q, 4 q + 4), zeros behind the block
  struct Quad {
    int4 va, vb;
    float4 wt;
  };
  auto fetch = [&](int q) -> Quad {  // (rows are 16-byte aligned with a stride of whole quads: the launcher checks)
    Quad r;
    r.va = r.vb = make_int4(0, 0, 0, 0);
    r.wt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (WEIGHTED && 4 * q < n) r.wt = *reinterpret_cast<const float4*>(wsrc + 4 * q);  // (rows of whole quads)
    if (4 * q + 3 < n) {
      r.va = *reinterpret_cast<const int4*>(rowA + 4 * q);
      if (STEREO && kind >= 2) r.vb = *reinterpret_cast<const int4*>(rowB + 4 * q);
    } else if (4 * q < n) {
      int32_t ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) {
        ta[u] = rowA[4 * q + u];
        if (STEREO && kind >= 2) tb[u] = rowB[4 * q + u];
      }
      r.va = make_int4(ta[0], ta[1], ta[2], ta[3]);
      r.vb = make_int4(tb[0], tb[1], tb[2], tb[3]);
    }
    return r;
  };
  auto place = [&](int q, const Quad& r, int at) {  // -> win[at .. at + 4)
    int4 v = r.va;
    if (STEREO && kind == 2) v = make_int4((r.va.x + r.vb.x) >> 1, (r.va.y + r.vb.y) >> 1, (r.va.z + r.vb.z) >> 1, (r.va.w + r.vb.w) >> 1);
    if (STEREO && kind == 3) v = make_int4(r.va.x - r.vb.x, r.va.y - r.vb.y, r.va.z - r.vb.z, r.va.w - r.vb.w);  // coding.rs:483
    float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab && 4 * q + 3 < n) w4 = *reinterpret_cast<const float4*>(wtab + 4 * q);
    else if (wtab) {
      float tw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) tw[u] = wtab[4 * q + u];
      w4 = make_float4(tw[0], tw[1], tw[2], tw[3]);
    }
    const float4 xv = make_float4((float)v.x * w4.x, (float)v.y * w4.y, (float)v.z * w4.z, (float)v.w * w4.w);
    *reinterpret_cast<double2*>(&win[at]) = make_double2((double)xv.x, (double)xv.y);
    *reinterpret_cast<double2*>(&win[at + 2]) = make_double2((double)xv.z, (double)xv.w);
    if (WEIGHTED) {
      *reinterpret_cast<float4*>(&xf[at]) = xv;
      *reinterpret_cast<float4*>(&ww[at]) = r.wt;
    }
  };
  // lane -> operand elements, as in the block form of direct_mse_kernel
  const int len = n - P;
  const int NB = (P + 1 + 3) >> 2;  // 1..3
  const int kq = lane >> 4, bs = (lane >> 2) & 3, r = lane & 3;
  auto blk = [&](int q, int slot, int& I, int& J) {
    if (NB <= 2) {  // (0,0) (0,1) (1,1) (1,0)
      I = slot == 2 || slot == 3 ? 1 : 0;
      J = slot == 1 || slot == 2 ? 1 : 0;
    } else if (q == 0) {  // (0,0) (0,1) (0,2) (1,1)
      I = slot == 3 ? 1 : 0;
      J = slot == 3 ? 1 : slot;
    } else {  // (1,2) (2,2) (1,0) (2,0)
      I = slot == 0 || slot == 2 ? 1 : 2;
      J = slot < 2 ? 2 : 0;
    }
  };
  const int NI = NB <= 2 ? 1 : 2;
  int I0, J0, I1 = 0, J1 = 0;
  blk(0, bs, I0, J0);
  if (NI > 1) blk(1, bs, I1, J1);
  auto offs = [&](int I, int J, int& oa, int& ob) {
    int arow = 4 * I + r, bcol = 4 * J + r;
    arow = arow > P ? P : arow;
    bcol = bcol > P ? P : bcol;
    oa = (P - arow) + kq;  // step k reads x_w[(P - row) + k + kq]
    ob = (P - bcol) + kq;
  };
  int oa0, ob0, oa1, ob1;
  offs(I0, J0, oa0, ob0);
  offs(I1, J1, oa1, ob1);
  double acc0 = 0.0, acc1 = 0.0;
  // window = samples [base, base + 512 + 16): two quads and a bit per lane
  Quad nx[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int q = lane + 64 * u;
    if (q < (kStreamPiece + 16) / 4) place(q, fetch(q), 4 * q);
  }
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < len; base += kStreamPiece) {
    // the next 512 samples (behind the 16 already ahead) on their way while this piece is walked
    const bool more = base + kStreamPiece < len;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) nx[u] = fetch((base + kStreamPiece + 16) / 4 + lane + 64 * u);
    }
    const int steps_here = (len - base) < kStreamPiece ? (len - base) : kStreamPiece;
    const double* __restrict__ pa0 = win + oa0;
    const double* __restrict__ pb0 = win + ob0;
    const double* __restrict__ pa1 = win + oa1;
    const double* __restrict__ pb1 = win + ob1;
    const float* __restrict__ fb0 = xf + ob0;
    const float* __restrict__ fb1 = xf + ob1;
    const float* __restrict__ pw = ww + P + kq;  // w[t' + 1]
    auto step = [&](int kk) __attribute__((always_inline)) {
      if (WEIGHTED) {
        const float wk = pw[kk];
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa0[kk], (double)(wk * fb0[kk]), acc0, 0, 0, 0);
        if (NI > 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa1[kk], (double)(wk * fb1[kk]), acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa0[kk], pb0[kk], acc0, 0, 0, 0);
        if (NI > 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa1[kk], pb1[kk], acc1, 0, 0, 0);
      }
    };
    int k = 0;
    for (; k + 32 <= steps_here; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) step(k + 4 * u);
    }
    for (; k + 4 <= steps_here; k += 4) step(k);
    if (k < steps_here) {  // the block's last one to three steps: the missing ones multiply by 0 (x + 0 * y == x)
      const bool in = k + kq < steps_here;
      const float wk = (WEIGHTED && in) ? pw[k] : 0.0f;
      const double b0 = !in ? 0.0 : (WEIGHTED ? (double)(wk * fb0[k]) : pb0[k]);
      const double b1 = !in ? 0.0 : (WEIGHTED ? (double)(wk * fb1[k]) : pb1[k]);
      acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(in ? pa0[k] : 0.0, b0, acc0, 0, 0, 0);
      if (NI > 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(in ? pa1[k] : 0.0, b1, acc1, 0, 0, 0);
    }
    if (more) {
      __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order; the compiler must not move them)
      if (lane < 4) {
        const double2 t0 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * lane]);
        const double2 t1 = *reinterpret_cast<const double2*>(&win[kStreamPiece + 4 * lane + 2]);
        *reinterpret_cast<double2*>(&win[4 * lane]) = t0;
        *reinterpret_cast<double2*>(&win[4 * lane + 2]) = t1;
        if (WEIGHTED) {
          const float4 f0 = *reinterpret_cast<const float4*>(&xf[kStreamPiece + 4 * lane]);
          const float4 f1 = *reinterpret_cast<const float4*>(&ww[kStreamPiece + 4 * lane]);
          *reinterpret_cast<float4*>(&xf[4 * lane]) = f0;
          *reinterpret_cast<float4*>(&ww[4 * lane]) = f1;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) place((base + kStreamPiece + 16) / 4 + lane + 64 * u, nx[u], 16 + 4 * (lane + 64 * u));
      __builtin_amdgcn_wave_barrier();
    }
  }
  // D_b[i][j] sits in lane 16 i + 4 b + j
  const int oi = lane >> 4, oj = lane & 3;
  auto put = [&](int I, int J, double v) {
    const int ra = 4 * I + oi, cb = 4 * J + oj;
    if (cb == 0 && ra <= P) {
      out[ra] = v;
      if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + ra] = v;
    } else if (ra >= 1 && ra <= cb && cb <= P) {
      out[33 + (ra - 1) + (cb - 1) * P] = v;
      out[33 + (cb - 1) + (ra - 1) * P] = v;
    }
  };
  put(I0, J0, acc0);
  if (NI > 1) put(I1, J1, acc1);
  if (lane > P && lane <= 32) {
    out[lane] = 0.0;
    if (a.autocorr && !irls) a.autocorr[(size_t)sf * 33 + lane] = 0.0;
  }
}

// compute_raw_errors (lpc.rs:602-618: f32 fma chain over the taps of the UNWINDOWED samples), the sum of |error| as ONE
// sequential f32 chain (Iterator::sum, lpc.rs:839), the best step so far (lpc.rs:840-844) and the next step's weights
// (lpc.rs:845-847), one workgroup per subframe: state[0..32) = this step's solution, [32..64) = the best,
// [64] = its error, [65] = have one, [66] = status.
constexpr int kIrlsErrThreads = 256;
template <bool STEREO>
__global__ void __launch_bounds__(kIrlsErrThreads) direct_mse_irls_error_kernel(DirectMseArgs a) {
  // (f32)s of one 1024-sample chunk + the 32 samples in front of it: what every tap and the leading term convert to.
  // A window, not the block: 8.3 KB of LDS per workgroup keep eighteen subframes' sequential sums going per CU where
  // the whole block (64 KB at 16384 samples) kept two.
  __shared__ __attribute__((aligned(16))) float xs[kErrChunk + 32];
  __shared__ __attribute__((aligned(16))) float echunk[kErrChunk];
  __shared__ float cf[32];
  __shared__ int smax;
  __shared__ float ssum;
  const int tid = threadIdx.x;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  double* const st = a.irls_state + (size_t)sf * kIrlsStateDoubles;
  if (st[66] != 0.0) return;  // (uniform)
  const int n4 = (n + 3) & ~3;
  float* __restrict__ wout = a.irls_weights + (size_t)sf * n4;
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
  auto sample = [&](int t) -> int32_t {
    int32_t s = rowA[t];
    if (STEREO && kind >= 2) {
      const int32_t r = rowB[t];
      s = kind == 2 ? (s + r) >> 1 : s - r;  // coding.rs:483
    }
    return s;
  };
  const bool last = a.irls_step == a.mae_steps;  // no step follows: its weights are never read
  if (tid == 0) {
    smax = 0;
    ssum = 0.0f;
  }
  if (tid < 32) cf[tid] = tid < P ? (float)st[tid] : 0.0f;
  __syncthreads();
  float normalizer = 1.0f;
  if (!last) {  // the normaliser of the weights: max |s| over the block (lpc.rs:823-826)
    int my_maxabs = 0;
    for (int t = tid; t < n; t += kIrlsErrThreads) {
      const int32_t s = sample(t);
      const int32_t ab = s < 0 ? (int32_t)(0u - (uint32_t)s) : s;  // i32::abs (wrapping)
      my_maxabs = ab > my_maxabs ? ab : my_maxabs;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_xor(my_maxabs, d, 64);
      my_maxabs = o > my_maxabs ? o : my_maxabs;
    }
    if ((tid & 63) == 0) atomicMax(&smax, my_maxabs);
    __syncthreads();
    normalizer = (float)smax;
  }
  for (int base = 0; base < n; base += kErrChunk) {
    __syncthreads();  // the previous chunk's errors are summed, its window is free
    for (int o = tid; o < kErrChunk + 32; o += kIrlsErrThreads) {
      const int t = base - 32 + o;
      xs[o] = (t >= 0 && t < n) ? (float)sample(t) : 0.0f;
    }
    __syncthreads();
    for (int o = tid; o < kErrChunk && base + o < n; o += kIrlsErrThreads) {
      const int t = base + o;
      float e = 0.0f;  // raw_errors[t] for t < order: never written, 0
      if (t >= P) {
        e = -xs[32 + o];  // (f32)(-s) == -(f32)s: rounding is symmetric
        for (int j = 0; j < P; ++j) e = __builtin_fmaf(cf[j], xs[32 + o - 1 - j], e);
        if (!last) {
          float x = __builtin_fabsf(e);
          x = x > 1.0f ? x : 1.0f;
          x = x / normalizer;
          x = x > 0.01f ? x : 0.01f;
          wout[t] = dev_powf_pos(x, -1.2f);
        }
      } else if (!last && a.irls_step == 0) {
        wout[t] = 1.0f;  // the first `order` weights stay 1 (lpc.rs:821-822)
      }
      echunk[o] = __builtin_fabsf(e);
    }
    __syncthreads();
    if (tid == 0) {
      // (Iterator::sum is one sequential chain: sixteen values per trip come as four 16-byte reads, the additions
      // stay in order)
      float sacc = ssum;
      const int cnt = n - base < kErrChunk ? n - base : kErrChunk;
      int o = 0;
      for (; o + 16 <= cnt; o += 16) {
        const float4 q0 = *reinterpret_cast<const float4*>(&echunk[o]);
        const float4 q1 = *reinterpret_cast<const float4*>(&echunk[o + 4]);
        const float4 q2 = *reinterpret_cast<const float4*>(&echunk[o + 8]);
        const float4 q3 = *reinterpret_cast<const float4*>(&echunk[o + 12]);
        sacc += q0.x; sacc += q0.y; sacc += q0.z; sacc += q0.w;
        sacc += q1.x; sacc += q1.y; sacc += q1.z; sacc += q1.w;
        sacc += q2.x; sacc += q2.y; sacc += q2.z; sacc += q2.w;
        sacc += q3.x; sacc += q3.y; sacc += q3.z; sacc += q3.w;
      }
      for (; o < cnt; ++o) sacc += echunk[o];
      ssum = sacc;
    }
  }
  __syncthreads();
  const float best_error = a.irls_step == 0 ? 3.40282347e+38f : (float)st[64];  // f32::MAX
  if (ssum < best_error) {  // (uniform)
    if (tid < 32) st[32 + tid] = st[tid];
    if (tid <= 32 && a.autocorr) {
      const double* g = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
      a.autocorr[(size_t)sf * 33 + tid] = tid <= P ? g[tid] : 0.0;
    }
    if (tid == 0) {
      st[64] = (double)ssum;
      st[65] = 1.0;
    }
  } else if (a.irls_step == 0 && tid == 0) {
    st[64] = (double)best_error;
    st[65] = 0.0;
  }
}

// solve_sym_mut with the regulariser loop (lpc.rs:887-896: nalgebra's Cholesky, restated in the oracle's
// orc_cholesky_solve) + quantize_parameters (lpc.rs:273-302) for a batch, ONE SUBFRAME PER LANE: every operation of a
// subframe's sequence is the lane's own, in the reference's order; the matrices live in LDS, element k of lane l at
// k * LW + l.  LW = lanes in use per 64-thread workgroup = what 64 KB hold of (P * P + 2 P) doubles per subframe.
// PHASE 0: solve + quantise (no IRLS); 1: one IRLS step's solve -> irls_state (a failed estimate ends the subframe's
// iteration, lpc.rs:829-830); 2: after the last step, the best solution -> quantise.
template <int PHASE>
__global__ void __launch_bounds__(64) direct_mse_solve_kernel(DirectMseArgs a, int LW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x * (uint32_t)LW + (uint32_t)lane;
  if (lane >= LW || sf >= a.n_subframes) return;  // (no barriers below: lanes are independent)
  double* const m = reinterpret_cast<double*>(smem_raw) + lane;        // [P * P]
  double* const v = m + (size_t)P * P * LW;                            // [P]
  double* const diag = v + (size_t)P * LW;                             // [P]: the diagonal with the regulariser added so far
#define M_(r, c) m[(size_t)((r) + (c) * P) * LW]
#define V_(i) v[(size_t)(i) * LW]
  const double* __restrict__ g = a.gram_scratch + (size_t)sf * direct_mse_gram_stride((uint32_t)P);
  double* const st = PHASE != 0 ? a.irls_state + (size_t)sf * kIrlsStateDoubles : nullptr;
  int status = 0;
  if (PHASE == 1) {
    if (a.irls_step == 0) st[66] = 0.0;
    else if (st[66] != 0.0) return;
  }
  if (PHASE == 2) {
    status = (int)st[66];
    const bool have = st[65] != 0.0;
    if (status == 0 && !have) status = FLACENC_HIP_SUBFRAME_NONFINITE;  // best_coefs.unwrap() would panic
    for (int i = 0; i < P; ++i) V_(i) = (status == 0) ? st[32 + i] : 0.0;
  }
  double regularizer = 0.0;
  int tries = 0;
  if (PHASE != 2) for (int i = 0; i < P; ++i) diag[(size_t)i * LW] = g[33 + i + i * P];
  for (; PHASE != 2;) {
    // mat.clone(); xy = corr[1..]
    for (int c = 0; c < P; ++c)
      for (int r = 0; r < P; ++r) M_(r, c) = r == c ? diag[(size_t)r * LW] : g[33 + r + c * P];
    for (int i = 0; i < P; ++i) V_(i) = g[i + 1];
    bool ok = true;
    for (int j = 0; j < P && ok; ++j) {
      for (int k = 0; k < j; ++k) {
        const double factor = -M_(j, k);
        for (int r = j; r < P; ++r) {
          const double ax = factor * M_(r, k);
          M_(r, j) = ax + M_(r, j);  // array_axcpy: (a * x) * 1 + 1 * y, no fma
        }
      }
      const double dj = M_(j, j);
      if (dj == 0.0 || !(dj >= 0.0)) {  // is_zero() / try_sqrt() == None
        ok = false;
        break;
      }
      const double denom = __builtin_sqrt(dj);
      M_(j, j) = denom;
      for (int r = j + 1; r < P; ++r) M_(r, j) = M_(r, j) / denom;
    }
    if (ok) {
      // solve_lower_triangular_vector_unchecked_mut
      for (int i = 0; i < P; ++i) {
        const double coeff = V_(i) / M_(i, i);
        V_(i) = coeff;
        const double na = -coeff;
        for (int r = i + 1; r < P; ++r) V_(r) = (na * M_(r, i)) + V_(r);
      }
      // ad_solve_lower_triangular: b[i] = (b[i] - dot(L[i+1.., i], b[i+1..])) / L[i][i], dotx's eight accumulators
      for (int i = P - 1; i >= 0; --i) {
        const int rows = P - (i + 1);
        double acc8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double res = 0.0;
        int q = 0;
        while (rows - q >= 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc8[u] += M_(i + 1 + q + u, i) * V_(i + 1 + q + u);
          q += 8;
        }
        res += acc8[0] + acc8[4];
        res += acc8[1] + acc8[5];
        res += acc8[2] + acc8[6];
        res += acc8[3] + acc8[7];
        for (int k = q; k < rows; ++k) res += M_(i + 1 + k, i) * V_(i + 1 + k);
        V_(i) = (V_(i) - res) / M_(i, i);
      }
      break;
    }
    // regularizer = max(1, 2 regularizer); diag += regularizer - old (lpc.rs:889-895)
    const double old = regularizer;
    const double twice = regularizer + regularizer;
    regularizer = 1.0 > twice ? 1.0 : twice;
    for (int i = 0; i < P; ++i) diag[(size_t)i * LW] += regularizer - old;
    if (++tries > 2000) {  // (NaN input: the reference would not terminate)
      status = FLACENC_HIP_SUBFRAME_NONFINITE;
      for (int i = 0; i < P; ++i) V_(i) = 0.0;
      break;
    }
  }
  if (PHASE == 1) {
    for (int i = 0; i < 32; ++i) st[i] = i < P ? V_(i) : 0.0;
    st[66] = (double)status;
    return;
  }
  // ---- quantize_parameters, lpc.rs:273-302 (find_shift :234-254, quantize_parameter :258-270) ----
  for (int i = 0; i < P; ++i) {
    const uint64_t b = (uint64_t)__double_as_longlong(V_(i));
    if (((b >> 52) & 0x7FF) == 0x7FF) status |= FLACENC_HIP_SUBFRAME_NONFINITE;
  }
  int32_t* pr = a.pred_out + (size_t)sf * 36;
  for (int i = 0; i < 36; ++i) pr[i] = 0;
  int shift = 0, order = 0;
  if (status == 0) {
    double max_abs = 0.0;
    for (int i = 0; i < P; ++i) max_abs = fmax(max_abs, fabs(V_(i)));
    int abs_log2 = dm_ceil_log2_pos(max_abs);
    if (abs_log2 < -32752) abs_log2 = -32752;
    const int precision = (int)a.precision;
    shift = (precision - 1) - abs_log2;
    shift = shift < 0 ? 0 : (shift > 15 ? 15 : shift);
    const double scalefac = (double)(1 << shift);
    const int lo = -(1 << (precision - 1)), hi = (1 << (precision - 1)) - 1;
    order = 1;
    for (int i = 0; i < P; ++i) {
      double s = round(V_(i) * scalefac);  // half away from zero
      s = s < -32768.0 ? -32768.0 : (s > 32767.0 ? 32767.0 : s);
      int q = (int)s;
      q = q < lo ? lo : (q > hi ? hi : q);
      pr[i] = q;
      if (q != 0) order = i + 1;  // tail-zero truncation, min 1
    }
    for (int i = order; i < P; ++i) pr[i] = 0;
  }
  pr[32] = order;
  pr[33] = shift;
  pr[34] = status;
  if (a.lpc_coefs)
    for (int i = 0; i < 32; ++i) a.lpc_coefs[(size_t)sf * 32 + i] = (i < P && status == 0) ? V_(i) : 0.0;
#undef M_
#undef V_
}

}  // namespace

size_t direct_mse_lds_bytes(uint32_t block_size, bool irls, uint32_t order) {
  const size_t n4 = ((size_t)block_size + 3) & ~(size_t)3;
  size_t b = n4 * 4;
  if (irls) b += (block_size > 16384u ? 0 : n4 * 4) + kErrChunk * 4;  // (above 16384 the weights live in HBM scratch)
  const size_t pp = ((size_t)order * order + 1) & ~(size_t)1;
  b += (pp * 2 + 33 + 32 * 3) * 8 + 64;
  return (b + 15) & ~(size_t)15;
}

hipError_t launch_direct_mse(const DirectMseArgs& a, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  if (a.lpc_order < 1 || a.lpc_order > 32) return hipErrorInvalidValue;
  if (a.stereo && (a.n_subframes & 3u)) return hipErrorInvalidValue;
  const bool irls = a.mae_steps > 0;
  auto launch_solve = [&](const DirectMseArgs& d, int phase) -> hipError_t {
    const size_t per = ((size_t)d.lpc_order * d.lpc_order + 2 * d.lpc_order) * 8;  // LDS per subframe
    int lw = 64;
    while (lw > 1 && per * (size_t)lw > 64 * 1024) lw >>= 1;
    const size_t solve_smem = per * (size_t)lw;
    static DynamicLdsOptIn solve_opt[3];
    const dim3 grid((d.n_subframes + (uint32_t)lw - 1u) / (uint32_t)lw);
#define FLACENC_DM_SOLVE(PH)                                                                                             \
  {                                                                                                                      \
    auto kern = direct_mse_solve_kernel<PH>;                                                                             \
    if (hipError_t e = solve_opt[PH].ensure(reinterpret_cast<const void*>(kern), solve_smem); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, dim3(64), solve_smem, stream, d, lw);                                                 \
  }
    if (phase == 0) FLACENC_DM_SOLVE(0) else if (phase == 1) FLACENC_DM_SOLVE(1) else FLACENC_DM_SOLVE(2)
#undef FLACENC_DM_SOLVE
    return hipGetLastError();
  };
  const bool streamable = a.gram_scratch != nullptr && (reinterpret_cast<uintptr_t>(a.samples) & 15) == 0 && (a.stride & 3) == 0;
  // the sliding-window chains: NI instructions of four 4 x 4 blocks, dealt to 1..4 waves with NIW accumulators each
  const uint32_t nb = (a.lpc_order + 1 + 3) >> 2;
  const uint32_t ni = (nb + nb * (nb + 1) / 2 - 1 + 3) / 4;  // 1 .. 14
  const uint32_t nw = (ni + 3) / 4, niw = (ni + nw - 1) / nw;
  auto launch_stream = [&](const DirectMseArgs& d, bool weighted) -> hipError_t {
    if (d.lpc_order <= 11) {
      if (d.stereo) {
        if (weighted) hipLaunchKernelGGL((direct_mse_stream_small_kernel<true, true>), dim3(d.n_subframes), dim3(64), 0, stream, d);
        else hipLaunchKernelGGL((direct_mse_stream_small_kernel<true, false>), dim3(d.n_subframes), dim3(64), 0, stream, d);
      } else {
        if (weighted) hipLaunchKernelGGL((direct_mse_stream_small_kernel<false, true>), dim3(d.n_subframes), dim3(64), 0, stream, d);
        else hipLaunchKernelGGL((direct_mse_stream_small_kernel<false, false>), dim3(d.n_subframes), dim3(64), 0, stream, d);
      }
      return hipGetLastError();
    }
    if (d.lpc_order <= 31) {
      if (d.stereo) {
        if (weighted) hipLaunchKernelGGL((direct_mse_stream_tiles_kernel<true, true>), dim3(d.n_subframes), dim3(256), 0, stream, d);
        else hipLaunchKernelGGL((direct_mse_stream_tiles_kernel<true, false>), dim3(d.n_subframes), dim3(256), 0, stream, d);
      } else {
        if (weighted) hipLaunchKernelGGL((direct_mse_stream_tiles_kernel<false, true>), dim3(d.n_subframes), dim3(256), 0, stream, d);
        else hipLaunchKernelGGL((direct_mse_stream_tiles_kernel<false, false>), dim3(d.n_subframes), dim3(256), 0, stream, d);
      }
      return hipGetLastError();
    }
    const dim3 grid(d.n_subframes), block(64 * nw);
#define FLACENC_DM_STREAM(ST, WT, N_) hipLaunchKernelGGL((direct_mse_stream_kernel<ST, WT, N_>), grid, block, 0, stream, d);
#define FLACENC_DM_STREAM_N(ST, WT)                        \
  {                                                        \
    if (niw == 1) FLACENC_DM_STREAM(ST, WT, 1)             \
    else if (niw == 2) FLACENC_DM_STREAM(ST, WT, 2)        \
    else if (niw == 3) FLACENC_DM_STREAM(ST, WT, 3)        \
    else FLACENC_DM_STREAM(ST, WT, 4)                      \
  }
    if (d.stereo) {
      if (weighted) FLACENC_DM_STREAM_N(true, true) else FLACENC_DM_STREAM_N(true, false)
    } else {
      if (weighted) FLACENC_DM_STREAM_N(false, true) else FLACENC_DM_STREAM_N(false, false)
    }
#undef FLACENC_DM_STREAM_N
#undef FLACENC_DM_STREAM
    return hipGetLastError();
  };
  // (orders up to 11: 4 x 4 blocks on one wave; 12..31: the four 16 x 16 tiles on four waves -- the 4 x 4 block form is
  // LDS-bound there, two 512-byte f64 operand reads per 16-cycle MFMA on each of four SIMDs: 2.40 against 1.87 ms per
  // 768 frames of 16384 samples at order 24; order 32: 4 x 4 blocks on four waves, whose third tile row would be mostly
  // padding: 5.6 -> 3.8 ms)
  if (!irls && streamable) {
    if (hipError_t e = launch_stream(a, false); e != hipSuccess) return e;
    return launch_solve(a, 0);
  }
  if (irls && streamable && a.weight_scratch != nullptr && a.irls_state != nullptr) {
    // IRLS: per step the (weighted) chains, the batched solve, the error pass; then the best solution
    DirectMseArgs d = a;
    d.irls_weights = a.weight_scratch;
    for (uint32_t it = 0; it <= a.mae_steps; ++it) {
      d.irls_step = it;
      // (the first step's weights are 1: the unweighted chain, operand for operand)
      if (hipError_t e = launch_stream(d, it != 0); e != hipSuccess) return e;
      if (hipError_t e = launch_solve(d, 1); e != hipSuccess) return e;
      if (a.stereo) hipLaunchKernelGGL(direct_mse_irls_error_kernel<true>, dim3(a.n_subframes), dim3(kIrlsErrThreads), 0, stream, d);
      else hipLaunchKernelGGL(direct_mse_irls_error_kernel<false>, dim3(a.n_subframes), dim3(kIrlsErrThreads), 0, stream, d);
      if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return launch_solve(d, 2);
  }
  const size_t smem = direct_mse_lds_bytes(a.block_size, irls, a.lpc_order);
  if (smem > 160 * 1024) return hipErrorNotSupported;
  const uint32_t P = a.lpc_order;
  const uint32_t nc = (P + 1) + P * (P + 1) / 2;
  const uint32_t threads = ((nc < 64 ? 64 : nc) + 63u) & ~63u;  // <= 576
  static DynamicLdsOptIn opt[6];
#define FLACENC_DM_LAUNCH(ST, IR, SLOT, WGT)                                                                       \
  {                                                                                                           \
    auto kern = direct_mse_kernel<ST, IR, WGT>;                                                                    \
    if (hipError_t e = opt[SLOT].ensure(reinterpret_cast<const void*>(kern), smem); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, dim3(a.n_subframes), dim3(threads), smem, stream, a);                            \
  }
  const bool wg = irls && a.block_size > 16384u;
  if (wg && a.weight_scratch == nullptr) return hipErrorInvalidValue;
  if (a.stereo) {
    if (wg) FLACENC_DM_LAUNCH(true, true, 4, true) else if (irls) FLACENC_DM_LAUNCH(true, true, 0, false) else FLACENC_DM_LAUNCH(true, false, 1, false)
  } else {
    if (wg) FLACENC_DM_LAUNCH(false, true, 5, true) else if (irls) FLACENC_DM_LAUNCH(false, true, 2, false) else FLACENC_DM_LAUNCH(false, false, 3, false)
  }
#undef FLACENC_DM_LAUNCH
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (!irls && a.gram_scratch != nullptr) return launch_solve(a, 0);
  return hipSuccess;
}

}  // namespace flacenc_hip
