// mfma_i8_fir.hip -- the gated experiment of VERDICT r3 #1: compute_error's FIR (src/lpc.rs:306-350) as a banded-
// Toeplitz GEMM on v_mfma_i32_16x16x64_i8, against the v_mad_i64_i32 window walk of bigblock_residual_kernel.
//
//   e[t] = s[t] - ((sum_{j<P} c_j s[t-1-j]) >> shift)  =  -(X[t] >> shift),
//   X[t] = sum_{j=-1}^{P-1} c'_j s[t-1-j],  c'_{-1} = -2^shift   (the "s[t] -" rides along as one more tap)
//
// GEMM form: a tile is 16 chunks of 16 samples; D[i][n] = X[16 c(n) + i] = sum_m A[i][m] B[m][n] with
// A[i][m] = c'[i - 1 - m] (Toeplitz in the coefficients, one operand per subframe) and B[m][n] = s[16 c(n) + m],
// m = -32..31 (K = 64; m = 16..31 multiplies zeros of A).  Exact integers: samples as two's complement bytes
// with the low ones biased by -128 (b ^ 0x80 read as i8; the bias is a per-subframe constant (128 + 2^15 [+ 2^23])
// * sum c' that enters through the accumulators' initial values), coefficients as two signed digits.  Limb products
// of equal weight chain through C, so NL sample limbs x 2 coefficient limbs cost 2 NL MFMAs and leave NL + 1
// accumulators, recombined as lo = a0 + a1 << 8, hi = a2 + a3 << 8 [+ a4 << 16], X = hi * 2^16 + lo (one
// v_mad_i64_i32), e = K - alignbit(X, shift) -- or, as the kernel below does it, samples stored complemented so
// that the accumulators hold -X + 2^shift - 1 = hi 2^16 + lo and e = (hi << (16 - shift)) + (lo >> shift).
//
// Part 1 probes the operand lane maps of the instruction with one-hot operands; part 2 checks both kernels against a
// CPU FIR; part 3 times them (whole launches by events, the same rows, 4 waves per workgroup, one row per wave).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

// ---------------------------------------------------------------- part 1: lane maps
__global__ void probe(const v4i* A, const v4i* B, v4i* D, int n) {
  const int l = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i * 64 + l], B[i * 64 + l], acc, 0, 0, 0);
    D[i * 64 + l] = acc;
  }
}

// ---------------------------------------------------------------- the two FIR kernels
constexpr int kN = 4096;  // samples per row (one pass of the big-block kernels)

struct RowCoefs {
  int32_t c[32];
  int32_t order;
  int32_t shift;
};

// (a) the window walk of bigblock_residual_kernel: lane l owns samples [64 l, 64 l + 64), 64-bit multiply-adds
constexpr int kSeg = 68;
constexpr int kBufDwords = 65 * kSeg + 8;
__device__ __forceinline__ int widx(int t) { return ((t >> 6) + 1) * kSeg + (t & 63); }

template <int MAXP>
__global__ void __launch_bounds__(256, 3) fir_valu(const int32_t* __restrict__ x, const RowCoefs* __restrict__ rc,
                                                   int32_t* __restrict__ out, int rows, int reps, int32_t* __restrict__ sink) {
  constexpr int HP = (MAXP + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int row = blockIdx.x * 4 + wave;
  if (row >= rows) row = rows - 1;
  int32_t* const buf = sm + wave * kBufDwords;
  const int32_t* __restrict__ src = x + (size_t)row * kN;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int t = (lane + it * 64) << 2;
    *reinterpret_cast<int4*>(&buf[widx(t)]) = *reinterpret_cast<const int4*>(src + t);
  }
  if (lane < 16) *reinterpret_cast<int4*>(&buf[widx((lane << 2) - 64)]) = make_int4(0, 0, 0, 0);
  __syncthreads();
  int32_t cq[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) cq[i] = __builtin_amdgcn_readfirstlane(rc[row].c[i]);
  const int shift = __builtin_amdgcn_readfirstlane(rc[row].shift);
  const int warm = __builtin_amdgcn_readfirstlane(rc[row].order);
  const int tl = lane << 6;
  int sw[HP + 16];
#pragma unroll
  for (int q = 0; q < HP; q += 4) {
    const int4 v = *reinterpret_cast<const int4*>(&buf[widx(tl - HP + q)]);
    sw[q] = v.x; sw[q + 1] = v.y; sw[q + 2] = v.z; sw[q + 3] = v.w;
  }
  int32_t* __restrict__ orow = out + (size_t)row * kN;
  // reps > 1: the walk is repeated without its loads from and stores to HBM (results folded into a checksum), so that
  // time(reps) - time(1) is the arithmetic alone
  int32_t fold = 0;
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
  if (rep > 0) {
#pragma unroll
    for (int q = 0; q < HP; q += 4) {
      const int4 v = *reinterpret_cast<const int4*>(&buf[widx(tl - HP + q)]);
      sw[q] = v.x; sw[q + 1] = v.y; sw[q + 2] = v.z; sw[q + 3] = v.w;
    }
  }
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int t0 = tl + 16 * i;
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
      const int4 v = *reinterpret_cast<const int4*>(&buf[widx(t0 + q)]);
      sw[HP + q] = v.x; sw[HP + q + 1] = v.y; sw[HP + q + 2] = v.z; sw[HP + q + 3] = v.w;
    }
    int32_t e[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int64_t pred = 0;
#pragma unroll
      for (int j = 0; j < MAXP; ++j) pred += (int64_t)cq[j] * (int64_t)sw[HP + q - 1 - j];
      e[q] = (int32_t)(uint32_t)(uint64_t)((int64_t)sw[HP + q] - (pred >> shift));
      if (t0 + q < warm) e[q] = 0;
    }
    if (rep == 0) {
#pragma unroll
      for (int q = 0; q < 16; q += 4)
        *reinterpret_cast<int4*>(orow + t0 + q) = make_int4(e[q], e[q + 1], e[q + 2], e[q + 3]);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) fold ^= e[q];
    }
#pragma unroll
    for (int q = 0; q < HP; ++q) sw[q] = sw[q + 16];
  }
  }
  if (reps > 1) sink[(size_t)row * 64 + lane] = fold;
}

// (b) the matrix-core form.  LDS per wave: NL limb planes of (32 + 4096) bytes (two chunks of zeros in front).
// LAYOUT 0: lane l holds k = 16 (l >> 4) + j in byte j of its 16 operand bytes; LAYOUT 1: k = 8 (l >> 4) + (j & 7)
// + 32 (j >> 3) (two K = 32 halves).  The probe says which one the hardware has.
constexpr int kPlane = 32 + kN;

template <int NL>
__device__ __forceinline__ void pack_limbs(const int4 v, uint32_t (&L)[NL]) {
  const uint32_t u0 = (uint32_t)v.x, u1 = (uint32_t)v.y, u2 = (uint32_t)v.z, u3 = (uint32_t)v.w;
  // byte transpose 4 x 4 (v_perm_b32: selector byte k picks byte k of D from {src0 : src1} = bytes 7..4 : 3..0)
  const uint32_t a_lo = __builtin_amdgcn_perm(u1, u0, 0x05010400u);  // u0.b0 u1.b0 u0.b1 u1.b1
  const uint32_t b_lo = __builtin_amdgcn_perm(u3, u2, 0x05010400u);
  L[0] = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u) ^ 0x7F7F7F7Fu;
  if (NL >= 2) {
    const uint32_t l1 = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
    L[1] = NL > 2 ? (l1 ^ 0x7F7F7F7Fu) : ~l1;
  }
  if (NL >= 3) {
    const uint32_t a_hi = __builtin_amdgcn_perm(u1, u0, 0x07030602u);  // u0.b2 u1.b2 u0.b3 u1.b3
    const uint32_t b_hi = __builtin_amdgcn_perm(u3, u2, 0x07030602u);
    const uint32_t l2 = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
    L[2] = NL > 3 ? (l2 ^ 0x7F7F7F7Fu) : ~l2;
    if (NL >= 4) L[3] = ~__builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
  }
}

template <int NL, int LAYOUT>
__global__ void __launch_bounds__(256, 3) fir_mfma(const int32_t* __restrict__ x, const RowCoefs* __restrict__ rc,
                                                   int32_t* __restrict__ out, int rows, int reps, int32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int row = blockIdx.x * 4 + wave;
  if (row >= rows) row = rows - 1;
  unsigned char* const planes = smem_raw + wave * (NL * kPlane + 512);
  int32_t* const ctab = reinterpret_cast<int32_t*>(planes + NL * kPlane);  // 128 ints: c'[x - 48], x = tap + 48
  const int32_t* __restrict__ src = x + (size_t)row * kN;
  // ---- load + limb split: lane handles quads lane + 64 it
  int4 v[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) v[it] = *reinterpret_cast<const int4*>(src + ((lane + it * 64) << 2));
  const int shift = __builtin_amdgcn_readfirstlane(rc[row].shift);
  const int warm = __builtin_amdgcn_readfirstlane(rc[row].order);
  // coefficient table with zero padding either side: tap -1 = -2^shift, taps 0..31 = c
  ctab[lane] = 0;
  ctab[lane + 64] = 0;
  __builtin_amdgcn_wave_barrier();
  if (lane < 32) ctab[48 + lane] = rc[row].c[lane];
  if (lane == 32) ctab[47] = -(1 << shift);
  if (lane < 8) {  // the 32 samples in front of the row are 0, stored complemented like every sample: 0xFF ^ 0x80 in the biased planes, 0xFF in the top one
#pragma unroll
    for (int a = 0; a < NL; ++a)
      *reinterpret_cast<uint32_t*>(planes + a * kPlane + 4 * lane) = a + 1 < NL ? 0x7F7F7F7Fu : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    uint32_t L[NL];
    pack_limbs<NL>(v[it], L);
    const int byte = 32 + ((lane + it * 64) << 2);
#pragma unroll
    for (int a = 0; a < NL; ++a) *reinterpret_cast<uint32_t*>(planes + a * kPlane + byte) = L[a];
  }
  __builtin_amdgcn_wave_barrier();
  // ---- A operand: row i = lane & 15, k block kb = lane >> 4; tap(i, k) = i + 31 - k
  const int i = lane & 15, kb = lane >> 4;
  v4i A0, A1;
  int csum = 0;
  {
    uint32_t w0[4], w1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t p0 = 0, p1 = 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * q + jj;
        const int k = LAYOUT == 0 ? 16 * kb + j : 8 * kb + (j & 7) + 32 * (j >> 3);
        const int c = ctab[48 + i + 31 - k];
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
    // sum of c' over all taps (wave-uniform): lanes 0..32 hold one each
    int mine = lane <= 32 ? ctab[47 + lane] : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mine += __shfl_xor(mine, d, 64);
    csum = __builtin_amdgcn_readfirstlane(mine);
  }
  // ---- accumulator seeds.  The planes hold the bytes of ~s = -s - 1, so the limb sums give -X - csum - (bias) csum:
  // the seeds add back (bias + 1) csum and 2^shift - 1, and what the accumulators then hold is Y = -X + 2^shift - 1
  // with e = -(X >> shift) = Y >> shift (ceiling of -X / 2^shift), split as Y = hi 2^16 + lo:
  //   e = (hi << (16 - shift)) + (lo >> shift)   (arithmetic shift; hi 2^16 is a multiple of 2^shift)
  //   NL = 2: bias 128                -> a0 += 128 csum
  //   NL = 3: bias 128 + 2^15         -> a0 += 128 csum, a1 += (csum & 1) << 7, a2 += csum >> 1
  //   NL = 4: bias 128 + 2^15 + 2^23  -> ... a2 += (csum & 1) << 7, a3 += csum >> 1
  const int half = csum >> 1, odd = (csum & 1) << 7;
  int s0 = 129 * csum + (1 << shift) - 1;
  int s1 = NL >= 3 ? odd : 0;
  int s2 = NL == 3 ? half : (NL == 4 ? half + odd : 0);
  int s3 = NL == 4 ? half : 0;
  v4i seed0 = {s0, s0, s0, s0}, seed1 = {s1, s1, s1, s1}, seed2 = {s2, s2, s2, s2}, seed3 = {s3, s3, s3, s3};
  // (opaque copies: left as wave-uniform values the compiler rebuilds the four quads from SGPRs in every tile)
  asm volatile("" : "+v"(seed0), "+v"(seed1), "+v"(seed2), "+v"(seed3));
  const v4i zero = {0, 0, 0, 0};
  const int lsh = 16 - shift;
  int32_t* __restrict__ orow = out + (size_t)row * kN;
  const int n = i;
  // ---- tiles: T = 4 Q + r; column n is chunk 64 Q + 4 n + r
  int32_t fold = 0;
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep)
#pragma unroll 1
  for (int T = 0; T < 16; ++T) {
    const int Q = T >> 2, r = T & 3;
    const int chunk = 64 * Q + 4 * n + r;  // the column's chunk
    v4i B[NL];
#pragma unroll
    for (int a = 0; a < NL; ++a) {
      if (LAYOUT == 0) {
        B[a] = *reinterpret_cast<const v4i*>(planes + a * kPlane + 16 * (chunk + kb));  // chunk + kb - 2, + 32 bytes
      } else {
        const int2 lo = *reinterpret_cast<const int2*>(planes + a * kPlane + 16 * chunk + 8 * kb);        // m = 8 kb - 32 ..
        const int2 hi = *reinterpret_cast<const int2*>(planes + a * kPlane + 16 * (chunk + 2) + 8 * kb);  // m = 8 kb ..
        B[a] = v4i{lo.x, lo.y, hi.x, hi.y};
      }
    }
    v4i a0, a1, a2 = zero, a3 = zero, a4 = zero;
    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[0], seed0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[0], seed1, 0, 0, 0);
    if (NL >= 2) {
      a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[1], seed2, 0, 0, 0);
    }
    if (NL >= 3) {
      a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[2], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[2], seed3, 0, 0, 0);
    }
    if (NL >= 4) {
      a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[3], a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[3], zero, 0, 0, 0);
    }
    int32_t e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int32_t lo = (int32_t)((uint32_t)a0[q] + ((uint32_t)a1[q] << 8));  // |lo| < 2^30
      uint32_t hi = (uint32_t)a2[q];
      if (NL >= 3) hi += (uint32_t)a3[q] << 8;
      if (NL >= 4) hi += (uint32_t)a4[q] << 16;
      e[q] = (int32_t)((hi << lsh) + (uint32_t)(lo >> shift));
    }
    const int t0 = 16 * chunk + 4 * kb;  // rows 4 kb .. 4 kb + 3 of the column
    if (T < 2) {  // (the warm-up of orders above 16 reaches chunk 1 = column 0 of tile 1)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t0 + q < warm) e[q] = 0;
    }
    if (rep == 0) *reinterpret_cast<int4*>(orow + t0) = make_int4(e[0], e[1], e[2], e[3]);
    else fold ^= e[0] ^ e[1] ^ e[2] ^ e[3];
  }
  if (reps > 1) sink[(size_t)row * 64 + lane] = fold;
}

// ---------------------------------------------------------------- host
static void cpu_fir(const int32_t* x, const RowCoefs& rc, int32_t* e) {
  for (int t = 0; t < kN; ++t) {
    int64_t pred = 0;
    for (int j = 0; j < 32; ++j) {
      const int64_t s = t - 1 - j >= 0 ? x[t - 1 - j] : 0;
      pred += (int64_t)rc.c[j] * s;
    }
    e[t] = t < rc.order ? 0 : (int32_t)(uint32_t)(uint64_t)((int64_t)x[t] - (pred >> rc.shift));
  }
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? std::atoi(argv[1]) : 24576;
  // ---------------- part 1
  int layout = -1;
  {
    // A one-hot at (lane la, byte ja) with value 1; B byte value = jb (pass 0) or lb (pass 1) everywhere
    const int NP = 64 * 16 * 2;
    std::vector<int32_t> hA((size_t)NP * 64 * 4, 0), hB((size_t)NP * 64 * 4), hD((size_t)NP * 64 * 4);
    for (int la = 0; la < 64; ++la)
      for (int ja = 0; ja < 16; ++ja)
        for (int pass = 0; pass < 2; ++pass) {
          const size_t p = ((size_t)(la * 16 + ja) * 2 + pass);
          hA[(p * 64 + la) * 4 + ja / 4] = 1 << (8 * (ja % 4));
          for (int lb = 0; lb < 64; ++lb)
            for (int q = 0; q < 4; ++q) {
              uint32_t w = 0;
              for (int jj = 0; jj < 4; ++jj) w |= (uint32_t)(pass == 0 ? 4 * q + jj : lb) << (8 * jj);
              hB[(p * 64 + lb) * 4 + q] = (int32_t)w;
            }
        }
    v4i *dA, *dB, *dD;
    CHECK(hipMalloc(&dA, hA.size() * 4));
    CHECK(hipMalloc(&dB, hB.size() * 4));
    CHECK(hipMalloc(&dD, hD.size() * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, NP);
    CHECK(hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost));
    // D[lane ld][reg]: row = 4 (ld >> 4) + reg, col = ld & 15 (the dtype-independent C/D map; verified below by the
    // FIR check).  For A position (la, ja): the non-zero row is A's row; per column the (jb, lb) found say which B
    // element shares its k.
    bool ok0 = true, ok1 = true, rows_ok = true;
    for (int la = 0; la < 64; ++la)
      for (int ja = 0; ja < 16; ++ja) {
        const size_t p0 = (size_t)(la * 16 + ja) * 2, p1 = p0 + 1;
        for (int ld = 0; ld < 64; ++ld)
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * (ld >> 4) + reg, col = ld & 15;
            const int jb = hD[(p0 * 64 + ld) * 4 + reg], lb = hD[(p1 * 64 + ld) * 4 + reg];
            if (row != (la & 15)) {
              if (jb != 0 || lb != 0) rows_ok = false;
              continue;
            }
            // the B element at (k of A(la, ja), col): hypothesis 0 and 1 both say lb = 16 (la >> 4) + col, jb = ja
            if (!(lb == 16 * (la >> 4) + col && jb == ja)) ok0 = ok1 = false;
            if (la == 17 && ja < 2 && col < 2) std::printf("  A(lane 17, byte %d) col %d pairs with B(lane %d, byte %d)\n", ja, col, lb, jb);
          }
      }
    std::printf("probe: A row = lane & 15: %s; B col = lane & 15 and same (lane >> 4, byte) as A: %s\n",
                rows_ok ? "yes" : "NO", ok0 ? "yes" : "NO");
    // which k a (lane >> 4, byte) pair is does not matter for a product as long as A and B agree -- but the FIR puts a
    // Toeplitz structure on k, so it does here; the FIR check below decides between the two candidate orders
    (void)ok1;
    (void)hipFree(dA);
    (void)hipFree(dB);
    (void)hipFree(dD);
  }
  // ---------------- parts 2 and 3
  std::mt19937 rng(12345);
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int bits = cfg == 0 ? 24 : (cfg == 1 ? 25 : (cfg == 2 ? 24 : 16));
    const int order = cfg == 2 ? 32 : (cfg == 3 ? 8 : 24);
    const int NL = (bits + 7) / 8;
    std::vector<int32_t> hx((size_t)rows * kN), he((size_t)rows * kN), hr((size_t)rows * kN);
    std::vector<RowCoefs> hc(rows);
    std::uniform_int_distribution<int> ds(-(1 << (bits - 1)), (1 << (bits - 1)) - 1), dc(-(1 << 14), (1 << 14) - 1), dsh(0, 15);
    for (auto& s : hx) s = ds(rng);
    // extremes in the first rows
    for (int t = 0; t < kN; ++t) {
      hx[t] = (t & 1) ? (1 << (bits - 1)) - 1 : -(1 << (bits - 1));
      hx[kN + t] = (1 << (bits - 1)) - 1;
      hx[2 * kN + t] = -(1 << (bits - 1));
    }
    for (int r = 0; r < rows; ++r) {
      std::memset(&hc[r], 0, sizeof(RowCoefs));
      for (int j = 0; j < order; ++j) hc[r].c[j] = r == 1 ? (1 << 14) - 1 : (r == 2 ? -(1 << 14) : dc(rng));
      hc[r].order = order;
      hc[r].shift = r < 16 ? r : dsh(rng);
    }
    const int check_rows = rows < 64 ? rows : 64;
    for (int r = 0; r < check_rows; ++r) cpu_fir(&hx[(size_t)r * kN], hc[r], &hr[(size_t)r * kN]);
    int32_t *dx, *de, *dsink;
    CHECK(hipMalloc(&dsink, (size_t)rows * 64 * 4));
    RowCoefs* dc_;
    CHECK(hipMalloc(&dx, hx.size() * 4));
    CHECK(hipMalloc(&de, he.size() * 4));
    CHECK(hipMalloc(&dc_, hc.size() * sizeof(RowCoefs)));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dc_, hc.data(), hc.size() * sizeof(RowCoefs), hipMemcpyHostToDevice));
    const int blocks = (rows + 3) / 4;
    auto check = [&](const char* name) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(he.data(), de, (size_t)check_rows * kN * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < (size_t)check_rows * kN; ++i) bad += he[i] != hr[i];
      std::printf("  %-28s %s (%zu of %zu differ)\n", name, bad ? "MISMATCH" : "== CPU", bad, (size_t)check_rows * kN);
      return bad == 0;
    };
    std::printf("config: %d-bit samples, order %d, %d rows of %d\n", bits, order, rows, kN);
    const size_t smem_v = 4 * kBufDwords * 4;
    static bool valu_opt_in = false;
    if (!valu_opt_in) {
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_valu<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_v));
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_valu<24>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_v));
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_valu<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_v));
      valu_opt_in = true;
    }
    int reps = 1;
    auto run_valu = [&]() {
      if (order <= 8) hipLaunchKernelGGL(fir_valu<8>, dim3(blocks), dim3(256), smem_v, 0, dx, dc_, de, rows, reps, dsink);
      else if (order <= 24) hipLaunchKernelGGL(fir_valu<24>, dim3(blocks), dim3(256), smem_v, 0, dx, dc_, de, rows, reps, dsink);
      else hipLaunchKernelGGL(fir_valu<32>, dim3(blocks), dim3(256), smem_v, 0, dx, dc_, de, rows, reps, dsink);
    };
    CHECK(hipMemset(de, 0xFF, he.size() * 4));
    run_valu();
    check("v_mad_i64_i32 window walk");
    const double t_valu = time_ms(run_valu, 20);
    reps = 5;
    const double t_valu5 = time_ms(run_valu, 10);
    reps = 1;
    auto run_mfma = [&](int lay) {
      const size_t smem_m = 4 * (size_t)(NL * kPlane + 512);
#define LAUNCH(NL_, LAY_)                                                                                          \
  {                                                                                                                \
    static bool once = false;                                                                                      \
    if (!once) {                                                                                                   \
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_mfma<NL_, LAY_>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_m));                         \
      once = true;                                                                                                 \
    }                                                                                                              \
    hipLaunchKernelGGL((fir_mfma<NL_, LAY_>), dim3(blocks), dim3(256), smem_m, 0, dx, dc_, de, rows, reps, dsink); \
  }
      if (NL == 2 && lay == 0) LAUNCH(2, 0)
      if (NL == 2 && lay == 1) LAUNCH(2, 1)
      if (NL == 3 && lay == 0) LAUNCH(3, 0)
      if (NL == 3 && lay == 1) LAUNCH(3, 1)
      if (NL == 4 && lay == 0) LAUNCH(4, 0)
      if (NL == 4 && lay == 1) LAUNCH(4, 1)
    };
    if (layout < 0) {
      for (int lay = 0; lay < 2 && layout < 0; ++lay) {
        CHECK(hipMemset(de, 0xFF, he.size() * 4));
        run_mfma(lay);
        if (check(lay == 0 ? "MFMA i8, k = 16 kb + j" : "MFMA i8, k = 8 kb + (j&7) + 32 (j>>3)")) layout = lay;
      }
      if (layout < 0) {
        std::printf("neither operand order reproduces the FIR -- stop\n");
        return 1;
      }
    } else {
      CHECK(hipMemset(de, 0xFF, he.size() * 4));
      run_mfma(layout);
      check("MFMA i8");
    }
    const int lay = layout;
    const double t_mfma = time_ms([&]() { run_mfma(lay); }, 20);
    reps = 5;
    const double t_mfma5 = time_ms([&]() { run_mfma(lay); }, 10);
    reps = 1;
    const double gs = (double)rows * kN / 1e9;
    const double c_valu = (t_valu5 - t_valu) / 4, c_mfma = (t_mfma5 - t_mfma) / 4;
    std::printf("  whole launch (HBM in + out): window walk %.3f ms (%.1f G samples/s)   MFMA i8 %.3f ms (%.1f G samples/s)\n", t_valu,
                gs / (t_valu * 1e-3), t_mfma, gs / (t_mfma * 1e-3));
    std::printf("  arithmetic alone, per pass over the rows [(t(5 passes) - t(1)) / 4]: window walk %.3f ms   MFMA i8 (%d limbs, %d MFMA per 256 outputs) %.3f ms   ratio %.2f\n",
                c_valu, NL, 2 * NL, c_mfma, c_valu / c_mfma);
    (void)hipFree(dsink);
    (void)hipFree(dx);
    (void)hipFree(de);
    (void)hipFree(dc_);
  }
  return 0;
}
