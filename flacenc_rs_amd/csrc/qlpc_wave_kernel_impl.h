// qlpc_wave_kernel_impl.h -- wave-per-subframe variant of the fused QLPC kernel for
// 4096-sample blocks (the default block size and every 44.1 kHz BASELINE config).
//
// The generic kernel (qlpc_kernel_impl.h) spreads one subframe over a whole
// workgroup and therefore stalls all of it on every serial step (Levinson, the
// cross-wave reductions).  Here ONE WAVE owns one subframe from the first LDS
// read to the parameter record: lane l holds the 64 samples [64 l, 64 l + 64)
// = four 16-sample chunks = exactly one finest Rice partition, so
//   * the autocorrelation tree is 2 in-lane levels + one 6-level lane butterfly
//     (identical balanced tree over the chunk index -> same canonical sums),
//   * Levinson + quantisation of the workgroup's four subframes run side by side on four
//     lanes of one wave (a wave instruction costs the same for 1 or 64 lanes),
//   * the residual, its bit-sliced population counts and the partition's bit table
//     never leave the lane's registers,
//   * orders 6..0 of the Rice search are a lane butterfly.
// A workgroup is 4 waves: in stereo mode the four roles L, R, M, S of one frame
// sharing the two channel images in LDS (HBM reads: each channel once); in plain
// mode four independent subframes.
//
// Template switches: STEREO (roles of a 2-channel frame vs independent subframes), DECIDE
// (encode_subframe's candidate choice, and try_stereo_coding's channel assignment when STEREO,
// on the device: only the chosen residual rows are written), FIXED (the fixed-LPC candidate of
// coding.rs:298-331 goes through the same Rice search in a rolled candidate loop).
//
// All `file:line` citations are relative to the flacenc-rs v0.5.1 tree.
#ifndef FLACENC_HIP_QLPC_WAVE_KERNEL_IMPL_H_
#define FLACENC_HIP_QLPC_WAVE_KERNEL_IMPL_H_

#include <type_traits>

#include "frame_bits.h"
#include "qlpc_kernel_impl.h"
#include "sumabs_chain.h"

namespace flacenc_hip {
namespace {

// waves per SIMD the register allocator must leave room for: 3 workgroups per CU is what the
// stereo LDS footprint (2 images + window) allows; plain mode (4 images) is LDS-bound earlier
#ifndef FLACENC_ORDER12_OCC3
#define FLACENC_ORDER12_OCC3 14  // bit v: variant v (FLACENC_STEREO) of the order-12 bucket at three workgroups per CU
#endif
#ifndef FLACENC_WAVE_OCC
// 3 where the instance fits 168 VGPRs with (next to) nothing of its common path in scratch -- checked with
// tools/kernel_resources.py and the Spill / Reload comments of tools/asm_variant.sh's output, then measured
// (same box, alternating runs, 24576 frames): the stereo kernels up to order 10, with and without the
// on-device decision and the fixed-LPC candidate.  What is still spilled there sits around the out-of-line
// call of the rare literal Rice search and in the decision tail.  Order 8: deciding kernel 0.313 -> 0.275 ms
// per 8192 frames, with the fixed-LPC candidate 0.446 -> 0.380 ms.  Order 12 (17 / 35 / 44 spilled dwords in the
// deciding / candidate / fixed-LPC variants, and an exchange area 256 bytes smaller, see kXqInWindow) gains too:
// 1.005 -> 0.815 ms, 1.12 -> 1.02 ms and 1.54 -> 1.29 ms per 24576 frames.  The fused bit writer (PACK) and the
// independent-channel kernel spill in their inner loops at 168 registers and stay at 2.
// (variants 6 / 7 are 3 / 4 with the order selector's chain walk: the same budget as their base variant)
#if defined(FLACENC_STEREO)
#define FLACENC_BASE_VARIANT (FLACENC_STEREO >= 6 ? FLACENC_STEREO - 3 : FLACENC_STEREO)
#endif
// Round 5 -- FOUR workgroups per CU (128 registers, window weights from L2 instead of a third LDS image; see
// window_image below): possible since the lane order keeps one accumulator set per lag where the chunk tree kept three.
// Same-box A/B, 24576 frames (98304: the deciding order-8 kernel 2.39 -> 2.10 ms): deciding kernels 0.614 -> 0.580
// (order 8), 0.70 -> 0.664 (10), 0.77 -> 0.74 ms (12) with 50-75 spilled dwords; four-candidate kernels +3.5 / +4 % at
// orders 8 / 10, -2 % at 12; with the fixed-LPC candidate (170-197 spilled dwords at 128 registers) +1.5 % slower at order
// 8, -0.7 % at 10, -3 % at 12.  Blocks of 4608 (two 19.8 KB images + the certificate's scratch = 43 KB) stay at three.
#if defined(FLACENC_STEREO) && defined(FLACENC_MAXP) && (!defined(FLACENC_SPL) || FLACENC_SPL == 64) && \
    ((FLACENC_BASE_VARIANT == 2) || (FLACENC_BASE_VARIANT == 1 && FLACENC_MAXP <= 10) || \
     (FLACENC_BASE_VARIANT == 3 && FLACENC_MAXP >= 10))
#define FLACENC_WAVE_OCC 4
#elif defined(FLACENC_STEREO) && defined(FLACENC_MAXP) && \
    ((FLACENC_MAXP <= 10 && (FLACENC_BASE_VARIANT == 1 || FLACENC_BASE_VARIANT == 2 || FLACENC_BASE_VARIANT == 3)) || \
     (FLACENC_MAXP == 12 && ((FLACENC_ORDER12_OCC3 >> FLACENC_BASE_VARIANT) & 1)))
#define FLACENC_WAVE_OCC 3
#else
#define FLACENC_WAVE_OCC 2
#endif
#endif
// The stereo 4096-sample instances at three workgroups per CU keep the window table as a third LDS image; at FOUR
// (round 5: 128 registers, possible since the lane order needs one accumulator set where the chunk tree needed three) the
// 40 KB a workgroup may take hold the two channel images, the exchange area and the certificate's scratch, and the weights
// come from the L2-resident table, loaded unconditionally (it holds exactly 1.0f inside the flat part) one step ahead.
constexpr bool kWindowImageBuild = FLACENC_WAVE_OCC < 4;
constexpr bool window_image(bool stereo, int spl) { return kWindowImageBuild && stereo && spl == 64; }
constexpr int kWaveN = 4096;        // block size handled by this kernel
constexpr int kSeg = 68;            // dwords per lane segment: 64 samples + 4 pad (conflict-free b128)
constexpr int kBufDwords = 65 * kSeg + 8;  // one leading all-zero segment (halo of lane 0) + look-ahead slack

__device__ __forceinline__ int widx(int t) { return ((t >> 6) + 1) * kSeg + (t & 63); }
// the same relative to a lane's segment base lb = widx(64 lane) for sample 64 lane + off, -64 <= off < 128: with a
// compile-time off this is lb + constant (folded into the LDS instruction's offset field) where widx(tl + off)
// on an opaque tl costs a shift, a v_mul_lo_u32, an and and two adds per access
__device__ __forceinline__ int widx_rel(int lb, int off) {
  return lb + off + (off >= 64 ? (kSeg - 64) : 0) - (off < 0 ? (kSeg - 64) : 0);
}

// Geometry of the wave kernel for SPL samples per lane: 64 (blocks of 4096) or 72 (blocks of 4608 = 64 finest Rice
// partitions of 72 samples, rice.rs:157-165 -- the CD-style block sizes 4608 / 2304 / 1152 / 576 all have
// 72-sample partitions; the full wave's worth, 4608, gets this kernel).  A lane's segment in an LDS image is its
// SPL samples + 4 dwords of padding (76 dwords for 72: consecutive lanes start 12 banks apart, 16 lanes cover the
// 64 banks with their 4-dword reads, as 68 does).
template <int SPL>
struct WaveGeom {
  static constexpr int N = 64 * SPL;
  static constexpr int Seg = SPL + 4;
  static constexpr int Buf = 65 * Seg + 8;  // one leading all-zero segment (halo of lane 0) + look-ahead slack
  static constexpr int Quads = N / 4;       // 16-byte pieces per row
  static constexpr int Chunks = N / 16;     // canonical 16-sample chunks per block (256 / 288)
  // LDS index of sample t, 0 <= t < N + SPL
  static __device__ __forceinline__ int idx(int t) {
    if (SPL == 64) return ((t >> 6) + 1) * Seg + (t & 63);
    const int sgm = (int)(((uint32_t)t * 58255u) >> 22);  // t / 72 (exact for t < 73727)
    return (sgm + 1) * Seg + (t - sgm * SPL);
  }
  // ... of the 16-byte piece q (sample 4 q), 0 <= q < Quads
  static __device__ __forceinline__ int qidx(int q) {
    if (SPL == 64) return ((q >> 4) + 1) * Seg + ((q & 15) << 2);
    const int sgm = (int)(((uint32_t)q * 3641u) >> 16);  // q / 18 (exact for q < 1300)
    return (sgm + 1) * Seg + ((q - sgm * (SPL / 4)) << 2);
  }
  // ... of sample SPL lane + off relative to the lane's segment base lb = (lane + 1) Seg, -SPL <= off < 2 SPL
  static __device__ __forceinline__ int rel(int lb, int off) {
    return lb + off + (off >= SPL ? (Seg - SPL) : 0) - (off < 0 ? (Seg - SPL) : 0);
  }
};

// The fixed-LPC order selector's rare path under the reference's summation orders (QlpcKernelArgs::sumabs_mode):
// an estimator partition whose exact sum of |e| reaches 2^24 is walked serially by its first lane, from the
// LDS images, with the f32 chains of find_sum_abs_f32 (sumabs_chain.h).  begin == end: nothing to do for this lane.
template <int SPL, bool NIGHTLY>
__device__ __forceinline__ void sumabs_chains_walk(const int32_t* bufA, const int32_t* bufB, int kind, int begin, int end,
                                                   float* out5) {
  typedef WaveGeom<SPL> G;
  auto sample = [&](int t) -> uint32_t {
    const int ix = t < 0 ? SPL + t : G::idx(t);  // (the segment in front of the block holds zeros)
    const int x = bufA[ix];
    if (kind < 2) return (uint32_t)x;
    const int y = bufB[ix];
    return (uint32_t)(kind == 2 ? (x + y) >> 1 : x - y);
  };
  SumAbsChains<NIGHTLY> ch;
  ch.init(begin, end);
  const int w0 = begin & ~15;
  ch.seed(sample(w0 - 4), sample(w0 - 3), sample(w0 - 2), sample(w0 - 1));
#pragma unroll 1
  for (int t0 = w0; t0 < end; t0 += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = t0 + j;
      ch.template step<true>(t < end ? sample(t) : 0u, t, j);
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) out5[k] = ch.result(k);
}
template <int SPL>
__device__ __attribute__((noinline)) void sumabs_chains_from_lds(const int32_t* bufA, const int32_t* bufB, int kind, int begin,
                                                                 int end, int nightly, float* out5) {
  if (nightly) sumabs_chains_walk<SPL, true>(bufA, bufB, kind, begin, end, out5);
  else sumabs_chains_walk<SPL, false>(bufA, bufB, kind, begin, end, out5);
}

// |v| with i32::MIN -> 2^31 - 1 + 1 handled by the caller's unsigned compare; inputs are <= 25 bits
__device__ __forceinline__ int abs_sat(int v) { return v < 0 ? -v : v; }

// ---- cross-lane helpers (DPP: no LDS traffic) --------------------------------
#define FLACENC_DPP(v, ctrl, rowmask) \
  ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rowmask), 0xF, false))

// value of lane + s for s in {1, 2, 4, 8} (row_shl), 16 (swizzle xor 16), 32 (readlane);
// only meaningful on lanes that are multiples of 2 s -- the group leaders that use it
template <int S>
__device__ __forceinline__ uint32_t from_upper_half(uint32_t v) {
  if (S == 1) return FLACENC_DPP(v, 0x101, 0xF);
  if (S == 2) return FLACENC_DPP(v, 0x102, 0xF);
  if (S == 4) return FLACENC_DPP(v, 0x104, 0xF);
  if (S == 8) return FLACENC_DPP(v, 0x108, 0xF);
  if (S == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);  // lane ^ 16
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}

// wave-wide sum (< 2^32), result uniform: row scan (row_shr 1,2,4,8), row_bcast15, row_bcast31
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
  v += FLACENC_DPP(v, 0x111, 0xF);
  v += FLACENC_DPP(v, 0x112, 0xF);
  v += FLACENC_DPP(v, 0x114, 0xF);
  v += FLACENC_DPP(v, 0x118, 0xF);
  v += FLACENC_DPP(v, 0x142, 0xA);
  v += FLACENC_DPP(v, 0x143, 0xC);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// exclusive prefix sum over the 64 lanes (same DPP row scan as wave_sum_dpp)
__device__ __forceinline__ uint32_t wave_excl_scan_dpp(uint32_t v) {
  uint32_t s = v;
  s += FLACENC_DPP(s, 0x111, 0xF);
  s += FLACENC_DPP(s, 0x112, 0xF);
  s += FLACENC_DPP(s, 0x114, 0xF);
  s += FLACENC_DPP(s, 0x118, 0xF);
  s += FLACENC_DPP(s, 0x142, 0xA);
  s += FLACENC_DPP(s, 0x143, 0xC);
  return s - v;
}

// The same canonical sums for NV values per lane at once, through a per-wave LDS area of NV x kTreeRow doubles
// instead of NV dependent DPP trees (35 VALU instructions each): every lane parks its NV values, lane
// 4 j + s fetches value j of lanes [16 s, 16 s + 16) and adds them exactly as the tree pairs them --
// ((v0 + v1) + (v2 + v3)) + ... -- and two row shifts fold the four sixteenths, (s0 + s1) + (s2 + s3).
// The total of value j arrives in lane 4 j + 3 (other lanes, and lanes >= 4 NV, return garbage).
// Layout: value j of lane l at j kTreeRow + (l >> 4) kTreeSeg + (l & 15) -- sixteenths 18 doubles apart, values 72:
// the 16-byte reads of lanes (j, s) then start 36 s + 144 j dwords apart, sixteen distinct bank quads for any sixteen
// consecutive lanes (dense rows of 64 put every other lane on the same banks: SQ_LDS_BANK_CONFLICT was 55 % of the
// autocorrelation kernel's LDS cycles).
constexpr int kTreeSeg = 18, kTreeRow = 4 * kTreeSeg;
template <int NV>
__device__ __forceinline__ double wave_tree_sums_lds_n(const double* v, double* buf, int lane) {
  static_assert(4 * NV <= 64, "one quad of lanes per value");
  const int mypos = (lane >> 4) * kTreeSeg + (lane & 15);
#pragma unroll
  for (int j = 0; j < NV; ++j) buf[j * kTreeRow + mypos] = v[j];
  __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order)
  const int j = (lane >> 2) < NV ? (lane >> 2) : NV - 1;
  const double2* src = reinterpret_cast<const double2*>(buf + j * kTreeRow + (lane & 3) * kTreeSeg);
  double x[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double2 q = src[i];
    x[2 * i] = q.x;
    x[2 * i + 1] = q.y;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
#pragma unroll
    for (int i = 0; i < 16; i += 2 * d) x[i] = x[i] + x[i + d];
  }
  double t = x[0];
#define FLACENC_F64_DPP_STEP(CTRL, ROWMASK)                                                              \
  {                                                                                                      \
    const unsigned long long b_ = (unsigned long long)__double_as_longlong(t);                          \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b_, CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF);        \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b_ >> 32), CTRL, ROWMASK, 0xF, (ROWMASK) == 0xF); \
    t = t + __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                    \
  }
  FLACENC_F64_DPP_STEP(0x111, 0xF)
  FLACENC_F64_DPP_STEP(0x112, 0xF)
#undef FLACENC_F64_DPP_STEP
  return t;
}

template <int NV>
__device__ __forceinline__ double wave_tree_sums_lds(const double (&v)[NV], double* buf, int lane) {
  return wave_tree_sums_lds_n<NV>(v, buf, lane);
}

__device__ __forceinline__ uint32_t wave_or_dpp(uint32_t v) {
  v |= FLACENC_DPP(v, 0x111, 0xF);
  v |= FLACENC_DPP(v, 0x112, 0xF);
  v |= FLACENC_DPP(v, 0x114, 0xF);
  v |= FLACENC_DPP(v, 0x118, 0xF);
  v |= FLACENC_DPP(v, 0x142, 0xA);
  v |= FLACENC_DPP(v, 0x143, 0xC);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {
  uint32_t o;
  o = FLACENC_DPP(v, 0x111, 0xF); v = o > v ? o : v;
  o = FLACENC_DPP(v, 0x112, 0xF); v = o > v ? o : v;
  o = FLACENC_DPP(v, 0x114, 0xF); v = o > v ? o : v;
  o = FLACENC_DPP(v, 0x118, 0xF); v = o > v ? o : v;
  o = FLACENC_DPP(v, 0x142, 0xA); v = o > v ? o : v;
  o = FLACENC_DPP(v, 0x143, 0xC); v = o > v ? o : v;
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// acc + c * s with 24-bit operands as ONE v_mad_i32_i24 (left to itself the compiler
// re-associates the prediction sum into multiplies + add3 trees, ~40 % more instructions)
__device__ __forceinline__ int32_t mad24(int32_t c, int32_t s, int32_t acc) {
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "s"(c), "v"(s), "v"(acc));
  return r;
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x111, 0xF, 0xF, false); v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x112, 0xF, 0xF, false); v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x114, 0xF, 0xF, false); v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x118, 0xF, 0xF, false); v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x142, 0xA, 0xF, false); v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x143, 0xC, 0xF, false); v = o < v ? o : v;
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// carry-save adder on bit-planes: (h, l) = a + b + c per bit position, two v_bitop3_b32
// (truth tables 0x96 = a ^ b ^ c, 0xE8 = majority)
#define FLACENC_CSA(h, l, a_, b_, c_)                                          \
  {                                                                            \
    const uint32_t x_ = (a_), y_ = (b_), z_ = (c_);                            \
    h = __builtin_amdgcn_bitop3_b32(x_, y_, z_, 0xE8);                         \
    l = __builtin_amdgcn_bitop3_b32(x_, y_, z_, 0x96);                         \
  }

// 16 words -> bit-planes {1, 2, 4, 8, 16} of the per-bit population counts (Harley-Seal)
// The counted words are NOT the zig-zag codes u = 2 m + neg (rice::encode_signbit, rice.rs:169)
// but w = m | neg << 31 with m = e ^ (e >> 31): one shift + one bitop per sample instead of
// three ops, and u >> p == m >> (p - 1) for p >= 1, sum u == 2 sum m + #neg (see plane_sum).
__device__ __forceinline__ void popcount_planes16(const int32_t* e, uint32_t (&pl)[5]) {
  uint32_t u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // (x ^ t) | (t & 2^31): one v_bitop3_b32 (table 0xBC over (x, t, 2^31)) behind the shift
    const uint32_t x = (uint32_t)e[k], t = (uint32_t)(e[k] >> 31);
    u[k] = __builtin_amdgcn_bitop3_b32(x, t, 0x80000000u, 0xBC);
  }
  uint32_t ones = 0, twos = 0, fours = 0, eights = 0, sixteens;
  uint32_t twosA, twosB, foursA, foursB, eightsA, eightsB;
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
  pl[0] = ones;
  pl[1] = twos;
  pl[2] = fours;
  pl[3] = eights;
  pl[4] = sixteens;
}

// bit-sliced add: a (NA planes) += b (NA planes) -> NA + 1 planes
template <int NA>
__device__ __forceinline__ void planes_add(uint32_t* a, const uint32_t* b) {
  uint32_t carry = a[0] & b[0];
  a[0] ^= b[0];
#pragma unroll
  for (int k = 1; k < NA; ++k) {
    const uint32_t x_ = a[k], y_ = b[k];
    a[k] = __builtin_amdgcn_bitop3_b32(x_, y_, carry, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(x_, y_, carry, 0xE8);
  }
  a[NA] = carry;
}

// what fixed_lpc (coding.rs:298-331) settled on for one subframe
struct FixedChoice {
  bool have;
  int order;                     // uniform
  unsigned long long key;        // the selector's key of `order` (estimate or real bits)
  int bestk;                     // Rice result of the order's error signal
  uint32_t my_p;
  unsigned long long code_bits, sum_q, sub_bits;
};

struct RiceResult {
  int bestk;                     // chosen order = 6 - bestk (uniform)
  unsigned long long best_bits;  // PrcParameter::code_bits (uniform)
  uint32_t my_p;                 // parameter of the chosen-order partition this lane leads
  bool saturated;                // the chosen order has a saturated table minimum
  uint32_t sat_levels;           // bit k: some group minimum at level k hit MAX_P_TO_BITS
};

// Orders 6..0 of PrcParameterFinder::find (rice.rs:246-298) for a 4096 block with one finest
// partition per lane.  Table entries are kept as  Wp[j] = table[p_lo + j] - 4  so that
//   merge   (rice.rs:144-152): min(a + b - 4, MAX) on tables == min(Wa + Wb, MAX - 4)
//   minimiser (rice.rs:115-141): min over p of (Wp << 5 | p), bits = (min >> 5) + 4, p = min & 31.
// Level k's tables are valid on the lanes that are multiples of 2^k (group leaders), which
// fetch their partner's entries from lane + 2^(k-1).
//
// Only the parameters p_lo .. p_lo + NP - 1 are evaluated; [p_lo, max_p] is a window that
// provably contains every minimiser of every partition at every order (see rice_window), so
// the result is identical to searching 0..=max_p.  Entries above max_p are set to the
// saturation value once: they can then never beat (or tie ahead of) a legal parameter.
// EXACT = false: literal chunk-clamped sums of rice.rs:75-98 for residuals >= 2^26.
// sum over the lane's 64 samples of (u_i >> p), u = zig-zag code, from the bit-planes of the
// w words (magnitude planes q[k] = pl[k] & 0x7FFFFFFF, sign counts in bit 31):
//   p >= 1: u >> p = m >> (p - 1);   p == 0: sum u = 2 sum m + #negative
struct PlaneSums {
  uint32_t q[7];     // magnitude planes
  uint32_t sum_m;    // sum of m over the lane's samples
  uint32_t negs;     // number of negative samples
};
__device__ __forceinline__ PlaneSums make_plane_sums(const uint32_t (&pl)[7]) {
  PlaneSums ps;
  ps.sum_m = 0;
  ps.negs = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    ps.q[k] = pl[k] & 0x7FFFFFFFu;
    ps.sum_m += ps.q[k] << k;
    ps.negs += (pl[k] >> 31) << k;
  }
  return ps;
}
__device__ __forceinline__ uint32_t plane_sum_ge1(const PlaneSums& ps, uint32_t p) {  // p >= 1
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) sum += (ps.q[k] >> ((p - 1u) & 31u)) << k;
  return sum;
}
__device__ __forceinline__ unsigned long long plane_sum_any64(const PlaneSums& ps, uint32_t p) {
  if (p == 0) return 2ull * ps.sum_m + ps.negs;
  unsigned long long sum = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) sum += (unsigned long long)(ps.q[k] >> ((p - 1u) & 31u)) << k;
  return sum;
}

// The parameter axis is walked in groups of 8 (a rolled, wave-uniform loop): different parameters never
// interact -- a merge adds table entries of the same p, a minimiser takes the minimum over p of
// (bits << 5 | p) -- so each group runs all seven levels on its own 8 entries and leaves its packed
// minimum per level in pk[level]; the minimum over the groups is the minimum over the window.  One
// instance of the level code instead of one per window width keeps the search at 8 + 8 + 7 registers
// whatever the span (typical material: one group).
// Table entries Wp[j] = table[p_base + j] - 4 of this lane's 64-sample partition (see rice_search).
// EXACT: from the bit-planes; otherwise the literal chunk-clamped sums of rice.rs:75-98 from e[].
// NOSAT (only with EXACT; chosen per subframe, see rice_nosat_ok): no entry of any level can reach the
// saturation value and every parameter of the window is legal or provably losing, so the clamps and the
// validity selects are dropped and the entries are kept pre-shifted (W << 5) for the packed minimiser.
template <bool EXACT, bool NOSAT = false, int SPL = 64, int NP>
__device__ __forceinline__ void rice_build_tables(const PlaneSums& ps, const int32_t* e, uint32_t len0,
                                                  uint32_t p_base, uint32_t max_p, int lane, int warm,
                                                  uint32_t (&Wp)[NP]) {
  constexpr uint32_t kWMax = kMaxPToBits - 4u;
  if (EXACT && NOSAT) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const uint32_t pp = p_base + (uint32_t)j;  // wave-uniform, <= 31
      uint32_t sum;
      if (j == 0) sum = (pp == 0) ? 2u * ps.sum_m + ps.negs : plane_sum_ge1(ps, pp);
      else sum = plane_sum_ge1(ps, pp);
      Wp[j] = (sum + len0 * (pp + 1u)) << 5;
    }
  } else if (EXACT) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const uint32_t pp = p_base + (uint32_t)j;  // wave-uniform
      uint32_t sum;  // sum_i (u_i >> pp); pp == 0 is only possible for j == 0
      if (j == 0) sum = (pp == 0) ? 2u * ps.sum_m + ps.negs : plane_sum_ge1(ps, pp);
      else sum = plane_sum_ge1(ps, pp);
      sum = sum < kMaxPToBits ? sum : kMaxPToBits;
      uint32_t v = sum + len0 * (pp + 1u);  // rice.rs:69-71, 95-98 (minus the 4)
      v = v < kWMax ? v : kWMax;
      Wp[j] = (pp <= max_p) ? v : kWMax;
    }
  } else {
    // the reference's slice of partition 0 starts at `warm`; its clamp cadence follows.
    // (rare path: runtime loop over p, registers selected by compare chains -- no scratch)
    const int off = (lane == 0) ? warm : 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) Wp[q] = kWMax;
#pragma unroll 1
    for (int j = 0; j < NP; ++j) {
      const uint32_t pp = p_base + (uint32_t)j;
      uint32_t accb = 0;
#pragma unroll
      for (int k = 0; k < SPL; ++k) {
        if (k >= off) {
          accb += zigzag(e[k]) >> (pp & 31u);
          if (((k - off) & 15) == 15) accb = accb < kMaxPToBits ? accb : kMaxPToBits;
        }
      }
      accb = accb < kMaxPToBits ? accb : kMaxPToBits;
      uint32_t v = accb + len0 * (pp + 1u);
      v = v < kWMax ? v : kWMax;
      if (pp > max_p) v = kWMax;
#pragma unroll
      for (int q = 0; q < NP; ++q) Wp[q] = (q == j) ? v : Wp[q];
    }
  }
}

// Levels 0..6 of one group of 8 parameters: merges in place (afterwards Wp holds, on lane 0, the table of
// all 64 partitions merged) and lowers pk[level] to the group's packed minimum (bits << 5 | p).
template <bool NOSAT = false, int NP = 8>
__device__ __forceinline__ void rice_group_levels(uint32_t (&Wp)[NP], uint32_t (&pk)[7], uint32_t p_base,
                                                  bool finest_only) {
  constexpr uint32_t kWMax = kMaxPToBits - 4u;  // (the packed minimiser takes the entries in pairs, an odd last one alone)
#define FLACENC_RICE_LEVEL(K, S)                                                              \
  {                                                                                           \
    if (K > 0) {                                                                              \
      uint32_t part[NP];                                                                      \
      _Pragma("unroll") for (int j = 0; j < NP; ++j) part[j] = from_upper_half<S>(Wp[j]);     \
      _Pragma("unroll") for (int j = 0; j < NP; ++j) {                                        \
        uint32_t v = Wp[j] + part[j];                                                         \
        Wp[j] = NOSAT ? v : (v < kWMax ? v : kWMax);                                          \
      }                                                                                       \
    }                                                                                         \
    uint32_t packed = pk[K];                                                                  \
    _Pragma("unroll") for (int j = 0; j + 1 < NP; j += 2) {                                   \
      const uint32_t c0 = (NOSAT ? Wp[j] : (Wp[j] << 5)) | (p_base + (uint32_t)j);            \
      const uint32_t c1 = (NOSAT ? Wp[j + 1] : (Wp[j + 1] << 5)) | (p_base + (uint32_t)j + 1u); \
      packed = umin3(packed, c0, c1);                                                         \
    }                                                                                         \
    if (NP & 1) {                                                                             \
      const uint32_t cl = (NOSAT ? Wp[NP - 1] : (Wp[NP - 1] << 5)) | (p_base + (uint32_t)(NP - 1)); \
      packed = cl < packed ? cl : packed;                                                     \
    }                                                                                         \
    pk[K] = packed;                                                                           \
  }
  FLACENC_RICE_LEVEL(0, 1)
  if (!finest_only) {  // (FLACENC_HIP_FLAG_FINEST_RICE_ORDER keeps the finest order)
    FLACENC_RICE_LEVEL(1, 1)
    FLACENC_RICE_LEVEL(2, 2)
    FLACENC_RICE_LEVEL(3, 4)
    FLACENC_RICE_LEVEL(4, 8)
    FLACENC_RICE_LEVEL(5, 16)
    FLACENC_RICE_LEVEL(6, 32)
  }
#undef FLACENC_RICE_LEVEL
}

// [p_lo, p_hi] is the parameter window (rice_window below); groups of 4: typical material -- partitions whose
// means lie within a factor of two of each other -- needs exactly one, and a wider window just takes more turns
// of the rolled loop (a group of 8 evaluated twice the entries for the common case).
template <bool EXACT, bool NOSAT = false, int SPL = 64>
__device__ __forceinline__ RiceResult rice_search(const PlaneSums& ps, const int32_t* e, uint32_t len0,
                                                  uint32_t p_lo, uint32_t p_hi, uint32_t max_p, bool small_bits,
                                                  int lane, int warm, bool finest_only) {
  // (groups of 5 since round 5: the window is p0min - 2 .. p0max + 1, four wide when all partition means share a binade and
  // five when they straddle one -- two thirds of the bench signal's subframes, which took a second turn of the loop for it)
#ifndef FLACENC_RICE_NP
#define FLACENC_RICE_NP 5
#endif
  constexpr int NP = EXACT ? FLACENC_RICE_NP : 8;
  constexpr uint32_t kWMax = kMaxPToBits - 4u;
  (void)kWMax;
  uint32_t pk[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) pk[k] = 0xFFFFFFFFu;
#pragma unroll 1
  for (uint32_t p_base = p_lo; p_base <= p_hi; p_base += (uint32_t)NP) {
    uint32_t Wp[NP];
    rice_build_tables<EXACT, NOSAT, SPL>(ps, e, len0, p_base, max_p, lane, warm, Wp);
    rice_group_levels<NOSAT>(Wp, pk, p_base, finest_only);
  }

  RiceResult r;
  r.bestk = 0;
  r.best_bits = 0;
  r.my_p = 0;
  r.saturated = false;
  uint32_t sat_any = 0;
  if (NOSAT && !finest_only) {
    // Level totals by ONE triangular reduction instead of seven wave sums (round 5): level K's minima live on the lanes
    // that are multiples of 2^K, so their sum needs only the tree levels from K on -- at spacing S the leaders of 2 S
    // add their partner's running totals of every level below log2(2 S): 1 + 2 + 3 + 4 + 5 + 6 = 21 adds (fifteen of
    // them with the fetch folded in as a DPP operand or done by the LDS crossbar) where seven full sums took 42 and the
    // leader masks 21 more.  NOSAT: no minimum can have saturated, totals fit 32 bits.  Lane 0 ends up with all seven.
    uint32_t t[7];
#pragma unroll
    for (int K = 0; K < 7; ++K) t[K] = (pk[K] >> 5) + 4u;
    t[0] += from_upper_half<1>(t[0]);
#pragma unroll
    for (int K = 0; K < 2; ++K) t[K] += from_upper_half<2>(t[K]);
#pragma unroll
    for (int K = 0; K < 3; ++K) t[K] += from_upper_half<4>(t[K]);
#pragma unroll
    for (int K = 0; K < 4; ++K) t[K] += from_upper_half<8>(t[K]);
#pragma unroll
    for (int K = 0; K < 5; ++K) t[K] += from_upper_half<16>(t[K]);
#pragma unroll
    for (int K = 0; K < 6; ++K) t[K] += from_upper_half<32>(t[K]);
    // strict < keeps the finer order on ties (rice.rs:285); everything below is wave-uniform
    uint32_t best = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0]);
    int bk = 0;
#pragma unroll
    for (int K = 1; K < 7; ++K) {
      const uint32_t tot = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[K]);
      if (tot < best) {
        best = tot;
        bk = K;
      }
    }
    r.best_bits = best;
    r.bestk = bk;
    uint32_t mp = pk[0];
#pragma unroll
    for (int K = 1; K < 7; ++K) mp = (bk == K) ? pk[K] : mp;
    r.my_p = mp & 31u;
    r.sat_levels = 0;
    return r;
  }
  // level totals: the group leaders' minima summed over the wave; strict < keeps the finer order on
  // ties (rice.rs:285)
#pragma unroll
  for (int K = 0; K < 7; ++K) {
    if (K > 0 && finest_only) break;
    const uint32_t bits = (pk[K] >> 5) + 4u;
    const bool lead = (lane & ((1 << K) - 1)) == 0;
    const uint32_t lb = lead ? bits : 0u;
    sat_any |= (lead && bits >= kMaxPToBits) ? (1u << K) : 0u;
    unsigned long long tot;
    if (small_bits) tot = wave_sum_dpp(lb);
    else tot = ((unsigned long long)wave_sum_dpp(lb >> 16) << 16) + wave_sum_dpp(lb & 0xFFFFu);
    if (K == 0 || tot < r.best_bits) {
      r.best_bits = tot;
      r.bestk = K;
      r.my_p = pk[K] & 31u;
    }
  }
  sat_any = wave_or_dpp(sat_any);
  r.saturated = (sat_any >> r.bestk) & 1u;
  r.sat_levels = sat_any;
  return r;
}

// The literal chunk-clamped search (residuals >= 2^26: full-scale 24/25-bit material with a useless
// predictor) as a real function call: it is rare, and inlined it made the register allocator park
// values of the common path in scratch across the branch.  Out of line, the call site -- inside the rare
// branch -- is where live registers are saved, and the residuals are handed over through private memory.
struct RiceLiteralResult {
  RiceResult rr;
  unsigned long long sum_q;
};
template <int SPL = 64>
__device__ __attribute__((noinline)) void rice_search_literal(const int32_t* e, uint32_t len0, uint32_t max_p,
                                                              int small_bits, int lane, int warm, int finest_only,
                                                              RiceLiteralResult* out) {
  int32_t ev[SPL];
#pragma unroll
  for (int k = 0; k < SPL; ++k) ev[k] = e[k];
  PlaneSums none;
#pragma unroll
  for (int k = 0; k < 7; ++k) none.q[k] = 0;
  none.sum_m = none.negs = 0;
  RiceResult rr = rice_search<false, false, SPL>(none, ev, len0, 0u, max_p, max_p, small_bits != 0, lane, warm, finest_only != 0);
  // the table sums of this path are the reference's wrapping u32 adds (rice.rs:88-93): code_bits
  // does not determine the true quotient sum any more, saturated or not -- always count it
  rr.saturated = true;
  const uint32_t gp = (uint32_t)__shfl((int)rr.my_p, lane & ~((1 << rr.bestk) - 1), 64);
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < SPL; ++k) {
    uint32_t qv = zigzag(ev[k]) >> gp;  // warm-up slots hold 0
    lo += qv & 0xFFFFu;
    hi += qv >> 16;
  }
  // per lane 64 x 16 bits = 22 bits; x 64 lanes = 28 bits
  out->sum_q = ((unsigned long long)wave_sum_dpp(hi) << 16) + (unsigned long long)wave_sum_dpp(lo);
  out->rr = rr;
}

// ---- the fused kernel's summation order (round 5) -------------------------------------------------------------------
// Lane l sums its 64 samples as one fma chain per lag; the 64 lane sums v[l] of a lag then meet as
//     w[i] = (v[i] + v[i + 32]) + (v[i + 16] + v[i + 48]),  i = 0 .. 15,
//     R = (((w15 + w14) + (w13 + w12)) + ((w11 + w10) + (w9 + w8))) + (((w7 + w6) + (w5 + w4)) + ((w3 + w2) + (w1 + w0)))
// (oracle: orc_auto_correlation_lane_order_f64).  The first two levels are reduce-scatters: v_permlane32_swap(A, B) leaves
// the lower half with (own A, the upper half's A) and the upper half with (the lower half's B, own B), so ONE add per
// PAIR of lags halves both -- three instructions per pair and level where a DPP tree spends six -- and v_permlane16_swap
// does the same between the rows of each half.  What is left, ceil(N / 4) values per lane, takes the four row shifts.
// N = 9 lags: 62 instructions for what nine DPP trees of six levels did in 162 (+ 27 adds of the in-lane tree that the
// single chain does not need).  An odd value out is paired with a register nobody defines; its "sum" is never read.
template <int N>
constexpr int kLaneOrderSlots = ((N + 1) / 2 + 1) / 2;

__device__ __forceinline__ double f64_from_halves(uint32_t lo, uint32_t hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int N>
__device__ __forceinline__ void lane_order_reduce(const double (&v)[N], double (&out)[kLaneOrderSlots<N>]) {
  constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2;
  double s1[N1];
#pragma unroll
  for (int p = 0; p < N1; ++p) {
    const unsigned long long ab = (unsigned long long)__double_as_longlong(v[2 * p]);
    unsigned long long bb;
    if (2 * p + 1 < N) bb = (unsigned long long)__double_as_longlong(v[2 * p + 1]);
    else asm volatile("" : "=v"(bb));  // (no partner: an undefined register)
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)ab, (uint32_t)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ab >> 32), (uint32_t)(bb >> 32), false, false);
    s1[p] = f64_from_halves(lo[0], hi[0]) + f64_from_halves(lo[1], hi[1]);
  }
#pragma unroll
  for (int q = 0; q < N2; ++q) {
    const unsigned long long ab = (unsigned long long)__double_as_longlong(s1[2 * q]);
    unsigned long long bb;
    if (2 * q + 1 < N1) bb = (unsigned long long)__double_as_longlong(s1[2 * q + 1]);
    else asm volatile("" : "=v"(bb));
    const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ab, (uint32_t)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ab >> 32), (uint32_t)(bb >> 32), false, false);
    double t = f64_from_halves(lo[0], hi[0]) + f64_from_halves(lo[1], hi[1]);
#define FLACENC_F64_ROW_STEP(CTRL)                                                                       \
    {                                                                                                    \
      const unsigned long long b_ = (unsigned long long)__double_as_longlong(t);                        \
      const uint32_t lo_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b_, CTRL, 0xF, 0xF, true);          \
      const uint32_t hi_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b_ >> 32), CTRL, 0xF, 0xF, true);  \
      t = t + f64_from_halves(lo_, hi_);                                                                 \
    }
    FLACENC_F64_ROW_STEP(0x111)
    FLACENC_F64_ROW_STEP(0x112)
    FLACENC_F64_ROW_STEP(0x114)
    FLACENC_F64_ROW_STEP(0x118)
#undef FLACENC_F64_ROW_STEP
    out[q] = t;
  }
}

// ---- the certificate's fallback: the reference's own chains, in the kernel -------------------------------------------
// A subframe whose chunk-tree sums do not certify its quantised parameters (levinson_quantize<.., CERT>) is redone from
// weighted_auto_correlation_nosimd's sums (lpc.rs:533-548): one sequential fma chain per lag over t = P .. n - 1.  Wave 0
// runs them for the workgroup's four subframes at once on v_mfma_f64_4x4x4_4b_f64 -- a block per subframe, sixteen lags
// tau = i + 4 j per block, A_b[i][k] = x_w[4 m + k - i - 12], B_b[k][j] = x_w[4 m + k - 12 + 4 j] (0 below t = P and
// from t = n on): acorr_reference_mfma_kernel's scheme (acorr_reference.cpp; the operand layout and the equality of the
// chained instruction with the sequential chain are probed in tools/microbench/mfma_f64_4x4x4_probe.hip), fed from the
// workgroup's LDS images instead of HBM.  `rows`: 4 x kCertRow values of LDS scratch ([history | a tile] per subframe);
// the sums come back in the same area, lag tau of subframe b at ((double*)rows)[16 b + tau].
// Out of line: it runs for a fraction of a per cent of the frames of noisy material (all of them on near-pure tones)
// and must not cost the common path a register.
// Round 6: the rows are staged as f64 -- a sample is converted ONCE, by the lane that lands it, where until then every
// step converted its two operands on the chain's own pipe (the f64 "matrix" instruction occupies the vector ALU, nothing
// overlaps it: 2 x 8 + 16 cycles a step for a lone wave; in-kernel stamps on the reference's real-audio fixtures had
// a recomputed frame's phase 2 at 140 k cycles against 15 k, the chains 100 k of it where 1088 bare steps need 18 k).
// What that bought is 3 % on music and nothing on near-pure tones, and an ablation says why (profiles/r06_fallback_ablation.txt):
// 0.70 of the fallback's 0.94 ms per 24576 recomputed frames are the 1032 DEPENDENT v_mfma_f64_4x4x4 themselves, ~65
// cycles each -- one chain per SIMD fills a quarter of a pipe that acorr_reference_mfma_kernel fills with 28 chains per
// CU; staging is 0.16, the history copy 0.05.  The fallback is latency-bound by construction; hard material is the order
// mode's business (flacenc_hip_api.cpp, launch_adaptive), not this function's.
// Tiles of 96 samples + 32 of history per subframe = 4.25 KB: what four workgroups per CU leave next to the two images.
// The plain 4608-sample instances (four 19.8 KB images, 1.7 KB left) keep f32 rows and tiles of 32.
constexpr bool cert_f64(bool stereo, int spl) { return stereo || spl == 64; }
constexpr int cert_hist(bool stereo, int spl) { return cert_f64(stereo, spl) ? 32 : 64; }
constexpr int cert_tile(bool stereo, int spl) { return cert_f64(stereo, spl) ? 96 : 32; }
constexpr int cert_scratch_bytes(bool stereo, int spl) {
  return 4 * (cert_hist(stereo, spl) + cert_tile(stereo, spl) + 8) * (cert_f64(stereo, spl) ? 8 : 4);
}

template <int SPL, bool STEREO>
__device__ __attribute__((noinline)) void reference_chains_from_lds(uint32_t rows_off, const float* wtab, int P) {
  // (the images and the scratch rows by their place in the workgroup's LDS, not by pointer: behind a generic pointer
  // every access of this out-of-line function paid a 64-bit address and an address-space cast -- 140 instructions per 16 steps)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int32_t* const sm = reinterpret_cast<const int32_t*>(smem_raw);
  constexpr bool F64 = cert_f64(STEREO, SPL);
  constexpr int kCertTile = cert_tile(STEREO, SPL), kCertHist = cert_hist(STEREO, SPL);
  using row_t = std::conditional_t<F64, double, float>;
  row_t* const rows = reinterpret_cast<row_t*>(smem_raw + rows_off);
  using G = WaveGeom<SPL>;
  constexpr int n = G::N;
  constexpr int kCertRow = kCertHist + kCertTile + 8;
  static_assert(kCertHist >= 28 && (kCertTile % 32) == 0 && (4 * kCertHist) % 64 == 0, "operands reach 27 samples back; whole groups of 8 steps");
  const int lane = threadIdx.x & 63;
  // Staging: TWO samples per lane on kCertTile / 2 lanes (round 6; four per lane kept a quarter of the wave busy with twice
  // the instructions: a lone wave pays ~6 cycles for each whatever its exec mask)
  int2 raw[STEREO ? 2 : 4];
  float2 wv;
  const bool stager = 2 * lane < kCertTile;
  auto issue = [&](int T0) __attribute__((always_inline)) {
    const int t = T0 + 2 * lane;
    if (!stager) return;
    const bool in = t < n;  // (n is even: a pair lies inside the block or behind it)
    const int ix = G::idx(in ? t : 0);
#pragma unroll
    for (int r = 0; r < (STEREO ? 2 : 4); ++r) {
      const int2 v = *reinterpret_cast<const int2*>(&sm[r * G::Buf + ix]);
      raw[r] = in ? v : make_int2(0, 0);
    }
    wv = make_float2(1.0f, 1.0f);
    if (wtab != nullptr && in) {
      // A GLOBAL load, said so: through the generic pointer of this out-of-line function it is a flat_load, which counts
      // as an LDS operation too (the waits for the operand reads then include the memory latency of the next tile's weights)
      typedef float v2f_t __attribute__((ext_vector_type(2)));
      typedef const v2f_t __attribute__((address_space(1))) gv2f_t;
      const v2f_t w2 = *(gv2f_t*)(wtab + t);
      wv = make_float2(w2.x, w2.y);
    }
  };
  // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754); widened here, once per sample (lpc.rs:545)
  auto land = [&]() __attribute__((always_inline)) {
    auto put = [&](int b, const int2& sv) {
      const float x0 = (float)sv.x * wv.x, x1 = (float)sv.y * wv.y;
      if (stager) {
        row_t* const dst = &rows[b * kCertRow + kCertHist + 2 * lane];
        if (F64) *reinterpret_cast<double2*>(dst) = make_double2((double)x0, (double)x1);
        else *reinterpret_cast<float2*>(dst) = make_float2(x0, x1);
      }
    };
    if (STEREO) {
      const int2 l = raw[0], r = raw[1];
      put(0, l);
      put(1, r);
      put(2, make_int2((l.x + r.x) >> 1, (l.y + r.y) >> 1));  // coding.rs:483
      put(3, make_int2(l.x - r.x, l.y - r.y));
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) put(b, raw[b]);
    }
  };
  // the history in front of the first tile: zeros (the samples in front of the block); 4 kCertHist values on 64 lanes
  constexpr int kHistPer = 4 * kCertHist / 64;  // values per lane: 2 (f64 rows) or 4
#pragma unroll
  for (int q = 0; q < kHistPer; ++q) {
    const int v = lane * kHistPer + q;  // value v of the 4 x kCertHist: row v / kCertHist, place v % kCertHist
    rows[(v / kCertHist) * kCertRow + (v % kCertHist)] = (row_t)0;
  }
  // this lane's operands: k = lane / 16, block b = (lane % 16) / 4, r = lane % 4 (i for A, j for B)
  const int k = lane >> 4, b = (lane >> 2) & 3, r = lane & 3;
  const row_t* const pa = rows + b * kCertRow + kCertHist + (k - r - 12);      // + 4 m: x_w[T0 + 4 m + k - i - 12]
  const row_t* const pb = rows + b * kCertRow + kCertHist + (k - 12 + 4 * r);  // + 4 m: x_w[T0 + 4 m + k - 12 + 4 j]
  double acc = 0.0;
  constexpr int n_steps = (n + 12 + 3) >> 2;                          // the last column trails by 12 samples
  constexpr int n_tiles = (4 * n_steps + kCertTile - 1) / kCertTile;  // (the tile behind the block's end is all zeros)
  constexpr int NG = kCertTile / 32;                                  // groups of 8 steps per tile
  issue(0);
#pragma unroll 1
  for (int tile = 0; tile < n_tiles; ++tile) {
    land();
    if (tile + 1 < n_tiles) issue((tile + 1) * kCertTile);
    const int left = n_steps - tile * (kCertTile / 4);
    if (left >= kCertTile / 4) {
      // a whole tile: NG groups of 8 steps, the operands of group g + 1 read while group g's chain runs (a lone wave has
      // nothing else to cover an LDS round trip with)
      double oa[8], ob[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        oa[u] = (double)pa[4 * u];
        ob[u] = (double)pb[4 * u];
      }
      if (tile == 0) {
        // B is the current sample: nothing below t = P (only the first group holds such samples: P + 12 < 32)
#pragma unroll
        for (int u = 0; u < 8; ++u) ob[u] = (4 * u + k - 12 + 4 * r) >= P ? ob[u] : 0.0;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        double na[8], nb[8];
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            na[u] = (double)pa[32 * (g + 1) + 4 * u];
            nb[u] = (double)pb[32 * (g + 1) + 4 * u];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(oa[u], ob[u], acc, 0, 0, 0);
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            oa[u] = na[u];
            ob[u] = nb[u];
          }
        }
      }
    } else {
      // the block's last, partial tile (never the first)
      for (int m = 0; m < left; ++m)
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa[4 * m], (double)pb[4 * m], acc, 0, 0, 0);
    }
    // the tile's last kCertHist samples become the next tile's history (the wave's own LDS operations are ordered)
    if (tile + 1 < n_tiles) {
      row_t h[kHistPer];
#pragma unroll
      for (int q = 0; q < kHistPer; ++q) {
        const int v = lane * kHistPer + q;
        h[q] = rows[(v / kCertHist) * kCertRow + kCertTile + (v % kCertHist)];
      }
#pragma unroll
      for (int q = 0; q < kHistPer; ++q) {
        const int v = lane * kHistPer + q;
        rows[(v / kCertHist) * kCertRow + (v % kCertHist)] = h[q];
      }
    }
  }
  // D_b[i][j] sits in lane 16 i + 4 b + j: lag i + 4 j of subframe b
  {
    const int i = lane >> 4, bo = (lane >> 2) & 3, j = lane & 3;
    reinterpret_cast<double*>(smem_raw + rows_off)[16 * bo + i + 4 * j] = acc;
  }
}

// Phase 2 of the fused kernel for one subframe (one lane): the recursion + quantisation (+ the order certificate) on
// R[0 .. MAXP] at `rsrc` (LDS), the quantised predictor to `xq_row` (LDS: [0 .. MAXP) coefficients, [12] order, [13]
// shift, [14] status), the unquantised coefficients to `coefs_out` (global, nullable).  Returns bit 0: certified,
// bit 1: the certificate needed the rows of T^-1.
#ifdef FLACENC_LEV_OUTLINE
#define FLACENC_LEV_ATTR __attribute__((noinline))
#else
#define FLACENC_LEV_ATTR __forceinline__
#endif
template <int MAXP, bool CERT>
__device__ FLACENC_LEV_ATTR int levinson_phase(const double* rsrc, int P, int precision, int32_t* xq_row,
                                                        double* coefs_out, uint32_t max_abs_s, int n_sum, bool do_cert) {
  const double* const Rl = rsrc;  // (levinson_quantize reads R[0 .. P] from it; the certificate's second tier once more)
  double coef[MAXP];
  int32_t cqv[MAXP];
  int warm_v, shift_v, st;
  bool certified = true, tier2 = false;
  if (CERT)
    st = levinson_quantize<MAXP, true>(Rl, P, precision, coef, cqv, &warm_v, &shift_v, max_abs_s, n_sum, &certified, &tier2, do_cert);
  else
    st = levinson_quantize<MAXP>(Rl, P, precision, coef, cqv, &warm_v, &shift_v);
#pragma unroll
  for (int i = 0; i < MAXP; ++i) xq_row[i] = cqv[i];
  xq_row[12] = warm_v;
  xq_row[13] = shift_v;
  xq_row[14] = st;
  if (coefs_out) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) coefs_out[i] = (i < P && st == 0) ? coef[i] : 0.0;
    for (int i = MAXP; i < 32; ++i) coefs_out[i] = 0.0;
  }
  return (certified ? 1 : 0) | (tier2 ? 2 : 0);
}

// ... and the same out of line, without the certificate: the second pass of a workgroup whose first was not certified
template <int MAXP>
__device__ __attribute__((noinline)) void levinson_phase_cold(const double* rsrc, int P, int precision, int32_t* xq_row,
                                                              double* coefs_out) {
  levinson_phase<MAXP, false>(rsrc, P, precision, xq_row, coefs_out, 0u, 0, false);
}

// DECIDE (stereo only): run encode_subframe's candidate choice and try_stereo_coding's channel
// assignment (coding.rs:384-418 without the fixed-LPC candidate, :493-522) on the device and
// write one flacenc_hip_stereo_frame_result + the TWO chosen residual rows per frame.
// CHAINS: the instances that serve QlpcKernelArgs::sumabs_mode (the selector's rare walk of the reference's f32 chains;
// instances of their own: the out-of-line call costs the common path of the others 19 spilled registers)
template <int MAXP, bool STEREO, bool DECIDE, bool FIXED, bool PACK, int SPL = 64, bool CHAINS = false>
__global__ void __launch_bounds__(256, FLACENC_WAVE_OCC) qlpc_wave4096_kernel(QlpcKernelArgs a) {
  static_assert(!CHAINS || (FIXED && !PACK), "the chain walk belongs to the fixed-LPC order selector");
  static_assert(!PACK || (STEREO && DECIDE && FIXED), "the fused bit writer extends the full stereo frame kernel");
  static_assert(SPL == 64 || (SPL == 72 && !PACK), "blocks of 4096 (64 samples per lane) or 4608 (72)");
  using G = WaveGeom<SPL>;
  // (shadow the file-level 4096-block constants, which the big-block kernels share)
  constexpr int kWaveN = G::N;
  constexpr int kSeg = G::Seg;
  constexpr int kBufDwords = G::Buf;
  auto widx = [](int t) { return G::idx(t); };
  // DECIDE with STEREO: encode_frame for a 2-channel frame (four roles + try_stereo_coding);
  // DECIDE without: encode_subframe for four independent channels (Independent(n) frames)
  static_assert(!FIXED || DECIDE, "the fixed-LPC candidate only exists inside encode_subframe's decision");
  constexpr int HP = (MAXP + 3) & ~3;
  constexpr int NLAG = MAXP + 1;
  constexpr int NBUF = STEREO ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* const sm = reinterpret_cast<int32_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // (The hardware already spreads the workgroups' waves: HW_REG_HW_ID stamps -- tools/hwid_probe.py on a -DFLACENC_STAMP_HWID
  // build -- show wave 0 evenly on the four SIMDs, a workgroup's four waves on four different ones, and the waves 0 of two
  // workgroups resident on a CU on one SIMD in 7 % of the pairs; rotating the roles by the workgroup index changes nothing.)
  const int wave = uni(tid >> 6);  // wave-uniform by construction; tell the compiler
  const int P = (int)a.lpc_order;
  const int n = kWaveN;

  // ---- which subframe does this wave own ----
  uint32_t blk = blockIdx.x;
  uint32_t sf;
  int role = 0;
  bool active = true;
  if (STEREO) {
    role = wave;
    sf = blk * 4u + (uint32_t)wave;
  } else {
    sf = blk * 4u + (uint32_t)wave;
    active = sf < a.n_subframes;
    // tail waves of the last workgroup redo the last subframe (identical bytes to the same
    // addresses) so that every wave reaches the workgroup barriers with valid data
    if (!active) sf = a.n_subframes - 1u;
  }
  if (a.stamps && lane == 0 && active) a.stamps[(size_t)sf * 8 + 0] = (unsigned long long)clock64();

  // ======================= phase 0: HBM -> LDS ==============================
  // Segment layout: sample t of a channel image lives at widx(t); segment 0 is zero.
  if (SPL == 64 && STEREO) {
    if (tid < kSeg) sm[tid] = sm[kBufDwords + tid] = 0;
  } else {
    for (int i = tid; i < NBUF * kSeg; i += 256) sm[(i / kSeg) * kBufDwords + (i % kSeg)] = 0;
  }
  // The window table (lpc.rs:96-120, computed on the host) is staged once per workgroup in the
  // same segment layout and shared by the four waves (3 workgroups x 53 KB fit one CU's LDS).
  // Plain mode keeps four sample images (71 KB) and reads the taper weights from the
  // L2-resident table instead, so that two workgroups still fit a CU.
  constexpr bool WINDOW_IN_LDS = window_image(STEREO, SPL);  // (4608: two images are 40 KB; a third would cost the third workgroup per CU)
  constexpr int NIMG = NBUF + (WINDOW_IN_LDS ? 1 : 0);
  // Order 12 at three workgroups per CU: the quantised coefficients wave 0 hands back overlay the R[] rows it was
  // handed (its four lanes have read them, in lockstep, before any of them writes; nobody else reads R[] after
  // the barrier) -- 256 bytes less, which is what 3 x 42 LDS granules leave room for.  The fused bit writer
  // reuses the area for larger things and keeps both.
  // (Round 5: the certificate's second tier reads R[] again after the predictor has been written, so instead of
  // overlaying R[] the quantised predictor of the order-12 stereo instances now goes into the window image -- dead once
  // phase 1 is over -- behind the 5.2 KB the certificate's fallback uses of it; instances without a window image have
  // the room for a predictor area of its own.)
  constexpr bool kXqInWindow = MAXP > 10 && !PACK && window_image(STEREO, SPL);
  constexpr int kXqWindowOff = 1344;  // floats into the window image
  float* const wlds = reinterpret_cast<float*>(sm + NBUF * kBufDwords);
  // The order certificate (levinson_quantize<.., CERT>) and its fallback (reference_chains_from_lds): everywhere but in
  // the fused bit writer (whose exchange area is something else; launch_qlpc hands it the reference's R[] instead).
  // Scratch: the window image, dead once phase 1 is over, where there is one; behind the exchange area otherwise (the
  // plain 4608-sample instances -- four 19.8 KB images -- walk tiles of 32 samples: 1.7 KB is what two workgroups per
  // CU leave).
#ifdef FLACENC_NO_CERT
  constexpr bool kCertSupported = false;  // (diagnostic build: the kernel without the order certificate)
#else
  constexpr bool kCertSupported = !PACK;
#endif
  float* const cert_rows = window_image(STEREO, SPL)
                               ? wlds
                               : reinterpret_cast<float*>(sm + NIMG * kBufDwords) + (4 * (MAXP + 1) * 8 + 256 + 16) / 4;
  const bool has_window = a.window != nullptr;  // nullptr = all ones (rectangle / Tukey(0))
  const float* __restrict__ wtab = a.window + 32;
  const int flat_lo = a.flat_lo, flat_hi = a.flat_hi;
  int tl = lane << 6;         // first sample of this lane
  int lb = (lane + 1) * kSeg;  // its place in an image: widx(tl) (made opaque again at the top of the candidate loop)
  // four window weights at sample tl + off (off a multiple of 4)
  auto window4_at = [&](int ix, int t) -> float4 {  // ix: LDS index (window in LDS), t: sample number (else)
    if (!has_window) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (WINDOW_IN_LDS) return *reinterpret_cast<const float4*>(&wlds[ix]);
    if (FLACENC_WAVE_OCC >= 4) return *reinterpret_cast<const float4*>(wtab + t);  // (exactly 1.0f inside the flat part)
    float4 wv = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // exactly 1.0f inside the flat part
    if (!(t >= flat_lo && t + 4 <= flat_hi)) wv = *reinterpret_cast<const float4*>(wtab + t);
    return wv;
  };

  // Blocks of 4096: quad q = tid + 256 it of a row lands at qidx(tid) + 1088 it -- one lane-dependent LDS address and one
  // lane-dependent byte offset (16 tid) for the whole phase; row bases and iteration strides are scalars / immediates
  // (the generic form below spent 113 VALU instructions per wave on selects and 64-bit address arithmetic).
  [[maybe_unused]] const int q0 = G::qidx(tid);
  [[maybe_unused]] const uint32_t e0 = (uint32_t)tid << 2;  // element offset of quad `tid`
  if (has_window && WINDOW_IN_LDS) {
    const float* __restrict__ wsrc = a.window + 32;
    if (tid < kSeg) wlds[tid] = 0.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *reinterpret_cast<float4*>(&wlds[q0 + it * 16 * kSeg]) = *reinterpret_cast<const float4*>(wsrc + it * 1024 + e0);
  }
  if (STEREO && SPL == 64) {
    const int32_t* __restrict__ srcl = a.samples + (size_t)(2u * blk) * a.stride;
    const int32_t* __restrict__ srcr = srcl + a.stride;
    int4 vl[4], vr[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) vl[it] = *reinterpret_cast<const int4*>(srcl + it * 1024 + e0);
#pragma unroll
    for (int it = 0; it < 4; ++it) vr[it] = *reinterpret_cast<const int4*>(srcr + it * 1024 + e0);
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<int4*>(&sm[q0 + it * 16 * kSeg]) = vl[it];
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<int4*>(&sm[kBufDwords + q0 + it * 16 * kSeg]) = vr[it];
    __syncthreads();
  } else if (STEREO) {
    const int32_t* __restrict__ src = a.samples + (size_t)(2u * blk) * a.stride;
    constexpr int NQ2 = 2 * G::Quads;  // 16-byte pieces of the two channels
#pragma unroll
    for (int it = 0; it < (NQ2 + 255) / 256; ++it) {
      const int q = tid + it * 256;
      if ((NQ2 % 256) == 0 || q < NQ2) {
        const int ch = q >= G::Quads ? 1 : 0;
        const int qq = q - ch * G::Quads;
        const int4 v = *reinterpret_cast<const int4*>(src + (size_t)ch * a.stride + (qq << 2));
        *reinterpret_cast<int4*>(&sm[ch * kBufDwords + G::qidx(qq)]) = v;
      }
    }
    __syncthreads();
  } else {
    {
      const int32_t* __restrict__ src = a.samples + (size_t)sf * a.stride;
#pragma unroll
      for (int it = 0; it < G::Quads / 64; ++it) {
        const int qq = lane + it * 64;
        const int4 v = *reinterpret_cast<const int4*>(src + (qq << 2));
        *reinterpret_cast<int4*>(&sm[wave * kBufDwords + G::qidx(qq)]) = v;
      }
    }
    __syncthreads();  // (also orders the zero segment written by other waves)
  }
  // (inactive tail waves of plain mode stay for the workgroup barriers and write nothing)
  if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + 1] = (unsigned long long)clock64();
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 0
  return;  // diagnostic build: dynamic instruction count of the load phase alone (tools/phase_insts.sh)
#endif

  const int32_t* const bufA = sm + (STEREO ? (role == 1 ? 1 : 0) : wave) * kBufDwords;
  const int32_t* const bufB = sm + kBufDwords;  // right channel (stereo roles 2, 3)
  // four samples of this wave's role starting at t (multiple of 4, >= -64).  The role is
  // wave-uniform; phases are instantiated per role kind so that no branch sits in their loops.
  auto ld4_at = [&](auto kind_tag, int ix) -> int4 {  // ix: index into an image
    constexpr int KIND = decltype(kind_tag)::value;  // 0 = own image, 2 = mid, 3 = side
    // (the empty asm takes the loaded quad as ONE 128-bit operand: with lane-base + constant addresses the
    // optimiser otherwise scalarises the 16-byte loads and re-pairs the dwords as ds_read2_b32 at each use -- and
    // single dwords at a 68-dword lane stride are an 8-way bank conflict where the b128 form is conflict-free)
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
  auto ld4k = [&](auto kind_tag, int off) -> int4 { return ld4_at(kind_tag, G::rel(lb, off)); };  // sample SPL lane + off
  auto ld4abs = [&](auto kind_tag, int t) -> int4 { return ld4_at(kind_tag, widx(t)); };             // sample t
  // run `f(kind_tag)` with the wave's role kind as a compile-time constant
  auto with_role = [&](auto&& f) {
    if (STEREO && role == 2) f(std::integral_constant<int, 2>{});
    else if (STEREO && role == 3) f(std::integral_constant<int, 3>{});
    else f(std::integral_constant<int, 0>{});
  };

  const unsigned long long bps_role = a.bps ? (unsigned long long)a.bps[sf]
                                            : (unsigned long long)(a.bps_uniform + ((STEREO && role == 3) ? 1u : 0u));

  // hoisted: produced by the candidate pass(es) below, consumed by the decision and the records
  // e[] sits behind four extra slots: the fixed-LPC error signal is differenced in place from the lane's
  // 64 samples + the 4 in front of them, so that no second 68-register array is alive next to e[]
  int32_t ebuf[SPL + 4];
  int32_t* const e = ebuf + 4;
  int32_t cq[MAXP];
  int warm = 0, shift = 0, status = 0;
  int role_max = 0, role_min = 0;
  int bestk = 0, rice_order = 0, best_parts = 1;
  unsigned long long best_bits = 0, sum_q = 0, sub_bits = 0;
  uint32_t my_p = 0;
  // Phase 1 and the first half of phase 2 of the QLPC candidate as one unit: window +
  // autocorrelation, R[] handed to wave 0, which solves the workgroup's four systems.  Its
  // results are picked up after a second barrier.  With the fixed-LPC candidate in the kernel this
  // runs BEFORE the fixed candidate's selection and coding pass, so that the other three waves
  // do that work instead of idling while wave 0 is in the serial recursion.
  uint32_t my_maxabs = 0;
  auto lpc_front = [&]() {
  // ======================= phase 1: window + autocorrelation ==============
  double R[NLAG];                        // R[] handed in (acorr_in): the same in every lane
  double Rq[kLaneOrderSlots<NLAG>];      // R[] summed here: slots of the rows' last lanes (lane_order_reduce)
  const bool from_in = a.acorr_in != nullptr;
  int vmax = INT32_MIN, vmin = INT32_MAX;
  if (a.acorr_in != nullptr) {
    // FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: R[] was computed by acorr_reference_kernel in the
    // reference's own summation order; what is left of this phase is the role's max / min
    const double* __restrict__ rin = a.acorr_in + (size_t)sf * 33;
#pragma unroll
    for (int k = 0; k < NLAG; ++k) R[k] = rin[k];
    with_role([&](auto kind) {
#pragma unroll 4
      for (int k = 0; k < SPL / 4; ++k) {
        const int4 v = ld4k(kind, 4 * k);
        vmax = max(max(vmax, v.x), max(v.y, max(v.z, v.w)));
        vmin = min(min(vmin, v.x), min(v.y, min(v.z, v.w)));
      }
    });
  } else
  with_role([&](auto kind) {
    // Phase-1 sample mapping: lane l takes the canonical chunks 4 l .. 4 l + 3 = samples [64 l, 64 l + 64).  For
    // 4096-sample blocks that is the lane's own LDS segment (addresses = lb + constant); for 4608 the 64 samples
    // straddle the 72-sample segments at a lane-dependent place (p1rem), and the block's last 32 chunks are taken
    // one each by lanes 0..31 afterwards (the "tail" below).  p1t = number of the mapping's first sample.
    int p1t = tl, p1base = lb, p1rem = 0;
    auto p1_set = [&](int t_first) {
      p1t = t_first;
      if (SPL != 64) {
        const int sg = (int)(((uint32_t)t_first * 58255u) >> 22);  // / 72
        p1rem = t_first - sg * SPL;
        p1base = (sg + 1) * kSeg + p1rem;
      }
    };
    p1_set(tl);
    auto p1ix = [&](int off, auto fwd_tag) -> int {
      constexpr bool FWD = decltype(fwd_tag)::value;
      if (SPL == 64) return FWD ? lb + off : G::rel(lb, off);
      const int r = p1rem + off;  // -12 <= r < 2 SPL
      return p1base + off + (r >= SPL ? (kSeg - SPL) : 0) - (r < 0 ? (kSeg - SPL) : 0);
    };
    // The lane walks its 64 samples as 4 chunks x 2 steps of 8 with a sliding f64 window of
    // HP halo + 8 new values.  A 16-sample chunk is one fma chain per lag, started with the
    // literal +0.0; chunk partials are combined (c0 + c1) + (c2 + c3): the in-lane levels of
    // the balanced tree over the chunk index c = 4 lane + i.  The chunk loop is deliberately
    // NOT unrolled: unrolled, the compiler hoists every chunk's loads and conversions to the
    // top and needs > 250 VGPRs; rolled it stays under 128 and four waves fit a SIMD.
    constexpr bool PINGPONG = (HP == 8);  // halo == step: two 8-value blocks swap roles, nothing slides
    constexpr int WN = HP + 8;
    double dw[WN];
    double acc[NLAG];
    // raw samples and weights of one 8-sample step, fetched one step ahead of their use
    int4 rv[2];
    float4 rw[2];
    // (orders above 8 carry a 12-value halo and three 11- or 13-entry accumulator sets: there the weights
    // are read where they are used instead of one step ahead, which keeps the loop inside 168 registers)
    constexpr bool PREFETCH_W = (HP == 8);
    int t_conv = 0;
    auto fetch = [&](int t0, auto fwd_tag) {  // fwd: t0 >= 0, plain lane-base + offset addressing
      t_conv = t0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ix = p1ix(t0 + 4 * q, fwd_tag);
        rv[q] = ld4_at(kind, ix);
        if (PREFETCH_W) rw[q] = window4_at(ix, p1t + t0 + 4 * q);
      }
    };
    // x_w[t] = (f32)s[t] * w[t]: one f32 rounding, then widen (lpc.rs:751-754)
    auto convert = [&](int base, auto fwd_tag) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (!PREFETCH_W) rw[q] = window4_at(p1ix(t_conv + 4 * q, fwd_tag), p1t + t_conv + 4 * q);
        dw[base + 4 * q + 0] = (double)((float)rv[q].x * rw[q].x);
        dw[base + 4 * q + 1] = (double)((float)rv[q].y * rw[q].y);
        dw[base + 4 * q + 2] = (double)((float)rv[q].z * rw[q].z);
        dw[base + 4 * q + 3] = (double)((float)rv[q].w * rw[q].w);
      }
    };
    // halo of the lane's first chunk (HP = 8 or 12 samples in front of it)
    fetch(-8, std::false_type{});
    convert(HP, std::false_type{});  // lands in dw[HP .. HP+8): the "previous step" of the first step
    auto halo12 = [&]() {
      if (HP > 8) {
        const int ix = p1ix(-12, std::false_type{});
        const int4 v = ld4_at(kind, ix);
        const float4 wv = window4_at(ix, p1t - 12);
        dw[HP - 4 + 0] = (double)((float)v.x * wv.x);
        dw[HP - 4 + 1] = (double)((float)v.y * wv.y);
        dw[HP - 4 + 2] = (double)((float)v.z * wv.z);
        dw[HP - 4 + 3] = (double)((float)v.w * wv.w);
      }
    };
    halo12();
    fetch(0, std::true_type{});
    // one 16-sample chunk = two 8-sample steps; MASKED only for the block's first chunk, the
    // only one containing t < P (P <= 12 < 16): common lower bound t = P for every lag (lpc.rs:542).
    // PINGPONG: step h writes its 8 values into block (h ^ 1) and finds the previous step's in block h
    // (the chunk body holds both parities, so every index is a compile-time constant); the halo
    // convert above landed in block 1 = what step h = 0 expects.  Otherwise (HP = 12) the last HP
    // values slide down to become the halo.
    auto chunk = [&](auto masked_tag, auto start_tag, int i) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      constexpr bool START = decltype(start_tag)::value;  // the lane's chains start here (literal +0.0), else they go on
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t0 = 16 * i + 8 * h;  // relative to tl
        const int cur0 = PINGPONG ? 8 * h : HP;        // where this step's values go
        const int old0 = PINGPONG ? 8 * (h ^ 1) : 0;   // PINGPONG: the previous step's block
        if (!PINGPONG) {
#pragma unroll
          for (int k = 0; k < HP; ++k) dw[k] = dw[k + 8];  // slide: last HP values become the halo
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          vmax = max(max(vmax, rv[q].x), max(rv[q].y, max(rv[q].z, rv[q].w)));
          vmin = min(min(vmin, rv[q].x), min(rv[q].y, min(rv[q].z, rv[q].w)));
        }
        convert(cur0, std::true_type{});
        fetch(t0 + 8, std::true_type{});  // next step (behind the lane's last step: its own pad + the next segment, unused)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double cur = dw[cur0 + k];
          if (MASKED) cur = (p1t + t0 + k >= P) ? cur : 0.0;
#pragma unroll
          for (int tau = 0; tau <= MAXP; ++tau) {
            // x_w[t - tau]: in this step's block, or tau - k values before the end of the previous one
            const double lagged = PINGPONG ? (tau <= k ? dw[cur0 + k - tau] : dw[old0 + 8 + k - tau])
                                           : dw[HP + k - tau];
            acc[tau] = (START && h == 0 && k == 0) ? __builtin_fma(cur, lagged, 0.0) : __builtin_fma(cur, lagged, acc[tau]);
          }
        }
      }
    };
    // Round 5: the lane's 64 samples are ONE chain per lag (no restart per 16-sample chunk, no in-lane tree: the order
    // certificate, not a shared summation order, is what ties the result to the reference's) ...
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      if (i == 0) chunk(std::true_type{}, std::true_type{}, i);
      else chunk(std::false_type{}, std::false_type{}, i);
    }
    // ... and the 64 lane sums meet by a reduce-scatter (lane_order_reduce): slot q of row r = lane >> 4 ends up with lag
    // 4 q + 2 (r & 1) + (r >> 1), complete in the row's lane 15
    lane_order_reduce<NLAG>(acc, Rq);
    if (SPL != 64) {
      // Blocks of 4608: the last 512 samples, 16 per lane on lanes 0..31 (the upper lanes shadow them and contribute
      // +0.0), a chain of their own per lag, reduced the same way and added to the main part's sums
      p1_set(64 * 64 + 16 * (lane & 31));
      fetch(-8, std::false_type{});
      convert(HP, std::false_type{});
      halo12();
      fetch(0, std::true_type{});
      chunk(std::false_type{}, std::true_type{}, 0);
#pragma unroll
      for (int k = 0; k < NLAG; ++k) acc[k] = lane < 32 ? acc[k] : 0.0;
      double Rt[kLaneOrderSlots<NLAG>];
      lane_order_reduce<NLAG>(acc, Rt);
#pragma unroll
      for (int q = 0; q < kLaneOrderSlots<NLAG>; ++q) Rq[q] = Rq[q] + Rt[q];
    }
  });
  // is_constant (arrayutils.rs:382): all samples of the role equal <=> max == min
  role_max = (int)(wave_max_dpp((uint32_t)vmax ^ 0x80000000u) ^ 0x80000000u);
  role_min = (int)(wave_min_dpp((uint32_t)vmin ^ 0x80000000u) ^ 0x80000000u);
  // max |s| (find_max_abs, arrayutils.rs:509) from the running max / min
  my_maxabs = (uint32_t)max(role_max, -role_min) | (role_min == INT32_MIN ? 0x80000000u : 0u);
  // where this lane's slots go: lag 4 q + lag0 (rows' last lanes only)
  const int lag0 = ((lane >> 3) & 2) + (lane >> 5);
  const bool owner = (lane & 15) == 15;
  if (a.autocorr) {
    if (from_in) {
      if (lane < 33) {
        double rv = 0.0;
#pragma unroll
        for (int k = 0; k < NLAG; ++k)
          if (k == lane && k <= P) rv = R[k];
        a.autocorr[(size_t)sf * 33 + lane] = rv;
      }
    } else {
      if (lane < 33 && lane > P) a.autocorr[(size_t)sf * 33 + lane] = 0.0;
      if (owner) {
#pragma unroll
        for (int q = 0; q < kLaneOrderSlots<NLAG>; ++q)
          if (4 * q + lag0 <= P) a.autocorr[(size_t)sf * 33 + 4 * q + lag0] = Rq[q];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + 2] = (unsigned long long)clock64();
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 1
  asm volatile("s_endpgm");
#endif

  // ======================= phase 2: Levinson + quantisation ================
  // Serial, ~650 VALU instructions, and a wave instruction costs the same with 1 or 64 active
  // lanes -- so the four waves of the workgroup hand their R[] to wave 0, whose lanes 0..3
  // run the recursion for the four subframes side by side (one instruction stream instead of
  // four), and pick their quantised coefficients up again from LDS.
  {
    // exchange area: kept small -- LDS is allocated in 1280-byte granules and three workgroups
    // must fit one CU (3 x 42 granules = 157.5 KB)
    constexpr int XR = NLAG;  // (not rounded up: at order 10 the 32 bytes decide whether three workgroups fit a CU)
    double* const xr = reinterpret_cast<double*>(sm + NIMG * kBufDwords);  // [4][XR]
    int32_t* const xq = kXqInWindow ? reinterpret_cast<int32_t*>(wlds + kXqWindowOff) : reinterpret_cast<int32_t*>(xr + 4 * XR);  // [4][16]
    // the roles' max |s| for the order certificate (16 bytes behind the exchange area: what 42 LDS granules leave at order 10)
    uint32_t* const xm = reinterpret_cast<uint32_t*>(kXqInWindow ? reinterpret_cast<int32_t*>(xr + 4 * XR) : xq + 64);
    const double* const xr_keep = xr;
    const bool certify = kCertSupported && a.certify != 0u && a.acorr_in == nullptr;
    if (from_in) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NLAG; ++k) xr[wave * XR + k] = R[k];
      }
    } else if (owner) {
#pragma unroll
      for (int q = 0; q < kLaneOrderSlots<NLAG>; ++q)
        if (4 * q + lag0 < NLAG) xr[wave * XR + 4 * q + lag0] = Rq[q];
    }
    if (kCertSupported && lane == 0) xm[wave] = my_maxabs;
    __syncthreads();
    // (always wave 0, measured: wave 1 instead +2 %, wave 3 +4 %, rotating with the workgroup index -- blk & 1,
    // blk & 3, (blk + (blk >> 8)) & 3, a hash -- +0.6 to +4 %: the three workgroups of a CU do not stack their
    // recursions on one SIMD, and wave 0 carries the lightest role)
    if (wave == 0) {
      // The recursion on the chunk tree's R[] (+ the order certificate when `certify`); then, only if a subframe was
      // not certified, the workgroup's four subframes once more from the reference's chains (reference_chains_from_lds),
      // adopted by the subframes that need them.  (Straight-line, the second pass through a copy of the recursion that
      // lives out of line: as a two-trip loop around one inlined copy the common path lost 6 %.)
      bool certified = true, need_rows = false;
      uint32_t sfl = blk * 4u + (uint32_t)(lane & 3);
      if (sfl >= a.n_subframes) sfl = a.n_subframes - 1u;
      if (lane < 4) {
        __builtin_amdgcn_s_setprio(3);  // the other three waves of the workgroup wait for this one
        const int fl = levinson_phase<MAXP, kCertSupported>(xr + lane * XR, P, (int)a.precision, xq + lane * 16,
                                                            a.lpc_coefs ? a.lpc_coefs + (size_t)sfl * 32 : nullptr,
                                                            kCertSupported ? xm[lane] : 0u, kWaveN, certify);
        __builtin_amdgcn_s_setprio(0);
        certified = (fl & 1) != 0;
        need_rows = (fl & 2) != 0;
        if (kCertSupported && a.cert_stats != nullptr && certify && blk * 4u + (uint32_t)lane < a.n_subframes) {
          atomicAdd(a.cert_stats + 0, 1u);
          if (need_rows) atomicAdd(a.cert_stats + 1, 1u);
          else if (!certified) atomicAdd(a.cert_stats + 2, 1u);
        }
      }
      // second tier of the certificate (the rows of T^-1): out of line, for the lanes whose first tier could not decide
      if (kCertSupported && __builtin_amdgcn_ballot_w64(lane < 4 && need_rows) != 0ull) {
        if (lane < 4 && need_rows) {
          certified = quant_certified_rows<MAXP>(xr_keep + lane * XR, P, (int)a.precision, xm[lane], kWaveN);
          if (a.cert_stats != nullptr && !certified && blk * 4u + (uint32_t)lane < a.n_subframes) atomicAdd(a.cert_stats + 2, 1u);
        }
      }
      if (kCertSupported && __builtin_amdgcn_ballot_w64(lane < 4 && !certified) != 0ull) {
#ifndef FLACENC_CERT_NO_SLOWPATH
        reference_chains_from_lds<SPL, STEREO>(
            (uint32_t)(reinterpret_cast<unsigned char*>(cert_rows) - smem_raw), has_window ? a.window + 32 : nullptr, P);
#endif
        if (lane < 4 && !certified) {
          const double* const rsrc = reinterpret_cast<const double*>(cert_rows) + 16 * lane;
          if (a.autocorr) {
            for (int k = 0; k <= P; ++k) a.autocorr[(size_t)sfl * 33 + k] = rsrc[k];
          }
          levinson_phase_cold<MAXP>(rsrc, P, (int)a.precision, xq + lane * 16,
                                    a.lpc_coefs ? a.lpc_coefs + (size_t)sfl * 32 : nullptr);
        }
      }
    }
  }
  };
  if (FIXED) lpc_front();

  // ======================= fixed-LPC candidate: order selection ============
  // fixed_lpc (coding.rs:298-331).  `cand` walks the candidates that go through the Rice search:
  // 0..4 = the fixed-LPC error signal of that order, 5 = the QLPC candidate (always last, so e[]
  // ends up holding its residual).  ApproxEnt (coding.rs:265-287) codes only its argmin order;
  // BitCount (:243-264) codes every order up to max_order and keeps the first minimum.
  int cand = 5;
  FixedChoice fx;
  fx.have = false;
  fx.order = 0;
  fx.key = ~0ull;
  fx.bestk = 0;
  fx.my_p = 0;
  fx.code_bits = fx.sum_q = fx.sub_bits = 0;
  // the lane's 64 samples + 4 in front of them (zeros in front of the block: the reference's
  // carry starts at 0, coding.rs:188), differenced `ord` times in place; valid from index ord on
  [[maybe_unused]] auto fixed_load = [&](uint32_t (&v)[SPL + 4]) {
    with_role([&](auto kind) {
#pragma unroll
      for (int k = 0; k < SPL / 4 + 1; ++k) {
        const int4 q = ld4k(kind, -4 + 4 * k);
        v[4 * k + 0] = (uint32_t)q.x;
        v[4 * k + 1] = (uint32_t)q.y;
        v[4 * k + 2] = (uint32_t)q.z;
        v[4 * k + 3] = (uint32_t)q.w;
      }
    });
  };
  auto fixed_load_into = [&](int32_t (&v)[SPL + 4]) {
    with_role([&](auto kind) {
#pragma unroll
      for (int k = 0; k < SPL / 4 + 1; ++k) {
        const int4 q = ld4k(kind, -4 + 4 * k);
        v[4 * k + 0] = q.x;
        v[4 * k + 1] = q.y;
        v[4 * k + 2] = q.z;
        v[4 * k + 3] = q.w;
      }
    });
  };
  if (FIXED && a.use_fixed) {
    if (a.fixed_order_sel == 1u) {
      // ---- estimate_entropy (coding.rs:200-227) for orders 0..max_order ----
      // per-lane sums of |e_k| as exact integers: v_sad_u32 on values biased by 2^31 gives
      // |x - y| of the signed values, i.e. the NEXT order's magnitude, straight from this order's
      // values; 16-sample sub-sums stay below 2^32 for inputs up to 25 bits, then go to f64
      double ls[5];
      const int g = (int)a.fixed_group_log2;
      uint32_t loud = 0;  // wave-uniform: bit k = some partition's order-k sum reaches 2^24 (sumabs_mode, below)
      if (a.sumabs_in != nullptr) {
        // FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: find_sum_abs_f32's own f32 chains (sumabs_reference_kernel)
        const float* __restrict__ sref = a.sumabs_in + (size_t)sf * (5 * 64) + (lane >> g);
#pragma unroll
        for (int ord = 0; ord < 5; ++ord) ls[ord] = (double)sref[ord * 64];
      } else {
      {
        // the role's samples with a bias that makes them non-negative, formed from the images in as few steps
        // as the unbiased value: own image x ^ 2^31 (bias 2^31), mid (l + r + 2^31) >>> 1 (bias 2^30: the sum
        // cannot wrap for inputs of at most 25 bits), side (r ^ 0x7FFFFFFF) + l (bias 2^31 - 1)
        uint32_t b[SPL + 4];
        uint32_t bias0 = 0x80000000u;
        with_role([&](auto kind) {
          constexpr int KIND = decltype(kind)::value;
          typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
          bias0 = KIND == 2 ? 0x40000000u : KIND == 3 ? 0x7FFFFFFFu : 0x80000000u;
#pragma unroll
          for (int k = 0; k < SPL / 4 + 1; ++k) {
            const int ix = G::rel(lb, -4 + 4 * k);
            v4u_t va = *reinterpret_cast<const v4u_t*>(&bufA[ix]);
            asm("" : "+v"(va));
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
#pragma unroll
        for (int ord = 0; ord < 5; ++ord) {
          uint32_t c[5] = {0u, 0u, 0u, 0u, 0u};  // (the fifth: samples 64..71 of a 72-sample lane)
          if (ord == 0) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) c[j >> 4] = sad_u32(b[4 + j], bias0, c[j >> 4]);
          } else {
#pragma unroll
            for (int j = 0; j < SPL; ++j) c[j >> 4] = sad_u32(b[4 + j], b[3 + j], c[j >> 4]);
            if (ord < 4) {
#pragma unroll
              for (int i = SPL + 3; i >= ord; --i) b[i] = xad_u32(b[i - 1], 0x7FFFFFFFu, b[i]);
            }
          }
          ls[ord] = ((double)c[0] + (double)c[1]) + ((double)c[2] + (double)c[3]);
          if (SPL > 64) ls[ord] += (double)c[4];  // (exact integers below 2^53: the order of these adds is immaterial)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // partition sums: partitions of 4096 / P samples = groups of 2^g lanes (P a power of two)
#pragma unroll 1
      for (int lvl = 0; lvl < g; ++lvl) {
#pragma unroll
        for (int ord = 0; ord < 5; ++ord) ls[ord] += __shfl_xor(ls[ord], 1 << lvl, 64);
      }
      // FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER / _NIGHTLY_SUM_ORDER on material of at most 16 bits (sumabs_mode; there
      // launch_qlpc leaves sumabs_reference_kernel out): the reference adds |e| as f32, in either order a sum of
      // non-negative integers whose every partial sum is bounded by the total -- exact, and equal to the sum
      // above, while the total stays below 2^24 (the sums' own conversions included: |e| <= total).  An order
      // with a partition at 2^24 or more ("loud") has roundings that depend on the order of the additions.
      if (CHAINS && a.sumabs_mode != 0u) {
#pragma unroll
        for (int ord = 0; ord < 5; ++ord) loud |= __builtin_amdgcn_ballot_w64(ls[ord] >= 16777216.0) != 0ull ? (1u << ord) : 0u;
      }
      }
      // estimate_entropy for several orders at once: the 2^g lanes of a partition all hold its
      // five sums, so lane j of a group takes order r 2^g + j in pass r.  Its partition estimates
      // are added up over the groups (lanes with equal j) by the butterfly levels >= g, and the
      // first minimum of the keys is the minimum of (key << 3 | order).
      const int G = 1 << g;
      const int jsub = lane & (G - 1);
      const uint32_t psize = (uint32_t)SPL << g;
      // A loud order's key is known only up to the roundings of its chains: each of a partition's `psize`
      // additions rounds by at most 2^-24 of the running sum, estimate_entropy's partition term
      // cnt (avg log2(1 + 1/avg) + log2(avg + 1)), avg = 2 sum / cnt, moves by at most cnt log2(1 + 1/avg) d(avg)
      // <= 1.45 cnt d(sum)/sum -- below 1.45 * 4608^2 * 2^-24 < 2 bits at the largest partition -- and its f32
      // evaluation by a fraction of a bit; 16 bits per partition cover both with a wide margin.  A loud order
      // whose exact-sum key lies further than that above the best quiet key cannot be the minimum, whatever its
      // chains give; only if one might be (or the keys themselves are asked for) are the chains walked.
      const uint32_t loud_slack = 16u * (64u >> g);
      uint32_t best_packed = 0xFFFFFFFFu;
      bool with_slack = CHAINS && loud != 0u;
#pragma unroll 1
      for (;;) {  // (one instance of the evaluation: twice only when chains had to be walked)
        best_packed = 0xFFFFFFFFu;
        uint32_t best_loud = 0xFFFFFFFFu;
#pragma unroll 1
        for (int r = 0; r * G <= (int)a.fixed_max_order; ++r) {
          const int ord = r * G + jsub;
          const bool valid = ord <= (int)a.fixed_max_order;
          double sv = ls[0];
#pragma unroll
          for (int q = 1; q < 5; ++q) sv = (q == ord) ? ls[q] : sv;
          // sample_count = min(end - warmup, partition_len): only partition 0 loses the warm-up
          const uint32_t cnt = psize - ((lane >> g) == 0 ? (uint32_t)ord : 0u);
          uint32_t pb = valid ? approx_ent_bits(sv, cnt) : 0u;
#pragma unroll 1
          for (int lvl = g; lvl < 6; ++lvl) pb += (uint32_t)__shfl_xor((int)pb, 1 << lvl, 64);
          const unsigned long long key = (unsigned long long)pb + bps_role * (unsigned long long)ord;
          if (!with_slack && a.fixed_keys && valid && lane < G) a.fixed_keys[(size_t)sf * 8 + ord] = key;
          const bool is_loud = with_slack && ((loud >> ord) & 1u) != 0u;
          const uint32_t k32 = (uint32_t)key;  // key < 2^29
          const uint32_t packed = (valid && !is_loud) ? ((k32 << 3) | (uint32_t)ord) : 0xFFFFFFFFu;
          const uint32_t m = wave_min_dpp(packed);
          best_packed = m < best_packed ? m : best_packed;
          if (with_slack) {
            const uint32_t lowered = (valid && is_loud) ? (k32 > loud_slack ? k32 - loud_slack : 0u) : 0xFFFFFFFFu;
            const uint32_t ml = wave_min_dpp(lowered);
            best_loud = ml < best_loud ? ml : best_loud;
          }
        }
        if (!CHAINS || !with_slack) break;
        if (a.fixed_keys == nullptr && best_packed != 0xFFFFFFFFu && best_loud > (best_packed >> 3)) break;
        // the partitions' first lanes walk them with the reference's chains (sumabs_chain.h)
        const int gl = G - 1;
        const bool first = (lane & gl) == 0;
        const int pbeg = first ? (lane * SPL) : 0, pend = first ? ((lane + gl + 1) * SPL) : 0;
        float chain[5];
        sumabs_chains_from_lds<SPL>(bufA, bufB, STEREO ? role : 0, pbeg, pend, a.sumabs_mode == 2u ? 1 : 0, chain);
#pragma unroll
        for (int ord = 0; ord < 5; ++ord) {
          const float v = __shfl(chain[ord], lane & ~gl, 64);
          ls[ord] = ((loud >> ord) & 1u) ? (double)v : ls[ord];
        }
        with_slack = false;
      }
      const unsigned long long best_key = (unsigned long long)(best_packed >> 3);
      const int best_ord = (int)(best_packed & 7u);
      fx.key = best_key;
      fx.order = uni(best_ord);
      // fixed_lpc returns None when the estimate does not beat verbatim_bits (coding.rs:284):
      // then no residual is ever coded for it
      cand = best_key < 8ull + (unsigned long long)n * bps_role ? fx.order : 5;
    } else {
      cand = 0;
    }
  }

  auto put_own_e = [&](int32_t* buf) {
#pragma unroll
    for (int k = 0; k < SPL; k += 4) {
      int4 v;
      v.x = e[k + 0];
      v.y = e[k + 1];
      v.z = e[k + 2];
      v.w = e[k + 3];
      *reinterpret_cast<int4*>(&buf[lb + k]) = v;
    }
  };
#pragma unroll 1
  for (;;) {
  // (the LDS addresses below are functions of tl alone: opaque here, they are recomputed per pass instead
  // of being hoisted out of the loop as two dozen registers that then live -- or spill -- across it)
  if (FIXED) {
    asm volatile("" : "+v"(lb));
    lb &= ~3;  // (a no-op: kSeg is a multiple of 4 dwords -- it tells the compiler that the 16-byte LDS accesses are aligned)
  }
  if (FIXED && cand != 5) {
    // ---- the order-`cand` fixed-LPC error signal (coding.rs:182-197) -> e[] ----
    fixed_load_into(ebuf);
#pragma unroll 1
    for (int lvl = 1; lvl <= cand; ++lvl) {
#pragma unroll
      for (int i = SPL + 3; i >= 1; --i) ebuf[i] = (int32_t)((uint32_t)ebuf[i] - (uint32_t)ebuf[i - 1]);
    }
    // the first `order` errors are never coded (Residual keeps zeros there, coding.rs:151-160)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane == 0 && k < cand) e[k] = 0;
    warm = cand;
    status = 0;
  } else {
  if (!FIXED) lpc_front();  // (FIXED: already run before the fixed-LPC work, see there)
  {
    constexpr int XR = NLAG;  // (not rounded up: at order 10 the 32 bytes decide whether three workgroups fit a CU)
    double* const xr = reinterpret_cast<double*>(sm + NIMG * kBufDwords);  // [4][XR]
    int32_t* const xq = kXqInWindow ? reinterpret_cast<int32_t*>(wlds + kXqWindowOff) : reinterpret_cast<int32_t*>(xr + 4 * XR);  // [4][16]
    (void)xr;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXP; ++i) cq[i] = uni(xq[wave * 16 + i]);
    warm = uni(xq[wave * 16 + 12]);
    shift = uni(xq[wave * 16 + 13]);
    status = uni(xq[wave * 16 + 14]);
  }
  int sumabs = 0;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) sumabs += cq[i] < 0 ? -cq[i] : cq[i];
  // compute_error's path choice, lpc.rs:361-377 (+ |s| < 2^23 for the 24-bit multiplier)
  const bool wide = !(((uint64_t)my_maxabs * (uint64_t)sumabs < 0x7FFFFFFFull) && (my_maxabs < (1u << 23)));
  __builtin_amdgcn_sched_barrier(0);
  if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + 3] = (unsigned long long)clock64();
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 2
  asm volatile("s_endpgm");
#endif

  // ======================= phase 3: residual -> registers ==================
  {
    // e[t] = s[t] - ((sum_j c_j s[t-1-j]) >> shift) (lpc.rs:306-350); the i32 / i64 choice of
    // lpc.rs:373-389 is wave-uniform, so it selects one of two straight-line bodies
    auto residual_pass = [&](auto wide_tag, auto kind) {
      constexpr bool WIDE = decltype(wide_tag)::value;
      auto ld4 = [&](int t) { return ld4k(kind, t); };
      int sw[HP + 16];
      constexpr int NCH = (SPL + 15) / 16;  // 4 chunks of 16; 4608-sample blocks: + one of 8
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int t0 = 16 * i;  // relative to the lane's first sample
        const int cn = (SPL - 16 * i) < 16 ? (SPL - 16 * i) : 16;  // samples in this chunk (compile-time per i)
        // compiler-level memory barrier: keeps the next chunk's LDS reads from being hoisted
        // above this chunk's arithmetic (which would cost VGPRs and an occupancy step)
        asm volatile("" ::: "memory");
        if (i > 0) {
#pragma unroll
          for (int k = 0; k < HP; ++k) sw[k] = sw[k + 16];
        }
        const int first = (i == 0) ? 0 : HP;
#pragma unroll
        for (int k = first; k < HP + 16; k += 4) {
          if (k >= HP + cn) continue;
          const int4 v = ld4(t0 - HP + k);
          sw[k + 0] = v.x;
          sw[k + 1] = v.y;
          sw[k + 2] = v.z;
          sw[k + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k >= cn) continue;
          if (!WIDE) {
            int32_t pred = __mul24(cq[0], sw[HP + k - 1]);
#pragma unroll
            for (int j = 1; j < MAXP; ++j) pred = mad24(cq[j], sw[HP + k - 1 - j], pred);
            e[16 * i + k] = sw[HP + k] - (pred >> shift);
          } else {
            int64_t pred = 0;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) pred += (int64_t)cq[j] * (int64_t)sw[HP + k - 1 - j];
            e[16 * i + k] = (int32_t)(uint32_t)(uint64_t)((int64_t)sw[HP + k] - (pred >> shift));
          }
        }
      }
    };
    // Samples and coefficients of at most 16 bits (L, R and M of <= 16-bit material; |c| < 2^14 for any precision
    // <= 15) with the reference's i32 criterion met: two taps per v_dot2_i32_i16 on packed sample pairs
    // P[u] = (lo: s[u], hi: s[u - 1]) -- pred = sum_m dot2((c_2m, c_2m+1), P[t - 1 - 2m]); every partial sum fits
    // i32 because max|s| * sum|c| does (lpc.rs:373-377).  One pack per sample replaces half the multiply-adds.
    auto residual_dot2 = [&](auto kind) {
      auto ld4 = [&](int t) { return ld4k(kind, t); };
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      auto pack = [](int lo, int hi) -> int {  // (hi & 0xFFFF) << 16 | (lo & 0xFFFF): one v_perm_b32
        return (int)__builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
      };
      auto dot2 = [](int c, int p, int acc) -> int {
        return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, c), __builtin_bit_cast(s16x2, p), acc, false);
      };
      // first pair of a chain: the three-operand form with the literal 0 (left to the compiler, every chain
      // starts with a v_mov 0 in front of the accumulate-in-place v_dot2c form)
      auto dot2_first = [](int c, int p) -> int {
        int r;
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "s"(c), "v"(p));
        return r;
      };
      int cp[MAXP / 2];
#pragma unroll
      for (int m = 0; m < MAXP / 2; ++m) cp[m] = uni((cq[2 * m] & 0xFFFF) | (cq[2 * m + 1] << 16));
      int P[HP + 16];  // P[j] pairs window positions j and j - 1 (position HP = the chunk's first sample)
      int prev;
      {
        int hw[HP];
#pragma unroll
        for (int k = 0; k < HP; k += 4) {
          const int4 v = ld4(-HP + k);
          hw[k + 0] = v.x;
          hw[k + 1] = v.y;
          hw[k + 2] = v.z;
          hw[k + 3] = v.w;
        }
#pragma unroll
        for (int j = 1; j < HP; ++j) P[j] = pack(hw[j], hw[j - 1]);
        prev = hw[HP - 1];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t0 = 16 * i;  // relative to tl
        asm volatile("" ::: "memory");
        if (i > 0) {
#pragma unroll
          for (int j = 1; j < HP; ++j) P[j] = P[j + 16];
        }
        int cw[16];
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          const int4 v = ld4(t0 + k);
          cw[k + 0] = v.x;
          cw[k + 1] = v.y;
          cw[k + 2] = v.z;
          cw[k + 3] = v.w;
        }
        P[HP] = pack(cw[0], prev);
#pragma unroll
        for (int k = 1; k < 16; ++k) P[HP + k] = pack(cw[k], cw[k - 1]);
        prev = cw[15];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          int32_t pred = dot2_first(cp[0], P[HP + k - 1]);
#pragma unroll
          for (int m = 1; m < MAXP / 2; ++m) pred = dot2(cp[m], P[HP + k - 1 - 2 * m], pred);
          e[16 * i + k] = cw[k] - (pred >> shift);
        }
      }
    };
    const bool dot_ok = SPL == 64 && !wide && role_min >= -32768 && role_max <= 32767;
    with_role([&](auto kind) {
      constexpr int KIND = decltype(kind)::value;
      // (no v_mad_i32_i24 variant any more: it issued as many instructions as the 64-bit one -- both cost one
      // multiply-add per tap -- and was a third copy of the pass per role in the instruction cache)
      if (KIND != 3 && dot_ok) residual_dot2(kind);
      else residual_pass(std::true_type{}, kind);
    });
    // e[0 .. order') = 0 (lpc.rs:349): only lane 0, only its first 16 slots (order' <= 12)
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if ((lane == 0 && k < warm) || status != 0) e[k] = 0;
    if (status != 0) {
#pragma unroll
      for (int k = 16; k < SPL; ++k) e[k] = 0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + 4] = (unsigned long long)clock64();
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 3
  asm volatile("s_endpgm");
#endif
  }  // QLPC candidate

  // ======================= residual store: registers -> LDS -> HBM ==========
  // (before the Rice search so the stores drain underneath it; with DECIDE only the two chosen
  // roles are stored, after the decision)
  if (!DECIDE) {
    auto put_own = [&](int32_t* buf) {
#pragma unroll
      for (int k = 0; k < SPL; k += 4) {
        int4 v;
        v.x = e[k + 0];
        v.y = e[k + 1];
        v.z = e[k + 2];
        v.w = e[k + 3];
        *reinterpret_cast<int4*>(&buf[lb + k]) = v;
      }
    };
    if (STEREO) {
      int32_t* __restrict__ dst0 = a.residual + (size_t)(blk * 4u) * a.residual_stride;
      __syncthreads();  // every wave is done reading the channel images
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if ((wave >> 1) == half) put_own(sm + (wave & 1) * kBufDwords);
        __syncthreads();
        constexpr int NQ2 = 2 * G::Quads;
#pragma unroll
        for (int it = 0; it < (NQ2 + 255) / 256; ++it) {
          const int q = tid + it * 256;
          if ((NQ2 % 256) == 0 || q < NQ2) {
            const int ch = q >= G::Quads ? 1 : 0;
            const int qq = q - ch * G::Quads;
            const int4 v = *reinterpret_cast<const int4*>(&sm[ch * kBufDwords + G::qidx(qq)]);
            *reinterpret_cast<int4*>(dst0 + (size_t)(2 * half + ch) * a.residual_stride + (qq << 2)) = v;
          }
        }
        if (half == 0) __syncthreads();
      }
    } else {
      int32_t* __restrict__ dst = a.residual + (size_t)sf * a.residual_stride;
      int32_t* own = sm + wave * kBufDwords;
      put_own(own);
#pragma unroll
      for (int it = 0; it < G::Quads / 64; ++it) {
        const int qq = lane + it * 64;
        const int4 v = *reinterpret_cast<const int4*>(&own[G::qidx(qq)]);
        *reinterpret_cast<int4*>(dst + (qq << 2)) = v;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (a.stamps && lane == 0) a.stamps[(size_t)sf * 8 + 5] = (unsigned long long)clock64();

  // ======================= phase 4: partitioned-Rice search ================
  // n = 4096: finest order 6, 64 partitions of 64 samples = one per lane (rice.rs:157-165).
  // bit-sliced population counts of the lane's 64 zig-zag coded residuals -> 7 planes
  // (exact for any magnitude); e[] itself stays intact for the store
  uint32_t pl[7];
  {
    // (scheduling barriers keep the four 16-word groups apart: interleaved, their zig-zag
    // temporaries alone cost ~50 VGPRs on top of the 64 live residuals)
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
      // samples 64..71 of a 72-sample partition: counts <= 72 still fit the seven planes
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
  // a bit is set in some word <=> its count is non-zero <=> it is set in some plane; and the
  // OR of the zig-zag codes u = 2 m + neg is (OR m) << 1 | (any neg)
  const uint32_t orw = wave_or_dpp(pl[0] | pl[1] | pl[2] | pl[3] | pl[4] | pl[5] | pl[6]);
  const uint32_t maxu = (orw << 1) | (orw >> 31);  // (m < 2^31, so nothing is lost by the shift)
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 4
  if (pl[0] + pl[1] + pl[2] + pl[3] + pl[4] + pl[5] + pl[6] == 0x12345u) a.residual[0] = 1;  // (keeps the planes alive)
  asm volatile("s_endpgm");
#endif
  const PlaneSums ps = make_plane_sums(pl);
  // Parameters beyond the residual's bit length can never win (see the generic kernel).
  const uint32_t bitlen = maxu ? (uint32_t)(32 - __builtin_clz(maxu)) : 0u;
  const uint32_t max_p = a.max_rice_parameter < bitlen ? a.max_rice_parameter : bitlen;
  // with the search not cut short by the configuration, every minimum is <= 4 + len*(bitlen+1)
  // and level totals fit 32 bits; otherwise sum in two 16-bit halves
  const bool small_bits = a.max_rice_parameter >= bitlen;
  const uint32_t len0 = (uint32_t)SPL - (lane == 0 ? (uint32_t)warm : 0u);

  const bool finest_only = a.rice_finest_only != 0;
  RiceResult rr;
  unsigned long long sat_sum_q = 0;  // exact sum of quotients, only evaluated if a minimum saturated
  // (the exact sums of a partition must fit 32 bits: 64 codes below 2^26, 72 codes below 2^25)
  if (maxu < (1u << (SPL == 64 ? 26 : 25))) {
    // rice_window: a lower end for the parameter search.  For a partition (or merged group)
    // with sum S over len samples and mean m = S / len let p0 = floor(log2(m + 1)).  From
    // S/2^p - len < sum_i (u_i >> p) <= S/2^p:  table[p0] - 4 < len (p0 + 3)  and, for
    // p <= p0 - 3,  table[p] - 4 > len (m / 2^p + p) >= len (p0 + 4).  So no p <= p0 - 3 can
    // win or tie.  A group's mean is at least the smallest 64-sample partition mean, so the
    // wave-minimum of the (conservatively rounded) per-lane p0 bounds every order.  If the
    // configured max_p lies below that, the same inequalities leave max_p as the only candidate.
    const uint32_t s0 = 2u * ps.sum_m + ps.negs;  // sum of the lane's codes (< 2^32, see above)
    const uint32_t q0 = (s0 >> 6) + 1u;
    // (72-sample partitions: floor(S / 128) + 1 <= S / 72 + 1 keeps the lower end conservative)
    const uint32_t q0lo = SPL == 64 ? q0 : (s0 >> 7) + 1u;
    const uint32_t p0min = wave_min_dpp(31u - (uint32_t)__builtin_clz(q0lo));
    uint32_t p_lo = p0min > 2u ? p0min - 2u : 0u;
    p_lo = p_lo < max_p ? p_lo : max_p;
    // Upper end: table[p + 1] - table[p] = len - sum_i ceil((u_i >> p) / 2) >= len - S / 2^p, which is positive
    // as soon as 2^p > m; 2^(p0 + 1) > m + 1, so from p0 + 1 on the entries grow strictly and no p > p0 + 1 can
    // win or tie (clamped entries can only tie among themselves, and ties go to the smaller p).  A merged
    // group's mean is at most the largest 64-sample partition mean: the wave-maximum of p0 bounds every order.
    // (q bounds the partition's mean from above: floor(S / 64) + 1 > S / 64, and lane 0, whose partition has
    // only 64 - warm >= 52 coded samples, adds S / 256 + 1: S / 64 + S / 256 >= S / 52)
    const uint32_t q0hi = q0 + (lane == 0 ? (s0 >> 8) + 1u : 0u);
    const uint32_t p0max = wave_max_dpp(31u - (uint32_t)__builtin_clz(q0hi));
    uint32_t p_hi = p0max + 1u;
    p_hi = p_hi < max_p ? p_hi : max_p;
    // NOSAT: (1) the configured limit does not cut the search (max_p == bitlen), so a parameter above
    // max_p inside the last group of 8 is legal-but-losing -- for p > bitlen every table entry is
    // len (p + 1), strictly above the entry at p = bitlen -- and needs no masking; (2) bitlen <= 24 keeps
    // every shift amount below 32; (3) no entry of any level reaches the saturation value: an entry of the
    // fully merged table is at most sum u + 4096 * 32, and the sum of the lanes' code sums is bounded from
    // their 64-sample means.  Then clamps never bind and the search equals the clamped one.
    // (Not in the fixed-LPC variants: a second instance of the search inside their candidate loop costs
    // them 35 more spilled dwords at 168 registers.)
    bool nosat = false;
    if (!FIXED) {
      const uint32_t tot_hi = wave_sum_dpp(s0 >> 6);  // sum over lanes of floor(s0 / 64): < 2^32
      nosat = small_bits && bitlen <= 24u && tot_hi < ((kMaxPToBits - 4u - (uint32_t)kWaveN * 32u) >> 6) - 64u;
    }
    if (!FIXED && nosat) rr = rice_search<true, true>(ps, nullptr, len0, p_lo, p_hi, max_p, small_bits, lane, warm, finest_only);
    else rr = rice_search<true>(ps, nullptr, len0, p_lo, p_hi, max_p, small_bits, lane, warm, finest_only);
    // The window argument compares unclamped table values.  If any group minimum saturated at
    // MAX_P_TO_BITS, clamped entries outside the window could tie with it (ties go to the
    // smallest p, rice.rs:123-124), so search the whole range then.
    if (rr.sat_levels != 0 && p_lo != 0)
      rr = rice_search<true>(ps, nullptr, len0, 0u, max_p, max_p, small_bits, lane, warm, finest_only);
    if (rr.saturated) {
      // sum_i (u_i >> p) of this lane's partition under its group's parameter, from the planes
      const uint32_t gp = (uint32_t)__shfl((int)rr.my_p, lane & ~((1 << rr.bestk) - 1), 64);
      const unsigned long long mine = plane_sum_any64(ps, gp);
      sat_sum_q = ((unsigned long long)wave_sum_dpp((uint32_t)(mine >> 16)) << 16) +
                  (unsigned long long)wave_sum_dpp((uint32_t)(mine & 0xFFFFu));
    }
  } else {
    int32_t handed[SPL];  // private memory, written on this path only; e[] itself stays in registers
#pragma unroll
    for (int k = 0; k < SPL; ++k) handed[k] = e[k];
    RiceLiteralResult lit;
    rice_search_literal<SPL>(handed, len0, max_p, small_bits ? 1 : 0, lane, warm, finest_only ? 1 : 0, &lit);
    rr = lit.rr;
    sat_sum_q = lit.sum_q;
  }
#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 5
  if (rr.best_bits + rr.my_p == 0x12345ull) a.residual[0] = 1;
  asm volatile("s_endpgm");
#endif
  bestk = rr.bestk;
  best_bits = rr.best_bits;
  my_p = rr.my_p;
  rice_order = 6 - bestk;
  best_parts = 1 << rice_order;

  // Residual::sum_quotients / count_bits (datatype.rs:2325-2331, bitrepr.rs:533-544)
  const bool leader = (lane & ((1 << bestk) - 1)) == 0;
  const uint32_t sum_p = wave_sum_dpp(leader ? my_p : 0u);
  const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_p);
  // (only the group leaders hold a parameter of the chosen order; the other lanes' my_p is whatever
  // their meaningless merged tables minimised to)
  const uint32_t rice2 = wave_or_dpp((leader && my_p > 14) ? 1u : 0u);
  const unsigned long long rem_bits = (unsigned long long)sum_p * (unsigned long long)(n >> rice_order) -
                                      (unsigned long long)warm * p0;
  sum_q = rr.saturated ? sat_sum_q
                       : best_bits - 4ull * (unsigned long long)best_parts - (unsigned long long)(n - warm) - rem_bits;

  // BitRepr for Residual / Lpc::count_bits, bitrepr.rs:533-544, 492-499 (all wave-uniform)
  const unsigned long long residual_bits = 2ull + 4ull + (unsigned long long)best_parts * (rice2 ? 5ull : 4ull) +
                                           (sum_q + (unsigned long long)(n - warm)) + rem_bits;
  if (FIXED && cand != 5) {
    // BitRepr for FixedLpc::count_bits, bitrepr.rs:473-477; the selector's key for BitCount is
    // bits_per_sample * order + code_bits (coding.rs:249)
    const unsigned long long key =
        a.fixed_order_sel == 1u ? fx.key : bps_role * (unsigned long long)cand + best_bits;
    if (a.fixed_order_sel != 1u && a.fixed_keys && lane == 0) a.fixed_keys[(size_t)sf * 8 + cand] = key;
    if (a.fixed_order_sel == 1u || key < fx.key) {  // first minimum
      fx.key = key;
      fx.order = cand;
      fx.bestk = bestk;
      fx.my_p = my_p;
      fx.code_bits = best_bits;
      fx.sum_q = sum_q;
      fx.sub_bits = 8ull + bps_role * (unsigned long long)cand + residual_bits;
    }
    cand = (a.fixed_order_sel != 1u && cand < (int)a.fixed_max_order) ? cand + 1 : 5;
    continue;
  }
  sub_bits = 8ull + bps_role * (unsigned long long)warm + 4ull + 5ull +
             (unsigned long long)a.precision * (unsigned long long)warm + residual_bits;
  break;
  }  // candidate loop

#if defined(FLACENC_EXIT_AFTER) && FLACENC_EXIT_AFTER == 6
  if (sub_bits + sum_q == 0x12345ull) a.residual[0] = 1;
  asm volatile("s_endpgm");
#endif
  flacenc_hip_subframe_params* rec = a.params ? a.params + sf : nullptr;
  bool fixed_record = false;
  int32_t pack_wsmp = 0;
  // a FixedLpc subframe was chosen: from here on the predictor / Rice variables describe it --
  // FIXED_LPC_COEFS[order] with shift 0 (decode.rs:179-201) and the Rice partition of its error signal
  auto use_fixed_record = [&]() {
    bestk = fx.bestk;
    rice_order = 6 - bestk;
    best_parts = 1 << rice_order;
    my_p = fx.my_p;
    best_bits = fx.code_bits;
    sum_q = fx.sum_q;
    sub_bits = fx.sub_bits;
    warm = fx.order;
    shift = 0;
    status = 0;
    fixed_record = true;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) cq[i] = 0;
    const int o = fx.order;
    cq[0] = o;  // 0, 1, 2, 3, 4
    cq[1] = o == 2 ? -1 : (o == 3 ? -3 : (o == 4 ? -6 : 0));
    cq[2] = o == 3 ? 1 : (o == 4 ? 4 : 0);
    cq[3] = o == 4 ? -1 : 0;
  };
  if (DECIDE) {
    // ---- encode_subframe for this role (coding.rs:384-418) ----
    const bool is_const = a.use_constant && (role_max == role_min);
    const unsigned long long verbatim_bits = 8ull + (unsigned long long)n * bps_role;  // datatype.rs:1944
    // fixed_lpc returns Some iff the selector's key beats verbatim_bits (coding.rs:262, :284)
    const bool have_fixed = FIXED && a.use_fixed && fx.key < verbatim_bits;
    const unsigned long long baseline_bits =
        (have_fixed && fx.sub_bits < verbatim_bits) ? fx.sub_bits : verbatim_bits;  // coding.rs:403-405
    uint32_t kind;
    unsigned long long bits;
    if (is_const) {
      kind = 0u;  // Constant
      bits = 8ull + bps_role;  // bitrepr.rs:445
    } else if (a.use_lpc && status == 0 && sub_bits < baseline_bits) {
      kind = 3u;  // Lpc
      bits = sub_bits;
    } else if (have_fixed && fx.sub_bits < verbatim_bits) {
      kind = 2u;  // FixedLpc
      bits = fx.sub_bits;
    } else {
      kind = 1u;  // Verbatim
      bits = verbatim_bits;
    }
    if (!STEREO) {
      // ---- Independent(n) frames: this wave's subframe is one output channel ----
      if (FIXED && kind == 2u) {
        fixed_load_into(ebuf);
#pragma unroll 1
        for (int lvl = 1; lvl <= fx.order; ++lvl) {
#pragma unroll
          for (int i = SPL + 3; i >= 1; --i) ebuf[i] = (int32_t)((uint32_t)ebuf[i] - (uint32_t)ebuf[i - 1]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lane == 0 && k < fx.order) e[k] = 0;
      } else if (kind < 2u) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) e[k] = 0;
      }
      {
        // own image -> coalesced store (only this wave touches its image)
        int32_t* own = sm + wave * kBufDwords;
        int32_t* __restrict__ dst = a.residual + (size_t)sf * a.residual_stride;
        put_own_e(own);
#pragma unroll
        for (int it = 0; it < G::Quads / 64; ++it) {
          const int qq = lane + it * 64;
          const int4 v = *reinterpret_cast<const int4*>(&own[G::qidx(qq)]);
          *reinterpret_cast<int4*>(dst + (qq << 2)) = v;
        }
      }
      flacenc_hip_channel_result* out = a.chan_results + sf;
      if (lane == 0) {
        out->kind = (uint8_t)kind;
        out->analysis_status = (uint8_t)status;
        out->pad[0] = out->pad[1] = 0;
        out->dc_offset = kind == 0u ? role_max : 0;
        out->bits = bits;
      }
      rec = &out->params;
      if (kind < 2u) {
        for (int i = lane; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += 64)
          reinterpret_cast<uint32_t*>(rec)[i] = 0u;
        rec = nullptr;
      }
    } else {
    // ---- try_stereo_coding (coding.rs:493-522): exchange the four candidates' sizes ----
    // (reuses the R[] exchange area, which nobody reads after the Levinson barriers)
    unsigned long long* const xb = reinterpret_cast<unsigned long long*>(sm + NIMG * kBufDwords);
    if (lane == 0) {
      xb[wave] = bits;
      xb[4 + wave] = ((unsigned long long)(kind | ((uint32_t)status << 8)) << 32) | (unsigned long long)(uint32_t)role_max;
    }
    __syncthreads();  // also: every wave is done reading the channel images
    const unsigned long long bl = xb[0], br = xb[1], bm = xb[2], bs = xb[3];
    unsigned long long min_bits = bl + br;
    int assignment = 0;  // Independent(2)
    if (a.use_leftside && bl + bs < min_bits) {
      min_bits = bl + bs;
      assignment = 1;
    }
    if (a.use_rightside && br + bs < min_bits) {
      min_bits = br + bs;
      assignment = 2;
    }
    if (a.use_midside && bm + bs < min_bits) {
      min_bits = bm + bs;
      assignment = 3;
    }
    assignment = uni(assignment);
    // ChannelAssignment::select_channels (datatype.rs:1173-1185): which role fills output channel 0 / 1
    const int role0 = assignment == 2 ? 3 : (assignment == 3 ? 2 : 0);
    const int role1 = (assignment == 0 || assignment == 2) ? 1 : 3;
    const int slot = (role == role0) ? 0 : ((role == role1) ? 1 : -1);
    flacenc_hip_stereo_frame_result* fr = a.frame_results + blk;
    if (wave == 0 && lane == 0) {
      fr->channel_assignment = (uint8_t)assignment;
      fr->role[0] = (uint8_t)role0;
      fr->role[1] = (uint8_t)role1;
      // the four analyses' status bits (non-zero where the reference panics, lpc.rs:646 / :786-799)
      fr->analysis_status = (uint8_t)(((xb[4] | xb[5] | xb[6] | xb[7]) >> 40) & 0xFFu);
      fr->pad[0] = fr->pad[1] = 0;
      const unsigned long long k0 = xb[4 + role0], k1 = xb[4 + role1];
      fr->kind[0] = (uint8_t)((k0 >> 32) & 0xFFu);
      fr->kind[1] = (uint8_t)((k1 >> 32) & 0xFFu);
      fr->dc_offset[0] = ((k0 >> 32) & 0xFFu) == 0 ? (int32_t)(uint32_t)k0 : 0;
      fr->dc_offset[1] = ((k1 >> 32) & 0xFFu) == 0 ? (int32_t)(uint32_t)k1 : 0;
      fr->bits[0] = bl;
      fr->bits[1] = br;
      fr->bits[2] = bm;
      fr->bits[3] = bs;
    }
    // the two chosen roles hand their residual (zeros unless the LPC candidate was kept) to the
    // channel images, then the whole workgroup streams both rows out
    if (FIXED) {
      // a chosen FixedLpc subframe: its error signal is rebuilt from the channel images, which
      // must therefore stay intact until every wave is past this point
      if (slot >= 0 && kind == 2u) {
        fixed_load_into(ebuf);
#pragma unroll 1
        for (int lvl = 1; lvl <= fx.order; ++lvl) {
#pragma unroll
          for (int i = SPL + 3; i >= 1; --i) ebuf[i] = (int32_t)((uint32_t)ebuf[i] - (uint32_t)ebuf[i - 1]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lane == 0 && k < fx.order) e[k] = 0;
      }
      if (PACK && slot >= 0) {
        // the bit buffer will reuse the image area: keep what the writer still needs from it --
        // the warm-up samples (lane i < 16 holds sample i) and a Verbatim subframe's samples
        with_role([&](auto kind_tag) {
          const int4 q = ld4abs(kind_tag, lane < 16 ? (lane & ~3) : 0);
          pack_wsmp = (lane & 3) == 0 ? q.x : ((lane & 3) == 1 ? q.y : ((lane & 3) == 2 ? q.z : q.w));
          if (kind == 1u) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              const int4 v4 = ld4k(kind_tag, 4 * k);
              e[4 * k + 0] = v4.x;
              e[4 * k + 1] = v4.y;
              e[4 * k + 2] = v4.z;
              e[4 * k + 3] = v4.w;
            }
          }
        });
      }
      __syncthreads();
    }
    if (PACK) {
      // ================= Frame::write (bitrepr.rs:289-319) in the workgroup =================
      // Same construction as frame_pack.cpp: every field is OR-ed into a zeroed LDS bit buffer at its
      // final position; here a lane's 64 residuals are still in its registers.
      uint32_t* const words = reinterpret_cast<uint32_t*>(sm);  // over the two channel images
      uint16_t* const crc_tab = reinterpret_cast<uint16_t*>(sm + NIMG * kBufDwords);  // exchange area
      const unsigned long long bits0 = role0 == 0 ? bl : (role0 == 2 ? bm : bs);
      for (uint32_t i = (uint32_t)tid; i < a.pack_lds_words / 4u; i += 256u)
        reinterpret_cast<int4*>(words)[i] = make_int4(0, 0, 0, 0);
      crc_tab[tid] = (uint16_t)crc16_byte(0u, (uint32_t)tid);
      __syncthreads();
      const uint32_t frame_number = a.pack_first_frame + blk * a.pack_frame_step;
      const uint32_t code_bits_fn = frame_number ? 32u - (uint32_t)__builtin_clz(frame_number) : 0u;
      const uint32_t utf8_len = code_bits_fn <= 7 ? 1u : 1u + (code_bits_fn - 2u) / 5u;
      const uint32_t header_bytes = 4u + utf8_len + a.pack_extra_len + 1u;
      if (wave == 0 && lane == 0) {  // FrameHeader::write, bitrepr.rs:373-419
        uint8_t hdr[16];
        uint32_t hn = 0;
        const uint32_t channel_tag = assignment == 0 ? 1u : 7u + (uint32_t)assignment;
        hdr[hn++] = 0xFF;
        hdr[hn++] = 0xF8;
        hdr[hn++] = (uint8_t)(a.pack_header_mid >> 8);
        hdr[hn++] = (uint8_t)((channel_tag << 4) | (a.pack_header_mid & 0x0Fu));
        if (utf8_len == 1) {
          hdr[hn++] = (uint8_t)frame_number;
        } else {
          const uint32_t trailing = utf8_len - 1u, first_bits = 6u - trailing;
          hdr[hn++] = (uint8_t)(((0xFFu << (8u - trailing - 1u)) & 0xFFu) |
                                ((frame_number >> (6u * trailing)) & ((1u << first_bits) - 1u)));
          for (uint32_t i = 0; i < trailing; ++i)
            hdr[hn++] = (uint8_t)(0x80u | ((frame_number >> (6u * (trailing - 1u - i))) & 0x3Fu));
        }
        for (uint32_t i = 0; i < a.pack_extra_len; ++i) hdr[hn++] = a.pack_extra[i];
        uint32_t crc = 0;
        for (uint32_t i = 0; i < hn; ++i) crc = crc8_byte(crc, hdr[i]);
        hdr[hn++] = (uint8_t)crc;
        for (uint32_t i = 0; i < hn; ++i) put_bits(words, 8u * i, hdr[i], 8u);
      }
      if (slot >= 0) {
        const uint32_t sub_base = header_bytes * 8u + (slot == 1 ? (uint32_t)bits0 : 0u);
        const uint32_t sbps = (uint32_t)bps_role;
        const uint32_t bps_mask = (1u << sbps) - 1u;  // sbps <= 25
        if (kind == 0u) {  // Constant, bitrepr.rs:449-454
          if (lane == 0) put_bits(words, sub_base + 8u, (uint32_t)role_max & bps_mask, sbps);
        } else if (kind == 1u) {  // Verbatim, bitrepr.rs:463-470
          if (lane == 0) put_bits(words, sub_base, 0x02u, 8u);
#pragma unroll
          for (int k = 0; k < 64; ++k)
            put_bits(words, sub_base + 8u + (uint32_t)(tl + k) * sbps, (uint32_t)e[k] & bps_mask, sbps);
        } else {
          if (kind == 2u) use_fixed_record();
          const uint32_t order = (uint32_t)warm;
          const uint32_t precision = fixed_record ? 0u : a.precision;
          const uint32_t head_bits = 8u + order * sbps + (kind == 3u ? 9u + order * precision : 0u);
          if ((uint32_t)lane < order) put_bits(words, sub_base + 8u + (uint32_t)lane * sbps, (uint32_t)pack_wsmp & bps_mask, sbps);
          if (kind == 3u && lane >= 16 && (uint32_t)(lane - 16) < order) {
            int32_t c = 0;
#pragma unroll
            for (int i = 0; i < MAXP; ++i)
              if (i == lane - 16) c = cq[i];
            put_bits(words, sub_base + 8u + order * sbps + 9u + (uint32_t)(lane - 16) * precision,
                     (uint32_t)c & ((1u << precision) - 1u), precision);
          }
          if (lane == 32) {  // FixedLpc::write bitrepr.rs:479-487 / Lpc::write :501-527
            put_bits(words, sub_base, kind == 3u ? (0x40u | ((order - 1u) << 1)) : (0x10u | (order << 1)), 8u);
            if (kind == 3u) {
              put_bits(words, sub_base + 8u + order * sbps, precision - 1u, 4u);
              put_bits(words, sub_base + 8u + order * sbps + 4u, (uint32_t)shift & 31u, 5u);
            }
          }
          // Residual::write, bitrepr.rs:550-597: a partition of the chosen order = 2^bestk lanes
          const bool pleader = (lane & ((1 << bestk) - 1)) == 0;
          const uint32_t r2 = wave_or_dpp((pleader && my_p > 14) ? 1u : 0u);
          const uint32_t pbits = r2 ? 5u : 4u;
          if (lane == 33) put_bits(words, sub_base + head_bits, (r2 << 4) | (uint32_t)(6 - bestk), 6u);
          const uint32_t gp = (uint32_t)__shfl((int)my_p, lane & ~((1 << bestk) - 1), 64);
          uint32_t my_bits = pleader ? pbits : 0u;
#pragma unroll
          for (int k = 0; k < 64; ++k) {
            const bool coded = k >= 16 || lane != 0 || k >= warm;  // the warm-up slots are not coded
            my_bits += coded ? (zigzag32(e[k]) >> gp) + 1u + gp : 0u;
          }
          uint32_t pos = sub_base + head_bits + 6u + wave_excl_scan_dpp(my_bits);
          if (pleader) {
            put_bits(words, pos, gp, pbits);
            pos += pbits;
          }
#pragma unroll
          for (int k = 0; k < 64; ++k) {
            const bool coded = k >= 16 || lane != 0 || k >= warm;
            if (coded) {
              const uint32_t u = zigzag32(e[k]);
              pos += u >> gp;  // unary quotient: zeros
              put_bits(words, pos, (u & ((1u << gp) - 1u)) | (1u << gp), gp + 1u);
              pos += gp + 1u;
            }
          }
        }
      }
      __syncthreads();
      // align_to_byte + CRC-16 (bitrepr.rs:306-316), slices combined as in frame_pack.cpp
      const uint32_t total_bits = header_bytes * 8u + (uint32_t)bits0 +
                                  (uint32_t)(role1 == 1 ? br : bs);
      const uint32_t body_bytes = (total_bits + 7u) >> 3;
      {
        const uint32_t per = a.pack_crc_per;
        const uint32_t k_after = (uint32_t)(255 - tid);
        const uint32_t after = k_after * per;
        uint32_t crc = 0;
        if (after < body_bytes) {
          const uint32_t hi = body_bytes - after;
          const uint32_t lo = hi > per ? hi - per : 0u;
          for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t byte = (words[i >> 2] >> (24u - 8u * (i & 3u))) & 0xFFu;
            crc = ((crc << 8) & 0xFFFFu) ^ crc_tab[(crc >> 8) ^ byte];
          }
          crc = gf_mulmod16(crc, a.pack_crc_pow[k_after & 15u]);
          crc = gf_mulmod16(crc, a.pack_crc_pow[16u + (k_after >> 4)]);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) crc ^= (uint32_t)__shfl_xor((int)crc, d, 64);
        __syncthreads();  // every thread is done with crc_tab's neighbours? (crc_tab is read-only here)
        if (lane == 0) crc_tab[256 + wave] = (uint16_t)crc;  // (the exchange area holds at least 4 x (9 x 8 + 64) = 544 bytes = 272 uint16)
      }
      __syncthreads();
      const uint32_t frame_bytes = body_bytes + 2u;
      if (tid == 0) {
        const uint32_t crc = (uint32_t)crc_tab[256] ^ crc_tab[257] ^ crc_tab[258] ^ crc_tab[259];
        put_bits(words, body_bytes * 8u, crc, 16u);
        a.pack_out_len[blk] = frame_bytes;
      }
      __syncthreads();
      int4* __restrict__ dstp = reinterpret_cast<int4*>(a.pack_out + (size_t)blk * a.pack_out_stride);
      const uint32_t nquads = (frame_bytes + 15u) >> 4;
      for (uint32_t i = (uint32_t)tid; i < nquads; i += 256u) {
        int4 v4 = reinterpret_cast<const int4*>(words)[i];
        v4.x = (int)__builtin_bswap32((uint32_t)v4.x);
        v4.y = (int)__builtin_bswap32((uint32_t)v4.y);
        v4.z = (int)__builtin_bswap32((uint32_t)v4.z);
        v4.w = (int)__builtin_bswap32((uint32_t)v4.w);
        dstp[i] = v4;
      }
    } else {
    if (slot >= 0) {
      if (kind < 2u) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) e[k] = 0;
      }
      put_own_e(sm + slot * kBufDwords);
    }
    __syncthreads();
    {
      int32_t* __restrict__ dst0 = a.residual + (size_t)(blk * 2u) * a.residual_stride;
      // (the thread index is laundered so that these addresses are recomputed here instead of being
      // kept alive -- and, under a tight register budget, spilled to scratch and reloaded at HBM
      // latency -- from the identical expressions of the load phase)
      // Written out per (channel, quarter) with everything but one 32-bit lane offset uniform: the stores then
      // address as scalar base + VGPR offset and the LDS reads as one base + immediates (left as a loop over
      // q = tid + 256 it with ch = q >> 10, each store cost 19 VALU instructions of 64-bit address arithmetic).
      uint32_t tid_out = (uint32_t)tid;
      asm volatile("" : "+v"(tid_out));
      const uint32_t goff = tid_out << 2;  // dwords
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int32_t* const simg = sm + ch * kBufDwords;
        int32_t* __restrict__ const drow = dst0 + (size_t)ch * a.residual_stride;
#pragma unroll
        for (int it = 0; it < (G::Quads + 255) / 256; ++it) {
          // (for 4096: ((tid >> 4) + 1 + 16 it) * 68 + ((tid & 15) << 2), one base + immediates)
          const int qq = (int)tid_out + it * 256;
          if ((G::Quads % 256) == 0 || qq < G::Quads) {
            const int4 v = *reinterpret_cast<const int4*>(&simg[G::qidx(qq)]);
            *reinterpret_cast<int4*>(drow + it * 1024 + goff) = v;
          }
        }
      }
    }
    }  // !PACK
    rec = (slot >= 0) ? &fr->lpc[slot] : nullptr;
    if (rec != nullptr && kind < 2u) {
      // neither an LPC nor a FixedLpc subframe: blank record
      for (int i = lane; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += 64)
        reinterpret_cast<uint32_t*>(rec)[i] = 0u;
      rec = nullptr;
    }
    }  // STEREO
    if (FIXED && rec != nullptr && kind == 2u) use_fixed_record();
  }
  if (rec != nullptr) {
    {
      // partition j of the chosen order lives on lane j << bestk
      const int srcl = (lane << bestk) & 63;
      const uint32_t pv = (uint32_t)__shfl((int)my_p, srcl, 64);
      uint32_t w0 = (lane < best_parts && status == 0) ? pv : 0u;
      rec->rice_params[lane] = (uint8_t)w0;
      rec->rice_params[lane + 64] = 0;
      rec->rice_params[lane + 128] = 0;
      rec->rice_params[lane + 192] = 0;
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
      rec->precision = (uint8_t)(fixed_record ? 0u : a.precision);
      rec->rice_order = (uint8_t)(status == 0 ? rice_order : 0);
      rec->status = status;
      rec->code_bits = status == 0 ? best_bits : 0ull;
      rec->subframe_bits = status == 0 ? sub_bits : 0ull;
      rec->sum_quotients = status == 0 ? sum_q : 0ull;
    }
  }
  if (a.stamps && lane == 0) {
    a.stamps[(size_t)sf * 8 + 6] = (unsigned long long)clock64();
#ifdef FLACENC_STAMP_HWID
    // (diagnostic build: where the wave ran -- HW_REG_HW_ID in the low word, HW_REG_XCC_ID in the high word)
    a.stamps[(size_t)sf * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#else
    a.stamps[(size_t)sf * 8 + 7] = (unsigned long long)clock64();
#endif
  }
}

template <int MAXP, bool STEREO, bool DECIDE, bool FIXED, bool PACK, int SPL = 64, bool CHAINS = false>
hipError_t launch_wave4096(const QlpcKernelArgs& a, hipStream_t stream) {
  auto kern = qlpc_wave4096_kernel<MAXP, STEREO, DECIDE, FIXED, PACK, SPL, CHAINS>;
  // images (+ window, 4096-sample stereo only) + exchange: [4][MAXP + 1] f64 (+ [4][16] i32 unless it sits in the window image, see kXqInWindow)
  // (+ the roles' max |s| and, where no window image can lend it, the scratch of the certificate's fallback)
  constexpr bool cert = !PACK;
  constexpr size_t smem = (size_t)(STEREO ? (window_image(STEREO, SPL) ? 3 : 2) : 4) * WaveGeom<SPL>::Buf * 4 +
                          ((MAXP > 10 && !PACK && window_image(STEREO, SPL)) ? 4 * (MAXP + 1) * 8 : 4 * ((MAXP + 1) * 8 + 64)) +
                          (cert ? 16 : 0) + ((cert && !window_image(STEREO, SPL)) ? cert_scratch_bytes(STEREO, SPL) : 0);
  static DynamicLdsOptIn opt_in;  // per instantiation, per device inside
  if (hipError_t err = opt_in.ensure(reinterpret_cast<const void*>(kern), smem); err != hipSuccess) return err;
  const uint32_t blocks = STEREO ? a.n_subframes / 4u : (a.n_subframes + 3u) / 4u;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, stream, a);
  return hipGetLastError();
}

}  // namespace
}  // namespace flacenc_hip
#endif
