// acorr_reference.cpp -- autocorrelation in the summation orders of the reference's builds.
//
// The stable order runs on the matrix cores (acorr_reference_mfma_kernel, further down); the lag-group kernel
// described first serves the simd-nightly order (and served the stable one until the MFMA form).
//
// weighted_auto_correlation_nosimd (src/lpc.rs:533-548) keeps ONE accumulator per lag and walks the
// block once:  for t in P..n { for tau in 0..=P { R[tau] = fma(x_w[t - tau], x_w[t], R[tau]) } }.
// A chain of n dependent fma per lag cannot be split over lanes without changing the roundings, but the
// chains of different lags are independent.  So a LANE owns (one subframe, four lags) and walks the block
// serially with four chains; a WAVE covers 64 subframes = 16 stereo frames x {L, R, M, S} for one group of
// four lags, and a workgroup is the ceil((P + 1) / 4) waves of those 64 subframes, sharing one staging of
// the windowed signal.  (Round 2 had a lane carry all P + 1 chains of its subframe: 1.5 waves per SIMD at
// the benchmark's batch, each a serial stream of 80 000 instructions -- 580 us where the arithmetic is 130.)
// Global loads stay coalesced by going through LDS tiles: the workgroup reads 64 consecutive samples of each
// of its rows (16 lanes x 16 bytes per row), windows them in f32 exactly as fill_windowed_signal does
// (src/lpc.rs:739-756; M = (l + r) >> 1 and S = l - r formed here, src/coding.rs:483), and stores
// them transposed-friendly (68-float rows) so that lane r then streams row r with ds_read_b128.  Two
// tiles form a ring: the lagged stream of lag group g trails the current one by 4 g samples and reads
// the previous tile's tail.  The next tile's loads are in flight while the current one is summed.
//
// Steps with t < P are masked to x_w[t] = 0 instead of skipped: fma(y, 0, acc) == acc for every
// accumulator value that can occur (accumulators start at +0 and can never become -0), so the chain
// is bit-identical to the reference's, which starts at t = P.  The same holds for the zero padding of
// the last tile of a block that is not a multiple of 64.
#include "acorr_reference.h"

#include <type_traits>

#include "lds_opt_in.h"

namespace flacenc_hip {
namespace {

constexpr int kTile = 64;  // samples per tile
constexpr int kRow = 68;   // floats per LDS row (64 + 4: lane r reading its row with b128 hits its own 4 banks)
constexpr int kLpl = 4;    // lags per lane (LPL = 8 from order 16 on: four waves instead of seven at order 24)
constexpr int kTileFloats = 64 * kRow;

// NIGHTLY: the simd-nightly build's order for blocks that are multiples of 16 samples (no scalar foot), in the same
// frame: per lag d the body's vector lane chains (8 for d < 8, 16 for d = 8..15: lane l takes the samples with
// t mod LANES == l, from the first multiple of LANES at or after P) are 8 / 16 accumulators of the LANE that owns
// the subframe, the scalar head (t = P .. that multiple) one more; waves: lags 0-3, 4-7, then pairs 8-9, 10-11, ...
template <bool STEREO, bool NIGHTLY, int LPL = kLpl>
__global__ void __launch_bounds__(NIGHTLY ? 384 : 576) acorr_reference_kernel(AcorrRefArgs a) {
  static_assert(LPL == 4 || (LPL == 8 && !NIGHTLY), "four lags per lane, or eight in the stable order");
  constexpr int CARRY = LPL;  // lagged values carried from step to step
  __shared__ __attribute__((aligned(16))) float tile[2 * kTileFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = lag group
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf0 = blockIdx.x * 64u;
  const int n_tiles = (n + kTile - 1) / kTile;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0) && ((a.stride & 3) == 0);
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;

  // the tile "in front of the block": zeros (read by the lagged streams, and multiplied by the masked
  // x_w[t] = 0 of the steps t < P -- it must not hold NaN patterns)
  for (int i = tid; i < kTileFloats; i += (int)blockDim.x) tile[kTileFloats + i] = 0.0f;

  // global -> registers, cooperatively and for TWO tiles at a time: item = (row pair or row, 16-byte piece), 16
  // consecutive threads read 256 consecutive bytes of a row for tile k and the 256 behind them for tile k + 1
  // (512-byte runs per row: 256-byte ones alone left the loads at 2.6 TB/s of DRAM page misses).  STEREO: 16
  // frames x 16 pieces, both channels per item (M and S are formed by the thread that holds l and r); plain:
  // 64 rows x 16 pieces.  The launch provides at least 128 (plain: 256) threads, so a thread has at most IT items.
  constexpr int NITEM = STEREO ? 256 : 1024;
  constexpr int IT = STEREO ? 2 : 4;
  constexpr bool PAIR = !NIGHTLY;  // (the nightly form needs the second set's 20 registers for its accumulators: one tile at a time)
  const int nthreads = (int)blockDim.x;
  int4 rA0[IT], rA1[PAIR ? IT : 1], rB0[STEREO ? IT : 1], rB1[(STEREO && PAIR) ? IT : 1];  // set 0: the even tile, set 1: the odd one
  float4 w0, w1;
  auto ld_row = [&](const int32_t* row, int t, bool full) __attribute__((always_inline)) -> int4 {
    if (full) return *reinterpret_cast<const int4*>(row + t);
    int4 v;
    v.x = t + 0 < n ? row[t + 0] : 0;
    v.y = t + 1 < n ? row[t + 1] : 0;
    v.z = t + 2 < n ? row[t + 2] : 0;
    v.w = t + 3 < n ? row[t + 3] : 0;
    return v;
  };
  auto ld_w = [&](int t) __attribute__((always_inline)) -> float4 {
    float4 w = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wtab) {
      w.x = t + 0 < n ? wtab[t + 0] : 0.0f;
      w.y = t + 1 < n ? wtab[t + 1] : 0.0f;
      w.z = t + 2 < n ? wtab[t + 2] : 0.0f;
      w.w = t + 3 < n ? wtab[t + 3] : 0.0f;
    }
    return w;
  };
  // row pointers of the thread's items, once (rows beyond the batch shadow the last valid one: same loads,
  // nothing stored for them at the end)
  const int col = (tid & 15) << 2;  // the same for all of a thread's items: the workgroup size is a multiple of 16
  const int32_t* rowp[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const uint32_t rr = (uint32_t)((tid + it * nthreads) >> 4);
    if (STEREO) {
      uint32_t frame = (sf0 >> 2) + rr;
      const uint32_t last = (a.n_subframes >> 2) - 1u;
      frame = frame < last ? frame : last;
      rowp[it] = a.samples + (size_t)(2u * frame) * a.stride;
    } else {
      uint32_t sf = sf0 + rr;
      sf = sf < a.n_subframes ? sf : a.n_subframes - 1u;
      rowp[it] = a.samples + (size_t)sf * a.stride;
    }
  }
  // tiles k (even) and k + 1
  auto issue_pair = [&](int k) __attribute__((always_inline)) {
    const bool full0 = vec_ok && (k + 1) * kTile <= n;
    const bool full1 = vec_ok && (k + 2) * kTile <= n;
    const bool have1 = PAIR && k + 1 < n_tiles;
    const int t = k * kTile + col;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (tid + it * nthreads < NITEM) {
        rA0[it] = ld_row(rowp[it], t, full0);
        if (STEREO) rB0[it] = ld_row(rowp[it] + a.stride, t, full0);
        if (PAIR && have1) {
          rA1[PAIR ? it : 0] = ld_row(rowp[it], t + kTile, full1);
          if (STEREO) rB1[PAIR ? it : 0] = ld_row(rowp[it] + a.stride, t + kTile, full1);
        }
      }
    }
    w0 = ld_w(t);
    if (have1) w1 = ld_w(t + kTile);
  };
  // registers -> LDS: x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754)
  auto land = [&](auto set_tag, float* buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    const float4 w = SET ? w1 : w0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int item = tid + it * nthreads;
      if (item < NITEM) {
        const int rr = item >> 4;
        auto put = [&](int row, const int4& sv) {
          float4 x;
          x.x = (float)sv.x * w.x;
          x.y = (float)sv.y * w.y;
          x.z = (float)sv.z * w.z;
          x.w = (float)sv.w * w.w;
          *reinterpret_cast<float4*>(&buf[row * kRow + col]) = x;
        };
        if (STEREO) {
          const int4 l = SET ? rA1[PAIR ? it : 0] : rA0[it], r = SET ? rB1[PAIR ? it : 0] : rB0[it];
          put(4 * rr + 0, l);
          put(4 * rr + 1, r);
          put(4 * rr + 2, make_int4((l.x + r.x) >> 1, (l.y + r.y) >> 1, (l.z + r.z) >> 1, (l.w + r.w) >> 1));
          put(4 * rr + 3, make_int4(l.x - r.x, l.y - r.y, l.z - r.z, l.w - r.w));
        } else {
          put(rr, SET ? rA1[PAIR ? it : 0] : rA0[it]);
        }
      }
    }
  };

  // this wave's lags: L0 .. L0 + nl - 1 -- four in the stable order, four or two in the nightly one (32 accumulators);
  // the lagged stream is read from the 16-byte aligned position B <= L0 and indexed OFF = L0 - B further back
  // (nightly: waves 0 and 1 take lags 0-3 and 4-7 with 8 accumulators per lag, the others a pair with 16 each)
  const int L0 = !NIGHTLY ? LPL * wave : (wave < 2 ? 4 * wave : 8 + 2 * (wave - 2));
  const int B = L0 & ~3;
  const int per_wave = (!NIGHTLY || wave < 2) ? LPL : 2;
  const int nl = P + 1 - L0 < per_wave ? P + 1 - L0 : per_wave;  // (<= 0: a wave that only helps with the loads)
  const float* const myrow = tile + lane * kRow;
  // stable: acc[j] is the chain of lag L0 + j.  nightly: acc[LANES j + l] is vector lane l of lag L0 + j, head[j] the
  // scalar chain of the samples between P and the first whole vector
  double acc[NIGHTLY ? 32 : LPL];
  double head[NIGHTLY ? kLpl : 1];
#pragma unroll
  for (int j = 0; j < (NIGHTLY ? 32 : LPL); ++j) acc[j] = 0.0;
#pragma unroll
  for (int j = 0; j < (NIGHTLY ? kLpl : 1); ++j) head[j] = 0.0;
  // lw[i] = x_w[t - B - CARRY + i] for the step at t: CARRY carried + the step's own 8
  double lw[CARRY + 8];
#pragma unroll
  for (int i = 0; i < CARRY + 8; ++i) lw[i] = 0.0;
  // one tile for a wave with NL lags; G0: the wave of lag 0, whose lagged stream is the current one;
  // LANES / OFF: nightly only
  auto sum_tile = [&](auto nl_tag, auto g0_tag, auto masked_tag, auto lanes_tag, auto off_tag, int k) __attribute__((always_inline)) {
    constexpr int NL = decltype(nl_tag)::value;
    constexpr bool G0 = decltype(g0_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int LANES = decltype(lanes_tag)::value;
    constexpr int OFF = decltype(off_tag)::value;
    const float* row = myrow + (k & 1) * kTileFloats;
    auto lag_ptr = [&](int u) { return myrow + ((u >> 6) & 1) * kTileFloats + (u & 63); };  // u % 4 == 0, u >= -64
    const int t0 = (P + LANES - 1) & ~(LANES - 1);  // nightly: the first whole vector of this lag
    auto half = [&](auto h_tag, int step) __attribute__((always_inline)) {
      constexpr int H = decltype(h_tag)::value;  // which half of a 16-sample stretch: the vector lane is kk + 8 H
      const float4 x0 = *reinterpret_cast<const float4*>(row + 8 * step);
      const float4 x1 = *reinterpret_cast<const float4*>(row + 8 * step + 4);
      double cur[8];
      cur[0] = (double)x0.x;
      cur[1] = (double)x0.y;
      cur[2] = (double)x0.z;
      cur[3] = (double)x0.w;
      cur[4] = (double)x1.x;
      cur[5] = (double)x1.y;
      cur[6] = (double)x1.z;
      cur[7] = (double)x1.w;
#pragma unroll
      for (int i = 0; i < CARRY; ++i) lw[i] = lw[8 + i];
      if (G0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lw[CARRY + i] = cur[i];
      } else {
        const int u = k * kTile + 8 * step - B;  // a multiple of 4, >= -32: quads never straddle tiles
        const float4 y0 = *reinterpret_cast<const float4*>(lag_ptr(u));
        const float4 y1 = *reinterpret_cast<const float4*>(lag_ptr(u + 4));
        lw[CARRY + 0] = (double)y0.x;
        lw[CARRY + 1] = (double)y0.y;
        lw[CARRY + 2] = (double)y0.z;
        lw[CARRY + 3] = (double)y0.w;
        lw[CARRY + 4] = (double)y1.x;
        lw[CARRY + 5] = (double)y1.y;
        lw[CARRY + 6] = (double)y1.z;
        lw[CARRY + 7] = (double)y1.w;
      }
      const int tb = k * kTile + 8 * step;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        // t < P: no step of the reference's loop; only the multiplier is masked -- as a lagged value the
        // sample is read by later steps like any other
        const double c = (!MASKED || tb + kk >= P) ? cur[kk] : 0.0;  // (cur[]: converted where it is used, see above)
        if (!NIGHTLY) {
#pragma unroll
          for (int j = 0; j < NL; ++j) acc[j] = __builtin_fma(lw[CARRY + kk - j], c, acc[j]);
        } else {
          const int l = (kk + 8 * H) & (LANES - 1);  // (a constant once the loop is unrolled)
          // the head chain takes the samples below the first whole vector, the lane chain the others; the
          // chain that is not the sample's gets a multiplier of 0 (x + 0 * y == x for every x that can occur),
          // so no value is ever selected between two accumulators -- which would move them to memory
          const bool in_head = MASKED && tb + kk < t0;  // (wave-uniform)
          const double c_head = in_head ? c : 0.0, c_lane = in_head ? 0.0 : c;
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            const double lagged = lw[CARRY + kk - OFF - j];
            if (MASKED) head[j] = __builtin_fma(c_head, lagged, head[j]);
            acc[LANES * j + l] = __builtin_fma(MASKED ? c_lane : c, lagged, acc[LANES * j + l]);
          }
        }
      }
    };
#pragma unroll 1
    for (int step = 0; step < 8; step += 2) {
      half(std::integral_constant<int, 0>{}, step);
      half(std::integral_constant<int, 1>{}, step + 1);
    }
  };
  issue_pair(0);
  __syncthreads();  // (the zero tile)
  land(std::integral_constant<int, 0>{}, tile);
  __syncthreads();
  // (the launch rounds the workgroup up to two -- plain: four -- waves for the loads, so a wave may have no lags)
  auto sum_nl = [&](auto g0_tag, auto masked_tag, int k) __attribute__((always_inline)) {
    constexpr bool G0 = decltype(g0_tag)::value;
    (void)G0;
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    if (nl <= 0) return;
    if (!NIGHTLY) {
      if (nl == 1) sum_tile(std::integral_constant<int, 1>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 2) sum_tile(std::integral_constant<int, 2>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 3) sum_tile(std::integral_constant<int, 3>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (LPL == 4 || nl == 4) sum_tile(std::integral_constant<int, 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 5) sum_tile(std::integral_constant<int, LPL == 8 ? 5 : 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 6) sum_tile(std::integral_constant<int, LPL == 8 ? 6 : 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 7) sum_tile(std::integral_constant<int, LPL == 8 ? 7 : 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else sum_tile(std::integral_constant<int, LPL == 8 ? 8 : 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
    } else if (L0 < 8) {
      if (nl == 1) sum_tile(std::integral_constant<int, 1>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 2) sum_tile(std::integral_constant<int, 2>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else if (nl == 3) sum_tile(std::integral_constant<int, 3>{}, g0_tag, masked_tag, I8{}, I0{}, k);
      else sum_tile(std::integral_constant<int, 4>{}, g0_tag, masked_tag, I8{}, I0{}, k);
    } else if (!G0) {
      if (L0 == B) {
        if (nl == 1) sum_tile(std::integral_constant<int, 1>{}, g0_tag, masked_tag, I16{}, I0{}, k);
        else sum_tile(std::integral_constant<int, 2>{}, g0_tag, masked_tag, I16{}, I0{}, k);
      } else {
        if (nl == 1) sum_tile(std::integral_constant<int, 1>{}, g0_tag, masked_tag, I16{}, I2{}, k);
        else sum_tile(std::integral_constant<int, 2>{}, g0_tag, masked_tag, I16{}, I2{}, k);
      }
    }
  };
  for (int k = 0; k < n_tiles; ++k) {
    if (!PAIR && k + 1 < n_tiles) issue_pair(k + 1);  // (one tile: in flight while tile k is summed)
    if (k * kTile < P || (NIGHTLY && k == 0)) {
      if (wave == 0) sum_nl(std::true_type{}, std::true_type{}, k);
      else sum_nl(std::false_type{}, std::true_type{}, k);
    } else {
      if (wave == 0) sum_nl(std::true_type{}, std::false_type{}, k);
      else sum_nl(std::false_type{}, std::false_type{}, k);
    }
    __syncthreads();  // every wave is done with tile k - 1, whose buffer takes tile k + 1
    if (k + 1 < n_tiles) {
      if (!PAIR || (k & 1)) land(std::integral_constant<int, 0>{}, tile + ((k + 1) & 1) * kTileFloats);
      else land(std::integral_constant<int, 1>{}, tile + kTileFloats);
    }
    __syncthreads();
    // both register sets are parked: the next pair of tiles has the whole of tile k + 1's sums to arrive
    if (PAIR && (k & 1) == 0 && k + 2 < n_tiles) issue_pair(k + 2);
  }
  const uint32_t sf = sf0 + (uint32_t)lane;
  if (sf < a.n_subframes) {
    double* __restrict__ o = a.out + (size_t)sf * 33;
    if (!NIGHTLY) {
#pragma unroll
      for (int j = 0; j < LPL; ++j)
        if (j < nl) o[L0 + j] = acc[j];  // (nl <= 0: a loading-only wave)
    } else {
      // acc = 0 + chain(head); acc += chain(foot) (empty here: + 0); R = acc + reduce_sum(lane chains), the lane
      // sum ordered ((0 + v0) + v1) + ... (weighted_delay_prod_sum_impl, lpc.rs:439-500, as the oracle restates it)
      auto finish = [&](auto lanes_tag) __attribute__((always_inline)) {
        constexpr int LANES = decltype(lanes_tag)::value;
#pragma unroll
        for (int j = 0; j < 32 / LANES; ++j) {
          if (j < nl) {
            double r = 0.0;
            r += head[j];
            r += 0.0;
            double lanesum = 0.0;
#pragma unroll
            for (int l = 0; l < LANES; ++l) lanesum += acc[LANES * j + l];
            o[L0 + j] = r + lanesum;
          }
        }
      };
      if (L0 < 8) finish(std::integral_constant<int, 8>{});
      else finish(std::integral_constant<int, 16>{});
    }
    if (wave == 0)
      for (int tau = P + 1; tau < 33; ++tau) o[tau] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// The simd-nightly build's order (weighted_auto_correlation_simd, src/lpc.rs:510-531, over
// weighted_delay_prod_sum_impl :439-500).  Per lag d the windowed signal from t = P on is split by
// `as_simd::<LANES>()` with LANES = 8 for d < 8 and 16 for d = 8..15 into a scalar head (up to the next
// LANES-aligned element; the buffer is a SimdVec<f32, 16>, 64-byte aligned, lpc.rs:710), a body of whole
// vectors and a scalar foot:
//   acc = 0 + chain(head);  acc_v[l] = fma(x[t + l], x[t + l - d], acc_v[l]) over the body's vectors;
//   acc += chain(foot);  R[d] = acc + reduce_sum(acc_v)   (ordered: ((0 + v0) + v1) + ...).
// For d >= 16 LANES is 32 or 64 and the split depends on where the allocator put the buffer, so the mode
// stops at order 15.  A vector lane's chain takes every LANES-th sample, so here a GPU LANE is a vector lane:
// threads 0..63 are the 8 x 8 lane chains of lags 0..7, threads 64..191 the 8 x 16 of lags 8..15; one
// workgroup per subframe, the windowed block staged once in LDS as f32.
template <bool STEREO>
__global__ void __launch_bounds__(256) acorr_nightly_kernel(AcorrRefArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xw[];
  const int tid = threadIdx.x;
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const uint32_t sf = blockIdx.x;
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
  {
    // x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754); M = (l + r) >> 1, S = l - r (coding.rs:483)
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
    for (int t = tid; t < n; t += 256) {
      int32_t s = rowA[t];
      if (STEREO && kind >= 2) {
        const int32_t r = rowB[t];
        s = kind == 2 ? (s + r) >> 1 : s - r;
      }
      xw[t] = (float)s * (wtab ? wtab[t] : 1.0f);
    }
  }
  __syncthreads();
  int d, l, L;
  if (tid < 64) {
    d = tid >> 3;
    l = tid & 7;
    L = 8;
  } else {
    d = 8 + ((tid - 64) >> 4);
    l = (tid - 64) & 15;
    L = 16;
  }
  // the split of signal[P..]: element i of the buffer is aligned iff i % L == 0
  const int len = n - P;  // >= 1: blocks are at least 64 samples, P <= 15
  const int mis = P % L;
  int head = mis ? L - mis : 0;
  if (head > len) head = len;
  const int nbody = (len - head) / L;
  const int t0 = P + head;
  const bool lag_used = d <= P && d <= 15;
  double acc_v = 0.0;
  if (lag_used) {
    const float* __restrict__ cur = xw + t0 + l;
    int b = 0;
    for (; b + 4 <= nbody; b += 4) {
      const float c0 = cur[0], c1 = cur[L], c2 = cur[2 * L], c3 = cur[3 * L];
      const float g0 = cur[0 - d], g1 = cur[L - d], g2 = cur[2 * L - d], g3 = cur[3 * L - d];
      acc_v = __builtin_fma((double)c0, (double)g0, acc_v);
      acc_v = __builtin_fma((double)c1, (double)g1, acc_v);
      acc_v = __builtin_fma((double)c2, (double)g2, acc_v);
      acc_v = __builtin_fma((double)c3, (double)g3, acc_v);
      cur += 4 * L;
    }
    for (; b < nbody; ++b) {
      acc_v = __builtin_fma((double)cur[0], (double)cur[0 - d], acc_v);
      cur += L;
    }
  }
  // ordered lane sum, gathered by the group's first lane (groups are aligned runs of L lanes of one wave)
  double lanesum = 0.0;
  const int base = (tid & 63) & ~(L - 1);
  for (int j = 0; j < L; ++j) {
    const double v = __shfl(acc_v, base + j, 64);
    lanesum += v;
  }
  if (l == 0 && d <= 32) {
    double r = 0.0;
    if (lag_used) {
      double acc = 0.0, h = 0.0, f = 0.0;
      for (int t = P; t < t0; ++t) h = __builtin_fma((double)xw[t - d], (double)xw[t], h);
      acc += h;
      for (int t = t0 + nbody * L; t < n; ++t) f = __builtin_fma((double)xw[t - d], (double)xw[t], f);
      acc += f;
      r = acc + lanesum;
    }
    a.out[(size_t)sf * 33 + d] = r;
  }
  // lags 16..32 are not produced in this mode: written as 0 like every lag above P
  if (tid >= 192 && tid < 192 + 17) a.out[(size_t)sf * 33 + 16 + (tid - 192)] = 0.0;
}

hipError_t launch_nightly(const AcorrRefArgs& a, hipStream_t stream) {
  if (a.lpc_order > 15) return hipErrorNotSupported;
  const size_t smem = (((size_t)a.block_size + 3) & ~(size_t)3) * sizeof(float);
  static DynamicLdsOptIn opt_s, opt_p;
  if (a.stereo) {
    if (hipError_t e = opt_s.ensure(reinterpret_cast<const void*>(acorr_nightly_kernel<true>), smem); e != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_nightly_kernel<true>), dim3(a.n_subframes), dim3(256), smem, stream, a);
  } else {
    if (hipError_t e = opt_p.ensure(reinterpret_cast<const void*>(acorr_nightly_kernel<false>), smem); e != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_nightly_kernel<false>), dim3(a.n_subframes), dim3(256), smem, stream, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The stable order on the matrix cores.  v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 x 4 blocks,
// D_b[i][j] += sum_k A_b[i][k] B_b[k][j], and chained through its C operand it performs, per output, exactly the
// sequential chain acc = fma(A_b[i][k], B_b[k][j], acc) with k ascending (tools/microbench/mfma_f64_4x4x4_probe.hip:
// lane 16 k + 4 b + i holds A_b[i][k], lane 16 k + 4 b + j holds B_b[k][j], lane 16 i + 4 b + j holds D_b[i][j];
// 32 000 outputs of 64-instruction chains over +-20 binades of operands, no mismatch).  With
//     A_b[i][k] = x_w[4 m + k - i - 12]        (the lagged sample)
//     B_b[k][j] = x_w[4 m + k - 12 + 4 j]      (the current sample; 0 below t = P and from t = n on)
// instruction m adds x_w[t - (i + 4 j)] x_w[t] for t = 4 m + k - 12 + 4 j, k = 0..3, to output (i, j): sixteen lags
// tau = i + 4 j per block, every output one chain over ascending t -- weighted_auto_correlation_nosimd's own
// (src/lpc.rs:533-548), the terms the reference does not have contributing fma(., 0, acc) = acc.  The Toeplitz
// structure that keeps the autocorrelation off a GEMM tile is exactly what a 4 x 4 block with one
// lag digit per axis absorbs.  A block is a subframe: a wave carries the four roles of one stereo frame (or four
// plain subframes), 256 fma per instruction in 16 cycles (the vector pipe's f64 rate; what the form saves is operand
// handling: one LDS value per four fma).  Orders above
// 15 add a second and third accumulator with A shifted by 16 and 32 lags.
// Each wave stages its own four rows (f32, windowed, M / S formed) in its own LDS region: [64 samples of history |
// a tile of 256], no workgroup barrier anywhere; the tile's loads are 1 KB runs per row.
constexpr int kMTile = 256;   // samples per staging step
constexpr int kMHist = 64;    // samples kept in front of the tile (lags up to 47 + the 12 of the column shift)
constexpr int kMRow = kMHist + kMTile + 8;   // row stride = 8 mod 32 banks: the four blocks' A reads (7 addresses each)
                                             // fall on disjoint banks, their B reads (16 each) two to a bank -- the minimum

template <bool STEREO, int NS>
__global__ void __launch_bounds__(256) acorr_reference_mfma_kernel(AcorrRefArgs a) {
  // (f64 in LDS -- no conversion per operand, twice the LDS traffic, 3 workgroups per CU -- is 2.4 x slower; round 4 tried it
  // again de-interleaved by sample phase, a lane's operand sequence contiguous and two steps per ds_read2_b64: 510 us
  // against 344 at order 24 on 6144 frames of 8192 -- the LDS, not the conversions, is what that form waits for)
  typedef float elem_t;
  __shared__ __attribute__((aligned(16))) elem_t lds[4][4 * kMRow];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  elem_t* const rows = lds[wave];  // row b at rows + b * kMRow
  const int n = (int)a.block_size;
  const int P = (int)a.lpc_order;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0) && ((a.stride & 3) == 0);
  const float* __restrict__ wtab = a.window ? a.window + 32 : nullptr;
  // the wave's four subframes: unit = blockIdx.x * 4 + wave (stereo: a frame; plain: a group of four subframes) -- or, in
  // the sub-wave kernel's clean-up (marked_params; round 6), the units of a small grid's stride that hold a record of
  // status -2 (int32 at byte 68 of the 352-byte record): the launch normally finds the count of marked records at 0 and
  // must cost next to nothing then
  auto process = [&](const uint32_t unit) __attribute__((always_inline)) {
  const uint32_t sf0 = unit * 4u;
  if (sf0 >= a.n_subframes) return;  // (whole wave; no barriers in this kernel)
  if (a.marked_params != nullptr) {
    const unsigned char* const recs = static_cast<const unsigned char*>(a.marked_params);
    bool any = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t sf = sf0 + (uint32_t)r;
      if (sf < a.n_subframes) any |= *reinterpret_cast<const int32_t*>(recs + (size_t)sf * 352u + 68u) == -2;
    }
    if (!any) return;
  }
  const int32_t* rowp[STEREO ? 2 : 4];
  if (STEREO) {
    rowp[0] = a.samples + (size_t)(2u * unit) * a.stride;
    rowp[1] = rowp[0] + a.stride;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t sf = sf0 + (uint32_t)r;
      sf = sf < a.n_subframes ? sf : a.n_subframes - 1u;  // (a ragged last group shadows the last subframe)
      rowp[r] = a.samples + (size_t)sf * a.stride;
    }
  }
  // global -> registers for the tile at T0: lane l holds samples T0 + 4 l .. + 3 of each row
  int4 raw[STEREO ? 2 : 4];
  float4 wv;
  // (whole quads: with aligned rows and n a multiple of four a lane's four samples are all inside the block or all behind
  // it, so the block's last, partial tile and the tile behind it -- a fifth of the tiles of a 1152-sample block, half of a
  // 256-sample block's -- load like the full ones instead of element by element)
  const bool quads = vec_ok && (n & 3) == 0;
  const bool wquads = wtab != nullptr && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(wtab) & 15) == 0;
  auto issue = [&](int T0) __attribute__((always_inline)) {
    const int t = T0 + 4 * lane;
    const bool full = vec_ok && T0 + kMTile <= n;
#pragma unroll
    for (int r = 0; r < (STEREO ? 2 : 4); ++r) {
      if (full || (quads && t < n)) {
        raw[r] = *reinterpret_cast<const int4*>(rowp[r] + t);
      } else if (quads) {
        raw[r] = make_int4(0, 0, 0, 0);
      } else {
        raw[r].x = t + 0 < n ? rowp[r][t + 0] : 0;
        raw[r].y = t + 1 < n ? rowp[r][t + 1] : 0;
        raw[r].z = t + 2 < n ? rowp[r][t + 2] : 0;
        raw[r].w = t + 3 < n ? rowp[r][t + 3] : 0;
      }
    }
    wv = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (wquads) {
      wv = t < n ? *reinterpret_cast<const float4*>(wtab + t) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else if (wtab) {
      wv.x = t + 0 < n ? wtab[t + 0] : 0.0f;
      wv.y = t + 1 < n ? wtab[t + 1] : 0.0f;
      wv.z = t + 2 < n ? wtab[t + 2] : 0.0f;
      wv.w = t + 3 < n ? wtab[t + 3] : 0.0f;
    }
  };
  // registers -> the tile part of the rows: x_w = (f32)s * w, one f32 rounding (lpc.rs:751-754)
  auto land = [&]() __attribute__((always_inline)) {
    auto put = [&](int b, const int4& sv) {
      float4 x;
      x.x = (float)sv.x * wv.x;
      x.y = (float)sv.y * wv.y;
      x.z = (float)sv.z * wv.z;
      x.w = (float)sv.w * wv.w;
      *reinterpret_cast<float4*>(&rows[b * kMRow + kMHist + 4 * lane]) = x;
    };
    if (STEREO) {
      const int4 l = raw[0], r = raw[1];
      put(0, l);
      put(1, r);
      put(2, make_int4((l.x + r.x) >> 1, (l.y + r.y) >> 1, (l.z + r.z) >> 1, (l.w + r.w) >> 1));  // coding.rs:483
      put(3, make_int4(l.x - r.x, l.y - r.y, l.z - r.z, l.w - r.w));
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) put(b, raw[b]);
    }
  };
  // the history in front of the first tile: zeros (the samples in front of the block)
  rows[(lane >> 4) * kMRow + (lane & 15)] = (elem_t)0;
  rows[(lane >> 4) * kMRow + 16 + (lane & 15)] = (elem_t)0;
  rows[(lane >> 4) * kMRow + 32 + (lane & 15)] = (elem_t)0;
  rows[(lane >> 4) * kMRow + 48 + (lane & 15)] = (elem_t)0;

  // this lane's operands: k = lane / 16, block b = (lane % 16) / 4, r = lane % 4 (i for A, j for B)
  const int k = lane >> 4, b = (lane >> 2) & 3, r = lane & 3;
  const elem_t* const pa = rows + b * kMRow + kMHist + (k - r - 12);      // + 4 m: x_w[T0 + 4 m + k - i - 12]
  const elem_t* const pb = rows + b * kMRow + kMHist + (k - 12 + 4 * r);  // + 4 m: x_w[T0 + 4 m + k - 12 + 4 j]
  double acc[NS];
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) acc[s_] = 0.0;

  const int n_steps = (n + 12 + 3) >> 2;                   // instructions: the last column trails by 12 samples
  const int n_tiles = (4 * n_steps + kMTile - 1) / kMTile;  // (the tile after the block's end, if any, is all zeros)
  issue(0);
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int T0 = tile * kMTile;
    land();
    if (tile + 1 < n_tiles) issue(T0 + kMTile);  // in flight while this tile is summed
    const int m_end = (n_steps - tile * (kMTile / 4)) < (kMTile / 4) ? (n_steps - tile * (kMTile / 4)) : (kMTile / 4);
    // B is the current sample: nothing below t = P (only the first tile can hold such samples; P + 12 < 64)
    int m = 0;
    if (tile == 0) {
      const int m_mask = (P + 12 + 3) >> 2;  // steps that can touch t < P
      for (; m < m_mask && m < m_end; ++m) {
        const int tcur = 4 * m + k - 12 + 4 * r;
        const double bd = tcur >= P ? (double)pb[4 * m] : 0.0;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
          acc[s_] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa[4 * m - 16 * s_], bd, acc[s_], 0, 0, 0);
      }
    }
    if (m == 0 && m_end == kMTile / 4) {
      // a whole tile: constant trip count, reads and MFMAs interleaved by the scheduler.  (Fetching 16 steps'
      // operands ahead of the previous batch's MFMAs costs 104 registers and occupancy: 484 us against 444.  A step
      // costs a SIMD 11.5 ns -- 7.4 for the MFMA, 2 x 2 for the conversions, one after the other.)
#pragma unroll 16
      for (int mm = 0; mm < kMTile / 4; ++mm) {
        const double bd = (double)pb[4 * mm];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
          acc[s_] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa[4 * mm - 16 * s_], bd, acc[s_], 0, 0, 0);
      }
    } else {
      for (; m < m_end; ++m) {
        const double bd = (double)pb[4 * m];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
          acc[s_] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)pa[4 * m - 16 * s_], bd, acc[s_], 0, 0, 0);
      }
    }
    // the tile's last 64 samples become the next tile's history (the wave's own LDS operations are ordered)
    if (tile + 1 < n_tiles) {
      const elem_t h = rows[(lane >> 4) * kMRow + kMTile + (lane & 15)];
      const elem_t h1 = rows[(lane >> 4) * kMRow + kMTile + 16 + (lane & 15)];
      const elem_t h2 = rows[(lane >> 4) * kMRow + kMTile + 32 + (lane & 15)];
      const elem_t h3 = rows[(lane >> 4) * kMRow + kMTile + 48 + (lane & 15)];
      rows[(lane >> 4) * kMRow + (lane & 15)] = h;
      rows[(lane >> 4) * kMRow + 16 + (lane & 15)] = h1;
      rows[(lane >> 4) * kMRow + 32 + (lane & 15)] = h2;
      rows[(lane >> 4) * kMRow + 48 + (lane & 15)] = h3;
    }
  }
  // D_b[i][j] sits in lane 16 i + 4 b + j: lag i + 4 j (+ 16 per further accumulator) of subframe sf0 + b
  {
    const int i = lane >> 4, bo = (lane >> 2) & 3, j = lane & 3;
    const uint32_t sf = sf0 + (uint32_t)bo;
    if (sf < a.n_subframes) {
      double* __restrict__ o = a.out + (size_t)sf * 33;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const int lag = i + 4 * j + 16 * s_;
        if (lag <= 32) o[lag] = lag <= P ? acc[s_] : 0.0;
      }
      if (NS < 3 && i + 4 * j == 0) {
        for (int lag = 16 * NS; lag < 33; ++lag) o[lag] = 0.0;
      }
    }
  }
  };  // process
  if (a.marked_params == nullptr) {
    process(blockIdx.x * 4u + (uint32_t)wave);
  } else {
    const uint32_t count = a.marked_count != nullptr ? __hip_atomic_load(a.marked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
    if (count == 0u) return;
    if (a.marked_list != nullptr && count <= a.marked_cap) {
      // the marks' own list: entry e = a record (marked_unit 1) or a stereo frame (4); this kernel's unit is four records.
      // (A group of four that holds several marked records is walked once per mark: the same sums, written again.)
      for (uint32_t i = blockIdx.x * 4u + (uint32_t)wave; i < count; i += gridDim.x * 4u) {
        const uint32_t e = a.marked_list[i];
        process(a.marked_unit == 4u ? e : e >> 2);
      }
      return;
    }
    for (uint32_t unit = blockIdx.x * 4u + (uint32_t)wave; unit * 4u < a.n_subframes; unit += gridDim.x * 4u) process(unit);
  }
}

template <int NS>
hipError_t launch_mfma(const AcorrRefArgs& a, hipStream_t stream) {
  const uint32_t units = a.stereo ? a.n_subframes / 4u : (a.n_subframes + 3u) / 4u;
  uint32_t blocks = (units + 3u) / 4u;
  if (a.marked_params != nullptr && blocks > 512u) blocks = 512u;  // (a grid-stride walk of the records: see the kernel)
  if (a.stereo) hipLaunchKernelGGL((acorr_reference_mfma_kernel<true, NS>), dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((acorr_reference_mfma_kernel<false, NS>), dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_lag_groups(const AcorrRefArgs& a, hipStream_t stream) {
  const uint32_t blocks = (a.n_subframes + 63u) / 64u;
  uint32_t groups;  // lag groups = waves (1..9; nightly: lags 0-3, 4-7, then pairs: 1..6)
  const bool wide = !a.nightly && a.lpc_order >= 16u;  // eight lags per lane: 3..5 waves where four per lane need 5..9
  if (!a.nightly) groups = wide ? (a.lpc_order + 1u + 7u) / 8u : (a.lpc_order + 1u + (uint32_t)kLpl - 1u) / (uint32_t)kLpl;
  else groups = a.lpc_order < 4u ? 1u : (a.lpc_order < 8u ? 2u : 2u + (a.lpc_order + 1u - 8u + 1u) / 2u);
  const uint32_t min_waves = a.stereo ? 2u : 4u;                     // (the cooperative loads want 128 / 256 threads)
  groups = groups < min_waves ? min_waves : groups;
  const dim3 block(64u * groups);
  if (a.nightly) {
    if (a.stereo) hipLaunchKernelGGL((acorr_reference_kernel<true, true>), dim3(blocks), block, 0, stream, a);
    else hipLaunchKernelGGL((acorr_reference_kernel<false, true>), dim3(blocks), block, 0, stream, a);
  } else if (wide) {
    if (a.stereo) hipLaunchKernelGGL((acorr_reference_kernel<true, false, 8>), dim3(blocks), block, 0, stream, a);
    else hipLaunchKernelGGL((acorr_reference_kernel<false, false, 8>), dim3(blocks), block, 0, stream, a);
  } else {
    if (a.stereo) hipLaunchKernelGGL((acorr_reference_kernel<true, false>), dim3(blocks), block, 0, stream, a);
    else hipLaunchKernelGGL((acorr_reference_kernel<false, false>), dim3(blocks), block, 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_acorr_reference(const AcorrRefArgs& a, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  if (a.stereo && (a.n_subframes & 3u)) return hipErrorInvalidValue;
  if (a.lpc_order > 32u) return hipErrorInvalidValue;
  if (a.nightly) {
    if (a.lpc_order > 15u) return hipErrorNotSupported;
    // blocks that are whole vectors of 16 (no scalar foot): the lane-per-subframe form; any other size: one
    // workgroup per subframe with a GPU lane per vector lane
    if ((a.block_size & 15u) != 0u || a.block_size < 64u) return launch_nightly(a, stream);
  }
  if (!a.nightly) {  // the stable order: on the matrix cores
    if (a.lpc_order <= 15u) return launch_mfma<1>(a, stream);
    if (a.lpc_order <= 31u) return launch_mfma<2>(a, stream);
    return launch_mfma<3>(a, stream);
  }
  return launch_lag_groups(a, stream);
}

}  // namespace flacenc_hip
