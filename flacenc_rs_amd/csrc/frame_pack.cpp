// frame_pack.cpp -- BitRepr::write for Frame / FrameHeader / SubFrame / Residual
// (src/component/bitrepr.rs:289-319, 373-419, 449-597 of flacenc-rs v0.5.1) on the GPU.
//
// One workgroup assembles one 2-channel frame in LDS from what flacenc_hip_encode_stereo_frames
// left on the device: the frame record (channel assignment, subframe kinds, predictor parameters,
// Rice partition) and the two residual rows.  The frame's bits are built in a zero-initialised LDS
// bit buffer; every field is OR-ed in at its final position, so nothing is written sequentially:
//   * the sizes of both subframes are already known exactly (SubFrame::count_bits in the record),
//   * inside a Rice-coded residual a block-wide prefix sum over per-thread bit counts gives each
//     thread the position of its first sample; a sample is `q` skipped (zero) bits followed by
//     (1 << p | r) in p + 1 bits (bitrepr.rs:581-586), i.e. one OR of at most two words,
//   * CRC-16 (CRC_16_UMTS, init 0, no xor-out: a plain polynomial remainder, hence linear) is
//     computed per thread over a slice of the bytes and combined with x^(8 * bytes after the slice).
#include "frame_pack.h"

#include "frame_bits.h"
#include "lds_opt_in.h"

#ifndef FLACENC_DPP
#define FLACENC_DPP(v, ctrl, rowmask) \
  ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rowmask), 0xF, false))
#endif

namespace flacenc_hip {
namespace {

constexpr int kPackThreads = 256;
#ifndef FLACENC_PACK_OCC
#define FLACENC_PACK_OCC 8  // waves per SIMD asked of the register allocator (<= 64 VGPRs)
#endif

// block-wide exclusive prefix sum of one value per thread (256 threads); `total` gets the sum
__device__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, int tid, uint32_t* total) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += scratch[w];
  *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return base + incl - v;
}

// slicing-by-4 tables of CRC_16_UMTS, made at compile time: t[b][k] = CRC-16 of byte b followed by k zero
// bytes (one 8-byte load per thread brings them into LDS)
struct Crc16Tables {
  uint16_t t[256][4];
};
constexpr uint32_t crc16_step(uint32_t crc, uint32_t byte) {
  crc ^= byte << 8;
  for (int b = 0; b < 8; ++b) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x8005u) & 0xFFFFu : (crc << 1) & 0xFFFFu;
  return crc;
}
constexpr Crc16Tables make_crc16_tables() {
  Crc16Tables r{};
  for (uint32_t b = 0; b < 256; ++b) {
    uint32_t c = crc16_step(0u, b);
    r.t[b][0] = (uint16_t)c;
    for (int k = 1; k < 4; ++k) {
      c = crc16_step(c, 0u);
      r.t[b][k] = (uint16_t)c;
    }
  }
  return r;
}
__device__ const Crc16Tables kCrc16Tables = make_crc16_tables();

// OR `c` (len <= 31 bits, K = 64 - len) into the bit buffer so that it starts at bit s: the 64-bit window
// over words s / 32 and s / 32 + 1 takes it whole (the second OR adds zero when it does not straddle)
__device__ __forceinline__ void or_code(uint32_t* words, uint32_t s, uint32_t c, uint32_t K) {
  const unsigned long long v = (unsigned long long)c << ((K - (s & 31u)) & 63u);
  uint32_t* const w = words + (s >> 5);
  atomicOr(w, (uint32_t)(v >> 32));
  atomicOr(w + 1, (uint32_t)v);
}

// Residual::write (bitrepr.rs:550-597) for blocks whose partitions are multiples of 16 samples: rounds of
// 4096 samples, 16 per thread.  MASKED: some lane of the wave holds warm-up samples (t < order), which
// emit nothing (their slots were zeroed on load: quotient 0).
template <bool MASKED>
__device__ __forceinline__ void rice_emit16(uint32_t* words, const uint32_t (&uc)[16], uint32_t p, uint32_t pos,
                                            int nskip) {
  const uint32_t bit = 1u << p, mask = bit - 1u, L = p + 1u, K = 64u - L;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t u = uc[k];
    uint32_t code = (u & mask) | bit;
    uint32_t len = L;
    if (MASKED) {
      code = k >= nskip ? code : 0u;
      len = k >= nskip ? L : 0u;
    }
    const uint32_t s = pos + (u >> p);  // unary quotient: zeros
    or_code(words, s, code, K);
    pos = s + len;
  }
}

// Scratch of the one-barrier block scans: two alternating buffers of (wave total, wave flag) pairs, so a
// scan's writes cannot overtake the reads of the scan before the previous barrier.
struct ScanScratch {
  uint32_t total[2][4];
  uint32_t flag[2][4];
};

// One thread = one run of 16 samples inside one partition; `params` is the subframe's Rice parameter
// list in global memory (each thread needs exactly one entry per round).  Writes the 6-bit residual header
// too (it needs RICE2 = any parameter > 14, which rides on the first round's scan barrier).
__device__ __forceinline__ void rice_runs16(uint32_t* words, const int32_t* __restrict__ e, int n, uint32_t order,
                                            uint32_t porder, const uint8_t* __restrict__ params, uint32_t hdr_pos,
                                            ScanScratch* ss, uint32_t* scan_id, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t part_len = (uint32_t)n >> porder;
  const float inv_pl16 = 1.0f / (float)(part_len >> 4);
  // any parameter above 14 switches the whole residual to 5-bit parameters (bitrepr.rs:540-543, 556-562)
  uint32_t big = (uint32_t)tid < (1u << porder) ? (params[tid] > 14 ? 1u : 0u) : 0u;
  big = __any((int)big) ? 1u : 0u;
  uint32_t pbits = 4u;
  uint32_t round_base = hdr_pos + 6u;
  for (int r0 = 0; r0 < n; r0 += 16 * kPackThreads) {
    const int t_lo = r0 + 16 * tid;
    const bool live = t_lo < n;
    uint32_t uc[16];
    uint32_t p = 0, nstarts = 0;
    bool starts = false;
    int nskip = 0;
    if (live) {
      if ((reinterpret_cast<uintptr_t>(e + t_lo) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int4 v = *reinterpret_cast<const int4*>(e + t_lo + 4 * k);
          uc[4 * k + 0] = zigzag32(v.x);
          uc[4 * k + 1] = zigzag32(v.y);
          uc[4 * k + 2] = zigzag32(v.z);
          uc[4 * k + 3] = zigzag32(v.w);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) uc[k] = zigzag32(e[t_lo + k]);
      }
      // t_lo / part_len: both are multiples of 16 below 2^16, so the quotient of the sixteenths is exact in
      // f32 with half a step of margin
      const uint32_t qi = (uint32_t)(((float)(t_lo >> 4) + 0.5f) * inv_pl16);
      p = params[qi & 255u];
      starts = qi * part_len == (uint32_t)t_lo;  // the partition's parameter goes first
      nstarts = qi + (starts ? 0u : 1u);          // partitions that began in front of this run
      nskip = (int)order - t_lo;
      nskip = nskip < 0 ? 0 : (nskip > 16 ? 16 : nskip);
      if (nskip > 0) {  // (whatever the producer left in the warm-up slots: they carry no quotient)
#pragma unroll
        for (int k = 0; k < 16; ++k) uc[k] = k >= nskip ? uc[k] : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) uc[k] = 0u;
    }
    uint32_t qsum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) qsum += uc[k] >> p;
    const uint32_t my_bits = live ? qsum + (uint32_t)(16 - nskip) * (p + 1u) : 0u;
    // block scan with one barrier: DPP scan inside the wave, wave totals through LDS
    uint32_t incl = my_bits;
    incl += FLACENC_DPP(incl, 0x111, 0xF);
    incl += FLACENC_DPP(incl, 0x112, 0xF);
    incl += FLACENC_DPP(incl, 0x114, 0xF);
    incl += FLACENC_DPP(incl, 0x118, 0xF);
    incl += FLACENC_DPP(incl, 0x142, 0xA);
    incl += FLACENC_DPP(incl, 0x143, 0xC);
    const uint32_t buf = (*scan_id)++ & 1u;
    if (lane == 63) {
      ss->total[buf][wave] = incl;
      ss->flag[buf][wave] = big;
    }
    __syncthreads();
    const uint32_t t0 = ss->total[buf][0], t1 = ss->total[buf][1], t2 = ss->total[buf][2], t3 = ss->total[buf][3];
    if (r0 == 0) {
      const uint32_t rice2 = ss->flag[buf][0] | ss->flag[buf][1] | ss->flag[buf][2] | ss->flag[buf][3];
      pbits = rice2 ? 5u : 4u;
      if (tid == 192) put_bits(words, hdr_pos, (rice2 << 4) | porder, 6u);
    }
    const uint32_t wave_base = (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
    uint32_t pos = round_base + wave_base + (incl - my_bits) + pbits * nstarts;
    round_base += t0 + t1 + t2 + t3;
    if (live) {
      const uint32_t lead = starts ? pbits : 0u;
      or_code(words, pos, starts ? p : 0u, 64u - lead);
      pos += lead;
      if (__any(nskip > 0)) rice_emit16<true>(words, uc, p, pos, nskip);
      else rice_emit16<false>(words, uc, p, pos, nskip);
    }
  }
}

// What the packer needs to know about one frame, for the two kinds of decision records.
struct StereoView {  // flacenc_hip_stereo_frame_result: 2 subframes, roles L / R / M / S
  const flacenc_hip_stereo_frame_result* fr;
  const int32_t* l;
  const int32_t* r;
  uint32_t bps0;
  size_t row0;
  __device__ uint32_t nsub() const { return 2u; }
  __device__ uint32_t channel_tag() const {  // ChannelAssignment::write, bitrepr.rs:329-356
    return fr->channel_assignment == 0 ? 1u : 7u + fr->channel_assignment;
  }
  __device__ uint32_t kind(int c) const { return fr->kind[c]; }
  __device__ uint32_t bps(int c) const { return bps0 + (fr->role[c] == 3 ? 1u : 0u); }
  __device__ unsigned long long bits(int c) const { return fr->bits[fr->role[c]]; }
  __device__ int32_t dc(int c) const { return fr->dc_offset[c]; }
  __device__ const flacenc_hip_subframe_params* rec(int c) const { return &fr->lpc[c]; }
  __device__ size_t residual_row(int c) const { return row0 + (size_t)c; }
  __device__ int32_t sample(int c, int t) const {  // the role's input sample (coding.rs:476-484 for M / S)
    const uint32_t role = fr->role[c];
    const int32_t lv = l[t];
    if (role == 0u) return lv;
    const int32_t rv = r[t];
    if (role == 1u) return rv;
    return role == 2u ? ((lv + rv) >> 1) : (lv - rv);
  }
};
struct ChannelView {  // flacenc_hip_channel_result x channels: Independent(n)
  const flacenc_hip_channel_result* ch;
  const int32_t* x;
  size_t stride;
  uint32_t nch, bps0;
  size_t row0;
  __device__ uint32_t nsub() const { return nch; }
  __device__ uint32_t channel_tag() const { return nch - 1u; }
  __device__ uint32_t kind(int c) const { return ch[c].kind; }
  __device__ uint32_t bps(int) const { return bps0; }
  __device__ unsigned long long bits(int c) const { return ch[c].bits; }
  __device__ int32_t dc(int c) const { return ch[c].dc_offset; }
  __device__ const flacenc_hip_subframe_params* rec(int c) const { return &ch[c].params; }
  __device__ size_t residual_row(int c) const { return row0 + (size_t)c; }
  __device__ int32_t sample(int c, int t) const { return x[(size_t)c * stride + t]; }
};

// ALIGNED: the block size is a multiple of 4096, so every partition order up to 8 gives partitions that are
// multiples of 16 samples and the walk over arbitrary partition boundaries is not compiled in.
template <bool ALIGNED, class View>
__device__ __forceinline__ void frame_pack_body(const FramePackArgs& a, const View& view, uint32_t f) {
  extern __shared__ __attribute__((aligned(16))) uint32_t words[];
  __shared__ uint32_t scan_scratch[4];
  __shared__ ScanScratch scan_pairs;
  __shared__ uint32_t crc_part[kPackThreads / 64];
  __shared__ uint8_t rice_p[FLACENC_HIP_MAX_RICE_PARTITIONS];
  // slicing-by-4 tables: crc_tab[k][b] = CRC-16 of byte b followed by k zero bytes
  __shared__ uint16_t crc_tab[4][256];
  const int tid = threadIdx.x;
  {
    const uint2 row = *reinterpret_cast<const uint2*>(&kCrc16Tables.t[tid][0]);
    crc_tab[0][tid] = (uint16_t)row.x;
    crc_tab[1][tid] = (uint16_t)(row.x >> 16);
    crc_tab[2][tid] = (uint16_t)row.y;
    crc_tab[3][tid] = (uint16_t)(row.y >> 16);
  }
  uint32_t scan_id = 0;
  const int n = (int)a.block_size;

  for (uint32_t i = tid; i < a.lds_words / 4u; i += kPackThreads)
    reinterpret_cast<int4*>(words)[i] = make_int4(0, 0, 0, 0);  // lds_words is a multiple of 4
  __syncthreads();

  // ---- FrameHeader::write, bitrepr.rs:373-419 (fixed blocking, FrameOffset::Frame) ----
  const uint32_t frame_number = a.first_frame_number + f * a.frame_number_step;
  uint32_t utf8_len;
  {
    const uint32_t code_bits = frame_number ? 32u - (uint32_t)__builtin_clz(frame_number) : 0u;
    utf8_len = code_bits <= 7 ? 1u : 1u + (code_bits - 2u) / 5u;  // utf8like_bytesize, bitrepr.rs:157-166
  }
  const uint32_t header_bytes = 4u + utf8_len + a.extra_len + 1u;
  if (tid == 0) {
    uint8_t hdr[16];
    uint32_t hn = 0;
    const uint32_t channel_tag = view.channel_tag();
    hdr[hn++] = 0xFF;
    hdr[hn++] = 0xF8;
    hdr[hn++] = (uint8_t)(a.header_mid >> 8);
    hdr[hn++] = (uint8_t)((channel_tag << 4) | (a.header_mid & 0x0Fu));
    if (utf8_len == 1) {
      hdr[hn++] = (uint8_t)frame_number;
    } else {  // encode_to_utf8like, bitrepr.rs:108-154
      const uint32_t trailing = utf8_len - 1u;
      const uint32_t first_bits = 6u - trailing;
      const uint32_t head = (0xFFu << (8u - trailing - 1u)) & 0xFFu;  // 0xC0, 0xE0, 0xF0, 0xF8, 0xFC
      hdr[hn++] = (uint8_t)(head | ((frame_number >> (6u * trailing)) & ((1u << first_bits) - 1u)));
      for (uint32_t i = 0; i < trailing; ++i)
        hdr[hn++] = (uint8_t)(0x80u | ((frame_number >> (6u * (trailing - 1u - i))) & 0x3Fu));
    }
    for (uint32_t i = 0; i < a.extra_len; ++i) hdr[hn++] = a.extra[i];
    uint32_t crc = 0;
    for (uint32_t i = 0; i < hn; ++i) crc = crc8_byte(crc, hdr[i]);
    hdr[hn++] = (uint8_t)crc;
    for (uint32_t i = 0; i < hn; ++i) put_bits(words, 8u * i, hdr[i], 8u);
  }

  // ---- the subframes ----
  uint32_t sub_base = header_bytes * 8u;
  const int nsub = (int)view.nsub();
  for (int c = 0; c < nsub; ++c) {
    const uint32_t kind = view.kind(c);
    const uint32_t bps = view.bps(c);
    const uint32_t sub_bits = (uint32_t)view.bits(c);
    const uint32_t bps_mask = bps >= 32u ? 0xFFFFFFFFu : ((1u << bps) - 1u);
    auto sample = [&](int t) -> int32_t { return view.sample(c, t); };
    if (kind == FLACENC_HIP_KIND_CONSTANT) {  // bitrepr.rs:449-454
      if (tid == 0) put_bits(words, sub_base + 8u, (uint32_t)view.dc(c) & bps_mask, bps);
    } else if (kind == FLACENC_HIP_KIND_VERBATIM) {  // bitrepr.rs:463-470
      if (tid == 0) put_bits(words, sub_base, 0x02u, 8u);
      for (int t = tid; t < n; t += kPackThreads)
        put_bits(words, sub_base + 8u + (uint32_t)t * bps, (uint32_t)sample(t) & bps_mask, bps);
    } else {
      const flacenc_hip_subframe_params* rec = view.rec(c);
      const uint32_t order = rec->order;
      const uint32_t precision = rec->precision;
      // FixedLpc::write bitrepr.rs:479-487 / Lpc::write :501-527 up to the residual
      const uint32_t head_bits = 8u + order * bps + (kind == FLACENC_HIP_KIND_LPC ? 9u + order * precision : 0u);
      if ((uint32_t)tid < order)  // warm-up samples
        put_bits(words, sub_base + 8u + (uint32_t)tid * bps, (uint32_t)sample(tid) & bps_mask, bps);
      if (kind == FLACENC_HIP_KIND_LPC && tid >= 64 && (uint32_t)(tid - 64) < order)  // quantised coefficients
        put_bits(words, sub_base + 8u + order * bps + 9u + (uint32_t)(tid - 64) * precision,
                 (uint32_t)(int32_t)rec->coefs[tid - 64] & ((1u << precision) - 1u), precision);
      if (tid == 128) {
        put_bits(words, sub_base, kind == FLACENC_HIP_KIND_LPC ? (0x40u | ((order - 1u) << 1)) : (0x10u | (order << 1)), 8u);
        if (kind == FLACENC_HIP_KIND_LPC) {
          put_bits(words, sub_base + 8u + order * bps, precision - 1u, 4u);
          put_bits(words, sub_base + 8u + order * bps + 4u, (uint32_t)rec->shift & 31u, 5u);
        }
      }
      // Residual::write, bitrepr.rs:550-597
      const uint32_t porder = rec->rice_order;
      const uint32_t nparts = 1u << porder;
      const uint32_t part_len = (uint32_t)n >> porder;
      const int32_t* __restrict__ e = a.residual + view.residual_row(c) * a.residual_stride;
      if (ALIGNED || ((n & 15) == 0 && (part_len & 15u) == 0u)) {
        // Aligned runs: a thread owns 16 consecutive samples of a round of 4096, which lie inside one
        // partition (its length is a multiple of 16) -- one Rice parameter per thread, no per-sample
        // partition walk.  A code of p + 1 <= 31 bits at bit offset <= 31 fits a 64-bit window over two
        // buffer words: one shift and two ORs per sample, no branch.
        rice_runs16(words, e, n, order, porder, rec->rice_params, sub_base + head_bits, &scan_pairs, &scan_id, tid);
      } else if (!ALIGNED) {
        __syncthreads();  // the previous subframe is done with rice_p
        rice_p[tid] = rec->rice_params[tid];
        __syncthreads();
        uint32_t rice2 = 0;
        for (uint32_t q = tid; q < nparts; q += kPackThreads) rice2 |= rice_p[q] > 14 ? 1u : 0u;
        rice2 = __syncthreads_or((int)rice2) ? 1u : 0u;
        const uint32_t pbits = rice2 ? 5u : 4u;
        if (tid == 192) put_bits(words, sub_base + head_bits, (rice2 << 4) | porder, 6u);
        const uint32_t res_base = sub_base + head_bits + 6u;
        // contiguous slice of samples per thread; pass 1 counts its bits, pass 2 writes them.
        // Up to 16 samples per thread (blocks <= 4096) are held in registers as zig-zag codes.
        const int per = (n + kPackThreads - 1) / kPackThreads;
        const int t_lo = tid * per < n ? tid * per : n;
        const int t_hi = t_lo + per < n ? t_lo + per : n;
        const bool cached = per <= 16;
        uint32_t uc[16];
        if (cached) {
          const bool vec = per == 16 && t_hi - t_lo == 16 && ((reinterpret_cast<uintptr_t>(e + t_lo) & 15) == 0);
          if (vec) {
  #pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int4 v = *reinterpret_cast<const int4*>(e + t_lo + 4 * k);
              uc[4 * k + 0] = zigzag32(v.x);
              uc[4 * k + 1] = zigzag32(v.y);
              uc[4 * k + 2] = zigzag32(v.z);
              uc[4 * k + 3] = zigzag32(v.w);
            }
          } else {
  #pragma unroll
            for (int k = 0; k < 16; ++k) uc[k] = (t_lo + k < t_hi) ? zigzag32(e[t_lo + k]) : 0u;
          }
        }
        const uint32_t q_lo = (uint32_t)t_lo / part_len;
        // walks the slice once; `emit(t, u, p, starts)` sees every coded sample in order
        auto walk = [&](auto&& emit) {
          uint32_t q = q_lo, next = (q_lo + 1u) * part_len, p = rice_p[q_lo & 255u];
          auto step = [&](uint32_t t, uint32_t u) {
            if (t == next) {
              ++q;
              next += part_len;
              p = rice_p[q & 255u];
            }
            if (t >= order) emit(u, p, q == 0u ? t == order : t == next - part_len);  // max(warmup, offset)
          };
          if (cached) {
  #pragma unroll
            for (int k = 0; k < 16; ++k)
              if (t_lo + k < t_hi) step((uint32_t)(t_lo + k), uc[k]);
          } else {
            for (int t = t_lo; t < t_hi; ++t) step((uint32_t)t, zigzag32(e[t]));
          }
        };
        uint32_t my_bits = 0;
        walk([&](uint32_t u, uint32_t p, bool starts) { my_bits += (starts ? pbits : 0u) + (u >> p) + 1u + p; });
        uint32_t total;
        uint32_t pos = res_base + block_exclusive_scan(my_bits, scan_scratch, tid, &total);
        walk([&](uint32_t u, uint32_t p, bool starts) {
          if (starts) {
            put_bits(words, pos, p, pbits);
            pos += pbits;
          }
          pos += u >> p;  // unary quotient: zeros
          put_bits(words, pos, (u & ((1u << p) - 1u)) | (1u << p), p + 1u);
          pos += p + 1u;
        });
        (void)total;
      }
    }
    sub_base += sub_bits;
  }
  __syncthreads();

  // ---- align_to_byte + CRC-16 over everything before it (bitrepr.rs:306-316) ----
  const uint32_t body_bytes = (sub_base + 7u) >> 3;
  {
    // slices of a.crc_per bytes counted from the END (so every slice but the first is full); the
    // slice of thread t is followed by k = 255 - t slices, and x^(8 * crc_per * k) mod P comes from
    // the host-made tables crc_pow[i] = y^i, crc_pow[16 + i] = y^(16 i) with y = x^(8 * crc_per)
    const uint32_t per = a.crc_per;
    const uint32_t k_after = (uint32_t)(kPackThreads - 1 - tid);
    const uint32_t after = k_after * per;
    uint32_t crc = 0;
    if (after < body_bytes) {
      const uint32_t hi = body_bytes - after;
      const uint32_t lo = hi > per ? hi - per : 0u;
      // Four bytes per step (slicing-by-4: one level of dependent table look-ups per word instead of
      // four).  The slice ends at byte hi; it is walked in 4-byte chunks [hi - 4m, hi) from the front,
      // and bytes in front of lo are masked to zero -- with the register still 0 there, leading zero
      // bytes leave a CRC with init 0 unchanged.  A chunk is assembled from two big-endian buffer words.
      const uint32_t nchunks = (hi - lo + 3u) >> 2;
      const uint32_t sh = (hi & 3u) * 8u;  // chunk start = hi - 4 (m - j) has the byte phase of hi
      const uint32_t w_end = hi >> 2;      // word holding byte hi (its first `hi & 3` bytes are the chunk's tail)
      uint32_t prev = (w_end >= nchunks) ? words[w_end - nchunks] : 0u;
      for (uint32_t j = 0; j < nchunks; ++j) {
        const uint32_t cur = words[w_end - nchunks + j + 1u];  // (one word past body_bytes at most: inside the buffer)
        // bytes [pos, pos + 4) with pos = 4 (w_end - nchunks + j) + (hi & 3)
        uint32_t x = sh ? ((prev << sh) | (cur >> (32u - sh))) : prev;
        if (j == 0) {
          const uint32_t pos = hi - 4u * nchunks;      // may be < lo (or wrap below 0): mask those bytes
          const uint32_t skip = lo - pos;              // 0..3 leading bytes that are not part of the slice
          x = skip ? (x & (0xFFFFFFFFu >> (8u * skip))) : x;
        }
        x ^= crc << 16;
        crc = (uint32_t)crc_tab[3][x >> 24] ^ crc_tab[2][(x >> 16) & 0xFFu] ^ crc_tab[1][(x >> 8) & 0xFFu] ^
              crc_tab[0][x & 0xFFu];
        prev = cur;
      }
      crc = gf_mulmod16(crc, a.crc_pow[k_after & 15u]);
      crc = gf_mulmod16(crc, a.crc_pow[16u + (k_after >> 4)]);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) crc ^= (uint32_t)__shfl_xor((int)crc, d, 64);
    if ((tid & 63) == 0) crc_part[tid >> 6] = crc;
  }
  __syncthreads();
  const uint32_t frame_bytes = body_bytes + 2u;
  if (tid == 0) {
    const uint32_t crc = crc_part[0] ^ crc_part[1] ^ crc_part[2] ^ crc_part[3];
    put_bits(words, body_bytes * 8u, crc, 16u);
    a.out_len[f] = frame_bytes;
  }
  __syncthreads();
  // out rows are 16-byte aligned (out 16-byte aligned, stride a multiple of 16): whole int4 stores
  int4* __restrict__ dst = reinterpret_cast<int4*>(a.out + (size_t)f * a.out_stride);
  const uint32_t nquads = (frame_bytes + 15u) >> 4;
  for (uint32_t i = tid; i < nquads; i += kPackThreads) {
    int4 v = reinterpret_cast<const int4*>(words)[i];
    v.x = (int)__builtin_bswap32((uint32_t)v.x);
    v.y = (int)__builtin_bswap32((uint32_t)v.y);
    v.z = (int)__builtin_bswap32((uint32_t)v.z);
    v.w = (int)__builtin_bswap32((uint32_t)v.w);
    dst[i] = v;
  }
}

template <bool ALIGNED>
__global__ void __launch_bounds__(kPackThreads, ALIGNED ? FLACENC_PACK_OCC : 4) frame_pack_kernel(FramePackArgs a) {
  const uint32_t f = blockIdx.x;
  StereoView v;
  v.fr = a.results + f;
  v.l = a.frames + (size_t)(2u * f) * a.stride;
  v.r = v.l + a.stride;
  v.bps0 = a.bits_per_sample;
  v.row0 = (size_t)(2u * f);
  frame_pack_body<ALIGNED>(a, v, f);
}

template <bool ALIGNED>
__global__ void __launch_bounds__(kPackThreads, ALIGNED ? FLACENC_PACK_OCC : 4) channel_pack_kernel(FramePackArgs a) {
  const uint32_t f = blockIdx.x;
  ChannelView v;
  v.ch = a.chan_results + (size_t)f * a.channels;
  v.x = a.frames + (size_t)f * a.channels * a.stride;
  v.stride = a.stride;
  v.nch = a.channels;
  v.bps0 = a.bits_per_sample;
  v.row0 = (size_t)f * a.channels;
  frame_pack_body<ALIGNED>(a, v, f);
}

// Frame::count_bits / 8 (bitrepr.rs:275-287) from the decision records alone
__global__ void frame_lengths_kernel(FramePackArgs a) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.n_frames) return;
  const uint32_t frame_number = a.first_frame_number + f * a.frame_number_step;
  const uint32_t code_bits = frame_number ? 32u - (uint32_t)__builtin_clz(frame_number) : 0u;
  const uint32_t utf8_len = code_bits <= 7 ? 1u : 1u + (code_bits - 2u) / 5u;
  unsigned long long bits = 8ull * (4u + utf8_len + a.extra_len + 1u);
  if (a.chan_results) {
    for (uint32_t c = 0; c < a.channels; ++c) bits += a.chan_results[(size_t)f * a.channels + c].bits;
  } else {
    const flacenc_hip_stereo_frame_result* fr = a.results + f;
    bits += fr->bits[fr->role[0]] + fr->bits[fr->role[1]];
  }
  a.out_len[f] = (uint32_t)((bits + 7ull) >> 3) + 2u;
}

// FrameBuf::fill_le_bytes (src/source.rs:288-298) for a run of consecutive frames of one stream:
// le_bytes_to_i32s (arrayutils.rs:273-290: little-endian, sign-extended) + deinterleave
// (arrayutils.rs:248-264: channel-major rows, the unfilled rest of a row is zero)
__global__ void __launch_bounds__(256) fill_le_bytes_kernel(const uint8_t* __restrict__ bytes, uint32_t channels,
                                                            uint32_t bytes_per_sample, uint64_t total_samples,
                                                            uint32_t block_size, int32_t* __restrict__ frames,
                                                            size_t stride) {
  const uint32_t f = blockIdx.y;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= block_size) return;
  const uint64_t idx = (uint64_t)f * block_size + t;  // inter-channel sample index in the stream
  const bool filled = idx < total_samples;
  const uint8_t* src = bytes + idx * channels * bytes_per_sample;
  for (uint32_t c = 0; c < channels; ++c) {
    int32_t v = 0;
    if (filled) {
      uint32_t u = 0;
      for (uint32_t b = 0; b < bytes_per_sample; ++b) u |= (uint32_t)src[c * bytes_per_sample + b] << (8u * b);
      const uint32_t sh = 32u - 8u * bytes_per_sample;
      v = (int32_t)(u << sh) >> sh;
    }
    frames[((size_t)f * channels + c) * stride + t] = v;
  }
}

}  // namespace

hipError_t launch_fill_le_bytes(const uint8_t* bytes, uint32_t channels, uint32_t bytes_per_sample,
                                uint64_t total_samples, uint32_t n_frames, uint32_t block_size, int32_t* frames,
                                size_t stride, hipStream_t stream) {
  if (n_frames == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_le_bytes_kernel, dim3((block_size + 255) / 256, n_frames), dim3(256), 0, stream, bytes,
                     channels, bytes_per_sample, total_samples, block_size, frames, stride);
  return hipGetLastError();
}

hipError_t launch_frame_lengths(const FramePackArgs& a, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  hipLaunchKernelGGL(frame_lengths_kernel, dim3((a.n_frames + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

size_t frame_bytes_bound(uint32_t channels, uint32_t block_size, uint32_t bits_per_sample) {
  const size_t bits = static_cast<size_t>(channels) * (8 + static_cast<size_t>(block_size) * bits_per_sample);
  return 15 + (bits + 7) / 8 + 2;
}

// One workgroup per frame.  Destination-aligned dword stores; the source dword that feeds each of them
// is assembled from two aligned loads with v_alignbyte_b32, so a frame moves at dword granularity
// whatever the two byte offsets are (frames are byte-aligned in a FLAC stream, nothing more).
__global__ __launch_bounds__(256) void place_frames_kernel(const uint8_t* __restrict__ src,
                                                           const uint64_t* __restrict__ src_offsets,
                                                           const uint32_t* __restrict__ lengths,
                                                           uint8_t* __restrict__ dst,
                                                           const uint64_t* __restrict__ dst_offsets) {
  const uint32_t f = blockIdx.x;
  const uint8_t* s = src + src_offsets[f];
  uint8_t* d = dst + dst_offsets[f];
  uint32_t len = lengths[f];
  uint32_t head = static_cast<uint32_t>(-reinterpret_cast<uintptr_t>(d)) & 3u;
  if (head > len) head = len;
  if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
  s += head;
  d += head;
  len -= head;
  const uint32_t n_dwords = len >> 2;
  const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s)) & 3u;
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(s - skew);
  uint32_t* dw = reinterpret_cast<uint32_t*>(d);
  if (skew == 0) {
    for (uint32_t k = threadIdx.x; k < n_dwords; k += blockDim.x) dw[k] = sw[k];
  } else {
    // bytes [4k + skew, 4k + skew + 4) of the aligned source: both dwords hold bytes of this frame
    for (uint32_t k = threadIdx.x; k < n_dwords; k += blockDim.x)
      dw[k] = __builtin_amdgcn_alignbyte(sw[k + 1], sw[k], skew);
  }
  const uint32_t tail = len & 3u;
  if (threadIdx.x < tail) d[n_dwords * 4 + threadIdx.x] = s[n_dwords * 4 + threadIdx.x];
}

hipError_t launch_place_frames(const uint8_t* src, const uint64_t* src_offsets, const uint32_t* lengths,
                               uint8_t* dst, const uint64_t* dst_offsets, uint32_t n_frames, hipStream_t stream) {
  if (n_frames == 0) return hipSuccess;
  hipLaunchKernelGGL(place_frames_kernel, dim3(n_frames), dim3(256), 0, stream, src, src_offsets, lengths, dst,
                     dst_offsets);
  return hipGetLastError();
}

__global__ __launch_bounds__(1024) void frame_offsets_kernel(const uint32_t* __restrict__ lengths, uint32_t n,
                                                             unsigned long long src_stride,
                                                             uint64_t* __restrict__ src_offsets,
                                                             uint64_t* __restrict__ dst_offsets,
                                                             uint64_t* __restrict__ total) {
  __shared__ unsigned long long wave_sums[16];
  __shared__ unsigned long long carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024u) {
    const uint32_t f = base + tid;
    const unsigned long long v = f < n ? lengths[f] : 0ull;
    unsigned long long s = v;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(s, d, 64);
      if ((int)lane >= d) s += o;
    }
    if (lane == 63) wave_sums[wave] = s;
    __syncthreads();
    unsigned long long before = carry;
    for (uint32_t w = 0; w < wave; ++w) before += wave_sums[w];
    if (f < n) {
      src_offsets[f] = (unsigned long long)f * src_stride;
      dst_offsets[f] = before + s - v;
    }
    __syncthreads();
    if (tid == 1023) carry = before + s;
    __syncthreads();
  }
  if (tid == 0) total[0] = carry;
}

hipError_t launch_frame_offsets(const uint32_t* lengths, uint32_t n_frames, size_t src_stride, uint64_t* src_offsets,
                                uint64_t* dst_offsets, uint64_t* total, hipStream_t stream) {
  hipLaunchKernelGGL(frame_offsets_kernel, dim3(1), dim3(1024), 0, stream, lengths, n_frames,
                     static_cast<unsigned long long>(src_stride), src_offsets, dst_offsets, total);
  return hipGetLastError();
}

// ---- the ordered gather's two device steps (ParSink, src/par.rs:67-95, across GPUs) --------------------------
// Wire format of a stereo frame record: its 48 bytes of frame fields and, of each 352-byte subframe record, the
// 96 fixed bytes + the first `parts` Rice parameters (a block has at most 2^finest_order partitions, rice.rs:157-165;
// the rest of rice_params[256] is always zero).  One thread moves W bytes; thread 0 of a record also derives the
// frame's byte length (frame_lengths_kernel's arithmetic), so records -> wire + lengths is one pass over the records.
template <int W>
__global__ __launch_bounds__(256) void frame_wire_kernel(FramePackArgs a, uint32_t keep, uint8_t* __restrict__ wire,
                                                         size_t wire_stride) {
  const uint32_t units = (48u + 2u * keep) / W;
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t f = (uint32_t)(t / units), u = (uint32_t)(t % units);
  if (f >= a.n_frames) return;
  const uint32_t w = u * W;
  const uint32_t s = w < 48u + keep ? w : w - keep + 352u;  // second subframe record starts at 48 + 352
  const uint8_t* src = reinterpret_cast<const uint8_t*>(a.results + f) + s;
  uint8_t* dst = wire + (size_t)f * wire_stride + w;
  if (W == 16) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
  else if (W == 4) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
  else *dst = *src;
  if (u == 0 && a.out_len) {
    const uint32_t frame_number = a.first_frame_number + f * a.frame_number_step;
    const uint32_t code_bits = frame_number ? 32u - (uint32_t)__builtin_clz(frame_number) : 0u;
    const uint32_t utf8_len = code_bits <= 7 ? 1u : 1u + (code_bits - 2u) / 5u;
    const flacenc_hip_stereo_frame_result* fr = a.results + f;
    const unsigned long long bits = 8ull * (4u + utf8_len + a.extra_len + 1u) + fr->bits[fr->role[0]] + fr->bits[fr->role[1]];
    a.out_len[f] = (uint32_t)((bits + 7ull) >> 3) + 2u;
  }
}

hipError_t launch_frame_wire(const FramePackArgs& a, uint32_t parts, uint8_t* wire, size_t wire_stride,
                             hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  const uint32_t keep = 96u + parts;
  const bool wide = parts % 16u == 0 && wire_stride % 16u == 0 && (reinterpret_cast<uintptr_t>(wire) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.results) & 15u) == 0;
  const bool mid = parts % 4u == 0 && wire_stride % 4u == 0 && (reinterpret_cast<uintptr_t>(wire) & 3u) == 0;
  const uint32_t W = wide ? 16u : mid ? 4u : 1u;
  const uint64_t threads = (uint64_t)a.n_frames * ((48u + 2u * keep) / W);
  const dim3 grid((uint32_t)((threads + 255) / 256));
  if (W == 16) hipLaunchKernelGGL(frame_wire_kernel<16>, grid, dim3(256), 0, stream, a, keep, wire, wire_stride);
  else if (W == 4) hipLaunchKernelGGL(frame_wire_kernel<4>, grid, dim3(256), 0, stream, a, keep, wire, wire_stride);
  else hipLaunchKernelGGL(frame_wire_kernel<1>, grid, dim3(256), 0, stream, a, keep, wire, wire_stride);
  return hipGetLastError();
}

// Stream offsets from the all-gathered frame lengths.  `lengths` is the collective's output as it arrives: rank-major,
// lengths[r * per_rank + j] = stream frame j * world + r (world 1: plain stream order).  A workgroup owns 4096
// consecutive stream frames; what precedes them is, per rank, a contiguous run of that rank's row, so every workgroup
// sums its own prefix with coalesced reads (no scratch, no second launch; the redundant reads are n / 8192 rows of L2
// traffic per workgroup, ~1 MB at 2 M frames) and scans its tile through the wave scans + one LDS step.
__global__ __launch_bounds__(1024) void stream_offsets_kernel(const uint32_t* __restrict__ lengths, uint32_t n,
                                                              uint32_t world, uint32_t per_rank,
                                                              unsigned long long header,
                                                              uint32_t* __restrict__ lengths_stream,
                                                              uint64_t* __restrict__ offsets,
                                                              uint64_t* __restrict__ total) {
  __shared__ unsigned long long wave_pre[16], wave_sums[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t base = blockIdx.x * 4096u;
  unsigned long long pre = 0;
  for (uint32_t r = 0; r < world && r < base; ++r) {
    uint32_t cnt = (base - r + world - 1u) / world;  // frames j * world + r below base
    if (cnt > per_rank) cnt = per_rank;
    const uint32_t* row = lengths + (size_t)r * per_rank;
    for (uint32_t j = tid; j < cnt; j += 1024u) pre += row[j];
  }
  uint32_t v[4];
  unsigned long long mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t f = base + 4u * tid + k;
    v[k] = f < n ? lengths[(size_t)(f % world) * per_rank + f / world] : 0u;
    mine += v[k];
  }
  unsigned long long s = mine;  // inclusive scan inside the wave; `pre` is only reduced
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(s, d, 64);
    if ((int)lane >= d) s += o;
    pre += __shfl_xor(pre, d, 64);
  }
  if (lane == 63) {
    wave_sums[wave] = s;
    wave_pre[wave] = pre;
  }
  __syncthreads();
  unsigned long long before = header;
  for (uint32_t w = 0; w < 16; ++w) before += wave_pre[w] + (w < wave ? wave_sums[w] : 0ull);
  unsigned long long off = before + s - mine;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t f = base + 4u * tid + k;
    if (f < n) {
      offsets[f] = off;
      if (lengths_stream) lengths_stream[f] = v[k];
    }
    off += v[k];
  }
  if (tid == 1023u && base + 4096u >= n) total[0] = off;  // the workgroup that holds the last frame (or n = 0)
}

hipError_t launch_stream_offsets(const uint32_t* lengths, uint32_t n_frames, uint32_t world, uint32_t per_rank,
                                 uint64_t header_bytes, uint32_t* lengths_stream, uint64_t* offsets, uint64_t* total,
                                 hipStream_t stream) {
  const uint32_t grid = n_frames ? (n_frames + 4095u) / 4096u : 1u;
  hipLaunchKernelGGL(stream_offsets_kernel, dim3(grid), dim3(1024), 0, stream, lengths, n_frames, world, per_rank,
                     static_cast<unsigned long long>(header_bytes), lengths_stream, offsets, total);
  return hipGetLastError();
}

size_t stereo_frame_bytes_bound(uint32_t block_size, uint32_t bits_per_sample) {
  // header <= 4 + 6 (frame number < 2^31) + 2 + 2 + 1, two subframes of at most Verbatim size
  // (encode_subframe never keeps anything larger, coding.rs:413-416), CRC-16
  const size_t bits = 2 * 8 + static_cast<size_t>(block_size) * (2 * bits_per_sample + 1);
  return 15 + (bits + 7) / 8 + 2;
}

hipError_t launch_frame_pack(const FramePackArgs& a, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  const size_t smem = static_cast<size_t>(a.lds_words) * 4;
  static DynamicLdsOptIn opt_in[4];  // per kernel, per device inside
  const bool aligned = a.block_size % 4096u == 0u;
  auto go = [&](auto kernel, DynamicLdsOptIn& opt) -> hipError_t {
    if (hipError_t err = opt.ensure(reinterpret_cast<const void*>(kernel), smem); err != hipSuccess) return err;
    hipLaunchKernelGGL(kernel, dim3(a.n_frames), dim3(kPackThreads), smem, stream, a);
    return hipSuccess;
  };
  hipError_t err;
  if (a.chan_results) err = aligned ? go(channel_pack_kernel<true>, opt_in[0]) : go(channel_pack_kernel<false>, opt_in[1]);
  else err = aligned ? go(frame_pack_kernel<true>, opt_in[2]) : go(frame_pack_kernel<false>, opt_in[3]);
  if (err != hipSuccess) return err;
  return hipGetLastError();
}

}  // namespace flacenc_hip
