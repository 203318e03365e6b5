// flacenc.hpp -- C++ host-side mirror of the flacenc-rs API surface around the GPU path.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the host side above the
// C ABI (include/flacenc_hip.h) is written in C++ with the reference's names, argument meaning
// and error behaviour:
//
//   flacenc::config::{Encoder, StereoCoding, SubFrameCoding, Qlpc, Prc, Fixed, OrderSel, Window}
//                                                        src/config.rs:80-432
//   flacenc::error::{VerifyError, EncodeError}           src/error.rs:178, 458
//   flacenc::source::{Source, FrameBuf, MemSource}       src/source.rs:445, 115, 543
//   flacenc::component::{QuantizedParameters, Residual, Lpc, Constant, Verbatim, SubFrame,
//                        ChannelAssignment, Frame, Stream, StreamInfo}   src/component/datatype.rs
//   flacenc::encode_with_fixed_block_size                src/coding.rs:645 (pub, lib.rs:162)
//   Decode (feature "decode")                            src/component/decode.rs
//
// The controller logic (`encode_subframe` src/coding.rs:384-418, `try_stereo_coding`
// :469-527) runs on the host exactly as in the reference; every `estimated_qlpc` call is served
// by the GPU through flacenc_hip_*_qlpc_batch.  There is no CPU implementation of the path here.
// Not mirrored (out of scope): bit writer, container metadata, MD5, serde.
// Fixed-LPC candidate (`fixed_lpc`, src/coding.rs:298): fused into the on-GPU controller where that
// exists (flacenc_hip_encode_stereo_frames: 2 channels, block 4096, lpc_order <= 12,
// ApproxEnt.partitions a power of two), served by flacenc_hip_fixed_lpc_batch everywhere else.
#ifndef FLACENC_HOST_FLACENC_HPP_
#define FLACENC_HOST_FLACENC_HPP_

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <variant>
#include <vector>

#include "flacenc_hip.h"

namespace flacenc {

// ---------------------------------------------------------------- constant.rs ----
namespace constant {
constexpr size_t DEFAULT_BLOCK_SIZE = 4096;            // src/constant.rs:32
constexpr size_t MIN_BLOCK_SIZE = 32;                  // :45
constexpr size_t MAX_BLOCK_SIZE = 32767;               // :57
constexpr size_t MIN_BLOCK_SIZE_FOR_PREDICTION = 64;   // :51
constexpr size_t MIN_BITS_PER_SAMPLE = 8;              // :38
constexpr size_t MAX_BITS_PER_SAMPLE = 24;             // :54
constexpr size_t MAX_CHANNELS = 8;                     // :60
namespace qlpc {
constexpr size_t DEFAULT_ORDER = 10;       // :109
constexpr size_t DEFAULT_PRECISION = 15;   // :112
constexpr float DEFAULT_TUKEY_ALPHA = 0.4f;  // :115
constexpr size_t MAX_ORDER = 24;           // :118
constexpr size_t MAX_PRECISION = 15;
}  // namespace qlpc
namespace rice {
constexpr size_t MAX_RICE_PARAMETER = 30;  // :143
}
}  // namespace constant

// ------------------------------------------------------------------- error.rs ----
namespace error {
// VerifyError, src/error.rs:178: component path + reason
struct VerifyError : std::runtime_error {
  std::string component;
  VerifyError(std::string comp, const std::string& reason)
      : std::runtime_error("verification error: `" + comp + "` is not valid. reason: " + reason),
        component(std::move(comp)) {}
  VerifyError within(const std::string& outer) const {  // src/error.rs:216
    const std::string what_s = what();
    const size_t pos = what_s.find("reason: ");
    return VerifyError(outer + "." + component, pos == std::string::npos ? what_s : what_s.substr(pos + 8));
  }
};
// EncodeError, src/error.rs:458: Source | Config
struct EncodeError : std::runtime_error {
  enum Kind { Source, Config, Device } kind;
  EncodeError(Kind k, const std::string& msg) : std::runtime_error(msg), kind(k) {}
};
}  // namespace error

// ------------------------------------------------------------------ config.rs ----
namespace config {
struct Window {  // src/config.rs:344-359
  enum Type { Rectangle, Tukey } type = Tukey;
  float alpha = constant::qlpc::DEFAULT_TUKEY_ALPHA;
  void verify() const {  // :371-387
    if (type == Tukey && !(alpha >= 0.0f && alpha <= 1.0f))
      throw error::VerifyError("tukey.alpha", "alpha must be in range between 0 and 1");
  }
};
struct OrderSel {  // :400-409
  enum Type { BitCount, ApproxEnt } type = ApproxEnt;
  size_t partitions = 16;
};
struct Prc {  // :211-214
  size_t max_parameter = constant::rice::MAX_RICE_PARAMETER;
  void verify() const {
    if (max_parameter > constant::rice::MAX_RICE_PARAMETER)
      throw error::VerifyError("max_parameter", "must be in range ..=30");
  }
};
struct Fixed {  // :236-244
  size_t max_order = 4;
  OrderSel order_sel;
  void verify() const {  // :246-255, :419-431
    if (max_order > 4) throw error::VerifyError("max_order", "must be in range ..=4");
    if (order_sel.type == OrderSel::ApproxEnt && (order_sel.partitions < 1 || order_sel.partitions > 64))
      throw error::VerifyError("order_sel.ApproxEnt.partitions", "must be in range 1..=64");
  }
};
struct Qlpc {  // :271-288
  size_t lpc_order = constant::qlpc::DEFAULT_ORDER;
  size_t quant_precision = constant::qlpc::DEFAULT_PRECISION;
  bool use_direct_mse = false;
  size_t mae_optimization_steps = 0;
  Window window;
  void verify() const {  // :302-326
    if (lpc_order < 1 || lpc_order > constant::qlpc::MAX_ORDER)
      throw error::VerifyError("lpc_order", "must be in range 1..=24");
    if (quant_precision < 1 || quant_precision > constant::qlpc::MAX_PRECISION)
      throw error::VerifyError("quant_precision", "must be in range 1..=15");
    // (the reference rejects these two outside its `experimental` cargo feature, :310-320; this mirror follows
    // the experimental build -- the GPU library implements both estimators, include/flacenc_hip.h)
    if (mae_optimization_steps > FLACENC_HIP_MAX_MAE_STEPS)
      throw error::VerifyError("mae_optimization_steps", "must be at most 64 here");
    try {
      window.verify();
    } catch (const error::VerifyError& e) {
      throw e.within("window");
    }
  }
};
struct SubFrameCoding {  // :167-183
  bool use_constant = true;
  bool use_fixed = true;
  bool use_lpc = true;
  Fixed fixed;
  Qlpc qlpc;
  Prc prc;
  void verify() const {  // :198-204
    try {
      fixed.verify();
    } catch (const error::VerifyError& e) {
      throw e.within("fixed");
    }
    try {
      qlpc.verify();
    } catch (const error::VerifyError& e) {
      throw e.within("qlpc");
    }
    try {
      prc.verify();
    } catch (const error::VerifyError& e) {
      throw e.within("prc");
    }
  }
};
struct StereoCoding {  // :137-144
  bool use_leftside = true;
  bool use_rightside = true;
  bool use_midside = true;
};
struct Encoder {  // :85-99
  size_t block_size = constant::DEFAULT_BLOCK_SIZE;
  bool multithread = true;
  StereoCoding stereo_coding;
  SubFrameCoding subframe_coding;
  void verify() const {  // :114-130
    if (block_size < constant::MIN_BLOCK_SIZE || block_size > constant::MAX_BLOCK_SIZE)
      throw error::VerifyError("block_size", "must be in range 32..=32767");
    try {
      subframe_coding.verify();
    } catch (const error::VerifyError& e) {
      throw e.within("subframe_coding");
    }
  }
};
}  // namespace config

// ------------------------------------------------------------------ source.rs ----
namespace source {
// FrameBuf, src/source.rs:115-127: channel-major i32, channel c at [c*size, c*size + filled)
class FrameBuf {
 public:
  FrameBuf(size_t channels, size_t size) : samples_(channels * size, 0), channels_(channels), size_(size) {}
  size_t size() const { return size_; }
  size_t channels() const { return channels_; }
  size_t filled_size() const { return filled_; }
  const int32_t* channel_slice(size_t ch) const { return samples_.data() + ch * size_; }  // :251-253
  int32_t* channel_slice_mut(size_t ch) { return samples_.data() + ch * size_; }
  const int32_t* raw() const { return samples_.data(); }
  // Fill::fill_interleaved, src/source.rs:42-70
  void fill_interleaved(const int32_t* interleaved, size_t n_samples_total) {
    const size_t per_ch = n_samples_total / channels_;
    for (size_t c = 0; c < channels_; ++c) {
      int32_t* dst = channel_slice_mut(c);
      for (size_t t = 0; t < per_ch; ++t) dst[t] = interleaved[t * channels_ + c];
      for (size_t t = per_ch; t < size_; ++t) dst[t] = 0;
    }
    filled_ = per_ch;
  }
  // verify_samples, src/source.rs:262-275
  void verify_samples(size_t bits_per_sample) const {
    const int32_t lo = -(1 << (bits_per_sample - 1)), hi = (1 << (bits_per_sample - 1)) - 1;
    for (size_t c = 0; c < channels_; ++c)
      for (size_t t = 0; t < filled_; ++t) {
        const int32_t v = channel_slice(c)[t];
        if (v < lo || v > hi) throw error::VerifyError("samples", "sample out of range for bits_per_sample");
      }
  }

 private:
  std::vector<int32_t> samples_;
  size_t channels_, size_, filled_ = 0;
};

// Source, src/source.rs:445-480
class Source {
 public:
  virtual ~Source() = default;
  virtual size_t channels() const = 0;
  virtual size_t bits_per_sample() const = 0;
  virtual size_t sample_rate() const = 0;
  // reads up to block_size inter-channel samples into dest; returns the count read (0 = end)
  virtual size_t read_samples(size_t block_size, FrameBuf& dest) = 0;
};

// MemSource, src/source.rs:543-600
class MemSource : public Source {
 public:
  static MemSource from_samples(const std::vector<int32_t>& interleaved, size_t channels,
                                size_t bits_per_sample, size_t sample_rate) {
    MemSource s;
    s.samples_ = interleaved;
    s.channels_ = channels;
    s.bps_ = bits_per_sample;
    s.rate_ = sample_rate;
    return s;
  }
  size_t channels() const override { return channels_; }
  size_t bits_per_sample() const override { return bps_; }
  size_t sample_rate() const override { return rate_; }
  size_t read_samples(size_t block_size, FrameBuf& dest) override {
    const size_t begin = pos_ * channels_;
    const size_t end = std::min(samples_.size(), begin + block_size * channels_);
    if (end <= begin) return 0;
    dest.fill_interleaved(samples_.data() + begin, end - begin);
    const size_t n = (end - begin) / channels_;
    pos_ += n;
    return n;
  }
  size_t len_hint() const { return samples_.size() / channels_; }

 private:
  std::vector<int32_t> samples_;
  size_t channels_ = 1, bps_ = 16, rate_ = 44100, pos_ = 0;
};
}  // namespace source

// ------------------------------------------------------------------ component ----
namespace component {
// rice::encode_signbit / decode_signbit, src/rice.rs:169-187
inline uint32_t encode_signbit(int32_t v) { return (static_cast<uint32_t>(v) << 1) ^ static_cast<uint32_t>(v >> 31); }
inline int32_t decode_signbit(uint32_t v) { return static_cast<int32_t>(v >> 1) ^ -static_cast<int32_t>(v & 1u); }

// QuantizedParameters, src/component/datatype.rs:2164-2170
struct QuantizedParameters {
  int16_t coefs[32] = {0};
  size_t order = 0;
  int8_t shift = 0;
  size_t precision = 0;
};

// Residual, src/component/datatype.rs:2269-2284 (quotients/remainders derived on demand,
// src/coding.rs:58-62, 140-170)
struct Residual {
  uint8_t partition_order = 0;
  size_t block_size = 0;
  size_t warmup_length = 0;
  std::vector<uint8_t> rice_params;
  std::vector<int32_t> errors;  // first warmup_length slots are zero
  uint64_t sum_quotients = 0;
  uint64_t sum_rice_params = 0;
  std::pair<uint32_t, uint32_t> quotient_and_remainder(size_t t) const {
    if (t < warmup_length) return {0u, 0u};
    const uint8_t p = rice_params[t / (block_size >> partition_order)];
    const uint32_t u = encode_signbit(errors[t]);
    return {u >> p, u & ((1u << p) - 1u)};
  }
  // BitRepr::count_bits, src/component/bitrepr.rs:533-544
  size_t count_bits() const {
    const size_t nparts = size_t(1) << partition_order;
    bool rice2 = false;
    for (size_t i = 0; i < nparts; ++i) rice2 |= rice_params[i] > 14;
    return 2 + 4 + nparts * (rice2 ? 5 : 4) + (sum_quotients + block_size - warmup_length) +
           (sum_rice_params * (block_size >> partition_order) - warmup_length * rice_params[0]);
  }
  // Decode::copy_signal, src/component/decode.rs:226-237
  void copy_signal(int32_t* dest) const {
    for (size_t t = 0; t < block_size; ++t) {
      auto qr = quotient_and_remainder(t);
      const uint8_t p = rice_params[t / (block_size >> partition_order)];
      dest[t] = decode_signbit((qr.first << p) + qr.second);
    }
  }
};

struct Constant {  // datatype.rs:1820
  size_t block_size;
  int32_t dc_offset;
  uint8_t bits_per_sample;
  size_t count_bits() const { return 8 + bits_per_sample; }  // bitrepr.rs:445
};
struct Verbatim {  // datatype.rs:1896
  std::vector<int32_t> samples;
  uint8_t bits_per_sample;
  static size_t count_bits_from_metadata(size_t n, size_t bps) { return 8 + n * bps; }  // datatype.rs:1944
  size_t count_bits() const { return count_bits_from_metadata(samples.size(), bits_per_sample); }
};
struct Lpc {  // datatype.rs:2057-2062
  QuantizedParameters parameters;
  std::vector<int32_t> warm_up;
  Residual residual;
  uint8_t bits_per_sample;
  size_t order() const { return parameters.order; }
  size_t count_bits() const {  // bitrepr.rs:492-499
    return 8 + size_t(bits_per_sample) * order() + 4 + 5 + parameters.precision * order() + residual.count_bits();
  }
};
struct FixedLpc {  // datatype.rs:1960-1964
  std::vector<int32_t> warm_up;
  Residual residual;
  uint8_t bits_per_sample;
  size_t order() const { return warm_up.size(); }
  size_t count_bits() const {  // bitrepr.rs:473-477
    return 8 + size_t(bits_per_sample) * order() + residual.count_bits();
  }
};
using SubFrame = std::variant<Constant, Verbatim, FixedLpc, Lpc>;  // datatype.rs:1782
inline size_t count_bits(const SubFrame& sf) {
  return std::visit([](const auto& c) { return c.count_bits(); }, sf);
}

// Decode for SubFrame, src/component/decode.rs:116-218
inline std::vector<int32_t> decode(const SubFrame& sf) {
  if (const auto* c = std::get_if<Constant>(&sf)) return std::vector<int32_t>(c->block_size, c->dc_offset);
  if (const auto* v = std::get_if<Verbatim>(&sf)) return v->samples;
  if (const auto* fx = std::get_if<FixedLpc>(&sf)) {  // decode.rs:179-201
    static const int32_t kFixedCoefs[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
    std::vector<int32_t> dest(fx->residual.block_size);
    fx->residual.copy_signal(dest.data());
    const size_t order = fx->order();
    for (size_t t = 0; t < order; ++t) dest[t] = fx->warm_up[t];
    for (size_t t = order; t < dest.size(); ++t) {
      int64_t pred = 0;
      for (size_t tau = 0; tau < order; ++tau) pred += int64_t(kFixedCoefs[order][tau]) * int64_t(dest[t - 1 - tau]);
      dest[t] = int32_t(uint32_t(dest[t]) + uint32_t(int32_t(pred)));
    }
    return dest;
  }
  const Lpc& l = std::get<Lpc>(sf);
  std::vector<int32_t> dest(l.residual.block_size);
  l.residual.copy_signal(dest.data());
  for (size_t t = 0; t < l.warm_up.size(); ++t) dest[t] = l.warm_up[t];
  for (size_t t = l.warm_up.size(); t < dest.size(); ++t) {  // decode_lpc, decode.rs:159-177
    int64_t pred = 0;
    for (size_t tau = 0; tau < l.order(); ++tau) pred += int64_t(l.parameters.coefs[tau]) * int64_t(dest[t - 1 - tau]);
    dest[t] = int32_t(uint32_t(dest[t]) + uint32_t(int32_t(pred >> l.parameters.shift)));
  }
  return dest;
}

enum class ChannelAssignment { Independent, LeftSide, RightSide, MidSide };  // datatype.rs:1083

struct Frame {  // datatype.rs:820
  uint32_t frame_number = 0;
  size_t block_size = 0;
  ChannelAssignment channel_assignment = ChannelAssignment::Independent;
  std::vector<SubFrame> subframes;
  // Frame::precomputed_bitstream (datatype.rs:820-830): the frame as it goes into the stream;
  // filled by the GPU bit writer (flacenc_hip_pack_stereo_frames) where that path is taken
  std::vector<uint8_t> precomputed_bitstream;
  size_t count_subframe_bits() const {
    size_t b = 0;
    for (const auto& s : subframes) b += count_bits(s);
    return b;
  }
  // Decode for Frame, src/component/decode.rs:61-113: returns channel-major samples
  std::vector<std::vector<int32_t>> decode_channels() const {
    std::vector<std::vector<int32_t>> ch;
    for (const auto& s : subframes) ch.push_back(decode(s));
    if (channel_assignment == ChannelAssignment::LeftSide) {
      for (size_t t = 0; t < block_size; ++t) ch[1][t] = ch[0][t] - ch[1][t];
    } else if (channel_assignment == ChannelAssignment::RightSide) {
      for (size_t t = 0; t < block_size; ++t) ch[0][t] += ch[1][t];
    } else if (channel_assignment == ChannelAssignment::MidSide) {
      for (size_t t = 0; t < block_size; ++t) {
        const int32_t s = ch[1][t];
        const int32_t m = int32_t(uint32_t(ch[0][t]) << 1) + (s & 1);
        ch[0][t] = (m + s) >> 1;
        ch[1][t] = (m - s) >> 1;
      }
    }
    return ch;
  }
};

struct StreamInfo {  // datatype.rs:435-445, defaults :476-479
  size_t sample_rate = 0, channels = 0, bits_per_sample = 0;
  size_t min_block_size = 0xFFFF, max_block_size = 0;
  size_t min_frame_size = 0xFFFFFFFFu, max_frame_size = 0;
  uint64_t total_samples = 0;
  uint8_t md5_digest[16] = {0};  // all zero = "not computed" (MD5 is host work outside this mirror)
  // update_frame_info, datatype.rs:514-523 (total_samples is counted by the encoder loop here)
  void update_frame_info(const Frame& f) {
    min_block_size = std::min(min_block_size, f.block_size);
    max_block_size = std::max(max_block_size, f.block_size);
    if (!f.precomputed_bitstream.empty()) {
      min_frame_size = std::min(min_frame_size, f.precomputed_bitstream.size());
      max_frame_size = std::max(max_frame_size, f.precomputed_bitstream.size());
    }
  }
};
struct Stream {  // datatype.rs:65
  StreamInfo stream_info;
  std::vector<Frame> frames;
  void add_frame(Frame f) {  // datatype.rs:184-187
    stream_info.update_frame_info(f);
    frames.push_back(std::move(f));
  }
  // BitRepr for Stream::write (bitrepr.rs:185-196): "fLaC", the STREAMINFO block (:207-216,
  // :246-270, last-block flag set as Stream::write does for a stream without further metadata),
  // then every frame's precomputed bitstream (:290-293).  Frames that did not go through the GPU
  // bit writer have none: the host-side bit writer is outside this mirror.
  std::vector<uint8_t> to_bytes() const {
    std::vector<uint8_t> out = {0x66, 0x4c, 0x61, 0x43, 0x80, 0x00, 0x00, 34};
    auto be = [&](uint64_t v, int bytes) {
      for (int i = bytes - 1; i >= 0; --i) out.push_back(uint8_t(v >> (8 * i)));
    };
    be(stream_info.min_block_size, 2);
    be(stream_info.max_block_size, 2);
    be(stream_info.min_frame_size & 0xFFFFFF, 3);
    be(stream_info.max_frame_size & 0xFFFFFF, 3);
    be((uint64_t(stream_info.sample_rate) << 44) | (uint64_t(stream_info.channels - 1) << 41) |
           (uint64_t(stream_info.bits_per_sample - 1) << 36) | (stream_info.total_samples & 0xFFFFFFFFFull),
       8);
    out.insert(out.end(), stream_info.md5_digest, stream_info.md5_digest + 16);
    for (const Frame& f : frames) {
      if (f.precomputed_bitstream.empty())
        throw std::runtime_error("Stream::to_bytes: a frame has no precomputed bitstream (only 2-channel "
                                 "streams go through the GPU bit writer)");
      out.insert(out.end(), f.precomputed_bitstream.begin(), f.precomputed_bitstream.end());
    }
    return out;
  }
};
}  // namespace component

// ----------------------------------------------------------------- the GPU side ----
// RAII owner of a flacenc_hip_handle (the role of the reference's per-thread scratch)
class HipContext {
 public:
  explicit HipContext(int device_id = 0) {
    // (flacenc_hip_qlpc_config is embedded by value in the frame config: a library of another ABI revision would read
    // every field behind it shifted)
    if (flacenc_hip_abi_version() != FLACENC_HIP_ABI_VERSION)
      throw error::EncodeError(error::EncodeError::Device, "libflacenc_hip.so has another ABI revision than include/flacenc_hip.h");
    const int rc = flacenc_hip_create(&h_, device_id);
    if (rc != FLACENC_HIP_OK)
      throw error::EncodeError(error::EncodeError::Device, "flacenc_hip_create failed (no usable GPU)");
  }
  ~HipContext() {
    if (h_) flacenc_hip_destroy(h_);
  }
  HipContext(const HipContext&) = delete;
  HipContext& operator=(const HipContext&) = delete;
  HipContext(HipContext&& o) noexcept : h_(o.h_), sum_order_(o.sum_order_) { o.h_ = nullptr; }  // std::vector<HipContext>: one per GPU
  HipContext& operator=(HipContext&& o) noexcept {
    if (this != &o) {
      if (h_) flacenc_hip_destroy(h_);
      h_ = o.h_;
      sum_order_ = o.sum_order_;
      o.h_ = nullptr;
    }
    return *this;
  }
  flacenc_hip_handle* get() const { return h_; }

  // Which build of the crate the floating-point sums of the path reproduce bit for bit: the stable build
  // (one mul_add chain per lag, src/lpc.rs:533-548; one f32 chain per estimator partition,
  // src/arrayutils.rs:435-506), the `simd-nightly` build (src/lpc.rs:439-531; LPC orders up to 15 -- above,
  // the unflagged order is used), or the library's unflagged mode (Canonical: since ABI 6 the stable build's quantised LPC
  // parameters on every shape -- certified on blocks of 4096 / 4608, the stable build's own chains elsewhere -- with the
  // fixed-LPC selector's sums exact integers rather than that build's f32 chains, which only matters above 16 bits).
  enum class SumOrder { Canonical, Stable, SimdNightly };
  void set_sum_order(SumOrder o) { sum_order_ = o; }
  SumOrder sum_order() const { return sum_order_; }
  uint32_t sum_order_flags(size_t lpc_order) const {
    // (the mirror consumes integers only: certified shapes keep their own order, INTEGER_PARITY_ONLY)
    if (sum_order_ == SumOrder::Stable) return FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER | FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY;
    if (sum_order_ == SumOrder::SimdNightly && lpc_order <= 15) return FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER;
    return 0u;
  }

 private:
  flacenc_hip_handle* h_ = nullptr;
  SumOrder sum_order_ = SumOrder::Canonical;
};

// Staging memory for the host-pointer entry points: page-locked (flacenc_hip_host_alloc), so that the
// library's copies are plain DMA transfers instead of going through the driver's bounce buffers -- the
// role of the reference's reused FrameBuf pool (src/par.rs:364-368).  Falls back to ordinary memory if
// page-locked memory cannot be had.
template <class T>
class PinnedBuffer {
 public:
  explicit PinnedBuffer(size_t count) : count_(count) {
    if (count == 0) return;
    ptr_ = static_cast<T*>(flacenc_hip_host_alloc(count * sizeof(T)));
    if (ptr_ == nullptr) {
      fallback_.resize(count);
      ptr_ = fallback_.data();
    }
  }
  ~PinnedBuffer() {
    if (ptr_ != nullptr && fallback_.empty()) flacenc_hip_host_free(ptr_);
  }
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  T* data() { return ptr_; }
  const T* data() const { return ptr_; }
  T& operator[](size_t i) { return ptr_[i]; }
  const T& operator[](size_t i) const { return ptr_[i]; }
  T* begin() { return ptr_; }
  size_t size() const { return count_; }

 private:
  T* ptr_ = nullptr;
  size_t count_ = 0;
  std::vector<T> fallback_;
};

namespace detail {
inline flacenc_hip_qlpc_config to_abi(const config::SubFrameCoding& c) {
  flacenc_hip_qlpc_config o{};
  o.lpc_order = static_cast<uint32_t>(c.qlpc.lpc_order);
  o.quant_precision = static_cast<uint32_t>(c.qlpc.quant_precision);
  o.window_type = c.qlpc.window.type == config::Window::Rectangle ? FLACENC_HIP_WINDOW_RECTANGLE
                                                                  : FLACENC_HIP_WINDOW_TUKEY;
  o.tukey_alpha = c.qlpc.window.alpha;
  o.max_rice_parameter = static_cast<uint32_t>(c.prc.max_parameter);
  o.flags = 0;
  o.use_direct_mse = c.qlpc.use_direct_mse ? 1u : 0u;
  o.mae_optimization_steps = static_cast<uint32_t>(c.qlpc.mae_optimization_steps);
  return o;
}

inline bool is_constant(const int32_t* s, size_t n) {  // arrayutils::is_constant, arrayutils.rs:382
  for (size_t t = 1; t < n; ++t)
    if (s[t] != s[0]) return false;
  return true;
}

// SubFrame::Lpc from one GPU record + residual row (Lpc::from_parts, coding.rs:373-380)
inline component::Lpc make_lpc(const flacenc_hip_subframe_params& p, const int32_t* residual,
                               const int32_t* signal, size_t n, uint8_t bps) {
  component::Lpc l;
  std::memcpy(l.parameters.coefs, p.coefs, sizeof(p.coefs));
  l.parameters.order = p.order;
  l.parameters.shift = p.shift;
  l.parameters.precision = p.precision;
  l.warm_up.assign(signal, signal + p.order);
  l.residual.partition_order = p.rice_order;
  l.residual.block_size = n;
  l.residual.warmup_length = p.order;
  l.residual.rice_params.assign(p.rice_params, p.rice_params + (size_t(1) << p.rice_order));
  l.residual.errors.assign(residual, residual + n);
  l.residual.sum_quotients = p.sum_quotients;
  l.residual.sum_rice_params = 0;
  for (uint8_t v : l.residual.rice_params) l.residual.sum_rice_params += v;
  l.bits_per_sample = bps;
  return l;
}

// SubFrame::FixedLpc from one GPU record + error-signal row (FixedLpc::from_parts, coding.rs:321-328)
inline component::FixedLpc make_fixed_lpc(const flacenc_hip_subframe_params& p, const int32_t* residual,
                                          const int32_t* signal, size_t n, uint8_t bps) {
  component::FixedLpc fx;
  fx.warm_up.assign(signal, signal + p.order);
  fx.residual.partition_order = p.rice_order;
  fx.residual.block_size = n;
  fx.residual.warmup_length = p.order;
  fx.residual.rice_params.assign(p.rice_params, p.rice_params + (size_t(1) << p.rice_order));
  fx.residual.errors.assign(residual, residual + n);
  fx.residual.sum_quotients = p.sum_quotients;
  fx.residual.sum_rice_params = 0;
  for (uint8_t v : fx.residual.rice_params) fx.residual.sum_rice_params += v;
  fx.bits_per_sample = bps;
  return fx;
}

// encode_subframe, src/coding.rs:384-418, with the LPC and fixed-LPC candidates supplied by the GPU
// (`fixed_key` = the order selector's key of the fixed candidate: fixed_lpc returned Some iff it is
// below verbatim_bits, coding.rs:262, :284)
inline component::SubFrame encode_subframe(const config::SubFrameCoding& cfg, const int32_t* samples, size_t n,
                                           uint8_t bps, const flacenc_hip_subframe_params* lpc_rec,
                                           const int32_t* lpc_residual,
                                           const flacenc_hip_subframe_params* fixed_rec = nullptr,
                                           const int32_t* fixed_residual = nullptr, uint64_t fixed_key = 0) {
  if (cfg.use_constant && is_constant(samples, n)) return component::Constant{n, samples[0], bps};
  const size_t verbatim_bits = component::Verbatim::count_bits_from_metadata(n, bps);
  const bool too_short = n < constant::MIN_BLOCK_SIZE_FOR_PREDICTION;
  const bool have_fixed = !too_short && cfg.use_fixed && fixed_rec != nullptr && fixed_key < verbatim_bits;
  size_t baseline_bits = verbatim_bits;
  if (have_fixed) baseline_bits = std::min<size_t>(verbatim_bits, fixed_rec->subframe_bits);
  if (!too_short && cfg.use_lpc && lpc_rec != nullptr) {
    if (lpc_rec->status != FLACENC_HIP_SUBFRAME_OK)
      throw std::runtime_error("LPC analysis reported a non-finite result (the reference panics here, lpc.rs:786)");
    component::Lpc cand = make_lpc(*lpc_rec, lpc_residual, samples, n, bps);
    if (cand.count_bits() < baseline_bits) return cand;  // then also < verbatim_bits
  }
  if (have_fixed) {
    component::FixedLpc cand = make_fixed_lpc(*fixed_rec, fixed_residual, samples, n, bps);
    if (cand.count_bits() < verbatim_bits) return cand;
  }
  return component::Verbatim{std::vector<int32_t>(samples, samples + n), bps};
}
}  // namespace detail

namespace detail {
// encode_fixed_size_frame (src/coding.rs:581-606) for a run of frames on ONE GPU: `bufs[i]` is stream
// frame first_number + i*number_step (number_step > 1 when frames were dealt round-robin over several
// GPUs).  Frames of equal filled size (all but the stream's last) are analysed in one batch; the
// reference's per-frame controller runs on the device where the ABI offers it and on the host otherwise.
inline std::vector<component::Frame> encode_frame_run(const config::Encoder& config,
                                                      const std::vector<const source::FrameBuf*>& bufs,
                                                      size_t first_number, size_t number_step, size_t nch,
                                                      size_t bps, size_t sample_rate, HipContext& gpu) {
  const config::SubFrameCoding& sc = config.subframe_coding;
  std::vector<component::Frame> out;
  out.reserve(bufs.size());
  flacenc_hip_qlpc_config abi_cfg = detail::to_abi(sc);
  abi_cfg.flags |= gpu.sum_order_flags(sc.qlpc.lpc_order);
  const bool stereo = (nch == 2);
  const size_t per_frame = stereo ? 4 : nch;  // analyses per frame (coding.rs:530-544)

  // 2. group frames by filled size (all but the last are block_size) and batch each group
  size_t f0 = 0;
  while (f0 < bufs.size()) {
    const size_t n = bufs[f0]->filled_size();
    size_t f1 = f0;
    while (f1 < bufs.size() && bufs[f1]->filled_size() == n) ++f1;
    const size_t nf = f1 - f0;
    const bool use_gpu = sc.use_lpc && n >= constant::MIN_BLOCK_SIZE_FOR_PREDICTION;
    // 2a. stereo: try the entry point that also runs the controller on the GPU
    //     (flacenc_hip_encode_stereo_frames: block 4096, order <= 12); its records are turned
    //     into Frames directly.  Anything else falls through to the 4-candidate path below and the
    //     host-side controller -- same result either way (tests/test_gpu_parity.py).
    const bool fused_ok = (sc.use_lpc || sc.use_fixed) && n >= constant::MIN_BLOCK_SIZE_FOR_PREDICTION;
    if (fused_ok && stereo) {
      PinnedBuffer<int32_t> staged(nf * 2 * n);
      for (size_t f = 0; f < nf; ++f)
        for (size_t c = 0; c < 2; ++c)
          std::memcpy(&staged[(f * 2 + c) * n], bufs[f0 + f]->channel_slice(c), n * sizeof(int32_t));
      flacenc_hip_frame_config fc{};
      fc.qlpc = abi_cfg;
      fc.use_constant = sc.use_constant;
      fc.use_fixed = sc.use_fixed;
      fc.fixed_max_order = static_cast<uint32_t>(sc.fixed.max_order);
      fc.fixed_order_sel = sc.fixed.order_sel.type == config::OrderSel::BitCount ? FLACENC_HIP_ORDERSEL_BITCOUNT
                                                                                 : FLACENC_HIP_ORDERSEL_APPROXENT;
      fc.fixed_partitions = static_cast<uint32_t>(sc.fixed.order_sel.partitions);
      fc.use_lpc = sc.use_lpc;
      fc.use_leftside = config.stereo_coding.use_leftside;
      fc.use_rightside = config.stereo_coding.use_rightside;
      fc.use_midside = config.stereo_coding.use_midside;
      std::vector<flacenc_hip_stereo_frame_result> fr(nf);
      PinnedBuffer<int32_t> resid2(nf * 2 * n);
      const int rc = flacenc_hip_encode_stereo_frames(gpu.get(), &fc, staged.data(), nf, static_cast<uint32_t>(n), n,
                                                      static_cast<uint32_t>(bps), fr.data(), resid2.data(), n,
                                                      FLACENC_HIP_MEM_HOST);
      if (rc == FLACENC_HIP_OK) {
        // Frame::write on the GPU too (bitrepr.rs:289-319): the frames' final bytes
        const size_t ostride = flacenc_hip_stereo_frame_bytes_bound(static_cast<uint32_t>(n), static_cast<uint32_t>(bps));
        PinnedBuffer<uint8_t> packed(nf * ostride);
        std::vector<uint32_t> packed_len(nf);
        const int prc = flacenc_hip_pack_stereo_frames(
            gpu.get(), staged.data(), nf, static_cast<uint32_t>(n), n, fr.data(), resid2.data(), n,
            static_cast<uint32_t>(bps), static_cast<uint32_t>(sample_rate), static_cast<uint32_t>(first_number + f0 * number_step),
            static_cast<uint32_t>(number_step),
            packed.data(), ostride, packed_len.data(), FLACENC_HIP_MEM_HOST);
        if (prc != FLACENC_HIP_OK && prc != FLACENC_HIP_ERR_UNSUPPORTED)
          throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
        for (size_t f = 0; f < nf; ++f) {
          // the reference panics where an analysis saw a non-finite or negative-energy autocorrelation
          // (lpc.rs:646, :786-799); the GPU reports it per frame instead
          if (fr[f].analysis_status != FLACENC_HIP_SUBFRAME_OK)
            throw std::runtime_error("LPC analysis reported a non-finite result (lpc.rs:786)");
          const source::FrameBuf& fb = *bufs[f0 + f];
          const int32_t* l = fb.channel_slice(0);
          const int32_t* r = fb.channel_slice(1);
          component::Frame frame;
          frame.frame_number = static_cast<uint32_t>(first_number + (f0 + f) * number_step);
          frame.block_size = n;
          frame.channel_assignment = static_cast<component::ChannelAssignment>(fr[f].channel_assignment);
          for (int c = 0; c < 2; ++c) {
            const int role = fr[f].role[c];
            std::vector<int32_t> sig(n);
            for (size_t t = 0; t < n; ++t)
              sig[t] = role == 0 ? l[t] : role == 1 ? r[t] : role == 2 ? ((l[t] + r[t]) >> 1) : (l[t] - r[t]);
            const uint8_t b = static_cast<uint8_t>(bps + (role == 3 ? 1 : 0));
            if (fr[f].kind[c] == FLACENC_HIP_KIND_CONSTANT) {
              frame.subframes.push_back(component::Constant{n, fr[f].dc_offset[c], b});
            } else if (fr[f].kind[c] == FLACENC_HIP_KIND_VERBATIM) {
              frame.subframes.push_back(component::Verbatim{std::move(sig), b});
            } else if (fr[f].kind[c] == FLACENC_HIP_KIND_FIXED) {
              frame.subframes.push_back(
                  detail::make_fixed_lpc(fr[f].lpc[c], &resid2[(f * 2 + c) * n], sig.data(), n, b));
            } else {
              if (fr[f].lpc[c].status != FLACENC_HIP_SUBFRAME_OK)
                throw std::runtime_error("LPC analysis reported a non-finite result (lpc.rs:786)");
              frame.subframes.push_back(detail::make_lpc(fr[f].lpc[c], &resid2[(f * 2 + c) * n], sig.data(), n, b));
            }
          }
          if (prc == FLACENC_HIP_OK)
            frame.precomputed_bitstream.assign(packed.begin() + f * ostride, packed.begin() + f * ostride + packed_len[f]);
          out.push_back(std::move(frame));
        }
        f0 = f1;
        continue;
      }
      if (rc == FLACENC_HIP_ERR_BAD_CONFIG)
        throw error::EncodeError(error::EncodeError::Config, flacenc_hip_last_error(gpu.get()));
      if (rc != FLACENC_HIP_ERR_UNSUPPORTED)
        throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
    }
    // 2a'. mono / multi-channel: encode_frame for Independent(n) on the GPU (flacenc_hip_encode_frames)
    //      and the frames' bytes (flacenc_hip_pack_frames)
    if (fused_ok && !stereo) {
      PinnedBuffer<int32_t> staged(nf * nch * n);
      for (size_t f = 0; f < nf; ++f)
        for (size_t c = 0; c < nch; ++c)
          std::memcpy(&staged[(f * nch + c) * n], bufs[f0 + f]->channel_slice(c), n * sizeof(int32_t));
      flacenc_hip_frame_config fc{};
      fc.qlpc = abi_cfg;
      fc.use_constant = sc.use_constant;
      fc.use_fixed = sc.use_fixed;
      fc.use_lpc = sc.use_lpc;
      fc.fixed_max_order = static_cast<uint32_t>(sc.fixed.max_order);
      fc.fixed_order_sel = sc.fixed.order_sel.type == config::OrderSel::BitCount ? FLACENC_HIP_ORDERSEL_BITCOUNT
                                                                                 : FLACENC_HIP_ORDERSEL_APPROXENT;
      fc.fixed_partitions = static_cast<uint32_t>(sc.fixed.order_sel.partitions);
      std::vector<flacenc_hip_channel_result> cr(nf * nch);
      PinnedBuffer<int32_t> resid2(nf * nch * n);
      const int rc = flacenc_hip_encode_frames(gpu.get(), &fc, staged.data(), nf, static_cast<uint32_t>(nch),
                                               static_cast<uint32_t>(n), n, static_cast<uint32_t>(bps), cr.data(),
                                               resid2.data(), n, FLACENC_HIP_MEM_HOST);
      if (rc == FLACENC_HIP_OK) {
        const size_t ostride = flacenc_hip_frame_bytes_bound(static_cast<uint32_t>(nch), static_cast<uint32_t>(n),
                                                             static_cast<uint32_t>(bps));
        PinnedBuffer<uint8_t> packed(nf * ostride);
        std::vector<uint32_t> packed_len(nf);
        const int prc = flacenc_hip_pack_frames(gpu.get(), staged.data(), nf, static_cast<uint32_t>(nch),
                                                static_cast<uint32_t>(n), n, cr.data(), resid2.data(), n,
                                                static_cast<uint32_t>(bps), static_cast<uint32_t>(sample_rate),
                                                static_cast<uint32_t>(first_number + f0 * number_step),
            static_cast<uint32_t>(number_step), packed.data(), ostride,
                                                packed_len.data(), FLACENC_HIP_MEM_HOST);
        if (prc != FLACENC_HIP_OK && prc != FLACENC_HIP_ERR_UNSUPPORTED)
          throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
        for (size_t f = 0; f < nf; ++f) {
          component::Frame frame;
          frame.frame_number = static_cast<uint32_t>(first_number + (f0 + f) * number_step);
          frame.block_size = n;
          for (size_t c = 0; c < nch; ++c) {
            const flacenc_hip_channel_result& r = cr[f * nch + c];
            if (r.analysis_status != FLACENC_HIP_SUBFRAME_OK)
              throw std::runtime_error("LPC analysis reported a non-finite result (lpc.rs:786)");
            const int32_t* sig = bufs[f0 + f]->channel_slice(c);
            const uint8_t b = static_cast<uint8_t>(bps);
            if (r.kind == FLACENC_HIP_KIND_CONSTANT) {
              frame.subframes.push_back(component::Constant{n, r.dc_offset, b});
            } else if (r.kind == FLACENC_HIP_KIND_VERBATIM) {
              frame.subframes.push_back(component::Verbatim{std::vector<int32_t>(sig, sig + n), b});
            } else if (r.kind == FLACENC_HIP_KIND_FIXED) {
              frame.subframes.push_back(detail::make_fixed_lpc(r.params, &resid2[(f * nch + c) * n], sig, n, b));
            } else {
              if (r.params.status != FLACENC_HIP_SUBFRAME_OK)
                throw std::runtime_error("LPC analysis reported a non-finite result (lpc.rs:786)");
              frame.subframes.push_back(detail::make_lpc(r.params, &resid2[(f * nch + c) * n], sig, n, b));
            }
          }
          if (prc == FLACENC_HIP_OK)
            frame.precomputed_bitstream.assign(packed.begin() + f * ostride, packed.begin() + f * ostride + packed_len[f]);
          out.push_back(std::move(frame));
        }
        f0 = f1;
        continue;
      }
      if (rc == FLACENC_HIP_ERR_BAD_CONFIG)
        throw error::EncodeError(error::EncodeError::Config, flacenc_hip_last_error(gpu.get()));
      if (rc != FLACENC_HIP_ERR_UNSUPPORTED)
        throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
    }
    // 2b. the LPC candidates of every channel (stereo: L, R, M, S) in one batch
    std::vector<flacenc_hip_subframe_params> recs(use_gpu ? nf * per_frame : 0);
    PinnedBuffer<int32_t> resid(use_gpu ? nf * per_frame * n : 0);
    if (use_gpu) {
      PinnedBuffer<int32_t> staged(nf * nch * n);
      for (size_t f = 0; f < nf; ++f)
        for (size_t c = 0; c < nch; ++c)
          std::memcpy(&staged[(f * nch + c) * n], bufs[f0 + f]->channel_slice(c), n * sizeof(int32_t));
      int rc;
      if (stereo) {
        rc = flacenc_hip_stereo_qlpc_batch(gpu.get(), &abi_cfg, staged.data(), nf, static_cast<uint32_t>(n), n,
                                           static_cast<uint32_t>(bps), recs.data(), resid.data(), n,
                                           FLACENC_HIP_MEM_HOST);
      } else {
        std::vector<uint8_t> bpsv(nf * nch, static_cast<uint8_t>(bps));
        rc = flacenc_hip_qlpc_batch(gpu.get(), &abi_cfg, staged.data(), nf * nch, static_cast<uint32_t>(n), n,
                                    bpsv.data(), recs.data(), resid.data(), n, nullptr, nullptr,
                                    FLACENC_HIP_MEM_HOST);
      }
      if (rc == FLACENC_HIP_ERR_BAD_CONFIG)
        throw error::EncodeError(error::EncodeError::Config, flacenc_hip_last_error(gpu.get()));
      if (rc != FLACENC_HIP_OK) throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
    }
    // 2c. the fixed-LPC candidates of the same subframes (fixed_lpc, coding.rs:298-331)
    const bool use_fixed_gpu = sc.use_fixed && n >= constant::MIN_BLOCK_SIZE_FOR_PREDICTION;
    std::vector<flacenc_hip_subframe_params> frecs;
    std::vector<int32_t> fresid;
    std::vector<uint64_t> fkeys;
    if (use_fixed_gpu) {
      frecs.resize(nf * per_frame);
      fresid.resize(nf * per_frame * n);
      fkeys.resize(nf * per_frame);
      PinnedBuffer<int32_t> staged(nf * nch * n);
      for (size_t f = 0; f < nf; ++f)
        for (size_t c = 0; c < nch; ++c)
          std::memcpy(&staged[(f * nch + c) * n], bufs[f0 + f]->channel_slice(c), n * sizeof(int32_t));
      flacenc_hip_frame_config fc{};
      fc.qlpc = abi_cfg;
      fc.use_fixed = 1;
      fc.fixed_max_order = static_cast<uint32_t>(sc.fixed.max_order);
      fc.fixed_order_sel = sc.fixed.order_sel.type == config::OrderSel::BitCount ? FLACENC_HIP_ORDERSEL_BITCOUNT
                                                                                 : FLACENC_HIP_ORDERSEL_APPROXENT;
      fc.fixed_partitions = static_cast<uint32_t>(sc.fixed.order_sel.partitions);
      const int rc = flacenc_hip_fixed_lpc_batch(
          gpu.get(), &fc, staged.data(), stereo ? nf : nf * nch, static_cast<uint32_t>(n), n, nullptr,
          static_cast<uint32_t>(bps), stereo ? FLACENC_HIP_LAYOUT_STEREO_FRAMES : FLACENC_HIP_LAYOUT_SUBFRAMES,
          frecs.data(), fresid.data(), n, fkeys.data(), FLACENC_HIP_MEM_HOST);
      if (rc == FLACENC_HIP_ERR_BAD_CONFIG)
        throw error::EncodeError(error::EncodeError::Config, flacenc_hip_last_error(gpu.get()));
      if (rc != FLACENC_HIP_OK) throw error::EncodeError(error::EncodeError::Device, flacenc_hip_last_error(gpu.get()));
    }
    // 3. the reference's controller per frame (encode_frame, coding.rs:530-544)
    for (size_t f = 0; f < nf; ++f) {
      const source::FrameBuf& fb = *bufs[f0 + f];
      component::Frame frame;
      frame.frame_number = static_cast<uint32_t>(first_number + (f0 + f) * number_step);
      frame.block_size = n;
      auto rec = [&](size_t k) { return use_gpu ? &recs[f * per_frame + k] : nullptr; };
      auto res = [&](size_t k) { return use_gpu ? &resid[(f * per_frame + k) * n] : nullptr; };
      auto frec = [&](size_t k) { return use_fixed_gpu ? &frecs[f * per_frame + k] : nullptr; };
      auto fres = [&](size_t k) { return use_fixed_gpu ? &fresid[(f * per_frame + k) * n] : nullptr; };
      auto fkey = [&](size_t k) { return use_fixed_gpu ? fkeys[f * per_frame + k] : uint64_t(0); };
      if (!stereo) {
        for (size_t c = 0; c < nch; ++c)
          frame.subframes.push_back(detail::encode_subframe(sc, fb.channel_slice(c), n, static_cast<uint8_t>(bps),
                                                            rec(c), res(c), frec(c), fres(c), fkey(c)));
      } else {
        // try_stereo_coding, coding.rs:469-527
        const int32_t* l = fb.channel_slice(0);
        const int32_t* r = fb.channel_slice(1);
        std::vector<int32_t> m(n), s(n);
        for (size_t t = 0; t < n; ++t) {
          m[t] = (l[t] + r[t]) >> 1;
          s[t] = l[t] - r[t];
        }
        component::SubFrame sl = detail::encode_subframe(sc, l, n, static_cast<uint8_t>(bps), rec(0), res(0), frec(0), fres(0), fkey(0));
        component::SubFrame sr = detail::encode_subframe(sc, r, n, static_cast<uint8_t>(bps), rec(1), res(1), frec(1), fres(1), fkey(1));
        component::SubFrame sm = detail::encode_subframe(sc, m.data(), n, static_cast<uint8_t>(bps), rec(2), res(2), frec(2), fres(2), fkey(2));
        component::SubFrame ss = detail::encode_subframe(sc, s.data(), n, static_cast<uint8_t>(bps + 1), rec(3), res(3), frec(3), fres(3), fkey(3));
        const size_t bl = component::count_bits(sl), br = component::count_bits(sr);
        const size_t bm = component::count_bits(sm), bs = component::count_bits(ss);
        size_t min_bits = bl + br;
        component::ChannelAssignment best = component::ChannelAssignment::Independent;
        if (config.stereo_coding.use_leftside && bl + bs < min_bits) {
          min_bits = bl + bs;
          best = component::ChannelAssignment::LeftSide;
        }
        if (config.stereo_coding.use_rightside && br + bs < min_bits) {
          min_bits = br + bs;
          best = component::ChannelAssignment::RightSide;
        }
        if (config.stereo_coding.use_midside && bm + bs < min_bits) {
          min_bits = bm + bs;
          best = component::ChannelAssignment::MidSide;
        }
        frame.channel_assignment = best;
        // ChannelAssignment::select_channels, datatype.rs:1145-1171
        switch (best) {
          case component::ChannelAssignment::Independent:
            frame.subframes = {std::move(sl), std::move(sr)};
            break;
          case component::ChannelAssignment::LeftSide:
            frame.subframes = {std::move(sl), std::move(ss)};
            break;
          case component::ChannelAssignment::RightSide:
            frame.subframes = {std::move(ss), std::move(sr)};
            break;
          case component::ChannelAssignment::MidSide:
            frame.subframes = {std::move(sm), std::move(ss)};
            break;
        }
      }
      out.push_back(std::move(frame));
    }
    f0 = f1;
  }
  return out;
}

// src/coding.rs:662-674 (serial) / the feed loop of src/par.rs:288-325
template <class SourceT>
std::vector<source::FrameBuf> drain_source(SourceT& src, size_t block_size, size_t nch, size_t bps,
                                           component::Stream& stream) {
  std::vector<source::FrameBuf> bufs;
  for (;;) {
    source::FrameBuf fb(nch, block_size);
    const size_t got = src.read_samples(block_size, fb);
    if (got == 0) break;
    fb.verify_samples(bps);
    stream.stream_info.total_samples += got;
    bufs.push_back(std::move(fb));
  }
  return bufs;
}

template <class SourceT>
component::Stream new_stream_for(const config::Encoder& config, const SourceT& src) {
  try {
    config.verify();
  } catch (const error::VerifyError& e) {
    throw error::EncodeError(error::EncodeError::Config, e.what());
  }
  const size_t nch = src.channels();
  const size_t bps = src.bits_per_sample();
  if (nch < 1 || nch > constant::MAX_CHANNELS || bps < constant::MIN_BITS_PER_SAMPLE ||
      bps > constant::MAX_BITS_PER_SAMPLE)
    throw error::EncodeError(error::EncodeError::Config, "stream_info: channels / bits_per_sample out of range");
  component::Stream stream;
  stream.stream_info.sample_rate = src.sample_rate();
  stream.stream_info.channels = nch;
  stream.stream_info.bits_per_sample = bps;
  return stream;
}
}  // namespace detail

// encode_with_fixed_block_size, src/coding.rs:645-700: reads the whole source, analyses all
// full-size frames in ONE GPU batch (the tail frame, if shorter, in a second one), then runs the
// reference's per-frame controller (on the device for the shapes the ABI covers, else on the host).
template <class SourceT>
component::Stream encode_with_fixed_block_size(const config::Encoder& config, SourceT src, size_t block_size,
                                               HipContext& gpu) {
  component::Stream stream = detail::new_stream_for(config, src);
  const size_t nch = src.channels(), bps = src.bits_per_sample();
  std::vector<source::FrameBuf> bufs = detail::drain_source(src, block_size, nch, bps, stream);
  std::vector<const source::FrameBuf*> run;
  run.reserve(bufs.size());
  for (const source::FrameBuf& fb : bufs) run.push_back(&fb);
  for (component::Frame& f : detail::encode_frame_run(config, run, 0, 1, nch, bps, src.sample_rate(), gpu))
    stream.add_frame(std::move(f));
  // fixed-block mode exposes one block size in STREAMINFO (coding.rs:676-690)
  stream.stream_info.min_block_size = block_size;
  stream.stream_info.max_block_size = block_size;
  return stream;
}

// Sink that stores encoding results by serial id and hands them out in id order: ParSink,
// src/par.rs:67-95 (a BTreeMap behind a Mutex there, a std::map behind a std::mutex here).
template <class T>
class ParSink {
 public:
  void push(size_t idx, T element) {
    std::lock_guard<std::mutex> lock(mutex_);
    data_.emplace(idx, std::move(element));
  }
  template <class F>
  void finalize(F f) {  // empties the sink, calling f in the order of the serial id
    std::map<size_t, T> data;
    {
      std::lock_guard<std::mutex> lock(mutex_);
      data.swap(data_);
    }
    for (auto& kv : data) f(std::move(kv.second));
  }

 private:
  std::mutex mutex_;
  std::map<size_t, T> data_;
};

// encode_with_fixed_block_size over several GPUs -- the shape of the reference's par-mode encoder
// (src/par.rs:355-449) with the worker pool replaced by the node's GPUs: one host thread and one
// handle per device (a handle is single-threaded, like the reference's thread-local scratch), stream
// frame f goes to device f mod G (BASELINE config 4's round-robin; frames are independent),
// every device analyses and packs its frames as one batch with the right frame numbers in the headers
// (first = r, step = G), and a ParSink puts the finished frames back into frame-number order.
// Errors raised on a worker thread are re-thrown here after all workers have been joined.
template <class SourceT>
component::Stream encode_with_fixed_block_size(const config::Encoder& config, SourceT src, size_t block_size,
                                               std::vector<HipContext>& gpus) {
  if (gpus.empty()) throw error::EncodeError(error::EncodeError::Device, "no GPU context given");
  component::Stream stream = detail::new_stream_for(config, src);
  const size_t nch = src.channels(), bps = src.bits_per_sample(), rate = src.sample_rate();
  std::vector<source::FrameBuf> bufs = detail::drain_source(src, block_size, nch, bps, stream);
  const size_t G = gpus.size();
  ParSink<component::Frame> sink;
  std::vector<std::exception_ptr> failures(G);
  std::vector<std::thread> workers;
  workers.reserve(G);
  for (size_t r = 0; r < G; ++r) {
    workers.emplace_back([&, r] {
      try {
        std::vector<const source::FrameBuf*> run;
        for (size_t f = r; f < bufs.size(); f += G) run.push_back(&bufs[f]);
        for (component::Frame& fr : detail::encode_frame_run(config, run, r, G, nch, bps, rate, gpus[r])) {
          const size_t number = fr.frame_number;
          sink.push(number, std::move(fr));
        }
      } catch (...) {
        failures[r] = std::current_exception();
      }
    });
  }
  for (std::thread& t : workers) t.join();
  for (const std::exception_ptr& e : failures)
    if (e) std::rethrow_exception(e);
  sink.finalize([&](component::Frame f) { stream.add_frame(std::move(f)); });
  stream.stream_info.min_block_size = block_size;
  stream.stream_info.max_block_size = block_size;
  return stream;
}

}  // namespace flacenc
#endif  // FLACENC_HOST_FLACENC_HPP_
