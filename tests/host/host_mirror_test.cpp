// host_mirror_test.cpp -- the reference's end-to-end integrity test, restated on the C++ host
// mirror (flacenc_rs_amd/host/flacenc.hpp) with the LPC candidates served by the GPU.
//
//   src/lib.rs:201-251  e2e_with_generated_sinusoids: channels {1,2,3,5,8}, 16-bit, 16123
//                       samples of Sine(36, 0.4) + noise(0.04), encode_with_fixed_block_size,
//                       decode, compare sample-exactly (test_helper::integrity_test,
//                       src/test_helper.rs:131-185; the decoder here is the crate's own Decode
//                       mirror instead of claxon).
//   src/coding.rs:869-942  tail-block regressions: the last block is shorter than block_size.
//   src/config.rs:438-470  verification errors surface as EncodeError::Config.
//
// Build + run: see tests/test_host_mirror_gpu.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "flacenc.hpp"
#include "flacenc_sigen.h"

using namespace flacenc;

static int failures = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                       \
    }                                                                   \
  } while (0)

static std::vector<int32_t> make_signal(size_t channels, size_t len, uint32_t bps, uint64_t seed) {
  // channel-major from sigen, then interleave like the reference test does
  std::vector<int32_t> chmajor(channels * len);
  flacenc_sigen_fill_frames(chmajor.data(), 1, static_cast<uint32_t>(channels), static_cast<uint32_t>(len), len,
                            bps, 36.0f, 0.4f, 0.04f, seed, 0, 1);
  std::vector<int32_t> inter(channels * len);
  for (size_t t = 0; t < len; ++t)
    for (size_t c = 0; c < channels; ++c) inter[t * channels + c] = chmajor[c * len + t];
  return inter;
}

static void integrity_test(HipContext& gpu, const config::Encoder& cfg, size_t channels, size_t len, uint32_t bps,
                           size_t block_size) {
  const std::vector<int32_t> signal = make_signal(channels, len, bps, 1000 + channels * 7 + block_size);
  auto src = source::MemSource::from_samples(signal, channels, bps, 16000);
  component::Stream stream = encode_with_fixed_block_size(cfg, src, block_size, gpu);
  CHECK(stream.stream_info.total_samples == len);
  CHECK(stream.frames.size() == (len + block_size - 1) / block_size);
  size_t pos = 0, lpc = 0, total = 0, bits = 0;
  for (const component::Frame& f : stream.frames) {
    const auto ch = f.decode_channels();
    CHECK(ch.size() == channels);
    for (size_t c = 0; c < channels; ++c)
      for (size_t t = 0; t < f.block_size; ++t)
        if (ch[c][t] != signal[(pos + t) * channels + c]) {
          std::printf("mismatch frame %u ch %zu t %zu\n", f.frame_number, c, t);
          ++failures;
          return;
        }
    for (const auto& sf : f.subframes) {
      lpc += std::holds_alternative<component::Lpc>(sf);
      ++total;
    }
    bits += f.count_subframe_bits();
    pos += f.block_size;
  }
  CHECK(pos == len);
  std::printf("  ch=%zu n=%zu bps=%u block=%zu: %zu frames, %zu/%zu LPC subframes, %.3f of raw size -- lossless\n",
              channels, len, bps, block_size, stream.frames.size(), lpc, total,
              double(bits) / double(len * channels * bps));
  CHECK(lpc > 0);  // a predictable sinusoid must pick the LPC candidate
}

int main(int argc, char** argv) {
  HipContext gpu(0);
  config::Encoder cfg;                    // defaults: order 10, precision 15, Tukey(0.4), max_p 30
  cfg.subframe_coding.use_fixed = false;  // see flacenc.hpp header note
  for (size_t channels : {1, 2, 3, 5, 8}) integrity_test(gpu, cfg, channels, 16123, 16, 4096);
  integrity_test(gpu, cfg, 2, 16123, 16, 1152);   // ragged partition size (72 samples)
  integrity_test(gpu, cfg, 2, 40000, 24, 8192);   // 24-bit, big block
  // the CPU builds' own summation orders (HipContext::set_sum_order): lossless like the canonical order, and a
  // stream of the same size to within a fraction of a percent (the orders differ in roundings, not in what is coded)
  {
    config::Encoder cd = cfg;
    cd.subframe_coding.use_fixed = true;
    const std::vector<int32_t> signal = make_signal(2, 20000, 16, 4242);
    size_t bits[3] = {0, 0, 0};
    int k = 0;
    for (HipContext::SumOrder so : {HipContext::SumOrder::Canonical, HipContext::SumOrder::Stable, HipContext::SumOrder::SimdNightly}) {
      gpu.set_sum_order(so);
      integrity_test(gpu, cd, 2, 20000, 16, 4096);
      auto src = source::MemSource::from_samples(signal, 2, 16, 44100);
      component::Stream st = encode_with_fixed_block_size(cd, src, 4096, gpu);
      for (const component::Frame& f : st.frames) bits[k] += f.count_subframe_bits();
      ++k;
    }
    gpu.set_sum_order(HipContext::SumOrder::Canonical);
    std::printf("  subframe bits canonical / stable / simd-nightly order: %zu / %zu / %zu\n", bits[0], bits[1], bits[2]);
    for (int j = 1; j < 3; ++j) {
      const double rel = double(bits[j] > bits[0] ? bits[j] - bits[0] : bits[0] - bits[j]) / double(bits[0]);
      CHECK(rel < 0.005);
    }
  }
  {
    config::Encoder c2 = cfg;  // stereo variants off -> Independent only
    c2.stereo_coding.use_leftside = c2.stereo_coding.use_rightside = c2.stereo_coding.use_midside = false;
    integrity_test(gpu, c2, 2, 9000, 16, 4096);
    c2.subframe_coding.qlpc.lpc_order = 8;
    c2.subframe_coding.qlpc.window.type = config::Window::Rectangle;
    c2.subframe_coding.prc.max_parameter = 14;
    integrity_test(gpu, c2, 2, 9000, 16, 4096);
  }
  // digital silence -> Constant subframes (coding.rs:389-391)
  {
    std::vector<int32_t> zeros(2 * 8192, 0);
    auto src = source::MemSource::from_samples(zeros, 2, 16, 44100);
    auto st = encode_with_fixed_block_size(cfg, src, 4096, gpu);
    for (const auto& f : st.frames)
      for (const auto& sf : f.subframes) CHECK(std::holds_alternative<component::Constant>(sf));
  }
  // config errors surface as EncodeError::Config (error.rs:458)
  {
    config::Encoder bad = cfg;
    bad.subframe_coding.qlpc.lpc_order = 25;
    bool thrown = false;
    try {
      auto src = source::MemSource::from_samples(std::vector<int32_t>(8192, 1), 1, 16, 44100);
      encode_with_fixed_block_size(bad, src, 4096, gpu);
    } catch (const error::EncodeError& e) {
      thrown = e.kind == error::EncodeError::Config;
      std::printf("  expected config error: %s\n", e.what());
    }
    CHECK(thrown);
    config::Encoder bad_fixed;
    bad_fixed.subframe_coding.fixed.max_order = 5;  // config.rs:246-255
    thrown = false;
    try {
      auto src = source::MemSource::from_samples(std::vector<int32_t>(8192, 1), 1, 16, 44100);
      encode_with_fixed_block_size(bad_fixed, src, 4096, gpu);
    } catch (const error::EncodeError& e) {
      thrown = e.kind == error::EncodeError::Config;
    }
    CHECK(thrown);
  }
  // the reference's default configuration (use_fixed = true): fused path for stereo 4096 blocks,
  // flacenc_hip_fixed_lpc_batch for the short tail block and every other shape
  {
    config::Encoder def;
    for (size_t channels : {1, 2, 3}) integrity_test(gpu, def, channels, 16123, 16, 4096);
    integrity_test(gpu, def, 2, 16123, 16, 1152);
    integrity_test(gpu, def, 2, 40000, 24, 8192);
    config::Encoder bc = def;
    bc.subframe_coding.fixed.order_sel.type = config::OrderSel::BitCount;
    integrity_test(gpu, bc, 2, 16123, 16, 4096);
    integrity_test(gpu, bc, 1, 9000, 16, 2304);
    // a slow ramp: FixedLpc must win over both Verbatim and the QLPC candidate (fixed_lpc_of_sine-like)
    for (size_t block : {size_t(4096), size_t(1152)}) {
      std::vector<int32_t> ramp(2 * 12000);
      for (size_t t = 0; t < 12000; ++t) {
        ramp[2 * t] = int32_t(t / 7);
        ramp[2 * t + 1] = int32_t((t * t) / 40000) - 1000;
      }
      auto src = source::MemSource::from_samples(ramp, 2, 16, 44100);
      auto st = encode_with_fixed_block_size(def, src, block, gpu);
      size_t fixed = 0, pos = 0;
      for (const auto& f : st.frames) {
        const auto ch = f.decode_channels();
        for (size_t c = 0; c < 2; ++c)
          for (size_t t = 0; t < f.block_size; ++t) CHECK(ch[c][t] == ramp[(pos + t) * 2 + c]);
        for (const auto& sf : f.subframes) fixed += std::holds_alternative<component::FixedLpc>(sf);
        pos += f.block_size;
      }
      std::printf("  ramp, block %zu: %zu FixedLpc subframes of %zu\n", block, fixed, st.frames.size() * 2);
      CHECK(fixed > 0);
    }
  }
  // Stream::write with every frame's bytes made by the GPU bit writer: a complete .flac for a
  // 2-channel stream incl. its short tail block; tests/test_host_mirror_gpu.py parses it back
  {
    config::Encoder def;
    const size_t len = 16123;
    const std::vector<int32_t> signal = make_signal(2, len, 16, 4242);
    auto src = source::MemSource::from_samples(signal, 2, 16, 44100);
    component::Stream st = encode_with_fixed_block_size(def, src, 4096, gpu);
    std::vector<uint8_t> bytes = st.to_bytes();
    CHECK(bytes.size() > 42 && bytes[0] == 'f' && bytes[3] == 'C');
    size_t frame_bytes = 0;
    for (const auto& f : st.frames) frame_bytes += f.precomputed_bitstream.size();
    CHECK(bytes.size() == 42 + frame_bytes);
    CHECK(st.stream_info.max_frame_size >= st.stream_info.min_frame_size && st.stream_info.min_frame_size > 0);
    if (argc > 1) {
      const std::string dir = argv[1];
      FILE* fo = std::fopen((dir + "/mirror.flac").c_str(), "wb");
      CHECK(fo != nullptr);
      if (fo) {
        std::fwrite(bytes.data(), 1, bytes.size(), fo);
        std::fclose(fo);
      }
      fo = std::fopen((dir + "/mirror.pcm").c_str(), "wb");
      if (fo) {
        std::fwrite(signal.data(), sizeof(int32_t), signal.size(), fo);
        std::fclose(fo);
      }
    }
    // mono and multi-channel streams go through flacenc_hip_encode_frames / pack_frames
    for (size_t channels : {size_t(1), size_t(3)}) {
      const std::vector<int32_t> sig = make_signal(channels, 9000, 16, 7 + channels);
      auto msrc = source::MemSource::from_samples(sig, channels, 16, 48000);
      component::Stream ms = encode_with_fixed_block_size(def, msrc, 4096, gpu);
      std::vector<uint8_t> mb = ms.to_bytes();
      CHECK(mb.size() > 42);
      if (argc > 1) {
        const std::string base = std::string(argv[1]) + "/mirror" + std::to_string(channels);
        FILE* fo = std::fopen((base + ".flac").c_str(), "wb");
        if (fo) {
          std::fwrite(mb.data(), 1, mb.size(), fo);
          std::fclose(fo);
        }
        fo = std::fopen((base + ".pcm").c_str(), "wb");
        if (fo) {
          std::fwrite(sig.data(), sizeof(int32_t), sig.size(), fo);
          std::fclose(fo);
        }
      }
    }
  }
  // several handles: the par-mode shape (src/par.rs:355-449; ParSink's ordering is its test
  // par_sink_finalization, par.rs:457-556).  One host thread + handle per device, frame f -> handle
  // f mod G; on a one-GPU box the handles share device 0, on a node each gets its own device.  The
  // stream must be byte-identical to the single-handle one for stereo, mono and 8-channel input, with
  // frame counts that do and do not divide by G, short tail block included.
  {
    const int n_dev = flacenc_hip_device_count();
    config::Encoder def;
    for (size_t G : {size_t(2), size_t(3)}) {
      std::vector<HipContext> gpus;
      for (size_t r = 0; r < G; ++r) gpus.emplace_back(n_dev > 1 ? int(r % size_t(n_dev)) : 0);
      for (size_t channels : {size_t(2), size_t(1), size_t(8)}) {
        const size_t len = 4096 * 7 + 1234;  // 8 frames, the last one short
        const std::vector<int32_t> sig = make_signal(channels, len, 16, 99 + channels);
        auto s1 = source::MemSource::from_samples(sig, channels, 16, 44100);
        auto sg = source::MemSource::from_samples(sig, channels, 16, 44100);
        component::Stream one = encode_with_fixed_block_size(def, s1, 4096, gpu);
        component::Stream many = encode_with_fixed_block_size(def, sg, 4096, gpus);
        CHECK(many.frames.size() == 8 && one.frames.size() == 8);
        for (size_t f = 0; f < many.frames.size(); ++f) CHECK(many.frames[f].frame_number == f);
        CHECK(many.to_bytes() == one.to_bytes());
        CHECK(many.stream_info.total_samples == len);
      }
      std::printf("  %zu handles (%d device%s): streams byte-identical to the single-handle encoder\n", G, n_dev,
                  n_dev == 1 ? "" : "s");
    }
    // a worker's error is re-thrown on the calling thread after the join
    bool thrown = false;
    try {
      std::vector<HipContext> gpus;
      gpus.emplace_back(0);
      gpus.emplace_back(0);
      config::Encoder bad;
      bad.subframe_coding.qlpc.lpc_order = 25;
      auto src = source::MemSource::from_samples(std::vector<int32_t>(4 * 8192, 1), 2, 16, 44100);
      encode_with_fixed_block_size(bad, src, 4096, gpus);
    } catch (const error::EncodeError& e) {
      thrown = e.kind == error::EncodeError::Config;
    }
    CHECK(thrown);
    // ParSink by itself: out-of-order pushes come back in id order (par.rs:457-556)
    ParSink<int> sink;
    for (int id : {5, 1, 4, 0, 3, 2}) sink.push(size_t(id), id * 10);
    std::vector<int> seen;
    sink.finalize([&](int v) { seen.push_back(v); });
    CHECK((seen == std::vector<int>{0, 10, 20, 30, 40, 50}));
  }
  if (failures) {
    std::printf("FAILED: %d\n", failures);
    return 1;
  }
  std::printf("host mirror: all integrity tests passed\n");
  return 0;
}
