// sigen.cpp -- host-side synthetic signal generator (see include/flacenc_sigen.h).
#include "flacenc_sigen.h"

#include <cmath>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// uniform in (0, 1), 24 random bits -- the role of rand's Open01 in sigen.rs:229
inline float open01(uint64_t seed, uint64_t t) {
  const uint64_t bits = splitmix64(seed * 0x100000001B3ull + t) >> 40;
  return (static_cast<float>(bits) + 0.5f) * (1.0f / 16777216.0f);
}

void fill_range(int32_t* dst, size_t f_begin, size_t f_end, uint32_t channels, uint32_t block_size,
                size_t stride, uint32_t bps, float period, float amp, float namp, uint64_t seed,
                uint64_t first_frame, uint64_t frame_step) {
  const float pi = 3.14159265358979323846f;
  const float scale = static_cast<float>(1u << (bps - 1));
  const float lo = -scale, hi = scale - 1.0f;
  for (size_t f = f_begin; f < f_end; ++f) {
    for (uint32_t c = 0; c < channels; ++c) {
      int32_t* out = dst + (f * channels + c) * stride;
      const float period_c = period + 7.0f * static_cast<float>(c);
      const float phase_c = 0.5f * static_cast<float>(c);
      const uint64_t t0 = (first_frame + f * frame_step) * block_size;
      for (uint32_t i = 0; i < block_size; ++i) {
        const uint64_t ti = t0 + i;
        const float t = static_cast<float>(ti);
        // Sine::fill_buffer, sigen.rs:159-168
        const float s = amp * sinf(phase_c + 2.0f * pi * t / period_c);
        // Noise::fill_buffer, sigen.rs:227-232
        const float nz = namp * 2.0f * (open01(seed + c, ti) - 0.5f);
        // Mix (weights 1, 1) + to_vec_quantized, sigen.rs:35-53
        float v = roundf(scale * (s + nz));
        v = v < lo ? lo : (v > hi ? hi : v);
        out[i] = static_cast<int32_t>(v);
      }
    }
  }
}

}  // namespace

extern "C" int flacenc_sigen_fill_frames(int32_t* dst, size_t n_frames, uint32_t channels,
                                         uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                         float sine_period, float sine_amplitude, float noise_amplitude,
                                         uint64_t seed, uint64_t first_frame, int nthreads) {
  return flacenc_sigen_fill_frames_strided(dst, n_frames, channels, block_size, stride, bits_per_sample,
                                           sine_period, sine_amplitude, noise_amplitude, seed,
                                           first_frame, 1, nthreads);
}

extern "C" int flacenc_sigen_fill_frames_strided(int32_t* dst, size_t n_frames, uint32_t channels,
                                                 uint32_t block_size, size_t stride,
                                                 uint32_t bits_per_sample, float sine_period,
                                                 float sine_amplitude, float noise_amplitude,
                                                 uint64_t seed, uint64_t first_frame,
                                                 uint64_t frame_step, int nthreads) {
  if (!dst || channels == 0 || block_size == 0 || stride < block_size || bits_per_sample < 5 ||
      bits_per_sample > 25 || !(sine_period > 0.0f))
    return -2;
  if (nthreads < 1) nthreads = 1;
  if (static_cast<size_t>(nthreads) > n_frames) nthreads = n_frames ? static_cast<int>(n_frames) : 1;
  std::vector<std::thread> th;
  for (int i = 0; i < nthreads; ++i) {
    const size_t b = n_frames * i / nthreads, e = n_frames * (i + 1) / nthreads;
    th.emplace_back(fill_range, dst, b, e, channels, block_size, stride, bits_per_sample, sine_period,
                    sine_amplitude, noise_amplitude, seed, first_frame, frame_step);
  }
  for (auto& t : th) t.join();
  return 0;
}
