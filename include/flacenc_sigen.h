/*
 * flacenc_sigen.h -- synthetic test-signal generator for measurement inputs.
 *
 * Mirrors the reference's `sigen` module (src/sigen.rs, cargo feature
 * `__export_sigen`): Sine (sigen.rs:133-168) mixed with uniform Noise
 * (sigen.rs:199-232), quantised like Signal::to_vec_quantized
 * (sigen.rs:35-53: x * 2^(bps-1), round half away from zero, clamp).  The noise
 * uses a counter-based splitmix64 stream instead of rand's StdRng (ChaCha12 is
 * not reproducible outside Rust), so the values differ from the reference's but
 * are identical on every machine and for every caller (GPU path, CPU baseline).
 */
#ifndef FLACENC_SIGEN_H_
#define FLACENC_SIGEN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Fills a batch of frames in FrameBuf layout (src/source.rs:115-127): channel c
 * of frame f occupies dst[(f*channels + c)*stride .. +block_size).  Channel c is
 * the continuous stream  Sine(period + 7c, amp, phase 0.5c) + Noise(seed + c, namp)
 * sampled at t = (first_frame + f)*block_size + i.  `nthreads` host threads.
 * Returns 0, or -2 on a bad argument.
 */
int flacenc_sigen_fill_frames(int32_t* dst, size_t n_frames, uint32_t channels, uint32_t block_size,
                              size_t stride, uint32_t bits_per_sample, float sine_period,
                              float sine_amplitude, float noise_amplitude, uint64_t seed,
                              uint64_t first_frame, int nthreads);

/* Same, but local frame f is stream frame first_frame + f*frame_step: the frames a rank owns
 * when a stream is dealt round-robin over `frame_step` GPUs (frame f -> GPU f mod G). */
int flacenc_sigen_fill_frames_strided(int32_t* dst, size_t n_frames, uint32_t channels,
                                      uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                      float sine_period, float sine_amplitude, float noise_amplitude,
                                      uint64_t seed, uint64_t first_frame, uint64_t frame_step,
                                      int nthreads);

#ifdef __cplusplus
}
#endif
#endif
