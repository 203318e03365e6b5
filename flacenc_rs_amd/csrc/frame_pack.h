// frame_pack.h -- launch interface of the frame bit-packing kernel (Frame::write on the GPU).
#ifndef FLACENC_HIP_FRAME_PACK_H_
#define FLACENC_HIP_FRAME_PACK_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "flacenc_hip.h"

namespace flacenc_hip {

struct FramePackArgs {
  const int32_t* frames;  // device; channel c of frame f at frames + (2f + c)*stride
  size_t stride;
  uint32_t block_size;
  uint32_t n_frames;
  const flacenc_hip_stereo_frame_result* results;  // device, [n_frames] (2-channel frames), or
  const flacenc_hip_channel_result* chan_results;  // device, [n_frames * channels] (Independent frames)
  uint32_t channels;                               // used with chan_results
  const int32_t* residual;                         // device; output channel c of frame f at (2f + c)*residual_stride
  size_t residual_stride;
  uint32_t bits_per_sample;
  // FrameHeader bytes 2..3 without the channel assignment (block-size tag << 12 | rate tag << 8 |
  // sample-size tag << 1) and the extra block-size / sample-rate bytes that follow the frame number
  uint32_t header_mid;
  uint32_t extra_len;
  uint8_t extra[4];
  uint32_t first_frame_number;
  uint32_t frame_number_step;
  uint8_t* out;  // device, 16-byte aligned; frame f at out + f*out_stride (multiple of 16)
  size_t out_stride;
  uint32_t* out_len;  // device, [n_frames]
  uint32_t lds_words;  // bit buffer size, a multiple of 4
  // CRC-16 combination: slices of crc_per bytes; crc_pow[i] = y^i, crc_pow[16 + i] = y^(16 i)
  // (i < 16) with y = x^(8 crc_per) mod (x^16 + x^15 + x^2 + 1)
  uint32_t crc_per;
  uint16_t crc_pow[32];
};

// worst-case frame length in bytes for a 2-channel frame (both subframes Verbatim, one a side channel)
size_t stereo_frame_bytes_bound(uint32_t block_size, uint32_t bits_per_sample);
size_t frame_bytes_bound(uint32_t channels, uint32_t block_size, uint32_t bits_per_sample);
hipError_t launch_frame_pack(const FramePackArgs& args, hipStream_t stream);
// only results, n_frames, extra_len, first_frame_number, frame_number_step, out_len are read
hipError_t launch_frame_lengths(const FramePackArgs& args, hipStream_t stream);
// packed little-endian interleaved PCM -> batched FrameBuf layout (int32, channel-major rows)
hipError_t launch_fill_le_bytes(const uint8_t* bytes, uint32_t channels, uint32_t bytes_per_sample,
                                uint64_t total_samples, uint32_t n_frames, uint32_t block_size, int32_t* frames,
                                size_t stride, hipStream_t stream);

// ParSink's reordering (src/par.rs:67-95) for packed frames in HBM: frame i = `lengths[i]` bytes at
// src + src_offsets[i]  ->  dst + dst_offsets[i]; any alignment.
hipError_t launch_place_frames(const uint8_t* src, const uint64_t* src_offsets, const uint32_t* lengths,
                               uint8_t* dst, const uint64_t* dst_offsets, uint32_t n_frames, hipStream_t stream);

// compaction plan of one chunk of packed frames: src_offsets[f] = f * src_stride, dst_offsets[f] = exclusive
// prefix sum of lengths (+ base), total[0] = sum of lengths.  One workgroup; n_frames <= 65535.
hipError_t launch_frame_offsets(const uint32_t* lengths, uint32_t n_frames, size_t src_stride, uint64_t* src_offsets,
                                uint64_t* dst_offsets, uint64_t* total, hipStream_t stream);

// records -> wire records (48 + 2 * (96 + parts) bytes each, at wire + f * wire_stride) and, when args.out_len is
// set, the frames' byte lengths in the same pass; reads what launch_frame_lengths reads
hipError_t launch_frame_wire(const FramePackArgs& args, uint32_t parts, uint8_t* wire, size_t wire_stride,
                             hipStream_t stream);
// exclusive prefix sum over the all-gathered lengths (rank-major [world][per_rank]) in stream order
hipError_t launch_stream_offsets(const uint32_t* lengths, uint32_t n_frames, uint32_t world, uint32_t per_rank,
                                 uint64_t header_bytes, uint32_t* lengths_stream, uint64_t* offsets, uint64_t* total,
                                 hipStream_t stream);

}  // namespace flacenc_hip
#endif
