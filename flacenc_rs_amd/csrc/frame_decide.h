// frame_decide.h -- encode_subframe + try_stereo_coding over candidate batches (any block size).
#ifndef FLACENC_HIP_FRAME_DECIDE_H_
#define FLACENC_HIP_FRAME_DECIDE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "flacenc_hip.h"

namespace flacenc_hip {

struct FrameDecideArgs {
  const int32_t* frames;  // device; channel c of frame f at frames + (2f + c)*stride
  size_t stride;
  uint32_t block_size;
  uint32_t n_frames;
  uint32_t bits_per_sample;
  uint32_t use_constant, use_fixed, use_lpc, use_leftside, use_rightside, use_midside;
  // candidates of subframe 4f + role (role = L, R, M, S); null when the candidate kind is off
  const flacenc_hip_subframe_params* lpc_params;
  const int32_t* lpc_residual;
  const flacenc_hip_subframe_params* fixed_params;
  const int32_t* fixed_residual;
  const unsigned long long* fixed_keys;  // the order selector's key of each fixed candidate
  size_t cand_stride;                    // row stride of both candidate residual buffers
  // big-block shapes: the LPC candidates of L and R already sit in output rows 2f / 2f + 1 (QlpcKernelArgs::
  // residual_lr) and the roles' min / max come from the residual kernel ([4 n_frames][2]; null: scan the frame)
  uint32_t lpc_lr_in_place;
  const int32_t* minmax;
  flacenc_hip_stereo_frame_result* results;  // out, [n_frames]
  int32_t* residual;                         // out; output channel c of frame f at (2f + c)*residual_stride
  size_t residual_stride;
  // clean-up behind qlpc_subwave_kernel's frame variant: only frames it marked (channel_assignment 0xFF), and nothing
  // at all when the count it left is 0
  uint32_t only_marked = 0;
  const uint32_t* marked_count = nullptr;
  const uint32_t* marked_list = nullptr;  // QlpcKernelArgs::marked_list (entries: frames resp. subframes)
  uint32_t marked_cap = 0;
};

hipError_t launch_frame_decide(const FrameDecideArgs& args, hipStream_t stream);

// Independent(channels) frames (mono / multi-channel, coding.rs:537-541): one encode_subframe per
// channel, no stereo decision.  Subframe k = f*channels + c everywhere.
struct ChannelDecideArgs {
  const int32_t* samples;  // device; subframe k at samples + k*stride
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;
  uint32_t bits_per_sample;
  uint32_t use_constant, use_fixed, use_lpc;
  const flacenc_hip_subframe_params* lpc_params;
  const int32_t* lpc_residual;
  const flacenc_hip_subframe_params* fixed_params;
  const int32_t* fixed_residual;
  const unsigned long long* fixed_keys;
  size_t cand_stride;
  flacenc_hip_channel_result* results;  // out, [n_subframes]
  int32_t* residual;                    // out; subframe k at k*residual_stride
  size_t residual_stride;
  // clean-up behind qlpc_subwave_kernel's independent-channel variant: only subframes it marked (kind 0xFF)
  uint32_t only_marked = 0;
  const uint32_t* marked_count = nullptr;
  const uint32_t* marked_list = nullptr;  // QlpcKernelArgs::marked_list (entries: frames resp. subframes)
  uint32_t marked_cap = 0;
};
hipError_t launch_channel_decide(const ChannelDecideArgs& args, hipStream_t stream);

}  // namespace flacenc_hip
#endif
