// frame_decide_device.h -- encode_subframe's choice for the four roles of a stereo frame and try_stereo_coding's
// channel assignment (src/coding.rs:384-418, 493-522; ChannelAssignment::select_channels, datatype.rs:1173-1185) from
// candidate records that are already in HBM: what frame_decide_kernel and the deciding store pass of the big-block
// pipeline (bigblock_residual_kernel, MODE 2) both run.
#ifndef FLACENC_HIP_FRAME_DECIDE_DEVICE_H_
#define FLACENC_HIP_FRAME_DECIDE_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flacenc_hip.h"

namespace flacenc_hip {

struct FrameCandidates {
  uint32_t block_size, bits_per_sample;
  uint32_t use_constant, use_fixed, use_lpc, use_leftside, use_rightside, use_midside;
  const flacenc_hip_subframe_params* lpc_params;    // [4 frames]: roles L, R, M, S; null when the candidate kind is off
  const flacenc_hip_subframe_params* fixed_params;
  const unsigned long long* fixed_keys;             // the order selector's key of each fixed candidate
};

// Shared-memory scratch of the decision.
struct FrameDecision {
  unsigned long long bits[4];
  uint32_t kind[4], status[4];
  int dc[4];
  uint32_t choice[3];  // assignment, role of output channel 0, of channel 1
};

// Threads 0..3 of the workgroup decide the roles, thread 0 the assignment; `lo` / `hi`: the roles' minima / maxima
// (thread k < 4 passes those of role k).  Two workgroup barriers inside; writes the result's header fields and copies
// the two chosen predictor records (all `nthreads` threads take part).
__device__ __forceinline__ void decide_frame(const FrameCandidates& c, uint32_t f, int tid, int nthreads, int lo, int hi,
                                             FrameDecision& sh, flacenc_hip_stereo_frame_result* results) {
  const int n = (int)c.block_size;
  if (tid < 4) {
    const int role = tid;
    const unsigned long long bps = c.bits_per_sample + (role == 3 ? 1u : 0u);  // coding.rs:444
    const unsigned long long verbatim_bits = 8ull + (unsigned long long)n * bps;  // datatype.rs:1944
    const size_t sf = (size_t)f * 4 + role;
    const bool have_fixed = c.use_fixed && c.fixed_params && c.fixed_keys[sf] < verbatim_bits;  // coding.rs:262, :284
    const unsigned long long fixed_bits = have_fixed ? c.fixed_params[sf].subframe_bits : ~0ull;
    const unsigned long long baseline = fixed_bits < verbatim_bits ? fixed_bits : verbatim_bits;  // coding.rs:403-405
    const bool lpc_ok = c.use_lpc && c.lpc_params && c.lpc_params[sf].status == 0;
    uint32_t kind;
    unsigned long long bits;
    if (c.use_constant && lo == hi) {
      kind = FLACENC_HIP_KIND_CONSTANT;
      bits = 8ull + bps;  // bitrepr.rs:445
    } else if (lpc_ok && c.lpc_params[sf].subframe_bits < baseline) {
      kind = FLACENC_HIP_KIND_LPC;
      bits = c.lpc_params[sf].subframe_bits;
    } else if (have_fixed && fixed_bits < verbatim_bits) {
      kind = FLACENC_HIP_KIND_FIXED;
      bits = fixed_bits;
    } else {
      kind = FLACENC_HIP_KIND_VERBATIM;
      bits = verbatim_bits;
    }
    sh.kind[role] = kind;
    sh.bits[role] = bits;
    sh.dc[role] = lo;
    sh.status[role] = (c.use_lpc && c.lpc_params) ? (uint32_t)c.lpc_params[sf].status : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned long long bl = sh.bits[0], br = sh.bits[1], bm = sh.bits[2], bs = sh.bits[3];
    unsigned long long min_bits = bl + br;
    uint32_t assignment = 0;  // Independent(2)
    if (c.use_leftside && bl + bs < min_bits) {
      min_bits = bl + bs;
      assignment = 1;
    }
    if (c.use_rightside && br + bs < min_bits) {
      min_bits = br + bs;
      assignment = 2;
    }
    if (c.use_midside && bm + bs < min_bits) {
      min_bits = bm + bs;
      assignment = 3;
    }
    sh.choice[0] = assignment;
    sh.choice[1] = assignment == 2 ? 3u : (assignment == 3 ? 2u : 0u);
    sh.choice[2] = (assignment == 0 || assignment == 2) ? 1u : 3u;
    flacenc_hip_stereo_frame_result* fr = results + f;
    fr->channel_assignment = (uint8_t)assignment;
    // analysis status of the four LPC candidates (the reference panics on these, lpc.rs:646 / :786-799)
    fr->analysis_status = (uint8_t)(sh.status[0] | sh.status[1] | sh.status[2] | sh.status[3]);
    fr->pad[0] = fr->pad[1] = 0;
    for (int ch = 0; ch < 2; ++ch) {
      const uint32_t role = sh.choice[1 + ch];
      fr->role[ch] = (uint8_t)role;
      fr->kind[ch] = (uint8_t)sh.kind[role];
      fr->dc_offset[ch] = sh.kind[role] == FLACENC_HIP_KIND_CONSTANT ? sh.dc[role] : 0;
    }
    fr->bits[0] = bl;
    fr->bits[1] = br;
    fr->bits[2] = bm;
    fr->bits[3] = bs;
  }
  __syncthreads();
  for (int ch = 0; ch < 2; ++ch) {
    const uint32_t role = sh.choice[1 + ch];
    const uint32_t kind = sh.kind[role];
    const size_t sf = (size_t)f * 4 + role;
    uint32_t* rec = reinterpret_cast<uint32_t*>(&results[f].lpc[ch]);
    const uint32_t* src_rec = kind == FLACENC_HIP_KIND_LPC     ? reinterpret_cast<const uint32_t*>(c.lpc_params + sf)
                              : kind == FLACENC_HIP_KIND_FIXED ? reinterpret_cast<const uint32_t*>(c.fixed_params + sf)
                                                               : nullptr;
    for (int i = tid; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += nthreads) rec[i] = src_rec ? src_rec[i] : 0u;
  }
}

}  // namespace flacenc_hip
#endif
