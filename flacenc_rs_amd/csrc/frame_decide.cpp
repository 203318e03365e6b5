// frame_decide.cpp -- the per-frame controller of flacenc-rs v0.5.1 for 2-channel frames of ANY
// block size, run on the GPU over candidate batches that are already in HBM:
//   encode_subframe      src/coding.rs:384-418  (Constant / Verbatim / FixedLpc / Lpc, strict '<')
//   try_stereo_coding    src/coding.rs:493-522  (LeftSide, RightSide, MidSide vs Independent)
//   select_channels      src/component/datatype.rs:1173-1185
// It produces exactly what the fused 4096-sample kernel (qlpc_wave_kernel_impl.h, DECIDE) writes:
// one flacenc_hip_stereo_frame_result and the two chosen residual rows per frame.  Inputs are the
// outputs of the stereo QLPC batch and of the stereo fixed-LPC batch (4 candidates each per frame).
#include "frame_decide.h"

#include "frame_decide_device.h"

namespace flacenc_hip {
namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads) frame_decide_kernel(FrameDecideArgs a) {
  __shared__ int smin[4][kThreads / 64], smax[4][kThreads / 64];
  __shared__ FrameDecision sdec;
  const int tid = threadIdx.x;
  // Every frame (grid = n_frames: one trip), or -- only_marked, a small grid -- the frames the marking kernel listed
  // (FrameDecideArgs::marked_list) or, when the list does not hold them all, a grid-stride walk of all frames.  Everything
  // that steers the loop is workgroup-uniform and in front of every barrier of a trip.
  uint32_t count = a.n_frames;
  bool listed = false;
  if (a.only_marked) {
    const uint32_t marks = a.marked_count != nullptr ? __hip_atomic_load(a.marked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
    if (marks == 0u) return;
    listed = a.marked_list != nullptr && marks <= a.marked_cap;
    if (listed) count = marks;
  }
  for (uint32_t trip = blockIdx.x; trip < count; trip += gridDim.x) {
  const uint32_t f = listed ? a.marked_list[trip] : trip;
  if (f >= a.n_frames) continue;
  if (a.only_marked && a.results[f].channel_assignment != 0xFF) continue;
  const int n = (int)a.block_size;
  const int32_t* __restrict__ l = a.frames + (size_t)(2u * f) * a.stride;
  const int32_t* __restrict__ r = l + a.stride;

  // ---- is_constant (arrayutils.rs:382) for L, R, M = (l + r) >> 1, S = l - r ----
  int mn[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
  int mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
  if (a.minmax != nullptr) {  // (already reduced by the residual kernel)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mn[k] = a.minmax[((size_t)f * 4 + k) * 2 + 0];
      mx[k] = a.minmax[((size_t)f * 4 + k) * 2 + 1];
    }
  } else
  for (int t = tid; t < n; t += kThreads) {
    const int lv = l[t], rv = r[t];
    const int v[4] = {lv, rv, (lv + rv) >> 1, lv - rv};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mn[k] = v[k] < mn[k] ? v[k] : mn[k];
      mx[k] = v[k] > mx[k] ? v[k] : mx[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o1 = __shfl_xor(mn[k], d, 64), o2 = __shfl_xor(mx[k], d, 64);
      mn[k] = o1 < mn[k] ? o1 : mn[k];
      mx[k] = o2 > mx[k] ? o2 : mx[k];
    }
    if ((tid & 63) == 0) {
      smin[k][tid >> 6] = mn[k];
      smax[k][tid >> 6] = mx[k];
    }
  }
  __syncthreads();

  // ---- encode_subframe per role (threads 0..3), then try_stereo_coding (thread 0); the two chosen records ----
  int lo = 0, hi = 0;
  if (tid < 4) {
    lo = smin[tid][0];
    hi = smax[tid][0];
    for (int w = 1; w < kThreads / 64; ++w) {
      lo = smin[tid][w] < lo ? smin[tid][w] : lo;
      hi = smax[tid][w] > hi ? smax[tid][w] : hi;
    }
  }
  FrameCandidates cand;
  cand.block_size = a.block_size;
  cand.bits_per_sample = a.bits_per_sample;
  cand.use_constant = a.use_constant;
  cand.use_fixed = a.use_fixed;
  cand.use_lpc = a.use_lpc;
  cand.use_leftside = a.use_leftside;
  cand.use_rightside = a.use_rightside;
  cand.use_midside = a.use_midside;
  cand.lpc_params = a.lpc_params;
  cand.fixed_params = a.fixed_params;
  cand.fixed_keys = a.fixed_keys;
  decide_frame(cand, f, tid, kThreads, lo, hi, sdec, a.results);

  // ---- the two chosen residual rows ----
  for (int c = 0; c < 2; ++c) {
    const uint32_t role = sdec.choice[1 + c];
    const uint32_t kind = sdec.kind[role];
    const size_t sf = (size_t)f * 4 + role;
    const int32_t* src = kind == FLACENC_HIP_KIND_LPC     ? a.lpc_residual + sf * a.cand_stride
                         : kind == FLACENC_HIP_KIND_FIXED ? a.fixed_residual + sf * a.cand_stride
                                                          : nullptr;
    int32_t* dst = a.residual + (size_t)(2u * f + (uint32_t)c) * a.residual_stride;
    // (the L / R candidate of an LPC subframe may already be where it belongs)
    if (a.lpc_lr_in_place && kind == FLACENC_HIP_KIND_LPC && role == (uint32_t)c) continue;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && (n & 3) == 0) {
      const int4* s4 = reinterpret_cast<const int4*>(src);
      int4* d4 = reinterpret_cast<int4*>(dst);
      for (int t = tid; t < (n >> 2); t += kThreads) d4[t] = src ? s4[t] : make_int4(0, 0, 0, 0);
    } else {
      for (int t = tid; t < n; t += kThreads) dst[t] = src ? src[t] : 0;
    }
  }
  __syncthreads();  // (the shared decision and extremes are rewritten by the next trip)
  }
}

// encode_subframe (coding.rs:384-418) for one channel of an Independent(n) frame
__global__ void __launch_bounds__(kThreads) channel_decide_kernel(ChannelDecideArgs a) {
  __shared__ int smin[kThreads / 64], smax[kThreads / 64];
  __shared__ uint32_t skind;
  const int tid = threadIdx.x;
  // (the loop of frame_decide_kernel: every subframe, the listed ones, or a grid-stride walk)
  uint32_t count = a.n_subframes;
  bool listed = false;
  if (a.only_marked) {
    const uint32_t marks = a.marked_count != nullptr ? __hip_atomic_load(a.marked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
    if (marks == 0u) return;
    listed = a.marked_list != nullptr && marks <= a.marked_cap;
    if (listed) count = marks;
  }
  for (uint32_t trip = blockIdx.x; trip < count; trip += gridDim.x) {
  const size_t sf = listed ? a.marked_list[trip] : trip;
  if (sf >= a.n_subframes) continue;
  if (a.only_marked && a.results[sf].kind != 0xFF) continue;
  const int n = (int)a.block_size;
  const int32_t* __restrict__ x = a.samples + sf * a.stride;
  int mn = INT32_MAX, mx = INT32_MIN;
  for (int t = tid; t < n; t += kThreads) {
    const int v = x[t];
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o1 = __shfl_xor(mn, d, 64), o2 = __shfl_xor(mx, d, 64);
    mn = o1 < mn ? o1 : mn;
    mx = o2 > mx ? o2 : mx;
  }
  if ((tid & 63) == 0) {
    smin[tid >> 6] = mn;
    smax[tid >> 6] = mx;
  }
  __syncthreads();
  flacenc_hip_channel_result* out = a.results + sf;
  if (tid == 0) {
    int lo = smin[0], hi = smax[0];
    for (int w = 1; w < kThreads / 64; ++w) {
      lo = smin[w] < lo ? smin[w] : lo;
      hi = smax[w] > hi ? smax[w] : hi;
    }
    const unsigned long long bps = a.bits_per_sample;
    const unsigned long long verbatim_bits = 8ull + (unsigned long long)n * bps;
    const bool have_fixed = a.use_fixed && a.fixed_params && a.fixed_keys[sf] < verbatim_bits;
    const unsigned long long fixed_bits = have_fixed ? a.fixed_params[sf].subframe_bits : ~0ull;
    const unsigned long long baseline = fixed_bits < verbatim_bits ? fixed_bits : verbatim_bits;
    const bool lpc_ok = a.use_lpc && a.lpc_params && a.lpc_params[sf].status == 0;
    uint32_t kind;
    unsigned long long bits;
    if (a.use_constant && lo == hi) {
      kind = FLACENC_HIP_KIND_CONSTANT;
      bits = 8ull + bps;
    } else if (lpc_ok && a.lpc_params[sf].subframe_bits < baseline) {
      kind = FLACENC_HIP_KIND_LPC;
      bits = a.lpc_params[sf].subframe_bits;
    } else if (have_fixed && fixed_bits < verbatim_bits) {
      kind = FLACENC_HIP_KIND_FIXED;
      bits = fixed_bits;
    } else {
      kind = FLACENC_HIP_KIND_VERBATIM;
      bits = verbatim_bits;
    }
    skind = kind;
    out->kind = (uint8_t)kind;
    out->analysis_status = (uint8_t)((a.use_lpc && a.lpc_params) ? a.lpc_params[sf].status : 0);
    out->pad[0] = out->pad[1] = 0;
    out->dc_offset = kind == FLACENC_HIP_KIND_CONSTANT ? lo : 0;
    out->bits = bits;
  }
  __syncthreads();
  const uint32_t kind = skind;
  uint32_t* rec = reinterpret_cast<uint32_t*>(&out->params);
  const uint32_t* src_rec = kind == FLACENC_HIP_KIND_LPC     ? reinterpret_cast<const uint32_t*>(a.lpc_params + sf)
                            : kind == FLACENC_HIP_KIND_FIXED ? reinterpret_cast<const uint32_t*>(a.fixed_params + sf)
                                                             : nullptr;
  for (int i = tid; i < (int)(sizeof(flacenc_hip_subframe_params) / 4); i += kThreads) rec[i] = src_rec ? src_rec[i] : 0u;
  const int32_t* src = kind == FLACENC_HIP_KIND_LPC     ? a.lpc_residual + sf * a.cand_stride
                       : kind == FLACENC_HIP_KIND_FIXED ? a.fixed_residual + sf * a.cand_stride
                                                        : nullptr;
  int32_t* dst = a.residual + sf * a.residual_stride;
  for (int t = tid; t < n; t += kThreads) dst[t] = src ? src[t] : 0;
  __syncthreads();
  }
}

}  // namespace

hipError_t launch_channel_decide(const ChannelDecideArgs& a, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  // (only_marked: a small grid that visits the listed subframes, or walks all of them in strides)
  const uint32_t grid = (a.only_marked && a.n_subframes > 512u) ? 512u : a.n_subframes;
  hipLaunchKernelGGL(channel_decide_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_frame_decide(const FrameDecideArgs& a, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  const uint32_t grid = (a.only_marked && a.n_frames > 512u) ? 512u : a.n_frames;
  hipLaunchKernelGGL(frame_decide_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace flacenc_hip
