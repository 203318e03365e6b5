// acorr_reference.h -- launch interface of the reference-order autocorrelation kernel
// (FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER).
#ifndef FLACENC_HIP_ACORR_REFERENCE_H_
#define FLACENC_HIP_ACORR_REFERENCE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace flacenc_hip {

struct AcorrRefArgs {
  const int32_t* samples;  // device; subframe k at samples + k*stride (stereo: channel c of frame f at (2f + c)*stride)
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;    // stereo: 4 per frame (L, R, M, S), a multiple of 4
  uint32_t stereo;
  const float* window;     // device table with 32 leading pad floats, nullptr = all ones
  uint32_t lpc_order;      // P: lags 0..P are produced
  uint32_t nightly;        // 0: the stable build's single chain per lag; 1: simd-nightly's lane chains (P <= 15)
  double* out;             // device, [n_subframes][33]; lags above P are written as 0
  // Clean-up behind the sub-wave kernel's order certificate (round 6): only the subframes whose record carries status -2
  // (flacenc_hip_subframe_params, 352 bytes each; stable order only) -- a wave whose four subframes have none returns at
  // once, the whole launch when *marked_count is 0.  nullptr: every subframe.
  const void* marked_params = nullptr;
  const uint32_t* marked_count = nullptr;
  const uint32_t* marked_list = nullptr;  // QlpcKernelArgs::marked_list / marked_cap / marked_unit
  uint32_t marked_cap = 0;
  uint32_t marked_unit = 1;
};

// R[tau] = the single sequential fma chain of weighted_auto_correlation_nosimd (src/lpc.rs:533-548):
// for t in P..n { R[tau] = fma(x_w[t - tau], x_w[t], R[tau]) }, one subframe per lane.
// nightly: weighted_auto_correlation_simd (src/lpc.rs:510-531): 8 / 16 strided lane chains per lag + scalar head
// and foot + ordered lane sum; hipErrorNotSupported above order 15 (alignment-dependent there).
hipError_t launch_acorr_reference(const AcorrRefArgs& args, hipStream_t stream);

}  // namespace flacenc_hip
#endif
