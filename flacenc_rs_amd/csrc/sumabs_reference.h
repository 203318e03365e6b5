// sumabs_reference.h -- launch interface of the reference-order sum-of-|e| kernel
// (FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER / FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER with the ApproxEnt selector).
#ifndef FLACENC_HIP_SUMABS_REFERENCE_H_
#define FLACENC_HIP_SUMABS_REFERENCE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace flacenc_hip {

struct SumAbsRefArgs {
  const int32_t* samples;  // device; subframe k at samples + k*stride (stereo: channel c of frame f at (2f + c)*stride)
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;    // stereo: 4 per frame (L, R, M, S), a multiple of 4
  uint32_t stereo;
  uint32_t partitions;     // OrderSel::ApproxEnt.partitions, 1..64
  uint32_t nightly;        // 0: stable build's single chain; 1: simd-nightly's 16 lanes + head / foot
  float* out;              // device, [n_subframes][5][64]: find_sum_abs_f32 of partition p of order k's errors
};

// find_sum_abs_f32::<16> (src/arrayutils.rs:496-506) over every estimator partition of
// estimate_entropy (src/coding.rs:200-227) for the five fixed-LPC orders, in the reference's own order.
hipError_t launch_sumabs_reference(const SumAbsRefArgs& args, hipStream_t stream);

}  // namespace flacenc_hip
#endif
