// direct_mse.h -- launch interface of the covariance-method ("direct MSE") / IRLS estimator kernel
// (config::Qlpc::use_direct_mse / mae_optimization_steps, src/config.rs:280-285; experimental in the reference).
#ifndef FLACENC_HIP_DIRECT_MSE_H_
#define FLACENC_HIP_DIRECT_MSE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace flacenc_hip {

struct DirectMseArgs {
  const int32_t* samples;  // device; subframe k at samples + k*stride (stereo: channel c of frame f at (2f + c)*stride)
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;    // stereo: 4 per frame (L, R, M, S), a multiple of 4
  uint32_t stereo;
  const float* window;     // device table with 32 leading pad floats, nullptr = all ones
  uint32_t lpc_order;      // 1..32
  uint32_t precision;      // quant_precision
  uint32_t mae_steps;      // 0: lpc_with_direct_mse; > 0: lpc_with_irls_mae with that many re-weighting steps
  int32_t* pred_out;       // device, [n][36]: qc[32], order, shift, status, 0 (as levinson_batch_kernel writes it)
  double* autocorr;        // device, nullable, [n][33]: R[0..=P] (of the chosen IRLS step)
  double* lpc_coefs;       // device, nullable, [n][32]: the unquantised solution
  float* weight_scratch;   // device, [n][(block_size + 3) & ~3] floats: the IRLS weights of blocks above 16384 samples (else unused)
  // Without IRLS steps and with this scratch ([n][direct_mse_gram_stride(order)] doubles: R[0..32], then the order x order
  // matrix column-major), the chains' kernel stops behind them and direct_mse_solve_kernel factorises, solves and
  // quantises with a LANE per subframe; nullptr: everything in the one kernel, as with IRLS.
  double* gram_scratch;
  // IRLS in the same form (orders up to 11, aligned rows): per step the weighted chains, the batched solve and an
  // error pass (raw errors, their sequential f32 sum, the best step so far, the next weights).  irls_state: [n][80]
  // doubles -- the step's solution [32], the best one [32], {best error, have a best, status, 0...}; irls_weights:
  // [n][(block_size + 3) & ~3] floats (= weight_scratch's layout).  Set by launch_direct_mse from `weight_scratch` and
  // the tail of `gram_scratch` when both are there; `irls_step` is the launcher's loop variable.
  double* irls_state;
  float* irls_weights;
  uint32_t irls_step;
};
constexpr size_t kIrlsStateDoubles = 80;

__host__ __device__ inline size_t direct_mse_gram_stride(uint32_t order) { return (33u + (size_t)order * order + 1u) & ~(size_t)1; }

size_t direct_mse_lds_bytes(uint32_t block_size, bool irls, uint32_t order);
// perform_qlpc's experimental branches (src/coding.rs:337-347) for a batch, one workgroup per subframe;
// hipErrorNotSupported when the block does not fit the LDS (never for block_size <= 32767: above 16384 samples the IRLS
// weights go to `weight_scratch`).
hipError_t launch_direct_mse(const DirectMseArgs& args, hipStream_t stream);

}  // namespace flacenc_hip
#endif
