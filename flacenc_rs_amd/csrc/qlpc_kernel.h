// qlpc_kernel.h -- launch interface between the C-ABI layer and the HIP kernels.
#ifndef FLACENC_HIP_QLPC_KERNEL_H_
#define FLACENC_HIP_QLPC_KERNEL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "flacenc_hip.h"

namespace flacenc_hip {

// Autocorrelation summation order implemented by the kernels (the "canonical
// order" of DESIGN.md): 16-sample chunk chains + balanced tree over chunk index.
constexpr int kAcorrChunk = 16;

struct QlpcKernelArgs {
  const int32_t* samples;   // device; subframe k at samples + k*stride
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;
  const uint8_t* bps;       // device, per subframe (nullable -> 16)
  const float* window;      // device table with 32 leading pad floats, nullptr = all ones
  int32_t flat_lo;          // window[t] == 1.0f for flat_lo <= t < flat_hi
  int32_t flat_hi;
  uint32_t lpc_order;
  uint32_t precision;
  uint32_t max_rice_parameter;
  flacenc_hip_subframe_params* params;  // device
  int32_t* residual;                    // device
  size_t residual_stride;
  double* autocorr;                     // device, nullable, [n][33]
  double* lpc_coefs;                    // device, nullable, [n][32]
  uint32_t* table_scratch;              // device, only for blocks > 16384 samples
};

struct QlpcLaunchPlan {
  int maxp;        // template bucket for the LPC order
  bool big;        // block_size > 16384: unpadded LDS image, bit tables in HBM scratch
  int threads;     // workgroup size (power of two, 64..1024)
  size_t smem_bytes;
  size_t table_scratch_bytes_per_subframe;
};

QlpcLaunchPlan plan_qlpc_launch(uint32_t block_size, uint32_t lpc_order);
hipError_t launch_qlpc(const QlpcKernelArgs& args, const QlpcLaunchPlan& plan, hipStream_t stream);

}  // namespace flacenc_hip
#endif
