// qlpc_bigblock_residual_inst.hip -- one (passes, limbs) instantiation of bigblock_residual_kernel per translation
// unit (compiled with -DFLACENC_BIG_K=<1|2|4> -DFLACENC_BIG_NLB=<2|3|4>; stereo and plain in each).
#include "qlpc_bigblock_residual_impl.h"

#define FLACENC_CAT2(a, b, c) launch_bigblock_residual_##a##_##b
#define FLACENC_CAT(a, b) FLACENC_CAT2(a, b, )

namespace flacenc_hip {
hipError_t FLACENC_CAT(FLACENC_BIG_K, FLACENC_BIG_NLB)(const QlpcKernelArgs& a, hipStream_t stream) {
  return launch_bigblock_residual_inst<FLACENC_BIG_K, FLACENC_BIG_NLB>(a, stream);
}
}  // namespace flacenc_hip
