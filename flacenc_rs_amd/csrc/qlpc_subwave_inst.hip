// qlpc_subwave_inst.hip -- one (order bucket, stereo, samples per lane) instantiation of the sub-wave kernel per
// translation unit (-DFLACENC_MAXP=<8|12> -DFLACENC_STEREO=<0|1> -DFLACENC_SPL=<64|72>), with its three segment widths.
#include "qlpc_subwave_kernel_impl.h"

#define FLACENC_CAT3(a, b, c) launch_qlpc_subwave_##a##_##b##_##c
#define FLACENC_CAT(a, b, c) FLACENC_CAT3(a, b, c)

namespace flacenc_hip {
hipError_t FLACENC_CAT(FLACENC_MAXP, FLACENC_STEREO, FLACENC_SPL)(const QlpcKernelArgs& a, hipStream_t stream) {
  return launch_subwave<FLACENC_MAXP, (FLACENC_STEREO != 0), FLACENC_SPL>(a, stream);
}
}  // namespace flacenc_hip
