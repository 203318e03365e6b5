// qlpc_subwave_inst.hip -- one (order bucket, stereo, samples per lane, variant) instantiation of the sub-wave kernel per
// translation unit (-DFLACENC_MAXP=<8|10|12> -DFLACENC_STEREO=<0|1> -DFLACENC_SPL=<64|72> -DFLACENC_VARIANT=<0|1|2>: QLPC
// candidates, fixed_lpc batch, 2-channel frame decision), with its three segment widths.
#include "qlpc_subwave_kernel_impl.h"

#define FLACENC_CAT4(a, b, c, d) launch_qlpc_subwave_##a##_##b##_##c##_##d
#define FLACENC_CAT(a, b, c, d) FLACENC_CAT4(a, b, c, d)

namespace flacenc_hip {
hipError_t FLACENC_CAT(FLACENC_MAXP, FLACENC_STEREO, FLACENC_SPL, FLACENC_VARIANT)(const QlpcKernelArgs& a, hipStream_t stream) {
  return launch_subwave<FLACENC_MAXP, (FLACENC_STEREO != 0), FLACENC_SPL, FLACENC_VARIANT>(a, stream);
}
}  // namespace flacenc_hip
