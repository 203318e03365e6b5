// qlpc_wave_inst.hip -- one instantiation of the wave-per-subframe kernel per translation
// unit (compiled with -DFLACENC_MAXP=<8|10|12> -DFLACENC_STEREO=<0|1|2|3|4>; 2 = stereo with the
// on-device candidate / channel-assignment decision, 3 = 2 + the fixed-LPC candidate, 4 = independent
// channels with encode_subframe's decision and the fixed-LPC candidate, 5 = 3 + Frame::write in the
// kernel: packed frame bytes instead of residual rows; 6 / 7 = 3 / 4 with the order selector's walk of the reference's
// f32 chains, QlpcKernelArgs::sumabs_mode).
#include "qlpc_wave_kernel_impl.h"

// -DFLACENC_SPL=72: the same kernel for blocks of 4608 samples (72 per lane), variants 0..4.
#ifndef FLACENC_SPL
#define FLACENC_SPL 64
#endif
#if FLACENC_SPL == 64
#define FLACENC_CAT2(a, b, c) launch_qlpc_wave_##a##_##b
#else
#define FLACENC_CAT2(a, b, c) launch_qlpc_wave72_##a##_##b
#endif
#define FLACENC_CAT(a, b) FLACENC_CAT2(a, b, )

namespace flacenc_hip {
hipError_t FLACENC_CAT(FLACENC_MAXP, FLACENC_STEREO)(const QlpcKernelArgs& a, hipStream_t stream) {
  return launch_wave4096<FLACENC_MAXP, (FLACENC_STEREO != 0 && FLACENC_STEREO != 4 && FLACENC_STEREO != 7), (FLACENC_STEREO >= 2),
                         (FLACENC_STEREO >= 3), (FLACENC_STEREO == 5), FLACENC_SPL, (FLACENC_STEREO >= 6)>(a, stream);
}
}  // namespace flacenc_hip
