// qlpc_inst.hip -- one instantiation of the fused QLPC kernel per translation unit
// (compiled with -DFLACENC_MAXP=<bucket> -DFLACENC_BIG=<0|1>) so the eight
// variants build in parallel.
#include "qlpc_kernel_impl.h"

#define FLACENC_CAT2(a, b, c) launch_qlpc_##a##_##b
#define FLACENC_CAT(a, b) FLACENC_CAT2(a, b, )

namespace flacenc_hip {
hipError_t FLACENC_CAT(FLACENC_MAXP, FLACENC_BIG)(const QlpcKernelArgs& a, int threads, size_t smem,
                                                   hipStream_t stream) {
  return launch_one<FLACENC_MAXP, (FLACENC_BIG != 0)>(a, threads, smem, stream);
}
#define FLACENC_LCAT2(a, b, c) launch_levinson_##a##_##b
#define FLACENC_LCAT(a, b) FLACENC_LCAT2(a, b, )
hipError_t FLACENC_LCAT(FLACENC_MAXP, FLACENC_BIG)(const QlpcKernelArgs& a, hipStream_t stream) {
  return launch_levinson_batch<FLACENC_MAXP>(a, stream);
}
}  // namespace flacenc_hip
