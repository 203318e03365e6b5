/* flacenc_hip_debug.h -- test and profiling hooks.  NOT part of the drop-in boundary (include/flacenc_hip.h) and NOT in
 * the product library: the Makefile links libflacenc_hip.so without them and a second library, libflacenc_hip_hooks.so --
 * the same objects, flacenc_hip_api.cpp compiled once more with -DFLACENC_HIP_DEBUG_HOOKS -- for tests/ and tools/
 * (flacenc_rs_amd/_capi.py: Handle(dev, hooks=True)). */
#ifndef FLACENC_HIP_DEBUG_H_
#define FLACENC_HIP_DEBUG_H_

#include "flacenc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook (no reference counterpart): when `device_keys` is non-NULL, launches with use_fixed
 * store the order selector's key for each tried fixed order (estimate_entropy + bps*order, or
 * the BitCount bits; src/coding.rs:249, :271) at device_keys[subframe*8 + order]. */
int flacenc_hip_debug_set_fixed_keys(flacenc_hip_handle* h, unsigned long long* device_keys);

/* Profiling hook (no reference counterpart): when `device_stamps` is non-NULL every
 * following launch makes each workgroup leader store 8 shader-clock timestamps
 * (phase boundaries of the fused kernel) at device_stamps[subframe*8 + phase].
 * Pass NULL to switch it off again.  See tools/phase_profile.py. */
int flacenc_hip_debug_set_stamps(flacenc_hip_handle* h, unsigned long long* device_stamps);

/* Statistics hook (no reference counterpart): when `device_counters` is non-NULL (3 x uint32, zeroed by the caller),
 * launches that certify their summation order (blocks of 4096 / 4608 samples, orders up to 12, no order flag) add to
 * [0] the subframes analysed, [1] the certificates that needed the rows of the inverse Toeplitz matrix, [2] the subframes
 * recomputed from the reference's chains. */
int flacenc_hip_debug_set_cert_stats(flacenc_hip_handle* h, uint32_t* device_counters);
/* 0: launches of the certified shapes always take the fused kernel's certificate; 1 (default): integer-only launches of
 * any size take the two-pass form (the reference's chains for every subframe, same integers) while the
 * certificate's counters of the launches before them say the material is hard (flacenc_hip_api.cpp, launch_adaptive) */
int flacenc_hip_debug_set_adaptive_order(flacenc_hip_handle* h, int on);
/* the current span of two-pass launches (0: the material last seen was easy) and how many of it are left */
int flacenc_hip_debug_adaptive_state(flacenc_hip_handle* h, int* span, int* left);

#ifdef __cplusplus
}
#endif
#endif /* FLACENC_HIP_DEBUG_H_ */
