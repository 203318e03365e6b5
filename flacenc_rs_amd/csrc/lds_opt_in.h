// lds_opt_in.h -- per-device bookkeeping of the dynamic-LDS opt-in of a kernel.
#ifndef FLACENC_HIP_LDS_OPT_IN_H_
#define FLACENC_HIP_LDS_OPT_IN_H_

#include <hip/hip_runtime.h>
#include <stddef.h>

#include <atomic>

namespace flacenc_hip {

// hipFuncAttributeMaxDynamicSharedMemorySize (the opt-in for more than 64 KiB of dynamic LDS) is an
// attribute of a kernel ON ONE DEVICE, and handles of several devices / host threads share a process:
// remember per device the largest size already configured.  One static instance per kernel.
struct DynamicLdsOptIn {
  static constexpr int kMaxDevices = 64;
  std::atomic<size_t> configured[kMaxDevices];  // static storage: zero-initialised

  hipError_t ensure(const void* kernel, size_t smem) {
    int dev = -1;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && smem <= configured[dev].load(std::memory_order_acquire)) return hipSuccess;
    err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
    if (err != hipSuccess) return err;
    if (tracked) {  // monotonic maximum; concurrent setters only ever raise it
      size_t seen = configured[dev].load(std::memory_order_relaxed);
      while (seen < smem && !configured[dev].compare_exchange_weak(seen, smem, std::memory_order_release)) {
      }
    }
    return hipSuccess;
  }
};

}  // namespace flacenc_hip
#endif
