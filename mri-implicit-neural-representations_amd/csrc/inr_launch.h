// inr_launch.h -- host-side launch helper shared by every kernel translation unit
#pragma once
#include <hip/hip_runtime.h>

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (kernel, DEVICE): a process may drive plans on several GPUs from one thread, so the "already done" bit is kept
// per device ordinal (one static word per kernel instantiation; bit d = device d).
#include <atomic>
namespace inr {
template <auto Kernel>
inline hipError_t allow_full_lds() {
  static std::atomic<unsigned long long> done{0ull};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
  return hipSuccess;
}
}  // namespace inr
