// mom_host.hpp -- host-side helpers shared by the translation units of libmomcore.so.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <utility>

// hipFuncAttributeMaxDynamicSharedMemorySize is set once per (device, kernel image) and raised only when a
// larger LDS image is requested -- not on every launch.
inline hipError_t mom_allow_lds(const void *fn, size_t bytes) {
  static thread_local std::map<std::pair<int, const void *>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  size_t &g = granted[std::make_pair(dev, fn)];
  if (bytes <= g) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) g = bytes;
  return e;
}
