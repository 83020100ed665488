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

// text for mom_last_global_error() (the thread's library-level error string, momcore.hip)
void mom_set_global_error(const char *msg);

// voigt.hip: one launch for all lines (device pointers, stream st); out[g] = acc  or  out[g] += factor * acc
hipError_t mom_voigt_launch(hipStream_t st, int nLines, const double *nu, const double *gamma_d, const double *y,
                            const double *S, const int *i0, const int *i1, int nGrid, const double *grid, double *out,
                            double factor, int accumulate, int sorted);
