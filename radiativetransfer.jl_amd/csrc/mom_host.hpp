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

// resident HITRAN table of one absorber + the TIPS spline tables of its isotopologues (device pointers)
struct MomLineTable {
  int nLines, nIso, nTmax;
  const double *nu0, *S0, *g_air, *g_self, *E, *n_air, *d_air, *sqw;  // [nLines]
  const int *iso;                                                     // [nLines] index into the spline tables
  const int *nT;                                                      // [nIso] knots per isotopologue
  const double *tT, *tQ, *tZ;                                         // [nIso, nTmax] knots, values, second derivatives
};
// voigt.hip: per-line prefactors of one (p, T) on the device; *unsorted is set when the windows are not monotone
hipError_t mom_line_prefactors_launch(hipStream_t st, const MomLineTable &tb, int nGrid, const double *grid, double p, double T,
                                      double vmr, double wing, double cgd, double *nu, double *gd, double *y, double *S, int *i0,
                                      int *i1, int *unsorted);
