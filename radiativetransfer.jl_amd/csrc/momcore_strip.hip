// momcore_strip.hip -- k_layer with the strip-chained doubling / interaction paths (mom_strip.hpp) for ONE
// operator size N = 4 * MOM_STRIP_KS, compiled once per size so that each kernel image carries a single
// unrolled variant (Makefile: KS = 13, 14, 15, i.e. N = 52, 56, 60, in the 8-wave build, namespace mom;
// KS = 9, 10, 11, i.e. N = 36, 40, 44, in the 4-wave build, namespace mom4, two workgroups per CU).
#ifndef MOM_STRIP_KS
#error "compile with -DMOM_STRIP_KS=<N/4>"
#endif
#include <hip/hip_runtime.h>

#include "mom_diag.hpp"
#include "mom_entry.hpp"
#include "mom_host.hpp"
// the lean image (three operator buffers, three workgroups per CU): Float64, 4-wave build, N = 36, 40
#if defined(MOM_WAVES) && MOM_WAVES == 4 && !defined(MOM_REAL_IS_FLOAT) && (MOM_STRIP_KS == 9 || MOM_STRIP_KS == 10)
#define MOM_HAVE_LEAN 1
#include "mom_lean.hpp"
#endif

using namespace MOM_NS;

#define MOM_CAT2(a, b) a##b
#define MOM_CAT(a, b) MOM_CAT2(a, b)
// entry-point prefix: mom_strip (Float64 builds) or momf_strip (the Float32 build of the same images: -DMOM_REAL=float
// -DMOM_REAL_IS_FLOAT=1 -DMOM_NS=momf -DMOM_STRIP_PREFIX=momf_strip)
#ifndef MOM_STRIP_PREFIX
#define MOM_STRIP_PREFIX mom_strip
#endif

// host entry used by momcore.hip: mom_strip<KS>_launch_layer(args, iface, grid, smem, stream)
hipError_t MOM_CAT(MOM_CAT(MOM_STRIP_PREFIX, MOM_STRIP_KS), _launch_layer)(const void *layer_args, int iface, int grid, size_t smem,
                                                                     hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  hipError_t e = hipSuccess;
#define STRIP_LAUNCH(IF)                                                                                            \
  if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<true, IF, MOM_STRIP_KS>), smem)) != hipSuccess)     \
    return e;                                                                                                       \
  hipLaunchKernelGGL((k_layer<true, IF, MOM_STRIP_KS>), dim3(grid), dim3(kThreads), smem, st, a);
  if (a.ntgt > 0) {  // multi-target form (interface code dispatched at run time)
    if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<true, -1, MOM_STRIP_KS, true>), smem)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_layer<true, -1, MOM_STRIP_KS, true>), dim3(grid), dim3(kThreads), smem, st, a);
    return hipGetLastError();
  }
  switch (iface) {
    case 0: STRIP_LAUNCH(0) break;
    case 1: STRIP_LAUNCH(1) break;
    case 2: STRIP_LAUNCH(2) break;
    default: STRIP_LAUNCH(3) break;
  }
#undef STRIP_LAUNCH
  return hipGetLastError();
}

// LDS bytes of one workgroup of this image (the vector area depends on the workgroup shape of the build)
size_t MOM_CAT(MOM_CAT(MOM_STRIP_PREFIX, MOM_STRIP_KS), _lds_bytes)() { return lds_bytes(4 * MOM_STRIP_KS, true); }

#ifdef MOM_HAVE_LEAN
// mom_strip<KS>_launch_lean(args, grid, stream): the lean sweep kernel; mom_strip<KS>_lean_lds_bytes(ns): its LDS bytes, 0 if the
// image does not apply to ns Stokes components per stream
hipError_t MOM_CAT(MOM_CAT(MOM_STRIP_PREFIX, MOM_STRIP_KS), _launch_lean)(const void *layer_args, int grid, hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  const size_t smem = lean_lds_bytes(4 * MOM_STRIP_KS);
  hipError_t e = mom_allow_lds(reinterpret_cast<const void *>(k_layer_lean<MOM_STRIP_KS>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_layer_lean<MOM_STRIP_KS>), dim3(grid), dim3(kThreads), smem, st, a);
  return hipGetLastError();
}
size_t MOM_CAT(MOM_CAT(MOM_STRIP_PREFIX, MOM_STRIP_KS), _lean_lds_bytes)(int ns) {
  return lean_applies(4 * MOM_STRIP_KS, ns) ? lean_lds_bytes(4 * MOM_STRIP_KS) : 0;
}
#endif

#ifdef MOM_DIAG_STAMPS
extern "C" int MOM_CAT(MOM_CAT(MOM_STRIP_PREFIX, _diag_read), MOM_STRIP_KS)(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mom_diag_acc), 128 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mom_diag_acc), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
