// momcore_gen.hip -- the general (non strip-chained) layer kernels of the 8-wave build, k_layer<LDSM, IFACE>, in a
// translation unit of their own: they are the largest kernel images of the library (eight of them), and compiling them
// next to the rest of momcore.hip serialised the build.  Host entry point used by momcore.hip.
#include <hip/hip_runtime.h>

#include "mom_diag.hpp"
#include "mom_entry.hpp"
#include "mom_host.hpp"

using namespace MOM_NS;

hipError_t mom_gen_launch_layer(const void *layer_args, int iface, bool lds, int grid, size_t smem, hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  hipError_t e = hipSuccess;
#define GEN_LAUNCH(IF)                                                                                         \
  if (lds) {                                                                                                   \
    if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<true, IF>), smem)) != hipSuccess) return e;  \
    hipLaunchKernelGGL((k_layer<true, IF>), dim3(grid), dim3(kThreads), smem, st, a);                          \
  } else {                                                                                                     \
    if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<false, IF>), smem)) != hipSuccess) return e; \
    hipLaunchKernelGGL((k_layer<false, IF>), dim3(grid), dim3(kThreads), smem, st, a);                         \
  }
  if (a.ntgt > 0) {  // multi-target form: one image per LDS mode, interface code dispatched at run time (IFACE = -1)
    if (lds) {
      if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<true, -1, 0, true>), smem)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_layer<true, -1, 0, true>), dim3(grid), dim3(kThreads), smem, st, a);
    } else {
      if ((e = mom_allow_lds(reinterpret_cast<const void *>(k_layer<false, -1, 0, true>), smem)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_layer<false, -1, 0, true>), dim3(grid), dim3(kThreads), smem, st, a);
    }
    return hipGetLastError();
  }
  switch (iface) {  // the interface code is a template argument: see interaction_core
    case 0: GEN_LAUNCH(0) break;
    case 1: GEN_LAUNCH(1) break;
    case 2: GEN_LAUNCH(2) break;
    default: GEN_LAUNCH(3) break;
  }
#undef GEN_LAUNCH
  return hipGetLastError();
}

#ifdef MOM_DIAG_STAMPS
// diagnostic builds: the stamp accumulators of THIS translation unit (the general layer kernels)
extern "C" int mom_diag_read_gen(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mom_diag_acc), 128 * sizeof(unsigned long long)) != hipSuccess) return -2;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mom_diag_acc), z, sizeof z) != hipSuccess) return -2;
  }
  return 0;
}
#endif
