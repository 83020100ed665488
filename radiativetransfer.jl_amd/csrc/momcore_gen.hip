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
  switch (iface) {  // the interface code is a template argument: see interaction_core
    case 0: GEN_LAUNCH(0) break;
    case 1: GEN_LAUNCH(1) break;
    case 2: GEN_LAUNCH(2) break;
    default: GEN_LAUNCH(3) break;
  }
#undef GEN_LAUNCH
  return hipGetLastError();
}
