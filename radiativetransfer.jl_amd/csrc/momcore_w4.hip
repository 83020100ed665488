// momcore_w4.hip -- the fused kernels instantiated for 4-wave (256-thread) workgroups, namespace mom4.
// Two such workgroups share a CU when 4 operators + vectors fit 80 KB of LDS (N <= 40); their phases then
// overlap each other's barriers and LDS latencies.  Host entry points are plain C++ functions used by
// momcore.hip; the argument blocks are layout-identical to mom::LayerArgs / mom::SurfArgs.
#define MOM_WAVES 4
#define MOM_TJ 3
#define MOM_NO_STRAIGHT  // operators of this build have at most 12 MFMA k-steps
#define MOM_NS mom4
#include <hip/hip_runtime.h>

#include "mom_diag.hpp"
#include "mom_entry.hpp"
#include "mom_host.hpp"

using namespace mom4;

size_t mom4_lds_bytes(int N, bool lds_mats) { return lds_bytes(N, lds_mats); }
size_t mom4_strip_lds_bytes(int N, int ns) { return strip_lds_bytes(N, ns); }  // strip images: + the persistent stream-pair tables
int mom4_generic_bufs_elems(int N) { return (int)(kGenericBufs * mat_elems(N)); }

template <class K>
static hipError_t allow(K kernel, size_t bytes) {
  return mom_allow_lds(reinterpret_cast<const void *>(kernel), bytes);
}

hipError_t mom4_launch_layer(const void *layer_args, int iface, bool lds, int grid, size_t smem, hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  hipError_t e = hipSuccess;
#define W4_LAUNCH(IF)                                                                         \
  if (lds) {                                                                                  \
    if ((e = allow(k_layer<true, IF>, smem)) != hipSuccess) return e;                         \
    hipLaunchKernelGGL((k_layer<true, IF>), dim3(grid), dim3(kThreads), smem, st, a);         \
  } else {                                                                                    \
    if ((e = allow(k_layer<false, IF>, smem)) != hipSuccess) return e;                        \
    hipLaunchKernelGGL((k_layer<false, IF>), dim3(grid), dim3(kThreads), smem, st, a);        \
  }
  if (a.ntgt > 0) {  // multi-target form (interface code dispatched at run time)
    if (lds) {
      if ((e = allow(k_layer<true, -1, 0, true>, smem)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_layer<true, -1, 0, true>), dim3(grid), dim3(kThreads), smem, st, a);
    } else {
      if ((e = allow(k_layer<false, -1, 0, true>, smem)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_layer<false, -1, 0, true>), dim3(grid), dim3(kThreads), smem, st, a);
    }
    return hipGetLastError();
  }
  switch (iface) {
    case 0: W4_LAUNCH(0) break;
    case 1: W4_LAUNCH(1) break;
    case 2: W4_LAUNCH(2) break;
    default: W4_LAUNCH(3) break;
  }
#undef W4_LAUNCH
  return hipGetLastError();
}

hipError_t mom4_launch_surface(const void *surf_args, bool lds, int grid, size_t smem, hipStream_t st) {
  const SurfArgs a = *reinterpret_cast<const SurfArgs *>(surf_args);
  hipError_t e = hipSuccess;
  if (lds) {
    if ((e = allow(k_surface<true>, smem)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_surface<true>, dim3(grid), dim3(kThreads), smem, st, a);
  } else {
    if ((e = allow(k_surface<false>, smem)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_surface<false>, dim3(grid), dim3(kThreads), smem, st, a);
  }
  return hipGetLastError();
}
