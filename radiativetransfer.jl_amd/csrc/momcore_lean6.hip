// momcore_lean6.hip -- the six-wave lean strip image (mom_lean.hpp, kLean6): half-strip doubling chains, two workgroups per CU,
// three chain waves on every SIMD.  One object per operator size N = 4 * MOM_STRIP_KS (KS = 9, 10), compiled with -DMOM_WAVES=6
// -DMOM_NS=mom6.  Host entry points used by momcore.hip.
#ifndef MOM_STRIP_KS
#error "compile with -DMOM_STRIP_KS=<N/4>"
#endif
#include <hip/hip_runtime.h>

#include "mom_diag.hpp"
#include "mom_lean.hpp"
#include "mom_host.hpp"

using namespace MOM_NS;

#define MOM_CAT2(a, b) a##b
#define MOM_CAT(a, b) MOM_CAT2(a, b)

hipError_t MOM_CAT(MOM_CAT(mom6_lean, MOM_STRIP_KS), _launch)(const void *layer_args, int grid, hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  const size_t smem = lean_lds_bytes(4 * MOM_STRIP_KS);
  hipError_t e = mom_allow_lds(reinterpret_cast<const void *>(k_layer_lean<MOM_STRIP_KS>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_layer_lean<MOM_STRIP_KS>), dim3(grid), dim3(kThreads), smem, st, a);
  return hipGetLastError();
}
size_t MOM_CAT(MOM_CAT(mom6_lean, MOM_STRIP_KS), _lds_bytes)(int ns) {
  return lean_applies(4 * MOM_STRIP_KS, ns) ? lean_lds_bytes(4 * MOM_STRIP_KS) : 0;
}
