// momcore_q4.hip -- the quad-block image (mom_q4.hpp): one wavefront per (spectral point, moment) unit of an N = 36 / 40 problem,
// v_mfma_f64_4x4x4_4b products, four units per CU.  One object per operator size N = 4 * MOM_STRIP_KS (KS = 9, 10), compiled with
// -DMOM_WAVES=1 -DMOM_NS=momq.  Host entry points used by momcore.hip.
#ifndef MOM_STRIP_KS
#error "compile with -DMOM_STRIP_KS=<N/4>"
#endif
#include <hip/hip_runtime.h>

#include "mom_diag.hpp"
#include "mom_q4.hpp"
#include "mom_host.hpp"

using namespace MOM_NS;

#define MOM_CAT2(a, b) a##b
#define MOM_CAT(a, b) MOM_CAT2(a, b)

hipError_t MOM_CAT(MOM_CAT(momq_q4_, MOM_STRIP_KS), _launch)(const void *layer_args, int grid, hipStream_t st) {
  const LayerArgs a = *reinterpret_cast<const LayerArgs *>(layer_args);
  const size_t smem = q4_lds_bytes(4 * MOM_STRIP_KS);
  hipError_t e = mom_allow_lds(reinterpret_cast<const void *>(k_layer_q4<MOM_STRIP_KS>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_layer_q4<MOM_STRIP_KS>), dim3(grid), dim3(64), smem, st, a);
  return hipGetLastError();
}
// workgroups of this image a CU holds: by LDS and by the registers the compiler gave the kernel (occupancy API)
int MOM_CAT(MOM_CAT(momq_q4_, MOM_STRIP_KS), _per_cu)() {
  int nb = 0;
  const size_t smem = q4_lds_bytes(4 * MOM_STRIP_KS);
  if (mom_allow_lds(reinterpret_cast<const void *>(k_layer_q4<MOM_STRIP_KS>), smem) != hipSuccess) return 4;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_layer_q4<MOM_STRIP_KS>, 64, smem) != hipSuccess || nb < 1) return 4;
  return nb;
}
// LDS bytes of one (one-wave) workgroup; 0 if the image does not apply to ns Stokes components per stream and K phase-matrix bases
size_t MOM_CAT(MOM_CAT(momq_q4_, MOM_STRIP_KS), _lds_bytes)(int ns, int K) {
  return q4_applies(4 * MOM_STRIP_KS, ns, K) ? q4_lds_bytes(4 * MOM_STRIP_KS) : 0;
}

#ifdef MOM_DIAG_STAMPS
extern "C" int MOM_CAT(momq_q4_diag_read, MOM_STRIP_KS)(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mom_diag_acc), 128 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mom_diag_acc), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
