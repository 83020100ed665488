// tools/q4_probe2.hip -- the product routine of the quad-block image (mom_q4.hpp q4_mul_c, as shipped) in isolation: ticks per 40^3
// product with 1 .. 4 one-wave workgroups per CU, with and without the riding block row.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DMOM_WAVES=1 -DMOM_TJ=3 -DMOM_NO_STRAIGHT -DMOM_NS=momq
//         -Iinclude -Iradiativetransfer.jl_amd/csrc tools/q4_probe2.hip -o scratch/bin/q4_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "mom_diag.hpp"
#include "mom_q4.hpp"
using namespace MOM_NS;
constexpr int KS = 10, N = 40;

template <int NV>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_chain(int iters, const double *Mg, double *out) {
  double *M = mom_smem, *ride = mom_smem + 3 * N * N;
  const int lane = threadIdx.x;
  for (int e = lane; e < N * N; e += 64) M[e] = Mg[e];
  for (int e = lane; e < 4 * N; e += 64) ride[e] = 1e-3 * e;
  __syncthreads();
  const Q4Lane q;
  double X[KS][3], C0[KS][3], rr[3] = {0, 0, 0};
  q4_load_T<KS>(M, q, X);
  q4_load_T<KS>(M, q, C0);
  for (int it = 0; it < iters; ++it) {
    double acc[KS][3];
    if (NV > 0) q4_mul_c<KS, 2>(M, X, C0, acc, ride, &rr);
    else q4_mul_c<KS>(M, X, C0, acc);
    q4_copy<KS>(X, acc);
  }
  double s = rr[0] + rr[1] + rr[2];
  for (int K = 0; K < KS; ++K) for (int J = 0; J < 3; ++J) s += X[K][J];
  if (s == 1.2345) out[blockIdx.x] = s;
}
template <int NV> void run() {
  double *hM = (double *)malloc(N * N * 8);
  for (int e = 0; e < N * N; ++e) hM[e] = 0.02 * ((e * 7919) % 1013) / 1013.0;
  double *dM, *dO;
  hipMalloc(&dM, N * N * 8); hipMalloc(&dO, 4096 * 8);
  hipMemcpy(dM, hM, N * N * 8, hipMemcpyHostToDevice);
  const size_t lds = q4_lds_bytes(N);
  hipFuncSetAttribute((const void *)k_chain<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int per_cu = 1; per_cu <= 4; per_cu += 3) {
    const int iters = 2000, grid = 256 * per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chain<NV>, dim3(grid), dim3(64), lds, 0, 10, dM, dO);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<NV>, dim3(grid), dim3(64), lds, 0, iters, dM, dO);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    printf("riding rows %d, %d per CU: %.3f us per product + copy (%.0f ticks at 2.39 GHz; %d MFMAs x 16 = %d)\n", NV, per_cu, us, us * 2390,
           300 + (NV ? 30 : 0), 16 * (300 + (NV ? 30 : 0)));
  }
}
int main() { run<0>(); run<2>(); return 0; }
