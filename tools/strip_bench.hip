// micro-benchmark: strip_mul chain, 1 wave per SIMD vs 2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../radiativetransfer.jl_amd/csrc/mom_kernels.hpp"
using namespace mom;
extern __shared__ double smem[];
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(int iters, double *out, int active_waves) {
  double *M = smem;
  for (int e = threadIdx.x; e < 66 * 64; e += 512) M[e] = 1e-3 * ((e * 7) % 13);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  d4 Y[4], T0[4];
  strip_load_lds<15>(M, lr, lq, 16 * (wave & 3), T0);
  strip_copy(Y, T0);
  if (wave < active_waves) {
#pragma nounroll
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {
        d4 acc[4];
        strip_copy(acc, T0);
        strip_mul<15>(M, lr, lq, Y, acc);
        strip_copy(Y, acc);
      } else {
        d4 a1[4], a2[4];
        strip_copy(a1, T0); strip_zero(a2);
        strip_mul2<15>(M, lr, lq, Y, a1, T0, a2);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) Y[rt] = a1[rt] + a2[rt];
      }
    }
  }
  double s = 0;
  for (int rt = 0; rt < 4; ++rt) for (int r = 0; r < 4; ++r) s += Y[rt][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  double *out; hipMalloc(&out, 1024 * 512 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
  for (int aw : {4, 8}) {
    hipFuncSetAttribute((const void*)(mode ? k<1> : k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 140000, 0, iters, out, aw);
      else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 140000, 0, iters, out, aw);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double prods = (double)iters * (mode ? 2 : 1);
      if (rep) printf("mode %d active waves %d: %.3f ms, %.3f us per 60-MFMA product per wave-slot (%.1f%% of 1.6us when 4 waves)\n", mode, aw, ms, ms * 1e3 / prods, 100 * 1.6 / (ms * 1e3 / prods) * (aw / 4));
    }
  }
  return 0;
}
