// tools/mfma_valu_overlap.hip -- do FP64 VALU instructions of ONE wave run under the FP64 MFMA stream of ANOTHER wave of the same
// SIMD (gfx950)?  8-wave workgroups, one per CU: waves 0..3 issue dependent-free v_mfma_f64_16x16x4 (VGPR accumulators), waves 4..7
// v_fma_f64 chains (or LDS reads); each part alone and both together.  If together = max(alone) the idle waves of the strip-chained
// kernels can do the elemental layer's arithmetic under the chains; if = sum, they cannot (r2 argued the latter from the equal peaks).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_valu_overlap.hip -o scratch/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
extern __shared__ double lds[];

template <int MODE>  // bit 0: MFMA waves work, bit 1: VALU waves work, bit 2: the second group reads LDS instead
__global__ void __launch_bounds__(512) k(int it_m, int it_v, double *out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double s = 0;
  if (wave < 4) {
    if (MODE & 1) {
      d4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
      const double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
      for (int it = 0; it < it_m; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    }
  } else if (MODE & 2) {
    if (MODE & 4) {
      const double *p = lds + (threadIdx.x & 255);
      double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] += p[256 * u];
        asm volatile("" ::: "memory");
      }
      for (int u = 0; u < 8; ++u) s += v[u];
    } else {
      double v[8], x = 1.0 + threadIdx.x * 1e-9, y = 1e-7 * blockIdx.x;
      for (int u = 0; u < 8; ++u) v[u] = u;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_fma(v[u], x, y);
      }
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  if (s == 1.2345) out[blockIdx.x] = s;
}
template <int MODE> float run(int it_m, int it_v) {
  double *out; hipMalloc(&out, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 32768, 0, it_m, it_v, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 32768, 0, it_m, it_v, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}
int main() {
  const int it_m = 20000;            // 80 000 MFMAs per wave
  for (int it_v : {40000, 80000, 160000}) {   // 8 FMAs per iteration per wave
    const float tm = run<1>(it_m, it_v), tv = run<2>(it_m, it_v), tb = run<3>(it_m, it_v);
    printf("FP64 FMA : MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms (max %.3f, sum %.3f)\n", tm, tv, tb, tm > tv ? tm : tv, tm + tv);
  }
  for (int it_v : {40000, 160000}) {
    const float tm = run<1>(it_m, it_v), tv = run<6>(it_m, it_v), tb = run<7>(it_m, it_v);
    printf("LDS reads: MFMA alone %.3f ms, LDS  alone %.3f ms, together %.3f ms (max %.3f, sum %.3f)\n", tm, tv, tb, tm > tv ? tm : tv, tm + tv);
  }
  return 0;
}
