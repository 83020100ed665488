// FP64 MFMA issue rate by the register class of each operand (one wave per SIMD, 256-thread workgroups, 16 independent
// accumulators): which operands of v_mfma_f64_16x16x4_f64 may sit in AGPRs without the half rate of the AGPR C/D form?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_agpr_operand.hip -o mfma_agpr_operand   (results: profiles/r04_C4_ab.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// MODE 0: A v, B v, C/D v   1: C/D a (A, B v)   2: B a   3: A a   4: A a, B a   5: A a, B a, C/D a
template <int MODE>
__global__ void __launch_bounds__(256) k_op(int iters, double *out) {
  constexpr int NI = 16;
  d4 acc[NI];
  for (int i = 0; i < NI; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (MODE == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b));
      if (MODE == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "v"(b));
      if (MODE == 4) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "a"(b));
      if (MODE == 5) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a), "a"(b));
    }
  }
  double s = 0;
  for (int i = 0; i < NI; ++i) s += acc[i][0] + acc[i][3];
  if (s == 1.2345) out[blockIdx.x] = s;
}
// the strip-chain pattern: X' = M^T X with the running strip X as the B operand held in AGPRs (32 doubles per lane =
// 8 row tiles x 4 k-steps... here 16 B registers cycled), accumulators in VGPRs, A from a VGPR (stands for the LDS read)
template <int BA>
__global__ void __launch_bounds__(256) k_chain(int iters, double *out) {
  constexpr int NT = 8;                      // output row tiles of the strip
  d4 acc[NT];
  double xb[16];                             // the input strip as B operands (16 k-steps)
  for (int i = 0; i < NT; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int k = 0; k < 16; ++k) xb[k] = 1.0 + threadIdx.x * 1e-4 + k;
  double a = threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (BA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(xb[k]));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(xb[k]));
      }
  }
  double s = 0;
  for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][3];
  if (s == 1.2345) out[blockIdx.x] = s;
}
template <class K> void timeit(const char *name, K kern, double mfma_per_wave) {
  double *out; hipMalloc(&out, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern(out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern(out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2048.0 * mfma_per_wave * 4 * 256;
  printf("%-44s %.2f TFLOP/s, %.1f ns per MFMA per wave\n", name, flops / ms * 1e-9, ms * 1e6 / mfma_per_wave);
  hipFree(out);
}
int main() {
  const int it = 10000;
#define RUN(M, label) timeit(label, [&](double *o) { hipLaunchKernelGGL((k_op<M>), dim3(256), dim3(256), 0, 0, it, o); }, 16.0 * it)
  RUN(0, "A v, B v, C/D v");
  RUN(1, "A v, B v, C/D a");
  RUN(2, "A v, B a, C/D v");
  RUN(3, "A a, B v, C/D v");
  RUN(4, "A a, B a, C/D v");
  RUN(5, "A a, B a, C/D a");
  timeit("strip chain, B strip in VGPRs", [&](double *o) { hipLaunchKernelGGL((k_chain<0>), dim3(256), dim3(256), 0, 0, 1000, o); }, 128.0 * 1000);
  timeit("strip chain, B strip in AGPRs", [&](double *o) { hipLaunchKernelGGL((k_chain<1>), dim3(256), dim3(256), 0, 0, 1000, o); }, 128.0 * 1000);
  return 0;
}
