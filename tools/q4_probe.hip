// tools/q4_probe.hip -- feasibility of the "quad-block" product for N = 40 (r6): one WAVE owns a whole 40 x 40 operator in the
// D / B layout of v_mfma_f64_4x4x4_4b (30 registers: block row K = 0..9, column group Jg = 0..2 of four 4 x 4 blocks; element
// (4K + k, 16 Jg + 4 b + j) in lane 16 k + 4 b + j) and multiplies it from the left by M^T read from an LDS buffer of M with
// broadcast reads (lane 16 k + 4 b + i reads M[4K + k + (4I + i) LD]: the four lane groups b read the same 16 values).
// 300 MFMAs + 100 ds_read_b64 per 40^3 product; four one-wave workgroups per CU with 40 960 B of LDS each.
//   hipcc -O3 --offload-arch=gfx950 tools/q4_probe.hip -o scratch/bin/q4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
constexpr int N = 40, NB = 10, NJ = 3, LD = 40;
extern __shared__ double smem[];

__device__ __forceinline__ double mma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// acc(I, Jg) += sum_K M^T(I, K) X(K, Jg)
__device__ __forceinline__ void q4_mul(const double *M, const double (&X)[NB][NJ], double (&acc)[NB][NJ]) {
  const int lane = threadIdx.x & 63, k = lane >> 4, i = lane & 3;
  const double *base = M + k + i * LD;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      const double a = base[4 * K + 4 * I * LD];
#pragma unroll
      for (int J = 0; J < NJ; ++J) acc[I][J] = mma4(a, X[K][J], acc[I][J]);
    }
  }
}

__global__ void __launch_bounds__(64) k_chain(int iters, const double *Mg, const double *Xg, double *out) {
  double *M = smem;
  const int lane = threadIdx.x, k = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
  for (int e = lane; e < N * N; e += 64) M[e] = Mg[e];
  __syncthreads();
  double X[NB][NJ];
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int row = 4 * K + k, col = 16 * J + 4 * b + j;
      X[K][J] = col < N ? Xg[row + col * N] : 0.0;
    }
  for (int it = 0; it < iters; ++it) {
    double acc[NB][NJ];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;
    q4_mul(M, X, acc);
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) X[I][J] = acc[I][J];
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 4 * K + k, col = 16 * J + 4 * b + j;
        if (col < N) out[row + col * N] = X[K][J];
      }
  }
}

int main() {
  double *hM = (double *)malloc(N * N * 8), *hX = (double *)malloc(N * N * 8), *hO = (double *)malloc(N * N * 8);
  for (int e = 0; e < N * N; ++e) { hM[e] = 0.02 * ((e * 7919) % 1013) / 1013.0; hX[e] = ((e * 104729) % 997) / 997.0; }
  double *dM, *dX, *dO;
  hipMalloc(&dM, N * N * 8); hipMalloc(&dX, N * N * 8); hipMalloc(&dO, N * N * 8);
  hipMemcpy(dM, hM, N * N * 8, hipMemcpyHostToDevice); hipMemcpy(dX, hX, N * N * 8, hipMemcpyHostToDevice);
  const size_t lds = 40960;
  hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_chain, 64, lds);
  printf("occupancy API: %d workgroups of one wave with %zu B of LDS per CU\n", nb, lds);
  // correctness: 2 iterations against the host
  hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), lds, 0, 2, dM, dX, dO);
  hipMemcpy(hO, dO, N * N * 8, hipMemcpyDeviceToHost);
  double *t1 = (double *)calloc(N * N, 8), *t2 = (double *)calloc(N * N, 8);
  for (int c = 0; c < N; ++c) for (int r = 0; r < N; ++r) { double s = 0; for (int q = 0; q < N; ++q) s += hM[q + r * N] * hX[q + c * N]; t1[r + c * N] = s; }
  for (int c = 0; c < N; ++c) for (int r = 0; r < N; ++r) { double s = 0; for (int q = 0; q < N; ++q) s += hM[q + r * N] * t1[q + c * N]; t2[r + c * N] = s; }
  double err = 0, mx = 0;
  for (int e = 0; e < N * N; ++e) { err = fmax(err, fabs(t2[e] - hO[e])); mx = fmax(mx, fabs(t2[e])); }
  printf("M^T (M^T X): max |err| %.3e of %.3e\n", err, mx);
  for (int per_cu = 1; per_cu <= 4; ++per_cu) {
    const int iters = 2000, grid = 256 * per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), lds, 0, 10, dM, dX, dO);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), lds, 0, iters, dM, dX, dO);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, fl = 2.0 * N * N * N * iters * (double)grid;
    printf("%d one-wave workgroups per CU: %.3f us per 40^3 product per wave (%.0f cycles at 2.4 GHz; 300 MFMAs = 4800), %.2f TFLOP/s useful = %.3f of 78.6\n",
           per_cu, us, us * 2400, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6);
  }
  return 0;
}
