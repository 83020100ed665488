// micro-benchmark of wg_gemm_big (N = 256 products on per-workgroup slabs), build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DPAIR] -Iinclude -Iradiativetransfer.jl_amd/csrc tools/gemm_big_bench.hip -o gemm_big_bench
// 'own slabs': every workgroup streams its own 2.6 MB slab (HBM); 'shared slab': all workgroups read the same one (L2);
// r6: `nslab` distinct slabs, workgroup b on slab b % nslab -- 64 slabs = 166 MB stay in the 256 MB Infinity Cache, 8 slabs = one
// per XCD stay in its 4 MB L2: the memory-system ceiling of ANY scheme that spreads a unit over several CUs to keep its slab set
// on chip, before the cost of its cross-workgroup hand-offs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "mom_kernels.hpp"
using namespace mom;
struct StoreEpi {
  double *o; int ld;
  __device__ void operator()(int i, int j, double v) const { o[i + j * ld] = v; }
#ifdef PAIR
  __device__ void operator()(int i, int j, double v0, double v1) const { typedef double d2_ __attribute__((ext_vector_type(2))); *(d2_ *)(o + i + j * ld) = (d2_){v0, v1}; }
#endif
};
template <int VAR>
__global__ void __launch_bounds__(kThreads) k_loop(int N, int ld, int iters, double *slab, size_t stride, double *out, int nslab) {
  double *base = slab + (size_t)(blockIdx.x % nslab) * stride;
  double *A = base, *B = base + (size_t)ld * N, *C = base + 2 * (size_t)ld * N, *D = base + 3 * (size_t)ld * N;
  for (int it = 0; it < iters; ++it) {
    double *o = (it & 1) ? D : C;
    if (VAR == 0) wg_gemm_big(N, N, ElP{A, ld}, ElP{B, ld}, StoreEpi{o, ld});
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = C[1] + D[2];
}
template <int VAR> void run(const char *name, int N, int iters, bool shared_slab, int nslab = 256) {
  const int ld = 16 * ((N + 15) / 16) + 2;
  const size_t stride = 5 * (size_t)ld * N;
  const int nwg = 256;
  double *slab, *out;
  hipMalloc(&slab, nwg * stride * 8); hipMalloc(&out, nwg * 8);
  double *h = (double *)malloc(nwg * stride * 8);
  for (size_t i = 0; i < nwg * stride; ++i) h[i] = 1e-3 * ((i * 7919) % 1013) / 1013.0;
  hipMemcpy(slab, h, nwg * stride * 8, hipMemcpyHostToDevice);
  size_t sm = lds_bytes(N, false); printf("lds %zu  ", sm);
  hipFuncSetAttribute((const void *)k_loop<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t st = shared_slab ? 0 : stride;
  hipLaunchKernelGGL(k_loop<VAR>, dim3(nwg), dim3(kThreads), sm, 0, N, ld, iters, slab, st, out, nslab);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_loop<VAR>, dim3(nwg), dim3(kThreads), sm, 0, N, ld, iters, slab, st, out, nslab);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * N * N * N * iters * nwg;
  // check one product on the host (workgroup 3, output C = A B of iteration 0 (even))
  double *hc = (double *)malloc(stride * 8);
  int wg = shared_slab ? 0 : 3;
  hipMemcpy(hc, slab + wg * stride, stride * 8, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 200; ++t) {
    int i = (t * 37) % N, j = (t * 91) % N; double s = 0;
    for (int k = 0; k < N; ++k) s += hc[i + k * ld] * hc[(size_t)ld * N + k + j * ld];
    double e = fabs(s - hc[2 * (size_t)ld * N + i + j * ld]); if (e > maxerr) maxerr = e;
  }
  printf("%-14s N=%d %3d slabs %s: %.1f us/product, %.2f TFLOP/s (%.3f of 78.6), err %.2e\n", name, N, shared_slab ? 1 : nslab, shared_slab ? "shared slab" : "own slabs ", ms * 1e3 / iters,
         flops / ms * 1e-9, flops / ms * 1e-9 / 78.6, maxerr);
  hipFree(slab); hipFree(out); free(h); free(hc);
}
int main(int argc, char **argv) {
  int N = argc > 1 ? atoi(argv[1]) : 256;
  run<0>("wg_gemm_big", N, 40, false);
  const int ns[] = {128, 96, 64, 32, 16, 8};
  for (int n : ns) run<0>("wg_gemm_big", N, 40, false, n);
  run<0>("wg_gemm_big", N, 40, true);
  return 0;
}
