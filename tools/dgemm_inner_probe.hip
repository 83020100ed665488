// What the inner loop of momd::k_dgemm can reach by itself: 256-thread workgroups, per k-step one B read + four A reads from LDS
// (ds_read_b64) and four v_mfma_f64_16x16x4, operands staged ONCE (no global loads, no LDS stores, no barriers in the loop).
// Operands: a 17-value pattern or random mantissas (data-dependent power: MI355X_MICROARCH.md).  Variants: waves per SIMD (workgroups per CU via LDS padding), and k-steps unrolled 1 / 8.
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/dgemm_inner_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KC = 32, LDA = 66, LDB = 34;
template <int UNROLL>
__global__ void __launch_bounds__(256) k_probe(double *out, int iters, int rnd) {
  extern __shared__ double sm[];
  double *As = sm, *Bs = sm + KC * LDA;
  for (int e = threadIdx.x; e < KC * LDA + 64 * LDB; e += 256) {
    unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1 + 977 * blockIdx.x);   // rnd: full random mantissas in [1, 2) - 1.5
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    sm[e] = rnd ? (__longlong_as_double((long long)((x >> 12) | 0x3FF0000000000000ull)) - 1.5) * 0.05 : 1e-3 * (e % 17);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lq = lane >> 4, lr = lane & 15;
  d4 acc[4];
  for (int tb = 0; tb < 4; ++tb) acc[tb] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll UNROLL
    for (int kk = 0; kk < KC; kk += 4) {
      const double bv = Bs[(kk + lq) + (16 * w + lr) * LDB];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const double av = As[(16 * tb + lr) + (kk + lq) * LDA];
        acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc[tb], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int tb = 0; tb < 4; ++tb) s += acc[tb][0] + acc[tb][1] + acc[tb][2] + acc[tb][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int UNROLL>
void run(int wg_per_cu, double *out, int rnd) {
  const size_t lds = 160 * 1024 / wg_per_cu - 512;   // forces wg_per_cu workgroups per CU
  hipFuncSetAttribute((const void *)k_probe<UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wg_per_cu, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_probe<UNROLL><<<grid, 256, lds>>>(out, 10, rnd);
  hipEventRecord(e0);
  k_probe<UNROLL><<<grid, 256, lds>>>(out, iters, rnd);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * (KC / 4) * 4 * (16.0 * 16 * 4 * 2);
  printf("%s operands, unroll %d, %d workgroup(s) per CU (%d waves per SIMD): %.1f TFLOP/s = %.2f of 78.6\n", rnd ? "random" : "patterned", UNROLL, wg_per_cu, wg_per_cu, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
}
int main() {
  double *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  for (int rnd : {0, 1})
    for (int w : {1, 2, 4}) { run<1>(w, out, rnd); run<8>(w, out, rnd); }
  return 0;
}
