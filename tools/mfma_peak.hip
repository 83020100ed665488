// FP64 MFMA issue rate against workgroup shape, active waves and accumulator register class (VGPR / AGPR form):
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o mfma_peak   (results: profiles/r03_C4_ab.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// MASK: bit w set -> wave w of the workgroup issues MFMAs; NI independent accumulators
template <int NW, int NI>
__global__ void __launch_bounds__(64 * NW) k_peak(int iters, unsigned mask, double *out) {
  d4 acc[NI];
  for (int i = 0; i < NI; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((mask >> wave) & 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < NI; ++i) s += acc[i][0] + acc[i][3];
  if (s == 1.2345) out[blockIdx.x] = s;
}
template <int NW, int NI> void run(int nwg, int iters, unsigned mask) {
  double *out; hipMalloc(&out, nwg * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_peak<NW, NI>), dim3(nwg), dim3(64 * NW), 0, 0, iters, mask, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_peak<NW, NI>), dim3(nwg), dim3(64 * NW), 0, 0, iters, mask, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int act = __builtin_popcount(mask & ((1u << NW) - 1));
  double flops = 2048.0 * NI * iters * act * nwg;
  printf("waves/WG %d (active mask 0x%02x), NI %2d, WGs %4d: %.2f TFLOP/s, %.1f ns per MFMA per active wave\n", NW, mask & ((1u << NW) - 1), NI, nwg, flops / ms * 1e-9,
         ms * 1e6 / ((double)NI * iters * ((nwg + 255) / 256)));
}
int main() {
  run<8, 16>(256, 10000, 0xff);
  run<8, 16>(256, 10000, 0x0f);
  run<8, 16>(256, 10000, 0x55);
  run<8, 16>(256, 10000, 0x33);
  run<8, 16>(256, 10000, 0x01);
  run<4, 16>(256, 10000, 0x0f);
  run<4, 16>(256, 10000, 0x01);
  run<4, 16>(256, 10000, 0x03);
  run<4, 16>(256, 10000, 0x05);
  run<1, 16>(256, 10000, 0x01);
  run<1, 16>(1024, 10000, 0x01);
  run<1, 16>(2048, 10000, 0x01);
  run<8, 4>(256, 40000, 0xff);
  run<8, 4>(256, 40000, 0x0f);
  run<8, 2>(256, 40000, 0x0f);
  run<8, 1>(256, 40000, 0x0f);
  return 0;
}
