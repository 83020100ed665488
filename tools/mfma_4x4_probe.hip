// v_mfma_f64_4x4x4_4b on gfx950: (1) issue rate against v_mfma_f64_16x16x4 (both 32 FLOP/clk/SIMD at peak?), (2) the register layout
// of A, B and D (which lane holds which element of which of the four 4 x 4 blocks), (3) the A-broadcast controls cbsz / abid.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_4x4_probe.hip -o scratch/bin/mfma_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NI, int SHAPE>
__global__ void __launch_bounds__(256) k_rate(int iters, double *out) {
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  double s = 0;
  if (SHAPE == 4) {
    double acc[NI];
    for (int i = 0; i < NI; ++i) acc[i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NI; ++i) s += acc[i];
  } else {
    d4 acc[NI];
    for (int i = 0; i < NI; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NI; ++i) s += acc[i][0] + acc[i][3];
  }
  if (s == 1.2345) out[blockIdx.x] = s;
}
template <int NI, int SHAPE> void rate(int iters) {
  double *out; hipMalloc(&out, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<NI, SHAPE>), dim3(256), dim3(256), 0, 0, iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<NI, SHAPE>), dim3(256), dim3(256), 0, 0, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (SHAPE == 4 ? 512.0 : 2048.0) * NI * iters * 4 * 256;
  printf("%s NI %2d: %.2f TFLOP/s, %.1f ns per instruction per wave (one wave per SIMD)\n", SHAPE == 4 ? "4x4x4_4b " : "16x16x4  ", NI,
         fl / ms * 1e-9, ms * 1e6 / ((double)NI * iters));
}

// layout probe: D = A x B per block with A, B filled so that the product identifies the mapping
__global__ void k_layout(const double *A, const double *B, double *D, int cbsz, int abid) {
  const int l = threadIdx.x;
  double d;
  switch (cbsz * 4 + abid) {  // immediates
    case 0: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0); break;
    case 8: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, 0, 0); break;
    case 9: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, 1, 0); break;
    case 10: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, 2, 0); break;
    case 11: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, 3, 0); break;
    case 4: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 1, 0, 0); break;
    case 5: d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 1, 1, 0); break;
    default: d = -1; break;
  }
  D[l] = d;
}

int main() {
  rate<16, 16>(20000);
  rate<16, 4>(80000);
  rate<4, 4>(80000);
  rate<2, 4>(80000);
  rate<1, 4>(80000);
  double hA[64], hB[64], hD[64], *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  // unit probes: A has a single 1 at lane la, B a single 1 at lane lb -> D has a 1 where (block, i, k) x (block, k, j) meet
  printf("layout (cbsz 0): for each A lane la and B lane lb, the D lanes that become 1\n");
  int Arow[64], Acol[64], Ablk[64], Brow[64], Bcol[64], Bblk[64];
  for (int i = 0; i < 64; ++i) Arow[i] = Acol[i] = Ablk[i] = Brow[i] = Bcol[i] = Bblk[i] = -1;
  // all-ones B: D lanes hit by A lane la tell (block, row i) of la; all-ones A: D lanes hit by B lane lb tell (block, col j)
  for (int la = 0; la < 64; ++la) {
    for (int i = 0; i < 64; ++i) { hA[i] = (i == la); hB[i] = 1.0; }
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    printf("A lane %2d -> D lanes:", la);
    for (int i = 0; i < 64; ++i) if (hD[i] != 0) printf(" %d", i);
    printf("\n");
  }
  for (int lb = 0; lb < 64; ++lb) {
    for (int i = 0; i < 64; ++i) { hB[i] = (i == lb); hA[i] = 1.0; }
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    printf("B lane %2d -> D lanes:", lb);
    for (int i = 0; i < 64; ++i) if (hD[i] != 0) printf(" %d", i);
    printf("\n");
  }
  // which (A lane, B lane) pairs interact: k index matching, block 0 only (lanes 0..15 if blocks are lane groups of 16)
  printf("k matching: A lane la x B lane lb gives a nonzero D (first 16 lanes each)\n");
  for (int la = 0; la < 16; ++la) {
    printf("A lane %2d with B lanes:", la);
    for (int lb = 0; lb < 64; ++lb) {
      for (int i = 0; i < 64; ++i) { hA[i] = (i == la); hB[i] = (i == lb); }
      hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0);
      hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; ++i) if (hD[i] != 0) printf(" %d->D%d", lb, i);
    }
    printf("\n");
  }
  // broadcast: cbsz = 2, abid = q: every block uses the A of block q?
  for (int q = 0; q < 4; ++q) {
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + ((i >> 2) & 3); hB[i] = 1.0; }   // A block b (lane bits 2..3) filled with b + 1
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 2, q);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    printf("cbsz 2 abid %d: D of blocks 0..3 (lanes 0,4,8,12) = %g %g %g %g (4 x the A block value in use)\n", q, hD[0], hD[4], hD[8], hD[12]);
  }
  for (int q = 0; q < 2; ++q) {
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + ((i >> 2) & 3); hB[i] = 1.0; }
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1, q);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    printf("cbsz 1 abid %d: D of blocks 0..3 = %g %g %g %g\n", q, hD[0], hD[4], hD[8], hD[12]);
  }
  {
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + ((i >> 2) & 3); hB[i] = 1.0; }
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    printf("cbsz 0       : D of blocks 0..3 = %g %g %g %g\n", hD[0], hD[4], hD[8], hD[12]);
  }
  return 0;
}
