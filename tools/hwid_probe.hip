// hwid_probe.hip -- where do the waves of co-resident 4-wave workgroups sit?  ("read HW_REG_HW_ID in the
// 4-wave images and make strip ownership SIMD-aware".)  Launches 2 x #CU persistent workgroups of 256 threads with the LDS
// footprint of the N0 = 40 image (76.5 KB: two per CU) and records per wave HW_ID {wave_id, simd_id, cu_id, se_id, tg_id} and
// the XCC id; prints how many workgroups have their 4 waves on 4 distinct SIMDs and, per CU, which (simd, tg) pairs occur.
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o scratch/bin/hwid_probe && scratch/bin/hwid_probe [lds_bytes] [wgs_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define GETREG(id, off, size) ((((size)-1) << 11) | ((off) << 6) | (id))
__global__ void __launch_bounds__(256) probe(unsigned *out, int spin) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  const unsigned hw = __builtin_amdgcn_s_getreg(GETREG(4, 0, 32));
  const unsigned xcc = __builtin_amdgcn_s_getreg(GETREG(20, 0, 4));
  lds[threadIdx.x] = hw;
  // keep the workgroup resident long enough for the whole grid to be placed
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(64);
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wave) * 2] = hw;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc;
  }
}
int main(int argc, char **argv) {
  const size_t lds = argc > 1 ? atol(argv[1]) : 78336;
  const int per_cu = argc > 2 ? atoi(argv[2]) : 2;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = per_cu * p.multiProcessorCount;
  unsigned *d; hipMalloc(&d, grid * 8 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 200000);  // 2 ms at 100 MHz
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  int distinct = 0, consecutive = 0;
  std::map<unsigned, std::vector<int>> cu_wgs;  // (xcc, se, cu) -> workgroups
  std::map<int, int> tg_hist;
  for (int b = 0; b < grid; ++b) {
    std::set<unsigned> simds;
    bool consec = true;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2];
      simds.insert((hw >> 4) & 3);
      consec = consec && (((hw >> 4) & 3) == (unsigned)w);
    }
    distinct += simds.size() == 4;
    consecutive += consec;
    const unsigned hw = h[b * 8], xcc = h[b * 8 + 1];
    cu_wgs[(xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)].push_back(b);
    tg_hist[(hw >> 16) & 15]++;
  }
  printf("grid %d workgroups of 4 waves, %zu B LDS: waves on 4 distinct SIMDs in %d, wave w on SIMD w in %d\n", grid, lds, distinct, consecutive);
  printf("CUs used %zu; tg_id histogram:", cu_wgs.size());
  for (auto &kv : tg_hist) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  int both_parity = 0, shown = 0;
  std::map<int, int> per_cu_count;
  for (auto &kv : cu_wgs) {
    per_cu_count[(int)kv.second.size()]++;
    std::set<unsigned> par;
    for (int b : kv.second) par.insert((h[b * 8] >> 16) & 1);
    both_parity += par.size() == 2;
    if (shown < 6) {
      printf("  xcc %u se %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 7, kv.first & 15);
      for (int b : kv.second) {
        printf("  wg %4d tg %u simd", b, (h[b * 8] >> 16) & 15);
        for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
      }
      printf("\n");
      ++shown;
    }
  }
  printf("workgroups per CU histogram:");
  for (auto &kv : per_cu_count) printf(" %d:%d", kv.first, kv.second);
  printf("; CUs whose co-resident workgroups differ in tg_id parity: %d of %zu\n", both_parity, cu_wgs.size());
  return 0;
}
