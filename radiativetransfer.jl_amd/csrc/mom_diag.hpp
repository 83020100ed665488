// mom_diag.hpp -- diagnostic builds only (make EXTRA=-DMOM_DIAG_STAMPS): s_memtime deltas of the middle
// workgroup, lane 0 of wave 0 (MOM_STAMP) and of wave 4 (MOM_STAMP4), accumulated per code section; each
// translation unit has its own accumulators and reader.  Never part of the shipped library.
#pragma once
#ifdef MOM_DIAG_STAMPS
#include <hip/hip_runtime.h>
static __device__ unsigned long long mom_diag_acc[128];
static __device__ unsigned long long mom_diag_last, mom_diag_last4;
__device__ __forceinline__ unsigned long long mom_diag_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define MOM_STAMP_(id, tid, last)                                        \
  do {                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                   \
    if (threadIdx.x == (tid) && blockIdx.x == (gridDim.x >> 1)) {        \
      unsigned long long n__ = mom_diag_now();                           \
      mom_diag_acc[id] += n__ - last;                                    \
      last = n__;                                                        \
    }                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                   \
  } while (0)
#define MOM_STAMP(id) MOM_STAMP_(id, 0, mom_diag_last)
#define MOM_STAMP4(id) MOM_STAMP_(id, 256, mom_diag_last4)
#endif
