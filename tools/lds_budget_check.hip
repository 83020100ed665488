// tools/lds_budget_check.hip -- host-only check of the LDS budgets the launch code relies on: for every strip
// image size N and every number of Stokes components per stream ns, strip_lds_bytes(N, ns) (image + persistent stream-pair
// tables, mom_kernels.hpp) must fit the CU's 160 KB; where the tables do not fit, ptab_reals must return 0.
// Built twice by tests/test_abi.py: the 8-wave build (default macros) and the 4-wave build (-DMOM_WAVES=4 -DMOM_TJ=3
// -DMOM_NO_STRAIGHT -DMOM_NS=mom4).  Prints one line per (N, ns); exit code 1 on a violation.  Needs no GPU.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "mom_kernels.hpp"

using namespace MOM_NS;

int main() {
  int bad = 0;
  const int sizes[] = {36, 40, 44, 52, 56, 60};
  for (int N : sizes)
    for (int ns = 1; ns <= 4; ++ns) {
      const size_t img = lds_bytes(N, true), all = strip_lds_bytes(N, ns);
      const int nt = ptab_reals(N, ns);
      const bool ok = all <= kLdsPerCU && (nt == 0 || img + (size_t)nt * sizeof(real) <= kLdsPerCU);
      printf("waves=%d N=%d ns=%d image=%zu tables=%d total=%zu %s\n", kWaves, N, ns, img, nt, all, ok ? "ok" : "OVER");
      bad += !ok;
    }
  return bad ? 1 : 0;
}
