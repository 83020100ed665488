// mom_ops.hpp -- the operator-level kernels (include/momcore.h: mom_elemental, mom_doubling, mom_interaction,
// mom_surface_lambertian): one reference operator per launch on [N,N,S] arrays in the reference's own layout, for per-op
// parity tests and for a host that keeps its layer loop.  Compiled once per real type: namespace mom (Float64, momcore.hip)
// and momf (Float32, momcore_f32.hip -- the reference's float_type = Float32 runs through every operator,
// parameters_from_yaml.jl:160, gpu_batched.jl:45-58).
#pragma once
#include "mom_entry.hpp"

namespace MOM_NS {


struct OpArgs {
  DevStreams q;
  int S, m, nd, iface, z_batch;
  const real *tau_sum, *dtau, *varpi, *Zpp, *Zmp;
  real *expk;
  real *added[6];  // r_pm, r_mp, t_mm, t_pp, j0p, j0m
  real *comp[6];
  real *scratch;
  int *info;
};

__device__ __forceinline__ void store_added(const Ctx &c, real *const added[6], size_t pt, bool with_mirror) {
  const int N = c.N, ld = c.ld;
  const size_t NN = (size_t)N * N;
  for (int e = threadIdx.x; e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    const real rv = c.r[i + j * ld], tv = c.t[i + j * ld];
    added[1][NN * pt + e] = rv;
    added[3][NN * pt + e] = tv;
    if (with_mirror) {
      const real s = c.sg[i] * c.sg[j];
      added[0][NN * pt + e] = s * rv;
      added[2][NN * pt + e] = s * tv;
    }
  }
  for (int i = threadIdx.x; i < N; i += kThreads) {
    added[4][(size_t)N * pt + i] = c.jp[i];
    added[5][(size_t)N * pt + i] = c.jm[i];
  }
}

__device__ __forceinline__ void load_added(const Ctx &c, real *const added[6], size_t pt) {
  const int N = c.N, ld = c.ld;
  const size_t NN = (size_t)N * N;
  for (int e = threadIdx.x; e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    c.r[i + j * ld] = added[1][NN * pt + e];
    c.t[i + j * ld] = added[3][NN * pt + e];
  }
  for (int i = threadIdx.x; i < N; i += kThreads) {
    c.jp[i] = added[4][(size_t)N * pt + i];
    c.jm[i] = added[5][(size_t)N * pt + i];
  }
}

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_op_elemental(OpArgs a) {
  const int N = a.q.N;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  const size_t NN = (size_t)N * N;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    const size_t zo = a.z_batch > 1 ? NN * pt : 0;
    El zpp{as_global(a.Zpp) + zo, N, N}, zmp{as_global(a.Zmp) + zo, N, N};
    elemental_build(c, a.q, a.m, a.nd, a.tau_sum[pt], a.dtau[pt], a.varpi[pt], zpp, zmp);
    // the reference leaves r+-/t-- untouched when nd >= 1 (elemental.jl:255-274)
    store_added(c, a.added, pt, a.nd < 1);
    __syncthreads();
  }
}

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_op_doubling(OpArgs a) {
  const int N = a.q.N;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    load_added(c, a.added, pt);
    __syncthreads();
    const real e = doubling_run<LDSM>(c, a.nd, a.expk[pt]);
    if (threadIdx.x == 0) a.expk[pt] = e;
    store_added(c, a.added, pt, true);
    __syncthreads();
  }
  if (threadIdx.x == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_op_interaction(OpArgs a) {
  const int N = a.q.N;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  const size_t NN = (size_t)N * N;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    load_added(c, a.added, pt);
    __syncthreads();
    CompPtrs g = comp_ptrs(a.comp, N, N, pt);  // operator-level arrays: natural pitch
    interaction_core<LDSM, -1>(c, a.iface, g, El{as_global(a.added[0]) + NN * pt, N, N}, El{as_global(a.added[2]) + NN * pt, N, N});
  }
  if (threadIdx.x == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

// surface layer arrays for the operator-level API (lambertian_surface.jl:20-75)
__global__ void k_op_surface_fill(DevStreams q, int S, int m, real albedo, const real *tau_tot, real *r_pm,
                                  real *r_mp, real *t_mm, real *t_pp, real *j0p, real *j0m) {
  const int N = q.N, n = q.nS;
  const size_t NN = (size_t)N * N;
  const size_t pt = blockIdx.x;
  const real rho = 2 * albedo;
  const real att = exp(-tau_tot[pt] / q.mu0);
  const int i_start = n * (q.imu0 - 1), i_end = n * q.imu0;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
    const int j = e / N, i = e - j * N;
    r_mp[NN * pt + e] = (m == 0 && (i % n == 0) && (j % n == 0)) ? rho * (q.mu[j] * q.wt[j]) : 0.0;
    if (m == 0) r_pm[NN * pt + e] = 0.0;  // not reset for m > 0 (:68-73)
    t_pp[NN * pt + e] = (i == j) ? 1.0 : 0.0;
    t_mm[NN * pt + e] = (i == j) ? 1.0 : 0.0;
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const bool in_sun = (i >= i_start) && (i < i_end);
    j0p[(size_t)N * pt + i] = (m == 0) ? (in_sun ? q.I0[i - i_start] : 0.0) * att : 0.0;
    j0m[(size_t)N * pt + i] = (m == 0 && (i % n == 0)) ? (q.mu0 * (rho * q.I0[0])) * att : 0.0;
  }
}

}  // namespace MOM_NS
