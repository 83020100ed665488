// mom_entry.hpp -- the fused per-layer kernels (k_layer, k_surface) and their argument blocks.
// Compiled once per workgroup shape (see mom_device.hpp: MOM_WAVES / MOM_NS); the argument structs are
// plain data with identical layout in every instantiation, so the host passes them across namespaces.
#pragma once
#include "mom_kernels.hpp"

namespace MOM_NS {

// waves per SIMD the layer kernels are compiled for (register budget 512 / MOM_LB_WAVES per lane)
#ifndef MOM_LB_WAVES
#define MOM_LB_WAVES 2
#endif
constexpr int kMaxSweepLayers = 96;
constexpr int kMaxTargets = 20;
struct LayerArgs {
  DevStreams q;
  int S, M, K;      // M: number of moments in THIS launch; m_first: Fourier index of the first of them
  int m_first;
  int nd, iface, first;
  int stagger;      // start offset step between CUs in 100 MHz ticks (persistent strip kernels), 0 = none
  const real *tau, *varpi, *zw, *tau_sum;  // slices of layer z: tau[n], varpi[n], zw[k + K*n], tau_sum[n]
  const real *Zpp, *Zmp;                   // [N,N,K,M] starting at moment m_first
  real *comp[6];                           // R_mp, R_pm, T_pp, T_mm [N,N,S,M]; J0p, J0m [N,S,M], from m_first
  real *scratch;                           // generic mode: per-workgroup slabs
  int *info;
  // sweep mode (Nz_sweep > 0): ONE launch walks all layers of a unit before it moves to the next unit -- the composite
  // blocks a workgroup stored for layer z are the ones it loads for layer z + 1 (same CU, L2-resident), there is one
  // tail per sweep instead of one per layer, and no launch gap.  tau/varpi/zw/tau_sum then point at layer 0 and layer z
  // lies z * S (z * K * S) elements further; nd / iface come from the tables below.
  int Nz_sweep;
  signed char nd_z[kMaxSweepLayers], iface_z[kMaxSweepLayers];
  // several composite targets fed by ONE added layer per layer (rt_kernel_multisensor!, rt_kernel_multisensor.jl:51-112:
  // the added layer of layer iz is built once and joins the composite above or below every sensor).  ntgt > 0: after the
  // doubling of layer z the workgroup walks the targets, act_z[z][t] = 0 nothing, 1 composite_t <- added (a slab's first
  // layer), 2 interaction into composite_t, 3 composite_t <- composite_0 (snapshot of the running top slab at a sensor
  // level; target 0 is processed first).  comp / first are unused then.  Per-layer launches (Nz_sweep = 0) use row 0.
  int ntgt;
  real *tgt[kMaxTargets][6];
  signed char act_z[kMaxSweepLayers][kMaxTargets];
  // lean strip image (mom_lean.hpp): resume[unit] = the layer at which the lean workgroup left the unit (Nz_sweep: finished).
  // The lean kernel writes it; the full image, launched afterwards with the same pointer, starts every unit at that layer
  // (its composite state in global memory is the one after layer resume - 1) and skips the finished ones.  nullptr: all layers.
  int *resume;
};

struct ZMix {
  const gdouble *base;  // Z[:,:,0,m]
  const real *w;        // K weights of this point (r5: staged in LDS with the layer's other scalars, see k_layer)
  int K, N;
  // Z(i,j) = sum_k w[k] Z_k(i,j), accumulated in k order (elemental_build walks the terms)
  __device__ __forceinline__ int terms() const { return K; }
  __device__ __forceinline__ real weight(int k) const { return w[k]; }
  __device__ __forceinline__ real basis(int k, int i, int j) const { return base[i + (size_t)j * N + (size_t)N * N * k]; }
};

// pitch: row pitch of the matrix blocks (N for the operator-level arrays, comp_pitch(N) for the scene-level state)
__device__ __forceinline__ CompPtrs comp_ptrs(real *const comp[6], int N, int pitch, size_t pt) {
  const size_t blk = (size_t)pitch * N;
  CompPtrs g;
  g.R_mp = as_global(comp[0]) + blk * pt;
  g.R_pm = as_global(comp[1]) + blk * pt;
  g.T_pp = as_global(comp[2]) + blk * pt;
  g.T_mm = as_global(comp[3]) + blk * pt;
  g.J0p = as_global(comp[4]) + (size_t)N * pt;
  g.J0m = as_global(comp[5]) + (size_t)N * pt;
  g.ld = pitch;
  return g;
}


// One launch per atmospheric layer: every (spectral point, Fourier moment) pair runs
// elemental -> nd doublings -> interaction with its composite state (rt_kernel!,
// rt_kernel.jl:173-235) inside one workgroup; the added layer never touches HBM.
// KS > 0: N = 4 KS exactly and the strip-chained paths of mom_strip.hpp are compiled in (8-wave LDS build only)
// MT: the multi-target form (a.ntgt composites fed by one added layer, see LayerArgs); a separate image so that the
// single-composite kernels carry none of its code (as a run-time branch it cost the C2 kernel 12 %)
template <bool LDSM, int IFACE, int KS = 0, bool MT = false>
#ifdef MOM_CONST_LAYER_ARGS
// r6 A/B (profiles/r06_C2_ab.txt): the argument block never written -- scratch 3 392 -> 200 B per lane, but 49 spilled VGPRs
// instead of 8 and the N = 60 launch 1.9 % SLOWER (292.8 -> 298.3 ms): the private copy of the block costs nothing that matters
// (it is read through scalar-uniform scratch loads outside the chains), the registers the compiler spends instead do.
__global__ void __launch_bounds__(kThreads, MOM_LB_WAVES) k_layer(const LayerArgs a) {
  const int N = KS > 0 ? 4 * KS : a.q.N;
#else
__global__ void __launch_bounds__(kThreads, MOM_LB_WAVES) k_layer(LayerArgs a) {
  if (KS > 0) a.q.N = 4 * KS;  // the host launches this instantiation only for that size: every dimension folds
  const int N = a.q.N;
#endif
  const size_t total = (size_t)a.S * a.M;
  Ctx c;
#ifdef MOM_DIAG_STAMPS
  if (wg_tid() == 0 && blockIdx.x == (gridDim.x >> 1)) mom_diag_last = mom_diag_now();
  if (wg_tid() == 256 && blockIdx.x == (gridDim.x >> 1)) mom_diag_last4 = mom_diag_now();
#endif
  wg_prologue<LDSM>(c, a.q, N, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  if constexpr (KS > 0) strip_slot_init(c);
  if constexpr (kF64 && KS > 0) {  // the persistent stream-pair tables of the elemental layer (mom_kernels.hpp ptab_reals)
    const int ns = a.q.regular ? a.q.nS : 1, nt = ptab_reals(N, ns), Nq = N / ns;
    if (nt > 0) {
      real *pt = mom_smem + lay_offset_reals(N, LDSM) + lay_cap_reals(N, LDSM);
      for (int e = wg_tid(); e < Nq * Nq; e += kThreads) {
        const int jq = e / Nq, iq = e - jq * Nq;
        const real mui = c.mu[iq * ns], muj = c.mu[jq * ns];
        pt[e] = muj / (mui + muj);
        pt[Nq * Nq + e] = muj / (mui - muj);
        pt[2 * Nq * Nq + e] = (1 / mui) + (1 / muj);
      }
      c.ptab = pt;
      __syncthreads();
    }
  }
  MOM_STAMP(40);
  if (KS > 0 && kWaves == 8 && a.stagger > 0) {
    // persistent workgroups run identical units in lockstep, so every CU would store (and load) its composite
    // blocks in the same microseconds and wait for the whole burst to drain; a start offset per CU (32 phases
    // spread over about one unit time, chosen by the host) spreads that traffic over the unit period
    const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)((blockIdx.x >> 3) & 31) * a.stagger;
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  const int nz = a.Nz_sweep > 0 ? a.Nz_sweep : 1;
  for (size_t pt = blockIdx.x; pt < total; pt += gridDim.x) {
    const int n = (int)(pt % a.S), mrel = (int)(pt / a.S), m = a.m_first + mrel;
    const size_t NN = (size_t)N * N;
    CompPtrs g = comp_ptrs(a.comp, N, comp_pitch(N), pt);
    const int z0 = (a.resume != nullptr) ? a.resume[pt] : 0;  // (workgroup-uniform)
    for (int z = z0; z < nz; ++z) {
      // The per-(point, layer) scalars tau, varpi, tau_sum and the K phase-matrix weights: kLayTab / (3 + K) layers at a time go
      // from their [S, Nz] tables into the LDS tail in ONE batch of independent loads -- a unit pays one memory latency per
      // batch instead of two dependent ones per layer (tau before the first exponential, the weights before the element math;
      // r4 had tried registers one layer ahead: -1.5 % from the ten live registers; LDS costs none)
      // (the host refuses K > kLayTab - 3 scatterer types)
      const int LW = 3 + a.K, LZ = lay_cap_reals(N, LDSM) / LW;
      real *lay = mom_smem + lay_offset_reals(N, LDSM);
      const int zl = z % LZ;
      if (zl == 0 || z == z0) {   // a batch starts at a multiple of LZ; a resumed unit may enter in the middle of one
        const int zb = z - zl, cnt = ((nz - zb < LZ) ? nz - zb : LZ) * LW;
        for (int i = wg_tid(); i < cnt; i += kThreads) {
          const int zz = i / LW, k = i - zz * LW;
          const size_t o = (size_t)n + (size_t)a.S * (zb + zz);
          lay[i] = (k == 0) ? as_global(a.tau)[o] : (k == 1) ? as_global(a.varpi)[o] : (k == 2) ? as_global(a.tau_sum)[o]
                                                                                                : as_global(a.zw)[(size_t)a.K * o + (k - 3)];
        }
        __syncthreads();
      }
      const int nd = a.Nz_sweep > 0 ? a.nd_z[z] : a.nd;
      const int iface = a.Nz_sweep > 0 ? a.iface_z[z] : a.iface;
      // first layer of a slab: the added layer becomes the composite (rt_kernel.jl:227-230); a sweep that CONTINUES a
      // composite already in memory (multi-sensor top slabs, a.first = 0) interacts from its first layer on
      const bool first = (a.Nz_sweep > 0 ? (z == 0) : true) && (a.first != 0);
      const real *ls = lay + zl * LW;
      const real tau = ls[0], varpi = ls[1], tau_sum = ls[2];
      const real dtau = ldexp(tau, -nd);         // τ ./ 2^ndoubl   (rt_kernel.jl:244)
      real expk = exp(-dtau / a.q.mu0);          // init_layer      (rt_kernel.jl:273)
      ZMix zpp{as_global(a.Zpp) + NN * a.K * mrel, ls + 3, a.K, N};
      ZMix zmp{as_global(a.Zmp) + NN * a.K * mrel, ls + 3, a.K, N};
#ifdef MOM_DIAG_STAMPS
      MOM_STAMP(43);
#endif
      elemental_build(c, a.q, m, nd, tau_sum, dtau, varpi, zpp, zmp);
      MOM_STAMP(41);
#ifdef MOM_DIAG_TWICE
      elemental_build(c, a.q, m, nd, tau_sum, dtau, varpi, zpp, zmp);
      MOM_STAMP(44);
#endif
#ifdef MOM_QPREFETCH
      expk = doubling_run<LDSM, KS>(c, nd, expk, (!MT && !first && iface == 3) ? &g : nullptr);
#else
      expk = doubling_run<LDSM, KS>(c, nd, expk);
#endif
      MOM_STAMP(30);
      if constexpr (MT) {
        const int zr = a.Nz_sweep > 0 ? z : 0;
        for (int t = 0; t < a.ntgt; ++t) {
          const int act = a.act_z[zr][t];
          if (act == 0) continue;
          CompPtrs gt = comp_ptrs(a.tgt[t], N, comp_pitch(N), pt);
          if (act == 1) {
            store_added_as_composite(c, gt);
            __syncthreads();
          } else if (act == 2) {
            interaction_core<LDSM, IFACE, KS>(c, iface, gt, ElSigP{c.r, c.sg, c.ld}, ElSigP{c.t, c.sg, c.ld});
          } else {
            copy_composite(c, comp_ptrs(a.tgt[0], N, comp_pitch(N), pt), gt);
            __syncthreads();
          }
        }
      } else {
        if (first) {
          store_added_as_composite(c, g);
          __syncthreads();
          MOM_STAMP(42);
        } else {
          interaction_core<LDSM, IFACE, KS>(c, iface, g, ElSigP{c.r, c.sg, c.ld}, ElSigP{c.t, c.sg, c.ld});
          MOM_STAMP(45);
        }
      }
    }
  }
  if (wg_tid() == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

struct SurfArgs {
  DevStreams q;
  int S, iface;
  real albedo;
  const real *tau_tot;  // [S]
  real *comp[6];        // slices of THIS moment
  real *hdrJ;           // [N,S]   hdr_J0- of interaction_hdrf! for this moment
  real *bhr_uw, *bhr_dw;  // [nS_out,S] (nS_out = the caller's nStokes; rows >= q.nS stay zero); written for m = 0
  int nS_out;
  real *scratch;
  int *info;
  // surface type (include/momcore.h, mom_scene_set_surface): 0 = LambertianSurfaceScalar, 1 = BRDF matrix of this
  // moment (rpv / Ross-Li ...: create_surface_layer!(::AbstractSurfaceType), rpv_surface.jl:20-66), 2 =
  // LambertianSurfaceLegendre (spectrally varying albedo, lambertian_surface.jl:77-138)
  int kind, m;
  const real *Rsurf;        // kind 1: [N,N] rho_m (column-major), factor 2 for m = 0 included
  const real *albedo_spec;  // kind 2: [S]
};

// The surface as an added layer + the closing interaction (rt_run.jl:169-185), then interaction_hdrf!.
// LambertianSurfaceScalar/Legendre: m = 0 only (for m > 0 the surface layer is r = 0, j = 0 and the interaction is the
// identity on everything post-processing reads, so no launch is made); BRDF surfaces: one launch per moment.
template <bool LDSM>
__global__ void __launch_bounds__(kThreads, 2) k_surface(SurfArgs a) {
  const int N = a.q.N, n = a.q.nS;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  const int ld = c.ld;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    const real att = exp(-a.tau_tot[pt] / a.q.mu0);
    const int i_start = n * (a.q.imu0 - 1), i_end = n * a.q.imu0;
    if (a.kind == 1) {
      // R_surf = rho_m ; j0+ = I0 exp(-tau/mu0) at the sun rows ; j0- = mu0 (R_surf I0N) exp(-tau/mu0) ;
      // r-+ = R_surf Diagonal(qp_muN .* wt_muN)                                   (rpv_surface.jl:48-62)
      for (int e = wg_tid(); e < N * N; e += kThreads) {
        int i, j;
        c.fd.split(e, i, j);
        c.r[i + j * ld] = a.Rsurf[i + (size_t)N * j] * (c.mu[j] * c.wt[j]);
        c.t[i + j * ld] = (i == j) ? 1.0 : 0.0;
      }
      for (int i = wg_tid(); i < N; i += kThreads) {
        const bool in_sun = (i >= i_start) && (i < i_end);
        real rI = 0.0;
        for (int k = 0; k < n; ++k) rI += a.Rsurf[i + (size_t)N * (i_start + k)] * a.q.I0[k];
        c.jp[i] = (in_sun ? a.q.I0[i - i_start] : 0.0) * att;
        c.jm[i] = (a.q.mu0 * rI) * att;
      }
    } else {
      const real rho = 2 * ((a.kind == 2) ? a.albedo_spec[pt] : a.albedo);  // lambertian_surface.jl:37 / :97
      for (int e = wg_tid(); e < N * N; e += kThreads) {
        int i, j;
        c.fd.split(e, i, j);
        c.r[i + j * ld] = ((i % n == 0) && (j % n == 0)) ? rho * (c.mu[j] * c.wt[j]) : 0.0;  // :41-43,:58
        c.t[i + j * ld] = (i == j) ? 1.0 : 0.0;
      }
      for (int i = wg_tid(); i < N; i += kThreads) {
        const bool in_sun = (i >= i_start) && (i < i_end);
        if (a.kind == 2) {
          c.jp[i] = 0.0;                                                              // :112 (the Legendre type sets j0+ = 0)
          c.jm[i] = (i % n == 0) ? (a.q.mu0 * a.q.I0[0]) * (rho * att) : 0.0;         // :114
        } else {
          c.jp[i] = (in_sun ? a.q.I0[i - i_start] : 0.0) * att;                 // :55
          c.jm[i] = (i % n == 0) ? (a.q.mu0 * (rho * a.q.I0[0])) * att : 0.0;  // :56
        }
      }
    }
    __syncthreads();
    CompPtrs g = comp_ptrs(a.comp, N, comp_pitch(N), pt);
    interaction_core<LDSM, -1>(c, a.iface, g, ElZero{}, ElEye{N});
    // interaction_hdrf! (CoreKernel/interaction_hdrf.jl:9-45): hdr_J0- = r-+_surf J0+ + j0-_surf with the
    // composite J0+ AFTER the surface interaction (still in c.Jp), then the m = 0 flux sums of the BHR
    wg_matvec(c, ElP{c.r, ld}, c.Jp, c.v1);
    for (int i = wg_tid(); i < N; i += kThreads) {
      const real hj = c.v1[i] + c.jm[i];
      c.v1[i] = hj;
      a.hdrJ[(size_t)N * pt + i] = hj;
    }
    __syncthreads();
    if (a.m == 0 && wg_tid() < a.nS_out) {
      const int k = wg_tid();
      real up = 0.0, dw = 0.0;
      if (k < n)  // components beyond the reduced problem's (I,Q) have exactly zero sums for m = 0
        for (int j = k; j < N; j += n) {
          up += c.v1[j] * c.wt[j] * c.mu[j];
          dw += c.Jp[j] * c.wt[j] * c.mu[j];
        }
      a.bhr_uw[k + (size_t)a.nS_out * pt] = up;
      a.bhr_dw[k + (size_t)a.nS_out * pt] = dw + c.jp[i_start] * c.mu[i_start];
    }
    __syncthreads();
  }
  if (wg_tid() == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

// Fields at a sensor between two composite slabs (interlayer_flux_helper!(::noRS), CoreKernel/interlayer_flux.jl:7-24):
//   dwJ = (I - topR+- botR-+)^-1 (topJ0+ + topR+- botJ0-)      uwJ = (I - botR-+ topR+-)^-1 (botJ0- + botR-+ topJ0+)
// per (spectral point, moment) unit; top = the layers above the sensor, bot = the layers below it and the surface.
struct InterArgs {
  DevStreams q;
  int S, M;
  real *top[6], *bot[6];  // scene-level composite states [.,.,S,M]
  real *dwJ, *uwJ;        // [N,S,M]
  real *scratch;
  int *info;
};

template <bool LDSM>
__global__ void __launch_bounds__(kThreads, 2) k_interlayer(InterArgs a) {
  const int N = a.q.N;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  const int ld = c.ld;
  const size_t units = (size_t)a.S * a.M;
  for (size_t pt = blockIdx.x; pt < units; pt += gridDim.x) {
    const CompPtrs gt = comp_ptrs(a.top, N, comp_pitch(N), pt), gb = comp_ptrs(a.bot, N, comp_pitch(N), pt);
    const int cl = gt.ld;
    wg_copy_mat(N, c.fd, gt.R_pm, cl, c.r, ld);  // r = topR+-
    wg_copy_mat(N, c.fd, gb.R_mp, cl, c.t, ld);  // t = botR-+
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.Jp[i] = gt.J0p[i];
      c.Jm[i] = gb.J0m[i];
    }
    __syncthreads();
    for (int dir = 0; dir < 2; ++dir) {
      // dir 0 (down): X = topR+-, Y = botR-+, x = topJ0+, y = botJ0-;  dir 1 (up): the roles exchanged
      real *X = dir ? c.t : c.r, *Y = dir ? c.r : c.t;
      const real *x = dir ? c.Jm : c.Jp, *y = dir ? c.Jp : c.Jm;
      real beta2;
      {
        real *Q = c.Q;
        real ss = 0.0;
        wg_gemm<false, !LDSM>(N, ElP{X, ld}, ElP{Y, ld}, [=, &ss](int i, int j, real v) {
          Q[i + j * ld] = v;
          ss += v * v;
        });
        wg_sumsq_put(c, ss);
        __syncthreads();
        beta2 = wg_sumsq_get(c);
      }
      times_inv<LDSM>(c, ElEye{N}, c.Q, c.P, beta2);  // P = (I - X Y)^-1
      wg_matvec(c, ElP{X, ld}, y, c.v1);
      for (int i = wg_tid(); i < N; i += kThreads) c.v1[i] = x[i] + c.v1[i];
      __syncthreads();
      wg_matvec(c, ElP{c.P, ld}, c.v1, c.v2);
      real *out = (dir ? a.uwJ : a.dwJ) + (size_t)N * pt;
      for (int i = wg_tid(); i < N; i += kThreads) out[i] = c.v2[i];
      __syncthreads();
    }
  }
  if (wg_tid() == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

// top <- top (+) bot for two composite slabs, bot below top: the adding equations of interaction_helper!(::ScatteringInterface_11)
// (CoreKernel/interaction.jl:69-117) with the lower slab in the role of the added layer -- its R-+, T++ and sources go to the
// unit's buffers, its R+- and T-- are read through element functors.  Used by mom_rt_run_multisensor: the slab below sensor k is
// the segment between sensors k and k + 1 (built in the shared sweep) joined to the slab below sensor k + 1.
template <bool LDSM>
__global__ void __launch_bounds__(kThreads, 2) k_combine(InterArgs a) {
  const int N = a.q.N;
  Ctx c;
  wg_prologue<LDSM>(c, a.q, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  const int ld = c.ld;
  const size_t units = (size_t)a.S * a.M;
  for (size_t pt = blockIdx.x; pt < units; pt += gridDim.x) {
    const CompPtrs gt = comp_ptrs(a.top, N, comp_pitch(N), pt), gb = comp_ptrs(a.bot, N, comp_pitch(N), pt);
    const int cl = gb.ld;
    wg_copy_mat(N, c.fd, gb.R_mp, cl, c.r, ld);
    wg_copy_mat(N, c.fd, gb.T_pp, cl, c.t, ld);
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.jp[i] = gb.J0p[i];
      c.jm[i] = gb.J0m[i];
    }
    __syncthreads();
    interaction_core<LDSM, -1>(c, 3, gt, El{gb.R_pm, cl, N}, El{gb.T_mm, cl, N});
  }
  if (wg_tid() == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

}  // namespace MOM_NS
