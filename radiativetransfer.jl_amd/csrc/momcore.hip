// momcore.hip -- __global__ kernels and the C ABI (include/momcore.h) of libmomcore.so.
// gfx950 (MI355X) only.  See DESIGN.md for the data layout and the kernel inventory.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <algorithm>
#include <vector>

#include <rccl/rccl.h>  // types and enums only: the library itself is dlopen'ed by mom_comm_init

#include "momcore.h"

#include "mom_diag.hpp"
#include "mom_entry.hpp"
#include "mom_ops.hpp"
#include "mom_host.hpp"
#include "mom_rrs.hpp"

using namespace mom;

// =========================================================================================
// kernels
// =========================================================================================

// postprocessing_vza! (postprocessing_vza.jl:9-60, SFI branch) and postprocessing_vza_hdrf! (:63-93), all
// moments in m order.  With the m = 0 reduction (see mom_scene_set) the m = 0 sources live in their own
// arrays with N0 = nS0 * Nquad rows; Stokes components >= nS0 get no m = 0 contribution (it is exactly 0).
struct PostArgs {
  int N, nS, S, M, nVza, red0, N0, nS0;
  int hdr_all;   // BRDF surfaces: hdr_J0- exists for every moment (hdrJm), not only m = 0
  int zeroT_hi;  // LambertianSurfaceLegendre: t++ = t-- = 0 for m > 0 (lambertian_surface.jl:131-132) -> J0+ = 0 there
  const int *node;
  const double *cos_mphi, *sin_mphi;
  const double *J0p, *J0m;    // [N,S,M] (moment 0 slice unused when red0)
  const double *J0p0, *J0m0;  // [N0,S] when red0
  const double *hdrJ;         // m = 0: [N,S] or [N0,S] when red0
  const double *hdrJm;        // hdr_all: [N,S,M] (slot 0 unused)
  double *R, *T, *hdr;
};
__global__ void k_postprocess(PostArgs a) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.nVza * a.nS * a.S;
  if (idx >= total) return;
  const int v = (int)(idx % a.nVza);
  const int k = (int)((idx / a.nVza) % a.nS);
  const size_t s = idx / ((size_t)a.nVza * a.nS);
  const int row = (a.node[v] - 1) * a.nS + k;
  double r = 0.0, t = 0.0, h = 0.0;
  for (int m = 0; m < a.M; ++m) {
    const double weight = (m == 0) ? 0.5 : 1.0;
    const double cs = weight * ((k < 2) ? a.cos_mphi[v + (size_t)a.nVza * m] : a.sin_mphi[v + (size_t)a.nVza * m]);
    if (m == 0 && a.red0) {
      if (k < a.nS0) {
        const size_t o = (size_t)(a.node[v] - 1) * a.nS0 + k + (size_t)a.N0 * s;
        r += cs * a.J0m0[o];
        t += cs * a.J0p0[o];
        h += cs * a.hdrJ[o];
      }
    } else {
      const size_t o = row + (size_t)a.N * (s + (size_t)a.S * m);
      r += cs * a.J0m[o];
      if (!(a.zeroT_hi && m > 0)) t += cs * a.J0p[o];
      if (m == 0) h += cs * a.hdrJ[row + (size_t)a.N * s];
      else if (a.hdr_all) h += cs * a.hdrJm[o];
    }
  }
  a.R[idx] = r;
  a.T[idx] = t;
  a.hdr[idx] = h;  // Lambertian surfaces: only m = 0 contributes (r-+ = 0, j0- = 0 for m > 0)
}

// operator-level kernels: mom_ops.hpp (shared with the Float32 build)

struct BlasArgs {
  int N, S;
  const double *A, *B;
  double *C;
  double *scratch;
  int *info;
};

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_batch_inv(BlasArgs a) {
  const int N = a.N;
  Ctx c;
  make_ctx<LDSM>(c, N, 1, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  zero_padding<LDSM>(c);
  if (threadIdx.x == 0) *c.bad = 0;
  __syncthreads();
  const size_t NN = (size_t)N * N;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    wg_copy_mat(N, c.fd, a.A + NN * pt, N, c.P, c.ld);
    __syncthreads();
    if (N <= 64) wg_inverse_reg(N, c.P, c.ld, c.part, c.prow, c.ipiv, c.bad);
    else wg_inverse(N, c.fd, c.P, c.ld, c.prow, c.pcol, c.rowk, c.ipiv, c.sh, c.bad);
    wg_copy_mat(N, c.fd, c.P, c.ld, a.C + NN * pt, N);
    __syncthreads();
  }
  if (threadIdx.x == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_batched_mul(BlasArgs a) {
  const int N = a.N;
  Ctx c;
  make_ctx<LDSM>(c, N, 1, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  zero_padding<LDSM>(c);
  __syncthreads();
  const size_t NN = (size_t)N * N;
  const int ld = c.ld;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    wg_copy_mat(N, c.fd, a.A + NN * pt, N, c.P, ld);
    wg_copy_mat(N, c.fd, a.B + NN * pt, N, c.Q, ld);
    __syncthreads();
    double *C = a.C + NN * pt;
    wg_gemm<false>(N, ElP{c.P, ld}, ElP{c.Q, ld}, [=](int i, int j, double v) { C[i + (size_t)j * N] = v; });
    __syncthreads();
  }
}

// elemental_inelastic!(RS_type::RRS, ...) (CoreKernel/elemental_inelastic.jl:23-91): the single-scattering layer of the
// rotational-Raman source operators, one thread per element (i, j, n1, dn) of the 4-D arrays -- get_elem_rt_RRS! (:93-160),
// get_elem_rt_SFI_RRS! (:320-382), apply_D_elemental_RRS! (:384-402; the SFI D kernel :404-412 / :478-490 changes nothing
// for any ndoubl).  n0 = n1 + i_l1l0[dn] is the incident-wavelength index (0-based here), dtau the elemental optical
// thickness per spectral point.  Entries whose n0 falls off the grid are written as zeros (the reference leaves the
// freshly allocated zeros in place).  HBM-write bound: 4 N^2 + 2 N doubles per (n1, dn).
struct RrsArgs {
  DevStreams q;
  int S, nR, m, nd, strict;
  const int *i_l1l0;                                                     // [nR]
  const double *varpi_l1l0, *fscatt, *tau_sum, *dtau, *varpi, *Zpp, *Zmp;  // [nR], [S] x4, [N,N] x2
  double *ier_mp, *iet_pp, *ier_pm, *iet_mm, *ieJ0p, *ieJ0m;              // [N,N,S,nR] x4, [N,S,nR] x2
};

__global__ void __launch_bounds__(256) k_elemental_rrs(RrsArgs a) {
#pragma clang fp contract(off)
  const int N = a.q.N, n = a.q.nS;
  const size_t NN = (size_t)N * N, total = NN * a.S * a.nR;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e % N), j = (int)((e / N) % N);
  const size_t u = e / NN;                       // n1 + S dn
  const int n1 = (int)(u % a.S), dn = (int)(u / a.S);
  const int n0 = n1 + a.i_l1l0[dn];
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const double mui = a.q.mu[i], muj = a.q.mu[j], wj = a.q.wt[j] / wdiv;
  double r = 0.0, t = 0.0;
  const bool in = (n0 >= 0) && (n0 < a.S);
  if (in && wj > 1.e-8) {
    const double d1 = a.dtau[n1], d0 = a.dtau[n0];
    const double pre = a.varpi_l1l0[dn] * a.varpi[n0] * a.fscatt[n0];
    // :118-120
    r = a.fscatt[n0] * a.varpi_l1l0[dn] * a.varpi[n0] * a.Zmp[i + (size_t)N * j] * (1 / ((mui / muj) + (d1 / d0))) *
        (1 - exp(-((d1 / mui) + (d0 / muj)))) * wj;
    if (mui == muj) {
      if (i == j) {
        const double wi = a.q.wt[i] / wdiv;
        if (fabs(d0 - d1) > 1.e-6)   // :130-134
          t = pre * a.Zpp[i + (size_t)N * i] * wi * (exp(-d0 / mui) - exp(-d1 / mui)) / (1 - (d1 / d0));
        else                          // :136-138
          t = pre * a.Zpp[i + (size_t)N * i] * wi * (1 - exp(-d0 / muj));
      }
    } else {                          // :147-151
      t = pre * a.Zpp[i + (size_t)N * j] * (1 / ((mui / muj) - (d1 / d0))) * wj * (exp(-d1 / mui) - exp(-d0 / muj));
    }
  }
  // apply_D_elemental_RRS! (:384-402), component rule of SURVEY Q1
  const int ci = a.strict ? ((i + 1) % n) : (i % n) + 1, cj = a.strict ? ((j + 1) % n) : (j % n) + 1;
  if (a.nd < 1) {
    const double s = (((ci <= 2) && (cj <= 2)) || ((ci > 2) && (cj > 2))) ? 1.0 : -1.0;
    a.ier_pm[e] = s * r;
    a.iet_mm[e] = s * t;
  } else {
    if (ci > 2) r = -r;
    a.ier_pm[e] = 0.0;  // left untouched by the reference for ndoubl >= 1 (apply_D_matrix_IE! fills them after doubling)
    a.iet_mm[e] = 0.0;
  }
  a.ier_mp[e] = r;
  a.iet_pp[e] = t;
  if (j == 0) {  // source vectors: one thread per (i, n1, dn)                                   (:320-382)
    const int i_start = n * (a.q.imu0 - 1), i_end = n * a.q.imu0;  // 0-based [i_start, i_end)
    double jp = 0.0, jm = 0.0;
    if (in) {
      const double d1 = a.dtau[n1], d0 = a.dtau[n0], mus = a.q.mu[i_start];
      double zpI = 0.0, zmI = 0.0;
      for (int ii = i_start; ii < i_end; ++ii) {
        zpI += a.Zpp[i + (size_t)N * ii] * a.q.I0[ii - i_start];
        zmI += a.Zmp[i + (size_t)N * ii] * a.q.I0[ii - i_start];
      }
      const double pre = a.varpi_l1l0[dn] * a.varpi[n0] * a.fscatt[n0];
      if (i >= i_start && i < i_end) {
        if (fabs(d0 - d1) > 1.e-6) jp = (exp(-d0 / mui) - exp(-d1 / mui)) / ((d1 / d0) - 1) * pre * zpI * wct02;  // :350-353
        else jp = wct02 * pre * zpI * (1 - exp(-d0 / mus));                                                        // :355-357
      } else {                                                                                                     // :361-364
        jp = wct02 * pre * zpI * (1 / ((mui / mus) - (d1 / d0))) * (exp(-d1 / mui) - exp(-d0 / mus));
      }
      jm = wct02 * pre * zmI * (1 / ((mui / mus) + (d1 / d0))) * (1 - exp(-((d1 / mui) + (d0 / mus))));            // :368-370
      const double att = exp(-a.tau_sum[n0] / mus);                                                               // :371-372
      jp *= att;
      jm *= att;
    }
    if (a.nd >= 1) jm = a.q.D[i % n] * jm;  // :374-376
    const size_t o = i + (size_t)N * u;
    a.ieJ0p[o] = jp;
    a.ieJ0m[o] = jm;
  }
}

// batched_mul / batch_inv! on ForwardDiff.Dual arrays (gpu_batched.jl:100-150): values [N,N,S] and P partials [N,N,S,P].
//   mul:  C = A B,      dC_i = A dB_i + dA_i B          inv:  X = A^-1,   dX_i = -X dA_i X
// One workgroup per batch item keeps the values in LDS (slab) while it walks the partials.
struct DualArgs {
  int N, S, P;
  const double *A, *dA, *B, *dB;
  double *C, *dC;
  double *scratch;
  int *info;
};

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_batched_mul_dual(DualArgs a) {
  const int N = a.N;
  Ctx c;
  make_ctx<LDSM>(c, N, 1, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  zero_padding<LDSM>(c);
  __syncthreads();
  const size_t NN = (size_t)N * N;
  const int ld = c.ld;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    wg_copy_mat(N, c.fd, a.A + NN * pt, N, c.P, ld);
    wg_copy_mat(N, c.fd, a.B + NN * pt, N, c.Q, ld);
    __syncthreads();
    double *C = a.C + NN * pt;
    wg_gemm<false, !LDSM>(N, ElP{c.P, ld}, ElP{c.Q, ld}, [=](int i, int j, double v) { C[i + (size_t)j * N] = v; });
    for (int ip = 0; ip < a.P; ++ip) {
      const size_t o = NN * (pt + (size_t)a.S * ip);
      __syncthreads();
      wg_copy_mat(N, c.fd, a.dB + o, N, c.r, ld);
      wg_copy_mat(N, c.fd, a.dA + o, N, c.t, ld);
      __syncthreads();
      double *dC = a.dC + o;
      wg_gemm<false, !LDSM>(N, ElP{c.P, ld}, ElP{c.r, ld}, [=](int i, int j, double v) { dC[i + (size_t)j * N] = v; });
      __syncthreads();
      wg_gemm<false, !LDSM>(N, ElP{c.t, ld}, ElP{c.Q, ld},
                            [=](int i, int j, double v) { dC[i + (size_t)j * N] = dC[i + (size_t)j * N] + v; });
    }
    __syncthreads();
  }
}

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_batch_inv_dual(DualArgs a) {
  const int N = a.N;
  Ctx c;
  make_ctx<LDSM>(c, N, 1, mom_smem, LDSM ? nullptr : a.scratch + (size_t)blockIdx.x * kGenericBufs * mat_elems(N));
  zero_padding<LDSM>(c);
  if (threadIdx.x == 0) *c.bad = 0;
  __syncthreads();
  const size_t NN = (size_t)N * N;
  const int ld = c.ld;
  for (size_t pt = blockIdx.x; pt < (size_t)a.S; pt += gridDim.x) {
    wg_copy_mat(N, c.fd, a.A + NN * pt, N, c.P, ld);
    __syncthreads();
    if (N <= 64) wg_inverse_reg(N, c.P, ld, c.part, c.prow, c.ipiv, c.bad);
    else wg_inverse(N, c.fd, c.P, ld, c.prow, c.pcol, c.rowk, c.ipiv, c.sh, c.bad);
    wg_copy_mat(N, c.fd, c.P, ld, a.C + NN * pt, N);
    for (int ip = 0; ip < a.P; ++ip) {
      const size_t o = NN * (pt + (size_t)a.S * ip);
      __syncthreads();
      wg_copy_mat(N, c.fd, a.dA + o, N, c.Q, ld);
      __syncthreads();
      double *r = c.r;
      wg_gemm<false, !LDSM>(N, ElP{c.P, ld}, ElP{c.Q, ld}, [=](int i, int j, double v) { r[i + j * ld] = v; });
      __syncthreads();
      double *dX = a.dC + o;
      wg_gemm<false, !LDSM>(N, ElP{c.r, ld}, ElP{c.P, ld}, [=](int i, int j, double v) { dX[i + (size_t)j * N] = -v; });
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

// =========================================================================================
// host side
// =========================================================================================

// momcore_w4.hip: the same kernels built for 4-wave workgroups (2 workgroups per CU when the operators are
// small enough for two LDS images: the m = 0 (I,Q) sub-problem of N = 60 is N0 = 40 -> 77 KB).
size_t mom4_lds_bytes(int N, bool lds_mats);
size_t mom4_strip_lds_bytes(int N, int ns);
hipError_t mom4_launch_layer(const void *layer_args, int iface, bool lds, int grid, size_t smem, hipStream_t st);
// momcore_gen.hip: the general layer kernels k_layer<LDSM, IFACE> of the 8-wave build
hipError_t mom_gen_launch_layer(const void *layer_args, int iface, bool lds, int grid, size_t smem, hipStream_t st);
// momcore_strip.hip, one object per operator size N = 4 KS
hipError_t mom_strip9_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom_strip10_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom_strip9_launch_lean(const void *layer_args, int grid, hipStream_t st);   // momcore_strip.hip with mom_lean.hpp
hipError_t mom_strip10_launch_lean(const void *layer_args, int grid, hipStream_t st);
size_t mom_strip9_lean_lds_bytes(int ns);
size_t mom_strip10_lean_lds_bytes(int ns);
hipError_t mom6_lean9_launch(const void *layer_args, int grid, hipStream_t st);        // momcore_lean6.hip: the six-wave lean image
hipError_t mom6_lean10_launch(const void *layer_args, int grid, hipStream_t st);
size_t mom6_lean9_lds_bytes(int ns);
size_t mom6_lean10_lds_bytes(int ns);
// the quad-block image (momcore_q4.hip, mom_q4.hpp): one wavefront per unit, v_mfma_f64_4x4x4 products, four units per CU
#define MOM_Q4_DECL(KS)                                                                  \
  hipError_t momq_q4_##KS##_launch(const void *layer_args, int grid, hipStream_t st);     \
  size_t momq_q4_##KS##_lds_bytes(int ns, int K);                                         \
  int momq_q4_##KS##_per_cu();
MOM_Q4_DECL(5) MOM_Q4_DECL(6) MOM_Q4_DECL(7) MOM_Q4_DECL(8) MOM_Q4_DECL(9) MOM_Q4_DECL(10)
#undef MOM_Q4_DECL
// the image of operator edge N = 4 KS: LDS bytes (0: does not apply), launch, workgroups per CU
static size_t q4_image_lds(int N, int ns, int K) {
  switch (N) {
    case 20: return momq_q4_5_lds_bytes(ns, K); case 24: return momq_q4_6_lds_bytes(ns, K); case 28: return momq_q4_7_lds_bytes(ns, K);
    case 32: return momq_q4_8_lds_bytes(ns, K); case 36: return momq_q4_9_lds_bytes(ns, K); case 40: return momq_q4_10_lds_bytes(ns, K);
    default: return 0;
  }
}
static hipError_t q4_image_launch(int N, const void *args, int num_cu, size_t units, hipStream_t st) {
  static int per_cu[6] = {0, 0, 0, 0, 0, 0};   // (per process: the occupancy of an image does not depend on the handle)
  const int k = N / 4 - 5;
  if (per_cu[k] == 0)
    per_cu[k] = (N == 20 ? momq_q4_5_per_cu : N == 24 ? momq_q4_6_per_cu : N == 28 ? momq_q4_7_per_cu : N == 32 ? momq_q4_8_per_cu
                 : N == 36 ? momq_q4_9_per_cu : momq_q4_10_per_cu)();
  const int grid = (int)std::min<size_t>(units, (size_t)per_cu[k] * num_cu);
  return (N == 20 ? momq_q4_5_launch : N == 24 ? momq_q4_6_launch : N == 28 ? momq_q4_7_launch : N == 32 ? momq_q4_8_launch
          : N == 36 ? momq_q4_9_launch : momq_q4_10_launch)(args, grid, st);
}
hipError_t mom_strip11_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom4_strip11_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);  // 4-wave build of N = 44
hipError_t mom_strip13_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom_strip14_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom_strip15_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t mom4_launch_surface(const void *surf_args, bool lds, int grid, size_t smem, hipStream_t st);
int mom4_generic_bufs_elems(int N);
// momcore_f32.hip: the Float32 build of the scene-level path (dtype = 1)
struct momf_scene;
int momf_create(momf_scene **out, int device, hipStream_t stream, int N, int nS, int S, int max_m, int *d_info);
void momf_destroy(momf_scene *s);
const char *momf_error(const momf_scene *s);
void momf_set_options(momf_scene *s, int inv_mode, int force_generic, int sweep, int small_n, int m0, int pad, int w4);
int momf_set_streams(momf_scene *s, const double *mu, const double *wt, const double *sg, int imu0, double mu0, const double *I0,
                     const double *D, int regular);
int momf_scene_set(momf_scene *s, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                   const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *tau_sum,
                   double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi);
int momf_scene_set_dev(momf_scene *s, int Nz, int K, int M, const double *d_tau, const double *d_varpi, const double *d_zw,
                       const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *d_tau_sum,
                       double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi);
int momf_scene_set_surface(momf_scene *s, int kind, int M, const double *Rsurf, const double *albedo_spec);
int momf_rt_run(momf_scene *s);
int momf_get_RT(momf_scene *s, double *R, double *T);
int momf_get_hdr(momf_scene *s, double *hdr, double *up, double *dw);
int momf_timers(momf_scene *s, double *ms, int *launches);
int momf_blas(momf_scene *s, int n, int batch, const double *A, const double *B, double *C, bool inv);
int momf_op_elemental(momf_scene *s, int m, int nd, const double *tau_sum, const double *dtau, const double *varpi,
                      const double *Zpp, const double *Zmp, int z_batch);
int momf_op_doubling(momf_scene *s, int nd, double *expk);
int momf_op_interaction(momf_scene *s, int iface, int with_surface_layer);
int momf_op_copy_added_to_composite(momf_scene *s);
int momf_op_surface_lambertian(momf_scene *s, int m, double albedo, const double *tau_tot);
int momf_op_upload(momf_scene *s, int which, const double *src);
int momf_op_download(momf_scene *s, int which, double *dst);
// mom_small.hip: N <= 4, one spectral point per lane, the whole sweep in one launch
hipError_t momsm_launch_sweep(const void *args, int N, hipStream_t st);
hipError_t momw_launch_sweep(const void *args, hipStream_t st);

static thread_local std::string g_err;
static int check_info(mom_t *h);
static void (*g_rccl_destroy)(void *) = nullptr;  // set once RCCL is loaded (mom_comm_init)

struct mom_handle {
  int device = 0, N = 0, nS = 0, S = 0, M = 0;
  int dtype = 0;              // 0 = Float64, 1 = Float32 (scene-level path only, momcore_f32.hip)
  momf_scene *f32 = nullptr;
  bool lds_mode = true;
  int opt_inverse = 0, opt_force_generic = 0;
  hipStream_t stream = nullptr;
  double *d_mu = nullptr, *d_wt = nullptr, *d_sg = nullptr;
  DevStreams q{};
  bool streams_set = false;
  std::vector<double> h_mu, h_wt;
  int strict = 1;
  double *added[6] = {}, *surf[6] = {}, *comp[6] = {};
  bool op_layers = false;     // added / surface layers of the operator-level API: allocated on first use
  bool comp_pitched = false;  // composite matrix blocks hold scene-level (row-pitched) state
  bool comp_on_chip = false;  // the last mom_rt_run kept the composite layer in registers (lane / wave kernels)
  double *d_post[2] = {};     // operator-level mom_postprocess: gathered J0-/J0+ rows [nVza*nS*S] x 2
  size_t post_cap = 0;
  // RCCL communicator (mom_comm_init); the library is dlopen'ed on first use
  void *comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  double *d_gather = nullptr;
  size_t gather_cap = 0;
  double *d_rrs_send = nullptr;  // packed owned spectra of the RRS run: send buffer of mom_allgather_rrs_device
  size_t rrs_send_cap = 0;
  // device-side layer optics (mom_absorption_* / mom_voigt_tau_abs / mom_scene_set_optics)
  double *d_tau_abs = nullptr, *d_grid = nullptr, *d_lines = nullptr, *d_tau_rayl = nullptr, *d_layer_max = nullptr,
         *d_aer = nullptr;
  int *d_aer_mode = nullptr;
  int abs_Nz = 0;
  size_t lines_cap = 0;   // line capacity of ONE layer's block of d_lines
  int lines_nz = 1;       // layers held in d_lines: arrays [nu | gamma_d | y | S][lines_nz][lines_cap], then the two window arrays as ints
  double *d_prof = nullptr;  // per-layer scalars of mom_voigt_tau_abs_profile
  size_t prof_cap = 0;
  double *d_vec[4] = {};  // S-length temporaries (tau_sum, dtau, varpi, expk)
  double *d_Zop[2] = {};
  size_t Zop_cap = 0;
  // scene
  int Nz = 0, K = 0, nVza = 0, scene_M = 0;
  double *d_tau = nullptr, *d_varpi = nullptr, *d_zw = nullptr, *d_Zpp = nullptr, *d_Zmp = nullptr,
         *d_tau_sum = nullptr, *d_cos = nullptr, *d_sin = nullptr, *d_R = nullptr, *d_T = nullptr, *d_hdr = nullptr,
         *d_hdrJ = nullptr, *d_bhr_uw = nullptr, *d_bhr_dw = nullptr;
  int *d_node = nullptr;
  std::vector<int> nd, iface;
  double albedo = 0.0;
  bool scene_set = false;
  // m = 0 reduction (see mom_scene_set)
  int opt_m0 = 1;
  int opt_w4 = 1;
  int opt_stagger = 1;
  int opt_rrs_kernels = -1;  // MOM_OPT_RRS_KERNELS (-1: momr::KOPT_DEFAULT)
  int opt_overlap = 1;       // MOM_OPT_OVERLAP: the m = 0 sub-problem on a second (high-priority) stream of the handle
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_go = nullptr;
  int surf_kind = 0;         // 0 Lambertian scalar, 1 BRDF matrices, 2 Lambertian Legendre (mom_scene_set_surface)
  // ForwardDiff.Dual run (mom_dual.hip): partials of the scene's inputs and of the outputs, the operator workspace
  int dual_P = 0;
  bool dual_ran = false;
  bool dual_last = false;   // the last run of the resident scene was mom_rt_run_dual: hdr / bhr hold nothing of it
  double *d_dual_in[8] = {};  // dtau, dvarpi, dzw, dZpp, dZmp, dalbedo, dRsurf, dalbedo_spec
  double *d_dual_out = nullptr, *d_dual_ts = nullptr;  // dR | dT [nVza,nS,S,P] x 2; d tau_sum [S,Nz+1,P]
  void *dual_work = nullptr;
  size_t dual_work_cap = 0;
  size_t opt_dual_budget = 0;  // MOM_OPT_DUAL_WORKSPACE_MB (0: 60 % of the free HBM at the time of the run)
  double *d_Rsurf = nullptr, *d_Rsurf0 = nullptr, *d_albedo_spec = nullptr, *d_hdrJm = nullptr;
  int opt_sweep = 1;       // one launch walks all layers of a unit (LayerArgs::Nz_sweep)
  double *comp_top[6] = {};  // mom_rt_run_multisensor: composite state of the slab above a sensor
  double *d_msJ[2] = {};     // interface fields dwJ, uwJ [Nk,S,M]
  std::vector<double *> ms_comp;  // multi-sensor: 6 arrays per composite set (snapshot of the top slab + bottom slab per sensor)
  size_t ms_sets = 0;
  double *d_ms_out = nullptr;  // [2][nVza*nS*S*nSensors]
  size_t ms_out_cap = 0;
  int opt_pad = 1;         // scene-level path: pad the operator edge to the next strip-chained kernel size (strip_pad)
  int opt_lean = 3;        // N = 36, 40: 3 = the quad-block image (one wavefront per unit, 4 x 4 x 4 MFMA blocks, four units per CU;
                           // mom_q4.hpp), 1 = the four-wave lean strip image (three workgroups per CU), 2 = the six-wave one (half-strip
                           // doubling chains, two per CU: measured slower, profiles/r05_mid_ab.txt), each followed by the full image's
                           // resume launch; 0 = the full image only
  int *d_resume = nullptr; // resume[unit] of the lean image (mom_lean.hpp)
  size_t resume_cap = 0;
  int Nk = 0;              // operator edge the scene-level kernels of the full problem run with (>= N)
  DevStreams qk{};         // q with N = Nk
  int opt_small = 1;       // N <= 4: lane-per-point sweep kernel (mom_small.hip)
  double *d_smtab = nullptr;  // F1 | F2 | SI tables [3][N,N]
  double *d_smpart = nullptr; // N <= 4, one (point, moment) per lane: the per-moment terms of R_SFI / T_SFI [M][2][nVza,nS,S]
  size_t smpart_cap = 0;
  int *d_ndif = nullptr;      // ndoubl | iface [2][Nz]
  size_t ndif_cap = 0;
  bool red0 = false;
  int N0 = 0, nS0 = 0;
  DevStreams q0{};
  double *d_mu0 = nullptr, *d_wt0 = nullptr, *d_sg0 = nullptr, *d_Zpp0 = nullptr, *d_Zmp0 = nullptr, *d_hdrJ0 = nullptr,
         *d_scratch0 = nullptr;
  double *comp0[6] = {};
  double *d_scratch = nullptr;
  int G = 0;  // workgroups in generic mode
  int num_cu = 256;
  int *d_info = nullptr;
  hipEvent_t ev[4] = {};
  hipEvent_t ev_voigt[2] = {};  // mom_voigt_tau_abs_profile's timing pair (created on first use, owned by the handle)
  std::vector<hipEvent_t> ev_full, ev_red;  // start/stop pairs around each full-problem / reduced layer launch
  int launches = 0, launches_full = 0, launches_red = 0;
  // rotational-Raman path (mom_rrs.hip): the persistent AddedLayerRS / CompositeLayerRS state and the scene's Raman inputs
  momr::State *rrs = nullptr;
  double *d_fscatt = nullptr, *d_Zr[2] = {};  // fScattRayleigh [S,Nz]; Raman phase matrices [N,N,M] x2
  double *d_rrs_op[8] = {};                   // operator-level inputs: tau_sum, dtau, varpi, fscatt [S]; Z x4 [N,N]
  bool rrs_scene = false;
  double rrs_ms = 0.0;
  // grow-only device workspace of the operator-level batched entry points (no hipMalloc / hipFree per call, no leak on an
  // error return): slot k holds ws_cap[k] bytes
  void *ws[4] = {};
  size_t ws_cap[4] = {};
  // resident HITRAN table + TIPS splines of one absorber (mom_absorption_set_lines)
  MomLineTable lt{};
  double *d_lt = nullptr;   // one allocation behind lt's double arrays
  int *d_lt_i = nullptr;    // iso index [nLines] | knots per isotopologue [nIso] | unsorted flag [1]
  double lt_Tmin = 0.0, lt_Tmax = 0.0;
  std::string err;
};

#define HIPCHK(h, call)                                                                            \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess) {                                                                       \
      char buf__[512];                                                                             \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      if (h) (h)->err = buf__;                                                                     \
      g_err = buf__;                                                                               \
      return MOM_EHIP;                                                                             \
    }                                                                                              \
  } while (0)

#define F64_ONLY(h, name)                                                                                 \
  if ((h) && (h)->dtype != 0)                                                                             \
  return fail(h, MOM_EINVAL, name ": not available on a Float32 (dtype = 1) handle (scene-level path only)")

static int fail(mom_t *h, int code, const char *msg);
static int fail(mom_t *h, int code, const char *msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}

template <class T>
static hipError_t dmalloc(T **p, size_t count) {
  return hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
}

// slot of the handle's grow-only workspace, at least `count` elements of T
template <class T>
static hipError_t ws_get(mom_t *h, int slot, T **p, size_t count) {
  const size_t bytes = count * sizeof(T);
  if (bytes > h->ws_cap[slot]) {
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return e;
    (void)hipFree(h->ws[slot]);
    h->ws[slot] = nullptr;
    h->ws_cap[slot] = 0;
    e = hipMalloc(&h->ws[slot], bytes);
    if (e != hipSuccess) return e;
    h->ws_cap[slot] = bytes;
  }
  *p = reinterpret_cast<T *>(h->ws[slot]);
  return hipSuccess;
}

static size_t smem_bytes(const mom_t *h) { return lds_bytes(h->N, h->lds_mode); }

// Scene-level path: an operator edge N for which no strip-chained kernel image exists is padded with up to 4 DUMMY
// STREAM ENTRIES (mu = 1, weight 0, zero rows and columns in every phase-matrix basis and BRDF matrix) when that
// reaches a size one exists for: most IQU stream counts (N = 3 k is a multiple of 4 only for every fourth k), and
// N = 32, 48 (measured: N = 48 as 52 runs 1.3x faster than the general path at 48).  A dummy entry is decoupled exactly:
// its column is zero in r and off-diagonal in t (zero weight, elemental.jl:198-205), its row is zero because its Z row
// is, so every product, series and pivoted inverse leaves the real rows and columns with the same terms plus exact zeros.
constexpr int kPadMax = 4;
static bool strip_size(int N) { return N == 36 || N == 40 || N == 44 || N == 52 || N == 56 || N == 60; }
static int strip_pad(int N) {
  if (strip_size(N)) return N;
  for (int p = N + 1; p <= N + kPadMax; ++p)
    if (strip_size(p)) return p;
  return N;
}
// [N,N,B] -> [Nk,Nk,B], zero padded
static std::vector<double> pad_blocks(const double *src, int N, int Nk, size_t B) {
  std::vector<double> out((size_t)Nk * Nk * B, 0.0);
  for (size_t b = 0; b < B; ++b)
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) out[i + (size_t)Nk * (j + (size_t)Nk * b)] = src[i + (size_t)N * (j + (size_t)N * b)];
  return out;
}

template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  return mom_allow_lds(reinterpret_cast<const void *>(kernel), bytes);
}

void mom_set_global_error(const char *msg) { g_err = msg ? msg : ""; }
extern "C" const char *mom_last_global_error(void) { return g_err.c_str(); }
extern "C" const char *mom_last_error(const mom_t *h) { return h ? h->err.c_str() : g_err.c_str(); }

// added layer + surface layer of the operator-level API (rt_run.jl:109-112), on first use
static int ensure_op_layers(mom_t *h) {
  if (h->op_layers) return MOM_OK;
  const size_t NN = (size_t)h->N * h->N;
  for (int k = 0; k < 6; ++k) {
    const size_t per = ((k < 4) ? NN : (size_t)h->N) * h->S;
    HIPCHK(h, dmalloc(&h->added[k], per));
    HIPCHK(h, dmalloc(&h->surf[k], per));
    HIPCHK(h, hipMemsetAsync(h->added[k], 0, per * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->surf[k], 0, per * sizeof(double), h->stream));
  }
  h->op_layers = true;
  return MOM_OK;
}
// the operator-level API keeps the composite matrices in the natural [N,N,S] layout; after a scene-level run the
// allocation holds row-pitched blocks, which the operator kernels must not be fed
static int op_composite_ready(mom_t *h, const char *who) {
  if (h->comp_pitched) {
    char buf[192];
    snprintf(buf, sizeof buf, "%s: the composite layer holds scene-level (mom_rt_run) state; start the operator-level "
             "sequence with mom_copy_added_to_composite or mom_upload", who);
    return fail(h, MOM_ESTATE, buf);
  }
  return MOM_OK;
}

extern "C" int mom_create(mom_t **out, int device, int N, int nStokes, int S, int max_m, int dtype) {
  if (!out || N <= 0 || S <= 0 || max_m <= 0 || nStokes <= 0 || nStokes > 4 || N % nStokes != 0)
    return fail(nullptr, MOM_EINVAL, "mom_create: bad argument");
  if (dtype != 0 && dtype != 1) return fail(nullptr, MOM_EINVAL, "mom_create: dtype must be 0 (Float64) or 1 (Float32)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(nullptr, MOM_EHIP, "mom_create: no HIP device available");
  if (device < 0 || device >= ndev) return fail(nullptr, MOM_EINVAL, "mom_create: device index out of range");
  mom_t *h = new mom_t();
  h->device = device; h->N = N; h->nS = nStokes; h->S = S; h->M = max_m; h->dtype = dtype;
  h->lds_mode = (N <= 64);
  *out = h;
  HIPCHK(h, hipSetDevice(device));
  HIPCHK(h, hipStreamCreate(&h->stream));
  {  // the second stream of MOM_OPT_OVERLAP: highest priority, so that its (shorter) launches are dispatched first
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_go, hipEventDisableTiming));
  }
  const size_t NN = (size_t)N * N;
  const int Na = N + kPadMax;  // room for the dummy entries of strip_pad
  HIPCHK(h, dmalloc(&h->d_mu, Na));
  HIPCHK(h, dmalloc(&h->d_wt, Na));
  HIPCHK(h, dmalloc(&h->d_sg, Na));
  (void)NN;
  for (int k = 0; k < 6 && dtype == 0; ++k) {
    // composite blocks: room for the scene-level row pitch (comp_pitch); the operator-level API uses the natural one.
    // The added / surface layers of the operator-level API (12 N^2 S doubles) are allocated on its first use
    // (ensure_op_layers): the scene-level path keeps the added layer in LDS and never needs them.
    const size_t perc = (k < 4) ? (size_t)comp_pitch(Na) * Na : (size_t)Na;
    HIPCHK(h, dmalloc(&h->comp[k], perc * S * max_m));
    HIPCHK(h, hipMemsetAsync(h->comp[k], 0, perc * S * max_m * sizeof(double), h->stream));
  }
  for (int k = 0; k < 4; ++k) HIPCHK(h, dmalloc(&h->d_vec[k], S));
  HIPCHK(h, dmalloc(&h->d_info, 1));
  HIPCHK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  h->G = 1024;
  {
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (dtype == 1) {  // Float32: the scene-level state lives in the f32 build's own object
    for (int k = 0; k < 4; ++k) HIPCHK(h, hipEventCreate(&h->ev[k]));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int rc = momf_create(&h->f32, device, h->stream, N, nStokes, S, max_m, h->d_info);
    if (rc) return fail(h, rc, momf_error(h->f32));
    return MOM_OK;
  }
  // + one padded matrix of slack: B-operand reads of the last column tile run past the stored columns
  HIPCHK(h, dmalloc(&h->d_scratch, (size_t)h->G * kGenericBufs * mat_elems(N) + (size_t)ld_for(N) * np_for(N)));
  HIPCHK(h, hipMemsetAsync(h->d_scratch, 0, ((size_t)h->G * kGenericBufs * mat_elems(N) + (size_t)ld_for(N) * np_for(N)) * sizeof(double), h->stream));
  for (int k = 0; k < 4; ++k) HIPCHK(h, hipEventCreate(&h->ev[k]));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_destroy(mom_t *h) {
  if (!h) return MOM_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->f32) momf_destroy(h->f32);
  momr::destroy(h->rrs);
  (void)hipFree(h->d_lt); (void)hipFree(h->d_lt_i);
  for (int k = 0; k < 4; ++k) (void)hipFree(h->ws[k]);
  (void)hipFree(h->d_fscatt); (void)hipFree(h->d_Zr[0]); (void)hipFree(h->d_Zr[1]);
  for (int k = 0; k < 8; ++k) (void)hipFree(h->d_rrs_op[k]);
  if (h->comm && g_rccl_destroy) g_rccl_destroy(h->comm);
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  fr(h->d_mu); fr(h->d_wt); fr(h->d_sg);
  for (int k = 0; k < 8; ++k) fr(h->d_dual_in[k]);
  fr(h->d_dual_out); fr(h->d_dual_ts); fr(h->dual_work);
  for (int k = 0; k < 6; ++k) { fr(h->added[k]); fr(h->surf[k]); fr(h->comp[k]); fr(h->comp_top[k]); }
  fr(h->d_msJ[0]); fr(h->d_msJ[1]); fr(h->d_ms_out);
  for (auto p : h->ms_comp) fr(p);
  for (int k = 0; k < 4; ++k) fr(h->d_vec[k]);
  fr(h->d_Zop[0]); fr(h->d_Zop[1]);
  fr(h->d_tau); fr(h->d_varpi); fr(h->d_zw); fr(h->d_Zpp); fr(h->d_Zmp); fr(h->d_tau_sum); fr(h->d_cos); fr(h->d_sin);
  fr(h->d_mu0); fr(h->d_wt0); fr(h->d_sg0); fr(h->d_Zpp0); fr(h->d_Zmp0); fr(h->d_hdrJ0); fr(h->d_scratch0);
  for (int k = 0; k < 6; ++k) fr(h->comp0[k]);
  fr(h->d_R); fr(h->d_hdr); fr(h->d_post[0]); fr(h->d_gather); fr(h->d_rrs_send); fr(h->d_Rsurf); fr(h->d_Rsurf0); fr(h->d_albedo_spec); fr(h->d_hdrJm); fr(h->d_smtab); fr(h->d_smpart); if (h->d_resume) (void)hipFree(h->d_resume); if (h->d_ndif) (void)hipFree(h->d_ndif); fr(h->d_tau_abs); fr(h->d_grid); fr(h->d_lines); fr(h->d_prof); fr(h->d_tau_rayl);
  fr(h->d_layer_max); fr(h->d_aer); if (h->d_aer_mode) (void)hipFree(h->d_aer_mode); fr(h->d_hdrJ); fr(h->d_bhr_uw); fr(h->d_bhr_dw); fr(h->d_node); fr(h->d_scratch); fr(h->d_info);
  for (int k = 0; k < 4; ++k) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
  for (int k = 0; k < 2; ++k) if (h->ev_voigt[k]) (void)hipEventDestroy(h->ev_voigt[k]);
  for (auto e : h->ev_full) (void)hipEventDestroy(e);
  for (auto e : h->ev_red) (void)hipEventDestroy(e);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_go) (void)hipEventDestroy(h->ev_go);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MOM_OK;
}

extern "C" int mom_sync(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_check(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  HIPCHK(h, hipSetDevice(h->device));
  return check_info(h);
}

extern "C" int mom_set_option(mom_t *h, int option, int value) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (option == MOM_OPT_INVERSE) { h->opt_inverse = value; h->q.inv_mode = value; h->qk.inv_mode = value; h->q0.inv_mode = value; }
  else if (option == MOM_OPT_M0_REDUCTION) h->opt_m0 = value;
  else if (option == MOM_OPT_SMALL_WG) h->opt_w4 = value;
  else if (option == MOM_OPT_STAGGER) h->opt_stagger = value;
  else if (option == MOM_OPT_SMALL_N) { h->opt_small = value; h->scene_set = false; }  // the padded edge Nk depends on it
  else if (option == MOM_OPT_LAYER_SWEEP) h->opt_sweep = value;
  else if (option == MOM_OPT_STRIP_PAD) { h->opt_pad = value; h->scene_set = false; }
  else if (option == MOM_OPT_LEAN) { h->opt_lean = value; h->scene_set = false; }  // the padded edge of the m = 0 sub-problem depends on it
  else if (option == MOM_OPT_OVERLAP) h->opt_overlap = value;
  else if (option == MOM_OPT_DUAL_WORKSPACE_MB) {
    if (value < 0) return fail(h, MOM_EINVAL, "mom_set_option: MOM_OPT_DUAL_WORKSPACE_MB takes megabytes >= 0 (0 = 60 % of the free HBM)");
    h->opt_dual_budget = (size_t)value << 20;
  }
  else if (option == MOM_OPT_RRS_KERNELS) {
    if (value < 0 || value > 63 || ((value & momr::KOPT_EL_FUSE_ON) && (value & momr::KOPT_EL_FUSE_OFF)))
      return fail(h, MOM_EINVAL, "mom_set_option: MOM_OPT_RRS_KERNELS takes a mask of bits 0..5 (bits 4 and 5 exclude each other)");
    h->opt_rrs_kernels = value;
    if (h->rrs) h->rrs->kopt = value;
  }
  else if (option == MOM_OPT_FORCE_GENERIC) {
    h->opt_force_generic = value;
    h->lds_mode = (h->N <= 64) && !value;
  } else return fail(h, MOM_EINVAL, "mom_set_option: unknown option");
  if (h->f32) momf_set_options(h->f32, h->opt_inverse, h->opt_force_generic, h->opt_sweep, h->opt_small, h->opt_m0, h->opt_pad, h->opt_w4);
  return MOM_OK;
}

extern "C" int mom_set_streams(mom_t *h, const double *qp_muN, const double *wt_muN, int N, int imu0_1based, double mu0,
                               const double *I0, const double *D, int strict) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (N != h->N || !qp_muN || !wt_muN || !I0 || !D || imu0_1based < 1 || imu0_1based * h->nS > N)
    return fail(h, MOM_EINVAL, "mom_set_streams: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  h->h_mu.assign(qp_muN, qp_muN + N);
  h->h_wt.assign(wt_muN, wt_muN + N);
  h->strict = strict;
  std::vector<double> sg(N);
  for (int i = 0; i < N; ++i) {
    const int comp = strict ? ((i + 1) % h->nS) : (i % h->nS) + 1;
    sg[i] = (comp > 2) ? -1.0 : 1.0;
  }
  {
    std::vector<double> mu(qp_muN, qp_muN + N), wt(wt_muN, wt_muN + N);
    mu.resize(N + kPadMax, 1.0); wt.resize(N + kPadMax, 0.0); sg.resize(N + kPadMax, 1.0);  // dummy entries (strip_pad)
    HIPCHK(h, hipMemcpyAsync(h->d_mu, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_wt, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_sg, sg.data(), sg.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  DevStreams &q = h->q;
  q.mu = h->d_mu; q.wt = h->d_wt; q.sg = h->d_sg;
  for (int k = 0; k < 4; ++k) { q.I0[k] = (k < h->nS) ? I0[k] : 0.0; q.D[k] = (k < h->nS) ? D[k] : 1.0; }
  q.N = N; q.nS = h->nS; q.imu0 = imu0_1based; q.mu0 = mu0; q.inv_mode = h->opt_inverse;
  q.regular = 1;
  for (int i = 0; i < N; ++i)
    if (qp_muN[i] != qp_muN[(i / h->nS) * h->nS]) q.regular = 0;
  if (h->f32) {
    momf_set_options(h->f32, h->opt_inverse, h->opt_force_generic, h->opt_sweep, h->opt_small, h->opt_m0, h->opt_pad, h->opt_w4);
    const int rc = momf_set_streams(h->f32, qp_muN, wt_muN, sg.data(), imu0_1based, mu0, I0, D, q.regular);
    if (rc) return fail(h, rc, momf_error(h->f32));
  }
  h->streams_set = true;
  h->scene_set = false;  // the reduced (I,Q) stream set of a resident scene was derived from the old streams
  return MOM_OK;
}

static int grid_for(const mom_t *h, size_t total) {
  if (h->lds_mode) return (int)total;
  return (int)std::min<size_t>(total, (size_t)h->G);
}
static int check_info(mom_t *h) {
  int info = 0;
  HIPCHK(h, hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (info) {
    HIPCHK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    char buf[128];
    snprintf(buf, sizeof buf, "zero pivot at elimination step %d while inverting (I - R r)", info);
    return fail(h, MOM_ESINGULAR, buf);
  }
  return MOM_OK;
}

#define LAUNCH(h, KERN, grid, args)                                                       \
  do {                                                                                    \
    const size_t sm__ = smem_bytes(h);                                                    \
    if ((h)->lds_mode) {                                                                  \
      HIPCHK(h, allow_lds(KERN<true>, sm__));                                             \
      hipLaunchKernelGGL(KERN<true>, dim3(grid), dim3(kThreads), sm__, (h)->stream, args); \
    } else {                                                                              \
      HIPCHK(h, allow_lds(KERN<false>, sm__));                                            \
      hipLaunchKernelGGL(KERN<false>, dim3(grid), dim3(kThreads), sm__, (h)->stream, args); \
    }                                                                                     \
    HIPCHK(h, hipGetLastError());                                                         \
  } while (0)

#define LAUNCH2(h, KERN, TARG, grid, args)                                                        \
  do {                                                                                            \
    const size_t sm__ = smem_bytes(h);                                                            \
    if ((h)->lds_mode) {                                                                          \
      HIPCHK(h, allow_lds(KERN<true, TARG>, sm__));                                               \
      hipLaunchKernelGGL((KERN<true, TARG>), dim3(grid), dim3(kThreads), sm__, (h)->stream, args); \
    } else {                                                                                      \
      HIPCHK(h, allow_lds(KERN<false, TARG>, sm__));                                              \
      hipLaunchKernelGGL((KERN<false, TARG>), dim3(grid), dim3(kThreads), sm__, (h)->stream, args); \
    }                                                                                             \
    HIPCHK(h, hipGetLastError());                                                                 \
  } while (0)

extern "C" int mom_elemental(mom_t *h, int m, int ndoubl, const double *tau_sum, const double *dtau, const double *varpi,
                             const double *Zpp, const double *Zmp, int z_batch) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_elemental: call mom_set_streams first");
  if (!tau_sum || !dtau || !varpi || !Zpp || !Zmp || (z_batch != 1 && z_batch != h->S) || m < 0 || ndoubl < 0)
    return fail(h, MOM_EINVAL, "mom_elemental: bad argument");
  if (h->f32) {
    const int rc = momf_op_elemental(h->f32, m, ndoubl, tau_sum, dtau, varpi, Zpp, Zmp, z_batch);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  const size_t NN = (size_t)h->N * h->N, zc = NN * z_batch;
  if (zc > h->Zop_cap) {
    if (h->d_Zop[0]) { (void)hipFree(h->d_Zop[0]); (void)hipFree(h->d_Zop[1]); }
    HIPCHK(h, dmalloc(&h->d_Zop[0], zc));
    HIPCHK(h, dmalloc(&h->d_Zop[1], zc));
    h->Zop_cap = zc;
  }
  const size_t sb = (size_t)h->S * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(h->d_vec[0], tau_sum, sb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_vec[1], dtau, sb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_vec[2], varpi, sb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_Zop[0], Zpp, zc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_Zop[1], Zmp, zc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  OpArgs a{};
  a.q = h->q; a.S = h->S; a.m = m; a.nd = ndoubl; a.z_batch = z_batch;
  a.tau_sum = h->d_vec[0]; a.dtau = h->d_vec[1]; a.varpi = h->d_vec[2]; a.Zpp = h->d_Zop[0]; a.Zmp = h->d_Zop[1];
  for (int k = 0; k < 6; ++k) a.added[k] = h->added[k];
  a.scratch = h->d_scratch; a.info = h->d_info;
  LAUNCH(h, k_op_elemental, grid_for(h, h->S), a);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_doubling(mom_t *h, int ndoubl, double *expk) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_doubling: call mom_set_streams first");
  if (!expk || ndoubl < 0) return fail(h, MOM_EINVAL, "mom_doubling: bad argument");
  if (h->f32) {
    const int rc = momf_op_doubling(h->f32, ndoubl, expk);
    return rc ? fail(h, rc, momf_error(h->f32)) : check_info(h);
  }
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  const size_t sb = (size_t)h->S * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(h->d_vec[3], expk, sb, hipMemcpyHostToDevice, h->stream));
  OpArgs a{};
  a.q = h->q; a.S = h->S; a.nd = ndoubl; a.expk = h->d_vec[3];
  for (int k = 0; k < 6; ++k) a.added[k] = h->added[k];
  a.scratch = h->d_scratch; a.info = h->d_info;
  if (ndoubl > 0) LAUNCH(h, k_op_doubling, grid_for(h, h->S), a);  // doubling.jl:28 returns early for 0
  HIPCHK(h, hipMemcpyAsync(expk, h->d_vec[3], sb, hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

extern "C" int mom_interaction(mom_t *h, int iface, int with_surface_layer) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_interaction: call mom_set_streams first");
  if (iface < 0 || iface > 3) return fail(h, MOM_EINVAL, "mom_interaction: iface must be 0..3");
  if (h->f32) {
    const int rc = momf_op_interaction(h->f32, iface, with_surface_layer);
    return rc ? fail(h, rc, momf_error(h->f32)) : check_info(h);
  }
  HIPCHK(h, hipSetDevice(h->device));
  { int rc_ = ensure_op_layers(h); if (rc_) return rc_; if ((rc_ = op_composite_ready(h, "mom_interaction"))) return rc_; }
  OpArgs a{};
  a.q = h->q; a.S = h->S; a.iface = iface;
  for (int k = 0; k < 6; ++k) { a.added[k] = with_surface_layer ? h->surf[k] : h->added[k]; a.comp[k] = h->comp[k]; }
  a.scratch = h->d_scratch; a.info = h->d_info;
  LAUNCH(h, k_op_interaction, grid_for(h, h->S), a);
  return check_info(h);
}

extern "C" int mom_copy_added_to_composite(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (h->f32) {
    const int rc = momf_op_copy_added_to_composite(h->f32);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  h->comp_pitched = false;
  const size_t NN = (size_t)h->N * h->N;
  // composite order: R_mp, R_pm, T_pp, T_mm, J0p, J0m ; added order: r_pm, r_mp, t_mm, t_pp, j0p, j0m
  const int src[6] = {1, 0, 3, 2, 4, 5};
  for (int k = 0; k < 6; ++k) {
    const size_t bytes = ((k < 4) ? NN : (size_t)h->N) * h->S * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(h->comp[k], h->added[src[k]], bytes, hipMemcpyDeviceToDevice, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_surface_lambertian(mom_t *h, int m, double albedo, const double *tau_tot) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_surface_lambertian: call mom_set_streams first");
  if (!tau_tot || m < 0) return fail(h, MOM_EINVAL, "mom_surface_lambertian: bad argument");
  if (h->f32) {
    const int rc = momf_op_surface_lambertian(h->f32, m, albedo, tau_tot);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  HIPCHK(h, hipMemcpyAsync(h->d_vec[0], tau_tot, (size_t)h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_op_surface_fill, dim3(h->S), dim3(256), 0, h->stream, h->q, h->S, m, albedo, h->d_vec[0],
                     h->surf[0], h->surf[1], h->surf[2], h->surf[3], h->surf[4], h->surf[5]);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

static double *which_ptr(mom_t *h, int which, size_t *count) {
  const size_t NN = (size_t)h->N * h->N;
  if (which < 0 || which > 17) return nullptr;
  const int grp = which / 6, k = which % 6;
  *count = ((k < 4) ? NN : (size_t)h->N) * h->S;
  return grp == 0 ? h->added[k] : (grp == 1 ? h->comp[k] : h->surf[k]);
}

extern "C" int mom_upload(mom_t *h, int which, const double *src) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (which < 0 || which > 17 || !src) return fail(h, MOM_EINVAL, "mom_upload: bad argument");
  if (h->f32) {
    const int rc = momf_op_upload(h->f32, which, src);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (which / 6 != 1) { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  else h->comp_pitched = false;  // the caller (re)starts an operator-level sequence: natural [N,N,S] layout
  size_t count = 0;
  double *p = which_ptr(h, which, &count);
  HIPCHK(h, hipMemcpyAsync(p, src, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_download(mom_t *h, int which, double *dst) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (which < 0 || which > 17 || !dst) return fail(h, MOM_EINVAL, "mom_download: bad argument");
  if (h->f32) {
    const int rc = momf_op_download(h->f32, which, dst);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (which / 6 != 1) { const int rc_ = ensure_op_layers(h); if (rc_) return rc_; }
  size_t count = 0;
  double *p = which_ptr(h, which, &count);
  if (which / 6 == 1 && h->comp_pitched) {
    // after mom_rt_run: moment slot 0 of the scene-level state.  With the m = 0 reduction that moment lives in the
    // (I,Q) sub-problem's own arrays, which have no [N,N,S] image
    if (h->comp_on_chip)
      return fail(h, MOM_ESTATE, "mom_download: the run kept the composite layer in registers (operator edge <= 32); set "
                                 "MOM_OPT_SMALL_N = 0 before mom_rt_run to read it back");
    if (h->red0)
      return fail(h, MOM_ESTATE, "mom_download: Fourier moment 0 ran on the (I,Q) sub-problem; set "
                                 "MOM_OPT_M0_REDUCTION = 0 before mom_scene_set to read the composite layer back");
    if (h->Nk != h->N)
      return fail(h, MOM_ESTATE, "mom_download: the scene ran on an operator edge padded to a strip-chained kernel size; "
                                 "set MOM_OPT_STRIP_PAD = 0 before mom_scene_set to read the composite layer back");
    if (which % 6 < 4) {  // de-pitch: columns of N doubles at a pitch of comp_pitch(N)
      HIPCHK(h, hipMemcpy2DAsync(dst, (size_t)h->N * sizeof(double), p, (size_t)comp_pitch(h->N) * sizeof(double),
                                 (size_t)h->N * sizeof(double), (size_t)h->N * h->S, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      return MOM_OK;
    }
  }
  HIPCHK(h, hipMemcpyAsync(dst, p, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

static int blas_common(mom_t *h, int n, int batch, const double *A, const double *B, double *C, bool inv) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (n <= 0 || batch <= 0 || !A || !C || (!inv && !B)) return fail(h, MOM_EINVAL, "batched op: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->f32) {  // Float32 handle: the f32 build's kernels (gpu_batched.jl:45-58)
    const int rc = momf_blas(h->f32, n, batch, A, B, C, inv);
    if (rc) return fail(h, rc, momf_error(h->f32));
    return inv ? check_info(h) : MOM_OK;
  }
  const size_t cnt = (size_t)n * n * batch;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr, *scr = nullptr;
  HIPCHK(h, ws_get(h, 0, &dA, cnt));
  HIPCHK(h, ws_get(h, 1, &dC, cnt));
  HIPCHK(h, hipMemcpyAsync(dA, A, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (!inv) {
    HIPCHK(h, ws_get(h, 2, &dB, cnt));
    HIPCHK(h, hipMemcpyAsync(dB, B, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  const bool lds = n <= 64 && !h->opt_force_generic;
  const int grid = lds ? batch : std::min(batch, 1024);
  if (!lds) {
    const size_t scn = (size_t)grid * kGenericBufs * mat_elems(n) + (size_t)ld_for(n) * np_for(n);
    HIPCHK(h, ws_get(h, 3, &scr, scn));
    HIPCHK(h, hipMemsetAsync(scr, 0, scn * sizeof(double), h->stream));
  }
  BlasArgs a{n, batch, dA, dB, dC, scr, h->d_info};
  const size_t sm = lds_bytes(n, lds);
  if (inv) {
    if (lds) { HIPCHK(h, allow_lds(k_batch_inv<true>, sm)); hipLaunchKernelGGL(k_batch_inv<true>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
    else { HIPCHK(h, allow_lds(k_batch_inv<false>, sm)); hipLaunchKernelGGL(k_batch_inv<false>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
  } else {
    if (lds) { HIPCHK(h, allow_lds(k_batched_mul<true>, sm)); hipLaunchKernelGGL(k_batched_mul<true>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
    else { HIPCHK(h, allow_lds(k_batched_mul<false>, sm)); hipLaunchKernelGGL(k_batched_mul<false>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(C, dC, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return inv ? check_info(h) : MOM_OK;
}

extern "C" int mom_batch_inv(mom_t *h, int n, int batch, const double *A, double *X) {
  return blas_common(h, n, batch, A, nullptr, X, true);
}
extern "C" int mom_batched_mul(mom_t *h, int n, int batch, const double *A, const double *B, double *C) {
  return blas_common(h, n, batch, A, B, C, false);
}

extern "C" int mom_elemental_inelastic_rrs(mom_t *h, int m, int ndoubl, int nRaman, const int *i_l1l0, const double *varpi_l1l0,
                                           const double *fscattRayl, const double *tau_sum, const double *dtau,
                                           const double *varpi, const double *Zpp_l1l0, const double *Zmp_l1l0,
                                           double *ier_mp, double *iet_pp, double *ier_pm, double *iet_mm, double *ieJ0p,
                                           double *ieJ0m) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_elemental_inelastic_rrs");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_elemental_inelastic_rrs: call mom_set_streams first");
  if (m < 0 || ndoubl < 0 || nRaman <= 0 || !i_l1l0 || !varpi_l1l0 || !fscattRayl || !tau_sum || !dtau || !varpi ||
      !Zpp_l1l0 || !Zmp_l1l0 || !ier_mp || !iet_pp || !ier_pm || !iet_mm || !ieJ0p || !ieJ0m)
    return fail(h, MOM_EINVAL, "mom_elemental_inelastic_rrs: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const int N = h->N;
  const size_t S = h->S, NN = (size_t)N * N, big = NN * S * nRaman, vec = (size_t)N * S * nRaman;
  double *buf = nullptr;
  int *dI = nullptr;
  HIPCHK(h, ws_get(h, 0, &buf, 4 * big + 2 * vec + nRaman + 4 * S + 2 * NN));
  HIPCHK(h, ws_get(h, 1, &dI, (size_t)nRaman));
  double *d_out = buf, *d_vp = buf + 4 * big + 2 * vec, *d_fs = d_vp + nRaman, *d_ts = d_fs + S, *d_dt = d_ts + S,
         *d_w = d_dt + S, *d_zp = d_w + S, *d_zm = d_zp + NN;
  HIPCHK(h, hipMemcpyAsync(dI, i_l1l0, nRaman * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_vp, varpi_l1l0, nRaman * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_fs, fscattRayl, S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_ts, tau_sum, S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_dt, dtau, S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_w, varpi, S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_zp, Zpp_l1l0, NN * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_zm, Zmp_l1l0, NN * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RrsArgs a{};
  a.q = h->q; a.S = h->S; a.nR = nRaman; a.m = m; a.nd = ndoubl; a.strict = h->strict;
  a.i_l1l0 = dI; a.varpi_l1l0 = d_vp; a.fscatt = d_fs; a.tau_sum = d_ts; a.dtau = d_dt; a.varpi = d_w; a.Zpp = d_zp; a.Zmp = d_zm;
  a.ier_mp = d_out; a.iet_pp = d_out + big; a.ier_pm = d_out + 2 * big; a.iet_mm = d_out + 3 * big;
  a.ieJ0p = d_out + 4 * big; a.ieJ0m = d_out + 4 * big + vec;
  hipLaunchKernelGGL(k_elemental_rrs, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  double *dst[6] = {ier_mp, iet_pp, ier_pm, iet_mm, ieJ0p, ieJ0m};
  for (int k = 0; k < 4; ++k) HIPCHK(h, hipMemcpyAsync(dst[k], d_out + k * big, big * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dst[4], a.ieJ0p, vec * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dst[5], a.ieJ0m, vec * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

static int dual_common(mom_t *h, int n, int batch, int P, const double *A, const double *dA, const double *B,
                       const double *dB, double *C, double *dC, bool inv) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_batch_inv_dual / mom_batched_mul_dual");
  if (n <= 0 || batch <= 0 || P < 0 || !A || !C || (P > 0 && (!dA || !dC)) || (!inv && (!B || (P > 0 && !dB))))
    return fail(h, MOM_EINVAL, "batched dual op: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t cnt = (size_t)n * n * batch, cntP = cnt * P;
  double *buf = nullptr, *scr = nullptr;
  // one allocation: A, B, C [cnt] and dA, dB, dC [cntP]
  HIPCHK(h, ws_get(h, 0, &buf, 3 * cnt + 3 * cntP + 1));
  double *dA_ = buf + 3 * cnt, *dB_ = dA_ + cntP, *dC_ = dB_ + cntP;
  HIPCHK(h, hipMemcpyAsync(buf, A, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (P) HIPCHK(h, hipMemcpyAsync(dA_, dA, cntP * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (!inv) {
    HIPCHK(h, hipMemcpyAsync(buf + cnt, B, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (P) HIPCHK(h, hipMemcpyAsync(dB_, dB, cntP * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  const bool lds = n <= 64 && !h->opt_force_generic;
  const int grid = lds ? batch : std::min(batch, 1024);
  if (!lds) {
    const size_t scn = (size_t)grid * kGenericBufs * mat_elems(n) + (size_t)ld_for(n) * np_for(n);
    HIPCHK(h, ws_get(h, 3, &scr, scn));
    HIPCHK(h, hipMemsetAsync(scr, 0, scn * sizeof(double), h->stream));
  }
  DualArgs a{n, batch, P, buf, dA_, buf + cnt, dB_, buf + 2 * cnt, dC_, scr, h->d_info};
  const size_t sm = lds_bytes(n, lds);
  if (inv) {
    if (lds) { HIPCHK(h, allow_lds(k_batch_inv_dual<true>, sm)); hipLaunchKernelGGL(k_batch_inv_dual<true>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
    else { HIPCHK(h, allow_lds(k_batch_inv_dual<false>, sm)); hipLaunchKernelGGL(k_batch_inv_dual<false>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
  } else {
    if (lds) { HIPCHK(h, allow_lds(k_batched_mul_dual<true>, sm)); hipLaunchKernelGGL(k_batched_mul_dual<true>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
    else { HIPCHK(h, allow_lds(k_batched_mul_dual<false>, sm)); hipLaunchKernelGGL(k_batched_mul_dual<false>, dim3(grid), dim3(kThreads), sm, h->stream, a); }
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(C, buf + 2 * cnt, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (P) HIPCHK(h, hipMemcpyAsync(dC, dC_, cntP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));

  return inv ? check_info(h) : MOM_OK;
}

extern "C" int mom_batch_inv_dual(mom_t *h, int n, int batch, int P, const double *A, const double *dA, double *X, double *dX) {
  return dual_common(h, n, batch, P, A, dA, nullptr, nullptr, X, dX, true);
}
extern "C" int mom_batched_mul_dual(mom_t *h, int n, int batch, int P, const double *A, const double *dA, const double *B,
                                    const double *dB, double *C, double *dC) {
  return dual_common(h, n, batch, P, A, dA, B, dB, C, dC, false);
}

// ---------------------------------------------------------------- scene-level

template <class T>
static int upload_new(mom_t *h, T **dst, const T *src, size_t count) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  HIPCHK(h, dmalloc(dst, count));
  HIPCHK(h, hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return MOM_OK;
}

// everything of a scene that does not depend on how the layer optics reach the device: phase-matrix bases, view
// geometry, output buffers, the m = 0 reduction
static int scene_common(mom_t *h, int Nz, int K, int M, const double *Zpp, const double *Zmp, double albedo, int nVza,
                        const int *node_1based, const double *cos_mphi, const double *sin_mphi);

extern "C" int mom_scene_set(mom_t *h, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                             const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface,
                             const double *tau_sum, double albedo, int nVza, const int *node_1based,
                             const double *cos_mphi, const double *sin_mphi) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_scene_set: call mom_set_streams first");
  if (Nz <= 0 || K <= 0 || M <= 0 || M > h->M || nVza <= 0 || !tau || !varpi || !zw || !Zpp || !Zmp || !ndoubl ||
      !iface || !tau_sum || !node_1based || !cos_mphi || !sin_mphi)
    return fail(h, MOM_EINVAL, "mom_scene_set: bad argument");
  if (K > 64) return fail(h, MOM_EINVAL, "mom_scene_set: at most 64 phase-matrix bases (Rayleigh + aerosol types)");
  for (int z = 0; z < Nz; ++z)
    if (ndoubl[z] < 0 || ndoubl[z] > 60 || iface[z] < 0 || iface[z] > 3)
      return fail(h, MOM_EINVAL, "mom_scene_set: ndoubl/iface out of range");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t S = h->S;
  h->scene_set = false;
  int rc;
  if (h->f32) {
    for (int v = 0; v < nVza; ++v)
      if (node_1based[v] < 1 || node_1based[v] * h->nS > h->N) return fail(h, MOM_EINVAL, "mom_scene_set: bad view node");
    if ((rc = momf_scene_set(h->f32, Nz, K, M, tau, varpi, zw, Zpp, Zmp, ndoubl, iface, tau_sum, albedo, nVza, node_1based,
                             cos_mphi, sin_mphi)))
      return fail(h, rc, momf_error(h->f32));
    h->Nz = Nz; h->K = K; h->scene_M = M; h->nVza = nVza; h->albedo = albedo; h->surf_kind = 0;
    h->nd.assign(ndoubl, ndoubl + Nz);
    h->iface.assign(iface, iface + Nz);
    h->scene_set = true;
    return MOM_OK;
  }
  if ((rc = upload_new(h, &h->d_tau, tau, S * Nz))) return rc;
  if ((rc = upload_new(h, &h->d_varpi, varpi, S * Nz))) return rc;
  if ((rc = upload_new(h, &h->d_zw, zw, (size_t)K * S * Nz))) return rc;
  if ((rc = upload_new(h, &h->d_tau_sum, tau_sum, S * (Nz + 1)))) return rc;
  if ((rc = scene_common(h, Nz, K, M, Zpp, Zmp, albedo, nVza, node_1based, cos_mphi, sin_mphi))) return rc;
  h->nd.assign(ndoubl, ndoubl + Nz);
  h->iface.assign(iface, iface + Nz);
  h->scene_set = true;
  return MOM_OK;
}

static int scene_common(mom_t *h, int Nz, int K, int M, const double *Zpp, const double *Zmp, double albedo, int nVza,
                        const int *node_1based, const double *cos_mphi, const double *sin_mphi) {
  for (int v = 0; v < nVza; ++v)
    if (node_1based[v] < 1 || node_1based[v] * h->nS > h->N) return fail(h, MOM_EINVAL, "mom_scene_set: bad view node");
  const size_t S = h->S, NN = (size_t)h->N * h->N;
  int rc;
  // a new scene: the partials of the previous one (mom_scene_set_partials) do not belong to it
  for (int k = 0; k < 8; ++k)
    if (h->d_dual_in[k]) { (void)hipFree(h->d_dual_in[k]); h->d_dual_in[k] = nullptr; }
  h->dual_P = 0; h->dual_ran = false; h->dual_last = false;
  // (edges up to 32 belong to the wave-per-point kernel, which takes the operators as they are)
  const int Nk = (h->opt_pad && !(h->N <= 32 && h->opt_small)) ? strip_pad(h->N) : h->N;
  h->Nk = Nk;
  h->qk = h->q;
  h->qk.N = Nk;
  if (Nk == h->N) {
    if ((rc = upload_new(h, &h->d_Zpp, Zpp, NN * K * M))) return rc;
    if ((rc = upload_new(h, &h->d_Zmp, Zmp, NN * K * M))) return rc;
  } else {
    const std::vector<double> zp = pad_blocks(Zpp, h->N, Nk, (size_t)K * M), zm = pad_blocks(Zmp, h->N, Nk, (size_t)K * M);
    if ((rc = upload_new(h, &h->d_Zpp, zp.data(), zp.size()))) return rc;
    if ((rc = upload_new(h, &h->d_Zmp, zm.data(), zm.size()))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the padded host copies go out of scope
  }
  if ((rc = upload_new(h, &h->d_node, node_1based, (size_t)nVza))) return rc;
  if ((rc = upload_new(h, &h->d_cos, cos_mphi, (size_t)nVza * M))) return rc;
  if ((rc = upload_new(h, &h->d_sin, sin_mphi, (size_t)nVza * M))) return rc;
  if (h->d_R) { (void)hipFree(h->d_R); (void)hipFree(h->d_hdr); h->d_R = h->d_T = h->d_hdr = nullptr; }
  // R_SFI || T_SFI in ONE buffer: it is the send buffer of the all-gather (mom_allgather_RT) as it stands
  HIPCHK(h, dmalloc(&h->d_R, 2 * (size_t)nVza * h->nS * S));
  h->d_T = h->d_R + (size_t)nVza * h->nS * S;
  HIPCHK(h, dmalloc(&h->d_hdr, (size_t)nVza * h->nS * S));
  if (!h->d_hdrJ) {
    HIPCHK(h, dmalloc(&h->d_hdrJ, (size_t)(h->N + kPadMax) * S));
    HIPCHK(h, dmalloc(&h->d_bhr_uw, (size_t)h->nS * S));
    HIPCHK(h, dmalloc(&h->d_bhr_dw, (size_t)h->nS * S));
  }
  // ---- m = 0 reduction (include/momcore.h): conditions checked on the data, bitwise
  {
    const int N = h->N, nS = h->nS, Nq = N / nS;
    bool ok = h->opt_m0 && nS >= 3 && h->q.regular && !(N <= 4 && h->opt_small && nVza <= 4 && K <= 4);
    for (int k = 2; k < nS && ok; ++k) ok = (h->q.I0[k] == 0.0);
    for (int kb = 0; kb < K && ok; ++kb)
      for (int j = 0; j < N && ok; ++j)
        for (int i = 0; i < N; ++i) {
          if (((i % nS) < 2) == ((j % nS) < 2)) continue;
          const size_t o = i + (size_t)N * (j + (size_t)N * kb);  // moment 0 block
          if (Zpp[o] != 0.0 || Zmp[o] != 0.0) { ok = false; break; }
        }
    auto fr = [](double *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    fr(h->d_mu0); fr(h->d_wt0); fr(h->d_sg0); fr(h->d_Zpp0); fr(h->d_Zmp0); fr(h->d_hdrJ0); fr(h->d_scratch0);
    for (int k = 0; k < 6; ++k) fr(h->comp0[k]);
    h->red0 = ok;
    if (ok) {
      // N0r real entries; the kernels run on N0 >= N0r (dummy entries of strip_pad at the end: mu = 1, weight 0, Z = 0)
      const int nS0 = 2, N0r = nS0 * Nq;
      int N0 = h->opt_pad ? strip_pad(N0r) : N0r;
      // r6: sub-problems of edge 18 .. 30 that are not a multiple of 4 take ONE dummy stream (two entries) to reach a quad-block
      // size (20, 24, 28, 32: mom_q4.hpp; IQUV scenes of 9 .. 15 streams)
      if (h->opt_pad && h->opt_lean >= 3 && N0 == N0r && N0r > 16 && N0r < 32 && (N0r % 4) != 0) N0 = N0r + 2;
      h->N0 = N0; h->nS0 = nS0;
      std::vector<double> mu0v(N0, 1.0), wt0v(N0, 0.0), sg0v(N0, 1.0), zp((size_t)N0 * N0 * K, 0.0), zm((size_t)N0 * N0 * K, 0.0);
      auto full = [&](int i0) { return (i0 / nS0) * nS + (i0 % nS0); };
      for (int i = 0; i < N0r; ++i) { mu0v[i] = h->h_mu[full(i)]; wt0v[i] = h->h_wt[full(i)]; }
      for (int kb = 0; kb < K; ++kb)
        for (int j = 0; j < N0r; ++j)
          for (int i = 0; i < N0r; ++i) {
            const size_t src = full(i) + (size_t)N * (full(j) + (size_t)N * kb);
            zp[i + (size_t)N0 * (j + (size_t)N0 * kb)] = Zpp[src];
            zm[i + (size_t)N0 * (j + (size_t)N0 * kb)] = Zmp[src];
          }
      if ((rc = upload_new(h, &h->d_mu0, mu0v.data(), (size_t)N0))) return rc;
      if ((rc = upload_new(h, &h->d_wt0, wt0v.data(), (size_t)N0))) return rc;
      if ((rc = upload_new(h, &h->d_sg0, sg0v.data(), (size_t)N0))) return rc;
      if ((rc = upload_new(h, &h->d_Zpp0, zp.data(), zp.size()))) return rc;
      if ((rc = upload_new(h, &h->d_Zmp0, zm.data(), zm.size()))) return rc;
      for (int k = 0; k < 6; ++k) {
        const size_t cnt = ((k < 4) ? (size_t)comp_pitch(N0) * N0 : (size_t)N0) * S;
        HIPCHK(h, dmalloc(&h->comp0[k], cnt));
        HIPCHK(h, hipMemsetAsync(h->comp0[k], 0, cnt * sizeof(double), h->stream));
      }
      HIPCHK(h, dmalloc(&h->d_hdrJ0, (size_t)N0 * S));
      const size_t scr = (size_t)h->G * kGenericBufs * mat_elems(N0) + (size_t)ld_for(N0) * np_for(N0);
      HIPCHK(h, dmalloc(&h->d_scratch0, scr));
      HIPCHK(h, hipMemsetAsync(h->d_scratch0, 0, scr * sizeof(double), h->stream));
      HIPCHK(h, hipMemsetAsync(h->d_bhr_uw, 0, (size_t)h->nS * S * sizeof(double), h->stream));
      HIPCHK(h, hipMemsetAsync(h->d_bhr_dw, 0, (size_t)h->nS * S * sizeof(double), h->stream));
      DevStreams &q0 = h->q0;
      q0 = h->q;
      q0.mu = h->d_mu0; q0.wt = h->d_wt0; q0.sg = h->d_sg0;
      q0.N = N0; q0.nS = nS0;
      for (int k = nS0; k < 4; ++k) { q0.I0[k] = 0.0; q0.D[k] = 1.0; }
    }
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->Nz = Nz; h->K = K; h->scene_M = M; h->nVza = nVza; h->albedo = albedo;
  h->surf_kind = 0;  // LambertianSurfaceScalar(albedo) until mom_scene_set_surface says otherwise
  return MOM_OK;
}

extern "C" int mom_scene_set_surface(mom_t *h, int kind, int M, const double *Rsurf, const double *albedo_spec) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_scene_set_surface: call mom_scene_set first");
  if (kind < 0 || kind > 2 || (kind == 1 && (!Rsurf || M != h->scene_M)) || (kind == 2 && !albedo_spec))
    return fail(h, MOM_EINVAL, "mom_scene_set_surface: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const int N = h->N, nS = h->nS;
  const size_t NN = (size_t)N * N, S = h->S;
  int rc;
  if (h->f32) {
    if ((rc = momf_scene_set_surface(h->f32, kind, M, Rsurf, albedo_spec))) return fail(h, rc, momf_error(h->f32));
    h->surf_kind = kind;
    return MOM_OK;
  }
  if (kind == 1) {
    if (h->Nk == N) {
      if ((rc = upload_new(h, &h->d_Rsurf, Rsurf, NN * M))) return rc;
    } else {
      const std::vector<double> rp = pad_blocks(Rsurf, N, h->Nk, (size_t)M);
      if ((rc = upload_new(h, &h->d_Rsurf, rp.data(), rp.size()))) return rc;
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (h->d_hdrJm) { (void)hipFree(h->d_hdrJm); h->d_hdrJm = nullptr; }
    HIPCHK(h, dmalloc(&h->d_hdrJm, (size_t)h->Nk * S * M));
    if (h->red0) {
      // moment 0 runs on the (I,Q) sub-problem: its surface matrix must not couple (I,Q) with (U,V) either
      const int nS0 = h->nS0, N0 = h->N0;
      std::vector<double> r0((size_t)N0 * N0);
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
          const bool iq_i = (i % nS) < nS0, iq_j = (j % nS) < nS0;
          const double v = Rsurf[i + (size_t)N * j];
          if (iq_i != iq_j && v != 0.0)
            return fail(h, MOM_EINVAL, "mom_scene_set_surface: the m = 0 BRDF matrix couples (I,Q) with (U,V); set "
                                       "MOM_OPT_M0_REDUCTION = 0 before mom_scene_set for this surface");
          if (iq_i && iq_j) r0[(i / nS) * nS0 + (i % nS) + (size_t)N0 * ((j / nS) * nS0 + (j % nS))] = v;
        }
      if ((rc = upload_new(h, &h->d_Rsurf0, r0.data(), r0.size()))) return rc;
    }
  } else if (kind == 2) {
    if ((rc = upload_new(h, &h->d_albedo_spec, albedo_spec, S))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->surf_kind = kind;
  return MOM_OK;
}

using SmallSweepArgs = MomSmallSweepArgs;  // mom_host.hpp

// N <= 4: one spectral point per lane, all moments / layers / surface / post-processing in ONE launch
static int rt_run_small(mom_t *h) {
  const int N = h->N, Nz = h->Nz;
  if (!h->d_smtab) HIPCHK(h, dmalloc(&h->d_smtab, 3 * 16));
  {  // mu_j/(mu_i + mu_j), mu_j/(mu_i - mu_j), (1/mu_i) + (1/mu_j): the expressions of elemental.jl:176-186, evaluated once
    double tab[48] = {0};
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        const double mui = h->h_mu[i], muj = h->h_mu[j];
        tab[i + N * j] = muj / (mui + muj);
        tab[16 + i + N * j] = muj / (mui - muj);
        tab[32 + i + N * j] = (1 / mui) + (1 / muj);
      }
    HIPCHK(h, hipMemcpyAsync(h->d_smtab, tab, sizeof tab, hipMemcpyHostToDevice, h->stream));
  }
  {
    if (h->ndif_cap < 2 * (size_t)Nz) {  // no allocation in steady state
      if (h->d_ndif) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_ndif); h->d_ndif = nullptr; }
      HIPCHK(h, dmalloc(&h->d_ndif, 2 * (size_t)Nz));
      h->ndif_cap = 2 * (size_t)Nz;
    }
    std::vector<int> v(h->nd);
    v.insert(v.end(), h->iface.begin(), h->iface.end());
    HIPCHK(h, hipMemcpyAsync(h->d_ndif, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the host vectors above go out of scope
  }
  SmallSweepArgs a{};
  a.S = h->S; a.M = h->scene_M; a.K = h->K; a.Nz = Nz; a.nVza = h->nVza; a.nS = h->nS; a.imu0 = h->q.imu0;
  a.mu0 = h->q.mu0; a.albedo = h->albedo;
  for (int k = 0; k < 4; ++k) { a.I0[k] = h->q.I0[k]; a.D[k] = h->q.D[k]; }
  a.mu = h->d_mu; a.wt = h->d_wt; a.sg = h->d_sg;
  a.F1 = h->d_smtab; a.F2 = h->d_smtab + 16; a.SI = h->d_smtab + 32;
  a.Zpp = h->d_Zpp; a.Zmp = h->d_Zmp;
  a.nd = h->d_ndif; a.iface = h->d_ndif + Nz; a.node = h->d_node; a.cos_mphi = h->d_cos; a.sin_mphi = h->d_sin;
  a.tau = h->d_tau; a.varpi = h->d_varpi; a.zw = h->d_zw; a.tau_sum = h->d_tau_sum;
  a.R = h->d_R; a.T = h->d_T; a.hdr = h->d_hdr; a.bhr_uw = h->d_bhr_uw; a.bhr_dw = h->d_bhr_dw;
  a.info = h->d_info;
  if (h->K > 4) return fail(h, MOM_EINVAL, "mom_rt_run: the N <= 4 sweep kernel handles at most 4 phase-matrix bases");
  if (a.M > 1 && h->opt_small != 2) {  // one (point, moment) per lane (mom_small.hip SPLIT); MOM_OPT_SMALL_N = 2: one point per lane
    const size_t need = (size_t)a.M * 2 * a.nVza * a.nS * a.S;
    if (need > h->smpart_cap) {  // grow-only: no allocation in steady state
      if (h->d_smpart) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_smpart); h->d_smpart = nullptr; h->smpart_cap = 0; }
      HIPCHK(h, dmalloc(&h->d_smpart, need));
      h->smpart_cap = need;
    }
    a.part = h->d_smpart;
  }
  while (h->ev_full.size() < 2) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->ev_full.push_back(e); }
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  HIPCHK(h, hipEventRecord(h->ev_full[0], h->stream));
  HIPCHK(h, momsm_launch_sweep(&a, N, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_full[1], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  h->launches = 1; h->launches_full = 1; h->launches_red = 0;
  return MOM_OK;
}

using WaveSweepArgs = MomWaveSweepArgs;  // mom_host.hpp

// the wave-per-point kernel covers ScatteringInterface_11 on every layer after the first and at the surface
static bool wave_sweep_applies(const mom_t *h) {
  if (!(h->N > 4 && h->N <= 32 && h->opt_small && !h->opt_force_generic && h->nVza * h->nS <= 256)) return false;
  for (int z = 1; z < h->Nz; ++z)
    if (h->iface[z] != 3) return false;
  return h->iface[h->Nz - 1] == 3;
}

// 4 < N <= 32: one spectral point per wavefront, operators in MFMA-layout registers, ONE launch
static int rt_run_wave(mom_t *h) {
  const int Nz = h->Nz;
  {
    if (h->ndif_cap < 2 * (size_t)Nz) {  // no allocation in steady state
      if (h->d_ndif) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_ndif); h->d_ndif = nullptr; }
      HIPCHK(h, dmalloc(&h->d_ndif, 2 * (size_t)Nz));
      h->ndif_cap = 2 * (size_t)Nz;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_ndif, h->nd.data(), (size_t)Nz * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  WaveSweepArgs a{};
  a.N = h->N; a.S = h->S; a.M = h->scene_M; a.K = h->K; a.Nz = Nz; a.nVza = h->nVza; a.nS = h->nS; a.imu0 = h->q.imu0;
  a.inv_mode = h->opt_inverse;
  // points per wavefront (mom_wave.hip, block-diagonal packing): MOM_OPT_SMALL_N = 2 keeps one point per wave
  a.pad = (h->opt_small == 1) ? (h->N == 5 ? 3 : (h->N >= 6 && h->N <= 8 ? 2 : 1)) : 1;
  a.mu0 = h->q.mu0; a.albedo = h->albedo;
  for (int k = 0; k < 4; ++k) { a.I0[k] = h->q.I0[k]; a.D[k] = h->q.D[k]; }
  a.mu = h->d_mu; a.wt = h->d_wt; a.sg = h->d_sg;
  a.Zpp = h->d_Zpp; a.Zmp = h->d_Zmp;
  a.nd = h->d_ndif; a.node = h->d_node; a.cos_mphi = h->d_cos; a.sin_mphi = h->d_sin;
  a.tau = h->d_tau; a.varpi = h->d_varpi; a.zw = h->d_zw; a.tau_sum = h->d_tau_sum;
  a.R = h->d_R; a.T = h->d_T; a.hdr = h->d_hdr; a.bhr_uw = h->d_bhr_uw; a.bhr_dw = h->d_bhr_dw;
  a.info = h->d_info;
  a.surf_kind = h->surf_kind; a.Rsurf = h->d_Rsurf; a.albedo_spec = h->d_albedo_spec;
  while (h->ev_full.size() < 2) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->ev_full.push_back(e); }
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  HIPCHK(h, hipEventRecord(h->ev_full[0], h->stream));
  HIPCHK(h, momw_launch_sweep(&a, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_full[1], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // h->nd may be rewritten by the next scene_set
  h->launches = 1; h->launches_full = 1; h->launches_red = 0;
  return MOM_OK;
}

// The general path of mom_rt_run for the layers [za, zb) of the column into the composite state `compF` (full problem;
// the m = 0 sub-problem keeps its own arrays and is used only when allow_red): layer kernels, then (do_surface) the
// surface layer with its closing interaction, then (do_post) the azimuthal post-processing into d_R / d_T / d_hdr.
// mom_rt_run: the whole column; mom_rt_run_multisensor: the slabs above and below a sensor.
// multi-target sweep (mom_rt_run_multisensor): composite targets and the per-layer action table of LayerArgs
struct TargetSpec {
  int ntgt = 0;
  double *tgt[kMaxTargets][6] = {};
  std::vector<signed char> act;  // [zb - za][kMaxTargets]
};

static int rt_run_core(mom_t *h, int za, int zb, bool allow_red, double *const compF[6], bool do_surface, bool do_post,
                       bool cont = false, const TargetSpec *tg = nullptr) {
  const size_t S = h->S;
  const int M = h->scene_M;
  const bool red0 = h->red0 && allow_red;
  const int nzr = zb - za;
  while (h->ev_full.size() < 2 * (size_t)h->Nz + 2) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->ev_full.push_back(e); }
  while (h->ev_red.size() < 2 * (size_t)h->Nz + 2) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->ev_red.push_back(e); }
  const int Nk = h->Nk;  // kernel-side edge of the full problem (strip_pad)
  const size_t NN = (size_t)Nk * Nk;
  // one k_layer launch over `Mcount` moments starting at `m_first` with stream set `q` (full or reduced)
  // sweep mode: every layer of a unit inside one launch (z < 0 selects it); needs one interface code for all z >= 1
  // (the code is a template argument of the kernel images) -- always the case once scattering has set in
  // cont: the slab continues the composite state already in compF (its first layer interacts like the others)
  bool can_sweep = h->opt_sweep && nzr <= kMaxSweepLayers && nzr > 1;
  for (int z = za + 2; z < zb && can_sweep; ++z) can_sweep = (h->iface[z] == h->iface[za + 1]);
  if (cont && can_sweep) can_sweep = (h->iface[za] == h->iface[za + 1]);
  for (int z = za; z < zb && can_sweep; ++z) can_sweep = (h->nd[z] <= 127);
  hipStream_t cur = h->stream;  // the stream launch_layer issues to (MOM_OPT_OVERLAP switches it for the m = 0 sub-problem)
  auto launch_layer = [&](int z, const DevStreams &q, int m_first, int Mcount, const double *Zpp, const double *Zmp,
                          double *const comp[6], double *scratch) -> int {
    LayerArgs a{};
    a.q = q; a.S = h->S; a.M = Mcount; a.K = h->K; a.m_first = m_first;
    const bool sweep = z < 0;
    if (sweep) {
      z = za;
      a.Nz_sweep = nzr;
      int ndsum = 0;
      for (int k = 0; k < nzr; ++k) { a.nd_z[k] = (signed char)h->nd[za + k]; a.iface_z[k] = (signed char)h->iface[za + k]; ndsum += h->nd[za + k]; }
      a.nd = ndsum / nzr; a.iface = h->iface[za + 1]; a.first = cont ? 0 : 1;
    } else {
      a.nd = h->nd[z]; a.iface = h->iface[z]; a.first = (z == za) && !cont;
    }
    a.tau = h->d_tau + S * z; a.varpi = h->d_varpi + S * z; a.zw = h->d_zw + (size_t)h->K * S * z;
    a.tau_sum = h->d_tau_sum + S * z;
    a.Zpp = Zpp; a.Zmp = Zmp;
    for (int k = 0; k < 6; ++k) a.comp[k] = comp[k];
    if (tg) {  // every moment of a target lies m_first moments into its arrays, like comp
      a.ntgt = tg->ntgt;
      for (int t = 0; t < tg->ntgt; ++t)
        for (int k = 0; k < 6; ++k) a.tgt[t][k] = tg->tgt[t][k];
      const int rows = sweep ? nzr : 1, r0 = sweep ? 0 : (z - za);
      for (int r = 0; r < rows; ++r)
        for (int t = 0; t < kMaxTargets; ++t) a.act_z[r][t] = tg->act[(size_t)(r0 + r) * kMaxTargets + t];
    }
    a.scratch = scratch; a.info = h->d_info;
    const bool lds = (q.N <= 64) && !h->opt_force_generic;
    // small operators: 4-wave workgroups, two per CU (momcore_w4.hip), when two LDS images fit
    const int ns_tab = q.regular ? q.nS : 1;  // Stokes components per stream of the elemental layer's stream-pair tables
    const bool strip4 = (q.N == 36 || q.N == 40 || q.N == 44);
    if (lds && h->opt_w4 && np_for(q.N) <= 48 &&
        2 * (strip4 ? mom4_strip_lds_bytes(q.N, ns_tab) : mom4_lds_bytes(q.N, true)) + 2048 <= 160 * 1024) {
      const int grid4 = (int)((S >= 2048) ? S : S * Mcount);
      // operator edges 20 .. 32 (multiples of 4) on the quad-block image first (mom_q4.hpp), the general 4-wave image finishes what
      // it left; the edges 36 / 40 take the same route below, with the strip image as the finisher
      if (!strip4 && h->opt_lean >= 3 && sweep && !tg && q.inv_mode == 0 && q4_image_lds(q.N, ns_tab, h->K) > 0) {
        bool quad = true;
        for (int k = 1; k < nzr && quad; ++k) quad = (a.iface_z[k] == 3);
        if (quad && !a.first) quad = (a.iface_z[0] == 3);
        if (quad) {
          const size_t units = S * (size_t)Mcount;
          if (units > h->resume_cap) {  // grow-only
            if (h->d_resume) { HIPCHK(h, hipStreamSynchronize(cur)); (void)hipFree(h->d_resume); h->d_resume = nullptr; h->resume_cap = 0; }
            HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_resume), units * sizeof(int)));
            h->resume_cap = units;
          }
          a.resume = h->d_resume;
          HIPCHK(h, q4_image_launch(q.N, &a, h->num_cu, units, cur));
          h->launches++;
        }
      }
      if (strip4) {  // strip-chained kernels of the 4-wave build (momcore_strip.hip)
        const int gridp = (int)std::min<size_t>(S * Mcount, (size_t)2 * h->num_cu);  // persistent, two per CU
        // N = 36, 40: the lean image first (three workgroups per CU; mom_lean.hpp), then the full image resumes what it left
        bool lean = h->opt_lean && sweep && !tg && q.inv_mode == 0 && (q.N == 36 || q.N == 40) &&
                    (q.N == 40 ? mom_strip10_lean_lds_bytes(ns_tab) : mom_strip9_lean_lds_bytes(ns_tab)) > 0;
        for (int k = 1; k < nzr && lean; ++k) lean = (a.iface_z[k] == 3);
        if (lean && !a.first) lean = (a.iface_z[0] == 3);
        if (lean) {
          const size_t units = S * (size_t)Mcount;
          if (units > h->resume_cap) {  // grow-only
            if (h->d_resume) { HIPCHK(h, hipStreamSynchronize(cur)); (void)hipFree(h->d_resume); h->d_resume = nullptr; h->resume_cap = 0; }
            HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_resume), units * sizeof(int)));
            h->resume_cap = units;
          }
          a.resume = h->d_resume;
          const bool quad = h->opt_lean >= 3 && q4_image_lds(q.N, ns_tab, h->K) > 0;
          const bool six = !quad && h->opt_lean == 2 && (q.N == 40 ? mom6_lean10_lds_bytes(ns_tab) : mom6_lean9_lds_bytes(ns_tab)) > 0;
#ifdef MOM_EXPERIMENTS
          static const int lean_per_cu = getenv("MOM_LEAN_PER_CU") ? atoi(getenv("MOM_LEAN_PER_CU")) : 0;
#else
          const int lean_per_cu = 0;
#endif
          const int per_cu = lean_per_cu > 0 ? lean_per_cu : (six ? 2 : 3);
          const int gridl = (int)std::min<size_t>(units, (size_t)per_cu * h->num_cu);
          if (quad) HIPCHK(h, q4_image_launch(q.N, &a, h->num_cu, units, cur));
          else if (six) HIPCHK(h, (q.N == 40 ? mom6_lean10_launch : mom6_lean9_launch)(&a, gridl, cur));
          else HIPCHK(h, (q.N == 40 ? mom_strip10_launch_lean : mom_strip9_launch_lean)(&a, gridl, cur));
          h->launches++;
        }
        HIPCHK(h, (q.N == 44 ? mom4_strip11_launch_layer : q.N == 40 ? mom_strip10_launch_layer : mom_strip9_launch_layer)(
                      &a, a.iface, gridp, mom4_strip_lds_bytes(q.N, ns_tab), cur));
        h->launches++;
        return MOM_OK;
      }
      HIPCHK(h, mom4_launch_layer(&a, a.iface, true, grid4, mom4_lds_bytes(q.N, true), cur));
      h->launches++;
      return MOM_OK;
    }
    size_t sm = lds_bytes(q.N, lds);
    if (lds && (q.N == 44 || q.N == 52 || q.N == 56 || q.N == 60)) {  // strip-chained kernels (momcore_strip.hip), one image per N
      sm = strip_lds_bytes(q.N, ns_tab);  // + the persistent stream-pair tables
      // persistent workgroups, one per CU (only one 135 KB LDS image fits a CU): the prologue is paid once;
      // their start is staggered over about one unit time (~ (44 + 17 nd) us at N = 60)
      if (S * Mcount >= 8 * (size_t)h->num_cu && h->opt_stagger) {
        const double f = (double)q.N / 60.0, unit_us = f * f * f * (44.0 + 17.0 * a.nd);
        a.stagger = (int)(unit_us * 100.0 / 32.0);
      }
      const int grid = (int)std::min<size_t>(S * Mcount, (size_t)h->num_cu);
      HIPCHK(h, (q.N == 60 ? mom_strip15_launch_layer : q.N == 56 ? mom_strip14_launch_layer : q.N == 52 ? mom_strip13_launch_layer : mom_strip11_launch_layer)(
                    &a, a.iface, grid, sm, cur));
      h->launches++;
      return MOM_OK;
    }
    const int grid = lds ? (int)((S >= 2048) ? S : S * Mcount) : (int)std::min<size_t>(S * Mcount, (size_t)h->G);
    HIPCHK(h, mom_gen_launch_layer(&a, a.iface, lds, grid, sm, cur));
    HIPCHK(h, hipGetLastError());
    h->launches++;
    return MOM_OK;
  };
  // MOM_OPT_OVERLAP: the two launches of a sweep -- moments 1..M-1 on the full problem, moment 0 on the (I,Q) sub-problem --
  // are independent, and each ends in a partial round of its persistent workgroups (C2: 20 000 units on 256 workgroups = 78.1
  // rounds, 10 000 on 768 = 13.02).  The sub-problem (with its surface interaction) goes first, on the handle's second,
  // high-priority stream; the full problem's workgroups take the CUs as the sub-problem's last round frees them, and the
  // workgroups that start late are those with the highest indices -- the ones WITHOUT a unit in the full problem's own partial
  // round.  The images cannot share a CU (149 + 48.5 KB of LDS), so nothing else overlaps.
  // ... and only then: where the two kernels CAN share a CU they contend for its matrix pipes and LDS bandwidth and the sweep
  // takes longer than the two launches in sequence (profiles/r06_C2_ab.txt (b'): N = 36 .. 44 with the m = 0 problem on the
  // wave-per-point kernel 28 -> 45 ms, C4 -5 %), so the overlap is reserved for the persistent one-per-CU strip images
  const bool two = red0 && can_sweep && h->opt_overlap && M > 1 && !tg && h->stream2 && !h->opt_force_generic &&
                   (Nk == 52 || Nk == 56 || Nk == 60);
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  for (int z = (can_sweep ? -1 : za); z < (can_sweep ? 0 : zb); ++z) {
    int rc;
    const int e = can_sweep ? 0 : z - za;  // event slot
    if (red0) {
      if (two) {  // the m = 0 sub-problem first, on the high-priority stream: see the comment at `two`
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        cur = h->stream2;
        // the full problem's launch is released only when the second stream has passed its own wait and stands right before
        // the sub-problem's launch: otherwise the main stream (no wait packet in front of its kernel) always dispatches first
        HIPCHK(h, hipEventRecord(h->ev_go, cur));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_go, 0));
        HIPCHK(h, hipEventRecord(h->ev_red[2 * e], cur));
        if ((rc = launch_layer(z, h->q0, 0, 1, h->d_Zpp0, h->d_Zmp0, h->comp0, h->d_scratch0))) return rc;
        HIPCHK(h, hipEventRecord(h->ev_red[2 * e + 1], cur));
        h->launches_red++;
        cur = h->stream;
      }
      if (M > 1) {  // moments 1..M-1 on the full problem
        double *comp1[6];
        for (int k = 0; k < 6; ++k) comp1[k] = compF[k] + ((k < 4) ? (size_t)comp_pitch(Nk) * Nk : (size_t)Nk) * S;
        HIPCHK(h, hipEventRecord(h->ev_full[2 * e], h->stream));
        if ((rc = launch_layer(z, h->qk, 1, M - 1, h->d_Zpp + NN * h->K, h->d_Zmp + NN * h->K, comp1, h->d_scratch))) return rc;
        HIPCHK(h, hipEventRecord(h->ev_full[2 * e + 1], h->stream));
        h->launches_full++;
      }
      if (!two) {
        HIPCHK(h, hipEventRecord(h->ev_red[2 * e], h->stream));
        if ((rc = launch_layer(z, h->q0, 0, 1, h->d_Zpp0, h->d_Zmp0, h->comp0, h->d_scratch0))) return rc;
        HIPCHK(h, hipEventRecord(h->ev_red[2 * e + 1], h->stream));
        h->launches_red++;
      }
    } else {
      HIPCHK(h, hipEventRecord(h->ev_full[2 * e], h->stream));
      if ((rc = launch_layer(z, h->qk, 0, M, h->d_Zpp, h->d_Zmp, compF, h->d_scratch))) return rc;
      HIPCHK(h, hipEventRecord(h->ev_full[2 * e + 1], h->stream));
      h->launches_full++;
    }
  }
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  // surface layer + closing interaction: m = 0 always; every moment for a BRDF surface (kind 1)
  for (int m = 0; do_surface && m < ((h->surf_kind == 1) ? M : 1); ++m) {
    SurfArgs a{};
    const bool red = red0 && m == 0;
    const DevStreams &q = red ? h->q0 : h->qk;
    a.q = q; a.S = h->S; a.iface = h->iface[h->Nz - 1];  // Q6: last layer's interface code (rt_run.jl:181)
    a.albedo = h->albedo; a.tau_tot = h->d_tau_sum + S * h->Nz;
    a.kind = h->surf_kind; a.m = m; a.albedo_spec = h->d_albedo_spec;
    a.Rsurf = (h->surf_kind == 1) ? (red ? h->d_Rsurf0 : h->d_Rsurf + NN * m) : nullptr;
    for (int k = 0; k < 6; ++k)
      a.comp[k] = red ? h->comp0[k] : compF[k] + ((k < 4) ? (size_t)comp_pitch(Nk) * Nk : (size_t)Nk) * S * m;
    a.hdrJ = red ? h->d_hdrJ0 : (m == 0 ? h->d_hdrJ : h->d_hdrJm + (size_t)Nk * S * m);
    a.bhr_uw = h->d_bhr_uw; a.bhr_dw = h->d_bhr_dw; a.nS_out = h->nS;
    a.scratch = red ? h->d_scratch0 : h->d_scratch; a.info = h->d_info;
    const bool lds = (q.N <= 64) && !h->opt_force_generic;
    const size_t sm = lds_bytes(q.N, lds);
    const int grid = lds ? (int)S : (int)std::min<size_t>(S, (size_t)h->G);
    const hipStream_t sst = (two && red) ? h->stream2 : h->stream;  // the sub-problem's surface follows its layers
    if (lds && h->opt_w4 && np_for(q.N) <= 48 && 2 * mom4_lds_bytes(q.N, true) + 2048 <= 160 * 1024) {
      HIPCHK(h, mom4_launch_surface(&a, true, (int)S, mom4_lds_bytes(q.N, true), sst));
    } else if (lds) {
      HIPCHK(h, allow_lds(k_surface<true>, sm));
      hipLaunchKernelGGL(k_surface<true>, dim3(grid), dim3(kThreads), sm, sst, a);
    } else {
      HIPCHK(h, allow_lds(k_surface<false>, sm));
      hipLaunchKernelGGL(k_surface<false>, dim3(grid), dim3(kThreads), sm, sst, a);
    }
    HIPCHK(h, hipGetLastError());
  }
  if (two) {  // join: everything below (post-processing, the caller's downloads) is ordered behind both streams
    HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  }
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  if (do_post) {
    const size_t total = (size_t)h->nVza * h->nS * S;
    PostArgs pa{};
    pa.N = Nk; pa.nS = h->nS; pa.S = h->S; pa.M = M; pa.nVza = h->nVza; pa.red0 = red0 ? 1 : 0;
    pa.N0 = h->N0; pa.nS0 = h->nS0;
    pa.node = h->d_node; pa.cos_mphi = h->d_cos; pa.sin_mphi = h->d_sin;
    pa.J0p = compF[4]; pa.J0m = compF[5]; pa.J0p0 = h->comp0[4]; pa.J0m0 = h->comp0[5];
    pa.hdrJ = red0 ? h->d_hdrJ0 : h->d_hdrJ;
    pa.hdr_all = (h->surf_kind == 1) ? 1 : 0; pa.zeroT_hi = (h->surf_kind == 2) ? 1 : 0; pa.hdrJm = h->d_hdrJm;
    pa.R = h->d_R; pa.T = h->d_T; pa.hdr = h->d_hdr;
    hipLaunchKernelGGL(k_postprocess, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, pa);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  return MOM_OK;
}

extern "C" int mom_rt_run(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_rt_run: call mom_scene_set first");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->f32) {
    const int rc = momf_rt_run(h->f32);
    return rc ? fail(h, rc, momf_error(h->f32)) : MOM_OK;
  }
  h->launches = 0; h->launches_full = 0; h->launches_red = 0;
  h->comp_pitched = true;
  h->comp_on_chip = true;
  h->dual_last = false;
  if (h->N <= 4 && h->opt_small && !h->opt_force_generic && h->nVza <= 4 && h->surf_kind == 0 && h->K <= 4) return rt_run_small(h);
  if (wave_sweep_applies(h)) return rt_run_wave(h);
  h->comp_on_chip = false;
  return rt_run_core(h, 0, h->Nz, true, h->comp, true, true);
}

// rt_run_test_ms(::noRS, sensor_levels, model, iBand) (rt_run_multisensor.jl:14-191).  Sensors are processed one after
// the other (in order of depth) with two composite states: the slab above the sensor (layers 1..L) and the slab below it (layers L+1..Nz and
// the surface), each built by the same fused layer kernels as mom_rt_run (sweep mode, strip chains, padded edges), then
// k_interlayer and the azimuthal post-processing of the interface fields.  The m = 0 (I,Q) reduction is not used here
// (the interface fields couple two states of the full problem); level 0 is mom_rt_run itself.
extern "C" int mom_rt_run_multisensor(mom_t *h, int nSensors, const int *sensor_levels, double *uwJ, double *dwJ) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_rt_run_multisensor");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_rt_run_multisensor: call mom_scene_set first");
  if (nSensors <= 0 || !sensor_levels || !uwJ || !dwJ) return fail(h, MOM_EINVAL, "mom_rt_run_multisensor: bad argument");
  for (int i = 0; i < nSensors; ++i)
    if (sensor_levels[i] < 0 || sensor_levels[i] >= h->Nz)
      return fail(h, MOM_EINVAL, "mom_rt_run_multisensor: sensor level must be in 0..Nz-1 (0 = TOA/BOA, L = below layer L)");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t S = h->S;
  const int M = h->scene_M, Nk = h->Nk;
  const size_t out1 = (size_t)h->nVza * h->nS * S;
  h->launches = 0; h->launches_full = 0; h->launches_red = 0;
  h->comp_pitched = true;
  h->comp_on_chip = false;
  if (!h->comp_top[0]) {
    const int Na = h->N + kPadMax;
    for (int k = 0; k < 6; ++k) {
      const size_t perc = (k < 4) ? (size_t)comp_pitch(Na) * Na : (size_t)Na;
      HIPCHK(h, dmalloc(&h->comp_top[k], perc * S * h->M));
    }
    for (int k = 0; k < 2; ++k) HIPCHK(h, dmalloc(&h->d_msJ[k], (size_t)Na * S * h->M));
  }
  if (h->ms_out_cap < 2 * out1 * nSensors) {
    if (h->d_ms_out) { (void)hipFree(h->d_ms_out); h->d_ms_out = nullptr; }
    HIPCHK(h, dmalloc(&h->d_ms_out, 2 * out1 * nSensors));
    h->ms_out_cap = 2 * out1 * nSensors;
  }
  double *d_uw = h->d_ms_out, *d_dw = h->d_ms_out + out1 * nSensors;
  // rt_kernel_multisensor! (rt_kernel_multisensor.jl:51-112): ONE sweep over the layers builds every layer's added operators
  // once and feeds all composites -- the running slab above the sensors (target 0, frozen into a per-sensor snapshot when
  // the sweep passes the sensor's level) and the slab below each sensor -- then per sensor the surface interaction, the
  // interface solve and the post-processing.  Sensors are processed in chunks of what one kernel's target table holds.
  const int Na = h->N + kPadMax;
  const size_t blk[6] = {(size_t)comp_pitch(Na) * Na, (size_t)comp_pitch(Na) * Na, (size_t)comp_pitch(Na) * Na,
                         (size_t)comp_pitch(Na) * Na, (size_t)Na, (size_t)Na};
  const int per_chunk = (kMaxTargets - 1) / 2;  // top + (snapshot + bottom) per sensor
  for (int c0 = 0; c0 < nSensors; c0 += per_chunk) {
    const int nc = std::min(per_chunk, nSensors - c0);
    const size_t need = (size_t)2 * nc;  // composite sets beyond h->comp_top: nc snapshots + nc bottoms
    if (h->ms_sets < need) {
      for (auto p : h->ms_comp) (void)hipFree(p);
      h->ms_comp.clear();
      h->ms_sets = 0;
      for (size_t sidx = 0; sidx < need; ++sidx)
        for (int k = 0; k < 6; ++k) {
          double *p = nullptr;
          HIPCHK(h, dmalloc(&p, blk[k] * S * h->M));
          h->ms_comp.push_back(p);
        }
      h->ms_sets = need;
    }
    // sensors of this chunk in order of depth: the slab below sensor i is the SEGMENT of layers [L_i, L_i+1) -- built in the
    // shared sweep, so every layer feeds the running top slab and exactly one segment whatever the number of sensors --
    // joined afterwards to the slab below sensor i + 1 (k_combine); the deepest sensor's segment runs to the last layer
    std::vector<int> ord(nc);
    for (int i = 0; i < nc; ++i) ord[i] = c0 + i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return sensor_levels[x] < sensor_levels[y]; });
    TargetSpec tg;
    tg.act.assign((size_t)h->Nz * kMaxTargets, 0);
    int maxL = 0;
    for (int i = 0; i < nc; ++i) maxL = std::max(maxL, sensor_levels[c0 + i]);
    int nt = 0;
    for (int k = 0; k < 6; ++k) tg.tgt[0][k] = h->comp_top[k];
    nt = 1;
    for (int z = 0; z < maxL; ++z) tg.act[(size_t)z * kMaxTargets + 0] = (z == 0) ? 1 : 2;
    std::vector<int> snap_t(nc, -1), bot_t(nc, -1);
    for (int i = 0; i < nc; ++i) {
      const int L = sensor_levels[ord[i]], Lnext = (i + 1 < nc) ? sensor_levels[ord[i + 1]] : h->Nz;
      if (L > 0) {
        snap_t[i] = nt;
        for (int k = 0; k < 6; ++k) tg.tgt[nt][k] = h->ms_comp[(size_t)(2 * i) * 6 + k];
        tg.act[(size_t)(L - 1) * kMaxTargets + nt] = 3;
        ++nt;
      }
      bot_t[i] = nt;
      for (int k = 0; k < 6; ++k) tg.tgt[nt][k] = h->ms_comp[(size_t)(2 * i + 1) * 6 + k];
      for (int z = L; z < Lnext; ++z) tg.act[(size_t)z * kMaxTargets + nt] = (z == L) ? 1 : 2;
      ++nt;
    }
    tg.ntgt = nt;
    int rc;
    if ((rc = rt_run_core(h, 0, h->Nz, false, h->comp, false, false, false, &tg))) return rc;
    for (int i = nc - 2; i >= 0; --i) {  // slab below sensor i = its segment (+) the slab below sensor i + 1
      const int L = sensor_levels[ord[i]], Lnext = sensor_levels[ord[i + 1]];
      if (L == Lnext) {  // same level: same slab
        for (int k = 0; k < 6; ++k)
          HIPCHK(h, hipMemcpyAsync(tg.tgt[bot_t[i]][k], tg.tgt[bot_t[i + 1]][k], blk[k] * S * h->M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        continue;
      }
      InterArgs a{};
      a.q = h->qk; a.S = h->S; a.M = M;
      for (int k = 0; k < 6; ++k) { a.top[k] = tg.tgt[bot_t[i]][k]; a.bot[k] = tg.tgt[bot_t[i + 1]][k]; }
      a.scratch = h->d_scratch; a.info = h->d_info;
      const bool lds = (Nk <= 64) && !h->opt_force_generic;
      const size_t sm = lds_bytes(Nk, lds);
      const size_t units = S * M;
      if (lds) {
        HIPCHK(h, allow_lds(k_combine<true>, sm));
        hipLaunchKernelGGL(k_combine<true>, dim3((unsigned)units), dim3(kThreads), sm, h->stream, a);
      } else {
        HIPCHK(h, allow_lds(k_combine<false>, sm));
        hipLaunchKernelGGL(k_combine<false>, dim3((unsigned)std::min<size_t>(units, (size_t)h->G)), dim3(kThreads), sm, h->stream, a);
      }
      HIPCHK(h, hipGetLastError());
    }
    for (int i = 0; i < nc; ++i) {
      const int ims = ord[i], L = sensor_levels[ims];
      double *bot[6], *top[6];
      for (int k = 0; k < 6; ++k) { bot[k] = tg.tgt[bot_t[i]][k]; top[k] = (L > 0) ? tg.tgt[snap_t[i]][k] : nullptr; }
      // surface interaction with the slab below the sensor (rt_run_multisensor.jl:150-159); L = 0: + post-processing of the
      // whole column (uwJ = R_SFI, dwJ = T_SFI, postprocessing_vza_ms.jl:34-36)
      if ((rc = rt_run_core(h, h->Nz, h->Nz, false, bot, true, L == 0))) return rc;
      if (L > 0) {
        InterArgs a{};
        a.q = h->qk; a.S = h->S; a.M = M;
        for (int k = 0; k < 6; ++k) { a.top[k] = top[k]; a.bot[k] = bot[k]; }
        a.dwJ = h->d_msJ[0]; a.uwJ = h->d_msJ[1]; a.scratch = h->d_scratch; a.info = h->d_info;
        const bool lds = (Nk <= 64) && !h->opt_force_generic;
        const size_t sm = lds_bytes(Nk, lds);
        const size_t units = S * M;
        if (lds) {
          HIPCHK(h, allow_lds(k_interlayer<true>, sm));
          hipLaunchKernelGGL(k_interlayer<true>, dim3((unsigned)units), dim3(kThreads), sm, h->stream, a);
        } else {
          HIPCHK(h, allow_lds(k_interlayer<false>, sm));
          hipLaunchKernelGGL(k_interlayer<false>, dim3((unsigned)std::min<size_t>(units, (size_t)h->G)), dim3(kThreads), sm, h->stream, a);
        }
        HIPCHK(h, hipGetLastError());
        PostArgs pa{};
        pa.N = Nk; pa.nS = h->nS; pa.S = h->S; pa.M = M; pa.nVza = h->nVza; pa.red0 = 0;
        pa.node = h->d_node; pa.cos_mphi = h->d_cos; pa.sin_mphi = h->d_sin;
        pa.J0p = h->d_msJ[0]; pa.J0m = h->d_msJ[1];
        pa.hdrJ = h->d_hdrJ; pa.hdr_all = 0; pa.zeroT_hi = 0; pa.hdrJm = nullptr;
        pa.R = h->d_R; pa.T = h->d_T; pa.hdr = h->d_hdr;
        hipLaunchKernelGGL(k_postprocess, dim3((unsigned)((out1 + 255) / 256)), dim3(256), 0, h->stream, pa);
        HIPCHK(h, hipGetLastError());
      }
      HIPCHK(h, hipMemcpyAsync(d_uw + out1 * ims, h->d_R, out1 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(d_dw + out1 * ims, h->d_T, out1 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
  }
  HIPCHK(h, hipMemcpyAsync(uwJ, d_uw, out1 * nSensors * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dwJ, d_dw, out1 * nSensors * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

// ---- ForwardDiff.Dual through rt_run (mom_dual.hip) ---------------------------------------------------------------
// The partials of everything mom_scene_set / mom_scene_set_surface uploaded, in the layout of the value arrays with the
// partial index as the slowest axis; NULL = that input does not depend on the parameters.
extern "C" int mom_scene_set_partials(mom_t *h, int P, const double *dtau, const double *dvarpi, const double *dzw,
                                      const double *dZpp, const double *dZmp, const double *dalbedo, const double *dRsurf,
                                      const double *dalbedo_spec) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_scene_set_partials");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_scene_set_partials: call mom_scene_set first");
  if (P < 0 || P > 64 || ((dZpp == nullptr) != (dZmp == nullptr)))
    return fail(h, MOM_EINVAL, "mom_scene_set_partials: 0 <= P <= 64; dZpp and dZmp come together");
  HIPCHK(h, hipSetDevice(h->device));
  for (int k = 0; k < 8; ++k)
    if (h->d_dual_in[k]) { (void)hipFree(h->d_dual_in[k]); h->d_dual_in[k] = nullptr; }
  if (h->d_dual_out) { (void)hipFree(h->d_dual_out); h->d_dual_out = nullptr; }
  if (h->d_dual_ts) { (void)hipFree(h->d_dual_ts); h->d_dual_ts = nullptr; }
  h->dual_P = P;
  h->dual_ran = false;
  if (P == 0) return MOM_OK;
  const size_t S = h->S, Nz = h->Nz, K = h->K, M = h->scene_M, N = h->N, Nk = h->Nk;
  int rc;
  if (dtau && (rc = upload_new(h, &h->d_dual_in[0], dtau, S * Nz * P))) return rc;
  if (dvarpi && (rc = upload_new(h, &h->d_dual_in[1], dvarpi, S * Nz * P))) return rc;
  if (dzw && (rc = upload_new(h, &h->d_dual_in[2], dzw, K * S * Nz * P))) return rc;
  if (dZpp) {
    if (Nk == N) {
      if ((rc = upload_new(h, &h->d_dual_in[3], dZpp, N * N * K * M * P))) return rc;
      if ((rc = upload_new(h, &h->d_dual_in[4], dZmp, N * N * K * M * P))) return rc;
    } else {  // the scene's operators carry strip_pad's dummy entries (Z = 0): so do the partials
      const std::vector<double> zp = pad_blocks(dZpp, (int)N, (int)Nk, K * M * P), zm = pad_blocks(dZmp, (int)N, (int)Nk, K * M * P);
      if ((rc = upload_new(h, &h->d_dual_in[3], zp.data(), zp.size()))) return rc;
      if ((rc = upload_new(h, &h->d_dual_in[4], zm.data(), zm.size()))) return rc;
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  if (dalbedo && (rc = upload_new(h, &h->d_dual_in[5], dalbedo, (size_t)P))) return rc;
  if (dRsurf && h->surf_kind == 1) {
    const std::vector<double> rp = pad_blocks(dRsurf, (int)N, (int)Nk, M * P);
    if ((rc = upload_new(h, &h->d_dual_in[6], rp.data(), rp.size()))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (dalbedo_spec && h->surf_kind == 2 && (rc = upload_new(h, &h->d_dual_in[7], dalbedo_spec, S * P))) return rc;
  HIPCHK(h, dmalloc(&h->d_dual_out, (3 * (size_t)h->nVza + 2) * h->nS * S * P));   // dR | dT | dhdr | dbhr_uw | dbhr_dw
  HIPCHK(h, dmalloc(&h->d_dual_ts, S * (Nz + 1) * P));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

// rt_run on Dual numbers for the resident scene: R_SFI / T_SFI (read with mom_get_RT) and their partials
// (mom_get_RT_partials).  Asynchronous on the handle's stream like mom_rt_run.
extern "C" int mom_rt_run_dual(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_rt_run_dual");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_rt_run_dual: call mom_scene_set first");
  HIPCHK(h, hipSetDevice(h->device));
  MomDualScene sc{};
  sc.N = h->Nk; sc.nS = h->nS; sc.S = h->S; sc.Nz = h->Nz; sc.K = h->K; sc.M = h->scene_M; sc.P = h->dual_P; sc.nVza = h->nVza;
  sc.imu0 = h->q.imu0; sc.strict = h->strict; sc.surf_kind = h->surf_kind; sc.mu0 = h->q.mu0; sc.albedo = h->albedo;
  for (int k = 0; k < 4; ++k) { sc.I0[k] = h->q.I0[k]; sc.D[k] = h->q.D[k]; }
  sc.mu = h->d_mu; sc.wt = h->d_wt;
  sc.tau = h->d_tau; sc.varpi = h->d_varpi; sc.zw = h->d_zw; sc.Zpp = h->d_Zpp; sc.Zmp = h->d_Zmp; sc.tau_sum = h->d_tau_sum;
  sc.dtau = h->d_dual_in[0]; sc.dvarpi = h->d_dual_in[1]; sc.dzw = h->d_dual_in[2]; sc.dZpp = h->d_dual_in[3];
  sc.dZmp = h->d_dual_in[4]; sc.dalbedo = h->d_dual_in[5]; sc.dRsurf = h->d_dual_in[6]; sc.dalbedo_spec = h->d_dual_in[7];
  sc.Rsurf = h->d_Rsurf; sc.albedo_spec = h->d_albedo_spec;
  sc.nd = h->nd.data(); sc.iface = h->iface.data(); sc.node = h->d_node; sc.cos_mphi = h->d_cos; sc.sin_mphi = h->d_sin;
  const size_t out = (size_t)h->nVza * h->nS * h->S;
  sc.R = h->d_R; sc.T = h->d_T; sc.dR = h->d_dual_out; sc.dT = h->d_dual_out ? h->d_dual_out + out * h->dual_P : nullptr;
  sc.hdr = h->d_hdr; sc.bhr_uw = h->d_bhr_uw; sc.bhr_dw = h->d_bhr_dw;
  sc.dhdr = h->d_dual_out ? h->d_dual_out + 2 * out * h->dual_P : nullptr;
  sc.dbhr_uw = h->d_dual_out ? h->d_dual_out + 3 * out * h->dual_P : nullptr;
  sc.dbhr_dw = sc.dbhr_uw ? sc.dbhr_uw + (size_t)h->nS * h->S * h->dual_P : nullptr;
  sc.dtau_sum_buf = h->d_dual_ts; sc.info = h->d_info; sc.stream = h->stream;
  sc.work = &h->dual_work; sc.work_cap = &h->dual_work_cap;
  size_t budget = h->opt_dual_budget;
  if (!budget) {
    size_t fr = 0, tot = 0;
    HIPCHK(h, hipMemGetInfo(&fr, &tot));
    budget = (size_t)(0.6 * (double)(fr + h->dual_work_cap));
  }
  sc.work_budget = budget;
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  std::string err;
  const int rc = momd_run(sc, &err);
  if (rc == 1) return fail(h, MOM_EUNSUPPORTED, err.c_str());
  if (rc) return fail(h, MOM_EHIP, err.c_str());
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  h->dual_ran = true;
  h->dual_last = true;
  h->comp_on_chip = true;  // no composite layer of this run is left in the handle's operator-level state
  return MOM_OK;
}

extern "C" int mom_get_RT_partials(mom_t *h, double *dR_SFI, double *dT_SFI) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->dual_ran || h->dual_P == 0 || !dR_SFI || !dT_SFI)
    return fail(h, MOM_ESTATE, "mom_get_RT_partials: no Dual run with P > 0 / null output");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t bytes = (size_t)h->nVza * h->nS * h->S * h->dual_P * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(dR_SFI, h->d_dual_out, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dT_SFI, h->d_dual_out + bytes / sizeof(double), bytes, hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

extern "C" int mom_get_hdr_partials(mom_t *h, double *dhdr, double *dbhr_uw, double *dbhr_dw) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->dual_ran || h->dual_P == 0 || !dhdr || !dbhr_uw || !dbhr_dw)
    return fail(h, MOM_ESTATE, "mom_get_hdr_partials: no Dual run with P > 0 / null output");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t out = (size_t)h->nVza * h->nS * h->S * h->dual_P, fl = (size_t)h->nS * h->S * h->dual_P;
  HIPCHK(h, hipMemcpyAsync(dhdr, h->d_dual_out + 2 * out, out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dbhr_uw, h->d_dual_out + 3 * out, fl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(dbhr_dw, h->d_dual_out + 3 * out + fl, fl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

extern "C" int mom_get_RT(mom_t *h, double *R_SFI, double *T_SFI) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->scene_set || !R_SFI || !T_SFI) return fail(h, MOM_ESTATE, "mom_get_RT: no scene / null output");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->f32) {
    const int rc = momf_get_RT(h->f32, R_SFI, T_SFI);
    return rc ? fail(h, rc, momf_error(h->f32)) : check_info(h);
  }
  const size_t bytes = (size_t)h->nVza * h->nS * h->S * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(R_SFI, h->d_R, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(T_SFI, h->d_T, bytes, hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

extern "C" int mom_get_hdr(mom_t *h, double *hdr, double *bhr_uw, double *bhr_dw) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->scene_set || !hdr || !bhr_uw || !bhr_dw) return fail(h, MOM_ESTATE, "mom_get_hdr: no scene / null output");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->f32) {
    const int rc = momf_get_hdr(h->f32, hdr, bhr_uw, bhr_dw);
    return rc ? fail(h, rc, momf_error(h->f32)) : check_info(h);
  }
  HIPCHK(h, hipMemcpyAsync(hdr, h->d_hdr, (size_t)h->nVza * h->nS * h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(bhr_uw, h->d_bhr_uw, (size_t)h->nS * h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(bhr_dw, h->d_bhr_dw, (size_t)h->nS * h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return check_info(h);
}

extern "C" int mom_get_RT_device(mom_t *h, void *dR, void *dT) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_get_RT_device");
  if (!h->scene_set || !dR || !dT) return fail(h, MOM_ESTATE, "mom_get_RT_device: no scene / null output");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t bytes = (size_t)h->nVza * h->nS * h->S * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(dR, h->d_R, bytes, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dT, h->d_T, bytes, hipMemcpyDeviceToDevice, h->stream));
  return MOM_OK;  // asynchronous: a singular-operator report surfaces at mom_get_RT / mom_check
}

// ---------------------------------------------------------------- operator-level post-processing

// cosd / sind as Julia evaluates them (base/special/trig.jl): reduction in degrees, double-double radians
// (deg2rad_ext), fdlibm kernels with the low word -- exact at the multiples of 30 and 90 degrees
#pragma clang fp contract(off)
static void mom_deg2rad_ext(double x, double &hi, double &lo) {
  const double m = 0.017453292519943295, m_hi = 0.01745329238474369, m_lo = 1.3519960527851425e-10;
  const volatile double u = 134217729.0 * x;
  const volatile double t = u - x;
  const double x_hi = u - t, x_lo = x - x_hi;
  hi = m * x;
  lo = x_hi * m_lo + (x_lo * m_hi + ((x_hi * m_hi - hi) + x_lo * m_lo));
}
static double mom_ksin(double deg) {
  double x, y;
  mom_deg2rad_ext(deg, x, y);
  const double z = x * x, v = z * x;
  const double r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01);
}
static double mom_kcos(double deg) {
  double x, y;
  mom_deg2rad_ext(deg, x, y);
  const double z = x * x;
  double w = z * z;
  const double r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                   w * w * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
  const double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}
static double mom_cosd(double x) {
  const double rx = std::fabs(std::fmod(x, 360.0));
  if (rx <= 45.0) return mom_kcos(rx);
  if (rx < 135.0) return mom_ksin(90.0 - rx);
  if (rx <= 225.0) return -mom_kcos(180.0 - rx);
  if (rx < 315.0) return mom_ksin(rx - 270.0);
  return mom_kcos(360.0 - rx);
}
static double mom_sind(double x) {
  const double rx = std::fmod(x, 360.0), arx = std::fabs(rx);
  if (rx == 0.0) return rx;
  if (arx < 45.0) return mom_ksin(rx);
  if (arx <= 135.0) return std::copysign(mom_kcos(90.0 - arx), rx);
  if (arx == 180.0) return std::copysign(0.0, rx);
  if (arx < 225.0) return mom_ksin((180.0 - arx) * std::copysign(1.0, rx));
  if (arx <= 315.0) return -std::copysign(mom_kcos(270.0 - arx), rx);
  return mom_ksin(rx - std::copysign(360.0, rx));
}

// postprocessing_vza!(RS_type::noRS, iμ₀, pol_type, composite_layer, vza, qp_μ, m, vaz, μ₀, weight, nSpec, SFI, R, R_SFI,
// T, T_SFI, ...) -- postprocessing_vza.jl:9-60 for ONE Fourier moment on the operator-level composite layer: gathers
// the nVza view rows of J0-/J0+ on the GPU (the reference copies the whole composite layer to the host, :17-20)
__global__ void k_op_postprocess(int N, int nS, int S, int nVza, const int *node, const double *cs_k, const double *J0p,
                                 const double *J0m, double *outR, double *outT) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nVza * nS * S;
  if (idx >= total) return;
  const int v = (int)(idx % nVza);
  const int k = (int)((idx / nVza) % nS);
  const size_t s = idx / ((size_t)nVza * nS);
  const size_t o = (size_t)(node[v] - 1) * nS + k + (size_t)N * s;
  const double cs = cs_k[v + (size_t)nVza * k];
  outR[idx] = cs * J0m[o];
  outT[idx] = cs * J0p[o];
}

extern "C" int mom_postprocess(mom_t *h, int m, int nVza, const int *node_1based, const double *vaz_deg, double weight,
                               double *R_SFI, double *T_SFI) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_postprocess");
  if (m < 0 || nVza <= 0 || !node_1based || !vaz_deg || !R_SFI || !T_SFI)
    return fail(h, MOM_EINVAL, "mom_postprocess: bad argument");
  for (int v = 0; v < nVza; ++v)
    if (node_1based[v] < 1 || node_1based[v] * h->nS > h->N) return fail(h, MOM_EINVAL, "mom_postprocess: bad view node");
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc_ = op_composite_ready(h, "mom_postprocess"); if (rc_) return rc_; }
  const size_t total = (size_t)nVza * h->nS * h->S;
  if (total > h->post_cap) {
    if (h->d_post[0]) (void)hipFree(h->d_post[0]);
    h->d_post[0] = nullptr; h->post_cap = 0;
    HIPCHK(h, dmalloc(&h->d_post[0], 2 * total));
    h->post_cap = total;
  }
  h->d_post[1] = h->d_post[0] + total;
  // bigCS = weight * Diagonal([cos(m φ), cos(m φ), sin(m φ), sin(m φ)][1:n])   (postprocessing_vza.jl:32-33)
  auto cosd = [](double x) { return mom_cosd(x); };
  auto sind = [](double x) { return mom_sind(x); };
  std::vector<double> cs((size_t)nVza * h->nS);
  for (int k = 0; k < h->nS; ++k)
    for (int v = 0; v < nVza; ++v) cs[v + (size_t)nVza * k] = weight * ((k < 2) ? cosd(m * vaz_deg[v]) : sind(m * vaz_deg[v]));
  int *d_node = nullptr;
  double *d_cs = nullptr;
  HIPCHK(h, ws_get(h, 2, &d_node, (size_t)nVza));   // grow-only workspace of the handle: no allocation per call, nothing to leak
  HIPCHK(h, ws_get(h, 3, &d_cs, cs.size()));
  HIPCHK(h, hipMemcpyAsync(d_node, node_1based, (size_t)nVza * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_op_postprocess, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->N, h->nS, h->S, nVza,
                     d_node, d_cs, h->comp[4], h->comp[5], h->d_post[0], h->d_post[1]);
  HIPCHK(h, hipGetLastError());
  std::vector<double> hr(total), ht(total);
  HIPCHK(h, hipMemcpyAsync(hr.data(), h->d_post[0], total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(ht.data(), h->d_post[1], total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < total; ++i) { R_SFI[i] += hr[i]; T_SFI[i] += ht[i]; }  // `+=` like :48-49
  return MOM_OK;
}

// ---------------------------------------------------------------- multi-GPU: RCCL behind the C ABI

namespace {
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
// librccl.so.1 is loaded on first use (the soname a host process such as PyTorch-ROCm may already have mapped: one
// copy per process); libmomcore.so itself stays loadable on machines without RCCL
int rccl_load(mom_t *h) {
  if (g_rccl.lib) return MOM_OK;
  // RCCL must sit on the SAME HIP runtime instance as this library.  A host process may hold two (PyTorch-ROCm wheels
  // bundle libamdhip64.so + librccl.so next to /opt/rocm's, and which one libmomcore.so was bound to depends on the
  // import order), so the copy next to the runtime that resolves OUR hip* symbols is taken first
  void *lib = nullptr;
  Dl_info info;
  if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash);
      for (const char *name : {"/librccl.so.1", "/librccl.so"}) {
        lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
      }
    }
  }
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(h, MOM_EHIP, "mom_comm: cannot load librccl.so.1 (RCCL)");
  Rccl r;
  r.lib = lib;
  *(void **)(&r.GetUniqueId) = dlsym(lib, "ncclGetUniqueId");
  *(void **)(&r.CommInitRank) = dlsym(lib, "ncclCommInitRank");
  *(void **)(&r.CommDestroy) = dlsym(lib, "ncclCommDestroy");
  *(void **)(&r.AllGather) = dlsym(lib, "ncclAllGather");
  *(void **)(&r.AllReduce) = dlsym(lib, "ncclAllReduce");
  *(void **)(&r.GetErrorString) = dlsym(lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString)
    return fail(h, MOM_EHIP, "mom_comm: librccl.so.1 lacks a required symbol");
  g_rccl = r;
  g_rccl_destroy = [](void *c) { (void)g_rccl.CommDestroy((ncclComm_t)c); };
  return MOM_OK;
}
int rccl_fail(mom_t *h, const char *what, ncclResult_t r) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return fail(h, MOM_EHIP, buf);
}
}  // namespace

extern "C" int mom_comm_unique_id(void *id_out, size_t bytes) {
  if (!id_out || bytes < sizeof(ncclUniqueId)) return fail(nullptr, MOM_EINVAL, "mom_comm_unique_id: need MOM_COMM_ID_BYTES bytes");
  const int rc = rccl_load(nullptr);
  if (rc) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", r);
  memcpy(id_out, &id, sizeof id);
  return MOM_OK;
}

extern "C" int mom_comm_init(mom_t *h, int rank, int nranks, const void *nccl_id) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (nranks < 1 || rank < 0 || rank >= nranks || !nccl_id) return fail(h, MOM_EINVAL, "mom_comm_init: bad argument");
  if (h->comm) return fail(h, MOM_ESTATE, "mom_comm_init: communicator already initialised");
  HIPCHK(h, hipSetDevice(h->device));
  const int rc = rccl_load(h);
  if (rc) return rc;
  // RCCL checks hipGetLastError() after its own launches: a stale (non-sticky) error code left behind by an earlier,
  // already reported failure in this process would be taken for its own
  (void)hipGetLastError();
  ncclUniqueId id;
  memcpy(&id, nccl_id, sizeof id);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) return rccl_fail(h, "ncclCommInitRank", r);
  h->comm = comm; h->comm_rank = rank; h->comm_size = nranks;
  return MOM_OK;
}

extern "C" int mom_comm_destroy(mom_t *h) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)g_rccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr; h->comm_size = 1; h->comm_rank = 0;
  }
  return MOM_OK;
}

extern "C" int mom_allgather(mom_t *h, const void *d_local, void *d_global, size_t count) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->comm) return fail(h, MOM_ESTATE, "mom_allgather: call mom_comm_init first");
  if (!d_local || !d_global) return fail(h, MOM_EINVAL, "mom_allgather: null buffer");
  HIPCHK(h, hipSetDevice(h->device));
  const ncclResult_t r = g_rccl.AllGather(d_local, d_global, count, ncclDouble, (ncclComm_t)h->comm, h->stream);
  if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather", r);
  return MOM_OK;
}

// The ONE collective of a sharded run: every rank contributes its R_SFI || T_SFI block (already contiguous in the
// handle, 2 * nVza * nStokes * S_loc doubles) and receives [nranks][2][nVza*nStokes*S_loc]
extern "C" int mom_allgather_RT_device(mom_t *h, void *d_global) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_allgather_RT_device");
  if (!h->scene_set || !d_global) return fail(h, MOM_ESTATE, "mom_allgather_RT_device: no scene / null output");
  return mom_allgather(h, h->d_R, d_global, 2 * (size_t)h->nVza * h->nS * h->S);
}

extern "C" int mom_allgather_RT(mom_t *h, double *R_SFI_global, double *T_SFI_global) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_allgather_RT");
  if (!h->scene_set || !R_SFI_global || !T_SFI_global) return fail(h, MOM_ESTATE, "mom_allgather_RT: no scene / null output");
  if (!h->comm) return fail(h, MOM_ESTATE, "mom_allgather_RT: call mom_comm_init first");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t nout = (size_t)h->nVza * h->nS * h->S, need = 2 * nout * h->comm_size;
  if (need > h->gather_cap) {
    if (h->d_gather) (void)hipFree(h->d_gather);
    h->d_gather = nullptr; h->gather_cap = 0;
    HIPCHK(h, dmalloc(&h->d_gather, need));
    h->gather_cap = need;
  }
  int rc = mom_allgather_RT_device(h, h->d_gather);
  if (rc) return rc;
  // [rank][R|T][nVza, nStokes, S_loc] -> R_SFI, T_SFI [nVza, nStokes, nranks * S_loc] (rank-major spectral axis)
  for (int r = 0; r < h->comm_size; ++r) {
    HIPCHK(h, hipMemcpyAsync(R_SFI_global + nout * r, h->d_gather + 2 * nout * r, nout * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(T_SFI_global + nout * r, h->d_gather + 2 * nout * r + nout, nout * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  return check_info(h);
}

// ---------------------------------------------------------------- device-side layer optics

extern "C" int mom_absorption_begin(mom_t *h, int Nz, const double *grid) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (Nz <= 0) return fail(h, MOM_EINVAL, "mom_absorption_begin: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t S = h->S;
  if (h->d_tau_abs) { (void)hipFree(h->d_tau_abs); h->d_tau_abs = nullptr; }
  HIPCHK(h, dmalloc(&h->d_tau_abs, S * Nz));
  HIPCHK(h, hipMemsetAsync(h->d_tau_abs, 0, S * Nz * sizeof(double), h->stream));  // τ_abs = zeros (model_from_parameters.jl:48)
  h->abs_Nz = Nz;
  if (grid) {
    int rc = upload_new(h, &h->d_grid, grid, S);
    if (rc) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_absorption_set(mom_t *h, int Nz, const double *tau_abs) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (Nz <= 0 || !tau_abs) return fail(h, MOM_EINVAL, "mom_absorption_set: bad argument");
  int rc = mom_absorption_begin(h, Nz, nullptr);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->d_tau_abs, tau_abs, (size_t)h->S * Nz * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_absorption_get(mom_t *h, double *tau_abs) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->d_tau_abs || !tau_abs) return fail(h, MOM_ESTATE, "mom_absorption_get: no resident tau_abs table / null output");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(tau_abs, h->d_tau_abs, (size_t)h->S * h->abs_Nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_voigt_tau_abs(mom_t *h, int iz_1based, int nLines, const double *nu, const double *gamma_d, const double *y,
                                 const double *S, const int *ind_start_1based, const int *ind_stop_1based, double factor) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->d_tau_abs || !h->d_grid) return fail(h, MOM_ESTATE, "mom_voigt_tau_abs: call mom_absorption_begin with the spectral grid first");
  if (iz_1based < 1 || iz_1based > h->abs_Nz || nLines < 0 ||
      (nLines > 0 && (!nu || !gamma_d || !y || !S || !ind_start_1based || !ind_stop_1based)))
    return fail(h, MOM_EINVAL, "mom_voigt_tau_abs: bad argument");
  for (int j = 0; j < nLines; ++j)
    if (ind_start_1based[j] < 1 || ind_stop_1based[j] > h->S) {
      char buf[160];
      snprintf(buf, sizeof buf, "mom_voigt_tau_abs: line %d: window [%d, %d] outside the grid 1..%d", j + 1, ind_start_1based[j],
               ind_stop_1based[j], h->S);
      return fail(h, MOM_EINVAL, buf);
    }
  if (nLines == 0) return MOM_OK;
  int sorted = 1;
  for (int j = 1; j < nLines; ++j)
    if (ind_start_1based[j] < ind_start_1based[j - 1] || ind_stop_1based[j] < ind_stop_1based[j - 1]) { sorted = 0; break; }
  HIPCHK(h, hipSetDevice(h->device));
  const size_t lb = (size_t)nLines;
  if (lb > h->lines_cap || h->lines_nz != 1) {  // 4 double + 2 int arrays per line, grown geometrically: no allocation in steady state
    h->lines_nz = 1;
    if (h->d_lines) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_lines); h->d_lines = nullptr; h->lines_cap = 0; }
    const size_t cap = std::max<size_t>(lb, 1024) * 2;
    HIPCHK(h, dmalloc(&h->d_lines, 5 * cap));
    h->lines_cap = cap;
  }
  const size_t cap = h->lines_cap;
  double *dl = h->d_lines;
  int *dw = reinterpret_cast<int *>(dl + 4 * cap);
  const double *src[4] = {nu, gamma_d, y, S};
  // the host arrays are borrowed for the call only: the copies must have left them before we return
  for (int k = 0; k < 4; ++k) HIPCHK(h, hipMemcpyAsync(dl + k * cap, src[k], lb * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dw, ind_start_1based, lb * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dw + cap, ind_stop_1based, lb * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, mom_voigt_launch(h->stream, nLines, dl, dl + cap, dl + 2 * cap, dl + 3 * cap, dw, dw + cap, h->S, h->d_grid,
                             h->d_tau_abs + (size_t)h->S * (iz_1based - 1), factor, 1, sorted));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

// Resident line table of one absorber: the HITRAN columns of the lines inside the padded grid (the host selects them once,
// compute_absorption_cross_section.jl:54-72) and the TIPS-2017 spline tables of their isotopologues (qoft! :197-214: knots,
// values and the second derivatives of DataInterpolations.CubicSpline, computed once by the host in the tables' Float32).
extern "C" int mom_absorption_set_lines(mom_t *h, int nLines, const double *nu0, const double *S0, const double *gamma_air,
                                        const double *gamma_self, const double *E_lower, const double *n_air,
                                        const double *delta_air, const double *sqrt_mol_weight, const int *iso_index, int nIso,
                                        int nTmax, const int *nT, const double *tips_T, const double *tips_Q, const double *tips_z) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (nLines < 0 || nIso < 0 || nTmax < 0 || (nLines > 0 && (!nu0 || !S0 || !gamma_air || !gamma_self || !E_lower || !n_air ||
      !delta_air || !sqrt_mol_weight || !iso_index)) || (nIso > 0 && (nTmax < 2 || !nT || !tips_T || !tips_Q || !tips_z)))
    return fail(h, MOM_EINVAL, "mom_absorption_set_lines: bad argument");
  for (int j = 0; j < nLines; ++j)
    if (E_lower[j] != -1.0 && (iso_index[j] < 0 || iso_index[j] >= nIso))
      return fail(h, MOM_EINVAL, "mom_absorption_set_lines: iso_index out of range");
  for (int k = 0; k < nIso; ++k)
    if (nT[k] < 2 || nT[k] > nTmax) return fail(h, MOM_EINVAL, "mom_absorption_set_lines: bad knot count");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  (void)hipFree(h->d_lt); (void)hipFree(h->d_lt_i);
  h->d_lt = nullptr; h->d_lt_i = nullptr;
  h->lt = MomLineTable{};
  const size_t L = (size_t)std::max(nLines, 1), Tn = (size_t)std::max(nIso, 1) * std::max(nTmax, 1);
  HIPCHK(h, dmalloc(&h->d_lt, 8 * L + 3 * Tn));
  HIPCHK(h, dmalloc(&h->d_lt_i, L + std::max(nIso, 1) + 1));
  const double *cols[8] = {nu0, S0, gamma_air, gamma_self, E_lower, n_air, delta_air, sqrt_mol_weight};
  for (int k = 0; k < 8 && nLines > 0; ++k) HIPCHK(h, hipMemcpy(h->d_lt + k * L, cols[k], (size_t)nLines * sizeof(double), hipMemcpyHostToDevice));
  const double *tabs[3] = {tips_T, tips_Q, tips_z};
  for (int k = 0; k < 3 && nIso > 0; ++k) HIPCHK(h, hipMemcpy(h->d_lt + 8 * L + k * Tn, tabs[k], (size_t)nIso * nTmax * sizeof(double), hipMemcpyHostToDevice));
  if (nLines > 0) HIPCHK(h, hipMemcpy(h->d_lt_i, iso_index, (size_t)nLines * sizeof(int), hipMemcpyHostToDevice));
  if (nIso > 0) HIPCHK(h, hipMemcpy(h->d_lt_i + L, nT, (size_t)nIso * sizeof(int), hipMemcpyHostToDevice));
  MomLineTable &t = h->lt;
  t.nLines = nLines; t.nIso = nIso; t.nTmax = nTmax;
  t.nu0 = h->d_lt; t.S0 = h->d_lt + L; t.g_air = h->d_lt + 2 * L; t.g_self = h->d_lt + 3 * L; t.E = h->d_lt + 4 * L;
  t.n_air = h->d_lt + 5 * L; t.d_air = h->d_lt + 6 * L; t.sqw = h->d_lt + 7 * L;
  t.tT = h->d_lt + 8 * L; t.tQ = t.tT + Tn; t.tZ = t.tQ + Tn;
  t.iso = h->d_lt_i; t.nT = h->d_lt_i + L;
  // the common validity range of the TIPS tables in use (qoft! asserts Tmin < T < Tmax, :204)
  h->lt_Tmin = -1e300; h->lt_Tmax = 1e300;
  for (int k = 0; k < nIso; ++k) {
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < nT[k]; ++i) { lo = std::min(lo, tips_T[(size_t)k * nTmax + i]); hi = std::max(hi, tips_T[(size_t)k * nTmax + i]); }
    h->lt_Tmin = std::max(h->lt_Tmin, lo); h->lt_Tmax = std::min(h->lt_Tmax, hi);
  }
  return MOM_OK;
}

// compute_absorption_profile! for ONE layer (atmo_prof.jl:427-449) with the per-line prefactors formed ON THE DEVICE from the
// resident table: only (p, T, vmr, wing_cutoff, factor) cross the bus.
extern "C" int mom_voigt_tau_abs_layer(mom_t *h, int iz_1based, double pressure, double temperature, double vmr,
                                       double wing_cutoff, double factor) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->d_tau_abs || !h->d_grid) return fail(h, MOM_ESTATE, "mom_voigt_tau_abs_layer: call mom_absorption_begin with the spectral grid first");
  if (!h->d_lt) return fail(h, MOM_ESTATE, "mom_voigt_tau_abs_layer: call mom_absorption_set_lines first");
  if (iz_1based < 1 || iz_1based > h->abs_Nz || !(temperature > 0.0)) return fail(h, MOM_EINVAL, "mom_voigt_tau_abs_layer: bad argument");
  if (h->lt.nIso > 0 && !(h->lt_Tmin < temperature && temperature < h->lt_Tmax)) {
    char buf[160];
    snprintf(buf, sizeof buf, "TIPS2017: T (%g) must be between %g K and %g K.", temperature, h->lt_Tmin, h->lt_Tmax);
    return fail(h, MOM_EINVAL, buf);
  }
  const int nLines = h->lt.nLines;
  if (nLines == 0) return MOM_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t lb = (size_t)nLines;
  if (lb > h->lines_cap || h->lines_nz != 1) {
    h->lines_nz = 1;
    if (h->d_lines) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_lines); h->d_lines = nullptr; h->lines_cap = 0; }
    const size_t cap = std::max<size_t>(lb, 1024) * 2;
    HIPCHK(h, dmalloc(&h->d_lines, 5 * cap));
    h->lines_cap = cap;
  }
  const size_t cap = h->lines_cap;
  double *dl = h->d_lines;
  int *dw = reinterpret_cast<int *>(dl + 4 * cap);
  int *flag = h->d_lt_i + (size_t)std::max(nLines, 1) + std::max(h->lt.nIso, 1);
  HIPCHK(h, hipMemsetAsync(flag, 0, sizeof(int), h->stream));
  // γ_d = (cSqrt2Ln2 / cc_) sqrt(cBolts_ / cMassMol) sqrt(T) ν₀ / sqrt(mol_weight)   (:87-88): the scalar part once
  const double cgd = (1.1774100225 / 2.99792458e8) * std::sqrt(1.3806503e-23 / 1.66053873e-27) * std::sqrt(temperature);
  HIPCHK(h, mom_line_prefactors_launch(h->stream, h->lt, h->S, h->d_grid, pressure, temperature, vmr, wing_cutoff, cgd, dl, dl + cap,
                                       dl + 2 * cap, dl + 3 * cap, dw, dw + cap, flag));
  int unsorted = 0;
  HIPCHK(h, hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, mom_voigt_launch(h->stream, nLines, dl, dl + cap, dl + 2 * cap, dl + 3 * cap, dw, dw + cap, h->S, h->d_grid,
                             h->d_tau_abs + (size_t)h->S * (iz_1based - 1), factor, 1, unsorted ? 0 : 1));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

// compute_absorption_profile! for ALL layers of a profile (atmo_prof.jl:427-449) in two launches: the reference walks the
// layers on the host and, per layer, launches one line-shape kernel per line; at its operating point (O2 A-band at
// 0.015 cm^-1, wing cut-off 40 cm^-1, 40 layers) a per-layer launch covers 90 workgroups -- a third of the GPU -- and the
// host round trips between the layers cost more than the arithmetic.  Here blockIdx.y = layer.  gpu_ms (optional): HIP-event
// time of the two kernels.
extern "C" int mom_voigt_tau_abs_profile(mom_t *h, int Nz, const double *pressure, const double *temperature, double vmr,
                                         double wing_cutoff, const double *factor, double *gpu_ms) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->d_tau_abs || !h->d_grid) return fail(h, MOM_ESTATE, "mom_voigt_tau_abs_profile: call mom_absorption_begin with the spectral grid first");
  if (!h->d_lt) return fail(h, MOM_ESTATE, "mom_voigt_tau_abs_profile: call mom_absorption_set_lines first");
  if (Nz < 1 || Nz > h->abs_Nz || !pressure || !temperature || !factor) return fail(h, MOM_EINVAL, "mom_voigt_tau_abs_profile: bad argument");
  for (int z = 0; z < Nz; ++z) {
    if (!(temperature[z] > 0.0)) return fail(h, MOM_EINVAL, "mom_voigt_tau_abs_profile: bad argument");
    if (h->lt.nIso > 0 && !(h->lt_Tmin < temperature[z] && temperature[z] < h->lt_Tmax)) {
      char buf[160];
      snprintf(buf, sizeof buf, "TIPS2017: T (%g) must be between %g K and %g K.", temperature[z], h->lt_Tmin, h->lt_Tmax);
      return fail(h, MOM_EINVAL, buf);
    }
  }
  if (gpu_ms) *gpu_ms = 0.0;
  const int nLines = h->lt.nLines;
  if (nLines == 0) return MOM_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t per = std::max<size_t>((size_t)nLines, 1024) * 2;       // line capacity of one layer's block
  const size_t need = per * (size_t)Nz;
  if (need > h->lines_cap * (size_t)std::max(h->lines_nz, 1) || h->lines_nz != Nz) {
    if (h->d_lines) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_lines); h->d_lines = nullptr; h->lines_cap = 0; }
    HIPCHK(h, dmalloc(&h->d_lines, 5 * need));
    h->lines_cap = per;
    h->lines_nz = Nz;
  }
  const size_t cap = h->lines_cap;
  // per-layer scalars [p | T | cgd | factor][Nz] and the Nz sortedness flags
  const size_t prm_doubles = 4 * (size_t)Nz + ((size_t)Nz + 1) / 2;
  if (prm_doubles > h->prof_cap) {
    if (h->d_prof) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_prof); h->d_prof = nullptr; h->prof_cap = 0; }
    HIPCHK(h, dmalloc(&h->d_prof, prm_doubles));
    h->prof_cap = prm_doubles;
  }
  std::vector<double> prm(4 * (size_t)Nz);
  for (int z = 0; z < Nz; ++z) {
    prm[z] = pressure[z];
    prm[Nz + z] = temperature[z];
    // γ_d = (cSqrt2Ln2 / cc_) sqrt(cBolts_ / cMassMol) sqrt(T) ν₀ / sqrt(mol_weight)   (:87-88): the scalar part
    prm[2 * (size_t)Nz + z] = (1.1774100225 / 2.99792458e8) * std::sqrt(1.3806503e-23 / 1.66053873e-27) * std::sqrt(temperature[z]);
    prm[3 * (size_t)Nz + z] = factor[z];
  }
  HIPCHK(h, hipMemcpyAsync(h->d_prof, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int *flags = reinterpret_cast<int *>(h->d_prof + 4 * (size_t)Nz);
  HIPCHK(h, hipMemsetAsync(flags, 0, sizeof(int) * (size_t)Nz, h->stream));
  double *pf = h->d_lines;
  int *win = reinterpret_cast<int *>(pf + 4 * cap * (size_t)Nz);
  if (gpu_ms) {
    for (int k = 0; k < 2; ++k)
      if (!h->ev_voigt[k]) HIPCHK(h, hipEventCreate(&h->ev_voigt[k]));
    HIPCHK(h, hipEventRecord(h->ev_voigt[0], h->stream));
  }
  HIPCHK(h, mom_voigt_profile_launch(h->stream, h->lt, Nz, cap, h->S, h->d_grid, h->d_prof, vmr, wing_cutoff, pf, win, flags,
                                     h->d_tau_abs, h->d_prof + 3 * (size_t)Nz));
  if (gpu_ms) HIPCHK(h, hipEventRecord(h->ev_voigt[1], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // prm is a host temporary
  if (gpu_ms) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_voigt[0], h->ev_voigt[1]) == hipSuccess) *gpu_ms = ms;
  }
  return MOM_OK;
}

// the prefactors of the last mom_voigt_tau_abs / mom_voigt_tau_abs_layer call (test access); n = its number of lines
extern "C" int mom_absorption_get_prefactors(mom_t *h, int n, double *nu, double *gamma_d, double *y, double *S, int *ind_start_1based,
                                             int *ind_stop_1based) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->d_lines || n < 0 || (size_t)n > h->lines_cap) return fail(h, MOM_ESTATE, "mom_absorption_get_prefactors: no prefactors resident");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t cap = h->lines_cap, nz = (size_t)std::max(h->lines_nz, 1), last = (nz - 1) * cap;  // the LAST layer of a profile call
  double *dst[4] = {nu, gamma_d, y, S};
  for (int k = 0; k < 4; ++k)
    if (dst[k]) HIPCHK(h, hipMemcpy(dst[k], h->d_lines + k * nz * cap + last, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  const int *dw = reinterpret_cast<const int *>(h->d_lines + 4 * nz * cap);
  if (ind_start_1based) HIPCHK(h, hipMemcpy(ind_start_1based, dw + last, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  if (ind_stop_1based) HIPCHK(h, hipMemcpy(ind_stop_1based, dw + nz * cap + last, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  return MOM_OK;
}

// constructCoreOpticalProperties (compEffectiveLayerProperties.jl:1-78) with the `+` of types.jl:632-678, createAero
// (:80-85), the gas term (:672-678) and the cumulative τ_sum of extractEffectiveProps (:108), one thread per spectral
// point walking the layers; per-layer max(τ ϖ) for get_dtau_ndoubl / `scatter` by atomic max on the bit pattern
// (non-negative doubles order like their unsigned bit patterns).  Contraction off: the same IEEE operations as the
// host (numpy / Julia) path, so both routes give bitwise equal τ, ϖ, weights.
struct OpticsArgs {
  int S, Nz, nAer;
  double varpi_rayl;
  const double *tau_rayl, *tau_abs;  // [S,Nz]
  const double *aer;                 // [2,nAer,Nz]: τ_y, w_y = τ_y ϖ_y per aerosol type and layer (spectrally flat)
  const int *aer_mode;               // [nAer,Nz]: 0 = Rayleigh side all zero, 1 = mix, 2 = aerosol side all zero
  double *tau, *varpi, *zw, *tau_sum, *layer_max;
};
#pragma clang fp contract(off)
__global__ void k_optics(OpticsArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = 1 + a.nAer;
  double tsum = 0.0;
  const bool live = n < a.S;
  if (live) a.tau_sum[n] = 0.0;
  for (int z = 0; z < a.Nz; ++z) {
    double tw = 0.0;
    if (live) {
      const size_t o = n + (size_t)a.S * z;
      double tau = a.tau_rayl[o], varpi = a.varpi_rayl;
      double w[8];
      w[0] = 1.0;
      for (int k = 1; k < K; ++k) w[k] = 0.0;
      for (int x = 0; x < a.nAer; ++x) {
        const double ty = a.aer[x + (size_t)a.nAer * z], wy = a.aer[a.nAer * a.Nz + x + (size_t)a.nAer * z];
        const int mode = a.aer_mode[x + (size_t)a.nAer * z];
        const double wx = tau * varpi, tot = wx + wy, tn = tau + ty;
        if (mode == 0) {
          for (int k = 0; k < K; ++k) w[k] = 0.0;
          w[x + 1] = 1.0;
        } else if (mode == 1) {
          const double fx = wx / tot;
          for (int k = 0; k < K; ++k) w[k] *= fx;
          w[x + 1] = wy / tot;
        }
        varpi = tot / tn;
        tau = tn;
      }
      const double tn = tau + a.tau_abs[o];
      varpi = (tau * varpi) / tn;
      tau = tn;
      a.tau[o] = tau;
      a.varpi[o] = varpi;
      for (int k = 0; k < K; ++k) a.zw[k + (size_t)K * o] = w[k];
      tsum = tsum + 1.0 * tau;
      a.tau_sum[o + a.S] = tsum;
      tw = tau * varpi;
    }
    // NaN (0/0 in an empty layer) must not win silently: fmax drops it like Julia's maximum would propagate it --
    // the host path would fail on such a scene as well; keep it visible as +inf
    if (tw != tw) tw = __longlong_as_double(0x7ff0000000000000ll);
    double m = tw;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.0)
      atomicMax(reinterpret_cast<unsigned long long *>(a.layer_max + z), (unsigned long long)__double_as_longlong(m));
  }
}

// doubling_number (rt_helper_functions.jl:31-57): log10 arithmetic and the eps test as in the reference
static int doubling_number_host(double dtau_max, double tau_end) {
  if (tau_end <= dtau_max) return 0;
  const double q1 = std::log10(2.0), q2 = std::log10(dtau_max), q3 = std::log10(tau_end);
  const double tlimit = (q3 - q2) / q1, nlimit = std::floor(tlimit);
  if (tlimit - nlimit < 2.220446049250313e-16) return (int)nlimit;
  return (int)nlimit + 1;
}

extern "C" int mom_scene_set_optics(mom_t *h, int Nz, int nAer, int M, const double *tau_rayl, double varpi_rayl,
                                    const double *tau_aer, const double *omega_aer, const double *ft_aer,
                                    const double *Zpp, const double *Zmp, double albedo, int nVza, const int *node_1based,
                                    const double *cos_mphi, const double *sin_mphi) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_scene_set_optics: call mom_set_streams first");
  if (Nz <= 0 || nAer < 0 || nAer > 7 || M <= 0 || M > h->M || nVza <= 0 || !tau_rayl || !Zpp || !Zmp || !node_1based ||
      !cos_mphi || !sin_mphi || (nAer > 0 && (!tau_aer || !omega_aer || !ft_aer)))
    return fail(h, MOM_EINVAL, "mom_scene_set_optics: bad argument");
  if (!h->d_tau_abs || h->abs_Nz != Nz)
    return fail(h, MOM_ESTATE, "mom_scene_set_optics: no resident tau_abs table of this Nz (mom_absorption_begin / _set)");
  HIPCHK(h, hipSetDevice(h->device));
  h->scene_set = false;
  const size_t S = h->S;
  const int K = 1 + nAer;
  int rc;
  if ((rc = upload_new(h, &h->d_tau_rayl, tau_rayl, S * Nz))) return rc;
  // createAero (compEffectiveLayerProperties.jl:80-85): τ' = (1 - fᵗ ω̃) τ_aer, ϖ' = (1 - fᵗ) ω̃ / (1 - fᵗ ω̃); the
  // all-zero tests of types.jl:641-661 are decided here on the host (they are properties of whole spectral columns)
  std::vector<double> aer((size_t)2 * std::max(nAer, 1) * Nz, 0.0);
  std::vector<int> mode((size_t)std::max(nAer, 1) * Nz, 2);
  for (int z = 0; z < Nz; ++z) {
    bool x_zero = true;  // all(τ ϖ == 0) of the accumulated left operand
    if (varpi_rayl != 0.0)
      for (size_t n = 0; n < S; ++n)
        if (tau_rayl[n + S * z] != 0.0) { x_zero = false; break; }
    for (int x = 0; x < nAer; ++x) {
      const double ty = (1 - ft_aer[x] * omega_aer[x]) * tau_aer[x + (size_t)nAer * z];
      const double vy = (1 - ft_aer[x]) * omega_aer[x] / (1 - ft_aer[x] * omega_aer[x]);
      const double wy = ty * vy;
      aer[x + (size_t)nAer * z] = ty;
      aer[(size_t)nAer * Nz + x + (size_t)nAer * z] = wy;
      mode[x + (size_t)nAer * z] = x_zero ? 0 : (wy != 0.0 ? 1 : 2);
      x_zero = x_zero && (wy == 0.0);
    }
  }
  if ((rc = upload_new(h, &h->d_aer, aer.data(), aer.size()))) return rc;
  if ((rc = upload_new(h, &h->d_aer_mode, mode.data(), mode.size()))) return rc;
  auto renew = [&](double **p, size_t cnt) -> hipError_t { if (*p) { (void)hipFree(*p); *p = nullptr; } return dmalloc(p, cnt); };
  HIPCHK(h, renew(&h->d_tau, S * Nz));
  HIPCHK(h, renew(&h->d_varpi, S * Nz));
  HIPCHK(h, renew(&h->d_zw, (size_t)K * S * Nz));
  HIPCHK(h, renew(&h->d_tau_sum, S * (Nz + 1)));
  HIPCHK(h, renew(&h->d_layer_max, (size_t)Nz));
  HIPCHK(h, hipMemsetAsync(h->d_layer_max, 0, (size_t)Nz * sizeof(double), h->stream));
  OpticsArgs a{};
  a.S = h->S; a.Nz = Nz; a.nAer = nAer; a.varpi_rayl = varpi_rayl;
  a.tau_rayl = h->d_tau_rayl; a.tau_abs = h->d_tau_abs; a.aer = h->d_aer; a.aer_mode = h->d_aer_mode;
  a.tau = h->d_tau; a.varpi = h->d_varpi; a.zw = h->d_zw; a.tau_sum = h->d_tau_sum; a.layer_max = h->d_layer_max;
  hipLaunchKernelGGL(k_optics, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  // get_dtau_ndoubl takes maximum(τ .* ϖ) over the WHOLE spectral axis (rt_kernel.jl:241-242): across the ranks of a
  // sharded run the per-layer maxima are combined first (one tiny all-reduce at set-up time, not in the sweep)
  if (h->comm) {
    const ncclResult_t r = g_rccl.AllReduce(h->d_layer_max, h->d_layer_max, (size_t)Nz, ncclDouble, ncclMax, (ncclComm_t)h->comm, h->stream);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllReduce", r);
  }
  std::vector<double> mx((size_t)Nz);
  HIPCHK(h, hipMemcpyAsync(mx.data(), h->d_layer_max, (size_t)Nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double mu_min = h->h_mu[0];
  for (double v : h->h_mu) mu_min = std::min(mu_min, v);
  h->nd.assign((size_t)Nz, 0);
  h->iface.assign((size_t)Nz, 0);
  int prev = 0;
  for (int z = 0; z < Nz; ++z) {
    if (!std::isfinite(mx[z])) return fail(h, MOM_EINVAL, "mom_scene_set_optics: a layer has non-finite τ ϖ (empty layer: τ = 0?)");
    h->nd[z] = doubling_number_host(std::min(mx[z], 0.001 * mu_min), mx[z]);
    if (h->nd[z] > 60) return fail(h, MOM_EINVAL, "mom_scene_set_optics: ndoubl out of range");
    const bool scatter = mx[z] > 2 * 2.220446049250313e-16;  // compEffectiveLayerProperties.jl:104
    prev = (z == 0) ? (scatter ? 3 : 0) : (prev == 0 ? (scatter ? 1 : 0) : (scatter ? 3 : 2));  // rt_helper_functions.jl:8-27
    h->iface[z] = prev;
  }
  if (h->f32) {  // Float32 handle: the Float64 assembly above is rounded to Float32 on the device (no host hop of tau_abs either)
    h->Nz = Nz; h->K = K; h->scene_M = M; h->nVza = nVza; h->albedo = albedo; h->surf_kind = 0;
    if ((rc = momf_scene_set_dev(h->f32, Nz, K, M, h->d_tau, h->d_varpi, h->d_zw, Zpp, Zmp, h->nd.data(), h->iface.data(), h->d_tau_sum,
                                 albedo, nVza, node_1based, cos_mphi, sin_mphi)))
      return fail(h, rc, momf_error(h->f32));
    h->scene_set = true;
    return MOM_OK;
  }
  if ((rc = scene_common(h, Nz, K, M, Zpp, Zmp, albedo, nVza, node_1based, cos_mphi, sin_mphi))) return rc;
  h->scene_set = true;
  return MOM_OK;
}

extern "C" int mom_scene_get_layers(mom_t *h, int *ndoubl, int *iface, double *tau, double *varpi, double *zw, double *tau_sum) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_scene_get_layers: no scene");
  if (h->f32 && (tau || varpi || zw || tau_sum) && !h->d_tau)
    return fail(h, MOM_ESTATE, "mom_scene_get_layers: a Float32 handle keeps the Float64 layer arrays only after mom_scene_set_optics");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t S = h->S, Nz = h->Nz;
  if (ndoubl) std::copy(h->nd.begin(), h->nd.end(), ndoubl);
  if (iface) std::copy(h->iface.begin(), h->iface.end(), iface);
  if (tau) HIPCHK(h, hipMemcpyAsync(tau, h->d_tau, S * Nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (varpi) HIPCHK(h, hipMemcpyAsync(varpi, h->d_varpi, S * Nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (zw) HIPCHK(h, hipMemcpyAsync(zw, h->d_zw, (size_t)h->K * S * Nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (tau_sum) HIPCHK(h, hipMemcpyAsync(tau_sum, h->d_tau_sum, S * (Nz + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}


// =========================================================================================
// rotational-Raman path (BASELINE config 5): rt_run(::RRS) -- kernels in mom_rrs.hip
// =========================================================================================
#define RRSCHK(h, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {                                                                                   \
      if ((h)->rrs && !(h)->rrs->err.empty()) {                                                                \
        const std::string m__ = (h)->rrs->err;                                                                 \
        (h)->rrs->err.clear();                                                                                 \
        return fail(h, MOM_EUNSUPPORTED, m__.c_str());                                                         \
      }                                                                                                        \
      char buf__[512];                                                                                         \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return fail(h, MOM_EHIP, buf__);                                                                         \
    }                                                                                                          \
  } while (0)

static momr::Streams rrs_streams(const mom_t *h) {
  momr::Streams q{};
  q.mu = h->d_mu; q.wt = h->d_wt;
  for (int k = 0; k < 4; ++k) { q.I0[k] = h->q.I0[k]; q.D[k] = h->q.D[k]; }
  q.N = h->N; q.nS = h->nS; q.imu0 = h->q.imu0; q.strict_idx = h->strict; q.mu0 = h->q.mu0;
  return q;
}
static int rrs_ready(mom_t *h, const char *who) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (h->dtype != 0) return fail(h, MOM_EINVAL, "the RRS path is Float64 only");
  if (!h->rrs) { static thread_local char b[128]; snprintf(b, sizeof b, "%s: call mom_rrs_set first", who); return fail(h, MOM_ESTATE, b); }
  if (!h->streams_set) return fail(h, MOM_ESTATE, "mom_set_streams must be called first");
  HIPCHK(h, hipSetDevice(h->device));
  h->rrs->fast = false;  // only mom_rt_run_rrs switches the deferred / derived mode on, for its own duration
  return MOM_OK;
}

extern "C" int mom_rrs_set(mom_t *h, int nRaman, const int *i_l1l0, const double *varpi_l1l0, int rrs_strict_reference) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  F64_ONLY(h, "mom_rrs_set");
  if (nRaman <= 0 || !i_l1l0 || !varpi_l1l0) return fail(h, MOM_EINVAL, "mom_rrs_set: bad argument");
  if (h->N > 64) return fail(h, MOM_EUNSUPPORTED, "mom_rrs_set: the RRS kernels cover operator edges N <= 64 (the reference's RRS shape is N = 15)");
  for (int k = 0; k < nRaman; ++k)
    if (std::abs(i_l1l0[k]) >= h->S) return fail(h, MOM_EINVAL, "mom_rrs_set: |i_l1l0| must be < nSpec (get_n0_n1 fails in the reference)");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  momr::destroy(h->rrs);
  h->rrs = nullptr;
  h->rrs_scene = false;
  const hipError_t e = momr::create(&h->rrs, h->stream, h->N, h->nS, h->S, nRaman, i_l1l0, varpi_l1l0, rrs_strict_reference ? 1 : 0);
  if (e != hipSuccess) {
    momr::destroy(h->rrs);
    h->rrs = nullptr;
    char buf[256];
    snprintf(buf, sizeof buf, "mom_rrs_set: allocating the RRS layers failed: %s", hipGetErrorString(e));
    return fail(h, MOM_EHIP, buf);
  }
  if (h->opt_rrs_kernels >= 0) h->rrs->kopt = h->opt_rrs_kernels;
  return MOM_OK;
}

extern "C" int mom_rrs_set_shard(mom_t *h, int nSpec_global, int n_glob0, int n1_lo, int n1_hi) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!h->rrs) return fail(h, MOM_ESTATE, "mom_rrs_set_shard: call mom_rrs_set first");
  if (n_glob0 < 0 || n_glob0 + h->S > nSpec_global || n1_lo < 0 || n1_lo > n1_hi || n1_hi > h->S)
    return fail(h, MOM_EINVAL, "mom_rrs_set_shard: need 0 <= n_glob0, n_glob0 + nSpec <= nSpec_global, 0 <= n1_lo <= n1_hi <= nSpec");
  if (n1_hi > n1_lo) {
    // every source index n1 + i_l1l0 of an owned point must be local or off the GLOBAL grid
    const int H = h->rrs->max_off;
    if ((n_glob0 > 0 && n1_lo < H) || (n_glob0 + h->S < nSpec_global && h->S - n1_hi < H))
      return fail(h, MOM_EINVAL, "mom_rrs_set_shard: the halo is shorter than max |i_l1l0| on an interior edge");
  }
  h->rrs->n_glob0 = n_glob0;
  h->rrs->n1_lo = n1_lo;
  h->rrs->n1_hi = n1_hi;
  return MOM_OK;
}

static int rrs_check(mom_t *h) {
  int info = 0;
  HIPCHK(h, hipMemcpyAsync(&info, h->rrs->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (info) {
    HIPCHK(h, hipMemsetAsync(h->rrs->d_info, 0, sizeof(int), h->stream));
    char buf[128];
    snprintf(buf, sizeof buf, "zero pivot at elimination step %d while inverting (I - R r) (RRS path)", info);
    return fail(h, MOM_ESINGULAR, buf);
  }
  return MOM_OK;
}

static double *rrs_which(mom_t *h, int which, bool *matrix, size_t *nblk) {
  momr::State *s = h->rrs;
  if (which < 0 || which >= 30) return nullptr;
  const int grp = which / 6, k = which % 6;
  *matrix = k < 4;
  *nblk = (size_t)s->S * (grp >= 3 ? (size_t)s->nR : 1);
  switch (grp) {
    case 0: return s->added[(k == momr::R_PM || k == momr::T_MM) ? 0 : s->cur][k];
    case 1: return s->comp[s->ccur][k];
    case 2: return s->surf[k];
    case 3: return s->ie_added[k];
    default: return s->ie_comp[k];
  }
}
extern "C" int mom_rrs_upload(mom_t *h, int which, const double *src) {
  int rc = rrs_ready(h, "mom_rrs_upload");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  bool matrix = false;
  size_t nblk = 0;
  double *p = rrs_which(h, which, &matrix, &nblk);
  if (!p || !src) return fail(h, MOM_EINVAL, "mom_rrs_upload: bad argument");
  if (which >= 18 && which < 24) {
    RRSCHK(h, momr::ensure_pm(h->rrs, rrs_streams(h)));
    momr::mark_uploaded(h->rrs);
  }
  HIPCHK(h, momr::upload(h->rrs, p, src, matrix, nblk));  // ABI memory order -> padded device blocks
  return MOM_OK;
}
extern "C" int mom_rrs_download(mom_t *h, int which, double *dst) {
  int rc = rrs_ready(h, "mom_rrs_download");
  if (rc) return rc;
  bool matrix = false;
  size_t nblk = 0;
  double *p = rrs_which(h, which, &matrix, &nblk);
  if (!p || !dst) return fail(h, MOM_EINVAL, "mom_rrs_download: bad argument");
  if (which >= 18 && which < 24) RRSCHK(h, momr::ensure_pm(h->rrs, rrs_streams(h)));
  HIPCHK(h, momr::download(h->rrs, dst, p, matrix, nblk));
  return MOM_OK;
}

extern "C" int mom_rrs_elemental(mom_t *h, int m, int ndoubl, const double *tau_sum, const double *dtau, const double *varpi,
                                 const double *Zpp, const double *Zmp, const double *fscattRayl, const double *Zpp_l1l0,
                                 const double *Zmp_l1l0) {
  int rc = rrs_ready(h, "mom_rrs_elemental");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  if (!tau_sum || !dtau || !varpi || !Zpp || !Zmp || !fscattRayl || !Zpp_l1l0 || !Zmp_l1l0 || ndoubl < 0 || ndoubl > 62)
    return fail(h, MOM_EINVAL, "mom_rrs_elemental: bad argument");
  const size_t S = h->S, NN = (size_t)h->N * h->N;
  const double *src[8] = {tau_sum, dtau, varpi, fscattRayl, Zpp, Zmp, Zpp_l1l0, Zmp_l1l0};
  for (int k = 0; k < 8; ++k) {
    const size_t cnt = (k < 4) ? S : NN;
    if (!h->d_rrs_op[k]) HIPCHK(h, dmalloc(&h->d_rrs_op[k], cnt));
    HIPCHK(h, hipMemcpyAsync(h->d_rrs_op[k], src[k], cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  double *const *d = h->d_rrs_op;
  RRSCHK(h, momr::elemental(h->rrs, rrs_streams(h), m, ndoubl, 0, d[0], d[1], d[2], d[4], d[5], 1, nullptr, d[3], d[6], d[7], true, true));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_rrs_doubling(mom_t *h, int ndoubl, double *expk) {
  int rc = rrs_ready(h, "mom_rrs_doubling");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  if (ndoubl < 0 || !expk) return fail(h, MOM_EINVAL, "mom_rrs_doubling: bad argument");
  momr::State *s = h->rrs;
  HIPCHK(h, hipMemcpyAsync(s->expk[s->cur], expk, (size_t)h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RRSCHK(h, momr::doubling(s, rrs_streams(h), ndoubl));
  HIPCHK(h, hipMemcpyAsync(expk, s->expk[s->cur], (size_t)h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return rrs_check(h);
}

extern "C" int mom_rrs_interaction(mom_t *h, int iface, int with_surface_layer) {
  int rc = rrs_ready(h, "mom_rrs_interaction");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  if (iface < 0 || iface > 3) return fail(h, MOM_EINVAL, "mom_rrs_interaction: iface must be 0..3");
  RRSCHK(h, momr::interaction(h->rrs, rrs_streams(h), iface, with_surface_layer != 0));
  return rrs_check(h);
}

extern "C" int mom_rrs_copy_added_to_composite(mom_t *h) {
  int rc = rrs_ready(h, "mom_rrs_copy_added_to_composite");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  RRSCHK(h, momr::copy_added_to_composite(h->rrs, rrs_streams(h)));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_rrs_surface_lambertian(mom_t *h, int m, double albedo, const double *tau_tot) {
  int rc = rrs_ready(h, "mom_rrs_surface_lambertian");
  if (rc) return rc;
  h->rrs->dirty = true;  // operator-level write: the next scene-level run starts from zeroed layers again
  if (!tau_tot) return fail(h, MOM_EINVAL, "mom_rrs_surface_lambertian: bad argument");
  HIPCHK(h, hipMemcpyAsync(h->d_vec[0], tau_tot, (size_t)h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RRSCHK(h, momr::surface(h->rrs, rrs_streams(h), m, 0, albedo, h->d_vec[0], nullptr, nullptr));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

extern "C" int mom_scene_set_rrs(mom_t *h, const double *fscattRayl, const double *Zpp_l1l0, const double *Zmp_l1l0) {
  int rc = rrs_ready(h, "mom_scene_set_rrs");
  if (rc) return rc;
  if (!h->scene_set) return fail(h, MOM_ESTATE, "mom_scene_set_rrs: call mom_scene_set / mom_scene_set_optics first");
  if (!fscattRayl || !Zpp_l1l0 || !Zmp_l1l0) return fail(h, MOM_EINVAL, "mom_scene_set_rrs: bad argument");
  if (h->Nk != h->N) return fail(h, MOM_ESTATE, "mom_scene_set_rrs: the scene was set with a padded operator edge (MOM_OPT_STRIP_PAD)");
  const size_t NN = (size_t)h->N * h->N;
  if ((rc = upload_new(h, &h->d_fscatt, fscattRayl, (size_t)h->S * h->Nz))) return rc;
  if ((rc = upload_new(h, &h->d_Zr[0], Zpp_l1l0, NN * h->scene_M))) return rc;
  if ((rc = upload_new(h, &h->d_Zr[1], Zmp_l1l0, NN * h->scene_M))) return rc;
  h->rrs_scene = true;
  return MOM_OK;
}

extern "C" int mom_rt_run_rrs(mom_t *h) {
  int rc = rrs_ready(h, "mom_rt_run_rrs");
  if (rc) return rc;
  if (!h->scene_set || !h->rrs_scene) return fail(h, MOM_ESTATE, "mom_rt_run_rrs: call mom_scene_set and mom_scene_set_rrs first");
  momr::State *s = h->rrs;
  const momr::Streams q = rrs_streams(h);
  const size_t S = h->S, NN = (size_t)h->N * h->N;
  const int Nz = h->Nz, K = h->K, M = h->scene_M;
  momr::timing_reset(s, true);
  s->fast = true;   // deferred inelastic elemental, derived ier+- / iet-- (corrected position)
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  RRSCHK(h, momr::begin_run(s, h->nVza));
  for (int m = 0; m < M; ++m) {
    for (int iz = 0; iz < Nz; ++iz) {                                              // rt_run.jl:143-165
      const int nd = h->nd[iz];
      RRSCHK(h, momr::elemental(s, q, m, nd, nd, h->d_tau_sum + S * iz, h->d_tau + S * iz, h->d_varpi + S * iz,
                                h->d_Zpp + NN * K * m, h->d_Zmp + NN * K * m, K, h->d_zw + (size_t)K * S * iz,
                                h->d_fscatt + S * iz, h->d_Zr[0] + NN * m, h->d_Zr[1] + NN * m, true, true));
      RRSCHK(h, momr::doubling(s, q, nd));
      if (iz == 0) RRSCHK(h, momr::copy_added_to_composite(s, q));                    // rt_kernel.jl:326-333
      else RRSCHK(h, momr::interaction(s, q, h->iface[iz], false));
    }
    RRSCHK(h, momr::surface(s, q, m, h->surf_kind, h->albedo, h->d_tau_sum + S * Nz,            // rt_run.jl:168-175
                            h->surf_kind == 1 ? h->d_Rsurf + NN * m : nullptr, h->d_albedo_spec));
    RRSCHK(h, momr::interaction(s, q, h->iface[Nz - 1], true));                    // rt_run.jl:179-185 (Q6)
    RRSCHK(h, momr::postprocess(s, q, m, h->nVza, h->d_node, h->d_cos, h->d_sin, M, m == 0 ? 0.5 : 1.0));
  }
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  s->timing = false;
  s->fast = false;
  return MOM_OK;
}

extern "C" int mom_get_hdr_rrs(mom_t *h, double *hdr, double *bhr_uw, double *bhr_dw) {
  int rc = rrs_ready(h, "mom_get_hdr_rrs");
  if (rc) return rc;
  momr::State *s = h->rrs;
  if (!s->d_out || !hdr || !bhr_uw || !bhr_dw) return fail(h, MOM_ESTATE, "mom_get_hdr_rrs: no run / null output");
  const size_t tot = (size_t)s->out_nVza * h->nS * h->S, fl = (size_t)h->nS * h->S;
  HIPCHK(h, hipMemcpyAsync(hdr, s->d_out + 4 * tot, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(bhr_uw, s->d_out + 5 * tot, fl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(bhr_dw, s->d_out + 5 * tot + fl, fl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MOM_OK;
}

// The seven spectra of rt_run(::RRS)'s return tuple (rt_run.jl:226), restricted to the points this rank owns
// (mom_rrs_set_shard: [n1_lo, n1_hi) of the window), packed on the device: [R | T | ieR | ieT | hdr][nVza, nStokes, per],
// then [bhr_uw | bhr_dw][nStokes, per]; `per` >= the owned count, the tail of every spectrum is zero (ragged last shard).
// The spectral index is the slowest one of every output array, so an owned slice is one contiguous piece per spectrum.
static int rrs_pack_owned(mom_t *h, int per, double *d_dst) {
  momr::State *s = h->rrs;
  if (!s->d_out) return fail(h, MOM_ESTATE, "mom_get_spectra_rrs_device: no run");
  const int own = s->n1_hi - s->n1_lo;
  if (per < own || per <= 0) return fail(h, MOM_EINVAL, "mom_get_spectra_rrs_device: per must be >= the owned point count");
  const size_t a = (size_t)s->out_nVza * h->nS, b = (size_t)h->nS, S = (size_t)h->S;
  if (own < per) HIPCHK(h, hipMemsetAsync(d_dst, 0, mom_rrs_spectra_count(h, per) * sizeof(double), h->stream));
  for (int k = 0; k < 7; ++k) {
    const size_t row = k < 5 ? a : b;
    const double *src = (k < 5 ? s->d_out + (size_t)k * a * S : s->d_out + 5 * a * S + (size_t)(k - 5) * b * S) + row * s->n1_lo;
    double *dst = k < 5 ? d_dst + (size_t)k * a * per : d_dst + 5 * a * per + (size_t)(k - 5) * b * per;
    if (own > 0) HIPCHK(h, hipMemcpyAsync(dst, src, row * own * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  }
  return MOM_OK;
}

extern "C" size_t mom_rrs_spectra_count(mom_t *h, int per) {
  if (!h || !h->rrs || per <= 0) return 0;
  const int nV = h->rrs->out_nVza > 0 ? h->rrs->out_nVza : h->nVza;
  return ((size_t)5 * nV * h->nS + (size_t)2 * h->nS) * (size_t)per;
}

extern "C" int mom_get_spectra_rrs_device(mom_t *h, int per, void *d_local) {
  int rc = rrs_ready(h, "mom_get_spectra_rrs_device");
  if (rc) return rc;
  if (!d_local) return fail(h, MOM_EINVAL, "mom_get_spectra_rrs_device: null buffer");
  return rrs_pack_owned(h, per, static_cast<double *>(d_local));
}

// The ONE collective of a sharded RRS run: every rank contributes the packed block of its owned points
// (above) and receives d_global [nranks][mom_rrs_spectra_count(h, per)]; asynchronous on the handle's stream, nothing
// crosses the host.
extern "C" int mom_allgather_rrs_device(mom_t *h, int per, void *d_global) {
  int rc = rrs_ready(h, "mom_allgather_rrs_device");
  if (rc) return rc;
  if (!h->comm) return fail(h, MOM_ESTATE, "mom_allgather_rrs_device: call mom_comm_init first");
  if (!d_global) return fail(h, MOM_EINVAL, "mom_allgather_rrs_device: null buffer");
  const size_t cnt = mom_rrs_spectra_count(h, per);
  if (cnt > h->rrs_send_cap) {
    if (h->d_rrs_send) (void)hipFree(h->d_rrs_send);
    h->d_rrs_send = nullptr; h->rrs_send_cap = 0;
    HIPCHK(h, dmalloc(&h->d_rrs_send, cnt));
    h->rrs_send_cap = cnt;
  }
  if ((rc = rrs_pack_owned(h, per, h->d_rrs_send))) return rc;
  return mom_allgather(h, h->d_rrs_send, d_global, cnt);
}

// test access: violations of the zero-padding invariant of the RRS layer arrays (mom_rrs.hip count_padding); 0 = intact
extern "C" int mom_rrs_check_padding(mom_t *h, unsigned long long *violations) {
  int rc = rrs_ready(h, "mom_rrs_check_padding");
  if (rc) return rc;
  if (!violations) return fail(h, MOM_EINVAL, "mom_rrs_check_padding: null output");
  RRSCHK(h, momr::ensure_pm(h->rrs, rrs_streams(h)));
  RRSCHK(h, momr::count_padding(h->rrs, violations));
  return MOM_OK;
}

extern "C" int mom_rrs_timers(mom_t *h, double *ms, int *launches, int n) {
  int rc = rrs_ready(h, "mom_rrs_timers");
  if (rc) return rc;
  if (!ms || !launches || n < momr::TK_COUNT + 1) return fail(h, MOM_EINVAL, "mom_rrs_timers: need room for 4 values");
  RRSCHK(h, momr::timing_read(h->rrs, ms, launches));
  float t = 0.f;
  if (hipEventElapsedTime(&t, h->ev[0], h->ev[3]) != hipSuccess) t = 0.f;
  ms[momr::TK_COUNT] = t;
  launches[momr::TK_COUNT] = 1;
  return MOM_OK;
}

extern "C" int mom_get_RT_rrs(mom_t *h, double *R_SFI, double *T_SFI, double *ieR_SFI, double *ieT_SFI, double *gpu_ms) {
  int rc = rrs_ready(h, "mom_get_RT_rrs");
  if (rc) return rc;
  momr::State *s = h->rrs;
  if (!s->d_out) return fail(h, MOM_ESTATE, "mom_get_RT_rrs: no run");
  const size_t tot = (size_t)s->out_nVza * h->nS * h->S;
  double *dst[4] = {R_SFI, T_SFI, ieR_SFI, ieT_SFI};
  for (int k = 0; k < 4; ++k)
    if (dst[k]) HIPCHK(h, hipMemcpyAsync(dst[k], s->d_out + tot * k, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if ((rc = rrs_check(h))) return rc;
  if (gpu_ms) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[3]) != hipSuccess) ms = 0.f;
    *gpu_ms = ms;
  }
  return MOM_OK;
}

#ifdef MOM_DIAG_STAMPS
extern "C" int mom_diag_read(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mom_diag_acc), 128 * sizeof(unsigned long long)) != hipSuccess) return MOM_EHIP;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mom_diag_acc), z, sizeof z) != hipSuccess) return MOM_EHIP;
  }
  return MOM_OK;
}
#endif

extern "C" int mom_timers(mom_t *h, double *ms, int n, int *kernel_launches) {
  if (!h) return fail(nullptr, MOM_EINVAL, "null handle");
  if (!ms || n < 4) return fail(h, MOM_EINVAL, "mom_timers: need room for 4 values");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->f32) {
    for (int k = 0; k < n; ++k) ms[k] = 0.0;
    int nl = 0;
    const int rc = momf_timers(h->f32, ms, &nl);
    if (rc) return fail(h, rc, momf_error(h->f32));
    if (n >= 8) { ms[4] = ms[0]; ms[6] = nl; }
    if (kernel_launches) *kernel_launches = nl;
    return MOM_OK;
  }
  HIPCHK(h, hipEventSynchronize(h->ev[3]));
  float t01, t12, t23, t03;
  HIPCHK(h, hipEventElapsedTime(&t01, h->ev[0], h->ev[1]));
  HIPCHK(h, hipEventElapsedTime(&t12, h->ev[1], h->ev[2]));
  HIPCHK(h, hipEventElapsedTime(&t23, h->ev[2], h->ev[3]));
  HIPCHK(h, hipEventElapsedTime(&t03, h->ev[0], h->ev[3]));
  ms[0] = t01; ms[1] = t12; ms[2] = t23; ms[3] = t03;
  if (n >= 8) {  // per-kernel sums: full-problem layer launches, reduced (m = 0) layer launches
    double full = 0.0, red = 0.0;
    float t;
    for (int z = 0; z < h->launches_full; ++z) { HIPCHK(h, hipEventElapsedTime(&t, h->ev_full[2 * z], h->ev_full[2 * z + 1])); full += t; }
    for (int z = 0; z < h->launches_red; ++z) { HIPCHK(h, hipEventElapsedTime(&t, h->ev_red[2 * z], h->ev_red[2 * z + 1])); red += t; }
    ms[4] = full; ms[5] = red; ms[6] = h->launches_full; ms[7] = h->launches_red;
  }
  if (kernel_launches) *kernel_launches = h->launches;
  return MOM_OK;
}
