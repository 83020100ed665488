// mom_host.hpp -- host-side helpers shared by the translation units of libmomcore.so.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <utility>

// hipFuncAttributeMaxDynamicSharedMemorySize is set once per (device, kernel image) and raised only when a
// larger LDS image is requested -- not on every launch.
inline hipError_t mom_allow_lds(const void *fn, size_t bytes) {
  static thread_local std::map<std::pair<int, const void *>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  size_t &g = granted[std::make_pair(dev, fn)];
  if (bytes <= g) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) g = bytes;
  return e;
}

// text for mom_last_global_error() (the thread's library-level error string, momcore.hip)
void mom_set_global_error(const char *msg);

// voigt.hip: one launch for all lines (device pointers, stream st); out[g] = acc  or  out[g] += factor * acc
hipError_t mom_voigt_launch(hipStream_t st, int nLines, const double *nu, const double *gamma_d, const double *y,
                            const double *S, const int *i0, const int *i1, int nGrid, const double *grid, double *out,
                            double factor, int accumulate, int sorted);

// resident HITRAN table of one absorber + the TIPS spline tables of its isotopologues (device pointers)
struct MomLineTable {
  int nLines, nIso, nTmax;
  const double *nu0, *S0, *g_air, *g_self, *E, *n_air, *d_air, *sqw;  // [nLines]
  const int *iso;                                                     // [nLines] index into the spline tables
  const int *nT;                                                      // [nIso] knots per isotopologue
  const double *tT, *tQ, *tZ;                                         // [nIso, nTmax] knots, values, second derivatives
};
// voigt.hip: per-line prefactors of one (p, T) on the device; *unsorted is set when the windows are not monotone
// every layer of a profile: prefactors of all (layer, line) pairs, then the line shapes of all (layer, grid point) pairs
hipError_t mom_voigt_profile_launch(hipStream_t st, const MomLineTable &tb, int Nz, size_t cap, int nGrid, const double *grid,
                                    const double *prm, double vmr, double wing, double *pf, int *win, int *unsorted, double *tau_abs,
                                    const double *factor);
hipError_t mom_line_prefactors_launch(hipStream_t st, const MomLineTable &tb, int nGrid, const double *grid, double p, double T,
                                      double vmr, double wing, double cgd, double *nu, double *gd, double *y, double *S, int *i0,
                                      int *i1, int *unsorted);

// mom_dual.hip: rt_run on ForwardDiff.Dual numbers (values + P partials) for the resident scene.  Device pointers unless noted;
// partial arrays have the layout of their value arrays with the partial index as the slowest axis, nullptr = zero partials.
struct MomDualScene {
  int N, nS, S, Nz, K, M, P, nVza, imu0, strict, surf_kind;
  double mu0, albedo;
  double I0[4], D[4];
  const double *mu, *wt;                                   // [N]
  const double *tau, *varpi, *zw, *Zpp, *Zmp, *tau_sum;    // [S,Nz], [S,Nz], [K,S,Nz], [N,N,K,M] x2, [S,Nz+1]
  const double *dtau, *dvarpi, *dzw, *dZpp, *dZmp;         // (.., P)
  const double *dalbedo;                                   // [P]
  const double *Rsurf, *dRsurf, *albedo_spec, *dalbedo_spec;  // [N,N,M](,P), [S](,P)
  const int *nd, *iface;                                   // HOST [Nz]
  const int *node;                                         // [nVza]
  const double *cos_mphi, *sin_mphi;                       // [nVza,M]
  double *R, *T, *dR, *dT;                                 // [nVza,nS,S], [nVza,nS,S,P]
  double *hdr, *dhdr, *bhr_uw, *bhr_dw, *dbhr_uw, *dbhr_dw; // [nVza,nS,S](,P); [nS,S](,P)
  double *dtau_sum_buf;                                    // [S,Nz+1,P] scratch
  int *info;
  hipStream_t stream;
  void **work;                                             // workspace owned by the handle (grown on demand)
  size_t *work_cap;
  size_t work_budget;                                      // bytes the operator workspace may take (units are chunked to fit)
};
size_t momd_bytes_per_unit(int N, int P);
int momd_run(const MomDualScene &sc, std::string *err);   // 0 ok, 1 unsupported, 2 HIP error (text in *err)

// argument blocks of the single-launch sweep kernels, shared by the launching translation unit (momcore.hip) and the
// kernels' own (mom_small.hip: momsm::k_sweep; mom_wave.hip: momw::k_wsweep)
template <class Real>
struct MomSmallSweepArgsT {
  int S, M, K, Nz, nVza, nS, imu0, pad;
  Real mu0, albedo;
  Real I0[4], D[4];
  // per-scene tables, the same for every spectral point (read through the scalar cache)
  const Real *mu, *wt, *sg;         // [N]
  const Real *F1, *F2, *SI;         // [N,N] i + N j: mu_j/(mu_i+mu_j), mu_j/(mu_i-mu_j), (1/mu_i)+(1/mu_j)
  const Real *Zpp, *Zmp;            // [N,N,K,M]
  const int *nd, *iface;              // [Nz]
  const int *node;                    // [nVza]
  const double *cos_mphi, *sin_mphi;  // [nVza,M]
  // per-point inputs
  const Real *tau, *varpi, *zw, *tau_sum;  // [S,Nz], [S,Nz], [K,S,Nz], [S,Nz+1]
  // outputs
  Real *R, *T, *hdr, *bhr_uw, *bhr_dw;  // [nVza,nS,S] x3, [nS,S] x2
  int *info;
  // r5: one (point, moment) per lane when `part` is given and M > 1: the terms of R_SFI / T_SFI go to part[m][R | T][nVza,nS,S]
  // and a second kernel adds them in ascending m (the order of the accumulator they replace); the lanes of m = 0 write hdr / bhr
  Real *part;
};
using MomSmallSweepArgs = MomSmallSweepArgsT<double>;
using MomSmallSweepArgsF = MomSmallSweepArgsT<float>;  // mom_small.hip with -DMOMS_FLOAT
template <class Real>
struct MomWaveSweepArgsT {
  int N, S, M, K, Nz, nVza, nS, imu0, inv_mode, pad;
  Real mu0, albedo;
  Real I0[4], D[4];
  const Real *mu, *wt, *sg;           // [N]
  const Real *Zpp, *Zmp;              // [N,N,K,M]
  const int *nd;                        // [Nz]
  const int *node;                      // [nVza]
  const double *cos_mphi, *sin_mphi;    // [nVza,M]
  const Real *tau, *varpi, *zw, *tau_sum;  // [S,Nz], [S,Nz], [K,S,Nz], [S,Nz+1]
  Real *R, *T, *hdr, *bhr_uw, *bhr_dw;
  int *info;
  // surface (mom_scene_set_surface): 0 LambertianSurfaceScalar(albedo), 1 BRDF matrices Rsurf [N,N,M] (every moment),
  // 2 LambertianSurfaceLegendre (albedo_spec [S]; j0+ = 0, T_SFI from m = 0 only: lambertian_surface.jl:112,131-132)
  int surf_kind, pad2;
  const Real *Rsurf, *albedo_spec;
};
using MomWaveSweepArgs = MomWaveSweepArgsT<double>;   // Float64 wave-per-point sweep (mom_wave.hip)
using MomWaveSweepArgsF = MomWaveSweepArgsT<float>;   // Float32 build of the same kernels (mom_wave.hip with -DMOMW_FLOAT)
