// mom_rrs.hip -- the rotational-Raman (RRS) branch of the CoreRT layer loop (BASELINE config 5):
//   rt_kernel!(::RRS)              src/CoreRT/CoreKernel/rt_kernel.jl:277-340
//   elemental_inelastic!(::RRS)    CoreKernel/elemental_inelastic.jl:23-91 (+ kernels :93-160, :320-402)
//   doubling_helper!(::RRS)        CoreKernel/doubling_inelastic.jl:13-134 (+ D kernels :291-311, :345-357)
//   interaction_helper!(::RRS, .)  CoreKernel/interaction_inelastic.jl:8-22, 28-76, 139-180, 230-340
//   postprocessing_vza!(::RRS)     tools/postprocessing_vza.jl:95-147
//
// Execution model.  The inelastic operators ie*[:, :, n1, dn] couple spectral point n1 with n0 = n1 + i_l1l0[dn], so a
// doubling step or an interaction cannot stay inside one spectral point's workgroup like the elastic path (mom_kernels.hpp):
// the ELASTIC state of all points is materialised in HBM per step, and each step is two launches:
//   * a POINT kernel, one wavefront per spectral point: the elastic update of the step (doubling.jl:43-68 /
//     interaction.jl:69-117 arithmetic) plus the operands every Raman pair of that point will need -- (I - r r)^-1 t,
//     t (I - r r)^-1 r, T-- (I - r R+-)^-1 ... -- written once, in the orientation their consumer loads coalesced;
//     the new elastic state goes to the other half of a double buffer, so the pairs still see the old one;
//   * a PAIR kernel, one wavefront per (n1, dn): the 9 (doubling) / 18 (interaction 11) operator products and 7 / 8
//     matrix-vector products of the reference's update formulas as register-tile MFMA products (mom_tile.hpp), reading
//     and writing each 4-D array block exactly once, in place.  N <= 16: one 16 x 16 tile per operator (the reference's own
//     RRS shape is N = 15: test/test_parameters/O2Parameters.yaml, IQU, l_trunc 5); N <= 32: 2 x 2 tiles; N <= 48 / <= 64: 3 x 3 / 4 x 4 tiles (second object, MOMR_BIG_TU).
//     r5: above N = 16 the pair kernels of the doubling step and of interface 11 run as one WORKGROUP of NT waves per pair instead
//     (mom_rrs_wg.hpp: a wave owns a tile column of every operator, left factors are read from LDS copies).
// Pairs are enumerated dn-major (n1 fastest): the blocks of the 4-D arrays are visited in memory order, i.e. every array is
// one sequential HBM stream per launch (n1-major order -- consecutive pairs 12 MB apart -- ran at 2.6 TB/s instead); the
// per-point operands (7 x N^2 x S doubles = 86 MB at C5) are re-read once per Raman line from L2 / Infinity Cache.  HBM-bound: 4 (doubling) / 12 (interaction) block transfers of N^2 doubles per pair against
// 36 / 72 MFMA instructions (NT = 1).
//
// The switch `strict_rrs` (rrs_strict_reference): 1 = the reference text as written, with the semantics of a single-threaded
// run; 0 = corrections D1..D5 listed in DESIGN.md ("RRS") and include/momcore.h (mom_rrs_set) -- and nothing else.
#include "mom_rrs.hpp"

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "mom_tile.hpp"

hipError_t momr_big_launch(int which, int nt, int v0, int v1, unsigned grid, void *stream, const void *args, int iface);

// The 3 x 3- and 4 x 4-tile images (32 < N <= 64) are built as a second object from this source (-DMOMR_BIG_TU, namespace
// momr_big, without -amdgpu-mfma-vgpr-form, which crashes the compiler on them): device code + ONE launcher, momr_big_launch
#ifdef MOMR_BIG_TU
#define MOMR_NS momr_big
#else
#define MOMR_NS momr
#endif
namespace MOMR_NS {
using namespace momt;
#ifdef MOMR_BIG_TU
using namespace momr;  // the enumerations and constants of mom_rrs.hpp
#endif

// diagnostic builds only (-DMOMR_DIAG_STAMPS, tools/phase_stamps_rrs.py): s_memtime deltas per code section of the pair
// kernels, wave 0 of the middle workgroup; every stamp first waits for the wave's outstanding memory operations, so a section's
// time includes the latency of the loads it consumes.  Never part of the shipped library.
#ifdef MOMR_DIAG_STAMPS
__device__ unsigned long long momr_diag_acc[64];
__device__ unsigned long long momr_diag_last;
#define MOMR_STAMP(id)                                                                                        \
  do {                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) {                                                 \
      unsigned long long n__;                                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory"); \
      momr_diag_acc[id] += n__ - momr_diag_last;                                                              \
      momr_diag_last = n__;                                                                                   \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
// ... without the wait for outstanding vector-memory operations (dbl_pair_body1: its prefetch must stay in flight)
#define MOMR_STAMP_NW(id)                                                                                     \
  do {                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) {                                                 \
      unsigned long long n__;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");                            \
      momr_diag_acc[id] += n__ - momr_diag_last;                                                              \
      momr_diag_last = n__;                                                                                   \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#else
#define MOMR_STAMP(id)
#define MOMR_STAMP_NW(id)
#endif


struct KArgs {
  int N, nS, S, nR, strict_idx, strict_rrs, n_glob0, n1_lo, n1_hi, last, nd, sh, m, imu0, nTerms;
  int P;          // row pitch of the device blocks: 16 (N <= 16) or 32; a matrix block is P x P doubles, a vector block P (zero padding)
  int derive_pm;  // corrected position: ier+- / iet-- are sgn (.) ier-+ / iet++ and are derived where they are read
  int fuse_el;    // first doubling step of a layer: the inelastic elemental layer is formed in registers, not loaded
  int dn_chunk;   // one-tile doubling pair kernel (dbl_pair_body1): Raman offsets per work item (n1, chunk)
  double mu0, albedo, weight;
  double I0[4], D[4];
  const double *mu, *wt;
  const int *off;
  const double *varpiR;
  double *a_cur[6], *a_nxt[6];
  double *expk_cur, *expk_nxt;
  double *sm[10];
  double *sv[6];
  double *jpseq;
  double *ie_a[6], *ie_c[6];
  double *c_cur[6], *c_nxt[6];
  double *x[6];  // the added layer of an interaction: the atmospheric one or the surface
  // elemental inputs
  const double *tau_sum, *tau, *varpi, *Zpp, *Zmp, *zw, *fscatt, *Zr_pp, *Zr_mp;
  int *info;
};

// Stokes component label of the D kernels: strict = mod(i_1based, n) -> 1, 2, .., n-1, 0
__device__ __forceinline__ int scomp(int i0, int n, int strict) { return strict ? ((i0 + 1) % n) : (i0 % n) + 1; }
__device__ __forceinline__ double dsgn(int ci, int cj) { return (((ci <= 2) && (cj <= 2)) || ((ci > 2) && (cj > 2))) ? 1.0 : -1.0; }

constexpr int kWavesPerBlock = 4;
// waves per SIMD the pair kernels are compiled for (register budget 512 / MOMR_WPE); measured in profiles/r03_C5_*.txt
#ifndef MOMR_FAST_DEFAULT
#define MOMR_FAST_DEFAULT 3
#endif
#ifndef MOMR_WPE
#define MOMR_WPE 2
#endif
#define MOMR_PAIR_ATTR __attribute__((amdgpu_waves_per_eu(MOMR_WPE, MOMR_WPE)))
#ifndef MOMR_WPE2
#define MOMR_WPE2 1
#endif
#define MOMR_PAIR_ATTR2 __attribute__((amdgpu_waves_per_eu(MOMR_WPE2, MOMR_WPE2)))
// software pipeline of the pair kernels' operand loads (next pair requested before the current one is computed): measured
// SLOWER on C5 (profiles/r04_C5_ab.txt: k_dbl_pair 160 -> 169 ms, k_int_pair 84 -> 119 ms per run) -- the kernels are
// bound by instruction issue, not by the latency of their first loads; kept as an experiment switch
#ifndef MOMR_PIPELINE
#define MOMR_PIPELINE 0
#endif

template <int NT>
__device__ __forceinline__ Geo make_geo(int N, unsigned char *smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Geo g;
  g.lr = lane & 15;
  g.lq = lane >> 4;
  g.N = N;
  unsigned char *base = smem + (size_t)wave * slice_bytes<NT>();
  g.xp = reinterpret_cast<double *>(base);
  g.ipiv = reinterpret_cast<int *>(base + slice_doubles<NT>() * 8);
  return g;
}

extern __shared__ __align__(16) unsigned char rrs_smem[];

// element (rho, kappa) of X_t is X[kappa][rho]: apply f(i = kappa, j = rho, value) to every element of an _t tile set
template <int NT, class F>
__device__ __forceinline__ void map_t(const Geo &g, Mat<NT> &X, F f) {
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) X.t[a][b][r] = f(g.col(b), g.row(a, r), X.t[a][b][r]);
}
template <int NT, class F>
__device__ __forceinline__ void vmap(const Geo &g, Vec<NT> &v, F f) {  // f(row, value)
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) v.t[a][r] = f(g.row(a, r), v.t[a][r]);
}

// ---------------------------------------------------------------------------------------------------------------------
// elastic elemental layer, one wavefront per spectral point: get_elem_rt! / get_elem_rt_SFI! / apply_D_elemental!
// (CoreKernel/elemental.jl:164-274) into the current half of the added-layer buffer, dtau = tau / 2^sh (sh = ndoubl when
// the caller passes the layer's tau, 0 when it passes dtau) and
// expk = exp(-dtau / mu0) (rt_kernel.jl:285-289).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_el_point(KArgs a) {
#pragma clang fp contract(off)
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int N = a.N, n = a.nS;
  const size_t NN = (size_t)N * N, BS = (size_t)a.P * a.P, VS = a.P;
  const int wave = threadIdx.x >> 6;
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
  for (int pt = blockIdx.x * kWavesPerBlock + wave; pt < a.S; pt += gridDim.x * kWavesPerBlock) {
    const double dtau = a.tau[pt] / (double)(1ull << a.sh), varpi = a.varpi[pt];
    Mat<NT> r_t, t_t;
    // element (rho, kappa) of X_t = X[i = kappa][j = rho]
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
      for (int tb = 0; tb < NT; ++tb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = g.col(tb), j = g.row(ta, rr);
          double rv = 0.0, tv = 0.0;
          if (i < N && j < N) {
            double zp = 0.0, zm = 0.0;
            for (int k = 0; k < a.nTerms; ++k) {
              const double w = a.zw ? a.zw[k + (size_t)a.nTerms * pt] : 1.0;
              zp += w * a.Zpp[i + (size_t)N * j + NN * k];
              zm += w * a.Zmp[i + (size_t)N * j + NN * k];
            }
            const double mui = a.mu[i], muj = a.mu[j], wj = a.wt[j] / wdiv;
            if (wj > 1.e-8) {
              rv = varpi * zm * (muj / (mui + muj)) * wj * (1 - exp(-dtau * ((1 / mui) + (1 / muj))));
              if (mui == muj) {
                if (i == j) tv = exp(-dtau / mui) * (1 + varpi * zp * (dtau / mui) * (a.wt[i] / wdiv));
              } else {
                tv = varpi * zp * (muj / (mui - muj)) * wj * (exp(-dtau / mui) - exp(-dtau / muj));
              }
            } else if (i == j) {
              tv = exp(-dtau / mui);
            }
          }
          r_t.t[ta][tb][rr] = rv;
          t_t.t[ta][tb][rr] = tv;
        }
    // sources: lanes of column 0 hold row i
    Vec<NT> jp = vzeros<NT>(), jm = vzeros<NT>();
    if (g.lr == 0) {
      const double mus = a.mu[i_start], att = exp(-a.tau_sum[pt] / mus);
#pragma unroll
      for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = g.row(ta, rr);
          if (i < N) {
            double zpI = 0.0, zmI = 0.0;
            for (int ii = i_start; ii < i_end; ++ii) {
              double zp = 0.0, zm = 0.0;
              for (int k = 0; k < a.nTerms; ++k) {
                const double w = a.zw ? a.zw[k + (size_t)a.nTerms * pt] : 1.0;
                zp += w * a.Zpp[i + (size_t)N * ii + NN * k];
                zm += w * a.Zmp[i + (size_t)N * ii + NN * k];
              }
              zpI += zp * a.I0[ii - i_start];
              zmI += zm * a.I0[ii - i_start];
            }
            const double mui = a.mu[i];
            double p;
            if (i >= i_start && i < i_end) p = wct02 * varpi * zpI * (dtau / mui) * exp(-dtau / mui);
            else p = wct02 * varpi * zpI * (mus / (mui - mus)) * (exp(-dtau / mui) - exp(-dtau / mus));
            double q = wct02 * varpi * zmI * (mus / (mui + mus)) * (1 - exp(-dtau * ((1 / mui) + (1 / mus))));
            p *= att;
            q *= att;
            if (a.nd >= 1) q = a.D[i % n] * q;
            jp.t[ta][rr] = p;
            jm.t[ta][rr] = q;
          }
        }
    }
    // apply_D_elemental! (elemental.jl:255-274)
    if (a.nd < 1) {
      Mat<NT> rpm = r_t, tmm = t_t;
      map_t<NT>(g, rpm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
      map_t<NT>(g, tmm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
      store_t<NT>(g, a.a_cur[R_PM] + BS * pt, rpm);
      store_t<NT>(g, a.a_cur[T_MM] + BS * pt, tmm);
    } else {
      map_t<NT>(g, r_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
    }
    store_t<NT>(g, a.a_cur[R_MP] + BS * pt, r_t);
    store_t<NT>(g, a.a_cur[T_PP] + BS * pt, t_t);
    storev<NT>(g, a.a_cur[J0P] + VS * pt, jp, 0);
    storev<NT>(g, a.a_cur[J0M] + VS * pt, jm, 0);
    if (g.lr == 0 && g.lq == 0) a.expk_cur[pt] = exp(-dtau / a.mu0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// elemental_inelastic!(::RRS) on the PERSISTENT added layer, one thread per element (i, j, n1, dn):
// get_elem_rt_RRS! (elemental_inelastic.jl:93-160) writes every element of ier-+ / iet++ (zeros off the grid);
// get_elem_rt_SFI_RRS! (:320-382) writes only on-grid entries of ieJ0+-, then multiplies EVERY entry of ieJ0- by D for
// ndoubl >= 1 (:378-380); apply_D_elemental_RRS! (:384-402).  dtau[n] = tau[n] / 2^nd.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ie_elemental(KArgs a) {
#pragma clang fp contract(off)
  const int N = a.N, n = a.nS;
  const size_t NN = (size_t)N * N;
  const size_t npairs = (size_t)(a.n1_hi - a.n1_lo) * a.nR;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NN * npairs) return;
  const int i = (int)(e % N), j = (int)((e / N) % N);
  const size_t pp = e / NN;
  const int span = a.n1_hi - a.n1_lo;
  const int n1 = a.n1_lo + (int)(pp % span), dn = (int)(pp / span);   // dn-major: consecutive blocks of the 4-D arrays
  const size_t u = (size_t)n1 + (size_t)a.S * dn;  // block index of the 4-D arrays
  const size_t o4 = (size_t)a.P * a.P * u + i + (size_t)a.P * j;
  const int n0 = n1 + a.off[dn];
  const double scl = (double)(1ull << a.sh);
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const double mui = a.mu[i], muj = a.mu[j], wj = a.wt[j] / wdiv;
  double r = 0.0, t = 0.0;
  const bool in = (n0 >= 0) && (n0 < a.S);
  if (in && wj > 1.e-8) {
    const double d1 = a.tau[n1] / scl, d0 = a.tau[n0] / scl;
    const double pre = a.varpiR[dn] * a.varpi[n0] * a.fscatt[n0];
    r = a.fscatt[n0] * a.varpiR[dn] * a.varpi[n0] * a.Zr_mp[i + (size_t)N * j] * (1 / ((mui / muj) + (d1 / d0))) *
        (1 - exp(-((d1 / mui) + (d0 / muj)))) * wj;
    if (mui == muj) {
      if (i == j) {
        const double wi = a.wt[i] / wdiv;
        if (fabs(d0 - d1) > 1.e-6) t = pre * a.Zr_pp[i + (size_t)N * i] * wi * (exp(-d0 / mui) - exp(-d1 / mui)) / (1 - (d1 / d0));
        else t = pre * a.Zr_pp[i + (size_t)N * i] * wi * (1 - exp(-d0 / muj));
      }
    } else {
      t = pre * a.Zr_pp[i + (size_t)N * j] * (1 / ((mui / muj) - (d1 / d0))) * wj * (exp(-d1 / mui) - exp(-d0 / muj));
    }
  }
  const int ci = scomp(i, n, a.strict_idx), cj = scomp(j, n, a.strict_idx);
  if (a.nd < 1) {
    const double s = dsgn(ci, cj);
    a.ie_a[R_PM][o4] = s * r;
    a.ie_a[T_MM][o4] = s * t;
  } else if (ci > 2) {
    r = -r;
  }
  a.ie_a[R_MP][o4] = r;
  a.ie_a[T_PP][o4] = t;
  if (j == 0) {
    const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
    const size_t o3 = i + (size_t)a.P * u;
    if (in) {
      const double d1 = a.tau[n1] / scl, d0 = a.tau[n0] / scl, mus = a.mu[i_start];
      double zpI = 0.0, zmI = 0.0;
      for (int ii = i_start; ii < i_end; ++ii) {
        zpI += a.Zr_pp[i + (size_t)N * ii] * a.I0[ii - i_start];
        zmI += a.Zr_mp[i + (size_t)N * ii] * a.I0[ii - i_start];
      }
      const double pre = a.varpiR[dn] * a.varpi[n0] * a.fscatt[n0];
      double jp, jm;
      if (i >= i_start && i < i_end) {
        if (fabs(d0 - d1) > 1.e-6) jp = (exp(-d0 / mui) - exp(-d1 / mui)) / ((d1 / d0) - 1) * pre * zpI * wct02;
        else jp = wct02 * pre * zpI * (1 - exp(-d0 / mus));
      } else {
        jp = wct02 * pre * zpI * (1 / ((mui / mus) - (d1 / d0))) * (exp(-d1 / mui) - exp(-d0 / mus));
      }
      jm = wct02 * pre * zmI * (1 / ((mui / mus) + (d1 / d0))) * (1 - exp(-((d1 / mui) + (d0 / mus))));
      const double att = exp(-a.tau_sum[n0] / mus);
      jp *= att;
      jm *= att;
      if (a.nd >= 1) jm = a.D[i % n] * jm;
      a.ie_a[J0P][o3] = jp;
      a.ie_a[J0M][o3] = jm;
    } else if (a.nd >= 1) {
      a.ie_a[J0M][o3] = a.D[i % n] * a.ie_a[J0M][o3];
    }
  }
}

// scratch matrix / vector slots of a doubling step
enum { SM_RT = 0, SM_TT = 1, SM_GT = 2, SM_GR = 3, SM_TTGP = 4, SM_TTGPR = 5 };
enum { SV_TMP1 = 0, SV_TMP2 = 1, SV_J1P = 2, SV_J1M = 3 };
// ... of an interaction (11)
enum { SI_T01 = 0, SI_T21 = 1, SI_RPM = 2, SI_TPP = 3, SI_R = 4, SI_TMM = 5, SI_G1RT = 6, SI_G1T = 7, SI_G2T = 8, SI_G2RT = 9 };
enum { SVI_G1V = 0, SVI_G2V = 1 };

// ---------------------------------------------------------------------------------------------------------------------
// doubling step, POINT kernel (doubling_inelastic.jl:47-59 and the elastic updates :90-95 (D1), :128-131)
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_dbl_point(KArgs a) {
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int n = a.nS, wave = threadIdx.x >> 6;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;  // block strides (padded pitch)
  int bad = 0;
  for (int pt = blockIdx.x * kWavesPerBlock + wave; pt < a.S; pt += gridDim.x * kWavesPerBlock) {
    const size_t om = NN * pt, ov = VS * pt;
    const Mat<NT> r_t = load_t<NT>(g, a.a_cur[R_MP] + om), t_t = load_t<NT>(g, a.a_cur[T_PP] + om);
    const Mat<NT> r_c = transpose<NT>(g, r_t), t_c = transpose<NT>(g, t_t);
    const Mat<NT> G_c = inv_one_minus<NT>(g, TN<NT>(g, r_t, r_c), &bad);   // (I - r r)^-1                      :47
    const Mat<NT> G_t = transpose<NT>(g, G_c);
    const Mat<NT> ttgp_t = TN<NT>(g, G_c, t_t);                            // (t G)^T                           :48
    const Mat<NT> ttgpr_t = TN<NT>(g, r_c, ttgp_t);                        // (t G r)^T
    store_t<NT>(g, a.sm[SM_RT] + om, r_c);
    store_t<NT>(g, a.sm[SM_TT] + om, t_c);
    store_t<NT>(g, a.sm[SM_GT] + om, TN<NT>(g, G_t, t_c));                 // G t, row-major
    store_t<NT>(g, a.sm[SM_GR] + om, TN<NT>(g, G_t, r_c));                 // G r, row-major
    store_t<NT>(g, a.sm[SM_TTGP] + om, ttgp_t);
    store_t<NT>(g, a.sm[SM_TTGPR] + om, ttgpr_t);
    double e = a.expk_cur[pt];
    Vec<NT> J = loadv2<NT>(g, a.a_cur[J0P] + ov, a.a_cur[J0M] + ov);      // (j0+ | j0-)
    const Vec<NT> J1 = vscale<NT>(J, e);                                   // (j1+ | j1-)                       :51-55
    storev<NT>(g, a.sv[SV_J1P] + ov, J1, 0);
    storev<NT>(g, a.sv[SV_J1M] + ov, J1, 1);
    // s = (j0+ + r j1- | j1- + r j0+),  tmp = G s = (tmp1 | tmp2)                                              :58-59
    auto mix = [&](const Vec<NT> &Jc) {
      Vec<NT> m;  // (j0+ | j1-)
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) m.t[ta] = (g.lr == 0) ? Jc.t[ta] : J1.t[ta];
      return m;
    };
    {
      const Vec<NT> mx = mix(J);
      const Vec<NT> s = vadd<NT>(mx, TNv<NT>(g, r_t, swap01<NT>(mx)));    // r (j1- | j0+)
      const Vec<NT> tmp = TNv<NT>(g, G_t, s);
      storev<NT>(g, a.sv[SV_TMP1] + ov, tmp, 0);
      storev<NT>(g, a.sv[SV_TMP2] + ov, tmp, 1);
    }
    // elastic source update: once (corrected) or nRaman times with expk squared every time (strict, D1)          :90-95
    const int reps = a.strict_rrs ? a.nR : 1;
    for (int k = 0; k < reps; ++k) {
      if (a.strict_rrs) storev<NT>(g, a.jpseq + ov + VS * a.S * k, J, 0);
      const Vec<NT> mx = mix(J);
      const Vec<NT> s = vadd<NT>(mx, TNv<NT>(g, r_t, swap01<NT>(mx)));
      const Vec<NT> q = TNv<NT>(g, ttgp_t, s);                             // (tG (j0+ + r j1-) | tG (j1- + r j0+))
      Vec<NT> Jn;
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) Jn.t[ta] = ((g.lr == 0) ? J1.t[ta] : J.t[ta]) + q.t[ta];
      J = Jn;
      e = e * e;
    }
    // r <- r + (tG r) t,  t <- tG t                                                                             :128-131
    Mat<NT> rn_t = add<NT>(r_t, TN<NT>(g, t_c, ttgpr_t));
    Mat<NT> tn_t = TN<NT>(g, t_c, ttgp_t);
    if (a.last) {  // apply_D_matrix! (doubling.jl:93-134) and apply_D_matrix_SFI! (:112-144)
      if (n > 1) {
        map_t<NT>(g, rn_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
        vmap<NT>(g, J, [&](int i, double v) { return (g.lr == 1 && scomp(i, n, a.strict_idx) > 2) ? -v : v; });
      }
      Mat<NT> rpm = rn_t, tmm = tn_t;
      if (n > 1) {
        map_t<NT>(g, rpm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
        map_t<NT>(g, tmm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
      }
      store_t<NT>(g, a.a_nxt[R_PM] + om, rpm);
      store_t<NT>(g, a.a_nxt[T_MM] + om, tmm);
    }
    store_t<NT>(g, a.a_nxt[R_MP] + om, rn_t);
    store_t<NT>(g, a.a_nxt[T_PP] + om, tn_t);
    storev<NT>(g, a.a_nxt[J0P] + ov, J, 0);
    storev<NT>(g, a.a_nxt[J0M] + ov, J, 1);
    if (g.lr == 0 && g.lq == 0) a.expk_nxt[pt] = e;
  }
  if (bad && g.lr == 0 && g.lq == 0) atomicMax(a.info, bad);
}

// The inelastic elemental layer of ONE pair in registers (the formulas of k_ie_elemental in tile form; ndoubl >= 1): ier-+
// and iet++ as _t tiles, ieJ0+- in column layout.  exp(-d1/mu_i) and exp(-d0/mu_j) are evaluated once per stream (2 exp per
// lane + 1 for the attenuation) and combined, i.e. 1 - e1 e0 stands for the reference's 1 - exp(-(d1/mu_i + d0/mu_j)):
// the same absolute accuracy (both are 1 minus a number rounded near 1).
// ND0: the layer has ndoubl = 0 (k_ie_elemental_tile below): no D sign on ier-+ and none on ieJ0- (elemental_inelastic.jl:378-402).
// ElemPre (k_ie_elemental_tile, NT <= 2): what an element needs that does not depend on the pair -- the two Raman phase-matrix
// entries, mu_i / mu_j and w_j / wdiv -- computed once per wave in front of its pair loop (the same values: two of the four
// divisions and three loads per element and pair less).
template <int NT>
struct ElemPre {
  Mat<NT> zmp, zpp, q;
  Vec<NT> wj;
};
template <int NT>
__device__ __forceinline__ ElemPre<NT> ie_elem_pre(const Geo &g, const KArgs &a) {
#pragma clang fp contract(off)
  ElemPre<NT> P;
  const int N = a.N;
  const double wdiv = (a.m == 0) ? 2.0 : 4.0;
#pragma unroll
  for (int ta = 0; ta < NT; ++ta)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = g.row(ta, rr);
      P.wj.t[ta][rr] = (j < N) ? a.wt[j] / wdiv : 0.0;
#pragma unroll
      for (int tb = 0; tb < NT; ++tb) {
        const int i = g.col(tb);
        const bool in = (i < N && j < N);
        P.zmp.t[ta][tb][rr] = in ? a.Zr_mp[i + (size_t)N * j] : 0.0;
        P.zpp.t[ta][tb][rr] = in ? a.Zr_pp[i + (size_t)N * j] : 0.0;
        P.q.t[ta][tb][rr] = in ? a.mu[i] / a.mu[j] : 1.0;
      }
    }
  return P;
}
template <int NT, bool ND0 = false, bool PRE = false>
__device__ __forceinline__ void ie_elem_tile(const Geo &g, const KArgs &a, int n1, int dn, int n0, Mat<NT> &a_t, Mat<NT> &b_t,
                                             CV<NT> &Jp, CV<NT> &Jm, const ElemPre<NT> *P = nullptr) {
#pragma clang fp contract(off)
  const int N = a.N, n = a.nS;
  const double scl = (double)(1ull << a.sh);
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const double d1 = a.tau[n1] / scl, d0 = a.tau[n0] / scl, ratio = d1 / d0;
  const double pre = a.varpiR[dn] * a.varpi[n0] * a.fscatt[n0];
  const double fs0 = a.fscatt[n0], vR = a.varpiR[dn], v0 = a.varpi[n0];
  CV<NT> e1C, e0C, muC;
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) {
    const int i = g.col(tb);
    const double mu = (i < N) ? a.mu[i] : 1.0;
    muC.c[tb] = mu;
    e1C.c[tb] = exp(-d1 / mu);
    e0C.c[tb] = exp(-d0 / mu);
  }
  const Vec<NT> e0R = c2r<NT>(g, e0C), muR = c2r<NT>(g, muC);
  const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
  const int base = (g.lq << 4);
  double e0s = 0.0, mus = 1.0;
#pragma unroll
  for (int tb = 0; tb < NT; ++tb)
    if ((i_start >> 4) == tb) {
      e0s = __shfl(e0C.c[tb], base | (i_start & 15));
      mus = __shfl(muC.c[tb], base | (i_start & 15));
    }
#pragma unroll
  for (int ta = 0; ta < NT; ++ta)
#pragma unroll
    for (int tb = 0; tb < NT; ++tb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = g.col(tb), j = g.row(ta, rr);
        double r = 0.0, t = 0.0;
        if (i < N && j < N) {
          const double mui = muC.c[tb], muj = muR.t[ta][rr], wj = PRE ? P->wj.t[ta][rr] : a.wt[j] / wdiv;
          if (wj > 1.e-8) {
            const double e1 = e1C.c[tb], e0 = e0R.t[ta][rr];
            const double q = PRE ? P->q.t[ta][tb][rr] : mui / muj;
            const double zmp = PRE ? P->zmp.t[ta][tb][rr] : a.Zr_mp[i + (size_t)N * j];
            r = fs0 * vR * v0 * zmp * (1 / (q + ratio)) * (1 - e1 * e0) * wj;
            if (mui == muj) {
              if (i == j) {
                const double wi = PRE ? wj : a.wt[i] / wdiv, e0i = e0C.c[tb];   // (i == j: the same weight)
                const double zpp = PRE ? P->zpp.t[ta][tb][rr] : a.Zr_pp[i + (size_t)N * i];
                if (fabs(d0 - d1) > 1.e-6) t = pre * zpp * wi * (e0i - e1) / (1 - ratio);
                else t = pre * zpp * wi * (1 - e0i);
              }
            } else {
              const double zpp = PRE ? P->zpp.t[ta][tb][rr] : a.Zr_pp[i + (size_t)N * j];
              t = pre * zpp * (1 / (q - ratio)) * wj * (e1 - e0);
            }
          }
          if (!ND0 && scomp(i, n, a.strict_idx) > 2) r = -r;  // apply_D_elemental_RRS!, ndoubl >= 1
        }
        a_t.t[ta][tb][rr] = r;
        b_t.t[ta][tb][rr] = t;
      }
  const double att = exp(-a.tau_sum[n0] / mus);
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) {
    const int i = g.col(tb);
    double jp = 0.0, jm = 0.0;
    if (i < N) {
      double zpI = 0.0, zmI = 0.0;
      for (int ii = i_start; ii < i_end; ++ii) {
        zpI += a.Zr_pp[i + (size_t)N * ii] * a.I0[ii - i_start];
        zmI += a.Zr_mp[i + (size_t)N * ii] * a.I0[ii - i_start];
      }
      const double mui = muC.c[tb], e1 = e1C.c[tb];
      if (i >= i_start && i < i_end) {
        if (fabs(d0 - d1) > 1.e-6) jp = (e0C.c[tb] - e1) / (ratio - 1) * pre * zpI * wct02;
        else jp = wct02 * pre * zpI * (1 - e0s);
      } else {
        jp = wct02 * pre * zpI * (1 / ((mui / mus) - ratio)) * (e1 - e0s);
      }
      jm = wct02 * pre * zmI * (1 / ((mui / mus) + ratio)) * (1 - e1 * e0s);
      jp *= att;
      jm = ND0 ? (jm * att) : a.D[i % n] * (jm * att);
    }
    Jp.c[tb] = jp;
    Jm.c[tb] = jm;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: the inelastic elemental layer in tile form, scene-level fast mode only: one wavefront per (n1, dn) forms ier-+ / iet++ /
// ieJ0+- of the pair with ie_elem_tile -- 2 exp per lane and stream instead of 3 exp and a handful of divisions per ELEMENT in
// k_ie_elemental (C5: 4.2 ms per launch, 12 % of a run, at 1.2 TB/s of stores) -- and stores whole zero-padded tiles.
// ND0 (ndoubl = 0): in the corrected position ier+- / iet-- are not written at all: they are sgn (.) ier-+ / sgn (.) iet++ and
// are derived where they are read (derive_pm, as after a doubling).  Off the grid: zeros (get_elem_rt_RRS! writes every
// element), sources untouched.  !ND0 (ndoubl >= 1; alternative to forming the layer inside the first doubling step, FUSE above):
// the D signs of apply_D_elemental_RRS!, off the grid ieJ0- is multiplied by D (elemental_inelastic.jl:378-380).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, bool ND0>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_ie_elemental_tile(KArgs a) {
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int n = a.nS, wave = threadIdx.x >> 6;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  const size_t span = (size_t)(a.n1_hi - a.n1_lo), npairs = span * a.nR, stride = (size_t)gridDim.x * kWavesPerBlock;
  constexpr bool PRE = (NT <= 2);  // (3 x 3 / 4 x 4 tiles: the three tile sets would not stay in registers)
  ElemPre<NT> P;
  if (PRE) P = ie_elem_pre<NT>(g, a);
  for (size_t p = (size_t)blockIdx.x * kWavesPerBlock + wave; p < npairs; p += stride) {
    const int n1 = a.n1_lo + (int)(p % span), dn = (int)(p / span);
    const int n0 = n1 + a.off[dn];
    const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
    if (n0 < 0 || n0 >= a.S) {
      store_t<NT>(g, a.ie_a[R_MP] + o4, zeros<NT>());
      store_t<NT>(g, a.ie_a[T_PP] + o4, zeros<NT>());
      if (ND0 && !a.derive_pm) {
        store_t<NT>(g, a.ie_a[R_PM] + o4, zeros<NT>());
        store_t<NT>(g, a.ie_a[T_MM] + o4, zeros<NT>());
      }
      if (!ND0) {
        CV<NT> Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
#pragma unroll
        for (int tb = 0; tb < NT; ++tb) Jm.c[tb] = a.D[g.col(tb) % n] * Jm.c[tb];
        storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
      }
      continue;
    }
    Mat<NT> a_t, b_t;
    CV<NT> Jp, Jm;
    ie_elem_tile<NT, ND0, PRE>(g, a, n1, dn, n0, a_t, b_t, Jp, Jm, &P);
    store_t<NT>(g, a.ie_a[R_MP] + o4, a_t);
    store_t<NT>(g, a.ie_a[T_PP] + o4, b_t);
    storeC<NT>(g, a.ie_a[J0P] + o3, Jp);
    storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
    if (ND0 && !a.derive_pm) {
      if (n > 1) {
        map_t<NT>(g, a_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
        map_t<NT>(g, b_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
      }
      store_t<NT>(g, a.ie_a[R_PM] + o4, a_t);
      store_t<NT>(g, a.ie_a[T_MM] + o4, b_t);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// doubling step, PAIR kernel (doubling_inelastic.jl:61-89 and :98-125)
// ---------------------------------------------------------------------------------------------------------------------
// MODE: 0 corrected position, not the last step; 1 corrected, last step (D2 / D3 folded into the stores); 2 strict position.
// The mode is a template parameter like FUSE: as run-time flags these branches cost a register move per tile element.
template <int NT, bool FUSE, int MODE>
__device__ __forceinline__ void dbl_pair_body(const KArgs &a) {
  constexpr bool STRICT = (MODE == 2);
#ifdef MOMR_DIAG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) {
    unsigned long long n__;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");
    momr_diag_last = n__;
  }
#endif
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int n = a.nS, wave = threadIdx.x >> 6;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;  // block strides (padded pitch)
  const size_t npairs = (size_t)(a.n1_hi - a.n1_lo) * a.nR;
  constexpr bool fuseD = (MODE == 1);  // a.last && !strict: D2/D3 (corrected) folded into the last step's stores
  // Software pipeline over the pairs of this wave (one-tile images: the register budget of the 2 x 2-tile ones is spent):
  // the operand tiles of pair p + stride are requested before pair p is computed, so their HBM / L2 latency (a wave waits
  // for ~40 loads per pair, and only two waves share a SIMD) runs under the MFMA chains of the current pair.
  constexpr bool PIPE = (NT == 1) && MOMR_PIPELINE;
  struct Pre {
    Mat<NT> a_t, b_t, r1_t, ttgp1_t, r0_c, gt0_c;
    CV<NT> Jp, Jm;
    bool have;
  } nx;
  nx.have = false;
  const size_t stride = (size_t)gridDim.x * kWavesPerBlock, span = (size_t)(a.n1_hi - a.n1_lo);
  auto prefetch = [&](size_t q) {
    nx.have = false;
    if (!PIPE || q >= npairs) return;
    const int qn1 = a.n1_lo + (int)(q % span), qdn = (int)(q / span);
    const int qn0 = qn1 + a.off[qdn];
    if (qn0 < 0 || qn0 >= a.S) return;
    const size_t qu = (size_t)qn1 + (size_t)a.S * qdn, q4 = NN * qu, q3 = VS * qu, qm1 = NN * qn1, qm0 = NN * qn0;
    if (!FUSE) {
      nx.a_t = load_t<NT>(g, a.ie_a[R_MP] + q4);
      nx.b_t = load_t<NT>(g, a.ie_a[T_PP] + q4);
      nx.Jp = loadC<NT>(g, a.ie_a[J0P] + q3);
      nx.Jm = loadC<NT>(g, a.ie_a[J0M] + q3);
    }
    nx.r1_t = load_t<NT>(g, a.a_cur[R_MP] + qm1);
    nx.ttgp1_t = load_t<NT>(g, a.sm[SM_TTGP] + qm1);
    nx.r0_c = load_t<NT>(g, a.sm[SM_RT] + qm0);
    nx.gt0_c = load_t<NT>(g, a.sm[SM_GT] + qm0);
    nx.have = true;
  };
  prefetch((size_t)blockIdx.x * kWavesPerBlock + wave);
  for (size_t p = (size_t)blockIdx.x * kWavesPerBlock + wave; p < npairs; p += stride) {
    const int n1 = a.n1_lo + (int)(p % span), dn = (int)(p / span);
    const int n0 = n1 + a.off[dn];
    const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
    if (n0 < 0 || n0 >= a.S) {  // get_n0_n1 (inelastic_helper.jl:13-21): no update off the grid ...
      if (PIPE) prefetch(p + stride);
      if (FUSE) {               // ... but the deferred elemental writes zeros there and multiplies ieJ0- by D (:378-380)
        store_t<NT>(g, a.ie_a[R_MP] + o4, zeros<NT>());
        store_t<NT>(g, a.ie_a[T_PP] + o4, zeros<NT>());
        CV<NT> Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
#pragma unroll
        for (int tb = 0; tb < NT; ++tb) Jm.c[tb] = a.D[g.col(tb) % n] * Jm.c[tb];
        storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
      }
      if (fuseD) {              // ... and the corrected D kernels visit every (n, dn)
        Mat<NT> an_t = FUSE ? zeros<NT>() : load_t<NT>(g, a.ie_a[R_MP] + o4);
        Mat<NT> bn_t = FUSE ? zeros<NT>() : load_t<NT>(g, a.ie_a[T_PP] + o4);
        if (n > 1) {
          map_t<NT>(g, an_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
          store_t<NT>(g, a.ie_a[R_MP] + o4, an_t);
          CV<NT> Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
#pragma unroll
          for (int tb = 0; tb < NT; ++tb)
            if (scomp(g.col(tb), n, a.strict_idx) > 2) Jm.c[tb] = -Jm.c[tb];
          storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
          map_t<NT>(g, an_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
          map_t<NT>(g, bn_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
        }
        if (!a.derive_pm) {
          store_t<NT>(g, a.ie_a[R_PM] + o4, an_t);
          store_t<NT>(g, a.ie_a[T_MM] + o4, bn_t);
        }
      }
      continue;
    }
    MOMR_STAMP(0);  // loop overhead, off-grid pairs
    const size_t m1 = NN * n1, m0 = NN * n0, v0 = VS * n0;
    Mat<NT> a_t, b_t;
    CV<NT> Jp, Jm;  // ieJ0+, ieJ0-
    Mat<NT> r1_t, ttgp1_t, r0_c, gt0_c;
    if (PIPE && nx.have) {  // requested one iteration ago
      if (!FUSE) { a_t = nx.a_t; b_t = nx.b_t; Jp = nx.Jp; Jm = nx.Jm; }
      r1_t = nx.r1_t; ttgp1_t = nx.ttgp1_t; r0_c = nx.r0_c; gt0_c = nx.gt0_c;
    } else {
      if (!FUSE) {
        a_t = load_t<NT>(g, a.ie_a[R_MP] + o4);
        b_t = load_t<NT>(g, a.ie_a[T_PP] + o4);
        Jp = loadC<NT>(g, a.ie_a[J0P] + o3);
        Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
      }
      r1_t = load_t<NT>(g, a.a_cur[R_MP] + m1); ttgp1_t = load_t<NT>(g, a.sm[SM_TTGP] + m1);
      r0_c = load_t<NT>(g, a.sm[SM_RT] + m0); gt0_c = load_t<NT>(g, a.sm[SM_GT] + m0);
    }
    if (FUSE) ie_elem_tile<NT>(g, a, n1, dn, n0, a_t, b_t, Jp, Jm);
    // the three operands of the last products, requested up front as well (they used to be loaded between the products)
    Mat<NT> gr0_c, t0_c, ttgpr1_t;
    if (PIPE) {
      gr0_c = load_t<NT>(g, a.sm[SM_GR] + m0); t0_c = load_t<NT>(g, a.sm[SM_TT] + m0);
      ttgpr1_t = load_t<NT>(g, a.sm[SM_TTGPR] + m1);
    }
    if (PIPE) prefetch(p + stride);
    MOMR_STAMP(1);  // the pair's first eight operand loads (+ the fused elemental layer)
    const Mat<NT> a_c = transpose<NT>(g, a_t);
    MOMR_STAMP(2);  // transpose of ier
    // X = ier r0 + r1 ier
    const Mat<NT> X_t = TNacc<NT>(g, a_c, r1_t, TN<NT>(g, r0_c, a_t));
    MOMR_STAMP(3);  // X: two products
    // ---- sources (matrix-vector products on the vector ALU: mom_tile.hpp)                                    :61-89
    {
      const double e1 = a.expk_cur[n1];
      const CV<NT> J1p = cscale<NT>(Jp, e1), J1m = cscale<NT>(Jm, e1);                           // ieJ1+, ieJ1-   :52-56
      const double *jp0 = STRICT ? a.jpseq + v0 + VS * a.S * dn : a.a_cur[J0P] + v0;
      const CV<NT> a_j1m = mv_t<NT>(g, a_t, loadR<NT>(g, a.sv[SV_J1M] + v0));                    // ier j1-[n0]
      const CV<NT> a_jp = mv_t<NT>(g, a_t, loadR<NT>(g, jp0));                                   // ier j0+[n0]
      const Vec<NT> tm1 = loadR<NT>(g, a.sv[SV_TMP1] + v0), tm2 = loadR<NT>(g, a.sv[SV_TMP2] + v0);
      const CV<NT> X1 = mv_t<NT>(g, X_t, tm1), X2 = mv_t<NT>(g, X_t, tm2);
      const CV<NT> b1 = mv_t<NT>(g, b_t, tm1);                                                   // iet++ tmp1
      const CV<NT> b2 = STRICT ? mv_t<NT>(g, load_t<NT>(g, a.ie_a[T_MM] + o4), tm2)        // D5: iet-- as the array holds it
                                     : mv_t<NT>(g, b_t, tm2);
      const CV<NT> uu = cadd<NT>(cadd<NT>(Jp, mv_t<NT>(g, r1_t, c2r<NT>(g, J1m))), cadd<NT>(a_j1m, X1));
      const CV<NT> Jpn = cadd<NT>(cadd<NT>(J1p, mv_t<NT>(g, ttgp1_t, c2r<NT>(g, uu))), b1);      // new ieJ0+
      const CV<NT> rv2 = mv_t<NT>(g, r1_t, c2r<NT>(g, Jpn));                                     // r1 ieJ0+(new)
      const CV<NT> u2 = cadd<NT>(cadd<NT>(J1m, rv2), cadd<NT>(a_jp, X2));
      CV<NT> Jmn = cadd<NT>(cadd<NT>(Jm, mv_t<NT>(g, ttgp1_t, c2r<NT>(g, u2))), b2);             // new ieJ0-
      if (fuseD && n > 1) {
#pragma unroll
        for (int tb = 0; tb < NT; ++tb)
          if (scomp(g.col(tb), n, a.strict_idx) > 2) Jmn.c[tb] = -Jmn.c[tb];
      }
      storeC<NT>(g, a.ie_a[J0P] + o3, Jpn);
      storeC<NT>(g, a.ie_a[J0M] + o3, Jmn);
    }
    MOMR_STAMP(4);  // source vectors: 4 vector loads, 10 mat-vecs, 4 column -> row conversions, 2 stores
    // ---- operators                                                                                            :98-125
    const Mat<NT> b_c = transpose<NT>(g, b_t);
    const Mat<NT> Y_c = TN<NT>(g, X_t, gt0_c);                                // X G t[n0]
    const Mat<NT> W_c = add<NT>(b_c, Y_c);
    Mat<NT> bn_t = TNacc<NT>(g, gt0_c, b_t, TN<NT>(g, W_c, ttgp1_t));         // tG (iet + Y) + iet G t[n0]
    const Mat<NT> bn_c = transpose<NT>(g, bn_t);
    const Mat<NT> V_c = add<NT>(bn_c, Y_c);
    MOMR_STAMP(5);  // iet: 2 transposes, 3 products
    if (!PIPE) {  // default: loaded where they are used
      gr0_c = load_t<NT>(g, a.sm[SM_GR] + m0); t0_c = load_t<NT>(g, a.sm[SM_TT] + m0);
      ttgpr1_t = load_t<NT>(g, a.sm[SM_TTGPR] + m1);
    }
    MOMR_STAMP(6);  // the three late operand loads
    const Mat<NT> Q_t = TNacc<NT>(g, a_c, ttgp1_t, TN<NT>(g, gr0_c, bn_t));  // iet(new) G r[n0] + tG ier
    Mat<NT> an_t = add<NT>(a_t, TNacc<NT>(g, t0_c, Q_t, TN<NT>(g, V_c, ttgpr1_t)));
    MOMR_STAMP(7);  // ier: 4 products
    if (fuseD) {  // apply_D_matrix_IE!, corrected indexing (D2)
      if (n > 1) map_t<NT>(g, an_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
      Mat<NT> apm = an_t, bmm = bn_t;
      if (n > 1) {
        map_t<NT>(g, apm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
        map_t<NT>(g, bmm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
      }
      if (!a.derive_pm) {
        store_t<NT>(g, a.ie_a[R_PM] + o4, apm);
        store_t<NT>(g, a.ie_a[T_MM] + o4, bmm);
      }
    }
    store_t<NT>(g, a.ie_a[R_MP] + o4, an_t);
    store_t<NT>(g, a.ie_a[T_PP] + o4, bn_t);
    MOMR_STAMP(8);  // D signs and the stores of the two (four) operator blocks
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: the one-tile doubling pair kernel with its operands PREFETCHED INTO LDS by direct global -> LDS loads
// (global_load_lds_dwordx4, gfx950), -DMOMR_LDSPF=0 restores the body above for NT = 1.
//
// What bounds the body above (tools/phase_stamps_rrs.py, profiles/r05_C5_ab.txt): a wave spends ~50 % of a pair waiting for
// memory -- the first eight operand loads 20 %, the three late ones 9 %, the four vector loads inside the source block, the
// stores -- at 3 waves per SIMD with ~1.1 us per round trip, and 14 % in the loop head (64-bit div / mod of the pair index and
// the dependent load of the offset).  A register prefetch of the next pair (r4, MOMR_PIPELINE) costs a wave per SIMD and was
// slower.  Here:
//   * a work item is (n1, a chunk of dn_chunk consecutive Raman offsets): the three n1-side tiles r[n1], (t G)[n1]^T,
//     (t G r)[n1]^T and expk[n1] are loaded ONCE per item and stay in registers; no division in the pair loop;
//   * while pair (n1, dn) is computed, the pair-specific and n0-side operands of the NEXT on-grid pair of the chunk -- ier-+,
//     iet++ (2 x 2 KB), r[n0]^T, (G t)[n0] (2 x 2 KB), ieJ0+-, j1-[n0], j0+[n0], tmp1, tmp2 (6 x 128 B) -- travel from global
//     memory straight into the wave's LDS slots: no VGPRs are held for them, the wave count per SIMD stays at 3;
//   * the two late operands (G r)[n0], t[n0]^T are requested at the top of the pair, before the prefetch, and are consumed in
//     its second half;
//   * the stores of a pair are issued at the top of the NEXT pair (results carried in registers), so that the one
//     s_waitcnt vmcnt(0) of a pair -- "my prefetch has landed" -- finds them a whole pair old.
// Pairs whose source point is off the grid are handled in line, as above.  Semantics identical to dbl_pair_body<1, ..>: every
// block is still read and written exactly once by exactly one wave (tests/test_gpu_rrs.py).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MOMR_LDSPF
#define MOMR_LDSPF 1
#endif
constexpr int kPfTile = 256;                              // doubles of a 16 x 16 block
constexpr int kPfDoubles = 4 * kPfTile + 6 * 16;          // ier-+, iet++, r[n0]^T, (G t)[n0] + six vectors
constexpr size_t kPfBytes = (size_t)kPfDoubles * 8;       // 8 960 B per wave
typedef __attribute__((address_space(1))) const void momr_gptr;
typedef __attribute__((address_space(3))) void momr_lptr;

// 2 KB block at `src` -> LDS at `dst` (wave-uniform), bit for bit: lane l moves bytes [16 l, 16 l + 16) of each KB
__device__ __forceinline__ void pf_tile(const double *src, double *dst, int lane) {
  __builtin_amdgcn_global_load_lds((momr_gptr *)(src + 2 * lane), (momr_lptr *)dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((momr_gptr *)(src + 128 + 2 * lane), (momr_lptr *)(dst + 128), 16, 0, 0);
}
__device__ __forceinline__ Mat<1> lds_tile(const Geo &g, const double *t) {  // the _t form of the block, as load_t<1>
  Mat<1> X;
#pragma unroll
  for (int r = 0; r < 4; ++r) X.t[0][0][r] = t[g.lr + 16 * (g.lq + 4 * r)];
  return X;
}

template <bool FUSE, int MODE>
__device__ __forceinline__ void dbl_pair_body1(const KArgs &a) {
  constexpr int NT = 1;
  constexpr bool STRICT = (MODE == 2), fuseD = (MODE == 1);
#ifdef MOMR_DIAG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) {
    unsigned long long n__;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");
    momr_diag_last = n__;
  }
#endif
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int n = a.nS, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t NN = 256, VS = 16;  // one-tile images: P = 16
  double *pf = reinterpret_cast<double *>(rrs_smem + (size_t)kWavesPerBlock * slice_bytes<NT>()) + (size_t)wave * kPfDoubles;
  double *pfA = pf, *pfB = pf + kPfTile, *pfR0 = pf + 2 * kPfTile, *pfGT = pf + 3 * kPfTile, *pfV = pf + 4 * kPfTile;
  const int span = a.n1_hi - a.n1_lo, CH = a.dn_chunk, nch = (a.nR + CH - 1) / CH;
  const unsigned items = (unsigned)span * (unsigned)nch, istride = gridDim.x * kWavesPerBlock;
  const bool sgnD = fuseD && n > 1;

  for (unsigned w = blockIdx.x * kWavesPerBlock + wave; w < items; w += istride) {
    const int ch = (int)(w / (unsigned)span), n1 = a.n1_lo + (int)(w - (unsigned)ch * (unsigned)span);
    const int dn_lo = ch * CH, dn_hi = (dn_lo + CH < a.nR) ? dn_lo + CH : a.nR;
    const size_t m1 = NN * n1;
    // the n1 side: once per item
    const Mat<NT> r1_t = load_t<NT>(g, a.a_cur[R_MP] + m1), ttgp1_t = load_t<NT>(g, a.sm[SM_TTGP] + m1);
    const Mat<NT> ttgpr1_t = load_t<NT>(g, a.sm[SM_TTGPR] + m1);
    const double e1 = a.expk_cur[n1];

    // The chunk's offsets live in the lanes of one register (lane k: offset dn_lo + k; dn_chunk <= 64) and its on-grid pairs in
    // a 64-bit scalar mask: the pair loop reads neither memory nor a divider for its bookkeeping
    const int cnt = dn_hi - dn_lo;
    const int off_l = (lane < cnt) ? a.off[dn_lo + lane] : 0;
    const unsigned long long gmask = __builtin_amdgcn_ballot_w64(lane < cnt && n1 + off_l >= 0 && n1 + off_l < a.S);
    auto off_of = [&](int dn) { return __builtin_amdgcn_readlane(off_l, dn - dn_lo); };
    // prefetch of pair (n1, dn) into the wave's LDS slots; dn on the grid
    auto prefetch = [&](int dn) {
      const int n0 = n1 + off_of(dn);
      const size_t u = (size_t)n1 + (size_t)a.S * dn, m0 = NN * n0, v0 = VS * n0;
      if (!FUSE) {
        pf_tile(a.ie_a[R_MP] + NN * u, pfA, lane);
        pf_tile(a.ie_a[T_PP] + NN * u, pfB, lane);
      }
      pf_tile(a.sm[SM_RT] + m0, pfR0, lane);
      pf_tile(a.sm[SM_GT] + m0, pfGT, lane);
      // six vectors of 16 doubles in ONE instruction: lane l moves 16 bytes of vector l / 8
      const double *jp0 = STRICT ? a.jpseq + v0 + VS * a.S * dn : a.a_cur[J0P] + v0;
      const int q = lane >> 3;
      const double *vp = (q == 0) ? a.ie_a[J0P] + VS * u : (q == 1) ? a.ie_a[J0M] + VS * u : (q == 2) ? a.sv[SV_J1M] + v0
                       : (q == 3) ? jp0 : (q == 4) ? a.sv[SV_TMP1] + v0 : a.sv[SV_TMP2] + v0;
      if (lane < 48 && !(FUSE && q < 2))
        __builtin_amdgcn_global_load_lds((momr_gptr *)(vp + 2 * (lane & 7)), (momr_lptr *)pfV, 16, 0, 0);
    };
    auto next_on_grid = [&](int dn) {  // first on-grid offset of the chunk after dn (dn_hi if none)
      const int k = dn + 1 - dn_lo;    // 0 .. cnt
      const unsigned long long rest = (k < 64) ? (gmask >> k) : 0ull;
      return rest ? dn + 1 + (int)__builtin_ctzll(rest) : dn_hi;
    };

    // results of the previous on-grid pair, stored at the top of the next one
    Mat<NT> an_d, bn_d;
    CV<NT> Jpn_d, Jmn_d;
    size_t o4_d = 0, o3_d = 0;
    bool have_d = false;
    auto flush = [&]() {
      if (!have_d) return;
      if (fuseD) {  // apply_D_matrix_IE!, corrected indexing (D2): the row signs are already in an_d
        if (!a.derive_pm) {
          Mat<NT> apm = an_d, bmm = bn_d;
          if (n > 1) {
            map_t<NT>(g, apm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
            map_t<NT>(g, bmm, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
          }
          store_t<NT>(g, a.ie_a[R_PM] + o4_d, apm);
          store_t<NT>(g, a.ie_a[T_MM] + o4_d, bmm);
        }
      }
      store_t<NT>(g, a.ie_a[R_MP] + o4_d, an_d);
      store_t<NT>(g, a.ie_a[T_PP] + o4_d, bn_d);
      storeC<NT>(g, a.ie_a[J0P] + o3_d, Jpn_d);
      storeC<NT>(g, a.ie_a[J0M] + o3_d, Jmn_d);
      have_d = false;
    };

    int dn_pf = dn_lo - 1;
    dn_pf = next_on_grid(dn_pf);
    if (dn_pf < dn_hi) prefetch(dn_pf);

    // (the variants without off-grid work -- not the fused first step, not the last corrected step -- walk the mask's set bits)
    constexpr bool OFFGRID_WORK = FUSE || fuseD;
    for (int dn = OFFGRID_WORK ? dn_lo : next_on_grid(dn_lo - 1); dn < dn_hi; dn = OFFGRID_WORK ? dn + 1 : dn_pf) {
      const int n0 = n1 + off_of(dn);
      const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
      if (OFFGRID_WORK && !((gmask >> (dn - dn_lo)) & 1ull)) {  // get_n0_n1 (inelastic_helper.jl:13-21): no update off the grid ...
        if (FUSE) {               // ... but the deferred elemental writes zeros there and multiplies ieJ0- by D (:378-380)
          store_t<NT>(g, a.ie_a[R_MP] + o4, zeros<NT>());
          store_t<NT>(g, a.ie_a[T_PP] + o4, zeros<NT>());
          CV<NT> Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
          Jm.c[0] = a.D[g.col(0) % n] * Jm.c[0];
          storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
        }
        if (fuseD) {              // ... and the corrected D kernels visit every (n, dn)
          Mat<NT> an_t = FUSE ? zeros<NT>() : load_t<NT>(g, a.ie_a[R_MP] + o4);
          Mat<NT> bn_t = FUSE ? zeros<NT>() : load_t<NT>(g, a.ie_a[T_PP] + o4);
          if (n > 1) {
            map_t<NT>(g, an_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
            store_t<NT>(g, a.ie_a[R_MP] + o4, an_t);
            CV<NT> Jm = loadC<NT>(g, a.ie_a[J0M] + o3);
            if (scomp(g.col(0), n, a.strict_idx) > 2) Jm.c[0] = -Jm.c[0];
            storeC<NT>(g, a.ie_a[J0M] + o3, Jm);
            map_t<NT>(g, an_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
            map_t<NT>(g, bn_t, [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; });
          }
          if (!a.derive_pm) {
            store_t<NT>(g, a.ie_a[R_PM] + o4, an_t);
            store_t<NT>(g, a.ie_a[T_MM] + o4, bn_t);
          }
        }
        continue;
      }
      const size_t m0 = NN * n0;
      MOMR_STAMP_NW(10);  // loop head, off-grid pairs, item set-up (n1-side loads)
      // ---- this pair's operands have landed in LDS (requested one pair ago); everything older has completed too
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MOMR_STAMP_NW(11);  // wait for the prefetch (and the previous pair's stores)
      Mat<NT> a_t, b_t;
      CV<NT> Jp, Jm;
      if (!FUSE) {
        a_t = lds_tile(g, pfA); b_t = lds_tile(g, pfB);
        Jp.c[0] = pfV[g.lr]; Jm.c[0] = pfV[16 + g.lr];
      }
      const Mat<NT> r0_c = lds_tile(g, pfR0), gt0_c = lds_tile(g, pfGT);
      Vec<NT> j1mR, jp0R, tm1, tm2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        j1mR.t[0][r] = pfV[32 + g.lq + 4 * r];
        jp0R.t[0][r] = pfV[48 + g.lq + 4 * r];
        tm1.t[0][r] = pfV[64 + g.lq + 4 * r];
        tm2.t[0][r] = pfV[80 + g.lq + 4 * r];
      }
      // the two late operands first (consumed in the second half of the pair), then the next pair's prefetch, then the
      // previous pair's stores
      const Mat<NT> gr0_c = load_t<NT>(g, a.sm[SM_GR] + m0), t0_c = load_t<NT>(g, a.sm[SM_TT] + m0);
      Mat<NT> bmm_strict;
      if (STRICT) bmm_strict = load_t<NT>(g, a.ie_a[T_MM] + o4);  // D5: iet-- as the array holds it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slots are free again
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      dn_pf = next_on_grid(dn);
      if (dn_pf < dn_hi) prefetch(dn_pf);
      flush();
      MOMR_STAMP_NW(12);  // LDS reads, late loads / next prefetch / deferred stores issued
      if (FUSE) ie_elem_tile<NT>(g, a, n1, dn, n0, a_t, b_t, Jp, Jm);

      const Mat<NT> a_c = transpose<NT>(g, a_t);
      // X = ier r0 + r1 ier
 const Mat<NT> X_t = TNacc<NT>(g, a_c, r1_t, TN<NT>(g, r0_c, a_t));
      MOMR_STAMP_NW(13);  // (fused elemental), transpose ier, X
      // ---- sources                                                                                            :61-89
      CV<NT> Jpn, Jmn;
      {
        const CV<NT> J1p = cscale<NT>(Jp, e1), J1m = cscale<NT>(Jm, e1);                         // ieJ1+, ieJ1-   :52-56
        const CV<NT> a_j1m = mv_t<NT>(g, a_t, j1mR);                                             // ier j1-[n0]
        const CV<NT> a_jp = mv_t<NT>(g, a_t, jp0R);                                              // ier j0+[n0]
        const CV<NT> X1 = mv_t<NT>(g, X_t, tm1), X2 = mv_t<NT>(g, X_t, tm2);
        const CV<NT> b1 = mv_t<NT>(g, b_t, tm1);                                                 // iet++ tmp1
        const CV<NT> b2 = STRICT ? mv_t<NT>(g, bmm_strict, tm2) : mv_t<NT>(g, b_t, tm2);
        const CV<NT> uu = cadd<NT>(cadd<NT>(Jp, mv_t<NT>(g, r1_t, c2r<NT>(g, J1m))), cadd<NT>(a_j1m, X1));
        Jpn = cadd<NT>(cadd<NT>(J1p, mv_t<NT>(g, ttgp1_t, c2r<NT>(g, uu))), b1);                 // new ieJ0+
        const CV<NT> rv2 = mv_t<NT>(g, r1_t, c2r<NT>(g, Jpn));                                   // r1 ieJ0+(new)
        const CV<NT> u2 = cadd<NT>(cadd<NT>(J1m, rv2), cadd<NT>(a_jp, X2));
        Jmn = cadd<NT>(cadd<NT>(Jm, mv_t<NT>(g, ttgp1_t, c2r<NT>(g, u2))), b2);                  // new ieJ0-
        if (sgnD && scomp(g.col(0), n, a.strict_idx) > 2) Jmn.c[0] = -Jmn.c[0];
      }
      MOMR_STAMP_NW(14);  // source vectors
      // ---- operators                                                                                          :98-125
      const Mat<NT> b_c = transpose<NT>(g, b_t);
      const Mat<NT> Y_c = TN<NT>(g, X_t, gt0_c);                                // X G t[n0]
      const Mat<NT> W_c = add<NT>(b_c, Y_c);
      Mat<NT> bn_t = TNacc<NT>(g, gt0_c, b_t, TN<NT>(g, W_c, ttgp1_t));         // tG (iet + Y) + iet G t[n0]
      const Mat<NT> bn_c = transpose<NT>(g, bn_t);
      const Mat<NT> V_c = add<NT>(bn_c, Y_c);
      MOMR_STAMP_NW(15);  // iet: 2 transposes, 3 products
      const Mat<NT> Q_t = TNacc<NT>(g, a_c, ttgp1_t, TN<NT>(g, gr0_c, bn_t));  // iet(new) G r[n0] + tG ier
      Mat<NT> an_t = add<NT>(a_t, TNacc<NT>(g, t0_c, Q_t, TN<NT>(g, V_c, ttgpr1_t)));
      if (sgnD) map_t<NT>(g, an_t, [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; });
      an_d = an_t; bn_d = bn_t; Jpn_d = Jpn; Jmn_d = Jmn; o4_d = o4; o3_d = o3; have_d = true;
      MOMR_STAMP_NW(16);  // ier: 4 products (waits for the two late operands)
    }
    flush();
  }
}

// The kernel images: the one-tile form at MOMR_WPE waves per SIMD (256 registers), the 2 x 2-tile form at MOMR_WPE2 (its ~320
// live registers spill to scratch at 256; at one wave per SIMD the register file holds them)
template <bool FUSE, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR k_dbl_pair1(KArgs a) {
  if constexpr (MOMR_LDSPF != 0) dbl_pair_body1<FUSE, MODE>(a);
  else dbl_pair_body<1, FUSE, MODE>(a);
}
template <bool FUSE, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_dbl_pair2(KArgs a) { dbl_pair_body<2, FUSE, MODE>(a); }
// 32 < N <= 64 (r4): the same bodies on 3 x 3 and 4 x 4 tiles.  Nine / sixteen tiles per operator do not fit the register file
// of a wave (a pair holds about ten operators), so these images live on scratch: they exist so that rt_run(::RRS) has no
// size limit below N = 64, not for speed (tools/bench_rrs_nt2.py has their timings)
#ifdef MOMR_BIG_TU
template <bool FUSE, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_dbl_pair3(KArgs a) { dbl_pair_body<3, FUSE, MODE>(a); }
template <bool FUSE, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_dbl_pair4(KArgs a) { dbl_pair_body<4, FUSE, MODE>(a); }
#endif

// apply_D_IE_RRS! / apply_D_SFI_IE_RRS! as written (doubling_inelastic.jl:291-311, :345-357), strict position: the work
// item (n, dn) addresses the RAMAN axis with n0 = n + i_l1l0[dn] (1-based) when 1 <= n0 <= nRaman.  One thread per
// (i, n) walks dn in ascending order (single-thread column-major semantics of the overlapping reads/writes of the SFI kernel).
__global__ void k_strict_D(KArgs a) {
  const int N = a.N, n = a.nS, P = a.P;
  const size_t NN = (size_t)P * P;  // block stride (padded pitch P)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int span = a.n1_hi - a.n1_lo;
  if (e >= N * span) return;
  const int i = e % N, nl = a.n1_lo + e / N;
  const int ng1 = a.n_glob0 + nl + 1;  // 1-based global spectral index
  const int ci = scomp(i, n, a.strict_idx);
  for (int dn = 0; dn < a.nR; ++dn) {
    const int k1 = ng1 + a.off[dn];
    if (k1 < 1 || k1 > a.nR) continue;
    const size_t ub = (size_t)nl + (size_t)a.S * (k1 - 1);
    for (int j = 0; j < N; ++j) {
      const size_t o = NN * ub + i + (size_t)P * j;
      double v = a.ie_a[R_MP][o];
      if (ci > 2) { v = -v; a.ie_a[R_MP][o] = v; }
      const double s = dsgn(ci, scomp(j, n, a.strict_idx));
      a.ie_a[R_PM][o] = s * v;
      a.ie_a[T_MM][o] = s * a.ie_a[T_PP][o];
    }
  }
  if (ci > 2)
    for (int dn = 0; dn < a.nR; ++dn) {
      const int k1 = ng1 + a.off[dn];
      if (k1 < 1 || k1 > a.nR) continue;
      a.ie_a[J0M][i + (size_t)P * ((size_t)nl + (size_t)a.S * (k1 - 1))] = -a.ie_a[J0M][i + (size_t)P * ((size_t)nl + (size_t)a.S * dn)];
    }
}
// ier+- = sgn (.) ier-+, iet-- = sgn (.) iet++ for the whole arrays (the derived form written out: downloads, strict reads)
__global__ void k_materialise_pm(KArgs a) {
  const size_t NN = (size_t)a.P * a.P, cnt = NN * a.S * a.nR;  // every element of the padded blocks (padding: s x 0)
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e % a.P), j = (int)((e / a.P) % a.P);
    const double s = a.nS > 1 ? dsgn(scomp(i, a.nS, a.strict_idx), scomp(j, a.nS, a.strict_idx)) : 1.0;
    a.ie_a[R_PM][e] = s * a.ie_a[R_MP][e];
    a.ie_a[T_MM][e] = s * a.ie_a[T_PP][e];
  }
}
// n_stokes == 1: ier+- = ier-+, iet-- = iet++ for the whole arrays (doubling_inelastic.jl:411-414)
__global__ void k_copy2(const double *s0, double *d0, const double *s1, double *d1, size_t count) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
    d0[e] = s0[e];
    d1[e] = s1[e];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// interaction, POINT kernel: the elastic adding equations (interaction.jl:8-117 arithmetic inside
// interaction_inelastic.jl) into the other half of the composite buffer + the operands of the pair kernel.
// x[] = the added layer of this interaction (atmospheric or surface).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_int_point(KArgs a, int iface) {
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int wave = threadIdx.x >> 6;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;  // block strides (padded pitch)
  int bad = 0;
  for (int pt = blockIdx.x * kWavesPerBlock + wave; pt < a.S; pt += gridDim.x * kWavesPerBlock) {
    const size_t om = NN * pt, ov = VS * pt;
    const Mat<NT> r_t = load_t<NT>(g, a.x[R_MP] + om), tpp_t = load_t<NT>(g, a.x[T_PP] + om);
    const Mat<NT> tmm_t = load_t<NT>(g, a.x[T_MM] + om);
    const Mat<NT> Rpm_t = load_t<NT>(g, a.c_cur[C_R_PM] + om), Tpp_t = load_t<NT>(g, a.c_cur[C_T_PP] + om);
    const Mat<NT> Tmm_t = load_t<NT>(g, a.c_cur[C_T_MM] + om), Rmp_t = load_t<NT>(g, a.c_cur[C_R_MP] + om);
    const Vec<NT> ja = loadv2<NT>(g, a.x[J0P] + ov, a.x[J0M] + ov);                    // (j0+ | j0-) added
    const Vec<NT> Jc = loadv2<NT>(g, a.c_cur[C_J0P] + ov, a.c_cur[C_J0M] + ov);        // (J0+ | J0-) composite
    const Mat<NT> tmm_c = transpose<NT>(g, tmm_t), Tpp_c = transpose<NT>(g, Tpp_t);
    Mat<NT> Rmp_n = Rmp_t, Rpm_n = Rpm_t, Tpp_n, Tmm_n;
    Vec<NT> Jn;  // (J0+ | J0-) new
    if (iface == 3) {
      const Mat<NT> r_c = transpose<NT>(g, r_t), Rpm_c = transpose<NT>(g, Rpm_t);
      const Mat<NT> G1_c = inv_one_minus<NT>(g, TN<NT>(g, r_t, Rpm_c), &bad);          // (I - r R+-)^-1           :244
      const Mat<NT> G1_t = transpose<NT>(g, G1_c);
      const Mat<NT> T01_t = TN<NT>(g, G1_c, Tmm_t);                                    // (T-- G1)^T               :247
      const Mat<NT> rT_c = TN<NT>(g, r_t, Tpp_c);                                      // r T++
      store_t<NT>(g, a.sm[SI_T01] + om, T01_t);
      store_t<NT>(g, a.sm[SI_RPM] + om, Rpm_c);
      store_t<NT>(g, a.sm[SI_TPP] + om, Tpp_c);
      store_t<NT>(g, a.sm[SI_R] + om, r_c);
      store_t<NT>(g, a.sm[SI_TMM] + om, tmm_c);
      store_t<NT>(g, a.sm[SI_G1RT] + om, TN<NT>(g, G1_t, rT_c));                       // G1 r T++
      store_t<NT>(g, a.sm[SI_G1T] + om, TN<NT>(g, G1_t, tmm_c));                       // G1 t--
      const Vec<NT> rJ = TNv<NT>(g, r_t, Jc);                                          // col 0: r J0+
      const Vec<NT> s1 = vadd<NT>(rJ, swap01<NT>(ja));                                 // col 0: r J0+ + j0-
      storev<NT>(g, a.sv[SVI_G1V] + ov, TNv<NT>(g, G1_t, s1), 0);                      // G1 (j0- + r J0+)
      const Vec<NT> dJm = TNv<NT>(g, T01_t, s1);                                       // col 0                    :267
      Rmp_n = add<NT>(Rmp_t, TN<NT>(g, rT_c, T01_t));                                  //                          :288
      Tmm_n = TN<NT>(g, tmm_c, T01_t);                                                 //                          :290
      const Mat<NT> G2_c = inv_one_minus<NT>(g, TN<NT>(g, Rpm_t, r_c), &bad);          // (I - R+- r)^-1           :295
      const Mat<NT> G2_t = transpose<NT>(g, G2_c);
      const Mat<NT> T21_t = TN<NT>(g, G2_c, tpp_t);                                    //                          :297
      const Mat<NT> Rt_c = TN<NT>(g, Rpm_t, tmm_c);                                    // R+- t--
      store_t<NT>(g, a.sm[SI_T21] + om, T21_t);
      store_t<NT>(g, a.sm[SI_G2T] + om, TN<NT>(g, G2_t, Tpp_c));
      store_t<NT>(g, a.sm[SI_G2RT] + om, TN<NT>(g, G2_t, Rt_c));
      const Vec<NT> Rj = swap01<NT>(TNv<NT>(g, Rpm_t, ja));                            // col 0: R+- j0-
      const Vec<NT> s2 = vadd<NT>(Jc, Rj);                                             // col 0: J0+ + R+- j0-
      storev<NT>(g, a.sv[SVI_G2V] + ov, TNv<NT>(g, G2_t, s2), 0);
      const Vec<NT> dJp = TNv<NT>(g, T21_t, s2);                                       //                          :315
      const Vec<NT> dJm1 = swap01<NT>(dJm);  // column 1: J0- + dJm (dJm sits in column 0)
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) Jn.t[ta] = (g.lr == 0) ? (ja.t[ta] + dJp.t[ta]) : (Jc.t[ta] + dJm1.t[ta]);
      Tpp_n = TN<NT>(g, Tpp_c, T21_t);                                                 //                          :338
      Rpm_n = add<NT>(load_t<NT>(g, a.x[R_PM] + om), TN<NT>(g, Rt_c, T21_t));          //                          :340
    } else {
      // 00 / 01 / 10 (D4): operands for the pair kernel are the plain layers in the right orientation
      store_t<NT>(g, a.sm[SI_TPP] + om, Tpp_c);
      store_t<NT>(g, a.sm[SI_TMM] + om, tmm_c);
      const Mat<NT> Tmm_c = transpose<NT>(g, Tmm_t);
      if (iface == 0) {                                                                //                          :8-22
        const Vec<NT> tJ = TNv<NT>(g, tpp_t, Jc);                                      // col 0: t++ J0+
        const Vec<NT> Tj = TNv<NT>(g, Tmm_t, ja);                                      // col 1: T-- j0-
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) Jn.t[ta] = (g.lr == 0) ? (ja.t[ta] + tJ.t[ta]) : (Jc.t[ta] + Tj.t[ta]);
        Tmm_n = TN<NT>(g, Tmm_c, tmm_t);                                               // t-- T--
        Tpp_n = TN<NT>(g, Tpp_c, tpp_t);                                               // t++ T++
      } else if (iface == 1) {                                                         //                          :28-76
        const Vec<NT> rJ = TNv<NT>(g, r_t, Jc);                                        // col 0: r J0+
        const Vec<NT> s1 = vadd<NT>(rJ, swap01<NT>(ja));                               // col 0: r J0+ + j0-
        const Vec<NT> dJm = swap01<NT>(TNv<NT>(g, Tmm_t, s1));                         // col 1
        const Vec<NT> tJ = TNv<NT>(g, tpp_t, Jc);
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) Jn.t[ta] = (g.lr == 0) ? (ja.t[ta] + tJ.t[ta]) : (Jc.t[ta] + dJm.t[ta]);
        const Mat<NT> rT_c = TN<NT>(g, r_t, Tpp_c);
        Rmp_n = TN<NT>(g, rT_c, Tmm_t);                                                // T-- r T++
        Rpm_n = load_t<NT>(g, a.x[R_PM] + om);
        Tpp_n = TN<NT>(g, Tpp_c, tpp_t);                                               // t++ T++
        Tmm_n = TN<NT>(g, tmm_c, Tmm_t);                                               // T-- t--
      } else {                                                                         // 10                       :139-180
        const Vec<NT> Rj = swap01<NT>(TNv<NT>(g, Rpm_t, ja));                          // col 0: R+- j0-
        const Vec<NT> dJp = TNv<NT>(g, tpp_t, vadd<NT>(Jc, Rj));                       // col 0
        const Vec<NT> Tj = TNv<NT>(g, Tmm_t, ja);                                      // col 1: T-- j0-
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) Jn.t[ta] = (g.lr == 0) ? (ja.t[ta] + dJp.t[ta]) : (Jc.t[ta] + Tj.t[ta]);
        Tpp_n = TN<NT>(g, Tpp_c, tpp_t);
        Tmm_n = TN<NT>(g, tmm_c, Tmm_t);
        const Mat<NT> Rt_c = TN<NT>(g, Rpm_t, tmm_c);
        Rpm_n = TN<NT>(g, Rt_c, tpp_t);                                                // t++ R+- t--
      }
    }
    store_t<NT>(g, a.c_nxt[C_R_MP] + om, Rmp_n);
    store_t<NT>(g, a.c_nxt[C_R_PM] + om, Rpm_n);
    store_t<NT>(g, a.c_nxt[C_T_PP] + om, Tpp_n);
    store_t<NT>(g, a.c_nxt[C_T_MM] + om, Tmm_n);
    storev<NT>(g, a.c_nxt[C_J0P] + ov, Jn, 0);
    storev<NT>(g, a.c_nxt[C_J0M] + ov, Jn, 1);
  }
  if (bad && g.lr == 0 && g.lq == 0) atomicMax(a.info, bad);
}

// ---------------------------------------------------------------------------------------------------------------------
// interaction, PAIR kernel.  SURF: the added layer is the surface (all its ie* arrays are zeros and never read).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, bool SURF, bool DERIVE>
__device__ __forceinline__ void int_pair_body(const KArgs &a, int iface) {
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int wave = threadIdx.x >> 6;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;  // block strides (padded pitch)
  const size_t npairs = (size_t)(a.n1_hi - a.n1_lo) * a.nR;
  // software pipeline of the interface-11 case (one-tile images), as in k_dbl_pair: the first operand tiles of pair
  // p + stride are requested before pair p is computed
  constexpr bool PIPE = (NT == 1) && MOMR_PIPELINE;
  struct Pre {
    Mat<NT> a_raw, bm_raw, E_t, C_t;   // the pair's own blocks of the 4-D arrays (HBM streams); the per-point operands are L2-hot
    CV<NT> Jap, Jam, Jcp, Jcm;
    bool have;
  } nx;
  nx.have = false;
  const size_t stride = (size_t)gridDim.x * kWavesPerBlock, span = (size_t)(a.n1_hi - a.n1_lo);
  auto raw_a = [&](size_t q4) { return SURF ? zeros<NT>() : load_t<NT>(g, a.ie_a[R_MP] + q4); };
  auto raw_bm = [&](size_t q4) { return SURF ? zeros<NT>() : load_t<NT>(g, a.ie_a[DERIVE ? T_PP : T_MM] + q4); };
  auto prefetch = [&](size_t q) {
    nx.have = false;
    if (!PIPE || iface != 3 || q >= npairs) return;
    const int qn1 = a.n1_lo + (int)(q % span), qdn = (int)(q / span);
    const int qn0 = qn1 + a.off[qdn];
    if (qn0 < 0 || qn0 >= a.S) return;
    const size_t qu = (size_t)qn1 + (size_t)a.S * qdn, q4 = NN * qu, q3 = VS * qu;
    nx.Jap = SURF ? czeros<NT>() : loadC<NT>(g, a.ie_a[J0P] + q3);
    nx.Jam = SURF ? czeros<NT>() : loadC<NT>(g, a.ie_a[J0M] + q3);
    nx.Jcp = loadC<NT>(g, a.ie_c[C_J0P] + q3);
    nx.Jcm = loadC<NT>(g, a.ie_c[C_J0M] + q3);
    nx.a_raw = raw_a(q4);
    nx.bm_raw = raw_bm(q4);
    nx.E_t = load_t<NT>(g, a.ie_c[C_R_PM] + q4);
    nx.C_t = load_t<NT>(g, a.ie_c[C_T_PP] + q4);
    nx.have = true;
  };
  prefetch((size_t)blockIdx.x * kWavesPerBlock + wave);
  for (size_t p = (size_t)blockIdx.x * kWavesPerBlock + wave; p < npairs; p += stride) {
    const int n1 = a.n1_lo + (int)(p % span), dn = (int)(p / span);
    const int n0 = n1 + a.off[dn];
    if (n0 < 0 || n0 >= a.S) {
      if (PIPE) prefetch(p + stride);
      continue;
    }
    const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
    const size_t m1 = NN * n1, m0 = NN * n0, v0 = VS * n0;
    auto flip_pm = [&](Mat<NT> X) {  // ier+- = sgn (.) ier-+, iet-- = sgn (.) iet++ (corrected D2)
      if (a.nS > 1)
        map_t<NT>(g, X, [&](int i, int j, double v) { return dsgn(scomp(i, a.nS, a.strict_idx), scomp(j, a.nS, a.strict_idx)) * v; });
      return X;
    };
    auto ldA = [&](int which) {
      if (SURF) return zeros<NT>();
      if (DERIVE && (which == T_MM || which == R_PM)) return flip_pm(load_t<NT>(g, a.ie_a[which == T_MM ? T_PP : R_MP] + o4));
      return load_t<NT>(g, a.ie_a[which] + o4);
    };
    const bool pre = PIPE && nx.have;   // (nx is only ever filled for iface == 3)
    const CV<NT> Jap = pre ? nx.Jap : (SURF ? czeros<NT>() : loadC<NT>(g, a.ie_a[J0P] + o3));
    const CV<NT> Jam = pre ? nx.Jam : (SURF ? czeros<NT>() : loadC<NT>(g, a.ie_a[J0M] + o3));
    const CV<NT> Jcp = pre ? nx.Jcp : loadC<NT>(g, a.ie_c[C_J0P] + o3);   // ieJ0+- added / composite
    const CV<NT> Jcm = pre ? nx.Jcm : loadC<NT>(g, a.ie_c[C_J0M] + o3);
    if (iface == 3) {
      const Mat<NT> a_t = pre ? nx.a_raw : raw_a(o4);
      Mat<NT> bm_t = pre ? nx.bm_raw : raw_bm(o4);
      if (DERIVE && !SURF) bm_t = flip_pm(bm_t);
      const Mat<NT> E_t = pre ? nx.E_t : load_t<NT>(g, a.ie_c[C_R_PM] + o4), C_t = pre ? nx.C_t : load_t<NT>(g, a.ie_c[C_T_PP] + o4);
      const Mat<NT> T01_t = load_t<NT>(g, a.sm[SI_T01] + m1), r1_t = load_t<NT>(g, a.x[R_MP] + m1);
      const Mat<NT> Rpm0_c = load_t<NT>(g, a.sm[SI_RPM] + m0), Tpp0_c = load_t<NT>(g, a.sm[SI_TPP] + m0);
      if (PIPE) prefetch(p + stride);
      const Mat<NT> a_c = transpose<NT>(g, a_t), bm_c = transpose<NT>(g, bm_t);
      const Mat<NT> E_c = transpose<NT>(g, E_t), C_c = transpose<NT>(g, C_t);
      // A = T01 (ier R+-[n0] + r ieR+-) + ieT--                                                                  :252-262
      const Mat<NT> M1_c = TNacc<NT>(g, r1_t, E_c, TN<NT>(g, a_t, Rpm0_c));
      const Mat<NT> A_t = TNacc<NT>(g, M1_c, T01_t, load_t<NT>(g, a.ie_c[C_T_MM] + o4));
      // ieJ0- += T01 (ier J0+[n0] + r ieJ0+ + ieJ0-(added)) + A G1 (j0-[n0] + r[n0] J0+[n0])                      :251-264
      {
        const CV<NT> v1 = mv_t<NT>(g, a_t, loadR<NT>(g, a.c_cur[C_J0P] + v0));
        const CV<NT> v2 = mv_t<NT>(g, r1_t, loadR<NT>(g, a.ie_c[C_J0P] + o3));
        const CV<NT> uu = cadd<NT>(cadd<NT>(v1, v2), Jam);
        const CV<NT> w = cadd<NT>(mv_t<NT>(g, T01_t, c2r<NT>(g, uu)), mv_t<NT>(g, A_t, loadR<NT>(g, a.sv[SVI_G1V] + v0)));
        storeC<NT>(g, a.ie_c[C_J0M] + o3, cadd<NT>(Jcm, w));
      }
      // ieR-+ += T01 (ier T++[n0] + r ieT++) + A G1 r[n0] T++[n0];  ieT-- = T01 iet-- + A G1 t--[n0]               :271-284
      {
        const Mat<NT> N1_c = TNacc<NT>(g, r1_t, C_c, TN<NT>(g, a_t, Tpp0_c));
        Mat<NT> Rm_t = load_t<NT>(g, a.ie_c[C_R_MP] + o4);
        Rm_t = TNacc<NT>(g, N1_c, T01_t, Rm_t);
        Rm_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G1RT] + m0), A_t, Rm_t);
        store_t<NT>(g, a.ie_c[C_R_MP] + o4, Rm_t);
        const Mat<NT> F_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G1T] + m0), A_t, TN<NT>(g, bm_c, T01_t));
        store_t<NT>(g, a.ie_c[C_T_MM] + o4, F_t);
      }
      // B = T21 (ieR+- r[n0] + R+- ier) + iet++                                                                   :302-310
      const Mat<NT> T21_t = load_t<NT>(g, a.sm[SI_T21] + m1), Rpm1_t = load_t<NT>(g, a.c_cur[C_R_PM] + m1);
      const Mat<NT> M2_c = TNacc<NT>(g, Rpm1_t, a_c, TN<NT>(g, E_t, load_t<NT>(g, a.sm[SI_R] + m0)));
      const Mat<NT> B_t = TNacc<NT>(g, M2_c, T21_t, ldA(T_PP));
      // ieJ0+ = ieJ0+(added) + T21 (ieJ0+ + ieR+- j0-[n0] + R+- ieJ0-(added)) + B G2 (J0+[n0] + R+-[n0] j0-[n0])    :301-312
      {
        const CV<NT> v3 = mv_t<NT>(g, E_t, loadR<NT>(g, a.x[J0M] + v0));
        const CV<NT> v4 = SURF ? czeros<NT>() : mv_t<NT>(g, Rpm1_t, loadR<NT>(g, a.ie_a[J0M] + o3));
        const CV<NT> uu = cadd<NT>(cadd<NT>(Jcp, v3), v4);
        const CV<NT> w = cadd<NT>(mv_t<NT>(g, T21_t, c2r<NT>(g, uu)), mv_t<NT>(g, B_t, loadR<NT>(g, a.sv[SVI_G2V] + v0)));
        storeC<NT>(g, a.ie_c[C_J0P] + o3, cadd<NT>(Jap, w));
      }
      // ieT++ = T21 ieT++ + B G2 T++[n0];  ieR+- = ier+- + T21 (ieR+- t--[n0] + R+- iet--) + B G2 R+-[n0] t--[n0]   :320-334
      {
        const Mat<NT> Cn_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G2T] + m0), B_t, TN<NT>(g, C_c, T21_t));
        store_t<NT>(g, a.ie_c[C_T_PP] + o4, Cn_t);
        const Mat<NT> N2_c = TNacc<NT>(g, Rpm1_t, bm_c, TN<NT>(g, E_t, load_t<NT>(g, a.sm[SI_TMM] + m0)));
        Mat<NT> En_t = ldA(R_PM);
        En_t = TNacc<NT>(g, N2_c, T21_t, En_t);
        En_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G2RT] + m0), B_t, En_t);
        store_t<NT>(g, a.ie_c[C_R_PM] + o4, En_t);
      }
    } else if (iface == 1) {                                                           // 01, corrected (D4)       :40-75
      const Mat<NT> a_t = ldA(R_MP), b_t = ldA(T_PP), bm_t = ldA(T_MM);
      const Mat<NT> Tmm1_t = load_t<NT>(g, a.c_cur[C_T_MM] + m1), Tpp0_c = load_t<NT>(g, a.sm[SI_TPP] + m0);
      const Vec<NT> J0 = loadR<NT>(g, a.c_cur[C_J0P] + v0);
      const CV<NT> v1 = cadd<NT>(mv_t<NT>(g, a_t, J0), Jam);                           // ier J0+[n0] + ieJ0-(added)
      storeC<NT>(g, a.ie_c[C_J0M] + o3, mv_t<NT>(g, Tmm1_t, c2r<NT>(g, v1)));
      storeC<NT>(g, a.ie_c[C_J0P] + o3, cadd<NT>(Jap, mv_t<NT>(g, b_t, J0)));
      store_t<NT>(g, a.ie_c[C_R_MP] + o4, TN<NT>(g, TN<NT>(g, a_t, Tpp0_c), Tmm1_t));  // T-- ier T++[n0]
      store_t<NT>(g, a.ie_c[C_R_PM] + o4, ldA(R_PM));
      store_t<NT>(g, a.ie_c[C_T_PP] + o4, TN<NT>(g, Tpp0_c, b_t));                     // iet++ T++[n0]
      store_t<NT>(g, a.ie_c[C_T_MM] + o4, TN<NT>(g, transpose<NT>(g, bm_t), Tmm1_t)); // T-- iet--
    } else if (iface == 2) {                                                           // 10, corrected (D4)       :150-178
      const Mat<NT> E_t = load_t<NT>(g, a.ie_c[C_R_PM] + o4), C_t = load_t<NT>(g, a.ie_c[C_T_PP] + o4);
      const Mat<NT> F_t = load_t<NT>(g, a.ie_c[C_T_MM] + o4);
      const Mat<NT> tpp1_t = load_t<NT>(g, a.x[T_PP] + m1), tmm0_c = load_t<NT>(g, a.sm[SI_TMM] + m0);
      const Vec<NT> j0 = loadR<NT>(g, a.x[J0M] + v0);
      storeC<NT>(g, a.ie_c[C_J0P] + o3, mv_t<NT>(g, tpp1_t, c2r<NT>(g, cadd<NT>(Jcp, mv_t<NT>(g, E_t, j0)))));
      storeC<NT>(g, a.ie_c[C_J0M] + o3, cadd<NT>(Jcm, mv_t<NT>(g, F_t, j0)));
      store_t<NT>(g, a.ie_c[C_T_PP] + o4, TN<NT>(g, transpose<NT>(g, C_t), tpp1_t));   // t++ ieT++
      store_t<NT>(g, a.ie_c[C_T_MM] + o4, TN<NT>(g, tmm0_c, F_t));                     // ieT-- t--[n0]
      store_t<NT>(g, a.ie_c[C_R_PM] + o4, TN<NT>(g, TN<NT>(g, E_t, tmm0_c), tpp1_t));  // t++ ieR+- t--[n0]
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: ScatteringInterface_11 of the one-tile interaction pair kernel in the form of dbl_pair_body1 -- work items (n1, chunk of
// offsets) with the four n1-side tiles T01, r, T21, R+-[n1] resident in registers, the pair's first six operand tiles (ier-+,
// iet-- / iet++, ieR+-, ieT++ of the pair, R+-[n0], T++[n0]) and its eight source vectors prefetched into LDS while the
// previous pair is computed (one global_load_lds per 1 KB), the on-grid pairs of the chunk as a scalar bit mask, the last two
// operator stores deferred to the top of the next pair.  The ten late operand tiles are loaded where they are used, as in
// int_pair_body.  The other interface cases (and -DMOMR_LDSPF=0) run int_pair_body<1, ..>.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPfIntDoubles = 6 * kPfTile + 8 * 16;      // 13 312 B per wave
constexpr size_t kPfIntBytes = (size_t)kPfIntDoubles * 8;

template <bool SURF, bool DERIVE>
__device__ __forceinline__ void int_pair_body1(const KArgs &a) {
  constexpr int NT = 1;
  const Geo g = make_geo<NT>(a.N, rrs_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t NN = 256, VS = 16;
  double *pf = reinterpret_cast<double *>(rrs_smem + (size_t)kWavesPerBlock * slice_bytes<NT>()) + (size_t)wave * kPfIntDoubles;
  double *pfA = pf, *pfBm = pf + kPfTile, *pfE = pf + 2 * kPfTile, *pfC = pf + 3 * kPfTile, *pfRpm0 = pf + 4 * kPfTile,
         *pfTpp0 = pf + 5 * kPfTile, *pfV = pf + 6 * kPfTile;
  const int span = a.n1_hi - a.n1_lo, CH = a.dn_chunk, nch = (a.nR + CH - 1) / CH;
  const unsigned items = (unsigned)span * (unsigned)nch, istride = gridDim.x * kWavesPerBlock;
  auto flip_pm = [&](Mat<NT> X) {  // ier+- = sgn (.) ier-+, iet-- = sgn (.) iet++ (corrected D2)
    if (a.nS > 1)
      map_t<NT>(g, X, [&](int i, int j, double v) { return dsgn(scomp(i, a.nS, a.strict_idx), scomp(j, a.nS, a.strict_idx)) * v; });
    return X;
  };

  for (unsigned w = blockIdx.x * kWavesPerBlock + wave; w < items; w += istride) {
    const int ch = (int)(w / (unsigned)span), n1 = a.n1_lo + (int)(w - (unsigned)ch * (unsigned)span);
    const int dn_lo = ch * CH, dn_hi = (dn_lo + CH < a.nR) ? dn_lo + CH : a.nR;
    const size_t m1 = NN * n1;
    const Mat<NT> T01_t = load_t<NT>(g, a.sm[SI_T01] + m1), r1_t = load_t<NT>(g, a.x[R_MP] + m1);
    const Mat<NT> T21_t = load_t<NT>(g, a.sm[SI_T21] + m1), Rpm1_t = load_t<NT>(g, a.c_cur[C_R_PM] + m1);
    const int cnt = dn_hi - dn_lo;
    const int off_l = (lane < cnt) ? a.off[dn_lo + lane] : 0;
    const unsigned long long gmask = __builtin_amdgcn_ballot_w64(lane < cnt && n1 + off_l >= 0 && n1 + off_l < a.S);
    auto off_of = [&](int dn) { return __builtin_amdgcn_readlane(off_l, dn - dn_lo); };
    auto next_on_grid = [&](int dn) {
      const int k = dn + 1 - dn_lo;
      const unsigned long long rest = (k < 64) ? (gmask >> k) : 0ull;
      return rest ? dn + 1 + (int)__builtin_ctzll(rest) : dn_hi;
    };
    auto prefetch = [&](int dn) {
      const int n0 = n1 + off_of(dn);
      const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u, m0 = NN * n0, v0 = VS * n0;
      if (!SURF) {
        pf_tile(a.ie_a[R_MP] + o4, pfA, lane);
        pf_tile(a.ie_a[DERIVE ? T_PP : T_MM] + o4, pfBm, lane);
      }
      pf_tile(a.ie_c[C_R_PM] + o4, pfE, lane);
      pf_tile(a.ie_c[C_T_PP] + o4, pfC, lane);
      pf_tile(a.sm[SI_RPM] + m0, pfRpm0, lane);
      pf_tile(a.sm[SI_TPP] + m0, pfTpp0, lane);
      // eight vectors of 16 doubles in one instruction: ieJ0+, ieJ0- (added), ieJ0+, ieJ0- (composite), J0+[n0], G1 v[n0],
      // j0-[n0] of the added layer, G2 v[n0]
      const int q = lane >> 3;
      const double *vp = (q == 0) ? a.ie_a[J0P] + o3 : (q == 1) ? a.ie_a[J0M] + o3 : (q == 2) ? a.ie_c[C_J0P] + o3
                       : (q == 3) ? a.ie_c[C_J0M] + o3 : (q == 4) ? a.c_cur[C_J0P] + v0 : (q == 5) ? a.sv[SVI_G1V] + v0
                       : (q == 6) ? a.x[J0M] + v0 : a.sv[SVI_G2V] + v0;
      if (!(SURF && q < 2))
        __builtin_amdgcn_global_load_lds((momr_gptr *)(vp + 2 * (lane & 7)), (momr_lptr *)pfV, 16, 0, 0);
    };
    // the last two operator blocks of the previous pair, stored at the top of the next one
    Mat<NT> Cn_d, En_d;
    size_t o4_d = 0;
    bool have_d = false;
    auto flush = [&]() {
      if (!have_d) return;
      store_t<NT>(g, a.ie_c[C_T_PP] + o4_d, Cn_d);
      store_t<NT>(g, a.ie_c[C_R_PM] + o4_d, En_d);
      have_d = false;
    };
    int dn_pf = next_on_grid(dn_lo - 1);
    if (dn_pf < dn_hi) prefetch(dn_pf);
    for (int dn = dn_pf; dn < dn_hi; dn = dn_pf) {
      const int n0 = n1 + off_of(dn);
      const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u, m0 = NN * n0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const Mat<NT> a_t = SURF ? zeros<NT>() : lds_tile(g, pfA);
      const Mat<NT> bm_raw = SURF ? zeros<NT>() : lds_tile(g, pfBm);
      const Mat<NT> E_t = lds_tile(g, pfE), C_t = lds_tile(g, pfC);
      const Mat<NT> Rpm0_c = lds_tile(g, pfRpm0), Tpp0_c = lds_tile(g, pfTpp0);
      CV<NT> Jap, Jam, Jcp, Jcm;
      Jap.c[0] = SURF ? 0.0 : pfV[g.lr]; Jam.c[0] = SURF ? 0.0 : pfV[16 + g.lr];
      Jcp.c[0] = pfV[32 + g.lr]; Jcm.c[0] = pfV[48 + g.lr];
      Vec<NT> JcpR, JamR, J0pR, g1vR, j0mR, g2vR;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = g.lq + 4 * r;
        JamR.t[0][r] = SURF ? 0.0 : pfV[16 + k];
        JcpR.t[0][r] = pfV[32 + k];
        J0pR.t[0][r] = pfV[64 + k];
        g1vR.t[0][r] = pfV[80 + k];
        j0mR.t[0][r] = pfV[96 + k];
        g2vR.t[0][r] = pfV[112 + k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      dn_pf = next_on_grid(dn);
      if (dn_pf < dn_hi) prefetch(dn_pf);
      flush();

      Mat<NT> bm_t = bm_raw;
      if (DERIVE && !SURF) bm_t = flip_pm(bm_t);
      const Mat<NT> a_c = transpose<NT>(g, a_t), bm_c = transpose<NT>(g, bm_t);
      const Mat<NT> E_c = transpose<NT>(g, E_t), C_c = transpose<NT>(g, C_t);
      // A = T01 (ier R+-[n0] + r ieR+-) + ieT--                                                                  :252-262
      const Mat<NT> M1_c = TNacc<NT>(g, r1_t, E_c, TN<NT>(g, a_t, Rpm0_c));
      const Mat<NT> A_t = TNacc<NT>(g, M1_c, T01_t, load_t<NT>(g, a.ie_c[C_T_MM] + o4));
      // ieJ0- += T01 (ier J0+[n0] + r ieJ0+ + ieJ0-(added)) + A G1 (j0-[n0] + r[n0] J0+[n0])                      :251-264
      {
        const CV<NT> v1 = mv_t<NT>(g, a_t, J0pR);
        const CV<NT> v2 = mv_t<NT>(g, r1_t, JcpR);
        const CV<NT> uu = cadd<NT>(cadd<NT>(v1, v2), Jam);
        const CV<NT> wv = cadd<NT>(mv_t<NT>(g, T01_t, c2r<NT>(g, uu)), mv_t<NT>(g, A_t, g1vR));
        storeC<NT>(g, a.ie_c[C_J0M] + o3, cadd<NT>(Jcm, wv));
      }
      // ieR-+ += T01 (ier T++[n0] + r ieT++) + A G1 r[n0] T++[n0];  ieT-- = T01 iet-- + A G1 t--[n0]               :271-284
      {
        const Mat<NT> N1_c = TNacc<NT>(g, r1_t, C_c, TN<NT>(g, a_t, Tpp0_c));
        Mat<NT> Rm_t = load_t<NT>(g, a.ie_c[C_R_MP] + o4);
        Rm_t = TNacc<NT>(g, N1_c, T01_t, Rm_t);
        Rm_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G1RT] + m0), A_t, Rm_t);
        store_t<NT>(g, a.ie_c[C_R_MP] + o4, Rm_t);
        const Mat<NT> F_t = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G1T] + m0), A_t, TN<NT>(g, bm_c, T01_t));
        store_t<NT>(g, a.ie_c[C_T_MM] + o4, F_t);
      }
      // B = T21 (ieR+- r[n0] + R+- ier) + iet++                                                                   :302-310
      const Mat<NT> M2_c = TNacc<NT>(g, Rpm1_t, a_c, TN<NT>(g, E_t, load_t<NT>(g, a.sm[SI_R] + m0)));
      // iet++ of the added layer: with DERIVE the block just read as iet-- IS iet++ (before the sign flip)
      const Mat<NT> b_t = SURF ? zeros<NT>() : (DERIVE ? bm_raw : load_t<NT>(g, a.ie_a[T_PP] + o4));
      const Mat<NT> B_t = TNacc<NT>(g, M2_c, T21_t, b_t);
      // ieJ0+ = ieJ0+(added) + T21 (ieJ0+ + ieR+- j0-[n0] + R+- ieJ0-(added)) + B G2 (J0+[n0] + R+-[n0] j0-[n0])    :301-312
      {
        const CV<NT> v3 = mv_t<NT>(g, E_t, j0mR);
        const CV<NT> v4 = SURF ? czeros<NT>() : mv_t<NT>(g, Rpm1_t, JamR);
        const CV<NT> uu = cadd<NT>(cadd<NT>(Jcp, v3), v4);
        const CV<NT> wv = cadd<NT>(mv_t<NT>(g, T21_t, c2r<NT>(g, uu)), mv_t<NT>(g, B_t, g2vR));
        storeC<NT>(g, a.ie_c[C_J0P] + o3, cadd<NT>(Jap, wv));
      }
      // ieT++ = T21 ieT++ + B G2 T++[n0];  ieR+- = ier+- + T21 (ieR+- t--[n0] + R+- iet--) + B G2 R+-[n0] t--[n0]   :320-334
      {
        Cn_d = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G2T] + m0), B_t, TN<NT>(g, C_c, T21_t));
        const Mat<NT> N2_c = TNacc<NT>(g, Rpm1_t, bm_c, TN<NT>(g, E_t, load_t<NT>(g, a.sm[SI_TMM] + m0)));
        // ier+- of the added layer: with DERIVE the sign-flipped ier-+
        Mat<NT> En_t = SURF ? zeros<NT>() : (DERIVE ? flip_pm(a_t) : load_t<NT>(g, a.ie_a[R_PM] + o4));
        En_t = TNacc<NT>(g, N2_c, T21_t, En_t);
        En_d = TNacc<NT>(g, load_t<NT>(g, a.sm[SI_G2RT] + m0), B_t, En_t);
        o4_d = o4; have_d = true;
      }
    }
    flush();
  }
}

template <bool SURF, bool DERIVE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR k_int_pair1(KArgs a, int iface) {
  if (MOMR_LDSPF != 0 && iface == 3) int_pair_body1<SURF, DERIVE>(a);
  else int_pair_body<1, SURF, DERIVE>(a, iface);
}
template <bool SURF, bool DERIVE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_int_pair2(KArgs a, int iface) { int_pair_body<2, SURF, DERIVE>(a, iface); }
#ifdef MOMR_BIG_TU
template <bool SURF, bool DERIVE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_int_pair3(KArgs a, int iface) { int_pair_body<3, SURF, DERIVE>(a, iface); }
template <bool SURF, bool DERIVE>
__global__ void __launch_bounds__(64 * kWavesPerBlock) MOMR_PAIR_ATTR2 k_int_pair4(KArgs a, int iface) { int_pair_body<4, SURF, DERIVE>(a, iface); }
#endif

// create_surface_layer! into the surface layer arrays: kind 0 LambertianSurfaceScalar (Surfaces/lambertian_surface.jl:20-75),
// 1 any BRDF type through its Fourier matrix Rsurf [N,N] of this moment (rpv_surface.jl:20-66), 2 LambertianSurfaceLegendre
// (lambertian_surface.jl:77-138: spectrally varying albedo, j0+ = 0, t = 0 for m > 0)
__global__ void k_surface_fill(KArgs a, const double *tau_tot, int kind, const double *Rsurf, const double *albedo_spec) {
  const int N = a.N, n = a.nS, P = a.P;
  const size_t BS = (size_t)P * P;  // block stride (padded pitch P; the padding stays zero)
  const size_t pt = blockIdx.x;
  const double att = exp(-tau_tot[pt] / a.mu0);
  const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
  double *const *x = a.x;
  if (kind == 1) {
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
      const int j = e / N, i = e - j * N;
      const size_t o = BS * pt + i + (size_t)P * j;
      x[R_MP][o] = Rsurf[i + (size_t)N * j] * (a.mu[j] * a.wt[j]);
      x[R_PM][o] = 0.0;
      x[T_PP][o] = (i == j) ? 1.0 : 0.0;
      x[T_MM][o] = (i == j) ? 1.0 : 0.0;
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      const bool in_sun = (i >= i_start) && (i < i_end);
      double rI = 0.0;
      for (int k = 0; k < n; ++k) rI += Rsurf[i + (size_t)N * (i_start + k)] * a.I0[k];
      x[J0P][(size_t)P * pt + i] = (in_sun ? a.I0[i - i_start] : 0.0) * att;
      x[J0M][(size_t)P * pt + i] = (a.mu0 * rI) * att;
    }
    return;
  }
  const double rho = 2 * ((kind == 2) ? albedo_spec[pt] : a.albedo);
  const double tdiag = (kind == 2 && a.m > 0) ? 0.0 : 1.0;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
    const int j = e / N, i = e - j * N;
    const size_t o = BS * pt + i + (size_t)P * j;
    x[R_MP][o] = (a.m == 0 && (i % n == 0) && (j % n == 0)) ? rho * (a.mu[j] * a.wt[j]) : 0.0;
    if (a.m == 0) x[R_PM][o] = 0.0;  // not reset for m > 0 (:68-73)
    x[T_PP][o] = (i == j) ? tdiag : 0.0;
    x[T_MM][o] = (i == j) ? tdiag : 0.0;
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const bool in_sun = (i >= i_start) && (i < i_end);
    double jp = 0.0, jm = 0.0;
    if (a.m == 0) {
      if (kind == 2) jm = (i % n == 0) ? (a.mu0 * a.I0[0]) * (rho * att) : 0.0;
      else {
        jp = (in_sun ? a.I0[i - i_start] : 0.0) * att;
        jm = (i % n == 0) ? (a.mu0 * (rho * a.I0[0])) * att : 0.0;
      }
    }
    x[J0P][(size_t)P * pt + i] = jp;
    x[J0M][(size_t)P * pt + i] = jm;
  }
}

// postprocessing_vza!(::RRS) (tools/postprocessing_vza.jl:95-147, SFI) and the elastic RAMI extras of rt_run.jl:187-213:
// interaction_hdrf! (CoreKernel/interaction_hdrf.jl:9-45: hdr_J0- = r-+_surf J0+ + j0-_surf with the composite J0+ after the
// surface interaction) + postprocessing_vza_hdrf! (postprocessing_vza.jl:63-93).
// out = [R | T | ieR | ieT | hdr][nVza, nS, S] then bhr_uw, bhr_dw [nS, S]
__global__ void k_post(KArgs a, int nVza, const int *node, const double *cosm, const double *sinm, int M, double *out) {
  const int N = a.N, n = a.nS, P = a.P;
  const size_t tot = (size_t)nVza * n * a.S;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= tot) return;
  const int v = (int)(e % nVza), s = (int)((e / nVza) % n), pt = (int)(e / ((size_t)nVza * n));
  if (pt < a.n1_lo || pt >= a.n1_hi) return;
  const int row = n * (node[v] - 1) + s;
  const double cs = a.weight * ((s < 2) ? cosm[v + nVza * a.m] : sinm[v + nVza * a.m]);
  const double *J0p = a.c_cur[C_J0P] + (size_t)P * pt;
  out[e] += cs * a.c_cur[C_J0M][row + (size_t)P * pt];
  out[tot + e] += cs * J0p[row];
  double sm = 0.0, sp = 0.0;
  for (int t = 0; t < a.nR; ++t) {
    const size_t o = row + (size_t)P * ((size_t)pt + (size_t)a.S * t);
    sm += cs * a.ie_c[C_J0M][o];
    sp += cs * a.ie_c[C_J0P][o];
  }
  out[2 * tot + e] += sm;
  out[3 * tot + e] += sp;
  const double *rs = a.x[R_MP] + (size_t)P * P * pt;   // the surface layer
  double hj = a.x[J0M][row + (size_t)P * pt];
  for (int j = 0; j < N; ++j) hj += rs[row + (size_t)P * j] * J0p[j];
  out[4 * tot + e] += cs * hj;
  if (a.m == 0 && v == 0) {  // bhr_uw / bhr_dw of Stokes component s (interaction_hdrf.jl:28-43)
    double up = 0.0, dw = 0.0;
    for (int jj = s; jj < N; jj += n) {
      double h2 = a.x[J0M][jj + (size_t)P * pt];
      for (int j = 0; j < N; ++j) h2 += rs[jj + (size_t)P * j] * J0p[j];
      up += h2 * a.wt[jj] * a.mu[jj];
      dw += J0p[jj] * a.wt[jj] * a.mu[jj];
    }
    const int i0 = n * (a.imu0 - 1);
    out[5 * tot + s + (size_t)n * pt] = up;
    out[5 * tot + (size_t)n * a.S + s + (size_t)n * pt] = dw + a.x[J0P][i0 + (size_t)P * pt] * a.mu[i0];
  }
}

// ABI order <-> device blocks: `nat` holds nblk blocks of rows x cols doubles (column-major, the reference's memory order),
// `dev` the same blocks at pitch P with zero padding (cols = 1: vectors, block stride P; else block stride P x P).
__global__ void k_pack(double *dev, const double *nat, int rows, int cols, int P, size_t nblk) {
  const size_t bs = (cols == 1) ? (size_t)P : (size_t)P * P, tot = bs * nblk;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / bs;
    const int w = (int)(e - b * bs), i = w % P, j = w / P;
    dev[e] = (i < rows && j < cols) ? nat[(size_t)rows * cols * b + i + (size_t)rows * j] : 0.0;
  }
}
__global__ void k_unpack(double *nat, const double *dev, int rows, int cols, int P, size_t nblk) {
  const size_t nb = (size_t)rows * cols, tot = nb * nblk, bs = (cols == 1) ? (size_t)P : (size_t)P * P;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / nb;
    const int w = (int)(e - b * nb), i = w % rows, j = w / rows;
    nat[e] = dev[bs * b + i + (size_t)P * j];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
#ifdef MOMR_BIG_TU
#include "mom_rrs_wg.hpp"
}  // namespace momr_big
// which: 0 k_el_point, 1 k_dbl_point, 2 k_int_point, 3 k_dbl_pair (v0 = fused elemental, v1 = mode), 4 k_int_pair (v0 = surface,
// v1 = derived +- / -- blocks), 5 k_dbl_pair_wg, 6 k_int_pair_wg (workgroup per pair, mom_rrs_wg.hpp; grid = workgroups), 7 k_dbl_point_wg (workgroup per point, nt >= 3), 8 k_ie_elemental_tile (v0 = layer without doublings), 9 k_int_point_wg; nt = 3 or 4; args: the
// KArgs of the caller (layout-identical in both namespaces)
hipError_t momr_big_launch(int which, int nt, int v0, int v1, unsigned grid, void *stream, const void *args, int iface) {
  using namespace momr_big;
  const KArgs a = *reinterpret_cast<const KArgs *>(args);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 gr(grid), bl(64 * kWavesPerBlock);
  const size_t lds = (size_t)kWavesPerBlock * (nt == 3 ? slice_bytes<3>() : slice_bytes<4>());
#define BIG_GO(KERN, ...)                                                                                                      \
  do {                                                                                                                         \
    const hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e__ != hipSuccess) return e__;                                                                                         \
    hipLaunchKernelGGL(KERN, gr, bl, lds, st, __VA_ARGS__);                                                                    \
    return hipGetLastError();                                                                                                  \
  } while (0)
#define BIG_NT(KERN3, KERN4, ...) do { if (nt == 3) BIG_GO(KERN3, __VA_ARGS__); else BIG_GO(KERN4, __VA_ARGS__); } while (0)
  const dim3 blw(64 * nt);
  const size_t ldw = (nt == 2) ? wg_lds_bytes<2>(4) : ((nt == 3) ? wg_lds_bytes<3>(4) : wg_lds_bytes<4>(4));
#define WG_GO(KERN)                                                                                                            \
  do {                                                                                                                         \
    const hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldw); \
    if (e__ != hipSuccess) return e__;                                                                                         \
    hipLaunchKernelGGL(KERN, gr, blw, ldw, st, a);                                                                             \
    return hipGetLastError();                                                                                                  \
  } while (0)
  switch (which) {
    case 0: BIG_NT(k_el_point<3>, k_el_point<4>, a);
    case 1: BIG_NT(k_dbl_point<3>, k_dbl_point<4>, a);
    case 2: BIG_NT(k_int_point<3>, k_int_point<4>, a, iface);
    case 3:
#define BIG_DBL(F, M) BIG_NT((k_dbl_pair3<F, M>), (k_dbl_pair4<F, M>), a)
      if (v0) { if (v1 == 0) BIG_DBL(true, 0); else if (v1 == 1) BIG_DBL(true, 1); else BIG_DBL(true, 2); }
      else { if (v1 == 0) BIG_DBL(false, 0); else if (v1 == 1) BIG_DBL(false, 1); else BIG_DBL(false, 2); }
#undef BIG_DBL
    case 4:
      if (v0) BIG_NT((k_int_pair3<true, false>), (k_int_pair4<true, false>), a, iface);
      else if (v1) BIG_NT((k_int_pair3<false, true>), (k_int_pair4<false, true>), a, iface);
      else BIG_NT((k_int_pair3<false, false>), (k_int_pair4<false, false>), a, iface);
    case 5: {
      // products without the per-k-step guards where no k-step lies in the zero padding, or one of sixteen (mom_rrs_wg.hpp NG;
      // one of twelve -- N = 42 ... 44 -- measured slower than the guards: profiles/r05_rrs_wg_ab.txt (13))
      const int padded_steps = (16 * nt - a.N) / 4;
      const bool ng = padded_steps == 0 || (padded_steps == 1 && nt == 4);
#define WG_DBL_G(F, M, G) do { if (nt == 2) WG_GO((k_dbl_pair_wg2<F, M, G>)); else if (nt == 3) WG_GO((k_dbl_pair_wg3<F, M, G>)); else WG_GO((k_dbl_pair_wg4<F, M, G>)); } while (0)
#define WG_DBL(F, M) do { if (ng) WG_DBL_G(F, M, true); else WG_DBL_G(F, M, false); } while (0)
      if (v0) { if (v1 == 0) WG_DBL(true, 0); else if (v1 == 1) WG_DBL(true, 1); else WG_DBL(true, 2); }
      else { if (v1 == 0) WG_DBL(false, 0); else if (v1 == 1) WG_DBL(false, 1); else WG_DBL(false, 2); }
#undef WG_DBL
#undef WG_DBL_G
    }
    case 6: {  // ScatteringInterface_11 only
#define WG_INT(SF, DV) do { if (nt == 2) WG_GO((k_int_pair_wg2<SF, DV>)); else if (nt == 3) WG_GO((k_int_pair_wg3<SF, DV>)); else WG_GO((k_int_pair_wg4<SF, DV>)); } while (0)
      if (v0) WG_INT(true, false);
      else if (v1) WG_INT(false, true);
      else WG_INT(false, false);
#undef WG_INT
    }
    case 9: {  // k_int_point_wg (interface 11): one workgroup per spectral point
      const size_t ldp = (nt == 2) ? wg_point_lds_bytes<2>() : ((nt == 3) ? wg_point_lds_bytes<3>() : wg_point_lds_bytes<4>());
      const void *kp = (nt == 2) ? reinterpret_cast<const void *>(k_int_point_wg2)
                                 : ((nt == 3) ? reinterpret_cast<const void *>(k_int_point_wg3) : reinterpret_cast<const void *>(k_int_point_wg4));
      const hipError_t e__ = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldp);
      if (e__ != hipSuccess) return e__;
      if (nt == 2) hipLaunchKernelGGL(k_int_point_wg2, gr, blw, ldp, st, a);
      else if (nt == 3) hipLaunchKernelGGL(k_int_point_wg3, gr, blw, ldp, st, a);
      else hipLaunchKernelGGL(k_int_point_wg4, gr, blw, ldp, st, a);
      return hipGetLastError();
    }
    case 8:
      if (v0) BIG_NT((k_ie_elemental_tile<3, true>), (k_ie_elemental_tile<4, true>), a);
      else BIG_NT((k_ie_elemental_tile<3, false>), (k_ie_elemental_tile<4, false>), a);
    case 7: {  // k_dbl_point_wg: one workgroup per spectral point (grid = workgroups)
      const size_t ldp = (nt == 2) ? wg_point_lds_bytes<2>() : ((nt == 3) ? wg_point_lds_bytes<3>() : wg_point_lds_bytes<4>());
      const void *kp = (nt == 2) ? reinterpret_cast<const void *>(k_dbl_point_wg2)
                                 : ((nt == 3) ? reinterpret_cast<const void *>(k_dbl_point_wg3) : reinterpret_cast<const void *>(k_dbl_point_wg4));
      const hipError_t e__ = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldp);
      if (e__ != hipSuccess) return e__;
      if (nt == 2) hipLaunchKernelGGL(k_dbl_point_wg2, gr, blw, ldp, st, a);
      else if (nt == 3) hipLaunchKernelGGL(k_dbl_point_wg3, gr, blw, ldp, st, a);
      else hipLaunchKernelGGL(k_dbl_point_wg4, gr, blw, ldp, st, a);
      return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
  }
#undef WG_GO
#undef BIG_NT
#undef BIG_GO
}
#else
#define RCHK(call)                                 \
  do {                                             \
    hipError_t e__ = (call);                       \
    if (e__ != hipSuccess) return e__;             \
  } while (0)

static KArgs base_args(const State *s, const Streams &q) {
  KArgs a{};
  a.N = s->N; a.nS = s->nS; a.S = s->S; a.nR = s->nR; a.P = s->P;
  a.strict_idx = q.strict_idx; a.strict_rrs = s->strict_rrs;
  a.n_glob0 = s->n_glob0; a.n1_lo = s->n1_lo; a.n1_hi = s->n1_hi;
  a.imu0 = q.imu0; a.mu0 = q.mu0;
  for (int k = 0; k < 4; ++k) { a.I0[k] = q.I0[k]; a.D[k] = q.D[k]; }
  a.mu = q.mu; a.wt = q.wt; a.off = s->d_off; a.varpiR = s->d_varpiR;
  for (int k = 0; k < 6; ++k) {
    a.a_cur[k] = s->added[s->cur][k]; a.a_nxt[k] = s->added[1 - s->cur][k];
    a.c_cur[k] = s->comp[s->ccur][k]; a.c_nxt[k] = s->comp[1 - s->ccur][k];
    a.ie_a[k] = s->ie_added[k]; a.ie_c[k] = s->ie_comp[k];
    a.sv[k] = s->svec[k];
  }
  // r+- / t-- of the added layer exist once
  a.a_cur[R_PM] = a.a_nxt[R_PM] = s->added[0][R_PM];
  a.a_cur[T_MM] = a.a_nxt[T_MM] = s->added[0][T_MM];
  a.expk_cur = s->expk[s->cur]; a.expk_nxt = s->expk[1 - s->cur];
  for (int k = 0; k < 10; ++k) a.sm[k] = s->smat[k];
  a.jpseq = s->jpseq;
  a.info = s->d_info;
  return a;
}

template <class T>
static hipError_t dm(T **p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
  if (e == hipSuccess) e = hipMemset(*p, 0, count * sizeof(T));
  return e;
}
// Layer arrays sit between two GUARD BANDS of kGuard doubles filled with a canary pattern (r5, ADVICE r4: the kernels store whole
// tiles without edge masks; the zero-padding invariant only sees writes that land INSIDE an array).  count_padding checks the
// bands of every array too: a store before the first or behind the last block of any layer array shows up as a violation
// (tests/test_gpu_rrs.py::test_rrs_zero_padding_invariant, N = 15 ... 60, both switch positions).
constexpr size_t kGuard = 2048;               // doubles per band (16 KB: more than a 4 x 4-tile block row)
constexpr unsigned kCanary = 0xDEADBEEFu;     // both halves of a canary double
static hipError_t dmg(State *s, double **p, size_t count) {
  double *base = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), (count + 2 * kGuard) * sizeof(double));
  if (e == hipSuccess) e = hipMemset(base + kGuard, 0, count * sizeof(double));
  if (e == hipSuccess) e = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(base), (int)kCanary, 2 * kGuard);
  if (e == hipSuccess) e = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(base + kGuard + count), (int)kCanary, 2 * kGuard);
  if (e != hipSuccess) { (void)hipFree(base); return e; }
  *p = base + kGuard;
  s->guarded.emplace_back(*p, count);
  return hipSuccess;
}
static void dfreeg(double *p) { if (p) (void)hipFree(p - kGuard); }

hipError_t create(State **out, hipStream_t st, int N, int nS, int S, int nR, const int *off_host, const double *varpi_host,
                  int strict_rrs) {
  State *s = new State;
  s->N = N; s->nS = nS; s->S = S; s->nR = nR; s->strict_rrs = strict_rrs; s->stream = st;
  s->n1_lo = 0; s->n1_hi = S;
  s->P = 16 * ((N + 15) / 16);  // device blocks are zero-padded to the MFMA tiling: P x P per matrix, P per vector
  *out = s;
  const size_t NN = (size_t)s->P * s->P, m3 = NN * S, v3 = (size_t)s->P * S, m4 = m3 * nR, v4 = v3 * nR;
  RCHK(dm(&s->d_off, nR));
  RCHK(dm(&s->d_varpiR, nR));
  RCHK(hipMemcpy(s->d_off, off_host, sizeof(int) * nR, hipMemcpyHostToDevice));
  RCHK(hipMemcpy(s->d_varpiR, varpi_host, sizeof(double) * nR, hipMemcpyHostToDevice));
  for (int k = 0; k < nR; ++k) s->max_off = std::max(s->max_off, std::abs(off_host[k]));
  for (int b = 0; b < 2; ++b) {
    for (int k = 0; k < 6; ++k) {
      if (b == 1 && (k == R_PM || k == T_MM)) continue;
      RCHK(dmg(s, &s->added[b][k], k < 4 ? m3 : v3));
    }
    for (int k = 0; k < 6; ++k) RCHK(dmg(s, &s->comp[b][k], k < 4 ? m3 : v3));
    RCHK(dmg(s, &s->expk[b], S));
  }
  s->added[1][R_PM] = s->added[0][R_PM];
  s->added[1][T_MM] = s->added[0][T_MM];
  for (int k = 0; k < 6; ++k) RCHK(dmg(s, &s->surf[k], k < 4 ? m3 : v3));
  for (int k = 0; k < 10; ++k) RCHK(dmg(s, &s->smat[k], m3));
  for (int k = 0; k < 6; ++k) RCHK(dmg(s, &s->svec[k], v3));
  if (strict_rrs) RCHK(dmg(s, &s->jpseq, v4));
  for (int k = 0; k < 6; ++k) {
    RCHK(dmg(s, &s->ie_added[k], k < 4 ? m4 : v4));
    RCHK(dmg(s, &s->ie_comp[k], k < 4 ? m4 : v4));
  }
  RCHK(dm(&s->d_info, 1));
  return hipSuccess;
}

// one array of the layers between the ABI's memory order (host) and the padded device blocks; nblk = S or S x nR
hipError_t upload(State *s, double *dev, const double *host, bool matrix, size_t nblk) {
  const size_t nat = (size_t)s->N * (matrix ? s->N : 1) * nblk;
  if (nat > s->stage_cap) {
    (void)hipFree(s->d_stage);
    s->d_stage = nullptr; s->stage_cap = 0;
    RCHK(hipMalloc(reinterpret_cast<void **>(&s->d_stage), nat * sizeof(double)));
    s->stage_cap = nat;
  }
  RCHK(hipMemcpyAsync(s->d_stage, host, nat * sizeof(double), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_pack, dim3(2048), dim3(256), 0, s->stream, dev, s->d_stage, s->N, matrix ? s->N : 1, s->P, nblk);
  RCHK(hipGetLastError());
  return hipStreamSynchronize(s->stream);
}
hipError_t download(State *s, double *host, const double *dev, bool matrix, size_t nblk) {
  const size_t nat = (size_t)s->N * (matrix ? s->N : 1) * nblk;
  if (nat > s->stage_cap) {
    (void)hipFree(s->d_stage);
    s->d_stage = nullptr; s->stage_cap = 0;
    RCHK(hipMalloc(reinterpret_cast<void **>(&s->d_stage), nat * sizeof(double)));
    s->stage_cap = nat;
  }
  hipLaunchKernelGGL(k_unpack, dim3(2048), dim3(256), 0, s->stream, s->d_stage, dev, s->N, matrix ? s->N : 1, s->P, nblk);
  RCHK(hipGetLastError());
  RCHK(hipMemcpyAsync(host, s->d_stage, nat * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  return hipStreamSynchronize(s->stream);
}

void destroy(State *s) {
  if (!s) return;
  (void)hipFree(s->d_stage);
  (void)hipFree(s->d_off); (void)hipFree(s->d_varpiR); (void)hipFree(s->d_out); (void)hipFree(s->d_info);
  for (auto &g : s->guarded) dfreeg(g.first);   // every layer array (added, comp, expk, surf, smat, svec, jpseq, ie_added, ie_comp)
  for (auto e : s->ev_pool) (void)hipEventDestroy(e);
  delete s;
}

// scene-level fast-mode features (A/B switch for profiling: MOM_RRS_FAST = bit 0 deferred elemental, bit 1 derived ier+- / iet--)
static int fast_bits() {
#ifdef MOM_EXPERIMENTS
  static const int bits = [] { const char *e = getenv("MOM_RRS_FAST"); return e ? atoi(e) : MOMR_FAST_DEFAULT; }();
  return bits;
#else
  return MOMR_FAST_DEFAULT;
#endif
}
// experiment-only knobs: environment variables in -DMOM_EXPERIMENTS builds, constants in the shipped library
static int exp_int(const char *name, int dflt) {
#ifdef MOM_EXPERIMENTS
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

void timing_reset(State *s, bool on) {
  s->timing = on;
  s->ev_kind.clear();
}
static hipError_t tick(State *s, int kind, bool start) {
  if (!s->timing) return hipSuccess;
  const size_t idx = 2 * s->ev_kind.size() + (start ? 0 : 1);
  while (s->ev_pool.size() <= idx) {
    hipEvent_t e;
    RCHK(hipEventCreate(&e));
    s->ev_pool.push_back(e);
  }
  RCHK(hipEventRecord(s->ev_pool[idx], s->stream));
  if (!start) s->ev_kind.push_back(kind);
  return hipSuccess;
}
hipError_t timing_read(State *s, double *ms, int *launches) {
  for (int k = 0; k < TK_COUNT; ++k) { ms[k] = 0.0; launches[k] = 0; }
  RCHK(hipStreamSynchronize(s->stream));
  for (size_t k = 0; k < s->ev_kind.size(); ++k) {
    float t = 0.f;
    RCHK(hipEventElapsedTime(&t, s->ev_pool[2 * k], s->ev_pool[2 * k + 1]));
    ms[s->ev_kind[k]] += t;
    launches[s->ev_kind[k]] += 1;
  }
  return hipSuccess;
}

// Kernel-form switches of the handle (State::kopt, MOM_OPT_RRS_KERNELS of mom_set_option; r5 read them from the environment):
// 16 < N <= 64: the pair kernels as one workgroup per pair (mom_rrs_wg.hpp); KOPT_WG off selects the wave-per-pair bodies
static bool wg_pairs(const State *s) { return (s->kopt & KOPT_WG) != 0; }
// tile count of the workgroup-per-pair image for this edge, 0 = wave-per-pair kernels (N <= 16; 16 < N <= 32 without KOPT_WG2)
static int wg_nt(const State *s) {
  if (!wg_pairs(s) || s->N <= 16) return 0;
  if (s->N <= 32) return (s->kopt & KOPT_WG2) ? 2 : 0;
  return s->N <= 48 ? 3 : 4;
}
static size_t wg_grid(const State *s, int nt) {  // persistent workgroups, one round of the chip
  static const int mult = std::max(1, exp_int("MOM_RRS_WG_GRID", 1));
  const size_t np = (size_t)(s->n1_hi - s->n1_lo) * s->nR, per_cu = (nt == 2) ? 4 : ((nt == 3) ? 2 : 1);
  return std::max<size_t>(1, std::min<size_t>(np, 256 * per_cu * mult));
}
static int grid_points(const State *s) { return std::max(1, std::min((s->S + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 8)); }
static int grid_pairs(const State *s) {
  const size_t np = (size_t)(s->n1_hi - s->n1_lo) * s->nR;
  return (int)std::max<size_t>(1, std::min<size_t>((np + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 16));
}
template <int NT>
static size_t lds() { return (size_t)kWavesPerBlock * slice_bytes<NT>(); }

// launch with the dynamic LDS of the image's tile count (above 64 KB the limit of the function has to be raised first)
template <class K, class... Args>
static hipError_t launch_lds(K kern, dim3 grid, size_t lds_bytes, hipStream_t st, Args... args) {
  if (lds_bytes > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * kWavesPerBlock), lds_bytes, st, args...);
  return hipGetLastError();
}
template <class T> static const void *first_arg(const T &a) { return &a; }
template <class T, class U> static const void *first_arg(const T &a, const U &) { return &a; }
template <class T> static int iface_arg(const T &) { return 0; }
template <class T> static int iface_arg(const T &, int iface) { return iface; }
#define LAUNCH_NT(s, kern, which, grid, ...)                                                               \
  do {                                                                                                    \
    if ((s)->N <= 16) RCHK(launch_lds(kern<1>, dim3(grid), lds<1>(), (s)->stream, __VA_ARGS__));          \
    else if ((s)->N <= 32) RCHK(launch_lds(kern<2>, dim3(grid), lds<2>(), (s)->stream, __VA_ARGS__));     \
    else RCHK(momr_big_launch(which, (s)->N <= 48 ? 3 : 4, 0, 0, (unsigned)(grid), (void *)(s)->stream,   \
                              first_arg(__VA_ARGS__), iface_arg(__VA_ARGS__)));                           \
  } while (0)

hipError_t elemental(State *s, const Streams &q, int m, int nd, int shift, const double *tau_sum, const double *tau, const double *varpi,
                     const double *Zpp, const double *Zmp, int nTerms, const double *zw, const double *fscatt,
                     const double *Zr_pp, const double *Zr_mp, bool elastic, bool inelastic) {
  KArgs a = base_args(s, q);
  a.m = m; a.nd = nd; a.sh = shift; a.tau_sum = tau_sum; a.tau = tau; a.varpi = varpi; a.Zpp = Zpp; a.Zmp = Zmp; a.nTerms = nTerms; a.zw = zw;
  a.fscatt = fscatt; a.Zr_pp = Zr_pp; a.Zr_mp = Zr_mp;
  if (elastic) LAUNCH_NT(s, k_el_point, 0, grid_points(s), a);
  s->el_pending = false;
  if (inelastic) {
    // scene-level fast mode: the tile form (k_ie_elemental_tile).  For a layer with doublings the alternative is to form the layer
    // inside the first doubling step (fuse_el): KOPT_EL_FUSE_ON / _OFF (neither: only above N = 48; below, the separate tile kernel +
    // the plain first step measured faster, profiles/r05_C5_ab.txt (5)).  KOPT_EL_TILE off: the element-wise kernel below / the fused form as in r4.
    const bool el_tile = (s->kopt & KOPT_EL_TILE) != 0;
    const int el_fuse_env = (s->kopt & KOPT_EL_FUSE_ON) ? 1 : ((s->kopt & KOPT_EL_FUSE_OFF) ? 0 : -1);
    const bool el_fuse = el_fuse_env >= 0 ? el_fuse_env == 1 : s->N > 48;  // 4 x 4 tiles: the fused form stays ahead (N = 60: 414 vs 420 ms per run)
    if (s->fast && nd >= 1 && (fast_bits() & 1) && (el_fuse || !el_tile)) {  // deferred into the first doubling step (k_dbl_pair, fuse_el)
      s->el_pending = true;
      s->el.m = m; s->el.nd = nd; s->el.sh = shift; s->el.tau_sum = tau_sum; s->el.tau = tau; s->el.varpi = varpi;
      s->el.fscatt = fscatt; s->el.Zr_pp = Zr_pp; s->el.Zr_mp = Zr_mp;
      return hipSuccess;
    }
    if (s->fast && (fast_bits() & 1) && el_tile) {
      const bool derive = !s->strict_rrs && (fast_bits() & 2);
      a.derive_pm = derive ? 1 : 0;  // (read by the ND0 image only)
      RCHK(tick(s, TK_IE_ELEMENTAL, true));
      if ((size_t)(s->n1_hi - s->n1_lo) * s->nR) {
        const dim3 gr(grid_pairs(s));
        if (s->N <= 16) { if (nd < 1) RCHK(launch_lds((k_ie_elemental_tile<1, true>), gr, lds<1>(), s->stream, a)); else RCHK(launch_lds((k_ie_elemental_tile<1, false>), gr, lds<1>(), s->stream, a)); }
        else if (s->N <= 32) { if (nd < 1) RCHK(launch_lds((k_ie_elemental_tile<2, true>), gr, lds<2>(), s->stream, a)); else RCHK(launch_lds((k_ie_elemental_tile<2, false>), gr, lds<2>(), s->stream, a)); }
        else RCHK(momr_big_launch(8, s->N <= 48 ? 3 : 4, nd < 1 ? 1 : 0, 0, gr.x, (void *)s->stream, &a, 0));
      }
      RCHK(tick(s, TK_IE_ELEMENTAL, false));
      if (nd < 1) { s->pm_valid = !derive; s->pm_derivable = true; }
      return hipSuccess;
    }
    const size_t tot = (size_t)s->N * s->N * (size_t)(s->n1_hi - s->n1_lo) * s->nR;
    RCHK(tick(s, TK_IE_ELEMENTAL, true));
    if (tot) hipLaunchKernelGGL(k_ie_elemental, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, a);  // empty owned range: nothing to do
    RCHK(hipGetLastError());
    RCHK(tick(s, TK_IE_ELEMENTAL, false));
    if (nd < 1) { s->pm_valid = true; s->pm_derivable = true; }  // apply_D_elemental_RRS! wrote both pairs consistently
  }
  return hipSuccess;
}

hipError_t ensure_pm(State *s, const Streams &q) {
  if (s->pm_valid || !s->pm_derivable) return hipSuccess;
  KArgs a = base_args(s, q);
  hipLaunchKernelGGL(k_materialise_pm, dim3(2048), dim3(256), 0, s->stream, a);
  RCHK(hipGetLastError());
  s->pm_valid = true;
  return hipSuccess;
}

hipError_t doubling(State *s, const Streams &q, int nd) {
  if (nd == 0) return hipSuccess;  // doubling_inelastic.jl:29
  const bool derive = s->fast && !s->strict_rrs && (fast_bits() & 2);
  for (int k = 0; k < nd; ++k) {
    KArgs a = base_args(s, q);
    a.last = (k == nd - 1);
    a.derive_pm = derive ? 1 : 0;
    if (k == 0 && s->el_pending) {
      a.fuse_el = 1;
      a.m = s->el.m; a.nd = s->el.nd; a.sh = s->el.sh; a.tau_sum = s->el.tau_sum; a.tau = s->el.tau; a.varpi = s->el.varpi;
      a.fscatt = s->el.fscatt; a.Zr_pp = s->el.Zr_pp; a.Zr_mp = s->el.Zr_mp;
      s->el_pending = false;
    }
    // above N = 16: one workgroup per point (mom_rrs_wg.hpp dbl_point_wg: products in strips from LDS copies, Gauss-Jordan inverse
    // over all waves); KOPT_WG_POINT off selects the wave-per-point kernel (operators in scratch, inverse on one wave)
    const bool wg_point = (s->kopt & KOPT_WG_POINT) != 0;
    static const int wg_point_min = exp_int("MOM_RRS_WG_POINT_MIN", 2);  // smallest tile count
    if (wg_point && wg_nt(s) >= wg_point_min)
      RCHK(momr_big_launch(7, wg_nt(s), 0, 0, (unsigned)std::max(1, std::min(s->S, 256 * 8)), (void *)s->stream, &a, 0));
    else LAUNCH_NT(s, k_dbl_point, 1, grid_points(s), a);
    RCHK(tick(s, TK_DBL_PAIR, true));
    {
      dim3 gr(grid_pairs(s));
      size_t lds1 = lds<1>();
      if (MOMR_LDSPF != 0 && s->N <= 16) {
        // work items (n1, chunk of Raman offsets) of dbl_pair_body1: about four items per resident wave, so that
        // the n1-side operands are loaded once per ~nR / nch pairs and the tail of the launch stays short
        static const int ipw = exp_int("MOMR_ITEMS_PER_WAVE", 4);
        const size_t span = (size_t)(s->n1_hi - s->n1_lo), want = (size_t)std::max(ipw, 1) * 3 * 4 * 256;
        const size_t nch = std::max<size_t>(1, std::min<size_t>((size_t)s->nR, (want + span - 1) / std::max<size_t>(span, 1)));
        a.dn_chunk = std::min(64, (int)(((size_t)s->nR + nch - 1) / nch));  // the offsets of a chunk live in the lanes of a wave
        const size_t items = span * (((size_t)s->nR + a.dn_chunk - 1) / a.dn_chunk);
        gr = dim3((unsigned)std::max<size_t>(1, std::min<size_t>((items + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 16)));
        lds1 = (size_t)kWavesPerBlock * (slice_bytes<1>() + kPfBytes);
      }
      const int mode = s->strict_rrs ? 2 : (a.last ? 1 : 0);
#define DBL_PAIR(NT_, FUSE_, MODE_) RCHK(launch_lds(k_dbl_pair##NT_<FUSE_, MODE_>, gr, (NT_ == 1) ? lds1 : lds<NT_>(), s->stream, a))
#define DBL_PAIR_M(NT_, FUSE_) do { if (mode == 0) DBL_PAIR(NT_, FUSE_, 0); else if (mode == 1) DBL_PAIR(NT_, FUSE_, 1); else DBL_PAIR(NT_, FUSE_, 2); } while (0)
#define DBL_PAIR_F(NT_) do { if (a.fuse_el) DBL_PAIR_M(NT_, true); else DBL_PAIR_M(NT_, false); } while (0)
      if (const int nt = wg_nt(s)) {  // one workgroup per pair (mom_rrs_wg.hpp): each walks ~np / grid pairs
        RCHK(momr_big_launch(5, nt, a.fuse_el ? 1 : 0, mode, (unsigned)wg_grid(s, nt), (void *)s->stream, &a, 0));
      } else if (s->N <= 16) DBL_PAIR_F(1);
      else if (s->N <= 32) DBL_PAIR_F(2);
      else RCHK(momr_big_launch(3, s->N <= 48 ? 3 : 4, a.fuse_el ? 1 : 0, mode, gr.x, (void *)s->stream, &a, 0));
#undef DBL_PAIR_F
#undef DBL_PAIR_M
#undef DBL_PAIR
      RCHK(hipGetLastError());
    }
    RCHK(tick(s, TK_DBL_PAIR, false));
    s->cur = 1 - s->cur;
  }
  KArgs a = base_args(s, q);
  if (s->strict_rrs) {
    if (s->nS == 1) {  // doubling_inelastic.jl:411-414: whole-array copies
      const size_t cnt = (size_t)s->P * s->P * s->S * s->nR;
      hipLaunchKernelGGL(k_copy2, dim3(2048), dim3(256), 0, s->stream, s->ie_added[R_MP], s->ie_added[R_PM], s->ie_added[T_PP],
                         s->ie_added[T_MM], cnt);
    } else {
      const int tot = s->N * (s->n1_hi - s->n1_lo);
      if (tot) hipLaunchKernelGGL(k_strict_D, dim3((tot + 127) / 128), dim3(128), 0, s->stream, a);
    }
    RCHK(hipGetLastError());
    s->pm_valid = true; s->pm_derivable = false;
  } else {
    s->pm_valid = !derive; s->pm_derivable = true;
  }
  return hipSuccess;
}

hipError_t copy_added_to_composite(State *s, const Streams &q) {
  RCHK(ensure_pm(s, q));
  const size_t NN = (size_t)s->P * s->P, m3 = NN * s->S * 8, v3 = (size_t)s->P * s->S * 8, m4 = m3 * s->nR, v4 = v3 * s->nR;
  static const int amap[6] = {R_MP, R_PM, T_PP, T_MM, J0P, J0M};  // composite field k <- added field amap[k]
  for (int k = 0; k < 6; ++k) {
    RCHK(hipMemcpyAsync(s->comp[s->ccur][k], s->added[s->cur][amap[k]], k < 4 ? m3 : v3, hipMemcpyDeviceToDevice, s->stream));
    if (s->fast && !s->strict_rrs && k < 4) {
      // scene-level run, corrected position (the strict one reads stale ier+- / iet-- of the added layer's OWN arrays, D5):
      // the four 4-D operator arrays (2.5 GB each at C5) change roles instead of being copied -- the next
      // layer's elemental overwrites every block of the added layer's arrays before anything reads them, and the entries
      // off the grid are zero in both sets (nothing ever writes them)
      std::swap(s->ie_comp[k], s->ie_added[amap[k]]);
    } else {
      RCHK(hipMemcpyAsync(s->ie_comp[k], s->ie_added[amap[k]], k < 4 ? m4 : v4, hipMemcpyDeviceToDevice, s->stream));
    }
  }
  return hipSuccess;
}

hipError_t interaction(State *s, const Streams &q, int iface, bool with_surface) {
  if (iface < 0 || iface > 3) return hipErrorInvalidValue;
  if (iface != 3 && s->strict_rrs) {
    s->err = "interaction_helper!(::RRS, ::ScatteringInterface_00/01/10): the reference raises a MethodError "
             "(interaction_inelastic.jl:8-12, 28-37, 139-148); available with rrs_strict_reference = 0";
    return hipErrorInvalidValue;
  }
  KArgs a = base_args(s, q);
  a.derive_pm = (!with_surface && !s->strict_rrs && s->fast && s->pm_derivable && (fast_bits() & 2)) ? 1 : 0;
  if (!with_surface && !a.derive_pm) RCHK(ensure_pm(s, q));
  for (int k = 0; k < 6; ++k) a.x[k] = with_surface ? s->surf[k] : s->added[s->cur][k];
  if (!with_surface) { a.x[R_PM] = s->added[0][R_PM]; a.x[T_MM] = s->added[0][T_MM]; }
  const bool wg_point = (s->kopt & KOPT_WG_POINT) != 0;
  static const int wg_point_min = exp_int("MOM_RRS_WG_POINT_MIN", 2);
  if (wg_point && iface == 3 && wg_nt(s) >= wg_point_min)  // one workgroup per point (mom_rrs_wg.hpp int_point_wg)
    RCHK(momr_big_launch(9, wg_nt(s), 0, 0, (unsigned)std::max(1, std::min(s->S, 256 * 8)), (void *)s->stream, &a, 0));
  else LAUNCH_NT(s, k_int_point, 2, grid_points(s), a, iface);
  if (iface == 0) {  // interaction_inelastic.jl:16-17
    const size_t v4 = (size_t)s->P * s->S * s->nR * 8;
    RCHK(hipMemsetAsync(s->ie_comp[C_J0P], 0, v4, s->stream));
    RCHK(hipMemsetAsync(s->ie_comp[C_J0M], 0, v4, s->stream));
  } else {
    RCHK(tick(s, TK_INT_PAIR, true));
    dim3 gr(grid_pairs(s));
    size_t lds1 = lds<1>();
    if (MOMR_LDSPF != 0 && s->N <= 16 && iface == 3) {  // int_pair_body1: (n1, chunk) items, as for the doubling pair kernel
      static const int ipw = exp_int("MOMR_ITEMS_PER_WAVE", 4);
      const size_t span = (size_t)(s->n1_hi - s->n1_lo), want = (size_t)std::max(ipw, 1) * 2 * 4 * 256;
      const size_t nch = std::max<size_t>(1, std::min<size_t>((size_t)s->nR, (want + span - 1) / std::max<size_t>(span, 1)));
      a.dn_chunk = std::min(64, (int)(((size_t)s->nR + nch - 1) / nch));
      const size_t items = span * (((size_t)s->nR + a.dn_chunk - 1) / a.dn_chunk);
      gr = dim3((unsigned)std::max<size_t>(1, std::min<size_t>((items + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 16)));
      lds1 = (size_t)kWavesPerBlock * (slice_bytes<1>() + kPfIntBytes);
    }
#define INT_PAIR(NT_)                                                                                                      \
  do {                                                                                                                     \
    const size_t l_ = (NT_ == 1) ? lds1 : lds<NT_>();                                                                       \
    if (with_surface) RCHK(launch_lds(k_int_pair##NT_<true, false>, gr, l_, s->stream, a, iface));                         \
    else if (a.derive_pm) RCHK(launch_lds(k_int_pair##NT_<false, true>, gr, l_, s->stream, a, iface));                     \
    else RCHK(launch_lds(k_int_pair##NT_<false, false>, gr, l_, s->stream, a, iface));                                     \
  } while (0)
    const int nt_wg = (iface == 3) ? wg_nt(s) : 0;
    if (nt_wg) {  // one workgroup per pair (mom_rrs_wg.hpp)
      RCHK(momr_big_launch(6, nt_wg, with_surface ? 1 : 0, a.derive_pm ? 1 : 0, (unsigned)wg_grid(s, nt_wg), (void *)s->stream, &a, iface));
    } else if (s->N <= 16) INT_PAIR(1);
    else if (s->N <= 32) INT_PAIR(2);
    else RCHK(momr_big_launch(4, s->N <= 48 ? 3 : 4, with_surface ? 1 : 0, a.derive_pm ? 1 : 0, gr.x, (void *)s->stream, &a, iface));
#undef INT_PAIR
    RCHK(hipGetLastError());
    RCHK(tick(s, TK_INT_PAIR, false));
  }
  s->ccur = 1 - s->ccur;
  return hipSuccess;
}

hipError_t surface(State *s, const Streams &q, int m, int kind, double albedo, const double *tau_tot, const double *Rsurf_m,
                   const double *albedo_spec) {
  KArgs a = base_args(s, q);
  a.m = m; a.albedo = albedo;
  for (int k = 0; k < 6; ++k) a.x[k] = s->surf[k];
  hipLaunchKernelGGL(k_surface_fill, dim3(s->S), dim3(256), 0, s->stream, a, tau_tot, kind, Rsurf_m, albedo_spec);
  return hipGetLastError();
}

// Invariant of the padded device layout (mom_tile.hpp: whole-tile stores): every entry of a layer block outside the
// N x N (or N) part is an exact zero.  Counts the violations over all layer arrays (test access: mom_rrs_check_padding).
__global__ void k_count_padding(const double *p, int N, int P, int matrix, size_t nblk, unsigned long long *cnt) {
  const size_t bs = matrix ? (size_t)P * P : (size_t)P, tot = bs * nblk;
  unsigned long long mine = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
    const size_t w = e % bs;
    const int i = (int)(w % P), j = (int)(w / P);
    if ((i >= N || (matrix && j >= N)) && p[e] != 0.0) ++mine;   // NaN counts as well
  }
  if (mine) atomicAdd(cnt, mine);
}
__global__ void k_count_canary(const unsigned *p, size_t n, unsigned long long *cnt) {
  unsigned long long mine = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    if (p[e] != kCanary) ++mine;
  if (mine) atomicAdd(cnt, mine);
}
hipError_t count_padding(State *s, unsigned long long *out) {
  unsigned long long *d = nullptr;
  RCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long)));
  RCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), s->stream));
  auto scan = [&](const double *p, bool matrix, size_t nblk) {
    if (p) hipLaunchKernelGGL(k_count_padding, dim3(1024), dim3(256), 0, s->stream, p, s->N, s->P, matrix ? 1 : 0, nblk, d);
  };
  const size_t S = s->S, SR = S * (size_t)s->nR;
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 6; ++k) {
      if (!(b == 1 && (k == R_PM || k == T_MM))) scan(s->added[b][k], k < 4, S);
      scan(s->comp[b][k], k < 4, S);
    }
  for (int k = 0; k < 6; ++k) {
    scan(s->surf[k], k < 4, S);
    scan(s->ie_added[k], k < 4, SR);
    scan(s->ie_comp[k], k < 4, SR);
  }
  for (auto &g : s->guarded) {
    hipLaunchKernelGGL(k_count_canary, dim3(8), dim3(256), 0, s->stream, reinterpret_cast<const unsigned *>(g.first - kGuard), 2 * kGuard, d);
    hipLaunchKernelGGL(k_count_canary, dim3(8), dim3(256), 0, s->stream, reinterpret_cast<const unsigned *>(g.first + g.second), 2 * kGuard, d);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  (void)hipFree(d);
  return e;
}

// rt_run allocates its added / composite / surface layers zeroed on every call (rt_run.jl:108-116: make_added_layer /
// make_composite_layer).  The persistent arrays of a handle are brought to that state before a scene-level run whenever
// anything but a corrected-position scene-level run has touched them (State::dirty): the strict position reads entries the
// current run has not written (D5: iet-- of the previous layer; k_strict_D / the off-grid ieJ0- are read-modify-written), and
// operator-level uploads leave arbitrary values.  A corrected-position scene-level run itself is stateless: every block
// it reads it has written before in the same run, and the entries off the grid are never written (they stay zero).
hipError_t reset_layers(State *s) {
  const size_t NN = (size_t)s->P * s->P, m3 = NN * s->S * 8, v3 = (size_t)s->P * s->S * 8, m4 = m3 * s->nR, v4 = v3 * s->nR;
  for (int b = 0; b < 2; ++b) {
    for (int k = 0; k < 6; ++k) {
      if (!(b == 1 && (k == R_PM || k == T_MM))) RCHK(hipMemsetAsync(s->added[b][k], 0, k < 4 ? m3 : v3, s->stream));
      RCHK(hipMemsetAsync(s->comp[b][k], 0, k < 4 ? m3 : v3, s->stream));
    }
    RCHK(hipMemsetAsync(s->expk[b], 0, (size_t)s->S * 8, s->stream));
  }
  for (int k = 0; k < 6; ++k) {
    RCHK(hipMemsetAsync(s->surf[k], 0, k < 4 ? m3 : v3, s->stream));
    RCHK(hipMemsetAsync(s->ie_added[k], 0, k < 4 ? m4 : v4, s->stream));
    RCHK(hipMemsetAsync(s->ie_comp[k], 0, k < 4 ? m4 : v4, s->stream));
  }
  if (s->jpseq) RCHK(hipMemsetAsync(s->jpseq, 0, v4, s->stream));
  return hipSuccess;
}

hipError_t begin_run(State *s, int nVza) {
  if (s->dirty || s->strict_rrs) RCHK(reset_layers(s));
  s->dirty = s->strict_rrs != 0;  // a strict-position run leaves state the next run must not see
  s->cur = 0; s->ccur = 0;
  s->pm_valid = true; s->pm_derivable = false; s->el_pending = false;
  const size_t cnt = (size_t)5 * nVza * s->nS * s->S + (size_t)2 * s->nS * s->S;
  if (s->out_nVza != nVza) {
    (void)hipFree(s->d_out);
    s->d_out = nullptr;
    RCHK(hipMalloc(reinterpret_cast<void **>(&s->d_out), cnt * 8));
    s->out_nVza = nVza;
  }
  return hipMemsetAsync(s->d_out, 0, cnt * 8, s->stream);
}

hipError_t postprocess(State *s, const Streams &q, int m, int nVza, const int *d_node, const double *d_cos, const double *d_sin,
                       int M, double weight) {
  KArgs a = base_args(s, q);
  a.m = m; a.weight = weight;
  for (int k = 0; k < 6; ++k) a.x[k] = s->surf[k];
  const size_t tot = (size_t)nVza * s->nS * s->S;
  hipLaunchKernelGGL(k_post, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, a, nVza, d_node, d_cos, d_sin, M, s->d_out);
  return hipGetLastError();
}

}  // namespace momr
#endif  // MOMR_BIG_TU

#if defined(MOMR_DIAG_STAMPS) && defined(MOMR_BIG_TU)
extern "C" int momr_big_diag_read(unsigned long long *out, int reset) {  // the stamps of the big-tile object (mom_rrs_wg.hpp)
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(momr_big::momr_diag_acc), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(momr_big::momr_diag_acc), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#if defined(MOMR_DIAG_STAMPS) && !defined(MOMR_BIG_TU)
extern "C" int momr_diag_read(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(momr::momr_diag_acc), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(momr::momr_diag_acc), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
