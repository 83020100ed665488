// mom_small.hip -- the small-operator regime (N = nStokes * Nquad <= 4, BASELINE config C1): ONE SPECTRAL POINT PER
// LANE, the whole of rt_run.jl:125-215 in ONE launch.
//
// A 4 x 4 operator is 16 doubles: the added layer (r-+, t++, j0+-) AND the composite layer (R-+, R+-, T++, T--, J0+-)
// of a spectral point fit the registers of one lane (about 190 doubles live at the peak; the kernel is built for one
// wave per SIMD, 512 VGPRs).  So a lane walks all Fourier moments and all layers of its point -- elemental ->
// doubling -> interaction -> surface -> post-processing -- with fully unrolled N x N arithmetic on the FP64 vector
// FMA pipe (which on MI355X has the same peak as the FP64 matrix pipe, 78.6 TFLOP/s), no LDS, no barrier, no HBM
// traffic for operators at all: per point the kernel reads tau, varpi, the phase-matrix weights and tau_sum of every
// layer once (consecutive lanes = consecutive spectral points: coalesced along the batch axis) and
// writes nVza * nStokes doubles of R_SFI / T_SFI / hdr.  Everything that is the same for all spectral points
// (quadrature, F = mu_j/(mu_i +- mu_j), Z bases, doubling numbers, interface codes) is read through the scalar
// cache.  An MFMA tile would be 1/16 used by a 4 x 4 operator (VERDICT r1 "n1").
//
// Arithmetic follows the reference's op list literally (file:line as in mom_kernels.hpp): elemental.jl:164-253,
// doubling.jl:43-68 with G = inv(I - r r) by Gauss-Jordan with partial pivoting (the LU of gpu_batched.jl:78-82),
// interaction.jl:8-117 (all four interface cases), lambertian_surface.jl:20-75, interaction_hdrf.jl:9-45,
// postprocessing_vza.jl:9-93.  Sums run over k in ascending order like the textbook triple loop.
#include <hip/hip_runtime.h>

#include "mom_host.hpp"

// The same source builds the Float32 lane-per-point kernels (-DMOMS_FLOAT: namespace momsmf, MomSmallSweepArgsF).
#ifdef MOMS_FLOAT
#define MOMS_NS momsmf
#define MOMS_LAUNCH momsmf_launch_sweep
typedef float real;
#else
#define MOMS_NS momsm
#define MOMS_LAUNCH momsm_launch_sweep
typedef double real;
#endif
namespace MOMS_NS {

template <int N>
struct Mat {
  real a[N][N];  // a[i][j]: row i, column j
};
template <int N>
struct Vec {
  real v[N];
};

template <int N>
__device__ __forceinline__ void mul(Mat<N> &C, const Mat<N> &A, const Mat<N> &B) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      real s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) s += A.a[i][k] * B.a[k][j];
      C.a[i][j] = s;
    }
}
template <int N>
__device__ __forceinline__ void mulv(Vec<N> &y, const Mat<N> &A, const Vec<N> &x) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    real s = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) s += A.a[i][k] * x.v[k];
    y.v[i] = s;
  }
}

// 1 / x for a pivot: the hardware reciprocal refined by two Newton steps (within 1 ulp of the IEEE quotient, 5 instructions
// instead of ~14; a zero pivot gives inf either way and is reported through `bad`).  Float32 build: the plain division.
__device__ __forceinline__ real rcp_pivot(real x) {
#ifdef MOMS_FLOAT
  return 1.0f / x;
#else
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#endif
}

// X = inv(I - B) by Gauss-Jordan elimination with partial (row) pivoting; the row interchange is a select, every index
// is a compile-time constant after unrolling.  Returns 0 or k + 1 for a zero pivot at step k.
template <int N>
__device__ __forceinline__ int inv_one_minus(Mat<N> &X, const Mat<N> &B) {
  Mat<N> A;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      A.a[i][j] = ((i == j) ? 1.0 : 0.0) - B.a[i][j];
      X.a[i][j] = (i == j) ? 1.0 : 0.0;
    }
  int bad = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    real best = fabs(A.a[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const real v = fabs(A.a[i][k]);
      if (v > best) { best = v; p = i; }
    }
    // the interchange itself only if SOME lane of the wavefront needs one (a wave-uniform branch): I - B with ||B|| < 1 is
    // close to diagonally dominant and hardly ever pivots, while the selects of an unconditional interchange (192 v_cndmask
    // per 4 x 4 inverse) outnumber the elimination's own 128 FMAs
    if (__builtin_amdgcn_ballot_w64(p != k) != 0) {
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool sw = (p == i);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const real ak = A.a[k][j], ai = A.a[i][j], xk = X.a[k][j], xi = X.a[i][j];
          A.a[k][j] = sw ? ai : ak;
          A.a[i][j] = sw ? ak : ai;
          X.a[k][j] = sw ? xi : xk;
          X.a[i][j] = sw ? xk : xi;
        }
      }
    }
    if (!(best > 0.0) && !bad) bad = k + 1;
    const real d = rcp_pivot(A.a[k][k]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      A.a[k][j] *= d;
      X.a[k][j] *= d;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i != k) {
        const real f = A.a[i][k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          A.a[i][j] -= f * A.a[k][j];
          X.a[i][j] -= f * X.a[k][j];
        }
      }
  }
  return bad;
}

// X = T (I - B)^-1 without forming the inverse: Gauss-Jordan on (I - B)^T with T^T as the right-hand side (all indices are
// compile-time constants, so the transpositions are only names) -- the elimination costs what the inverse costs, the N^3 product
// that followed it is gone.  Partial pivoting over the columns of I - B; interchange only if some lane needs one (see above).
template <int N>
__device__ __forceinline__ int solve_right(Mat<N> &X, const Mat<N> &B, const Mat<N> &T) {
  Mat<N> A, R;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      A.a[i][j] = ((i == j) ? 1.0 : 0.0) - B.a[j][i];
      R.a[i][j] = T.a[j][i];
    }
  int bad = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    real best = fabs(A.a[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const real v = fabs(A.a[i][k]);
      if (v > best) { best = v; p = i; }
    }
    if (__builtin_amdgcn_ballot_w64(p != k) != 0) {
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool sw = (p == i);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const real ak = A.a[k][j], ai = A.a[i][j], xk = R.a[k][j], xi = R.a[i][j];
          A.a[k][j] = sw ? ai : ak;
          A.a[i][j] = sw ? ak : ai;
          R.a[k][j] = sw ? xi : xk;
          R.a[i][j] = sw ? xk : xi;
        }
      }
    }
    if (!(best > 0.0) && !bad) bad = k + 1;
    const real d = rcp_pivot(A.a[k][k]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      A.a[k][j] *= d;
      R.a[k][j] *= d;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i != k) {
        const real f = A.a[i][k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          A.a[i][j] -= f * A.a[k][j];
          R.a[i][j] -= f * R.a[k][j];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) X.a[i][j] = R.a[j][i];
  return bad;
}

using SweepArgs = ::MomSmallSweepArgsT<real>;  // mom_host.hpp: the one definition shared with momcore.hip / momcore_f32.hip

// Register budget (r5).  Up to r4 the N = 3, 4 images were built for two waves per SIMD (256 VGPRs) and SPILLED: k_sweep<4> 522
// VGPRs, 996 B of scratch per lane -- 6.5 GB of scratch traffic per 1.7 ms launch against 88 MB of algorithmic bytes (74 x; the
// kernel ran at the speed of its spill traffic, 3.8 TB/s, not of the FP64 pipe).  Now: ONE wave per SIMD for N >= 3 (512
// registers: the ~190 live doubles of a point fit, no scratch), and what made one wave per SIMD slower than two in r2 is gone:
//   * the per-(point, layer) inputs tau, varpi, tau_sum, zw of layer z + 1 are requested before layer z is computed (one wave
//     per SIMD has nobody to hide a dependent HBM round trip per layer behind);
//   * the view accumulators (48 doubles held across the whole kernel for at most 4 x 4 outputs) live in LDS, [3][nVza nStokes]
//     [thread] (dynamic: 6 KB for C1) -- NOT in the output arrays: a global store inside the loops costs every table read of the
//     kernel its scalar-cache path (the compiler can no longer prove the tables unclobbered: 201 s_load became 190 per-lane
//     global loads and the kernel ran 12 % slower than the spilling one; measured, profiles/r05_C1_ab.txt);
//   * r+- = D r-+ D and t-- = D t++ D are formed where they are consumed (sg = +-1: exact) instead of being held per layer.
#ifndef MOMS_WAVES_N34
#define MOMS_WAVES_N34 1  // waves per SIMD of the N = 3, 4 images (A/B: profiles/r05_C1_ab.txt)
#endif
//   * SPLIT (r5): one (point, MOMENT) per lane instead of one point.  The moments of a point are independent until the final sum
//     over m, so S M lanes of a third of the work each fill the GPU's 65 536 lane slots far more evenly than S lanes do (S =
//     2 10^5: 3.05 rounds of whole points = 4, against 9.16 rounds of thirds = 10: 1.2 x); each lane stores its term of R_SFI /
//     T_SFI after its last load, k_sum adds the terms in ascending m.
template <int N, bool SPLIT>
__global__ void __launch_bounds__(256, (N >= 3) ? MOMS_WAVES_N34 : 2) k_sweep(SweepArgs a) {
  // SPLIT: ceil(S / 256) workgroups per moment, so that m is a scalar (the Z bases of the moment stay on the scalar cache path)
  const int bpm = (a.S + (int)blockDim.x - 1) / (int)blockDim.x;
  const int m_lo = SPLIT ? (int)blockIdx.x / bpm : 0, m_hi = SPLIT ? m_lo + 1 : a.M;
  const int n = (SPLIT ? (int)blockIdx.x - m_lo * bpm : (int)blockIdx.x) * (int)blockDim.x + (int)threadIdx.x;
  if (n >= a.S) return;
  const int nS = a.nS, S = a.S, K = a.K;
  const int i_start = nS * (a.imu0 - 1), i_end = nS * a.imu0;
  int bad = 0;
  real bup[4] = {0, 0, 0, 0}, bdw[4] = {0, 0, 0, 0};
  extern __shared__ real acc_lds[];  // [R | T | hdr][nVza nS][blockDim.x] (unsplit kernel only)
  const int nslot = a.nVza * nS;
  real *accR = acc_lds + threadIdx.x, *accT = accR + (size_t)nslot * blockDim.x, *accH = accT + (size_t)nslot * blockDim.x;
  (void)accR; (void)accT; (void)accH;

  for (int m = m_lo; m < m_hi; ++m) {
    // wt / wdiv with wdiv = 2 (m = 0) or 4: the same value as wt * 0.5 or wt * 0.25 (a power of two: exact), without the division
    const real winv = (m == 0) ? 0.5 : 0.25, wct02 = (m == 0) ? 0.5 : 0.25;
    Mat<N> Rmp, Rpm, Tpp, Tmm;  // composite layer
    Vec<N> Jp, Jm;
    const real *Zp_m = a.Zpp + (size_t)N * N * K * m, *Zm_m = a.Zmp + (size_t)N * N * K * m;
    // inputs of layer 0; inside the loop: of layer z + 1 while layer z is computed
    real tau_n = a.tau[n], varpi_n = a.varpi[n], tsum_n = a.tau_sum[n], zw_n[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) zw_n[k] = (k < K) ? a.zw[k + (size_t)K * n] : 0.0;
    for (int z = 0; z < a.Nz; ++z) {
      const int nd = a.nd[z], iface = a.iface[z];
      const real tau = tau_n, varpi = varpi_n, tau_sum = tsum_n;
      real zw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) zw[k] = zw_n[k];
      if (z + 1 < a.Nz) {
        const size_t o1 = n + (size_t)S * (z + 1);
        tau_n = a.tau[o1]; varpi_n = a.varpi[o1]; tsum_n = a.tau_sum[o1];
#pragma unroll
        for (int k = 0; k < 4; ++k) zw_n[k] = (k < K) ? a.zw[k + (size_t)K * o1] : 0.0;
      }
      const real dtau = ldexp(tau, -nd);  // τ ./ 2^ndoubl (rt_kernel.jl:244)
      real expk = exp(-dtau / a.mu0);     // init_layer (rt_kernel.jl:273)
      // ---------------- elemental! (elemental.jl:164-253)
      Mat<N> r, t;
      Vec<N> jp, jm;
      {
        real ei[N];
#pragma unroll
        for (int i = 0; i < N; ++i) ei[i] = exp(-dtau / a.mu[i]);
        Vec<N> zpI, zmI;  // Z I0 over the sun's Stokes block (:225-228)
#pragma unroll
        for (int i = 0; i < N; ++i) zpI.v[i] = zmI.v[i] = 0.0;
        // E[i][j] = 1 - exp(-dtau (1/mu_i + 1/mu_j)): SI[i + N j] and SI[j + N i] are the same sum of the same two quotients, so
        // the N (N + 1) / 2 entries with i <= j are evaluated and mirrored (same operands, same value: 10 exponentials instead
        // of the 12 + 4 of the element loop and the j0- vector)
        Mat<N> E;
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int i = 0; i <= j; ++i) {
            const real e = 1 - exp(-dtau * a.SI[i + N * j]);
            E.a[i][j] = e;
            E.a[j][i] = e;
          }
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const real wj = a.wt[j] * winv;
#pragma unroll
          for (int i = 0; i < N; ++i) {
            real zp = 0.0, zm = 0.0;  // Z = sum_k w_k Z_k, in k order (types.jl:656-661)
            for (int k = 0; k < K; ++k) {
              zp += zw[k] * Zp_m[i + N * j + N * N * k];
              zm += zw[k] * Zm_m[i + N * j + N * N * k];
            }
            if (j >= i_start && j < i_end) {
              zpI.v[i] += zp * a.I0[j - i_start];
              zmI.v[i] += zm * a.I0[j - i_start];
            }
            real rr, tt;
            if (wj > 1.e-8) {
              rr = varpi * zm * a.F1[i + N * j] * wj * E.a[i][j];
              if (a.mu[i] == a.mu[j]) {
                tt = (i == j) ? ei[i] * (1 + varpi * zp * (dtau / a.mu[i]) * (a.wt[i] * winv)) : 0.0;
              } else {
                tt = varpi * zp * a.F2[i + N * j] * wj * (ei[i] - ei[j]);
              }
            } else {
              rr = 0.0;
              tt = (i == j) ? ei[i] : 0.0;
            }
            if (nd >= 1) rr *= a.sg[i];  // apply_D_elemental! (elemental.jl:265-269)
            r.a[i][j] = rr;
            t.a[i][j] = tt;
          }
        }
        const real mus = a.mu[i_start], att = exp(-tau_sum / mus);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const real mui = a.mu[i];
          real p, q;
          if (i >= i_start && i < i_end)
            p = wct02 * varpi * zpI.v[i] * (dtau / mui) * ei[i];
          else
            p = wct02 * varpi * zpI.v[i] * a.F2[i + N * i_start] * (ei[i] - ei[i_start]);
          real Eis = 0.0;  // E[i][i_start] (i_start is a run-time index)
#pragma unroll
          for (int j = 0; j < N; ++j)
            if (j == i_start) Eis = E.a[i][j];
          q = wct02 * varpi * zmI.v[i] * a.F1[i + N * i_start] * Eis;
          p *= att;
          q *= att;
          if (nd >= 1) q = a.D[i % nS] * q;  // elemental.jl:249-251
          jp.v[i] = p;
          jm.v[i] = q;
        }
      }
      // ---------------- doubling_helper! (doubling.jl:43-68) + apply_D! (:93-118)
      for (int it = 0; it < nd; ++it) {
        Mat<N> B, G, A, W;
        mul(B, r, r);
        const int e = solve_right(A, B, t);  // t (I - r r)^-1: tt++_gp_refl (doubling.jl:44-48)
        if (e && !bad) bad = e;
        Vec<N> j1p, j1m, v1, v2;
#pragma unroll
        for (int i = 0; i < N; ++i) { j1p.v[i] = jp.v[i] * expk; j1m.v[i] = jm.v[i] * expk; }
        mulv(v1, r, jp);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = j1m.v[i] + v1.v[i];
        mulv(v2, A, v1);
        mulv(v1, r, j1m);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = jp.v[i] + v1.v[i];  // OLD j0+ (:60)
#pragma unroll
        for (int i = 0; i < N; ++i) jm.v[i] = jm.v[i] + v2.v[i];  // :57
        mulv(v2, A, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) jp.v[i] = j1p.v[i] + v2.v[i];  // :60
        expk = expk * expk;                                        // :61
        mul(B, A, r);
        mul(W, B, t);  // (A r) t with the OLD t (:64)
        mul(G, A, t);  // :67
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j < N; ++j) {
            r.a[i][j] = r.a[i][j] + W.a[i][j];
            t.a[i][j] = G.a[i][j];
          }
      }
      if (nd >= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
          for (int j = 0; j < N; ++j) r.a[i][j] *= a.sg[i];
          jm.v[i] *= a.sg[i];
        }
      }
      // r+- = D r-+ D, t-- = D t++ D (elemental.jl:255-263 for nd < 1, doubling.jl:95-108 otherwise): formed where consumed
      auto dsd = [&](Mat<N> &o, const Mat<N> &x) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j < N; ++j) o.a[i][j] = (a.sg[i] * a.sg[j]) * x.a[i][j];
      };
      // ---------------- composite <- added (rt_kernel.jl:227-230) or interaction! (interaction.jl:8-117)
      if (z == 0) {
        Rmp = r; dsd(Rpm, r); Tpp = t; dsd(Tmm, t); Jp = jp; Jm = jm;
      } else if (iface == 0) {
        Vec<N> v1, v2;
        mulv(v1, t, Jp);
        mulv(v2, Tmm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) { Jp.v[i] = jp.v[i] + v1.v[i]; Jm.v[i] = Jm.v[i] + v2.v[i]; }
        Mat<N> W, tmm;
        dsd(tmm, t);
        mul(W, tmm, Tmm); Tmm = W;
        mul(W, t, Tpp); Tpp = W;
      } else if (iface == 1) {
        Vec<N> v1, v2;
        mulv(v1, r, Jp);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = v1.v[i] + jm.v[i];
        mulv(v2, Tmm, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jm.v[i] = Jm.v[i] + v2.v[i];
        mulv(v1, t, Jp);
#pragma unroll
        for (int i = 0; i < N; ++i) Jp.v[i] = jp.v[i] + v1.v[i];
        Mat<N> W1, W2;
        mul(W1, Tmm, r); mul(W2, W1, Tpp); Rmp = W2;
        dsd(Rpm, r);
        mul(W1, t, Tpp); Tpp = W1;
        dsd(W2, t);
        mul(W1, Tmm, W2); Tmm = W1;
      } else if (iface == 2) {
        Vec<N> v1, v2;
        mulv(v1, Rpm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = Jp.v[i] + v1.v[i];
        mulv(v2, t, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jp.v[i] = jp.v[i] + v2.v[i];
        mulv(v1, Tmm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) Jm.v[i] = Jm.v[i] + v1.v[i];
        Mat<N> W1, W2, tmm;
        dsd(tmm, t);
        mul(W1, t, Tpp); Tpp = W1;
        mul(W1, Tmm, tmm); Tmm = W1;
        mul(W1, t, Rpm); mul(W2, W1, tmm); Rpm = W2;
      } else {
        Mat<N> W1, W2, W3;
        Vec<N> v1, v2;
        mul(W1, r, Rpm);
        int e = solve_right(W3, W1, Tmm);  // T01 = T-- (I - r-+ R+-)^-1 (:81-87)
        if (e && !bad) bad = e;
        mulv(v1, r, Jp);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = v1.v[i] + jm.v[i];
        mulv(v2, W3, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jm.v[i] = Jm.v[i] + v2.v[i];  // :90
        mul(W1, W3, r); mul(W2, W1, Tpp);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j < N; ++j) Rmp.a[i][j] = Rmp.a[i][j] + W2.a[i][j];  // :93
        dsd(W2, t);
        mul(W1, W3, W2); Tmm = W1;                                           // :96
        mul(W1, Rpm, r);
        e = solve_right(W3, W1, t);  // T21 = t++ (I - R+- r-+)^-1 (:104-107)
        if (e && !bad) bad = e;
        mulv(v1, Rpm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = Jp.v[i] + v1.v[i];
        mulv(v2, W3, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jp.v[i] = jp.v[i] + v2.v[i];  // :110
        mul(W1, W3, Tpp); Tpp = W1;                              // :113
        mul(W1, W3, Rpm); dsd(W3, t); mul(W2, W1, W3);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j < N; ++j) Rpm.a[i][j] = (a.sg[i] * a.sg[j]) * r.a[i][j] + W2.a[i][j];  // :116
      }
    }
    // ---------------- Lambertian surface (m = 0) + closing interaction with the LAST layer's interface code (Q6)
    Vec<N> hdrJ;
#pragma unroll
    for (int i = 0; i < N; ++i) hdrJ.v[i] = 0.0;
    if (m == 0) {
      const int iface = a.iface[a.Nz - 1];
      const real rho = 2 * a.albedo;
      const real att = exp(-a.tau_sum[n + (size_t)S * a.Nz] / a.mu0);
      Mat<N> rs;
      Vec<N> jp, jm;
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) rs.a[i][j] = ((i % nS == 0) && (j % nS == 0)) ? rho * (a.mu[j] * a.wt[j]) : 0.0;
        const bool in_sun = (i >= i_start) && (i < i_end);
        jp.v[i] = (in_sun ? a.I0[i - i_start] : 0.0) * att;
        jm.v[i] = (i % nS == 0) ? (a.mu0 * (rho * a.I0[0])) * att : 0.0;
      }
      // the surface layer has t++ = t-- = I, r+- = 0: the four cases of interaction.jl with those operands
      if (iface == 0) {
        Vec<N> v2;
        mulv(v2, Tmm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) { Jp.v[i] = jp.v[i] + Jp.v[i]; Jm.v[i] = Jm.v[i] + v2.v[i]; }
      } else if (iface == 1) {
        Vec<N> v1, v2;
        mulv(v1, rs, Jp);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = v1.v[i] + jm.v[i];
        mulv(v2, Tmm, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) { Jm.v[i] = Jm.v[i] + v2.v[i]; Jp.v[i] = jp.v[i] + Jp.v[i]; }
      } else if (iface == 2) {
        Vec<N> v1;
        mulv(v1, Rpm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) Jp.v[i] = jp.v[i] + (Jp.v[i] + v1.v[i]);
        mulv(v1, Tmm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) Jm.v[i] = Jm.v[i] + v1.v[i];
      } else {
        Mat<N> W1, W2, W3;
        Vec<N> v1, v2;
        mul(W1, rs, Rpm);
        int e = solve_right(W3, W1, Tmm);  // T01
        if (e && !bad) bad = e;
        mulv(v1, rs, Jp);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = v1.v[i] + jm.v[i];
        mulv(v2, W3, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jm.v[i] = Jm.v[i] + v2.v[i];
        mul(W1, Rpm, rs);
        e = inv_one_minus(W2, W1);  // T21 = I (I - R+- r-+)^-1
        if (e && !bad) bad = e;
        mulv(v1, Rpm, jm);
#pragma unroll
        for (int i = 0; i < N; ++i) v1.v[i] = Jp.v[i] + v1.v[i];
        mulv(v2, W2, v1);
#pragma unroll
        for (int i = 0; i < N; ++i) Jp.v[i] = jp.v[i] + v2.v[i];
      }
      // interaction_hdrf! (interaction_hdrf.jl:9-45): hdr_J0- = r-+_surf J0+ + j0-_surf, and the m = 0 flux sums
      mulv(hdrJ, rs, Jp);
#pragma unroll
      for (int i = 0; i < N; ++i) hdrJ.v[i] = hdrJ.v[i] + jm.v[i];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nS) {
          real up = 0.0, dw = 0.0;
#pragma unroll
          for (int j = 0; j < N; ++j)
            if (j % nS == k) {
              up += hdrJ.v[j] * a.wt[j] * a.mu[j];
              dw += Jp.v[j] * a.wt[j] * a.mu[j];
            }
          bup[k] = up;
          bdw[k] = dw + jp.v[i_start] * a.mu[i_start];
        }
    }
    // ---------------- postprocessing_vza! / postprocessing_vza_hdrf! (postprocessing_vza.jl:9-93)
    const real weight = (m == 0) ? 0.5 : 1.0;
    for (int v = 0; v < a.nVza; ++v) {
      const int row0 = (a.node[v] - 1) * nS;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nS) {
          const real cs = weight * ((k < 2) ? a.cos_mphi[v + a.nVza * m] : a.sin_mphi[v + a.nVza * m]);
          real jmv = 0.0, jpv = 0.0, hv = 0.0;
#pragma unroll
          for (int i = 0; i < N; ++i)
            if (i == row0 + k) { jmv = Jm.v[i]; jpv = Jp.v[i]; hv = hdrJ.v[i]; }
          // the sum over the moments is kept in LDS (same order of additions as a register accumulator that starts at zero)
          if constexpr (SPLIT) {  // (the loop over m runs once: these stores follow every load of the lane)
            const size_t cnt = (size_t)a.nVza * nS * S, idx = v + (size_t)a.nVza * (k + (size_t)nS * n);
            a.part[(size_t)(2 * m) * cnt + idx] = cs * jmv;
            a.part[(size_t)(2 * m + 1) * cnt + idx] = cs * jpv;
            if (m == 0) a.hdr[idx] = cs * hv;
          } else {
            const int x = (v * nS + k) * blockDim.x;
            if (m == 0) {
              accR[x] = cs * jmv;
              accT[x] = cs * jpv;
              accH[x] = cs * hv;
            } else {
              accR[x] += cs * jmv;
              accT[x] += cs * jpv;
            }
          }
        }
    }
  }
  if constexpr (!SPLIT) {
    for (int v = 0; v < a.nVza; ++v)
      for (int k = 0; k < nS; ++k) {
        const size_t idx = v + (size_t)a.nVza * (k + (size_t)nS * n);
        const int x = (v * nS + k) * blockDim.x;
        a.R[idx] = accR[x];
        a.T[idx] = accT[x];
        a.hdr[idx] = accH[x];
      }
  }
  if (!SPLIT || m_lo == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nS) {
        a.bhr_uw[k + (size_t)nS * n] = bup[k];
        a.bhr_dw[k + (size_t)nS * n] = bdw[k];
      }
  }
  if (bad) atomicMax(a.info, bad);
}

// R_SFI / T_SFI = sum over the moments of the SPLIT kernel's terms, in ascending m from zero like the accumulator of the
// unsplit kernel (postprocessing_vza.jl:9-60 adds the moments into R_SFI in the order of rt_run.jl:125)
__global__ void k_sum(SweepArgs a) {
  const size_t cnt = (size_t)a.nVza * a.nS * a.S, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cnt) return;
  real r = 0.0, t = 0.0;
  for (int m = 0; m < a.M; ++m) {
    r += a.part[(size_t)(2 * m) * cnt + idx];
    t += a.part[(size_t)(2 * m + 1) * cnt + idx];
  }
  a.R[idx] = r;
  a.T[idx] = t;
}

}  // namespace MOMS_NS

// host entry used by momcore.hip (argument block = MOMS_NS::SweepArgs, passed as bytes)
#ifndef MOMS_FLOAT
size_t momsm_args_bytes() { return sizeof(MOMS_NS::SweepArgs); }
#endif
hipError_t MOMS_LAUNCH(const void *args, int N, hipStream_t st) {
  const MOMS_NS::SweepArgs a = *reinterpret_cast<const MOMS_NS::SweepArgs *>(args);
  const dim3 block(256);
  if (a.part != nullptr && a.M > 1) {  // one (point, moment) per lane, then the sum over the moments
    const dim3 grid((unsigned)(((size_t)a.S + 255) / 256 * a.M));
    switch (N) {
      case 1: hipLaunchKernelGGL((MOMS_NS::k_sweep<1, true>), grid, block, 0, st, a); break;
      case 2: hipLaunchKernelGGL((MOMS_NS::k_sweep<2, true>), grid, block, 0, st, a); break;
      case 3: hipLaunchKernelGGL((MOMS_NS::k_sweep<3, true>), grid, block, 0, st, a); break;
      case 4: hipLaunchKernelGGL((MOMS_NS::k_sweep<4, true>), grid, block, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
    const size_t cnt = (size_t)a.nVza * a.nS * a.S;
    hipLaunchKernelGGL(MOMS_NS::k_sum, dim3((unsigned)((cnt + 255) / 256)), block, 0, st, a);
    return hipGetLastError();
  }
  const dim3 grid((unsigned)((a.S + 255) / 256));
  // the view accumulators: up to 96 KB in Float64 at nVza = nS = 4 -- above the 64 KB a kernel gets without asking
  const size_t lds = (size_t)3 * a.nVza * a.nS * 256 * sizeof(real);
  hipError_t e = hipSuccess;
#define MOMS_UNSPLIT(NN)                                                                                         \
  if ((e = mom_allow_lds(reinterpret_cast<const void *>(MOMS_NS::k_sweep<NN, false>), lds)) != hipSuccess) return e; \
  hipLaunchKernelGGL((MOMS_NS::k_sweep<NN, false>), grid, block, lds, st, a);
  switch (N) {
    case 1: MOMS_UNSPLIT(1) break;
    case 2: MOMS_UNSPLIT(2) break;
    case 3: MOMS_UNSPLIT(3) break;
    case 4: MOMS_UNSPLIT(4) break;
    default: return hipErrorInvalidValue;
  }
#undef MOMS_UNSPLIT
  return hipGetLastError();
}
