/*
 * oracle/momref.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's (vSmartMOM.jl) elastic Matrix-Operator
 * hot path and of its Voigt line-shape kernel.  It exists to CHECK the HIP library
 * (tests/, __graft_entry__.smoke()) and to be timed as bench.py's `cpu_baseline`
 * ("port": the reference is Julia and no Julia toolchain exists in this image).
 * Nothing under radiativetransfer.jl_amd/ links, loads or calls this file.
 *
 * Pinning: the numpy twin oracle/momref.py reproduces the reference's own known-answer
 * tests (test/test_CoreRT.jl:3-83, Natraj + 6SV1 tables committed as
 * tests/golden/reference_tables.json); this C file is checked against that twin and
 * against the same tables in tests/test_oracle_*.py.
 *
 * Memory layout = the reference's: column-major [i,j,n] (i fastest, spectral index n
 * slowest, batch stride N*N), sources [i,1,n].  Each function cites the reference
 * file:line it follows (paths relative to the reference root, src/CoreRT/...).
 *
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -fPIC -shared oracle/momref.c -o oracle/libmomref.so -lm  (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define IDX(i, j, N) ((size_t)(i) + (size_t)(j) * (size_t)(N))

/* ---------------------------------------------------------------- dense helpers */

/* C = A*B, column-major N x N.  (`⊠` = NNlib.batched_mul, gpu_batched.jl:90-97)
 * Register-blocked micro-kernel (12 x 4 block of C in 12 AVX2 accumulators, k innermost and sequential, so every
 * element is the same left-to-right sum over k as the textbook triple loop; only the FMA contraction differs): the
 * CPU baseline of bench.py should not be an un-blocked loop. */
#ifdef ORA_EXT
/* extended-precision build (momref_ext.c): plain column sweeps, the same left-to-right sum over k per element */
static void gemm(int N, const double *A, const double *B, double *C) {
  for (int j = 0; j < N; ++j) {
    double *c = C + (size_t)j * N;
    for (int i = 0; i < N; ++i) c[i] = 0.0;
    for (int k = 0; k < N; ++k) {
      const double b = B[IDX(k, j, N)];
      const double *a = A + (size_t)k * N;
      for (int i = 0; i < N; ++i) c[i] += a[i] * b;
    }
  }
}
#else
typedef double v4d __attribute__((vector_size(32), aligned(8)));
static inline v4d ld4(const double *p) { return *(const v4d *)p; }
static inline void st4(double *p, v4d v) { *(v4d *)p = v; }
static void gemm(int N, const double *A, const double *B, double *C) {
  int j = 0;
  for (; j + 4 <= N; j += 4) {
    const double *b0 = B + (size_t)j * N, *b1 = b0 + N, *b2 = b1 + N, *b3 = b2 + N;
    int i = 0;
    for (; i + 12 <= N; i += 12) {
      v4d c00 = {0, 0, 0, 0}, c10 = c00, c20 = c00, c01 = c00, c11 = c00, c21 = c00, c02 = c00, c12 = c00, c22 = c00,
          c03 = c00, c13 = c00, c23 = c00;
      const double *a = A + i;
      for (int k = 0; k < N; ++k, a += N) {
        const v4d a0 = ld4(a), a1 = ld4(a + 4), a2 = ld4(a + 8);
        v4d b = {b0[k], b0[k], b0[k], b0[k]};
        c00 += a0 * b; c10 += a1 * b; c20 += a2 * b;
        b = (v4d){b1[k], b1[k], b1[k], b1[k]};
        c01 += a0 * b; c11 += a1 * b; c21 += a2 * b;
        b = (v4d){b2[k], b2[k], b2[k], b2[k]};
        c02 += a0 * b; c12 += a1 * b; c22 += a2 * b;
        b = (v4d){b3[k], b3[k], b3[k], b3[k]};
        c03 += a0 * b; c13 += a1 * b; c23 += a2 * b;
      }
      double *c = C + IDX(i, j, N);
      st4(c, c00); st4(c + 4, c10); st4(c + 8, c20); c += N;
      st4(c, c01); st4(c + 4, c11); st4(c + 8, c21); c += N;
      st4(c, c02); st4(c + 4, c12); st4(c + 8, c22); c += N;
      st4(c, c03); st4(c + 4, c13); st4(c + 8, c23);
    }
    for (; i + 4 <= N; i += 4) {
      v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      const double *a = A + i;
      for (int k = 0; k < N; ++k, a += N) {
        const v4d a0 = ld4(a);
        c0 += a0 * (v4d){b0[k], b0[k], b0[k], b0[k]};
        c1 += a0 * (v4d){b1[k], b1[k], b1[k], b1[k]};
        c2 += a0 * (v4d){b2[k], b2[k], b2[k], b2[k]};
        c3 += a0 * (v4d){b3[k], b3[k], b3[k], b3[k]};
      }
      double *c = C + IDX(i, j, N);
      st4(c, c0); st4(c + N, c1); st4(c + 2 * (size_t)N, c2); st4(c + 3 * (size_t)N, c3);
    }
    for (; i < N; ++i) {
      double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      for (int k = 0; k < N; ++k) {
        const double a = A[IDX(i, k, N)];
        c0 += a * b0[k]; c1 += a * b1[k]; c2 += a * b2[k]; c3 += a * b3[k];
      }
      C[IDX(i, j, N)] = c0; C[IDX(i, j + 1, N)] = c1; C[IDX(i, j + 2, N)] = c2; C[IDX(i, j + 3, N)] = c3;
    }
  }
  for (; j < N; ++j) {
    double *c = C + (size_t)j * N;
    for (int i = 0; i < N; ++i) c[i] = 0.0;
    for (int k = 0; k < N; ++k) {
      const double b = B[IDX(k, j, N)];
      const double *a = A + (size_t)k * N;
      for (int i = 0; i < N; ++i) c[i] += a[i] * b;
    }
  }
}

#endif

/* y = A*x */
static void gemv(int N, const double *A, const double *x, double *y) {
  for (int i = 0; i < N; ++i) y[i] = 0.0;
  for (int k = 0; k < N; ++k) {
    const double b = x[k];
    const double *a = A + (size_t)k * N;
    for (int i = 0; i < N; ++i) y[i] += a[i] * b;
  }
}

/* X = inv(A) by LU with partial pivoting (A\I: gpu_batched.jl:78-82 -> LAPACK getrf/getrs).
 * A is destroyed.  piv: N ints.  Returns 0, or k+1 if U[k,k]==0. */
static int inv_lu(int N, double *A, double *X, int *piv) {
  int info = 0;
  for (int k = 0; k < N; ++k) {
    int p = k;
    double mx = fabs(A[IDX(k, k, N)]);
    for (int i = k + 1; i < N; ++i) {
      double v = fabs(A[IDX(i, k, N)]);
      if (v > mx) { mx = v; p = i; }
    }
    piv[k] = p;
    if (p != k)
      for (int j = 0; j < N; ++j) {
        double tmp = A[IDX(k, j, N)]; A[IDX(k, j, N)] = A[IDX(p, j, N)]; A[IDX(p, j, N)] = tmp;
      }
    double d = A[IDX(k, k, N)];
    if (d == 0.0) { if (!info) info = k + 1; continue; }
    double rd = 1.0 / d;
    for (int i = k + 1; i < N; ++i) A[IDX(i, k, N)] *= rd;
    for (int j = k + 1; j < N; ++j) {
      double f = A[IDX(k, j, N)];
      if (f != 0.0)
        for (int i = k + 1; i < N; ++i) A[IDX(i, j, N)] -= A[IDX(i, k, N)] * f;
    }
  }
  /* X = P (row-permuted identity), then L\, U\ */
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) X[IDX(i, j, N)] = (i == j) ? 1.0 : 0.0;
  for (int k = 0; k < N; ++k)
    if (piv[k] != k)
      for (int j = 0; j < N; ++j) {
        double tmp = X[IDX(k, j, N)]; X[IDX(k, j, N)] = X[IDX(piv[k], j, N)]; X[IDX(piv[k], j, N)] = tmp;
      }
  for (int j = 0; j < N; ++j) {
    double *x = X + (size_t)j * N;
    for (int k = 0; k < N; ++k) {
      double f = x[k];
      if (f != 0.0)
        for (int i = k + 1; i < N; ++i) x[i] -= A[IDX(i, k, N)] * f;
    }
    for (int k = N - 1; k >= 0; --k) {
      x[k] /= A[IDX(k, k, N)];
      double f = x[k];
      for (int i = 0; i < k; ++i) x[i] -= A[IDX(i, k, N)] * f;
    }
  }
  return info;
}

/* Stokes-component label of 0-based stream index i.
 * strict: mod(i_1based, n) -> 1,2,..,n-1,0 ; else 1..n. */
static inline int scomp(int i0, int n, int strict) { return strict ? ((i0 + 1) % n) : (i0 % n) + 1; }

static inline double dsign(int ci, int cj) {
  return (((ci <= 2) && (cj <= 2)) || ((ci > 2) && (cj > 2))) ? 1.0 : -1.0;
}

/* ---------------------------------------------------------------- per-point ops */

typedef struct {
  int N, nS, imu0; /* imu0: 1-based stream (not Stokes-expanded) index of the sun */
  const double *mu, *wt, *I0, *D;
  int strict;
  double mu0; /* quad_points.μ₀ = cosd(sza) (types.jl:458); normally == mu[nS*(imu0-1)] */
} ora_streams;

/* elemental! elemental.jl:109-162; get_elem_rt! :164-207; get_elem_rt_SFI! :209-253;
 * apply_D_elemental! :255-274 (the SFI D kernel is a no-op, :276-307). */
static void elemental_pt(const ora_streams *q, int m, int nd, double tau_sum, double dtau, double varpi,
                         const double *Zpp, const double *Zmp, double *r_mp, double *t_pp, double *r_pm,
                         double *t_mm, double *j0p, double *j0m) {
  const int N = q->N, n = q->nS;
  const double *mu = q->mu;
  const double wct02 = (m == 0) ? 0.5 : 0.25;
  for (int j = 0; j < N; ++j) {
    const double wj = (m == 0) ? q->wt[j] / 2 : q->wt[j] / 4;
    for (int i = 0; i < N; ++i) {
      double r, t;
      if (wj > 1.e-8) {
        r = varpi * Zmp[IDX(i, j, N)] * (mu[j] / (mu[i] + mu[j])) * wj *
            (1 - exp(-dtau * ((1 / mu[i]) + (1 / mu[j]))));
        if (mu[i] == mu[j]) {
          if (i == j) {
            const double wi = (m == 0) ? q->wt[i] / 2 : q->wt[i] / 4;
            t = exp(-dtau / mu[i]) * (1 + varpi * Zpp[IDX(i, i, N)] * (dtau / mu[i]) * wi);
          } else
            t = 0.0;
        } else {
          t = varpi * Zpp[IDX(i, j, N)] * (mu[j] / (mu[i] - mu[j])) * wj *
              (exp(-dtau / mu[i]) - exp(-dtau / mu[j]));
        }
      } else {
        r = 0.0;
        t = (i == j) ? exp(-dtau / mu[i]) : 0.0;
      }
      r_mp[IDX(i, j, N)] = r;
      t_pp[IDX(i, j, N)] = t;
    }
  }
  const int i_start = n * (q->imu0 - 1), i_end = n * q->imu0; /* 0-based, end exclusive */
  const double mus = mu[i_start];
  for (int i = 0; i < N; ++i) {
    double zp = 0.0, zm = 0.0;
    for (int ii = i_start; ii < i_end; ++ii) {
      zp += Zpp[IDX(i, ii, N)] * q->I0[ii - i_start];
      zm += Zmp[IDX(i, ii, N)] * q->I0[ii - i_start];
    }
    double jp, jm;
    if (i >= i_start && i < i_end)
      jp = wct02 * varpi * zp * (dtau / mu[i]) * exp(-dtau / mu[i]);
    else
      jp = wct02 * varpi * zp * (mus / (mu[i] - mus)) * (exp(-dtau / mu[i]) - exp(-dtau / mus));
    jm = wct02 * varpi * zm * (mus / (mu[i] + mus)) * (1 - exp(-dtau * ((1 / mu[i]) + (1 / mus))));
    jp *= exp(-tau_sum / mus);
    jm *= exp(-tau_sum / mus);
    if (nd >= 1) jm = q->D[i % n] * jm;
    j0p[i] = jp;
    j0m[i] = jm;
  }
  if (nd < 1) {
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        double s = dsign(scomp(i, n, q->strict), scomp(j, n, q->strict));
        r_pm[IDX(i, j, N)] = s * r_mp[IDX(i, j, N)];
        t_mm[IDX(i, j, N)] = s * t_pp[IDX(i, j, N)];
      }
  } else {
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i)
        if (scomp(i, n, q->strict) > 2) r_mp[IDX(i, j, N)] = -r_mp[IDX(i, j, N)];
  }
}

/* doubling_helper! doubling.jl:13-79, apply_D! :93-110, apply_D_SFI! :112-118.
 * work: 4*N*N + 4*N doubles, piv: N ints.  Returns LU info (0 = ok). */
static int doubling_pt(const ora_streams *q, int nd, double *expk, double *r, double *t, double *r_pm,
                       double *t_mm, double *j0p, double *j0m, double *work, int *piv) {
  const int N = q->N, n = q->nS;
  if (nd == 0) return 0;
  const size_t NN = (size_t)N * N;
  double *tmp2 = work, *tmp1 = work + NN, *ttgp = work + 2 * NN, *tmp3 = work + 3 * NN;
  double *j1p = work + 4 * NN, *j1m = j1p + N, *v1 = j1m + N, *v2 = v1 + N;
  int info = 0;
  for (int it = 0; it < nd; ++it) {
    gemm(N, r, r, tmp2);
    for (size_t x = 0; x < NN; ++x) tmp2[x] = -tmp2[x];
    for (int i = 0; i < N; ++i) tmp2[IDX(i, i, N)] += 1.0; /* I_static .- r⊠r (:44) */
    int e = inv_lu(N, tmp2, tmp1, piv);                    /* :47 */
    if (e && !info) info = e;
    gemm(N, t, tmp1, ttgp);                                /* :48 */
    for (int i = 0; i < N; ++i) { j1p[i] = j0p[i] * (*expk); j1m[i] = j0m[i] * (*expk); } /* :51,:54 */
    gemv(N, r, j0p, v1);
    for (int i = 0; i < N; ++i) v1[i] = j1m[i] + v1[i];
    gemv(N, ttgp, v1, v2);
    gemv(N, r, j1m, v1);
    for (int i = 0; i < N; ++i) v1[i] = j0p[i] + v1[i]; /* uses OLD j0+ (:60) */
    for (int i = 0; i < N; ++i) j0m[i] = j0m[i] + v2[i]; /* :57 */
    gemv(N, ttgp, v1, v2);
    for (int i = 0; i < N; ++i) j0p[i] = j1p[i] + v2[i]; /* :60 */
    *expk = (*expk) * (*expk);                            /* :61 */
    gemm(N, ttgp, r, tmp2);
    gemm(N, tmp2, t, tmp3);                               /* (ttgp⊠r)⊠t, old t (:64) */
    for (size_t x = 0; x < NN; ++x) r[x] = r[x] + tmp3[x];
    gemm(N, ttgp, t, tmp3);                               /* :67 */
    memcpy(t, tmp3, NN * sizeof(double));
  }
  if (n == 1) { /* apply_D_matrix! :121-124 */
    memcpy(r_pm, r, NN * sizeof(double));
    memcpy(t_mm, t, NN * sizeof(double));
    return info;
  }
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      int ci = scomp(i, n, q->strict), cj = scomp(j, n, q->strict);
      if (ci > 2) r[IDX(i, j, N)] = -r[IDX(i, j, N)];
      double s = dsign(ci, cj);
      r_pm[IDX(i, j, N)] = s * r[IDX(i, j, N)];
      t_mm[IDX(i, j, N)] = s * t[IDX(i, j, N)];
    }
  for (int i = 0; i < N; ++i)
    if (scomp(i, n, q->strict) > 2) j0m[i] = -j0m[i];
  return info;
}

/* interaction_helper! interaction.jl:8-22 (00), :27-43 (01), :49-64 (10), :69-117 (11).
 * iface 0..3 = 00,01,10,11.  work: 4*N*N+2*N doubles. */
static int interaction_pt(int N, int iface, double *R_mp, double *R_pm, double *T_pp, double *T_mm,
                          double *J0p, double *J0m, const double *r_pm, const double *r_mp,
                          const double *t_mm, const double *t_pp, const double *j0p, const double *j0m,
                          double *work, int *piv) {
  const size_t NN = (size_t)N * N;
  double *W1 = work, *W2 = work + NN, *W3 = work + 2 * NN, *W4 = work + 3 * NN;
  double *v1 = work + 4 * NN, *v2 = v1 + N;
  int info = 0;
  if (iface == 0) {
    gemv(N, t_pp, J0p, v1);
    gemv(N, T_mm, j0m, v2);
    for (int i = 0; i < N; ++i) { J0p[i] = j0p[i] + v1[i]; J0m[i] = J0m[i] + v2[i]; }
    gemm(N, t_mm, T_mm, W1); memcpy(T_mm, W1, NN * sizeof(double));
    gemm(N, t_pp, T_pp, W1); memcpy(T_pp, W1, NN * sizeof(double));
  } else if (iface == 1) {
    gemv(N, r_mp, J0p, v1);
    for (int i = 0; i < N; ++i) v1[i] = v1[i] + j0m[i];
    gemv(N, T_mm, v1, v2);
    for (int i = 0; i < N; ++i) J0m[i] = J0m[i] + v2[i];
    gemv(N, t_pp, J0p, v1);
    for (int i = 0; i < N; ++i) J0p[i] = j0p[i] + v1[i];
    gemm(N, T_mm, r_mp, W1); gemm(N, W1, T_pp, W2); memcpy(R_mp, W2, NN * sizeof(double));
    memcpy(R_pm, r_pm, NN * sizeof(double));
    gemm(N, t_pp, T_pp, W1); memcpy(T_pp, W1, NN * sizeof(double));
    gemm(N, T_mm, t_mm, W1); memcpy(T_mm, W1, NN * sizeof(double));
  } else if (iface == 2) {
    gemv(N, R_pm, j0m, v1);
    for (int i = 0; i < N; ++i) v1[i] = J0p[i] + v1[i];
    gemv(N, t_pp, v1, v2);
    for (int i = 0; i < N; ++i) J0p[i] = j0p[i] + v2[i];
    gemv(N, T_mm, j0m, v1);
    for (int i = 0; i < N; ++i) J0m[i] = J0m[i] + v1[i];
    gemm(N, t_pp, T_pp, W1); memcpy(T_pp, W1, NN * sizeof(double));
    gemm(N, T_mm, t_mm, W1); memcpy(T_mm, W1, NN * sizeof(double));
    gemm(N, t_pp, R_pm, W1); gemm(N, W1, t_mm, W2); memcpy(R_pm, W2, NN * sizeof(double));
  } else {
    /* temp2 = I - r⁻⁺ ⊠ R⁺⁻ (:81); temp1 = inv (:83) */
    gemm(N, r_mp, R_pm, W1);
    for (size_t x = 0; x < NN; ++x) W1[x] = -W1[x];
    for (int i = 0; i < N; ++i) W1[IDX(i, i, N)] += 1.0;
    int e = inv_lu(N, W1, W2, piv); if (e && !info) info = e;
    gemm(N, T_mm, W2, W3); /* T01_inv (:87) */
    gemv(N, r_mp, J0p, v1);
    for (int i = 0; i < N; ++i) v1[i] = v1[i] + j0m[i];
    gemv(N, W3, v1, v2);
    for (int i = 0; i < N; ++i) J0m[i] = J0m[i] + v2[i]; /* :90 */
    gemm(N, W3, r_mp, W1); gemm(N, W1, T_pp, W2);
    for (size_t x = 0; x < NN; ++x) R_mp[x] = R_mp[x] + W2[x]; /* :93 */
    gemm(N, W3, t_mm, W1); memcpy(T_mm, W1, NN * sizeof(double)); /* :96 */
    gemm(N, R_pm, r_mp, W1);
    for (size_t x = 0; x < NN; ++x) W1[x] = -W1[x];
    for (int i = 0; i < N; ++i) W1[IDX(i, i, N)] += 1.0; /* :104 */
    e = inv_lu(N, W1, W2, piv); if (e && !info) info = e;
    gemm(N, t_pp, W2, W3); /* T21_inv (:107) */
    gemv(N, R_pm, j0m, v1);
    for (int i = 0; i < N; ++i) v1[i] = J0p[i] + v1[i];
    gemv(N, W3, v1, v2);
    for (int i = 0; i < N; ++i) J0p[i] = j0p[i] + v2[i]; /* :110 */
    gemm(N, W3, T_pp, W1); memcpy(T_pp, W1, NN * sizeof(double)); /* :113 */
    gemm(N, W3, R_pm, W1); gemm(N, W1, t_mm, W2);
    for (size_t x = 0; x < NN; ++x) R_pm[x] = r_pm[x] + W2[x]; /* :116 */
  }
  (void)W4;
  return info;
}

/* create_surface_layer!(::LambertianSurfaceScalar) lambertian_surface.jl:20-75 */
static void surface_lambertian_pt(const ora_streams *q, int m, double albedo, double tau_tot, double *r_pm,
                                  double *r_mp, double *t_mm, double *t_pp, double *j0p, double *j0m) {
  const int N = q->N, n = q->nS;
  const size_t NN = (size_t)N * N;
  const double mu0 = q->mu0;
  for (size_t x = 0; x < NN; ++x) { t_pp[x] = 0; t_mm[x] = 0; r_mp[x] = 0; }
  for (int i = 0; i < N; ++i) { t_pp[IDX(i, i, N)] = 1.0; t_mm[IDX(i, i, N)] = 1.0; }
  if (m == 0) {
    const double rho = 2 * albedo;
    const double att = exp(-tau_tot / mu0);
    for (int i = 0; i < N; ++i) {
      int in_sun = (i >= n * (q->imu0 - 1)) && (i < n * q->imu0);
      double I0N = in_sun ? q->I0[i - n * (q->imu0 - 1)] : 0.0;
      j0p[i] = I0N * att;
    }
    /* R_surf*I0N: row i gets rho * sum_{j: comp 0} I0N[j] if i is an I row */
    double sI = 0.0;
    for (int j = 0; j < N; j += n) {
      int in_sun = (j >= n * (q->imu0 - 1)) && (j < n * q->imu0);
      sI += rho * (in_sun ? q->I0[0] : 0.0);
    }
    for (int i = 0; i < N; ++i) j0m[i] = (i % n == 0) ? (mu0 * sI) * att : (mu0 * 0.0) * att;
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i)
        r_mp[IDX(i, j, N)] = ((i % n == 0) && (j % n == 0)) ? rho * (q->mu[j] * q->wt[j]) : 0.0;
    for (size_t x = 0; x < NN; ++x) r_pm[x] = 0;
  } else {
    for (int i = 0; i < N; ++i) { j0p[i] = 0; j0m[i] = 0; }
    /* r⁺⁻ deliberately NOT reset (lambertian_surface.jl:68-69 resets r⁻⁺ twice) */
  }
}

/* create_surface_layer!(brdf::AbstractSurfaceType, ...) rpv_surface.jl:20-66 for ONE Fourier moment:
 * Rs = rho_m [N,N] column-major (reflectance(brdf, pol_type, qp_mu, m), x2 for m = 0, computed by the host),
 * j0+ = I0N e^{-tau/mu0}, j0- = mu0 (Rs I0N) e^{-tau/mu0}, r-+ = Rs Diagonal(mu .* wt), r+- = 0, t = I. */
static void surface_brdf_pt(const ora_streams *q, const double *Rs, double tau_tot, double *r_pm, double *r_mp,
                            double *t_mm, double *t_pp, double *j0p, double *j0m) {
  const int N = q->N, n = q->nS;
  const size_t NN = (size_t)N * N;
  const int i0 = n * (q->imu0 - 1);
  const double att = exp(-tau_tot / q->mu0);
  for (size_t x = 0; x < NN; ++x) { t_pp[x] = 0; t_mm[x] = 0; r_pm[x] = 0; }
  for (int i = 0; i < N; ++i) { t_pp[IDX(i, i, N)] = 1.0; t_mm[IDX(i, i, N)] = 1.0; }
  for (int i = 0; i < N; ++i) {
    const int in_sun = (i >= i0) && (i < n * q->imu0);
    j0p[i] = (in_sun ? q->I0[i - i0] : 0.0) * att;
    double rI = 0.0;
    for (int k = 0; k < n; ++k) rI += Rs[IDX(i, i0 + k, N)] * q->I0[k];
    j0m[i] = (q->mu0 * rI) * att;
  }
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) r_mp[IDX(i, j, N)] = Rs[IDX(i, j, N)] * (q->mu[j] * q->wt[j]);
}

/* create_surface_layer!(::LambertianSurfaceLegendre) lambertian_surface.jl:77-138: albedo = P * legendre_coeff per
 * spectral point (host); m = 0: j0+ = 0 (:112), j0- = mu0 (R_surf I0N) (rho e^{-tau/mu0}) (:114); m > 0: everything 0
 * INCLUDING t++ and t-- (:127-134). */
static void surface_legendre_pt(const ora_streams *q, int m, double albedo, double tau_tot, double *r_pm, double *r_mp,
                                double *t_mm, double *t_pp, double *j0p, double *j0m) {
  const int N = q->N, n = q->nS;
  const size_t NN = (size_t)N * N;
  for (size_t x = 0; x < NN; ++x) { t_pp[x] = 0; t_mm[x] = 0; r_mp[x] = 0; }
  for (int i = 0; i < N; ++i) { j0p[i] = 0; j0m[i] = 0; }
  if (m != 0) return; /* r+- is not touched for m > 0 (it still holds the m = 0 zeros) */
  const double rho = 2 * albedo, att = exp(-tau_tot / q->mu0);
  for (size_t x = 0; x < NN; ++x) r_pm[x] = 0;
  for (int i = 0; i < N; ++i) { t_pp[IDX(i, i, N)] = 1.0; t_mm[IDX(i, i, N)] = 1.0; }
  for (int i = 0; i < N; ++i) j0m[i] = (i % n == 0) ? (q->mu0 * q->I0[0]) * (rho * att) : 0.0;
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i)
      r_mp[IDX(i, j, N)] = ((i % n == 0) && (j % n == 0)) ? rho * (q->mu[j] * q->wt[j]) : 0.0;
}

/* ---------------------------------------------------------------- batched op-level API */

void ora_elemental(int N, int nS, int S, int m, int nd, int imu0, const double *mu, const double *wt,
                   const double *I0, const double *D, int strict, const double *tau_sum, const double *dtau,
                   const double *varpi, const double *Zpp, const double *Zmp, int z_batch, double *r_pm,
                   double *r_mp, double *t_mm, double *t_pp, double *j0p, double *j0m) {
  ora_streams q = {N, nS, imu0, mu, wt, I0, D, strict, mu[nS * (imu0 - 1)]};
  const size_t NN = (size_t)N * N;
#pragma omp parallel for schedule(static)
  for (int s = 0; s < S; ++s) {
    const size_t zo = z_batch > 1 ? NN * s : 0;
    elemental_pt(&q, m, nd, tau_sum[s], dtau[s], varpi[s], Zpp + zo, Zmp + zo, r_mp + NN * s, t_pp + NN * s,
                 r_pm + NN * s, t_mm + NN * s, j0p + (size_t)N * s, j0m + (size_t)N * s);
  }
}

int ora_doubling(int N, int nS, int S, int nd, int strict, double *expk, double *r_pm, double *r_mp,
                 double *t_mm, double *t_pp, double *j0p, double *j0m) {
  ora_streams q = {N, nS, 1, 0, 0, 0, 0, strict, 1.0};
  const size_t NN = (size_t)N * N;
  int info = 0;
#pragma omp parallel
  {
    double *work = (double *)malloc((4 * NN + 4 * N) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
#pragma omp for schedule(static)
    for (int s = 0; s < S; ++s) {
      int e = doubling_pt(&q, nd, expk + s, r_mp + NN * s, t_pp + NN * s, r_pm + NN * s, t_mm + NN * s,
                          j0p + (size_t)N * s, j0m + (size_t)N * s, work, piv);
      if (e) {
#pragma omp atomic write
        info = e;
      }
    }
    free(work); free(piv);
  }
  return info;
}

int ora_interaction(int N, int S, int iface, double *R_mp, double *R_pm, double *T_pp, double *T_mm,
                    double *J0p, double *J0m, const double *r_pm, const double *r_mp, const double *t_mm,
                    const double *t_pp, const double *j0p, const double *j0m) {
  const size_t NN = (size_t)N * N;
  int info = 0;
#pragma omp parallel
  {
    double *work = (double *)malloc((4 * NN + 2 * N) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
#pragma omp for schedule(static)
    for (int s = 0; s < S; ++s) {
      int e = interaction_pt(N, iface, R_mp + NN * s, R_pm + NN * s, T_pp + NN * s, T_mm + NN * s,
                             J0p + (size_t)N * s, J0m + (size_t)N * s, r_pm + NN * s, r_mp + NN * s,
                             t_mm + NN * s, t_pp + NN * s, j0p + (size_t)N * s, j0m + (size_t)N * s, work, piv);
      if (e) {
#pragma omp atomic write
        info = e;
      }
    }
    free(work); free(piv);
  }
  return info;
}

void ora_surface_lambertian(int N, int nS, int S, int m, int imu0, double mu0, const double *mu, const double *wt,
                            const double *I0, double albedo, const double *tau_tot, double *r_pm, double *r_mp,
                            double *t_mm, double *t_pp, double *j0p, double *j0m) {
  ora_streams q = {N, nS, imu0, mu, wt, I0, 0, 1, mu0};
  const size_t NN = (size_t)N * N;
#pragma omp parallel for schedule(static)
  for (int s = 0; s < S; ++s)
    surface_lambertian_pt(&q, m, albedo, tau_tot[s], r_pm + NN * s, r_mp + NN * s, t_mm + NN * s, t_pp + NN * s,
                          j0p + (size_t)N * s, j0m + (size_t)N * s);
}

/* batched inverse and gemm, exported for per-op tests */
int ora_batch_inv(int N, int S, const double *A, double *X) {
  const size_t NN = (size_t)N * N;
  int info = 0;
#pragma omp parallel
  {
    double *w = (double *)malloc(NN * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
#pragma omp for schedule(static)
    for (int s = 0; s < S; ++s) {
      memcpy(w, A + NN * s, NN * sizeof(double));
      int e = inv_lu(N, w, X + NN * s, piv);
      if (e) {
#pragma omp atomic write
        info = e;
      }
    }
    free(w); free(piv);
  }
  return info;
}

void ora_batched_mul(int N, int S, const double *A, const double *B, double *C) {
  const size_t NN = (size_t)N * N;
#pragma omp parallel for schedule(static)
  for (int s = 0; s < S; ++s) gemm(N, A + NN * s, B + NN * s, C + NN * s);
}

/* ---------------------------------------------------------------- full run (rt_run.jl:41-230) */

typedef struct {
  int N, nS, S, Nz, K, M, imu0, strict;
  double mu0;
  const double *mu, *wt, *I0, *D;
  const double *tau;      /* [S,Nz] column-major: tau[n + S*z]          */
  const double *varpi;    /* [S,Nz]                                      */
  const double *zw;       /* [K,S,Nz]: zw[k + K*(n + S*z)]               */
  const double *Zpp;      /* [N,N,K,M]                                   */
  const double *Zmp;      /* [N,N,K,M]                                   */
  const int *nd;          /* [Nz] (global over the spectral axis)        */
  const int *iface;       /* [Nz] 0..3                                   */
  const double *tau_sum;  /* [S,Nz+1]                                    */
  double albedo;
  int nVza;
  const int *node;        /* [nVza] 1-based stream index nearest to vza  */
  const double *cos_mphi; /* [nVza,M] cosd(m*vaz): cos_mphi[v + nVza*m]  */
  const double *sin_mphi; /* [nVza,M]                                    */
  int surf_kind;          /* 0 LambertianSurfaceScalar(albedo), 1 BRDF matrices, 2 LambertianSurfaceLegendre */
  const double *Rsurf;    /* kind 1: [N,N,M] rho_m, x2 for m = 0 included  */
  const double *albedo_spec; /* kind 2: [S]                                */
} ora_scene;

/* R_SFI,T_SFI: [nVza,nStokes,S] column-major, zero-initialised by the caller.
 * pts: optional list of spectral indices to process (npts entries) or NULL for all.
 * Returns first nonzero LU info. */
int ora_rt_run_full(const ora_scene *sc, const int *pts, int npts, int nthreads, double *R_SFI, double *T_SFI,
                    double *hdr, double *bhr_uw, double *bhr_dw);
int ora_rt_run(const ora_scene *sc, const int *pts, int npts, int nthreads, double *R_SFI, double *T_SFI) {
  return ora_rt_run_full(sc, pts, npts, nthreads, R_SFI, T_SFI, 0, 0, 0);
}

/* hdr [nVza,nStokes,S], bhr_uw/bhr_dw [nStokes,S]: interaction_hdrf! (interaction_hdrf.jl:9-45) and
 * postprocessing_vza_hdrf! (postprocessing_vza.jl:63-93); may be NULL. */
int ora_rt_run_full(const ora_scene *sc, const int *pts, int npts, int nthreads, double *R_SFI, double *T_SFI,
                    double *hdr, double *bhr_uw, double *bhr_dw) {
  const int N = sc->N, n = sc->nS, S = sc->S, Nz = sc->Nz, K = sc->K, M = sc->M;
  const size_t NN = (size_t)N * N;
  ora_streams q = {N, n, sc->imu0, sc->mu, sc->wt, sc->I0, sc->D, sc->strict, sc->mu0};
  const int count = pts ? npts : S;
  int info = 0;
  (void)nthreads;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    double *buf = (double *)malloc((16 * NN + 16 * N) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
    double *Zp = buf, *Zm = buf + NN;
    double *a_rpm = buf + 2 * NN, *a_rmp = buf + 3 * NN, *a_tmm = buf + 4 * NN, *a_tpp = buf + 5 * NN;
    double *c_Rmp = buf + 6 * NN, *c_Rpm = buf + 7 * NN, *c_Tpp = buf + 8 * NN, *c_Tmm = buf + 9 * NN;
    double *work = buf + 10 * NN; /* 4*NN + 4*N */
    double *vec = buf + 14 * NN + 4 * N;
    double *a_j0p = vec, *a_j0m = vec + N, *c_J0p = vec + 2 * N, *c_J0m = vec + 3 * N;
    double *s_rpm = buf + 15 * NN + 12 * N; /* surface r⁺⁻ persists across m (see surface_lambertian_pt) */
#pragma omp for schedule(dynamic, 1)
    for (int ip = 0; ip < count; ++ip) {
      const int s = pts ? pts[ip] : ip;
      for (size_t x = 0; x < NN; ++x) s_rpm[x] = 0.0;
      for (int m = 0; m < M; ++m) {
        const double weight = (m == 0) ? 0.5 : 1.0;
        for (int z = 0; z < Nz; ++z) {
          const double tau = sc->tau[s + (size_t)S * z], varpi = sc->varpi[s + (size_t)S * z];
          const int nd = sc->nd[z];
          const double dtau = tau / ldexp(1.0, nd);
          double expk = exp(-dtau / sc->mu0); /* init_layer rt_kernel.jl:269-275 */
          /* Z[:,:] = sum_k w_k Z_k  (types.jl:656-661 chained mix carried on the weights) */
          for (size_t x = 0; x < NN; ++x) { Zp[x] = 0; Zm[x] = 0; }
          for (int k = 0; k < K; ++k) {
            const double w = sc->zw[k + (size_t)K * (s + (size_t)S * z)];
            const double *bp = sc->Zpp + NN * (k + (size_t)K * m), *bm = sc->Zmp + NN * (k + (size_t)K * m);
            for (size_t x = 0; x < NN; ++x) { Zp[x] += w * bp[x]; Zm[x] += w * bm[x]; }
          }
          elemental_pt(&q, m, nd, sc->tau_sum[s + (size_t)S * z], dtau, varpi, Zp, Zm, a_rmp, a_tpp, a_rpm, a_tmm,
                       a_j0p, a_j0m);
          int e = doubling_pt(&q, nd, &expk, a_rmp, a_tpp, a_rpm, a_tmm, a_j0p, a_j0m, work, piv);
          if (z == 0) { /* rt_kernel.jl:227-230 */
            memcpy(c_Tpp, a_tpp, NN * sizeof(double)); memcpy(c_Tmm, a_tmm, NN * sizeof(double));
            memcpy(c_Rmp, a_rmp, NN * sizeof(double)); memcpy(c_Rpm, a_rpm, NN * sizeof(double));
            memcpy(c_J0p, a_j0p, N * sizeof(double)); memcpy(c_J0m, a_j0m, N * sizeof(double));
          } else {
            int e2 = interaction_pt(N, sc->iface[z], c_Rmp, c_Rpm, c_Tpp, c_Tmm, c_J0p, c_J0m, a_rpm, a_rmp, a_tmm,
                                    a_tpp, a_j0p, a_j0m, work, piv);
            if (e2 && !e) e = e2;
          }
          if (e) {
#pragma omp atomic write
            info = e;
          }
        }
        /* surface: rt_run.jl:169-185 (interface code of the LAST layer, Q6) */
        if (sc->surf_kind == 1)
          surface_brdf_pt(&q, sc->Rsurf + NN * m, sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp, a_j0p, a_j0m);
        else if (sc->surf_kind == 2)
          surface_legendre_pt(&q, m, sc->albedo_spec[s], sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp,
                              a_j0p, a_j0m);
        else
          surface_lambertian_pt(&q, m, sc->albedo, sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp, a_j0p,
                                a_j0m);
        int e3 = interaction_pt(N, sc->iface[Nz - 1], c_Rmp, c_Rpm, c_Tpp, c_Tmm, c_J0p, c_J0m, s_rpm, a_rmp, a_tmm,
                                a_tpp, a_j0p, a_j0m, work, piv);
        if (e3) {
#pragma omp atomic write
          info = e3;
        }
        /* interaction_hdrf!: hdr_J0- = r-+_surf J0+ + j0-_surf (work[0..N)) */
        double *hdrJ = work;
        if (hdr) {
          gemv(N, a_rmp, c_J0p, hdrJ);
          for (int i = 0; i < N; ++i) hdrJ[i] += a_j0m[i];
          if (m == 0) {
            const int i0 = n * (sc->imu0 - 1);
            for (int k = 0; k < n; ++k) {
              double up = 0.0, dw = 0.0;
              for (int j = k; j < N; j += n) { up += hdrJ[j] * sc->wt[j] * sc->mu[j]; dw += c_J0p[j] * sc->wt[j] * sc->mu[j]; }
              bhr_uw[k + (size_t)n * s] = up;
              bhr_dw[k + (size_t)n * s] = dw + a_j0p[i0] * sc->mu[i0];
            }
          }
        }
        /* postprocessing_vza! postprocessing_vza.jl:9-60 (SFI branch) and _hdrf! :63-93 */
        for (int v = 0; v < sc->nVza; ++v) {
          const int istart = (sc->node[v] - 1) * n;
          const double c = sc->cos_mphi[v + (size_t)sc->nVza * m], sn = sc->sin_mphi[v + (size_t)sc->nVza * m];
          for (int k = 0; k < n; ++k) {
            const double cs = weight * ((k < 2) ? c : sn);
            const size_t o = v + (size_t)sc->nVza * (k + (size_t)n * s);
            R_SFI[o] += cs * c_J0m[istart + k];
            T_SFI[o] += cs * c_J0p[istart + k];
            if (hdr) hdr[o] += cs * hdrJ[istart + k];
          }
        }
      }
    }
    free(buf); free(piv);
  }
  return info;
}

/* rt_run_test_ms(::noRS, sensor_levels, model, iBand) rt_run_multisensor.jl:14-191: rt_kernel_multisensor!(::noRS)
 * (rt_kernel_multisensor.jl:2-113: which composite -- above (top) or below (bot) sensor ims -- the added layer of iz
 * joins), the surface interaction with every bottom composite (rt_run_multisensor.jl:150-159), interlayer_flux_helper!
 * (interlayer_flux.jl:7-24) and postprocessing_vza_ms!(::noRS) (postprocessing_vza_ms.jl:9-77).
 * levels[ims] in 0..Nz-1: 0 = the TOA/BOA pair, L >= 1 = below layer L (counted from the top).
 * uwJ, dwJ: [nVza,nStokes,S,nSens] column-major, zero-initialised by the caller. */
int ora_rt_run_ms(const ora_scene *sc, const int *pts, int npts, int nthreads, int nSens, const int *levels, double *uwJ,
                  double *dwJ) {
  const int N = sc->N, n = sc->nS, S = sc->S, Nz = sc->Nz, K = sc->K, M = sc->M;
  const size_t NN = (size_t)N * N, CS = 4 * NN + 2 * (size_t)N; /* one composite: R-+, R+-, T++, T--, J0+, J0- */
  ora_streams q = {N, n, sc->imu0, sc->mu, sc->wt, sc->I0, sc->D, sc->strict, sc->mu0};
  const int count = pts ? npts : S;
  int info = 0;
  for (int i = 0; i < nSens; ++i)
    if (levels[i] < 0 || levels[i] >= Nz) return -1;
  (void)nthreads;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    double *buf = (double *)malloc((12 * NN + 16 * N + 2 * (size_t)nSens * CS) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
    double *Zp = buf, *Zm = buf + NN;
    double *a_rpm = buf + 2 * NN, *a_rmp = buf + 3 * NN, *a_tmm = buf + 4 * NN, *a_tpp = buf + 5 * NN;
    double *work = buf + 6 * NN; /* 4*NN + 4*N */
    double *vec = buf + 10 * NN + 4 * N;
    double *a_j0p = vec, *a_j0m = vec + N, *tdw = vec + 2 * N, *tuw = vec + 3 * N;
    double *s_rpm = buf + 11 * NN + 12 * N;
    double *comps = buf + 12 * NN + 16 * N; /* [top, bot] x nSens */
#define MS_COMP(which, ims, k) (comps + ((size_t)(2 * (ims) + (which))) * CS + ((k) < 4 ? (size_t)(k) * NN : 4 * NN + (size_t)((k) - 4) * N))
#pragma omp for schedule(dynamic, 1)
    for (int ip = 0; ip < count; ++ip) {
      const int s = pts ? pts[ip] : ip;
      for (size_t x = 0; x < NN; ++x) s_rpm[x] = 0.0;
      for (int m = 0; m < M; ++m) {
        const double weight = (m == 0) ? 0.5 : 1.0;
        int e = 0;
        for (int z = 0; z < Nz; ++z) {
          const int iz = z + 1;
          const double tau = sc->tau[s + (size_t)S * z], varpi = sc->varpi[s + (size_t)S * z];
          const int nd = sc->nd[z];
          const double dtau = tau / ldexp(1.0, nd);
          double expk = exp(-dtau / sc->mu0);
          for (size_t x = 0; x < NN; ++x) { Zp[x] = 0; Zm[x] = 0; }
          for (int k = 0; k < K; ++k) {
            const double w = sc->zw[k + (size_t)K * (s + (size_t)S * z)];
            const double *bp = sc->Zpp + NN * (k + (size_t)K * m), *bm = sc->Zmp + NN * (k + (size_t)K * m);
            for (size_t x = 0; x < NN; ++x) { Zp[x] += w * bp[x]; Zm[x] += w * bm[x]; }
          }
          elemental_pt(&q, m, nd, sc->tau_sum[s + (size_t)S * z], dtau, varpi, Zp, Zm, a_rmp, a_tpp, a_rpm, a_tmm,
                       a_j0p, a_j0m);
          int e1 = doubling_pt(&q, nd, &expk, a_rmp, a_tpp, a_rpm, a_tmm, a_j0p, a_j0m, work, piv);
          if (e1 && !e) e = e1;
          for (int ims = 0; ims < nSens; ++ims) {
            const int L = levels[ims];
            int which, copy; /* which: 0 top, 1 bot */
            if (iz == 1) { which = (L == 0) ? 1 : 0; copy = 1; }
            else if (L == 0) { which = 1; copy = 0; }
            else if (L == iz - 1) { which = 1; copy = 1; }
            else if (L < iz - 1) { which = 1; copy = 0; }
            else { which = 0; copy = 0; }
            double *cR_mp = MS_COMP(which, ims, 0), *cR_pm = MS_COMP(which, ims, 1), *cT_pp = MS_COMP(which, ims, 2),
                   *cT_mm = MS_COMP(which, ims, 3), *cJ0p = MS_COMP(which, ims, 4), *cJ0m = MS_COMP(which, ims, 5);
            if (copy) {
              memcpy(cT_pp, a_tpp, NN * sizeof(double)); memcpy(cT_mm, a_tmm, NN * sizeof(double));
              memcpy(cR_mp, a_rmp, NN * sizeof(double)); memcpy(cR_pm, a_rpm, NN * sizeof(double));
              memcpy(cJ0p, a_j0p, N * sizeof(double)); memcpy(cJ0m, a_j0m, N * sizeof(double));
            } else {
              int e2 = interaction_pt(N, sc->iface[z], cR_mp, cR_pm, cT_pp, cT_mm, cJ0p, cJ0m, a_rpm, a_rmp, a_tmm, a_tpp,
                                      a_j0p, a_j0m, work, piv);
              if (e2 && !e) e = e2;
            }
          }
        }
        if (sc->surf_kind == 1)
          surface_brdf_pt(&q, sc->Rsurf + NN * m, sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp, a_j0p, a_j0m);
        else if (sc->surf_kind == 2)
          surface_legendre_pt(&q, m, sc->albedo_spec[s], sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp,
                              a_j0p, a_j0m);
        else
          surface_lambertian_pt(&q, m, sc->albedo, sc->tau_sum[s + (size_t)S * Nz], s_rpm, a_rmp, a_tmm, a_tpp, a_j0p,
                                a_j0m);
        for (int ims = 0; ims < nSens; ++ims) {
          int e3 = interaction_pt(N, sc->iface[Nz - 1], MS_COMP(1, ims, 0), MS_COMP(1, ims, 1), MS_COMP(1, ims, 2),
                                  MS_COMP(1, ims, 3), MS_COMP(1, ims, 4), MS_COMP(1, ims, 5), s_rpm, a_rmp, a_tmm, a_tpp,
                                  a_j0p, a_j0m, work, piv);
          if (e3 && !e) e = e3;
        }
        for (int ims = 0; ims < nSens; ++ims) {
          const double *uw, *dw;
          if (levels[ims] == 0) {
            uw = MS_COMP(1, ims, 5); dw = MS_COMP(1, ims, 4);
          } else {
            /* interlayer_flux.jl:14-23 */
            const double *tR_pm = MS_COMP(0, ims, 1), *bR_mp = MS_COMP(1, ims, 0), *tJ0p = MS_COMP(0, ims, 4),
                         *bJ0m = MS_COMP(1, ims, 5);
            double *W1 = work, *W2 = work + NN, *v1 = work + 4 * NN, *v2 = v1 + N;
            gemm(N, tR_pm, bR_mp, W1);
            for (size_t x = 0; x < NN; ++x) W1[x] = -W1[x];
            for (int i = 0; i < N; ++i) W1[IDX(i, i, N)] += 1.0;
            int e4 = inv_lu(N, W1, W2, piv); if (e4 && !e) e = e4;
            gemv(N, tR_pm, bJ0m, v1);
            for (int i = 0; i < N; ++i) v1[i] = tJ0p[i] + v1[i];
            gemv(N, W2, v1, tdw);
            gemm(N, bR_mp, tR_pm, W1);
            for (size_t x = 0; x < NN; ++x) W1[x] = -W1[x];
            for (int i = 0; i < N; ++i) W1[IDX(i, i, N)] += 1.0;
            e4 = inv_lu(N, W1, W2, piv); if (e4 && !e) e = e4;
            gemv(N, bR_mp, tJ0p, v2);
            for (int i = 0; i < N; ++i) v2[i] = bJ0m[i] + v2[i];
            gemv(N, W2, v2, tuw);
            uw = tuw; dw = tdw;
          }
          for (int v = 0; v < sc->nVza; ++v) {
            const int istart = (sc->node[v] - 1) * n;
            const double c = sc->cos_mphi[v + (size_t)sc->nVza * m], sn = sc->sin_mphi[v + (size_t)sc->nVza * m];
            for (int k = 0; k < n; ++k) {
              const double cs = weight * ((k < 2) ? c : sn);
              const size_t o = v + (size_t)sc->nVza * (k + (size_t)n * (s + (size_t)S * ims));
              uwJ[o] += cs * uw[istart + k];
              dwJ[o] += cs * dw[istart + k];
            }
          }
        }
        if (e) {
#pragma omp atomic write
          info = e;
        }
      }
    }
#undef MS_COMP
    free(buf); free(piv);
  }
  return info;
}

/* ---------------------------------------------------------------- Voigt (src/Absorption) */

/* humlicek2: complex_error_functions.jl:24-30; weideman32a: :170-190;
 * w(::HumlicekWeidemann32SDErrorFunction, z): :226-234.  Returns Re w(x+iy). */
static const double W32A[32] = {
    2.5722534081245696e+00,  2.2635372999002676e+00,  1.8256696296324824e+00,  1.3455441692345453e+00,
    9.0192548936480144e-01,  5.4601397206393498e-01,  2.9544451071508926e-01,  1.4060716226893769e-01,
    5.7304403529837900e-02,  1.9006155784845689e-02,  4.5195411053501429e-03,  3.9259136070122748e-04,
    -2.4532980269928922e-04, -1.3075449254548613e-04, -2.1409619200870880e-05, 6.8210319440412389e-06,
    4.4015317319048931e-06,  4.2558331390536872e-07,  -4.1840763666294341e-07, -1.4813078891201116e-07,
    2.2930439569075392e-08,  2.3797557105844622e-08,  8.1248960947953431e-10,  -3.2080150458594088e-09,
    -5.2310170266050247e-10, 4.1537465934749353e-10,  1.1658312885903929e-10,  -5.5441820344468828e-11,
    -2.1542618451370239e-11, 8.0314997274316680e-12,  3.7424975634801558e-12,  -1.3031797863050087e-12};

typedef struct { double re, im; } cplx;
static inline cplx cmul(cplx a, cplx b) { cplx c = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return c; }
static inline cplx cadd(cplx a, cplx b) { cplx c = {a.re + b.re, a.im + b.im}; return c; }
/* Julia's complex division (Base.:/ for Complex{Float64}) is a robust scaled algorithm; the
 * operands here are O(1..1e3) so plain Smith division agrees to ~1 ulp. */
static inline cplx cdiv(cplx a, cplx b) {
  cplx c;
  if (fabs(b.re) >= fabs(b.im)) {
    double r = b.im / b.re, den = b.re + b.im * r;
    c.re = (a.re + a.im * r) / den; c.im = (a.im - a.re * r) / den;
  } else {
    double r = b.re / b.im, den = b.re * r + b.im;
    c.re = (a.re * r + a.im) / den; c.im = (a.im * r - a.re) / den;
  }
  return c;
}

double ora_w_hw32sd_re(double x, double y) {
  const double rsp = 1.0 / sqrt(M_PI);
  if (fabs(x) + y >= 8.0) {
    cplx t = {y, -x};
    cplx u = cmul(t, t);
    cplx num = cmul(t, (cplx){1.410474 + u.re * rsp, u.im * rsp});
    cplx up3 = {3.0 + u.re, u.im};
    cplx den = cadd((cplx){0.75, 0.0}, cmul(u, up3));
    return cdiv(num, den).re;
  }
  const double L = sqrt(32.0 / sqrt(2.0));
  cplx lpiz = {L - y, x}, lmiz = {L + y, -x};
  cplx rec = cdiv((cplx){1.0, 0.0}, lmiz);
  cplx Z = cmul(lpiz, rec);
  cplx p = {W32A[31], 0.0};
  for (int k = 30; k >= 0; --k) { p = cmul(p, Z); p.re += W32A[k]; }
  cplx inner = cmul((cplx){2 * p.re, 2 * p.im}, rec);
  inner.re += rsp;
  return cmul(inner, rec).re;
}

/* line_shape!(::Voigt) compute_absorption_cross_section.jl:179-183 accumulated over lines in
 * line order (host loop :73-126).  Per-line inputs are the host-side prefactors of :79-107:
 * nu (pressure-shifted centre), gamma_d, y, S (temperature-corrected), and the 1-based
 * inclusive window [ind_start, ind_stop] on the grid. */
void ora_voigt_xsec(int nLines, const double *nu, const double *gamma_d, const double *y, const double *Sline,
                    const int *ind_start, const int *ind_stop, int nGrid, const double *grid, double *sigma) {
  const double cSqrtLn2divSqrtPi = 0.469718639319144059835, cSqrtLn2 = 0.8325546111577;
  for (int i = 0; i < nGrid; ++i) sigma[i] = 0.0;
  for (int j = 0; j < nLines; ++j) {
    const int a = ind_start[j] - 1, b = ind_stop[j] - 1;
#pragma omp parallel for schedule(static)
    for (int i = a; i <= b; ++i)
      sigma[i] += Sline[j] * cSqrtLn2divSqrtPi / gamma_d[j] *
                  ora_w_hw32sd_re(cSqrtLn2 / gamma_d[j] * (grid[i] - nu[j]), y[j]);
  }
}

/* ---------------------------------------------------------------- rotational Raman: rt_run(RS_type::RRS, model, iBand)
 *
 * C twin of oracle/rrsref.py (which states the defects D1..D5 of the reference's RRS text and the meaning of the switch
 * `rrs_strict`): rt_kernel!(::RRS) rt_kernel.jl:277-340 = elemental_inelastic!(::RRS) (elemental_inelastic.jl:23-91) +
 * elemental! + doubling_helper!(::RRS) (doubling_inelastic.jl:13-134) + interaction_helper!(::RRS, iface)
 * (interaction_inelastic.jl:8-340), the surface interaction with an added layer whose ie* arrays are zeros, and
 * postprocessing_vza!(::RRS) (tools/postprocessing_vza.jl:95-147).  PARITY UNPINNED like the numpy twin (the reference holds
 * no known-answer test of its Raman path); this file is checked against the twin in tests/test_oracle_rrs.py.
 *
 * A Raman operator ie[.,.,n1,dn] takes radiation from spectral index n0 = n1 + off[dn] to n1.  Every pair (n1, dn) reads
 * elastic operators at n1 and n0 and inelastic ones at (n1, dn) only, so the run can be restricted to an OWNED window
 * [own_lo, own_hi) of n1: the owned entries are those of the unrestricted run.  Elastic arrays cover all S points;
 * inelastic arrays are [N,N,W,nR] / [N,1,W,nR] with W = own_hi - own_lo (element (dn, n1) at (dn*W + n1 - own_lo)).
 * In the strict position the defects D2/D3 index the RAMAN axis with n + off[dn]: they stay inside one spectral index n,
 * so the window applies to them as well. */

typedef struct {
  int nR;
  const int *off;             /* [nR] i_l1l0 = n0 - n1 in grid points (inelastic_helper.jl:13-21) */
  const double *wR;           /* [nR] varpi_l1l0 */
  const double *ZRpp, *ZRmp;  /* [N,N,M] Raman phase-matrix moments (computeRamanZλ!, inelastic_helper.jl:457-464) */
  const double *fscatt;       /* [S,Nz] fScattRayleigh (compEffectiveLayerProperties.jl:58) */
  int rrs_strict;             /* 1 = the reference's text as written, 0 = corrections D1..D5 */
  int own_lo, own_hi;         /* 0-based window of n1, hi exclusive */
} ora_rrs;

typedef struct { /* AddedLayerRS / CompositeLayerRS (types.jl:145-205): elastic part [.,.,S], inelastic part [.,.,W,nR] */
  double *pm, *mp, *mm, *pp, *Jp, *Jm;             /* r+-/R+-, r-+/R-+, t--/T--, t++/T++, j0+/J0+, j0-/J0- */
  double *ie_pm, *ie_mp, *ie_mm, *ie_pp, *ieJp, *ieJm;
} rrs_layer;

static int rrs_layer_alloc(rrs_layer *L, size_t NN, size_t N, size_t S, size_t P, int with_ie) {
  memset(L, 0, sizeof(*L));
  L->pm = (double *)calloc(NN * S, sizeof(double)); L->mp = (double *)calloc(NN * S, sizeof(double));
  L->mm = (double *)calloc(NN * S, sizeof(double)); L->pp = (double *)calloc(NN * S, sizeof(double));
  L->Jp = (double *)calloc(N * S, sizeof(double)); L->Jm = (double *)calloc(N * S, sizeof(double));
  int ok = L->pm && L->mp && L->mm && L->pp && L->Jp && L->Jm;
  if (with_ie) {
    L->ie_pm = (double *)calloc(NN * P, sizeof(double)); L->ie_mp = (double *)calloc(NN * P, sizeof(double));
    L->ie_mm = (double *)calloc(NN * P, sizeof(double)); L->ie_pp = (double *)calloc(NN * P, sizeof(double));
    L->ieJp = (double *)calloc(N * P, sizeof(double)); L->ieJm = (double *)calloc(N * P, sizeof(double));
    ok = ok && L->ie_pm && L->ie_mp && L->ie_mm && L->ie_pp && L->ieJp && L->ieJm;
  }
  return ok;
}

static void rrs_layer_free(rrs_layer *L) {
  free(L->pm); free(L->mp); free(L->mm); free(L->pp); free(L->Jp); free(L->Jm);
  free(L->ie_pm); free(L->ie_mp); free(L->ie_mm); free(L->ie_pp); free(L->ieJp); free(L->ieJm);
}

static inline void madd(size_t n, const double *A, const double *B, double *C) {
  for (size_t x = 0; x < n; ++x) C[x] = A[x] + B[x];
}

/* get_elem_rt_RRS! elemental_inelastic.jl:93-160 and get_elem_rt_SFI_RRS! :320-382 for ONE on-grid pair (n1, dn):
 * d1 = dtau[n1], d0 = dtau[n0], pre-factors at n0, att = exp(-tau_sum[n0]/mu_sun). */
static void rrs_elemental_pair(const ora_streams *q, int m, double wR, double varpi0, double fs0, double d1, double d0,
                               double tau_sum0, const double *Zpp, const double *Zmp, double *ier, double *iet,
                               double *jp, double *jm) {
  const int N = q->N, n = q->nS;
  const double *mu = q->mu;
  const double wct02 = (m == 0) ? 0.5 : 0.25;
  const double pre = wR * varpi0 * fs0;
  const int far = fabs(d0 - d1) > 1.e-6;
  for (int j = 0; j < N; ++j) {
    const double wj = (m == 0) ? q->wt[j] / 2 : q->wt[j] / 4;
    for (int i = 0; i < N; ++i) {
      double r = 0.0, t = 0.0;
      if (wj > 1.e-8) {
        r = fs0 * wR * varpi0 * Zmp[IDX(i, j, N)] * (1 / ((mu[i] / mu[j]) + (d1 / d0))) *
            (1 - exp(-((d1 / mu[i]) + (d0 / mu[j])))) * wj;
        if (mu[i] == mu[j]) {
          if (i == j) {
            const double wi = (m == 0) ? q->wt[i] / 2 : q->wt[i] / 4;
            if (far)
              t = pre * Zpp[IDX(i, i, N)] * wi * (exp(-d0 / mu[i]) - exp(-d1 / mu[i])) / (1 - (d1 / d0));
            else
              t = pre * Zpp[IDX(i, i, N)] * wi * (1 - exp(-d0 / mu[j]));
          } else
            t = 0.0;
        } else
          t = pre * Zpp[IDX(i, j, N)] * (1 / ((mu[i] / mu[j]) - (d1 / d0))) * wj * (exp(-d1 / mu[i]) - exp(-d0 / mu[j]));
      }
      ier[IDX(i, j, N)] = r;
      iet[IDX(i, j, N)] = t;
    }
  }
  const int i_start = n * (q->imu0 - 1), i_end = n * q->imu0;
  const double mus = mu[i_start];
  const double att = exp(-tau_sum0 / mus);
  for (int i = 0; i < N; ++i) {
    double zp = 0.0, zm = 0.0;
    for (int ii = i_start; ii < i_end; ++ii) {
      zp += Zpp[IDX(i, ii, N)] * q->I0[ii - i_start];
      zm += Zmp[IDX(i, ii, N)] * q->I0[ii - i_start];
    }
    double p;
    if (i >= i_start && i < i_end) {
      if (far)
        p = (exp(-d0 / mu[i]) - exp(-d1 / mu[i])) / ((d1 / d0) - 1) * pre * zp * wct02;
      else
        p = wct02 * pre * zp * (1 - exp(-d0 / mus));
    } else
      p = wct02 * pre * zp * (1 / ((mu[i] / mus) - (d1 / d0))) * (exp(-d1 / mu[i]) - exp(-d0 / mus));
    jp[i] = p * att;
    jm[i] = wct02 * pre * zm * (1 / ((mu[i] / mus) + (d1 / d0))) * (1 - exp(-((d1 / mu[i]) + (d0 / mus)))) * att;
  }
}

/* per-thread scratch of the pair loops: 10 matrices + 8 vectors */
#define RRS_WORK(NN, N) (10 * (NN) + 8 * (size_t)(N))

/* interaction_helper!(::RRS, iface, ...) interaction_inelastic.jl:8-22 (00), :28-76 (01), :139-180 (10), :230-340 (11) on
 * whole layers; `a` = added layer (a->ie_* == NULL: a surface layer, whose inelastic arrays are zeros).  wsE: 2*NN*S
 * doubles (tmp_inv, T01 / T21 per point).  Returns LU info, -3 for a non-11 interface in the strict position (D4). */
static int rrs_interaction(const ora_scene *sc, const ora_rrs *rs, int iface, rrs_layer *c, const rrs_layer *a, double *wsE,
                           const double *zeroM, int nthreads) {
  const int N = sc->N, S = sc->S, nR = rs->nR, lo = rs->own_lo, W = rs->own_hi - rs->own_lo;
  const size_t NN = (size_t)N * N;
  int info = 0;
  if (iface != 3 && rs->rrs_strict) return -3;
  double *tinv = wsE, *T1 = wsE + NN * S;
#define PAIR(dn, n1) ((size_t)(dn) * W + (size_t)((n1) - lo))
#define AIE(arr, p) (a->arr ? a->arr + NN * (p) : zeroM)
#define AIEV(arr, p) (a->arr ? a->arr + (size_t)N * (p) : zeroM)
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    double *w = (double *)malloc(RRS_WORK(NN, N) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
    double *M1 = w, *M2 = w + NN, *M3 = w + 2 * NN, *M4 = w + 3 * NN;
    double *v1 = w + 10 * NN, *v2 = v1 + N, *v3 = v2 + N, *v4 = v3 + N;
    if (iface == 0) { /* :8-22 */
#pragma omp for schedule(static)
      for (size_t x = 0; x < (size_t)N * W * nR; ++x) { c->ieJp[x] = 0.0; c->ieJm[x] = 0.0; }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) {
        double *Jp = c->Jp + (size_t)N * s, *Jm = c->Jm + (size_t)N * s;
        gemv(N, a->pp + NN * s, Jp, v1);
        gemv(N, c->mm + NN * s, a->Jm + (size_t)N * s, v2);
        for (int i = 0; i < N; ++i) { Jp[i] = a->Jp[(size_t)N * s + i] + v1[i]; Jm[i] = Jm[i] + v2[i]; }
        gemm(N, a->mm + NN * s, c->mm + NN * s, M1); memcpy(c->mm + NN * s, M1, NN * sizeof(double));
        gemm(N, a->pp + NN * s, c->pp + NN * s, M1); memcpy(c->pp + NN * s, M1, NN * sizeof(double));
      }
    } else if (iface == 1) { /* :28-76 (D4) */
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn)
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          gemv(N, AIE(ie_mp, p), c->Jp + (size_t)N * n0, v1);
          for (int i = 0; i < N; ++i) v1[i] = v1[i] + AIEV(ieJm, p)[i];
          gemv(N, c->mm + NN * n1, v1, c->ieJm + (size_t)N * p);
          gemv(N, AIE(ie_pp, p), c->Jp + (size_t)N * n0, v1);
          for (int i = 0; i < N; ++i) c->ieJp[(size_t)N * p + i] = AIEV(ieJp, p)[i] + v1[i];
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* new J0 into T1 (read by nobody else: the operators below use T++/T-- only) */
        double *Jp = c->Jp + (size_t)N * s, *Jm = c->Jm + (size_t)N * s;
        gemv(N, a->mp + NN * s, Jp, v1);
        for (int i = 0; i < N; ++i) v1[i] = v1[i] + a->Jm[(size_t)N * s + i];
        gemv(N, c->mm + NN * s, v1, v2);
        gemv(N, a->pp + NN * s, Jp, v3);
        for (int i = 0; i < N; ++i) { Jm[i] = Jm[i] + v2[i]; Jp[i] = a->Jp[(size_t)N * s + i] + v3[i]; }
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn)
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          gemm(N, c->mm + NN * n1, AIE(ie_mp, p), M1); gemm(N, M1, c->pp + NN * n0, c->ie_mp + NN * p);
          memcpy(c->ie_pm + NN * p, AIE(ie_pm, p), NN * sizeof(double));
          gemm(N, AIE(ie_pp, p), c->pp + NN * n0, c->ie_pp + NN * p);
          gemm(N, c->mm + NN * n1, AIE(ie_mm, p), c->ie_mm + NN * p);
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) {
        gemm(N, c->mm + NN * s, a->mp + NN * s, M1); gemm(N, M1, c->pp + NN * s, c->mp + NN * s);
        memcpy(c->pm + NN * s, a->pm + NN * s, NN * sizeof(double));
        gemm(N, a->pp + NN * s, c->pp + NN * s, M1); memcpy(c->pp + NN * s, M1, NN * sizeof(double));
        gemm(N, c->mm + NN * s, a->mm + NN * s, M1); memcpy(c->mm + NN * s, M1, NN * sizeof(double));
      }
    } else if (iface == 2) { /* :139-180 (D4) */
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn)
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          double *iJp = c->ieJp + (size_t)N * p, *iJm = c->ieJm + (size_t)N * p;
          gemv(N, c->ie_pm + NN * p, a->Jm + (size_t)N * n0, v1);
          for (int i = 0; i < N; ++i) v1[i] = iJp[i] + v1[i];
          gemv(N, a->pp + NN * n1, v1, v2);
          gemv(N, c->ie_mm + NN * p, a->Jm + (size_t)N * n0, v3);
          for (int i = 0; i < N; ++i) { iJp[i] = v2[i]; iJm[i] = iJm[i] + v3[i]; }
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) {
        double *Jp = c->Jp + (size_t)N * s, *Jm = c->Jm + (size_t)N * s;
        gemv(N, c->pm + NN * s, a->Jm + (size_t)N * s, v1);
        for (int i = 0; i < N; ++i) v1[i] = Jp[i] + v1[i];
        gemv(N, a->pp + NN * s, v1, v2);
        gemv(N, c->mm + NN * s, a->Jm + (size_t)N * s, v3);
        for (int i = 0; i < N; ++i) { Jp[i] = a->Jp[(size_t)N * s + i] + v2[i]; Jm[i] = Jm[i] + v3[i]; }
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn)
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          gemm(N, a->pp + NN * n1, c->ie_pp + NN * p, M1); memcpy(c->ie_pp + NN * p, M1, NN * sizeof(double));
          gemm(N, c->ie_mm + NN * p, a->mm + NN * n0, M1); memcpy(c->ie_mm + NN * p, M1, NN * sizeof(double));
          gemm(N, a->pp + NN * n1, c->ie_pm + NN * p, M1); gemm(N, M1, a->mm + NN * n0, c->ie_pm + NN * p);
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) {
        gemm(N, a->pp + NN * s, c->pp + NN * s, M1); memcpy(c->pp + NN * s, M1, NN * sizeof(double));
        gemm(N, c->mm + NN * s, a->mm + NN * s, M1); memcpy(c->mm + NN * s, M1, NN * sizeof(double));
        gemm(N, a->pp + NN * s, c->pm + NN * s, M1); gemm(N, M1, a->mm + NN * s, c->pm + NN * s);
      }
    } else { /* ScatteringInterface_11 :230-340 */
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* tmp_inv = (I - r-+ R+-)^-1 (:244), T01 = T-- tmp_inv (:247) */
        gemm(N, a->mp + NN * s, c->pm + NN * s, M1);
        for (size_t x = 0; x < NN; ++x) M1[x] = -M1[x];
        for (int i = 0; i < N; ++i) M1[IDX(i, i, N)] += 1.0;
        int e = inv_lu(N, M1, tinv + NN * s, piv);
        if (e) {
#pragma omp atomic write
          info = e;
        }
        gemm(N, c->mm + NN * s, tinv + NN * s, T1 + NN * s);
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn) /* :249-265 */
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          const double *r1 = a->mp + NN * n1, *r0 = a->mp + NN * n0, *T01 = T1 + NN * n1;
          gemm(N, AIE(ie_mp, p), c->pm + NN * n0, M1); gemm(N, r1, c->ie_pm + NN * p, M2); madd(NN, M1, M2, M3);
          gemm(N, T01, M3, M1); madd(NN, M1, c->ie_mm + NN * p, M2);       /* A */
          gemm(N, M2, tinv + NN * n0, M3);                                  /* A tmp_inv[n0] */
          gemv(N, AIE(ie_mp, p), c->Jp + (size_t)N * n0, v1);
          gemv(N, r1, c->ieJp + (size_t)N * p, v2);
          for (int i = 0; i < N; ++i) v1[i] = (v1[i] + v2[i]) + AIEV(ieJm, p)[i];
          gemv(N, T01, v1, v3);
          gemv(N, r0, c->Jp + (size_t)N * n0, v1);
          for (int i = 0; i < N; ++i) v1[i] = a->Jm[(size_t)N * n0 + i] + v1[i];
          gemv(N, M3, v1, v4);
          double *iJm = c->ieJm + (size_t)N * p;
          for (int i = 0; i < N; ++i) iJm[i] = (iJm[i] + v3[i]) + v4[i];
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :267 */
        gemv(N, a->mp + NN * s, c->Jp + (size_t)N * s, v1);
        for (int i = 0; i < N; ++i) v1[i] = v1[i] + a->Jm[(size_t)N * s + i];
        gemv(N, T1 + NN * s, v1, v2);
        for (int i = 0; i < N; ++i) c->Jm[(size_t)N * s + i] += v2[i];
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn) /* :269-285 */
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          const double *r1 = a->mp + NN * n1, *r0 = a->mp + NN * n0, *T01 = T1 + NN * n1;
          gemm(N, AIE(ie_mp, p), c->pm + NN * n0, M1); gemm(N, r1, c->ie_pm + NN * p, M2); madd(NN, M1, M2, M3);
          gemm(N, T01, M3, M1); madd(NN, M1, c->ie_mm + NN * p, M2);       /* A */
          gemm(N, M2, tinv + NN * n0, M3);                                  /* A tmp_inv[n0] */
          gemm(N, AIE(ie_mp, p), c->pp + NN * n0, M1); gemm(N, r1, c->ie_pp + NN * p, M2); madd(NN, M1, M2, M4);
          gemm(N, T01, M4, M1);
          gemm(N, M3, r0, M2); gemm(N, M2, c->pp + NN * n0, M4);
          double *ieR = c->ie_mp + NN * p;
          for (size_t x = 0; x < NN; ++x) ieR[x] = (ieR[x] + M1[x]) + M4[x];
          gemm(N, T01, AIE(ie_mm, p), M1); gemm(N, M3, a->mm + NN * n0, M2);
          madd(NN, M1, M2, c->ie_mm + NN * p);
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :288, :290, then tmp_inv = (I - R+- r-+)^-1 (:295), T21 = t++ tmp_inv (:297) */
        gemm(N, T1 + NN * s, a->mp + NN * s, M1); gemm(N, M1, c->pp + NN * s, M2);
        double *R = c->mp + NN * s;
        for (size_t x = 0; x < NN; ++x) R[x] = R[x] + M2[x];
        gemm(N, T1 + NN * s, a->mm + NN * s, c->mm + NN * s);
        gemm(N, c->pm + NN * s, a->mp + NN * s, M1);
        for (size_t x = 0; x < NN; ++x) M1[x] = -M1[x];
        for (int i = 0; i < N; ++i) M1[IDX(i, i, N)] += 1.0;
        int e = inv_lu(N, M1, tinv + NN * s, piv);
        if (e) {
#pragma omp atomic write
          info = e;
        }
        gemm(N, a->pp + NN * s, tinv + NN * s, T1 + NN * s);
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn) /* :299-313 */
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          const double *r0 = a->mp + NN * n0, *T21 = T1 + NN * n1;
          gemm(N, c->ie_pm + NN * p, r0, M1); gemm(N, c->pm + NN * n1, AIE(ie_mp, p), M2); madd(NN, M1, M2, M3);
          gemm(N, T21, M3, M1); madd(NN, M1, AIE(ie_pp, p), M2);           /* B */
          gemm(N, M2, tinv + NN * n0, M3);                                  /* B tmp_inv[n0] */
          double *iJp = c->ieJp + (size_t)N * p;
          gemv(N, c->ie_pm + NN * p, a->Jm + (size_t)N * n0, v1);
          gemv(N, c->pm + NN * n1, AIEV(ieJm, p), v2);
          for (int i = 0; i < N; ++i) v1[i] = (iJp[i] + v1[i]) + v2[i];
          gemv(N, T21, v1, v3);
          gemv(N, c->pm + NN * n0, a->Jm + (size_t)N * n0, v1);
          for (int i = 0; i < N; ++i) v1[i] = c->Jp[(size_t)N * n0 + i] + v1[i];
          gemv(N, M3, v1, v4);
          for (int i = 0; i < N; ++i) iJp[i] = (AIEV(ieJp, p)[i] + v3[i]) + v4[i];
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :315 */
        gemv(N, c->pm + NN * s, a->Jm + (size_t)N * s, v1);
        for (int i = 0; i < N; ++i) v1[i] = c->Jp[(size_t)N * s + i] + v1[i];
        gemv(N, T1 + NN * s, v1, v2);
        for (int i = 0; i < N; ++i) c->Jp[(size_t)N * s + i] = a->Jp[(size_t)N * s + i] + v2[i];
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn) /* :317-335 */
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          const double *r0 = a->mp + NN * n0, *T21 = T1 + NN * n1;
          gemm(N, c->ie_pm + NN * p, r0, M1); gemm(N, c->pm + NN * n1, AIE(ie_mp, p), M2); madd(NN, M1, M2, M3);
          gemm(N, T21, M3, M1); madd(NN, M1, AIE(ie_pp, p), M2);           /* B */
          gemm(N, M2, tinv + NN * n0, M3);                                  /* B tmp_inv[n0] */
          gemm(N, T21, c->ie_pp + NN * p, M1); gemm(N, M3, c->pp + NN * n0, M2);
          madd(NN, M1, M2, c->ie_pp + NN * p);
          gemm(N, c->ie_pm + NN * p, a->mm + NN * n0, M1); gemm(N, c->pm + NN * n1, AIE(ie_mm, p), M2); madd(NN, M1, M2, M4);
          gemm(N, T21, M4, M1);
          gemm(N, M3, c->pm + NN * n0, M2); gemm(N, M2, a->mm + NN * n0, M4);
          double *ieR = c->ie_pm + NN * p;
          const double *aie = AIE(ie_pm, p);
          for (size_t x = 0; x < NN; ++x) ieR[x] = (aie[x] + M1[x]) + M4[x];
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :338, :340 */
        gemm(N, T1 + NN * s, c->pp + NN * s, M1); memcpy(c->pp + NN * s, M1, NN * sizeof(double));
        gemm(N, T1 + NN * s, c->pm + NN * s, M1); gemm(N, M1, a->mm + NN * s, M2);
        madd(NN, a->pm + NN * s, M2, c->pm + NN * s);
      }
    }
    free(w); free(piv);
  }
#undef AIE
#undef AIEV
  return info;
}

/* doubling_helper!(::RRS) doubling_inelastic.jl:13-134 on the whole added layer, then apply_D_matrix! (doubling.jl:120-134),
 * apply_D_matrix_IE!(::RRS) (:410-425 -> apply_D_IE_RRS! :291-311, D2), apply_D_matrix_SFI! and apply_D_matrix_SFI_IE!
 * (:441-451 -> apply_D_SFI_IE_RRS! :345-357, D3).  expk [S] is updated in place.
 * wsE: (3*NN + 4*N + 1) * S doubles (gp_refl, tt++ gp_refl, (tt++ gp_refl) r-+, j1+, j1-, tmp1, tmp2, expk of the step's
 * start per point). */
static int rrs_doubling(const ora_scene *sc, const ora_rrs *rs, const ora_streams *q, int nd, double *expk, rrs_layer *a,
                        double *wsE, int nthreads) {
  const int N = sc->N, n = sc->nS, S = sc->S, nR = rs->nR, lo = rs->own_lo, W = rs->own_hi - rs->own_lo;
  const int strict = rs->rrs_strict;
  const size_t NN = (size_t)N * N;
  int info = 0;
  if (nd == 0) return 0;
  double *gp = wsE, *ttgp = gp + NN * S, *ttgpr = ttgp + NN * S;
  double *j1p = ttgpr + NN * S, *j1m = j1p + (size_t)N * S, *tmp1 = j1m + (size_t)N * S, *tmp2 = tmp1 + (size_t)N * S;
  double *expk0 = tmp2 + (size_t)N * S;
  double *r = a->mp, *t = a->pp, *jp = a->Jp, *jm = a->Jm;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    double *w = (double *)malloc(RRS_WORK(NN, N) * sizeof(double));
    int *piv = (int *)malloc(N * sizeof(int));
    double *M1 = w, *M2 = w + NN, *X = w + 2 * NN, *Y = w + 3 * NN, *Wm = w + 4 * NN, *G = w + 5 * NN, *M3 = w + 6 * NN,
           *M4 = w + 7 * NN;
    double *v1 = w + 10 * NN, *v2 = v1 + N, *v3 = v2 + N, *v4 = v3 + N, *e1p = v4 + N, *e1m = e1p + N, *v5 = e1m + N;
    for (int it = 0; it < nd; ++it) {
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :47-59 */
        gemm(N, r + NN * s, r + NN * s, M1);
        for (size_t x = 0; x < NN; ++x) M1[x] = -M1[x];
        for (int i = 0; i < N; ++i) M1[IDX(i, i, N)] += 1.0;
        int e = inv_lu(N, M1, gp + NN * s, piv);
        if (e) {
#pragma omp atomic write
          info = e;
        }
        gemm(N, t + NN * s, gp + NN * s, ttgp + NN * s);
        gemm(N, ttgp + NN * s, r + NN * s, ttgpr + NN * s);
        expk0[s] = expk[s];
        double *p1 = j1p + (size_t)N * s, *m1 = j1m + (size_t)N * s;
        for (int i = 0; i < N; ++i) { p1[i] = jp[(size_t)N * s + i] * expk[s]; m1[i] = jm[(size_t)N * s + i] * expk[s]; }
        gemv(N, r + NN * s, m1, v1);
        for (int i = 0; i < N; ++i) v1[i] = jp[(size_t)N * s + i] + v1[i];
        gemv(N, gp + NN * s, v1, tmp1 + (size_t)N * s);
        gemv(N, r + NN * s, jp + (size_t)N * s, v1);
        for (int i = 0; i < N; ++i) v1[i] = m1[i] + v1[i];
        gemv(N, gp + NN * s, v1, tmp2 + (size_t)N * s);
      }
      for (int dn = 0; dn < nR; ++dn) { /* :61-96; ieJ1+- = ieJ0+- .* expk' of :52,:56 formed per pair */
        if (abs(rs->off[dn]) < S) {
#pragma omp for schedule(dynamic, 16)
          for (int n1 = lo; n1 < lo + W; ++n1) {
            const int n0 = n1 + rs->off[dn];
            if (n0 < 0 || n0 >= S) continue;
            const size_t p = PAIR(dn, n1);
            const double *r1 = r + NN * n1, *r0 = r + NN * n0, *e = a->ie_mp + NN * p;
            double *iJp = a->ieJp + (size_t)N * p, *iJm = a->ieJm + (size_t)N * p;
            const double ek = expk0[n1]; /* ieJ1+- are formed ONCE per step (:52,:56), before D1 squares expk inside the loop */
            for (int i = 0; i < N; ++i) { e1p[i] = iJp[i] * ek; e1m[i] = iJm[i] * ek; }
            gemm(N, r1, e, M1); gemm(N, e, r0, M2); madd(NN, M1, M2, X);
            gemv(N, r1, e1m, v1); gemv(N, e, j1m + (size_t)N * n0, v2); gemv(N, X, tmp1 + (size_t)N * n0, v3);
            for (int i = 0; i < N; ++i) v1[i] = ((iJp[i] + v1[i]) + v2[i]) + v3[i];
            gemv(N, ttgp + NN * n1, v1, v4);
            gemv(N, a->ie_pp + NN * p, tmp1 + (size_t)N * n0, v5);
            for (int i = 0; i < N; ++i) iJp[i] = (e1p[i] + v4[i]) + v5[i];
            const double *itm = strict ? a->ie_mm + NN * p : a->ie_pp + NN * p; /* D5 */
            gemv(N, e, jp + (size_t)N * n0, v1); gemv(N, r1, iJp, v2); gemv(N, X, tmp2 + (size_t)N * n0, v3);
            for (int i = 0; i < N; ++i) v1[i] = ((e1m[i] + v1[i]) + v2[i]) + v3[i];
            gemv(N, ttgp + NN * n1, v1, v4);
            gemv(N, itm, tmp2 + (size_t)N * n0, v5);
            for (int i = 0; i < N; ++i) iJm[i] = (iJm[i] + v4[i]) + v5[i];
          }
        }
        if (strict) { /* D1: :90-95 inside the loop over dn */
#pragma omp for schedule(static)
          for (int s = 0; s < S; ++s) {
            double *p0 = jp + (size_t)N * s, *m0 = jm + (size_t)N * s;
            const double *p1 = j1p + (size_t)N * s, *m1 = j1m + (size_t)N * s;
            gemv(N, r + NN * s, p0, v1);
            for (int i = 0; i < N; ++i) v1[i] = m1[i] + v1[i];
            gemv(N, ttgp + NN * s, v1, v2);
            gemv(N, r + NN * s, m1, v1);
            for (int i = 0; i < N; ++i) v1[i] = p0[i] + v1[i];
            gemv(N, ttgp + NN * s, v1, v3);
            for (int i = 0; i < N; ++i) { m0[i] = m0[i] + v2[i]; p0[i] = p1[i] + v3[i]; }
            expk[s] = expk[s] * expk[s];
          }
        }
      }
      if (!strict) {
#pragma omp for schedule(static)
        for (int s = 0; s < S; ++s) {
          double *p0 = jp + (size_t)N * s, *m0 = jm + (size_t)N * s;
          const double *p1 = j1p + (size_t)N * s, *m1 = j1m + (size_t)N * s;
          gemv(N, r + NN * s, p0, v1);
          for (int i = 0; i < N; ++i) v1[i] = m1[i] + v1[i];
          gemv(N, ttgp + NN * s, v1, v2);
          gemv(N, r + NN * s, m1, v1);
          for (int i = 0; i < N; ++i) v1[i] = p0[i] + v1[i];
          gemv(N, ttgp + NN * s, v1, v3);
          for (int i = 0; i < N; ++i) { m0[i] = m0[i] + v2[i]; p0[i] = p1[i] + v3[i]; }
          expk[s] = expk[s] * expk[s];
        }
      }
#pragma omp for schedule(dynamic, 8) collapse(2)
      for (int dn = 0; dn < nR; ++dn) /* :98-125 */
        for (int n1 = lo; n1 < lo + W; ++n1) {
          const int n0 = n1 + rs->off[dn];
          if (n0 < 0 || n0 >= S) continue;
          const size_t p = PAIR(dn, n1);
          const double *r1 = r + NN * n1, *r0 = r + NN * n0, *tg1 = ttgp + NN * n1;
          double *e = a->ie_mp + NN * p, *et = a->ie_pp + NN * p;
          gemm(N, e, r0, M1); gemm(N, r1, e, M2); madd(NN, M1, M2, X);
          gemm(N, X, gp + NN * n0, M1); gemm(N, M1, t + NN * n0, Y);
          madd(NN, et, Y, Wm);
          gemm(N, et, gp + NN * n0, G);
          gemm(N, tg1, Wm, M1); gemm(N, G, t + NN * n0, M2);
          madd(NN, M1, M2, et);                                            /* new iet++ */
          madd(NN, et, Y, Wm);
          gemm(N, ttgpr + NN * n1, Wm, M1);                                /* (tt++ gp r-+)[n1] (iet++ + Y) */
          gemm(N, et, gp + NN * n0, G); gemm(N, G, r0, M2); gemm(N, tg1, e, M3); madd(NN, M2, M3, M4);
          gemm(N, M4, t + NN * n0, M2);
          for (size_t x = 0; x < NN; ++x) e[x] = (e[x] + M1[x]) + M2[x];
        }
#pragma omp for schedule(static)
      for (int s = 0; s < S; ++s) { /* :128, :131 */
        gemm(N, ttgpr + NN * s, t + NN * s, M1);
        double *rs_ = r + NN * s;
        for (size_t x = 0; x < NN; ++x) rs_[x] = rs_[x] + M1[x];
        gemm(N, ttgp + NN * s, t + NN * s, M1);
        memcpy(t + NN * s, M1, NN * sizeof(double));
      }
    }
    free(w); free(piv);
  }
  /* D back-transformations */
  const size_t P = (size_t)W * nR;
  if (n == 1) {
    memcpy(a->pm, r, NN * S * sizeof(double)); memcpy(a->mm, t, NN * S * sizeof(double));
    memcpy(a->ie_pm, a->ie_mp, NN * P * sizeof(double)); memcpy(a->ie_mm, a->ie_pp, NN * P * sizeof(double));
    return info;
  }
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
  for (int s = 0; s < S; ++s) {
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        const int ci = scomp(i, n, q->strict), cj = scomp(j, n, q->strict);
        const size_t o = NN * s + IDX(i, j, N);
        if (ci > 2) r[o] = -r[o];
        const double sg = dsign(ci, cj);
        a->pm[o] = sg * r[o];
        a->mm[o] = sg * t[o];
      }
    for (int i = 0; i < N; ++i)
      if (scomp(i, n, q->strict) > 2) jm[(size_t)N * s + i] = -jm[(size_t)N * s + i];
  }
  if (!strict) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (size_t p = 0; p < P; ++p) {
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
          const int ci = scomp(i, n, q->strict), cj = scomp(j, n, q->strict);
          const size_t o = NN * p + IDX(i, j, N);
          if (ci > 2) a->ie_mp[o] = -a->ie_mp[o];
          const double sg = dsign(ci, cj);
          a->ie_pm[o] = sg * a->ie_mp[o];
          a->ie_mm[o] = sg * a->ie_pp[o];
        }
      for (int i = 0; i < N; ++i)
        if (scomp(i, n, q->strict) > 2) a->ieJm[(size_t)N * p + i] = -a->ieJm[(size_t)N * p + i];
    }
    return info;
  }
  /* strict: work item (n, dn) touches the element [.., n, kk] with kk = n + off[dn] taken as a RAMAN index (D2, D3);
   * single-thread column-major order, dn slowest */
  for (int dn = 0; dn < nR; ++dn)
    for (int n1 = lo; n1 < lo + W; ++n1) {
      const long kk = (long)n1 + rs->off[dn];
      if (kk < 0 || kk >= nR) continue;
      const size_t p = PAIR(kk, n1);
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
          const int ci = scomp(i, n, q->strict), cj = scomp(j, n, q->strict);
          const size_t o = NN * p + IDX(i, j, N);
          if (ci > 2) a->ie_mp[o] = -a->ie_mp[o];
          const double sg = dsign(ci, cj);
          a->ie_pm[o] = sg * a->ie_mp[o];
          a->ie_mm[o] = sg * a->ie_pp[o];
        }
    }
  for (int dn = 0; dn < nR; ++dn)
    for (int n1 = lo; n1 < lo + W; ++n1) {
      const long kk = (long)n1 + rs->off[dn];
      if (kk < 0 || kk >= nR) continue;
      const size_t p = PAIR(kk, n1), ps = PAIR(dn, n1);
      for (int i = 0; i < N; ++i)
        if (scomp(i, n, q->strict) > 2) a->ieJm[(size_t)N * p + i] = -a->ieJm[(size_t)N * ps + i];
    }
  return info;
}

/* rt_run(RS_type::RRS, model, iBand) rt_run.jl:41-230 with SFI = true.  R_SFI, T_SFI, ieR_SFI, ieT_SFI: [nVza,nStokes,S]
 * column-major, zero-initialised by the caller; the inelastic spectra are produced for the owned points only.
 * The added, composite and surface layers persist over layers and Fourier moments like the reference's (rt_run.jl:108-116).
 * Returns 0, an LU info > 0, -1 (allocation), -2 (an offset with |off| >= S: get_n0_n1 raises a BoundsError,
 * inelastic_helper.jl:13-21), -3 (a 00 / 01 / 10 interface in the strict position: MethodError in the reference, D4). */
int ora_rt_run_rrs(const ora_scene *sc, const ora_rrs *rs, int nthreads, double *R_SFI, double *T_SFI, double *ieR_SFI,
                   double *ieT_SFI) {
  const int N = sc->N, n = sc->nS, S = sc->S, Nz = sc->Nz, K = sc->K, M = sc->M, nR = rs->nR;
  const int lo = rs->own_lo, W = rs->own_hi - rs->own_lo;
  const size_t NN = (size_t)N * N, P = (size_t)W * nR;
  if (nthreads <= 0) nthreads = 1;
  if (lo < 0 || rs->own_hi > S || W <= 0) return -1;
  for (int dn = 0; dn < nR; ++dn)
    if (abs(rs->off[dn]) >= S) return -2;
  ora_streams q = {N, n, sc->imu0, sc->mu, sc->wt, sc->I0, sc->D, sc->strict, sc->mu0};
  rrs_layer added, comp, surf;
  int ok = rrs_layer_alloc(&added, NN, N, S, P, 1);
  ok = rrs_layer_alloc(&comp, NN, N, S, P, 1) && ok;
  ok = rrs_layer_alloc(&surf, NN, N, S, P, 0) && ok;
  double *wsE = (double *)malloc(((3 * NN + 4 * (size_t)N + 1) * S) * sizeof(double));
  double *expk = (double *)malloc((size_t)S * sizeof(double)), *dtau = (double *)malloc((size_t)S * sizeof(double));
  double *zeroM = (double *)calloc(NN, sizeof(double));
  int info = 0;
  if (!(ok && wsE && expk && dtau && zeroM)) { info = -1; goto done; }
  for (int m = 0; m < M && info >= 0; ++m) {
    const double weight = (m == 0) ? 0.5 : 1.0;
    const double *ZRp = rs->ZRpp + NN * m, *ZRm = rs->ZRmp + NN * m;
    for (int z = 0; z < Nz && info >= 0; ++z) {
      const int nd = sc->nd[z];
      const double *tau = sc->tau + (size_t)S * z, *varpi = sc->varpi + (size_t)S * z, *fs = rs->fscatt + (size_t)S * z;
      const double *tsum = sc->tau_sum + (size_t)S * z;
      for (int s = 0; s < S; ++s) { dtau[s] = tau[s] / ldexp(1.0, nd); expk[s] = exp(-dtau[s] / sc->mu0); } /* rt_kernel.jl:269-275 */
      /* elemental_inelastic!(::RRS) elemental_inelastic.jl:23-91 on the persistent added layer */
#pragma omp parallel num_threads(nthreads)
      {
        double *Zp = (double *)malloc(2 * NN * sizeof(double)), *Zm = Zp + NN;
#pragma omp for schedule(dynamic, 8) collapse(2)
        for (int dn = 0; dn < nR; ++dn)
          for (int n1 = lo; n1 < lo + W; ++n1) {
            const int n0 = n1 + rs->off[dn];
            const size_t p = PAIR(dn, n1);
            double *ier = added.ie_mp + NN * p, *iet = added.ie_pp + NN * p;
            double *iJp = added.ieJp + (size_t)N * p, *iJm = added.ieJm + (size_t)N * p;
            if (n0 < 0 || n0 >= S) { /* get_elem_rt_RRS! zeroes the operators (:153-160); the SFI kernel leaves the sources (:345) */
              for (size_t x = 0; x < NN; ++x) { ier[x] = 0.0; iet[x] = 0.0; }
            } else
              rrs_elemental_pair(&q, m, rs->wR[dn], varpi[n0], fs[n0], dtau[n1], dtau[n0], tsum[n0], ZRp, ZRm, ier, iet, iJp, iJm);
            if (nd >= 1) /* :378-380: every entry, on the grid or not */
              for (int i = 0; i < N; ++i) iJm[i] = sc->D[i % n] * iJm[i];
            /* apply_D_elemental_RRS! :384-402 */
            if (nd < 1) {
              for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) {
                  const double sg = dsign(scomp(i, n, q.strict), scomp(j, n, q.strict));
                  added.ie_pm[NN * p + IDX(i, j, N)] = sg * ier[IDX(i, j, N)];
                  added.ie_mm[NN * p + IDX(i, j, N)] = sg * iet[IDX(i, j, N)];
                }
            } else {
              for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i)
                  if (scomp(i, n, q.strict) > 2) ier[IDX(i, j, N)] = -ier[IDX(i, j, N)];
            }
          }
        /* elemental! (elastic) on the same added layer */
#pragma omp for schedule(static)
        for (int s = 0; s < S; ++s) {
          for (size_t x = 0; x < NN; ++x) { Zp[x] = 0; Zm[x] = 0; }
          for (int k = 0; k < K; ++k) {
            const double w = sc->zw[k + (size_t)K * (s + (size_t)S * z)];
            const double *bp = sc->Zpp + NN * (k + (size_t)K * m), *bm = sc->Zmp + NN * (k + (size_t)K * m);
            for (size_t x = 0; x < NN; ++x) { Zp[x] += w * bp[x]; Zm[x] += w * bm[x]; }
          }
          elemental_pt(&q, m, nd, tsum[s], dtau[s], varpi[s], Zp, Zm, added.mp + NN * s, added.pp + NN * s, added.pm + NN * s,
                       added.mm + NN * s, added.Jp + (size_t)N * s, added.Jm + (size_t)N * s);
        }
        free(Zp);
      }
      int e = rrs_doubling(sc, rs, &q, nd, expk, &added, wsE, nthreads);
      if (e && !info) info = e;
      if (z == 0) { /* rt_kernel.jl:326-333 */
        memcpy(comp.pp, added.pp, NN * S * sizeof(double)); memcpy(comp.mm, added.mm, NN * S * sizeof(double));
        memcpy(comp.mp, added.mp, NN * S * sizeof(double)); memcpy(comp.pm, added.pm, NN * S * sizeof(double));
        memcpy(comp.Jp, added.Jp, (size_t)N * S * sizeof(double)); memcpy(comp.Jm, added.Jm, (size_t)N * S * sizeof(double));
        memcpy(comp.ie_pp, added.ie_pp, NN * P * sizeof(double)); memcpy(comp.ie_mm, added.ie_mm, NN * P * sizeof(double));
        memcpy(comp.ie_mp, added.ie_mp, NN * P * sizeof(double)); memcpy(comp.ie_pm, added.ie_pm, NN * P * sizeof(double));
        memcpy(comp.ieJp, added.ieJp, (size_t)N * P * sizeof(double)); memcpy(comp.ieJm, added.ieJm, (size_t)N * P * sizeof(double));
      } else {
        e = rrs_interaction(sc, rs, sc->iface[z], &comp, &added, wsE, zeroM, nthreads);
        if (e < 0) info = e; else if (e && !info) info = e;
      }
    }
    if (info < 0) break;
    /* surface (rt_run.jl:169-185): its inelastic arrays are never written (zeros) */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int s = 0; s < S; ++s) {
      double *s_rpm = surf.pm + NN * s, *s_rmp = surf.mp + NN * s, *s_tmm = surf.mm + NN * s, *s_tpp = surf.pp + NN * s;
      double *s_jp = surf.Jp + (size_t)N * s, *s_jm = surf.Jm + (size_t)N * s;
      const double tt = sc->tau_sum[s + (size_t)S * Nz];
      if (sc->surf_kind == 1) surface_brdf_pt(&q, sc->Rsurf + NN * m, tt, s_rpm, s_rmp, s_tmm, s_tpp, s_jp, s_jm);
      else if (sc->surf_kind == 2) surface_legendre_pt(&q, m, sc->albedo_spec[s], tt, s_rpm, s_rmp, s_tmm, s_tpp, s_jp, s_jm);
      else surface_lambertian_pt(&q, m, sc->albedo, tt, s_rpm, s_rmp, s_tmm, s_tpp, s_jp, s_jm);
    }
    int e = rrs_interaction(sc, rs, sc->iface[Nz - 1], &comp, &surf, wsE, zeroM, nthreads);
    if (e < 0) { info = e; break; } else if (e && !info) info = e;
    /* postprocessing_vza!(::RRS) tools/postprocessing_vza.jl:95-147 (SFI branch): the sum runs over EVERY Raman index */
    for (int s = 0; s < S; ++s)
      for (int v = 0; v < sc->nVza; ++v) {
        const int istart = (sc->node[v] - 1) * n;
        const double c = sc->cos_mphi[v + (size_t)sc->nVza * m], sn = sc->sin_mphi[v + (size_t)sc->nVza * m];
        for (int k = 0; k < n; ++k) {
          const double cs = weight * ((k < 2) ? c : sn);
          const size_t o = v + (size_t)sc->nVza * (k + (size_t)n * s);
          R_SFI[o] += cs * comp.Jm[(size_t)N * s + istart + k];
          T_SFI[o] += cs * comp.Jp[(size_t)N * s + istart + k];
          if (s >= lo && s < lo + W)
            for (int dn = 0; dn < nR; ++dn) {
              const size_t p = PAIR(dn, s);
              ieR_SFI[o] += cs * comp.ieJm[(size_t)N * p + istart + k];
              ieT_SFI[o] += cs * comp.ieJp[(size_t)N * p + istart + k];
            }
        }
      }
  }
done:
  rrs_layer_free(&added); rrs_layer_free(&comp); rrs_layer_free(&surf);
  free(wsE); free(expk); free(dtau); free(zeroM);
  return info;
}
#undef PAIR
