// voigt.hip -- Voigt line-by-line absorption cross section on gfx950.
//
// Restates line_shape!(::Voigt) (src/Absorption/compute_absorption_cross_section.jl:179-183)
// with w(::HumlicekWeidemann32SDErrorFunction, z) (complex_error_functions.jl:226-234:
// humlicek2 :24-30 for |x|+y >= 8, weideman32a :170-190 otherwise), accumulated over the
// host loop over lines (:73-126).  The reference launches one kernel per line; here ONE
// launch covers all lines: a workgroup owns 256 consecutive grid points, compacts (in line
// order) the lines whose window overlaps its range into LDS, and every thread sums its grid
// point's contributions in ascending line order -- the same accumulation order as the
// reference's sequential `A[I] += ...`, with no atomics, so results are reproducible.
// FP64 VALU-bound (about 300 flop per (line, grid point) evaluation), negligible HBM bytes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "momcore.h"
#include "mom_host.hpp"

namespace {

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__constant__ double kW32A[32] = {
    2.5722534081245696e+00,  2.2635372999002676e+00,  1.8256696296324824e+00,  1.3455441692345453e+00,
    9.0192548936480144e-01,  5.4601397206393498e-01,  2.9544451071508926e-01,  1.4060716226893769e-01,
    5.7304403529837900e-02,  1.9006155784845689e-02,  4.5195411053501429e-03,  3.9259136070122748e-04,
    -2.4532980269928922e-04, -1.3075449254548613e-04, -2.1409619200870880e-05, 6.8210319440412389e-06,
    4.4015317319048931e-06,  4.2558331390536872e-07,  -4.1840763666294341e-07, -1.4813078891201116e-07,
    2.2930439569075392e-08,  2.3797557105844622e-08,  8.1248960947953431e-10,  -3.2080150458594088e-09,
    -5.2310170266050247e-10, 4.1537465934749353e-10,  1.1658312885903929e-10,  -5.5441820344468828e-11,
    -2.1542618451370239e-11, 8.0314997274316680e-12,  3.7424975634801558e-12,  -1.3031797863050087e-12};

// 1 / d and n / d for d in a range where neither scaling nor special cases are needed (here 22 <= d <= 1e60): v_rcp_f64
// refined by two Newton steps, the quotient by one residual step -- 8 instructions instead of the 12 of the IEEE division
// sequence (2 v_div_scale, v_div_fmas, v_div_fixup around the same Newton steps).  The result is within 1 ulp of the correctly
// rounded quotient (2e-16 relative, inside the 1e-13 parity bar of tests/test_gpu_voigt.py).
__device__ __forceinline__ double rcp_refined(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double div_refined(double n, double d) {
  const double r = rcp_refined(d), q = n * r;
  return fma(fma(-d, q, n), r, q);
}

__device__ __forceinline__ double w_hw32sd_re(double x, double y) {
  const double rsp = 0.5641895835477563;  // 1/sqrt(pi)
  if (fabs(x) + y >= 8.0) {               // humlicek2, t = y - i x
    const cplx t = {y, -x};
    const cplx u = cmul(t, t);
    const cplx num = cmul(t, cplx{1.410474 + u.re * rsp, u.im * rsp});
    const cplx up3 = {3.0 + u.re, u.im};
    cplx den = cmul(u, up3);
    den.re += 0.75;
    // only the real part of num/den is needed: one division (|den| ~ |t|^4 >= 4e3 here and < 1e30 for any
    // line/grid distance in cm^-1 units, so the squares neither overflow nor underflow)
    return div_refined(num.re * den.re + num.im * den.im, den.re * den.re + den.im * den.im);
  }
  const double L = 4.756828460010884;  // sqrt(32/sqrt(2))
  const cplx lpiz = {L - y, x}, lmiz = {L + y, -x};
  // 1 / (L - i z) = conj / |.|^2: ONE division and no branch (Julia's complex division, complex.jl, is Smith's
  // branching three-division scheme; |L - i z|^2 lies in [22, 200] here, so the plain form differs from it by rounding
  // only -- a few 1e-16 relative, inside the 1e-13 parity bar -- and a wave no longer executes both branches)
  const double inv = rcp_refined(lmiz.re * lmiz.re + lmiz.im * lmiz.im);
  const cplx rec = {lmiz.re * inv, -lmiz.im * inv};
  const cplx Z = cmul(lpiz, rec);
  cplx p = {kW32A[31], 0.0};
#pragma unroll
  for (int k = 30; k >= 0; --k) {
    p = cmul(p, Z);
    p.re += kW32A[k];
  }
  cplx inner = cmul(cplx{2 * p.re, 2 * p.im}, rec);
  inner.re += rsp;
  return cmul(inner, rec).re;
}

#ifndef MOM_VOIGT_BLOCK
#define MOM_VOIGT_BLOCK 256
#endif
constexpr int kBlock = MOM_VOIGT_BLOCK;  // grid points per workgroup

__device__ __forceinline__ void voigt_block(int nLines, const double *__restrict__ nu,
                                            const double *__restrict__ gamma_d, const double *__restrict__ y,
                                            const double *__restrict__ S, const int *__restrict__ i0,
                                            const int *__restrict__ i1, int nGrid,
                                            const double *__restrict__ grid, double *__restrict__ sigma,
                                            double factor, int accumulate, int sorted) {
  // per-line constants of the candidates, staged once per workgroup: centre, S c/gamma_d, c'/gamma_d, y and the
  // 0-based window -- the two divisions by gamma_d are per LINE here, not per evaluation (same expressions, same values)
  __shared__ double c_nu[kBlock], c_a[kBlock], c_b[kBlock], c_y[kBlock];
  __shared__ int2 c_win[kBlock];  // {first point, last - first} of the window: ONE read and ONE unsigned compare per candidate
  __shared__ int wcount[kBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = blockIdx.x * kBlock;             // 0-based first grid index of this block
  const int g1 = min(nGrid, g0 + kBlock) - 1;     // last
  const int gi = g0 + tid;
  const double gx = (gi < nGrid) ? grid[gi] : 0.0;
  const double cSqrtLn2divSqrtPi = 0.469718639319144059835, cSqrtLn2 = 0.8325546111577;
  double acc = 0.0;
  // Range [jlo, jhi] of line indices whose window touches this block: one cheap strided pass (two loads and a compare
  // per line, no barrier) instead of running the ordered compaction below over the whole list -- line lists are
  // sorted by wavenumber, so the range is tight (an unsorted list still works, it only gets no benefit).  The sum
  // stays in ascending line order.
  __shared__ int s_lo, s_hi;
  if (tid == 0) { s_lo = nLines; s_hi = -1; }
  __syncthreads();
  if (sorted) {
    // window starts and stops both non-decreasing in the line index (the host checked): two binary searches
    if (tid == 0) {
      int a = 0, b = nLines;
      while (a < b) { const int mid = (a + b) >> 1; if (i1[mid] - 1 >= g0) b = mid; else a = mid + 1; }
      s_lo = a;   // first line whose window ends at or after the block's first point
      a = 0; b = nLines;
      while (a < b) { const int mid = (a + b) >> 1; if (i0[mid] - 1 > g1) b = mid; else a = mid + 1; }
      s_hi = a - 1;  // last line whose window starts at or before the block's last point
    }
  } else {
    int mylo = nLines, myhi = -1;
    for (int j = tid; j < nLines; j += kBlock) {
      const int lo = i0[j] - 1, hi = i1[j] - 1;
      if (lo <= g1 && hi >= g0 && hi >= lo) { mylo = min(mylo, j); myhi = max(myhi, j); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mylo = min(mylo, __shfl_xor(mylo, off));
      myhi = max(myhi, __shfl_xor(myhi, off));
    }
    if (lane == 0) { atomicMin(&s_lo, mylo); atomicMax(&s_hi, myhi); }
  }
  __syncthreads();
  const int jlo = s_lo, jhi = s_hi;
  for (int base = jlo; base <= jhi; base += kBlock) {
    const int j = base + tid;
    bool hit = false;
    int lo = 0, hi = -1;
    if (j <= jhi) {
      lo = i0[j] - 1;
      hi = i1[j] - 1;
      hit = (lo <= g1) && (hi >= g0) && (hi >= lo);
    }
    const unsigned long long mask = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(mask);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (hit) {  // ordered compaction: candidates keep the line order (the reference's accumulation order)
      const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
      const double gd = gamma_d[j];
      c_nu[pos] = nu[j];
      c_a[pos] = S[j] * cSqrtLn2divSqrtPi / gd;
      c_b[pos] = cSqrtLn2 / gd;
      c_y[pos] = y[j];
      c_win[pos] = make_int2(lo, hi - lo);  // hi >= lo for a hit
    }
    int nc = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) nc += wcount[w];
    nc = __builtin_amdgcn_readfirstlane(nc);  // workgroup-uniform: scalar loop control
    __syncthreads();
    // r5 (0.37 -> 0.40 of the vector peak, profiles/r05_voigt_ab.txt): the window as ONE LDS read and one unsigned compare
    // (r4: first point, last point and then the four doubles in three dependent round trips, each behind its own exec-masked
    // region); the evaluation is skipped only when the WHOLE wavefront lies outside the window; the quotient of the Humlicek
    // branch by the refined reciprocal.  Measured and NOT shipped: two / four grid points per thread (0.39 / 0.34: the two
    // evaluations of a thread do not overlap better than two wavefronts do, and the core workload loses its early exits), a
    // straight-line two-point Humlicek path, the strided range pass instead of the bisection for short line lists.
    const int gt = (gi < nGrid) ? gi : -1;  // a point past the end of the grid lies in no window: (unsigned)(-1 - lo) > any span
#ifndef MOM_VOIGT_DIAG_NOEVAL
    for (int c = 0; c < nc; ++c) {
      const int2 win = c_win[c];
      const double cn = c_nu[c], ca = c_a[c], cb = c_b[c], cy = c_y[c];
      const bool in = (unsigned)(gt - win.x) <= (unsigned)win.y;
      if (__builtin_amdgcn_ballot_w64(in) == 0) continue;
      const double w = w_hw32sd_re(cb * (gx - cn), cy);
      if (in) acc = fma(ca, w, acc);   // acc += a w, one rounding as before
    }
#else
    (void)gt;
    if (nc > 0) acc += c_a[nc - 1] + c_nu[0] + c_b[0] + c_y[0] + c_win[0].x;   // (diagnostic build: the setup without the evaluations)
#endif
    __syncthreads();
  }
  // accumulate: tau_abs[:, iz] += sigma * (vcd_dry[iz] * vmr)  (atmo_prof.jl:446), fused into the line-shape kernel
  // (separately rounded product and sum, like the host expression: no FMA contraction)
  if (gi < nGrid) {
#pragma clang fp contract(off)
    const double scaled = acc * factor;
    sigma[gi] = accumulate ? sigma[gi] + scaled : acc;
  }
}

__global__ void __launch_bounds__(kBlock) k_voigt(int nLines, const double *__restrict__ nu,
                                                  const double *__restrict__ gamma_d, const double *__restrict__ y,
                                                  const double *__restrict__ S, const int *__restrict__ i0,
                                                  const int *__restrict__ i1, int nGrid,
                                                  const double *__restrict__ grid, double *__restrict__ sigma,
                                                  double factor, int accumulate, int sorted) {
  voigt_block(nLines, nu, gamma_d, y, S, i0, i1, nGrid, grid, sigma, factor, accumulate, sorted);
}

// All layers of a profile in ONE launch (blockIdx.y = layer): the per-line prefactors of layer z sit at [k][z][cap]
// (k = nu, gamma_d, y, S; the two window arrays likewise as ints), tau_abs[:, z] += sigma_z * factor[z]; whether the
// bisection applies is read from the layer's flag on the device (no host round trip between the two kernels).
__global__ void __launch_bounds__(kBlock) k_voigt_profile(int nLines, int Nz, size_t cap, const double *__restrict__ pf,
                                                          const int *__restrict__ win, int nGrid, const double *__restrict__ grid,
                                                          double *__restrict__ tau_abs, const double *__restrict__ factor,
                                                          const int *__restrict__ unsorted) {
  const int z = blockIdx.y;
  const size_t lz = (size_t)z * cap, ks = (size_t)Nz * cap;
  voigt_block(nLines, pf + lz, pf + ks + lz, pf + 2 * ks + lz, pf + 3 * ks + lz, win + lz, win + ks + lz, nGrid, grid,
              tau_abs + (size_t)nGrid * z, factor[z], 1, unsorted[z] ? 0 : 1);
}

thread_local double v_last_ms = 0.0;

}  // namespace

// one launch for all lines on `st` (device pointers); used by mom_voigt_xsec below and by the handle-level
// mom_voigt_tau_abs (momcore.hip), which accumulates straight into the resident tau_abs table
// ---------------------------------------------------------------------------------------------------------------------
// Per-line prefactors of compute_absorption_cross_section (compute_absorption_cross_section.jl:73-107) on the device, from
// ONE resident HITRAN table per absorber: pressure shift (:79), Lorentz half width (:82-84), Doppler half width (:87-88),
// y (:91), the temperature correction of the strength with the TIPS-2017 partition-sum ratio qoft! (:95-101, :197-214:
// cubic spline of the isotopologue's table, evaluated at T_ref and T) and the grid window of the line (:104-107: linear
// interpolation grid -> index with the constant fill values 1 / n outside the grid, rounded half-to-even).  One thread per line; per layer only (p, T, vmr, wing) are
// kernel arguments.  Same expression order as the host route (absorption.line_prefactors); exp / pow come from the
// device math library, so the two routes agree to a few ulp, not bitwise.
// ---------------------------------------------------------------------------------------------------------------------
// searchsortedlast restricted to [0, n - 2]: the interval lo with a[lo] <= x < a[lo + 1] (lo = 0 below a[0], n - 2 from a[n - 1]
// on) of an ascending table a[0 .. n - 1].  r5: a guess from the mean spacing of a[first .. n - 1] and a walk of at most three
// steps replace the bisection wherever the table is (nearly) uniform -- the wavenumber grid always, the TIPS temperature
// knots from the second one on: two dependent memory latencies instead of log2(n).  k_line_prefactors_profile ran 76 dependent
// loads per thread (four window indices on a 22 801-point grid, two spline intervals on 251 knots): 23 us of the operating
// point's 132 us.  Any other table falls through to the bisection; the interval, hence every value computed from it, is the
// same either way.
__device__ __forceinline__ int locate_interval(const double *a, int n, double x, int first) {
  if (n < 2) return 0;
  const double a0 = a[first], a1 = a[n - 1];
  if (n - 1 > first && a1 > a0) {
    const double g = (x - a0) * ((double)(n - 1 - first) / (a1 - a0));
    int lo = first + (int)fmin(fmax(g, 0.0), (double)(n - 2 - first));
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
      if (a[lo] > x) { if (lo == 0) break; --lo; }
      else if (a[lo + 1] <= x) { if (lo == n - 2) break; ++lo; }
      else break;
    }
    const bool below = (lo == 0 && x < a[0]), above = (lo == n - 2 && a[n - 1] <= x);
    if (below || above || (a[lo] <= x && x < a[lo + 1])) return lo;
  }
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ double spline_eval(const double *t, const double *u, const double *z, int n1, double x) {
  // DataInterpolations.CubicSpline evaluation (restated in absorption.CubicSpline.__call__): interval by searchsortedlast
  const int i = min(max(locate_interval(t, n1 + 1, x, n1 >= 2 ? 1 : 0), 0), n1 - 1);  // knots t[0 .. n1]
  // the tables are Float32 (TIPS_2017.nc); products of two table entries are Float32 operations, as in the reference's
  // (and the host route's) evaluation -- only the terms that involve the Float64 argument x are Float64
  const float hf = (float)t[i + 1] - (float)t[i];
  const float cu = (float)u[i + 1] / hf - (float)z[i + 1] * hf / 6.0f;
  const float du = (float)u[i] / hf - (float)z[i] * hf / 6.0f;
  const double h6 = (double)(6.0f * hf);
  const double a = t[i + 1] - x, b = x - t[i];
  const double I = z[i] * (a * a * a) / h6 + z[i + 1] * (b * b * b) / h6;
  const double C = (double)cu * b;
  const double D = (double)du * a;
  return I + C + D;
}
__device__ __forceinline__ double interp_index(const double *grid, int n, double x, double fill) {
  // LinearInterpolation(grid, 1:n, extrapolation_bc = fill) (compute_absorption_cross_section.jl:60-61): linear inside the
  // grid, the CONSTANT `fill` on BOTH sides outside it (fill = 1 for the window start, n for the stop); grid ascending
  if (n == 1) return 1.0;
  if (x < grid[0] || x > grid[n - 1]) return fill;
  if (x == grid[n - 1]) return (double)n;
  const int lo = locate_interval(grid, n, x, 0);
  const double slope = 1.0 / (grid[lo + 1] - grid[lo]);
  return slope * (x - grid[lo]) + (double)(lo + 1);
}
__device__ __forceinline__ void line_prefactors_one(int j, const MomLineTable &tb, int nGrid, const double *grid, double p, double T,
                                                    double vmr, double wing, double cgd, double *nu, double *gd, double *yy, double *SS,
                                                    int *i0, int *i1, int *unsorted) {
#pragma clang fp contract(off)
  if (j >= tb.nLines) return;
  const double p_ref = 1013.25, t_ref = 296.0, c2 = 1.4387769, cLn2 = 0.6931471805599;
  const double nu0 = tb.nu0[j], E = tb.E[j];
  const double v = nu0 + p / p_ref * tb.d_air[j];
  const double gl = (tb.g_air[j] * (1 - vmr) * p / p_ref + tb.g_self[j] * vmr * p / p_ref) * pow(t_ref / T, tb.n_air[j]);
  const double g = cgd * nu0 / tb.sqw[j];
  double S = tb.S0[j];
  if (E != -1.0) {
    const int is = tb.iso[j];
    const double *t = tb.tT + (size_t)is * tb.nTmax, *u = tb.tQ + (size_t)is * tb.nTmax, *z = tb.tZ + (size_t)is * tb.nTmax;
    const int n1 = tb.nT[is] - 1;
    const double rate = spline_eval(t, u, z, n1, t_ref) / spline_eval(t, u, z, n1, T);
    const double corr = rate * exp(c2 * E * (1 / t_ref - 1 / T)) * (1 - exp(-c2 * nu0 / T)) / (1 - exp(-c2 * nu0 / t_ref));
    S = S * corr;
  }
  nu[j] = v;
  gd[j] = g;
  yy[j] = sqrt(cLn2) * gl / g;
  SS[j] = S;
  const int a = (int)rint(interp_index(grid, nGrid, v - wing, 1.0)), b = (int)rint(interp_index(grid, nGrid, v + wing, (double)nGrid));
  i0[j] = a;
  i1[j] = b;
  if (j > 0) {  // does the Voigt kernel's bisection apply?  (window starts and stops non-decreasing in the line index)
    const double vp = tb.nu0[j - 1] + p / p_ref * tb.d_air[j - 1];
    const int ap = (int)rint(interp_index(grid, nGrid, vp - wing, 1.0)), bp = (int)rint(interp_index(grid, nGrid, vp + wing, (double)nGrid));
    if (a < ap || b < bp) atomicOr(unsorted, 1);
  }
}
__global__ void k_line_prefactors(MomLineTable tb, int nGrid, const double *grid, double p, double T, double vmr, double wing,
                                  double cgd, double *nu, double *gd, double *yy, double *SS, int *i0, int *i1, int *unsorted) {
  line_prefactors_one(blockIdx.x * blockDim.x + threadIdx.x, tb, nGrid, grid, p, T, vmr, wing, cgd, nu, gd, yy, SS, i0, i1, unsorted);
}
// blockIdx.y = layer; prm = [p | T | cgd][Nz]; outputs at [k][z][cap] (see k_voigt_profile)
__global__ void k_line_prefactors_profile(MomLineTable tb, int Nz, size_t cap, int nGrid, const double *grid, const double *prm,
                                          double vmr, double wing, double *pf, int *win, int *unsorted) {
  const int z = blockIdx.y;
  const size_t lz = (size_t)z * cap, ks = (size_t)Nz * cap;
  line_prefactors_one(blockIdx.x * blockDim.x + threadIdx.x, tb, nGrid, grid, prm[z], prm[Nz + z], vmr, wing, prm[2 * Nz + z], pf + lz,
                      pf + ks + lz, pf + 2 * ks + lz, pf + 3 * ks + lz, win + lz, win + ks + lz, unsorted + z);
}
hipError_t mom_voigt_profile_launch(hipStream_t st, const MomLineTable &tb, int Nz, size_t cap, int nGrid, const double *grid,
                                    const double *prm, double vmr, double wing, double *pf, int *win, int *unsorted, double *tau_abs,
                                    const double *factor) {
  if (tb.nLines <= 0 || Nz <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_line_prefactors_profile, dim3((tb.nLines + 255) / 256, Nz), dim3(256), 0, st, tb, Nz, cap, nGrid, grid, prm, vmr,
                     wing, pf, win, unsorted);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_voigt_profile, dim3((nGrid + kBlock - 1) / kBlock, Nz), dim3(kBlock), 0, st, tb.nLines, Nz, cap, pf, win, nGrid,
                     grid, tau_abs, factor, unsorted);
  return hipGetLastError();
}
hipError_t mom_line_prefactors_launch(hipStream_t st, const MomLineTable &tb, int nGrid, const double *grid, double p, double T,
                                      double vmr, double wing, double cgd, double *nu, double *gd, double *y, double *S, int *i0,
                                      int *i1, int *unsorted) {
  if (tb.nLines <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_line_prefactors, dim3((tb.nLines + 255) / 256), dim3(256), 0, st, tb, nGrid, grid, p, T, vmr, wing, cgd, nu,
                     gd, y, S, i0, i1, unsorted);
  return hipGetLastError();
}

hipError_t mom_voigt_launch(hipStream_t st, int nLines, const double *nu, const double *gamma_d, const double *y,
                            const double *S, const int *i0, const int *i1, int nGrid, const double *grid, double *out,
                            double factor, int accumulate, int sorted) {
  hipLaunchKernelGGL(k_voigt, dim3((nGrid + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nLines, nu, gamma_d, y, S, i0, i1,
                     nGrid, grid, out, factor, accumulate, sorted);
  return hipGetLastError();
}

#define VCHK(call)                                                                                     \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      char buf__[384];                                                                                 \
      snprintf(buf__, sizeof buf__, "mom_voigt_xsec: %s failed: %s", #call, hipGetErrorString(e__));   \
      mom_set_global_error(buf__);                                                                     \
      rc = MOM_EHIP;                                                                                   \
      goto done;                                                                                       \
    }                                                                                                  \
  } while (0)

extern "C" int mom_voigt_xsec(int device, int nLines, const double *nu, const double *gamma_d, const double *y,
                              const double *S, const int *ind_start, const int *ind_stop, int nGrid, const double *grid,
                              double *sigma) {
  if (nLines < 0 || nGrid <= 0 || !grid || !sigma || (nLines > 0 && (!nu || !gamma_d || !y || !S || !ind_start || !ind_stop))) {
    mom_set_global_error("mom_voigt_xsec: bad argument (null pointer or non-positive size)");
    return MOM_EINVAL;
  }
  int sorted = 1;  // window starts and stops non-decreasing: the kernel finds a block's lines by bisection
  for (int j = 1; j < nLines; ++j)
    if (ind_start[j] < ind_start[j - 1] || ind_stop[j] < ind_stop[j - 1]) { sorted = 0; break; }
  for (int j = 0; j < nLines; ++j)
    if (ind_start[j] < 1 || ind_stop[j] > nGrid) {  // empty windows (start > stop) are allowed
      char buf[160];
      snprintf(buf, sizeof buf, "mom_voigt_xsec: line %d: window [%d, %d] outside the grid 1..%d", j + 1, ind_start[j], ind_stop[j], nGrid);
      mom_set_global_error(buf);
      return MOM_EINVAL;
    }
  int rc = MOM_OK;
  // one device allocation (lines | windows | grid | sigma) and one pinned-free staging copy per array on a private
  // stream; callers that evaluate many layers should use the handle-level mom_voigt_tau_abs, which keeps all of
  // this resident
  const size_t lb = (size_t)(nLines > 0 ? nLines : 1);
  const size_t bytes = (4 * lb + 2 * (size_t)nGrid) * sizeof(double) + 2 * lb * sizeof(int);
  char *base = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { mom_set_global_error("mom_voigt_xsec: no HIP device available"); return MOM_EHIP; }
  if (device < 0 || device >= ndev) { mom_set_global_error("mom_voigt_xsec: device index out of range"); return MOM_EINVAL; }
  VCHK(hipSetDevice(device));
  VCHK(hipStreamCreate(&st));
  VCHK(hipMalloc((void **)&base, bytes));
  {
    double *d_line = (double *)base, *d_grid = d_line + 4 * lb, *d_sig = d_grid + nGrid;
    int *d_win = (int *)(d_sig + nGrid);
    const double *hsrc[4] = {nu, gamma_d, y, S};
    for (int k = 0; k < 4; ++k)
      if (nLines) VCHK(hipMemcpyAsync(d_line + k * lb, hsrc[k], (size_t)nLines * sizeof(double), hipMemcpyHostToDevice, st));
    if (nLines) {
      VCHK(hipMemcpyAsync(d_win, ind_start, (size_t)nLines * sizeof(int), hipMemcpyHostToDevice, st));
      VCHK(hipMemcpyAsync(d_win + lb, ind_stop, (size_t)nLines * sizeof(int), hipMemcpyHostToDevice, st));
    }
    VCHK(hipMemcpyAsync(d_grid, grid, (size_t)nGrid * sizeof(double), hipMemcpyHostToDevice, st));
    VCHK(hipEventCreate(&e0));
    VCHK(hipEventCreate(&e1));
    VCHK(hipEventRecord(e0, st));
    VCHK(mom_voigt_launch(st, nLines, d_line, d_line + lb, d_line + 2 * lb, d_line + 3 * lb, d_win, d_win + lb, nGrid, d_grid,
                          d_sig, 1.0, 0, sorted));
    VCHK(hipEventRecord(e1, st));
    VCHK(hipMemcpyAsync(sigma, d_sig, (size_t)nGrid * sizeof(double), hipMemcpyDeviceToHost, st));
    VCHK(hipStreamSynchronize(st));
    float ms = 0.f;
    VCHK(hipEventElapsedTime(&ms, e0, e1));
    v_last_ms = ms;
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (base) (void)hipFree(base);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

// GPU time of the k_voigt launch of the last mom_voigt_xsec call on this thread (HIP events), ms.
extern "C" double mom_voigt_last_kernel_ms(void) { return v_last_ms; }
