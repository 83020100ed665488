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

#include <string>

#include "momcore.h"

namespace {

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
  cplx c;
  if (fabs(b.re) >= fabs(b.im)) {
    const double r = b.im / b.re, den = b.re + b.im * r;
    c.re = (a.re + a.im * r) / den; c.im = (a.im - a.re * r) / den;
  } else {
    const double r = b.re / b.im, den = b.re * r + b.im;
    c.re = (a.re * r + a.im) / den; c.im = (a.im * r - a.re) / den;
  }
  return c;
}

__constant__ double kW32A[32] = {
    2.5722534081245696e+00,  2.2635372999002676e+00,  1.8256696296324824e+00,  1.3455441692345453e+00,
    9.0192548936480144e-01,  5.4601397206393498e-01,  2.9544451071508926e-01,  1.4060716226893769e-01,
    5.7304403529837900e-02,  1.9006155784845689e-02,  4.5195411053501429e-03,  3.9259136070122748e-04,
    -2.4532980269928922e-04, -1.3075449254548613e-04, -2.1409619200870880e-05, 6.8210319440412389e-06,
    4.4015317319048931e-06,  4.2558331390536872e-07,  -4.1840763666294341e-07, -1.4813078891201116e-07,
    2.2930439569075392e-08,  2.3797557105844622e-08,  8.1248960947953431e-10,  -3.2080150458594088e-09,
    -5.2310170266050247e-10, 4.1537465934749353e-10,  1.1658312885903929e-10,  -5.5441820344468828e-11,
    -2.1542618451370239e-11, 8.0314997274316680e-12,  3.7424975634801558e-12,  -1.3031797863050087e-12};

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
    return (num.re * den.re + num.im * den.im) / (den.re * den.re + den.im * den.im);
  }
  const double L = 4.756828460010884;  // sqrt(32/sqrt(2))
  const cplx lpiz = {L - y, x}, lmiz = {L + y, -x};
  const cplx rec = cdiv(cplx{1.0, 0.0}, lmiz);
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

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock) k_voigt(int nLines, const double *__restrict__ nu,
                                                  const double *__restrict__ gamma_d, const double *__restrict__ y,
                                                  const double *__restrict__ S, const int *__restrict__ i0,
                                                  const int *__restrict__ i1, int nGrid,
                                                  const double *__restrict__ grid, double *__restrict__ sigma) {
  // per-line constants of the candidates, staged once per workgroup: centre, S c/gamma_d, c'/gamma_d, y and the
  // 0-based window -- the two divisions by gamma_d are per LINE here, not per evaluation (same expressions, same values)
  __shared__ double c_nu[kBlock], c_a[kBlock], c_b[kBlock], c_y[kBlock];
  __shared__ int c_lo[kBlock], c_hi[kBlock];
  __shared__ int wcount[kBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = blockIdx.x * kBlock;             // 0-based first grid index of this block
  const int g1 = min(nGrid, g0 + kBlock) - 1;     // last
  const int gi = g0 + tid;
  const double gx = (gi < nGrid) ? grid[gi] : 0.0;
  const double cSqrtLn2divSqrtPi = 0.469718639319144059835, cSqrtLn2 = 0.8325546111577;
  double acc = 0.0;
  for (int base = 0; base < nLines; base += kBlock) {
    const int j = base + tid;
    bool hit = false;
    int lo = 0, hi = -1;
    if (j < nLines) {
      lo = i0[j] - 1;
      hi = i1[j] - 1;
      hit = (lo <= g1) && (hi >= g0);
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
      c_lo[pos] = lo;
      c_hi[pos] = hi;
    }
    const int nc = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      if (gi >= c_lo[c] && gi <= c_hi[c] && gi < nGrid) acc += c_a[c] * w_hw32sd_re(c_b[c] * (gx - c_nu[c]), c_y[c]);
    }
    __syncthreads();
  }
  if (gi < nGrid) sigma[gi] = acc;
}

thread_local std::string v_err;
thread_local double v_last_ms = 0.0;

}  // namespace

#define VCHK(call)                                                        \
  do {                                                                    \
    hipError_t e__ = (call);                                              \
    if (e__ != hipSuccess) { v_err = hipGetErrorString(e__); rc = MOM_EHIP; goto done; } \
  } while (0)

extern "C" int mom_voigt_xsec(int device, int nLines, const double *nu, const double *gamma_d, const double *y,
                              const double *S, const int *ind_start, const int *ind_stop, int nGrid, const double *grid,
                              double *sigma) {
  if (nLines < 0 || nGrid <= 0 || !grid || !sigma || (nLines > 0 && (!nu || !gamma_d || !y || !S || !ind_start || !ind_stop)))
    return MOM_EINVAL;
  for (int j = 0; j < nLines; ++j)
    if (ind_start[j] < 1 || ind_stop[j] > nGrid) return MOM_EINVAL;  // empty windows (start > stop) are allowed
  int rc = MOM_OK;
  double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int *di[2] = {nullptr, nullptr};
  const double *hsrc[4] = {nu, gamma_d, y, S};
  const size_t lb = (size_t)(nLines > 0 ? nLines : 1);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MOM_EHIP;
  VCHK(hipSetDevice(device));
  for (int k = 0; k < 4; ++k) {
    VCHK(hipMalloc((void **)&d[k], lb * sizeof(double)));
    if (nLines) VCHK(hipMemcpy(d[k], hsrc[k], (size_t)nLines * sizeof(double), hipMemcpyHostToDevice));
  }
  VCHK(hipMalloc((void **)&di[0], lb * sizeof(int)));
  VCHK(hipMalloc((void **)&di[1], lb * sizeof(int)));
  if (nLines) {
    VCHK(hipMemcpy(di[0], ind_start, (size_t)nLines * sizeof(int), hipMemcpyHostToDevice));
    VCHK(hipMemcpy(di[1], ind_stop, (size_t)nLines * sizeof(int), hipMemcpyHostToDevice));
  }
  VCHK(hipMalloc((void **)&d[4], (size_t)nGrid * sizeof(double)));
  VCHK(hipMalloc((void **)&d[5], (size_t)nGrid * sizeof(double)));
  VCHK(hipMemcpy(d[4], grid, (size_t)nGrid * sizeof(double), hipMemcpyHostToDevice));
  {
    hipEvent_t e0, e1;
    VCHK(hipEventCreate(&e0));
    VCHK(hipEventCreate(&e1));
    VCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_voigt, dim3((nGrid + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, nLines, d[0], d[1], d[2], d[3],
                       di[0], di[1], nGrid, d[4], d[5]);
    VCHK(hipGetLastError());
    VCHK(hipEventRecord(e1, 0));
    VCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    VCHK(hipEventElapsedTime(&ms, e0, e1));
    v_last_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  VCHK(hipMemcpy(sigma, d[5], (size_t)nGrid * sizeof(double), hipMemcpyDeviceToHost));
done:
  for (int k = 0; k < 6; ++k) if (d[k]) (void)hipFree(d[k]);
  for (int k = 0; k < 2; ++k) if (di[k]) (void)hipFree(di[k]);
  return rc;
}

// GPU time of the k_voigt launch of the last mom_voigt_xsec call on this thread (HIP events), ms.
extern "C" double mom_voigt_last_kernel_ms(void) { return v_last_ms; }
