// momcore_f32.hip -- the Float32 build of the scene-level path (mom_create(..., dtype = 1)).
//
// The reference selects its float type with `float_type` (parameters_from_yaml.jl:160; batched ops for Float32 at
// gpu_batched.jl:45-58, the shapes of its own GPU tests: test/gpu_tests/gpu_batched_interaction.jl, n = 32, S = 20 000,
// Float32).  The same device templates as the Float64 library are compiled here with MOM_REAL = float in namespace
// momf: operators and sources in f32, products on v_mfma_f32_16x16x4_f32 (32 cycles per instruction per SIMD: twice
// the f64 rate), LDS images half the size.  What runs: the fused per-layer kernels of the general path (LDS-resident
// for N <= 64, generic mode above), layer-sweep mode, the surface layer (all three surface kinds) with HDRF/BHR, and
// post-processing; r4: the strip-chained images of the 8-wave build (N = 44, 52, 56, 60: momcore_strip.hip compiled for
// float, momf_strip<KS>_launch_layer), the (I,Q) reduction of moment 0 (a nested sub-scene: momf_scene::sub) and the padding of
// other edges to the strip sizes (strip_pad_f), and the operator-level API on dtype = 1 handles (mom_ops.hpp compiled for float:
// mom_elemental ... mom_download).  Not built for f32: multi-sensor, RRS, the device-side optics route.
//
// The C ABI keeps Float64 host arrays for both dtypes (a Float32 Julia host passes Float64.(x) and converts back):
// inputs are rounded to f32 on upload, outputs widened on download.
#define MOM_REAL float
#define MOM_REAL_IS_FLOAT 1
#define MOM_NS momf
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "momcore.h"

#include "mom_diag.hpp"
#include "mom_entry.hpp"
#include "mom_ops.hpp"
#include "mom_host.hpp"

using namespace momf;

namespace {

struct PostArgsF {
  int N, nS, S, M, nVza, hdr_all, zeroT_hi, m_first;  // M moments starting at Fourier index m_first
  const int *node;
  const double *cos_mphi, *sin_mphi;
  const float *J0p, *J0m, *hdrJ0, *hdrJm;
  float *R, *T, *hdr;
};
// postprocessing_vza! / postprocessing_vza_hdrf! (postprocessing_vza.jl:9-93); the azimuthal weights stay Float64
// like the reference's host-side `bigCS`, the sources are f32
__global__ void k_postprocess_f32(PostArgsF a) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.nVza * a.nS * a.S;
  if (idx >= total) return;
  const int v = (int)(idx % a.nVza);
  const int k = (int)((idx / a.nVza) % a.nS);
  const size_t s = idx / ((size_t)a.nVza * a.nS);
  const int row = (a.node[v] - 1) * a.nS + k;
  float r = 0.f, t = 0.f, h = 0.f;
  for (int mr = 0; mr < a.M; ++mr) {
    const int m = a.m_first + mr;
    const double weight = (m == 0) ? 0.5 : 1.0;
    const float cs = (float)(weight * ((k < 2) ? a.cos_mphi[v + (size_t)a.nVza * m] : a.sin_mphi[v + (size_t)a.nVza * m]));
    const size_t o = row + (size_t)a.N * (s + (size_t)a.S * mr);
    r += cs * a.J0m[o];
    if (!(a.zeroT_hi && m > 0)) t += cs * a.J0p[o];
    if (m == 0) h += cs * a.hdrJ0[row + (size_t)a.N * s];
    else if (a.hdr_all) h += cs * a.hdrJm[row + (size_t)a.N * (s + (size_t)a.S * m)];
  }
  a.R[idx] = r;
  a.T[idx] = t;
  a.hdr[idx] = h;
}

// m = 0 reduction: the (I,Q) sub-scene's spectra [nVza,nS0,S] and BHR [nS0,S] are added to / become the first nS0 Stokes
// components of the full scene's (whose own moments start at m = 1); components >= nS0 get no m = 0 term (exactly 0)
struct CombineArgsF {
  int nVza, nS, nS0, S;
  float *R, *T, *hdr, *bhr_uw, *bhr_dw;
  const float *R0, *T0, *hdr0, *bhr0_uw, *bhr0_dw;
};
__global__ void k_combine_m0_f32(CombineArgsF a) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.nVza * a.nS * a.S;
  if (idx < total) {
    const int v = (int)(idx % a.nVza), k = (int)((idx / a.nVza) % a.nS);
    const size_t s = idx / ((size_t)a.nVza * a.nS);
    if (k < a.nS0) {
      const size_t o = v + (size_t)a.nVza * (k + (size_t)a.nS0 * s);
      a.R[idx] += a.R0[o];
      a.T[idx] += a.T0[o];
      a.hdr[idx] += a.hdr0[o];
    }
  }
  if (idx < (size_t)a.nS * a.S) {
    const int k = (int)(idx % a.nS);
    const size_t s = idx / a.nS;
    a.bhr_uw[idx] = (k < a.nS0) ? a.bhr0_uw[k + (size_t)a.nS0 * s] : 0.f;
    a.bhr_dw[idx] = (k < a.nS0) ? a.bhr0_dw[k + (size_t)a.nS0 * s] : 0.f;
  }
}

template <class T>
hipError_t dmallocf(T **p, size_t count) { return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)); }

// batch_inv! / batched_mul for Float32 arrays (gpu_batched.jl:45-58, 90-97: cuBLAS Sgetrf/Sgetri, Sgemm)
struct BlasArgsF {
  int N, S;
  const float *A, *B;
  float *C;
  float *scratch;
  int *info;
};

template <bool LDSM>
__global__ void __launch_bounds__(kThreads) k_batch_inv_f32(BlasArgsF a) {
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
__global__ void __launch_bounds__(kThreads) k_batched_mul_f32(BlasArgsF a) {
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
    float *C = a.C + NN * pt;
    wg_gemm<false>(N, ElP{c.P, ld}, ElP{c.Q, ld}, [=](int i, int j, float v) { C[i + (size_t)j * N] = v; });
    __syncthreads();
  }
}

std::vector<float> tof(const double *src, size_t n) {
  std::vector<float> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = (float)src[i];
  return v;
}

}  // namespace

// Edges with a Float32 strip-chained image; other edges are padded with up to 4 dummy stream entries (mu = 1, weight 0, zero
// rows and columns in every phase-matrix basis and BRDF matrix: decoupled exactly, see strip_pad in momcore.hip) to reach one
constexpr int kPadMaxF = 4;
static bool strip_size_f(int N) { return N == 36 || N == 40 || N == 44 || N == 52 || N == 56 || N == 60; }
static int strip_pad_f(int N) {
  if (strip_size_f(N)) return N;
  for (int p = N + 1; p <= N + kPadMaxF; ++p)
    if (strip_size_f(p)) return p;
  return N;
}
// [N,N,B] -> [Nk,Nk,B], zero padded
static std::vector<double> pad_blocks_f(const double *src, int N, int Nk, size_t B) {
  std::vector<double> out((size_t)Nk * Nk * B, 0.0);
  for (size_t b = 0; b < B; ++b)
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) out[i + (size_t)Nk * (j + (size_t)Nk * b)] = src[i + (size_t)N * (j + (size_t)N * b)];
  return out;
}

struct momf_scene {
  int device = 0, N = 0, nS = 0, S = 0, Mmax = 0;
  int Nu = 0, Nmax = 0;  // Nu: the caller's operator edge; N: the kernels' (strip_pad_f, decided in momf_set_streams); Nmax: allocated
  // m = 0 reduction (include/momcore.h, MOM_OPT_M0_REDUCTION): moment 0 runs as its own scene `sub` on the (I,Q) streams,
  // this scene's launches start at Fourier index m_first = 1
  momf_scene *sub = nullptr;
  int m_first = 0;
  bool opt_m0 = true, opt_pad = true;
  // operator-level API (mom_ops.hpp): added / surface / composite layers in the reference's [N,N,S] layout, allocated on first use
  float *op_added[6] = {}, *op_surf[6] = {}, *op_comp[6] = {}, *op_vec[4] = {}, *op_Z[2] = {};
  size_t op_Zcap = 0;
  float *blas_buf[4] = {};  // mom_batch_inv / mom_batched_mul: A, B, C, generic-mode scratch (grow-only)
  size_t blas_cap[4] = {};
  bool op_ready = false, op_comp_set = false;
  std::vector<double> hd_mu, hd_wt, hd_sg;  // the caller's Float64 streams (the sub-scene is cut from them)
  double hd_I0[4] = {}, hd_D[4] = {}, hd_mu0 = 0;
  hipStream_t stream = nullptr;
  DevStreams q{};
  float *d_mu = nullptr, *d_wt = nullptr, *d_sg = nullptr;
  float *comp[6] = {};
  float *d_tau = nullptr, *d_varpi = nullptr, *d_zw = nullptr, *d_tau_sum = nullptr, *d_Zpp = nullptr, *d_Zmp = nullptr;
  float *d_R = nullptr, *d_T = nullptr, *d_hdr = nullptr, *d_hdrJ = nullptr, *d_hdrJm = nullptr, *d_bhr_uw = nullptr,
        *d_bhr_dw = nullptr, *d_scratch = nullptr, *d_Rsurf = nullptr, *d_albedo_spec = nullptr;
  double *d_cos = nullptr, *d_sin = nullptr;
  float *d_smtab = nullptr;               // N <= 4: the three stream-pair tables of the lane-per-point kernel
  float *d_smpart = nullptr;              // ... and the per-moment terms of R_SFI / T_SFI of its (point, moment) form
  size_t smpart_cap = 0;
  int *d_node = nullptr, *d_info = nullptr, *d_nd = nullptr;  // d_nd: ndoubl per layer for the wave-per-point kernel
  bool pack = true;                       // MOM_OPT_SMALL_N = 1 (2: one point per wavefront)
  bool small_n = true;                    // MOM_OPT_SMALL_N: 4 < N <= 32 on the wave-per-point kernels (mom_wave.hip, Float32 build)
  int Nz = 0, K = 0, M = 0, nVza = 0, surf_kind = 0, G = 1024;
  float albedo = 0.f;
  std::vector<int> nd, iface;
  std::vector<float> h_mu;
  bool lds = true, force_generic = false, sweep = true, strips = true, w4 = true;  // w4: MOM_OPT_SMALL_WG
  hipEvent_t ev[4] = {};
  int launches = 0;
  std::string err;
};

#define FCHK(s, call)                                                                                         \
  do {                                                                                                        \
    hipError_t e__ = (call);                                                                                  \
    if (e__ != hipSuccess) {                                                                                  \
      char b__[384];                                                                                          \
      snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);  \
      (s)->err = b__;                                                                                         \
      return MOM_EHIP;                                                                                        \
    }                                                                                                         \
  } while (0)

const char *momf_error(const momf_scene *s) { return s->err.c_str(); }

int momf_create(momf_scene **out, int device, hipStream_t stream, int N, int nS, int S, int max_m, int *d_info) {
  momf_scene *s = new momf_scene();
  *out = s;
  s->device = device; s->N = s->Nu = N; s->nS = nS; s->S = S; s->Mmax = max_m; s->stream = stream; s->d_info = d_info;
  const int Nm = s->Nmax = strip_pad_f(N);
  s->lds = N <= 64;
  FCHK(s, hipSetDevice(device));
  FCHK(s, dmallocf(&s->d_mu, Nm));
  FCHK(s, dmallocf(&s->d_wt, Nm));
  FCHK(s, dmallocf(&s->d_sg, Nm));
  for (int k = 0; k < 6; ++k) {
    const size_t per = (k < 4) ? (size_t)comp_pitch(Nm) * Nm : (size_t)Nm;
    FCHK(s, dmallocf(&s->comp[k], per * S * max_m));
    FCHK(s, hipMemsetAsync(s->comp[k], 0, per * S * max_m * sizeof(float), stream));
  }
  const size_t scr = (size_t)s->G * kGenericBufs * mat_elems(Nm) + (size_t)ld_for(Nm) * np_for(Nm);
  FCHK(s, dmallocf(&s->d_scratch, scr));
  FCHK(s, hipMemsetAsync(s->d_scratch, 0, scr * sizeof(float), stream));
  for (int k = 0; k < 4; ++k) FCHK(s, hipEventCreate(&s->ev[k]));
  FCHK(s, hipStreamSynchronize(stream));
  return MOM_OK;
}

void momf_destroy(momf_scene *s) {
  if (!s) return;
  momf_destroy(s->sub);
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  fr(s->d_mu); fr(s->d_wt); fr(s->d_sg);
  for (int k = 0; k < 6; ++k) fr(s->comp[k]);
  fr(s->d_tau); fr(s->d_varpi); fr(s->d_zw); fr(s->d_tau_sum); fr(s->d_Zpp); fr(s->d_Zmp); fr(s->d_R); fr(s->d_hdr);
  fr(s->d_hdrJ); fr(s->d_hdrJm); fr(s->d_bhr_uw); fr(s->d_bhr_dw); fr(s->d_scratch); fr(s->d_Rsurf); fr(s->d_albedo_spec);
  fr(s->d_cos); fr(s->d_sin); fr(s->d_node); fr(s->d_nd); fr(s->d_smtab); fr(s->d_smpart);
  for (int k = 0; k < 6; ++k) { fr(s->op_added[k]); fr(s->op_surf[k]); fr(s->op_comp[k]); }
  for (int k = 0; k < 4; ++k) fr(s->op_vec[k]);
  fr(s->op_Z[0]); fr(s->op_Z[1]);
  for (int k = 0; k < 4; ++k) fr(s->blas_buf[k]);
  for (int k = 0; k < 4; ++k) if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
  delete s;
}

void momf_set_options(momf_scene *s, int inv_mode, int force_generic, int sweep, int small_n, int m0, int pad, int w4) {
  s->w4 = w4 != 0;
  s->small_n = small_n != 0;
  s->pack = small_n == 1;
  s->q.inv_mode = inv_mode;
  s->force_generic = force_generic != 0;
  s->lds = (s->N <= 64) && !s->force_generic;
  s->sweep = sweep != 0;
  s->opt_m0 = m0 != 0;
  s->opt_pad = pad != 0;
  if (s->sub) momf_set_options(s->sub, inv_mode, force_generic, sweep, small_n, 0, 0, w4);
}

int momf_set_streams(momf_scene *s, const double *mu, const double *wt, const double *sg, int imu0, double mu0,
                     const double *I0, const double *D, int regular) {
  const int Nu = s->Nu;
  s->hd_mu.assign(mu, mu + Nu); s->hd_wt.assign(wt, wt + Nu); s->hd_sg.assign(sg, sg + Nu);
  for (int k = 0; k < 4; ++k) { s->hd_I0[k] = (k < s->nS) ? I0[k] : 0.0; s->hd_D[k] = (k < s->nS) ? D[k] : 1.0; }
  s->hd_mu0 = mu0;
  DevStreams &q = s->q;
  for (int k = 0; k < 4; ++k) { q.I0[k] = (float)s->hd_I0[k]; q.D[k] = (float)s->hd_D[k]; }
  q.nS = s->nS; q.imu0 = imu0; q.mu0 = (float)mu0; q.regular = regular;
  // `regular` was decided on the Float64 streams; two distinct f64 nodes may round to one f32 value, which only makes
  // more stream pairs take the equal-mu branch of get_elem_rt! -- exactly what a Float32 reference run does
  return MOM_OK;
}

// The kernel edge and the device copies of the streams, at scene time (the options that decide the edge may be set after
// momf_set_streams): padded to a strip-chained image where one is within reach; edges up to 32 belong to the wave-per-point
// kernel, which takes the operators as they are
static int apply_streams(momf_scene *s) {
  const int Nu = s->Nu;
  if ((int)s->hd_mu.size() != Nu) { s->err = "Float32 scene: streams not set"; return MOM_ESTATE; }
  const int N = s->N = (s->opt_pad && !s->force_generic && !(Nu <= 32 && s->small_n)) ? strip_pad_f(Nu) : Nu;
  s->lds = (N <= 64) && !s->force_generic;
  std::vector<float> fm = tof(s->hd_mu.data(), Nu), fw = tof(s->hd_wt.data(), Nu), fs = tof(s->hd_sg.data(), Nu);
  fm.resize(N, 1.f); fw.resize(N, 0.f); fs.resize(N, 1.f);  // dummy entries
  s->h_mu = fm;
  FCHK(s, hipMemcpyAsync(s->d_mu, fm.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipMemcpyAsync(s->d_wt, fw.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipMemcpyAsync(s->d_sg, fs.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  s->q.mu = s->d_mu; s->q.wt = s->d_wt; s->q.sg = s->d_sg; s->q.N = N;
  return MOM_OK;
}

template <class T, class U>
static int upload_f(momf_scene *s, T **dst, const U *src, size_t n) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  FCHK(s, dmallocf(dst, n));
  std::vector<T> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = (T)src[i];
  FCHK(s, hipMemcpyAsync(*dst, v.data(), n * sizeof(T), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  return MOM_OK;
}

// the layer optics assembled on the device in Float64 (mom_scene_set_optics): rounded to the scene's Float32 there
__global__ void k_cvt_d2f(const double *src, float *dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
static int convert_f(momf_scene *s, float **dst, const double *d_src, size_t n) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  FCHK(s, dmallocf(dst, n));
  hipLaunchKernelGGL(k_cvt_d2f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, d_src, *dst, n);
  FCHK(s, hipGetLastError());
  return MOM_OK;
}

static bool wave_applies_f32(const momf_scene *s);

static int scene_set_impl(momf_scene *s, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                          const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *tau_sum,
                          double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi, bool dev_layers);

int momf_scene_set(momf_scene *s, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                   const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *tau_sum,
                   double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi) {
  return scene_set_impl(s, Nz, K, M, tau, varpi, zw, Zpp, Zmp, ndoubl, iface, tau_sum, albedo, nVza, node, cos_mphi, sin_mphi, false);
}
// the same with tau, varpi, zw, tau_sum as DEVICE Float64 arrays (the device-side layer optics of mom_scene_set_optics)
int momf_scene_set_dev(momf_scene *s, int Nz, int K, int M, const double *d_tau, const double *d_varpi, const double *d_zw,
                       const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *d_tau_sum,
                       double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi) {
  return scene_set_impl(s, Nz, K, M, d_tau, d_varpi, d_zw, Zpp, Zmp, ndoubl, iface, d_tau_sum, albedo, nVza, node, cos_mphi, sin_mphi, true);
}

static int scene_set_impl(momf_scene *s, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                          const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *tau_sum,
                          double albedo, int nVza, const int *node, const double *cos_mphi, const double *sin_mphi, bool dev_layers) {
  FCHK(s, hipSetDevice(s->device));
  int rc;
  if ((rc = apply_streams(s))) return rc;
  const int N = s->N, Nu = s->Nu, nS = s->nS;
  const size_t S = s->S, NN = (size_t)N * N;
  if (dev_layers) {
    if ((rc = convert_f(s, &s->d_tau, tau, S * Nz))) return rc;
    if ((rc = convert_f(s, &s->d_varpi, varpi, S * Nz))) return rc;
    if ((rc = convert_f(s, &s->d_zw, zw, (size_t)K * S * Nz))) return rc;
    if ((rc = convert_f(s, &s->d_tau_sum, tau_sum, S * (Nz + 1)))) return rc;
  } else {
    if ((rc = upload_f(s, &s->d_tau, tau, S * Nz))) return rc;
    if ((rc = upload_f(s, &s->d_varpi, varpi, S * Nz))) return rc;
    if ((rc = upload_f(s, &s->d_zw, zw, (size_t)K * S * Nz))) return rc;
    if ((rc = upload_f(s, &s->d_tau_sum, tau_sum, S * (Nz + 1)))) return rc;
  }
  if (N == Nu) {
    if ((rc = upload_f(s, &s->d_Zpp, Zpp, NN * K * M))) return rc;
    if ((rc = upload_f(s, &s->d_Zmp, Zmp, NN * K * M))) return rc;
  } else {
    const std::vector<double> zp = pad_blocks_f(Zpp, Nu, N, (size_t)K * M), zm = pad_blocks_f(Zmp, Nu, N, (size_t)K * M);
    if ((rc = upload_f(s, &s->d_Zpp, zp.data(), zp.size()))) return rc;
    if ((rc = upload_f(s, &s->d_Zmp, zm.data(), zm.size()))) return rc;
  }
  if ((rc = upload_f(s, &s->d_node, node, (size_t)nVza))) return rc;
  if ((rc = upload_f(s, &s->d_cos, cos_mphi, (size_t)nVza * M))) return rc;
  if ((rc = upload_f(s, &s->d_sin, sin_mphi, (size_t)nVza * M))) return rc;
  auto renew = [&](float **p, size_t n) -> hipError_t { if (*p) { (void)hipFree(*p); *p = nullptr; } return dmallocf(p, n); };
  const size_t nout = (size_t)nVza * nS * S;
  FCHK(s, renew(&s->d_R, 2 * nout));
  s->d_T = s->d_R + nout;
  FCHK(s, renew(&s->d_hdr, nout));
  FCHK(s, renew(&s->d_hdrJ, (size_t)N * S));
  FCHK(s, renew(&s->d_bhr_uw, (size_t)nS * S));
  FCHK(s, renew(&s->d_bhr_dw, (size_t)nS * S));
  s->Nz = Nz; s->K = K; s->M = M; s->nVza = nVza; s->albedo = (float)albedo; s->surf_kind = 0;
  s->nd.assign(ndoubl, ndoubl + Nz);
  s->iface.assign(iface, iface + Nz);
  // ---- m = 0 reduction (include/momcore.h): the same conditions as the Float64 driver, checked on the data, bitwise.
  // Scenes that run on the lane- or wave-per-point kernels (all moments in one launch) keep moment 0 there.
  s->m_first = 0;
  const int Nq = Nu / nS;
  bool ok = s->opt_m0 && nS >= 3 && s->q.regular && !(Nu <= 4 && s->small_n) && !wave_applies_f32(s);
  for (int k = 2; k < nS && ok; ++k) ok = (s->hd_I0[k] == 0.0);
  for (int kb = 0; kb < K && ok; ++kb)
    for (int j = 0; j < Nu && ok; ++j)
      for (int i = 0; i < Nu; ++i) {
        if (((i % nS) < 2) == ((j % nS) < 2)) continue;
        const size_t o = i + (size_t)Nu * (j + (size_t)Nu * kb);  // moment 0 block
        if (Zpp[o] != 0.0 || Zmp[o] != 0.0) { ok = false; break; }
      }
  if (!ok) {
    momf_destroy(s->sub);
    s->sub = nullptr;
    return MOM_OK;
  }
  // N0r real entries; the kernels run on N0 >= N0r (dummy entries at the end) unless the wave-per-point kernel takes the scene
  const int nS0 = 2, N0r = nS0 * Nq;
  const int N0 = (s->opt_pad && !s->force_generic && !(N0r <= 32 && s->small_n)) ? strip_pad_f(N0r) : N0r;
  if (s->sub && !(s->sub->Nu == N0 && s->sub->S == s->S)) { momf_destroy(s->sub); s->sub = nullptr; }
  if (!s->sub) {
    if ((rc = momf_create(&s->sub, s->device, s->stream, N0, nS0, s->S, 1, s->d_info))) {
      s->err = s->sub->err;
      momf_destroy(s->sub);
      s->sub = nullptr;
      return rc;
    }
  }
  momf_scene *u = s->sub;
  momf_set_options(u, s->q.inv_mode, s->force_generic, s->sweep, s->small_n, 0, 0, s->w4);
  u->strips = s->strips;
  std::vector<double> mu0v(N0, 1.0), wt0v(N0, 0.0), sg0v(N0, 1.0), zp((size_t)N0 * N0 * K, 0.0), zm((size_t)N0 * N0 * K, 0.0);
  auto full = [&](int i0) { return (i0 / nS0) * nS + (i0 % nS0); };
  for (int i = 0; i < N0r; ++i) { mu0v[i] = s->hd_mu[full(i)]; wt0v[i] = s->hd_wt[full(i)]; }
  for (int kb = 0; kb < K; ++kb)
    for (int j = 0; j < N0r; ++j)
      for (int i = 0; i < N0r; ++i) {
        const size_t src = full(i) + (size_t)Nu * (full(j) + (size_t)Nu * kb);
        zp[i + (size_t)N0 * (j + (size_t)N0 * kb)] = Zpp[src];
        zm[i + (size_t)N0 * (j + (size_t)N0 * kb)] = Zmp[src];
      }
  const double one[4] = {1.0, 1.0, 1.0, 1.0};
  auto sub_fail = [&](int code) { s->err = u->err; return code; };
  if ((rc = momf_set_streams(u, mu0v.data(), wt0v.data(), sg0v.data(), s->q.imu0, s->hd_mu0, s->hd_I0, one, s->q.regular)))
    return sub_fail(rc);
  if ((rc = scene_set_impl(u, Nz, K, 1, tau, varpi, zw, zp.data(), zm.data(), ndoubl, iface, tau_sum, albedo, nVza, node, cos_mphi,
                           sin_mphi, dev_layers)))
    return sub_fail(rc);
  s->m_first = 1;
  return MOM_OK;
}

int momf_scene_set_surface(momf_scene *s, int kind, int M, const double *Rsurf, const double *albedo_spec) {
  FCHK(s, hipSetDevice(s->device));
  const int N = s->N, Nu = s->Nu, nS = s->nS;
  int rc;
  if (kind == 1) {
    if (N == Nu) {
      if ((rc = upload_f(s, &s->d_Rsurf, Rsurf, (size_t)N * N * M))) return rc;
    } else {
      const std::vector<double> rp = pad_blocks_f(Rsurf, Nu, N, (size_t)M);
      if ((rc = upload_f(s, &s->d_Rsurf, rp.data(), rp.size()))) return rc;
    }
    if (s->d_hdrJm) { (void)hipFree(s->d_hdrJm); s->d_hdrJm = nullptr; }
    FCHK(s, dmallocf(&s->d_hdrJm, (size_t)N * s->S * M));
    if (s->m_first) {
      // moment 0 runs on the (I,Q) sub-scene: its surface matrix must not couple (I,Q) with (U,V) either
      const int nS0 = s->sub->nS, N0 = s->sub->Nu;
      std::vector<double> r0((size_t)N0 * N0, 0.0);
      for (int j = 0; j < Nu; ++j)
        for (int i = 0; i < Nu; ++i) {
          const bool iq_i = (i % nS) < nS0, iq_j = (j % nS) < nS0;
          const double v = Rsurf[i + (size_t)Nu * j];
          if (iq_i != iq_j && v != 0.0) {
            s->err = "mom_scene_set_surface: the m = 0 BRDF matrix couples (I,Q) with (U,V); set MOM_OPT_M0_REDUCTION = 0 before "
                     "mom_scene_set for this surface";
            return MOM_EINVAL;
          }
          if (iq_i && iq_j) r0[(i / nS) * nS0 + (i % nS) + (size_t)N0 * ((j / nS) * nS0 + (j % nS))] = v;
        }
      if ((rc = momf_scene_set_surface(s->sub, 1, 1, r0.data(), nullptr))) { s->err = s->sub->err; return rc; }
    }
  } else if (kind == 2) {
    if ((rc = upload_f(s, &s->d_albedo_spec, albedo_spec, (size_t)s->S))) return rc;
    if (s->m_first && (rc = momf_scene_set_surface(s->sub, 2, 1, nullptr, albedo_spec))) { s->err = s->sub->err; return rc; }
  } else if (s->m_first) {
    s->sub->surf_kind = 0;
  }
  s->surf_kind = kind;
  return MOM_OK;
}

template <class K>
static hipError_t allow(K kernel, size_t bytes) { return mom_allow_lds(reinterpret_cast<const void *>(kernel), bytes); }

hipError_t momwf_launch_sweep(const void *args, hipStream_t st);  // mom_wave.hip built with -DMOMW_FLOAT
// momcore_strip.hip built for float (Makefile: momcore_fs<KS>.o)
hipError_t momf_strip11_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t momf_strip13_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t momf_strip14_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
hipError_t momf_strip15_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st);
// ... and their 4-wave builds (momcore_f4s<KS>.o, namespace momf4): two workgroups per CU
#define MOMF4_DECL(KS)                                                                                          \
  hipError_t momf4_strip##KS##_launch_layer(const void *layer_args, int iface, int grid, size_t smem, hipStream_t st); \
  size_t momf4_strip##KS##_lds_bytes();
MOMF4_DECL(9) MOMF4_DECL(10) MOMF4_DECL(11) MOMF4_DECL(13) MOMF4_DECL(14) MOMF4_DECL(15)
#undef MOMF4_DECL

// 4 < N <= 32: one spectral point per wavefront, operators in MFMA-layout registers, the whole run in ONE launch -- the
// Float32 build of momw::k_wsweep (the Float64 path: rt_run_wave in momcore.hip).  Covers ScatteringInterface_11 on every
// layer after the first and at the surface.
static bool wave_applies_f32(const momf_scene *s) {
  if (!(s->N > 4 && s->N <= 32 && s->small_n && !s->force_generic && s->nVza * s->nS <= 256)) return false;
  for (int z = 1; z < s->Nz; ++z)
    if (s->iface[z] != 3) return false;
  return s->iface[s->Nz - 1] == 3;
}
static int rt_run_wave_f32(momf_scene *s) {
  if (!s->d_nd) FCHK(s, hipMalloc((void **)&s->d_nd, sizeof(int) * kMaxSweepLayers * 4));
  if (s->Nz > kMaxSweepLayers * 4) { s->err = "Float32 wave sweep: too many layers"; return MOM_EINVAL; }
  FCHK(s, hipMemcpyAsync(s->d_nd, s->nd.data(), sizeof(int) * s->Nz, hipMemcpyHostToDevice, s->stream));
  MomWaveSweepArgsF a{};
  a.N = s->N; a.S = s->S; a.M = s->M; a.K = s->K; a.Nz = s->Nz; a.nVza = s->nVza; a.nS = s->nS; a.imu0 = s->q.imu0;
  a.inv_mode = s->q.inv_mode;
  a.pad = s->pack ? (s->N == 5 ? 3 : (s->N >= 6 && s->N <= 8 ? 2 : 1)) : 1;  // points per wavefront (k_wsweep's PK)
  a.mu0 = s->q.mu0; a.albedo = s->albedo;
  for (int k = 0; k < 4; ++k) { a.I0[k] = s->q.I0[k]; a.D[k] = s->q.D[k]; }
  a.mu = s->d_mu; a.wt = s->d_wt; a.sg = s->d_sg;
  a.Zpp = s->d_Zpp; a.Zmp = s->d_Zmp;
  a.nd = s->d_nd; a.node = s->d_node; a.cos_mphi = s->d_cos; a.sin_mphi = s->d_sin;
  a.tau = s->d_tau; a.varpi = s->d_varpi; a.zw = s->d_zw; a.tau_sum = s->d_tau_sum;
  a.R = s->d_R; a.T = s->d_T; a.hdr = s->d_hdr; a.bhr_uw = s->d_bhr_uw; a.bhr_dw = s->d_bhr_dw;
  a.info = s->d_info;
  a.surf_kind = s->surf_kind; a.Rsurf = s->d_Rsurf; a.albedo_spec = s->d_albedo_spec;
  FCHK(s, hipEventRecord(s->ev[0], s->stream));
  FCHK(s, momwf_launch_sweep(&a, s->stream));
  for (int k = 1; k < 4; ++k) FCHK(s, hipEventRecord(s->ev[k], s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));  // s->nd may be rewritten by the next scene_set
  s->launches = 1;
  return MOM_OK;
}

hipError_t momsmf_launch_sweep(const void *args, int N, hipStream_t st);  // mom_small.hip built with -DMOMS_FLOAT

// N <= 4: one spectral point per lane, the whole run in ONE launch (the Float64 path: rt_run_small in momcore.hip)
static int rt_run_small_f32(momf_scene *s) {
  const int N = s->N, Nz = s->Nz;
  if (!s->d_smtab) FCHK(s, hipMalloc((void **)&s->d_smtab, sizeof(float) * 48));
  if (!s->d_nd) FCHK(s, hipMalloc((void **)&s->d_nd, sizeof(int) * kMaxSweepLayers * 4));
  if (2 * Nz > kMaxSweepLayers * 4) { s->err = "Float32 lane sweep: too many layers"; return MOM_EINVAL; }
  float tab[48] = {0};
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {  // the expressions of elemental.jl:176-186 in Float32, evaluated once
      const float mui = s->h_mu[i], muj = s->h_mu[j];
      tab[i + N * j] = muj / (mui + muj);
      tab[16 + i + N * j] = muj / (mui - muj);
      tab[32 + i + N * j] = (1 / mui) + (1 / muj);
    }
  FCHK(s, hipMemcpyAsync(s->d_smtab, tab, sizeof tab, hipMemcpyHostToDevice, s->stream));
  std::vector<int> v(s->nd);
  v.insert(v.end(), s->iface.begin(), s->iface.end());
  FCHK(s, hipMemcpyAsync(s->d_nd, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  MomSmallSweepArgsF a{};
  a.S = s->S; a.M = s->M; a.K = s->K; a.Nz = Nz; a.nVza = s->nVza; a.nS = s->nS; a.imu0 = s->q.imu0;
  a.mu0 = s->q.mu0; a.albedo = s->albedo;
  for (int k = 0; k < 4; ++k) { a.I0[k] = s->q.I0[k]; a.D[k] = s->q.D[k]; }
  a.mu = s->d_mu; a.wt = s->d_wt; a.sg = s->d_sg;
  a.F1 = s->d_smtab; a.F2 = s->d_smtab + 16; a.SI = s->d_smtab + 32;
  a.Zpp = s->d_Zpp; a.Zmp = s->d_Zmp;
  a.nd = s->d_nd; a.iface = s->d_nd + Nz; a.node = s->d_node; a.cos_mphi = s->d_cos; a.sin_mphi = s->d_sin;
  a.tau = s->d_tau; a.varpi = s->d_varpi; a.zw = s->d_zw; a.tau_sum = s->d_tau_sum;
  a.R = s->d_R; a.T = s->d_T; a.hdr = s->d_hdr; a.bhr_uw = s->d_bhr_uw; a.bhr_dw = s->d_bhr_dw;
  a.info = s->d_info;
  if (a.M > 1 && s->pack) {  // one (point, moment) per lane; MOM_OPT_SMALL_N = 2: one point per lane
    const size_t need = (size_t)a.M * 2 * a.nVza * a.nS * a.S;
    if (need > s->smpart_cap) {
      if (s->d_smpart) { FCHK(s, hipStreamSynchronize(s->stream)); (void)hipFree(s->d_smpart); s->d_smpart = nullptr; s->smpart_cap = 0; }
      FCHK(s, hipMalloc((void **)&s->d_smpart, need * sizeof(float)));
      s->smpart_cap = need;
    }
    a.part = s->d_smpart;
  }
  FCHK(s, hipEventRecord(s->ev[0], s->stream));
  FCHK(s, momsmf_launch_sweep(&a, N, s->stream));
  for (int k = 1; k < 4; ++k) FCHK(s, hipEventRecord(s->ev[k], s->stream));
  s->launches = 1;
  return MOM_OK;
}

int momf_rt_run(momf_scene *s) {
  FCHK(s, hipSetDevice(s->device));
  if (s->N <= 4 && s->small_n && !s->force_generic && s->nVza <= 4 && s->surf_kind == 0 && s->K <= 4) return rt_run_small_f32(s);
  if (wave_applies_f32(s)) return rt_run_wave_f32(s);
  const size_t S = s->S;
  const int N = s->N, Nz = s->Nz;
  const int m_first = s->m_first, M = s->M - m_first;  // this scene's own moments: m_first ... s->M - 1
  const bool lds = s->lds;
  const size_t sm = lds_bytes(N, lds);
  s->launches = 0;
  bool can_sweep = s->sweep && Nz <= kMaxSweepLayers && Nz > 1;
  for (int z = 2; z < Nz && can_sweep; ++z) can_sweep = (s->iface[z] == s->iface[1]);
  for (int z = 0; z < Nz && can_sweep; ++z) can_sweep = (s->nd[z] <= 127);
  FCHK(s, hipEventRecord(s->ev[0], s->stream));
  if (m_first) {  // moment 0 on the (I,Q) sub-scene (same stream: in order with what follows)
    const int rc = momf_rt_run(s->sub);
    if (rc) { s->err = s->sub->err; return rc; }
    s->launches += s->sub->launches;
    FCHK(s, hipSetDevice(s->device));
  }
  for (int z = (can_sweep ? -1 : 0); z < (can_sweep ? 0 : Nz) && M > 0; ++z) {
    LayerArgs a{};
    a.q = s->q; a.S = s->S; a.M = M; a.K = s->K; a.m_first = m_first;
    int zz = z;
    if (z < 0) {
      zz = 0;
      a.Nz_sweep = Nz;
      for (int k = 0; k < Nz; ++k) { a.nd_z[k] = (signed char)s->nd[k]; a.iface_z[k] = (signed char)s->iface[k]; }
      a.nd = s->nd[0]; a.iface = s->iface[1]; a.first = 1;
    } else {
      a.nd = s->nd[z]; a.iface = s->iface[z]; a.first = (z == 0);
    }
    a.tau = s->d_tau + S * zz; a.varpi = s->d_varpi + S * zz; a.zw = s->d_zw + (size_t)s->K * S * zz;
    a.tau_sum = s->d_tau_sum + S * zz;
    a.Zpp = s->d_Zpp + (size_t)N * N * s->K * m_first; a.Zmp = s->d_Zmp + (size_t)N * N * s->K * m_first;
    for (int k = 0; k < 6; ++k) a.comp[k] = s->comp[k];
    a.scratch = s->d_scratch; a.info = s->d_info;
    const int grid = lds ? (int)((S >= 2048) ? S : S * M) : (int)std::min<size_t>(S * M, (size_t)s->G);
#define F32_LAUNCH(IF)                                                                            \
  if (lds) {                                                                                      \
    FCHK(s, allow(k_layer<true, IF>, sm));                                                        \
    hipLaunchKernelGGL((k_layer<true, IF>), dim3(grid), dim3(kThreads), sm, s->stream, a);         \
  } else {                                                                                        \
    FCHK(s, allow(k_layer<false, IF>, sm));                                                       \
    hipLaunchKernelGGL((k_layer<false, IF>), dim3(grid), dim3(kThreads), sm, s->stream, a);        \
  }
    // strip-chained images (Float32 builds of mom_strip.hpp's chains) for the edges that have one; MOM_OPT_INVERSE != 0 keeps
    // the general path inside the same image, MOM_OPT_STRIPS_F32 = 0 (s->strips) the general image
    const int ks4 = (N % 4 == 0) ? N / 4 : 0;
    if (lds && s->strips && s->w4 && (ks4 == 9 || ks4 == 10 || ks4 == 11 || ks4 == 13 || ks4 == 14 || ks4 == 15)) {
      // 4-wave images (N = 36, 40 have no other): two workgroups per CU
      hipError_t e = hipSuccess;
      switch (ks4) {
        case 9: e = momf4_strip9_launch_layer(&a, a.iface, grid, momf4_strip9_lds_bytes(), s->stream); break;
        case 10: e = momf4_strip10_launch_layer(&a, a.iface, grid, momf4_strip10_lds_bytes(), s->stream); break;
        case 11: e = momf4_strip11_launch_layer(&a, a.iface, grid, momf4_strip11_lds_bytes(), s->stream); break;
        case 13: e = momf4_strip13_launch_layer(&a, a.iface, grid, momf4_strip13_lds_bytes(), s->stream); break;
        case 14: e = momf4_strip14_launch_layer(&a, a.iface, grid, momf4_strip14_lds_bytes(), s->stream); break;
        default: e = momf4_strip15_launch_layer(&a, a.iface, grid, momf4_strip15_lds_bytes(), s->stream); break;
      }
      FCHK(s, e);
      s->launches++;
      continue;
    }
    if (lds && s->strips && (ks4 == 11 || ks4 == 13 || ks4 == 14 || ks4 == 15)) {
      hipError_t e = hipSuccess;
      switch (ks4) {
        case 11: e = momf_strip11_launch_layer(&a, a.iface, grid, sm, s->stream); break;
        case 13: e = momf_strip13_launch_layer(&a, a.iface, grid, sm, s->stream); break;
        case 14: e = momf_strip14_launch_layer(&a, a.iface, grid, sm, s->stream); break;
        default: e = momf_strip15_launch_layer(&a, a.iface, grid, sm, s->stream); break;
      }
      FCHK(s, e);
      s->launches++;
      continue;
    }
    switch (a.iface) {
      case 0: F32_LAUNCH(0) break;
      case 1: F32_LAUNCH(1) break;
      case 2: F32_LAUNCH(2) break;
      default: F32_LAUNCH(3) break;
    }
#undef F32_LAUNCH
    FCHK(s, hipGetLastError());
    s->launches++;
  }
  FCHK(s, hipEventRecord(s->ev[1], s->stream));
  // surface: every moment of a BRDF surface, moment 0 only otherwise (m > 0: r = 0, t = I, j = 0 -- the interaction is the identity)
  for (int m = m_first; m < ((s->surf_kind == 1) ? s->M : 1); ++m) {
    SurfArgs a{};
    a.q = s->q; a.S = s->S; a.iface = s->iface[Nz - 1];
    a.albedo = s->albedo; a.tau_tot = s->d_tau_sum + S * Nz;
    a.kind = s->surf_kind; a.m = m; a.albedo_spec = s->d_albedo_spec;
    a.Rsurf = (s->surf_kind == 1) ? s->d_Rsurf + (size_t)N * N * m : nullptr;
    for (int k = 0; k < 6; ++k)
      a.comp[k] = s->comp[k] + ((k < 4) ? (size_t)comp_pitch(N) * N : (size_t)N) * S * (m - m_first);
    a.hdrJ = (m == 0) ? s->d_hdrJ : s->d_hdrJm + (size_t)N * S * m;
    a.bhr_uw = s->d_bhr_uw; a.bhr_dw = s->d_bhr_dw; a.nS_out = s->nS;
    a.scratch = s->d_scratch; a.info = s->d_info;
    const int grid = lds ? (int)S : (int)std::min<size_t>(S, (size_t)s->G);
    if (lds) {
      FCHK(s, allow(k_surface<true>, sm));
      hipLaunchKernelGGL(k_surface<true>, dim3(grid), dim3(kThreads), sm, s->stream, a);
    } else {
      FCHK(s, allow(k_surface<false>, sm));
      hipLaunchKernelGGL(k_surface<false>, dim3(grid), dim3(kThreads), sm, s->stream, a);
    }
    FCHK(s, hipGetLastError());
  }
  FCHK(s, hipEventRecord(s->ev[2], s->stream));
  {
    const size_t total = (size_t)s->nVza * s->nS * S;
    PostArgsF pa{};
    pa.N = N; pa.nS = s->nS; pa.S = s->S; pa.M = M; pa.nVza = s->nVza; pa.m_first = m_first;
    pa.hdr_all = (s->surf_kind == 1); pa.zeroT_hi = (s->surf_kind == 2);
    pa.node = s->d_node; pa.cos_mphi = s->d_cos; pa.sin_mphi = s->d_sin;
    pa.J0p = s->comp[4]; pa.J0m = s->comp[5]; pa.hdrJ0 = s->d_hdrJ; pa.hdrJm = s->d_hdrJm;
    pa.R = s->d_R; pa.T = s->d_T; pa.hdr = s->d_hdr;
    hipLaunchKernelGGL(k_postprocess_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->stream, pa);
    FCHK(s, hipGetLastError());
    if (m_first) {
      const momf_scene *u = s->sub;
      CombineArgsF ca{s->nVza, s->nS, u->nS, s->S, s->d_R, s->d_T, s->d_hdr, s->d_bhr_uw, s->d_bhr_dw,
                      u->d_R, u->d_T, u->d_hdr, u->d_bhr_uw, u->d_bhr_dw};
      const size_t tot2 = std::max(total, (size_t)s->nS * S);
      hipLaunchKernelGGL(k_combine_m0_f32, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, s->stream, ca);
      FCHK(s, hipGetLastError());
    }
  }
  FCHK(s, hipEventRecord(s->ev[3], s->stream));
  return MOM_OK;
}

static int download_f(momf_scene *s, double *dst, const float *src, size_t n) {
  std::vector<float> v(n);
  FCHK(s, hipMemcpyAsync(v.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  for (size_t i = 0; i < n; ++i) dst[i] = (double)v[i];
  return MOM_OK;
}

int momf_get_RT(momf_scene *s, double *R, double *T) {
  FCHK(s, hipSetDevice(s->device));
  const size_t nout = (size_t)s->nVza * s->nS * s->S;
  int rc = download_f(s, R, s->d_R, nout);
  if (rc) return rc;
  return download_f(s, T, s->d_T, nout);
}

int momf_get_hdr(momf_scene *s, double *hdr, double *up, double *dw) {
  FCHK(s, hipSetDevice(s->device));
  int rc = download_f(s, hdr, s->d_hdr, (size_t)s->nVza * s->nS * s->S);
  if (rc) return rc;
  if ((rc = download_f(s, up, s->d_bhr_uw, (size_t)s->nS * s->S))) return rc;
  return download_f(s, dw, s->d_bhr_dw, (size_t)s->nS * s->S);
}

int momf_timers(momf_scene *s, double *ms, int *launches) {
  FCHK(s, hipSetDevice(s->device));
  FCHK(s, hipEventSynchronize(s->ev[3]));
  float t01, t12, t23, t03;
  FCHK(s, hipEventElapsedTime(&t01, s->ev[0], s->ev[1]));
  FCHK(s, hipEventElapsedTime(&t12, s->ev[1], s->ev[2]));
  FCHK(s, hipEventElapsedTime(&t23, s->ev[2], s->ev[3]));
  FCHK(s, hipEventElapsedTime(&t03, s->ev[0], s->ev[3]));
  ms[0] = t01; ms[1] = t12; ms[2] = t23; ms[3] = t03;
  *launches = s->launches;
  return MOM_OK;
}

// batch_inv!(X, A) / A ⊠ B on a Float32 handle: Float64 host arrays at the ABI, f32 on the device; the device buffers are a
// grow-only workspace of the handle (no hipMalloc / hipFree per call, nothing to leak on an early return)
static int up_vec(momf_scene *s, float *dst, const double *src, size_t n);
static int blas_ws(momf_scene *s, int slot, size_t count, float **out) {
  if (s->blas_cap[slot] < count) {
    if (s->blas_buf[slot]) { FCHK(s, hipStreamSynchronize(s->stream)); (void)hipFree(s->blas_buf[slot]); s->blas_buf[slot] = nullptr; s->blas_cap[slot] = 0; }
    FCHK(s, dmallocf(&s->blas_buf[slot], count));
    s->blas_cap[slot] = count;
  }
  *out = s->blas_buf[slot];
  return MOM_OK;
}
int momf_blas(momf_scene *s, int n, int batch, const double *A, const double *B, double *C, bool inv) {
  FCHK(s, hipSetDevice(s->device));
  const size_t cnt = (size_t)n * n * batch;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr, *scr = nullptr;
  int rc;
  if ((rc = blas_ws(s, 0, cnt, &dA)) || (rc = blas_ws(s, 2, cnt, &dC))) return rc;
  if ((rc = up_vec(s, dA, A, cnt))) return rc;
  if (!inv) {
    if ((rc = blas_ws(s, 1, cnt, &dB)) || (rc = up_vec(s, dB, B, cnt))) return rc;
  }
  const bool lds = n <= 64 && !s->force_generic;
  const int grid = lds ? batch : std::min(batch, 1024);
  if (!lds) {
    const size_t scn = (size_t)grid * kGenericBufs * mat_elems(n) + (size_t)ld_for(n) * np_for(n);
    if ((rc = blas_ws(s, 3, scn, &scr))) return rc;
    FCHK(s, hipMemsetAsync(scr, 0, scn * sizeof(float), s->stream));
  }
  BlasArgsF a{n, batch, dA, dB, dC, scr, s->d_info};
  const size_t sm = lds_bytes(n, lds);
  if (inv) {
    if (lds) { FCHK(s, allow(k_batch_inv_f32<true>, sm)); hipLaunchKernelGGL(k_batch_inv_f32<true>, dim3(grid), dim3(kThreads), sm, s->stream, a); }
    else { FCHK(s, allow(k_batch_inv_f32<false>, sm)); hipLaunchKernelGGL(k_batch_inv_f32<false>, dim3(grid), dim3(kThreads), sm, s->stream, a); }
  } else {
    if (lds) { FCHK(s, allow(k_batched_mul_f32<true>, sm)); hipLaunchKernelGGL(k_batched_mul_f32<true>, dim3(grid), dim3(kThreads), sm, s->stream, a); }
    else { FCHK(s, allow(k_batched_mul_f32<false>, sm)); hipLaunchKernelGGL(k_batched_mul_f32<false>, dim3(grid), dim3(kThreads), sm, s->stream, a); }
  }
  FCHK(s, hipGetLastError());
  return download_f(s, C, dC, cnt);
}

// =========================================================================================
// operator-level API on a Float32 handle: Float64 host arrays at the ABI (rounded on upload, widened on download), the
// kernels of mom_ops.hpp in namespace momf.  The operators run on the caller's edge (no padding) and their own layer arrays.
// =========================================================================================
static int op_begin(momf_scene *s, DevStreams *q) {
  FCHK(s, hipSetDevice(s->device));
  const int N = s->Nu;
  if ((int)s->hd_mu.size() != N) { s->err = "operator-level call: streams not set"; return MOM_ESTATE; }
  if (!s->op_ready) {
    const size_t NN = (size_t)N * N;
    for (int k = 0; k < 6; ++k) {
      const size_t per = ((k < 4) ? NN : (size_t)N) * s->S;
      FCHK(s, dmallocf(&s->op_added[k], per));
      FCHK(s, dmallocf(&s->op_surf[k], per));
      FCHK(s, dmallocf(&s->op_comp[k], per));
      FCHK(s, hipMemsetAsync(s->op_added[k], 0, per * sizeof(float), s->stream));
      FCHK(s, hipMemsetAsync(s->op_surf[k], 0, per * sizeof(float), s->stream));
      FCHK(s, hipMemsetAsync(s->op_comp[k], 0, per * sizeof(float), s->stream));
    }
    for (int k = 0; k < 4; ++k) FCHK(s, dmallocf(&s->op_vec[k], (size_t)s->S));
    s->op_ready = true;
  }
  // the caller's streams (a scene on this handle may have padded the device copies BEHIND entry Nu - 1; rewritten here anyway)
  const std::vector<float> fm = tof(s->hd_mu.data(), N), fw = tof(s->hd_wt.data(), N), fs = tof(s->hd_sg.data(), N);
  FCHK(s, hipMemcpyAsync(s->d_mu, fm.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipMemcpyAsync(s->d_wt, fw.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipMemcpyAsync(s->d_sg, fs.data(), N * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  *q = s->q;
  q->mu = s->d_mu; q->wt = s->d_wt; q->sg = s->d_sg; q->N = N;
  return MOM_OK;
}
static int up_vec(momf_scene *s, float *dst, const double *src, size_t n) {
  const std::vector<float> v = tof(src, n);
  FCHK(s, hipMemcpyAsync(dst, v.data(), n * sizeof(float), hipMemcpyHostToDevice, s->stream));
  FCHK(s, hipStreamSynchronize(s->stream));
  return MOM_OK;
}
#define OP_LAUNCH(s, KERN, grid, args)                                                            \
  do {                                                                                            \
    const bool l__ = (s)->Nu <= 64 && !(s)->force_generic;                                        \
    const size_t sm__ = lds_bytes((s)->Nu, l__);                                                  \
    if (l__) {                                                                                    \
      FCHK(s, allow(KERN<true>, sm__));                                                           \
      hipLaunchKernelGGL(KERN<true>, dim3(grid), dim3(kThreads), sm__, (s)->stream, args);        \
    } else {                                                                                      \
      FCHK(s, allow(KERN<false>, sm__));                                                          \
      hipLaunchKernelGGL(KERN<false>, dim3(grid), dim3(kThreads), sm__, (s)->stream, args);       \
    }                                                                                             \
    FCHK(s, hipGetLastError());                                                                   \
  } while (0)
static int op_grid(const momf_scene *s) {
  return (s->Nu <= 64 && !s->force_generic) ? s->S : (int)std::min<size_t>((size_t)s->S, (size_t)s->G);
}

int momf_op_elemental(momf_scene *s, int m, int nd, const double *tau_sum, const double *dtau, const double *varpi,
                      const double *Zpp, const double *Zmp, int z_batch) {
  OpArgs a{};
  int rc;
  if ((rc = op_begin(s, &a.q))) return rc;
  const size_t zc = (size_t)s->Nu * s->Nu * z_batch;
  if (zc > s->op_Zcap) {
    if (s->op_Z[0]) { (void)hipFree(s->op_Z[0]); (void)hipFree(s->op_Z[1]); s->op_Z[0] = s->op_Z[1] = nullptr; }
    FCHK(s, dmallocf(&s->op_Z[0], zc));
    FCHK(s, dmallocf(&s->op_Z[1], zc));
    s->op_Zcap = zc;
  }
  if ((rc = up_vec(s, s->op_vec[0], tau_sum, s->S)) || (rc = up_vec(s, s->op_vec[1], dtau, s->S)) ||
      (rc = up_vec(s, s->op_vec[2], varpi, s->S)) || (rc = up_vec(s, s->op_Z[0], Zpp, zc)) || (rc = up_vec(s, s->op_Z[1], Zmp, zc)))
    return rc;
  a.S = s->S; a.m = m; a.nd = nd; a.z_batch = z_batch;
  a.tau_sum = s->op_vec[0]; a.dtau = s->op_vec[1]; a.varpi = s->op_vec[2]; a.Zpp = s->op_Z[0]; a.Zmp = s->op_Z[1];
  for (int k = 0; k < 6; ++k) a.added[k] = s->op_added[k];
  a.scratch = s->d_scratch; a.info = s->d_info;
  OP_LAUNCH(s, k_op_elemental, op_grid(s), a);
  FCHK(s, hipStreamSynchronize(s->stream));
  return MOM_OK;
}

int momf_op_doubling(momf_scene *s, int nd, double *expk) {
  OpArgs a{};
  int rc;
  if ((rc = op_begin(s, &a.q))) return rc;
  if ((rc = up_vec(s, s->op_vec[3], expk, s->S))) return rc;
  a.S = s->S; a.nd = nd; a.expk = s->op_vec[3];
  for (int k = 0; k < 6; ++k) a.added[k] = s->op_added[k];
  a.scratch = s->d_scratch; a.info = s->d_info;
  if (nd > 0) OP_LAUNCH(s, k_op_doubling, op_grid(s), a);  // doubling.jl:28 returns early for 0
  return download_f(s, expk, s->op_vec[3], s->S);
}

int momf_op_interaction(momf_scene *s, int iface, int with_surface_layer) {
  OpArgs a{};
  int rc;
  if ((rc = op_begin(s, &a.q))) return rc;
  if (!s->op_comp_set) {
    s->err = "mom_interaction: start the operator-level sequence with mom_copy_added_to_composite or mom_upload";
    return MOM_ESTATE;
  }
  a.S = s->S; a.iface = iface;
  for (int k = 0; k < 6; ++k) { a.added[k] = with_surface_layer ? s->op_surf[k] : s->op_added[k]; a.comp[k] = s->op_comp[k]; }
  a.scratch = s->d_scratch; a.info = s->d_info;
  OP_LAUNCH(s, k_op_interaction, op_grid(s), a);
  FCHK(s, hipStreamSynchronize(s->stream));
  return MOM_OK;
}

int momf_op_copy_added_to_composite(momf_scene *s) {
  DevStreams q;
  int rc;
  if ((rc = op_begin(s, &q))) return rc;
  const size_t NN = (size_t)s->Nu * s->Nu;
  const int src[6] = {1, 0, 3, 2, 4, 5};  // composite order R_mp, R_pm, T_pp, T_mm, J0p, J0m ; added order r_pm, r_mp, t_mm, t_pp, j0p, j0m
  for (int k = 0; k < 6; ++k) {
    const size_t bytes = ((k < 4) ? NN : (size_t)s->Nu) * s->S * sizeof(float);
    FCHK(s, hipMemcpyAsync(s->op_comp[k], s->op_added[src[k]], bytes, hipMemcpyDeviceToDevice, s->stream));
  }
  FCHK(s, hipStreamSynchronize(s->stream));
  s->op_comp_set = true;
  return MOM_OK;
}

int momf_op_surface_lambertian(momf_scene *s, int m, double albedo, const double *tau_tot) {
  DevStreams q;
  int rc;
  if ((rc = op_begin(s, &q))) return rc;
  if ((rc = up_vec(s, s->op_vec[0], tau_tot, s->S))) return rc;
  hipLaunchKernelGGL(k_op_surface_fill, dim3(s->S), dim3(256), 0, s->stream, q, s->S, m, (float)albedo, s->op_vec[0], s->op_surf[0],
                     s->op_surf[1], s->op_surf[2], s->op_surf[3], s->op_surf[4], s->op_surf[5]);
  FCHK(s, hipGetLastError());
  FCHK(s, hipStreamSynchronize(s->stream));
  return MOM_OK;
}

static float *op_which(momf_scene *s, int which, size_t *count) {
  const int grp = which / 6, k = which % 6;
  *count = ((k < 4) ? (size_t)s->Nu * s->Nu : (size_t)s->Nu) * s->S;
  return grp == 0 ? s->op_added[k] : (grp == 1 ? s->op_comp[k] : s->op_surf[k]);
}
int momf_op_upload(momf_scene *s, int which, const double *src) {
  DevStreams q;
  int rc;
  if ((rc = op_begin(s, &q))) return rc;
  size_t count = 0;
  float *p = op_which(s, which, &count);
  if (which / 6 == 1) s->op_comp_set = true;
  return up_vec(s, p, src, count);
}
int momf_op_download(momf_scene *s, int which, double *dst) {
  DevStreams q;
  int rc;
  if ((rc = op_begin(s, &q))) return rc;
  if (which / 6 == 1 && !s->op_comp_set) {
    s->err = "mom_download: a Float32 handle keeps the operator-level composite layer apart from the scene-level state; start "
             "the operator-level sequence with mom_copy_added_to_composite or mom_upload";
    return MOM_ESTATE;
  }
  size_t count = 0;
  float *p = op_which(s, which, &count);
  return download_f(s, dst, p, count);
}
