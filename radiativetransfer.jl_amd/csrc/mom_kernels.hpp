// mom_kernels.hpp -- the per-layer Matrix-Operator kernels (elemental -> doubling ->
// interaction), written against the workgroup primitives of mom_device.hpp.
//
// Reference semantics restated here (file:line relative to the reference root, src/CoreRT):
//   elemental!        CoreKernel/elemental.jl:109-162, kernels :164-285
//   doubling_helper!  CoreKernel/doubling.jl:13-79, apply_D! :93-118
//   interaction_helper! CoreKernel/interaction.jl:8-22 (00) :27-43 (01) :49-64 (10) :69-117 (11)
//   create_surface_layer!(::LambertianSurfaceScalar) Surfaces/lambertian_surface.jl:20-75
//   postprocessing_vza! tools/postprocessing_vza.jl:9-60
//
// Sign bookkeeping: with sg[i] = -1 where the (reference-indexed) Stokes component of row i
// is "> 2" and +1 elsewhere, the reference's D kernels amount to
//   r-+ rows scaled by sg (elemental when nd>=1, and again after doubling), j0- scaled by sg
//   after doubling, and  r+- = diag(sg) r-+ diag(sg),  t-- = diag(sg) t++ diag(sg)  always.
// So r+- and t-- are never materialised in the fused path: they are operand functors.
//
// T (I - B)^-1 (the reference forms inv(I - B) by LU, gpu_batched.jl:36-87, then multiplies):
//   with beta = ||B||_F < 1/2 and p the smallest integer with beta^p / (1 - beta) <= 2^-56, the
//   truncated Neumann series sum_{k<p} B^k differs from (I - B)^-1 by at most 2^-56 in every
//   element (||B^k||_F <= beta^k) -- below FP64 rounding of the O(1) result -- and needs only
//   MFMA products: Horner (p <= 4) or repeated squaring (p <= 32).  Otherwise: Gauss-Jordan
//   with partial pivoting.  MOM_OPT_INVERSE = 1 forces the pivoted path.
#pragma once
#include "mom_device.hpp"

// diagnostic builds (scratch/phase_stamp.hip) define MOM_STAMP(id) to accumulate s_memtime deltas;
// in the product build it expands to nothing.
#ifndef MOM_STAMP
#define MOM_STAMP(id)
#endif
#ifndef MOM_STAMP4
#define MOM_STAMP4(id)
#endif

namespace MOM_NS {

// beta^2 thresholds: kNeumannThr2[p-1] = largest ||B||_F^2 for which p series terms suffice
// (beta^p / (1 - beta) <= 2^-56), p = 1..32 -- except that the first-order term B is always kept (p >= 2 unless
// B = 0): the bound is ABSOLUTE, i.e. against the O(1) diagonal, while the off-diagonal (diffuse) elements of t++ are
// themselves only of order rho = sqrt(beta) in a doubling step; dropping B there would cost them a relative rho (up to
// 4e-9), whereas the tail after B is of relative size rho^3 <= 2e-13.
#ifdef MOM_REAL_IS_FLOAT
// Float32 build: beta^p / (1 - beta) <= 2^-26 (a quarter of the f32 epsilon), first-order term always kept
__device__ const real kNeumannThr2[32] = {
    0.000000000e+00f, 1.489934232e-08f, 6.045524421e-06f, 1.213959655e-04f,
    7.320204349e-04f, 2.419753539e-03f, 5.676200029e-03f, 1.075029447e-02f,
    1.765854019e-02f, 2.625958398e-02f, 3.632882835e-02f, 4.761229038e-02f,
    5.985942527e-02f, 7.284045577e-02f, 8.635379794e-02f, 1.002277241e-01f,
    1.143189633e-01f, 1.285098733e-01f, 1.427051184e-01f, 1.568283537e-01f,
    1.708191621e-01f, 1.846303449e-01f, 1.982255889e-01f, 2.115774909e-01f,
    2.246659062e-01f, 2.374765770e-01f, 2.500000000e-01f, 2.622304965e-01f,
    2.741654486e-01f, 2.858046753e-01f, 2.971499218e-01f, 3.082044441e-01f};
#else
__device__ const real kNeumannThr2[32] = {
    0.0 /* p = 1 only for B = 0 */, 1.38777877561156685e-17, 5.77492213356056750e-12, 3.72517661162420568e-09,
    1.80656771560518035e-07, 2.40186660760962690e-06, 1.52417448931310540e-05, 6.09157135028591602e-05,
    1.78874927371965362e-04, 4.23309807394842467e-04, 8.56292603484697687e-04, 1.53988783074545245e-03,
    2.52966413875025916e-03, 3.87056942792606993e-03, 5.59509450064154569e-03, 7.72318484591632843e-03,
    1.02632763666295982e-02, 1.32139270473634555e-02, 1.65656642989196294e-02, 2.03028052759899880e-02,
    2.44051136922726897e-02, 2.88492300492497432e-02, 3.36098586497813809e-02, 3.86607216385354419e-02,
    4.39753040322414940e-02, 4.95274191527264318e-02, 5.52916244610413068e-02, 6.12435157546498479e-02,
    6.73599244364155580e-02, 7.36190389296255826e-02, 8.00004677634075928e-02, 8.64852586225294262e-02};
#endif

struct DevStreams {
  const real *mu;  // [N] qp_μN
  const real *wt;  // [N] wt_μN
  const real *sg;  // [N] +-1 (see above)
  real I0[4];
  real D[4];
  int N, nS, imu0;  // imu0: 1-based stream index of the sun
  int inv_mode;     // 0 auto, 1 force pivoted Gauss-Jordan
  int regular;      // qp_μN repeats each stream value nS times (always true for the reference's QuadPoints)
  real mu0;
};

// workgroup context: where this workgroup's matrices and vectors live
struct Ctx {
  int N, Np, nc, ld, ldv;  // nc: stored columns per buffer (cols_for)
  FastDiv fd;
  real *r, *t, *P, *Q, *X;  // padded N x N buffers (X: spare, generic mode only)
  real *jp, *jm, *j1p, *j1m, *v1, *v2, *Jp, *Jm, *prow, *pcol, *rowk, *ei, *mu, *wt, *sg, *part, *thr;
  int *ipiv, *sh, *bad;
  int inv_mode;
  int qpre;  // MOM_QPREFETCH experiment: c.Q already holds the composite T++ (+ J0+ riding) for the coming interaction
  int slot;  // MOM_SIMD_AWARE experiment: the column strip this wave owns (mom_strip.hpp strip_slot)
  const real *ptab;  // strip images (r5): F1 | F2 | SI per stream pair, built once per workgroup (nullptr: per layer, in Q)
  real *tabE, *tabZS;  // table space of elemental_build: E | F1 | F2 per stream pair and the sun-block Z columns (nullptr: the current Q / P)
};

__host__ __device__ inline int np_for(int N) { return 16 * ((N + 15) / 16); }
// Row pitch of the padded operator buffers: 16 NT + 2 (ds_read_b64 of 16 columns x 2 k conflict-free: pitch / 2 odd).  One
// exception (r4): the Float64 4-wave image of N = 44 takes N + 2 = 46 (23 odd as well) -- 76.9 instead of 82.9 KB of LDS, so that
// TWO workgroups share a CU like the N = 36, 40 images do (no k-step of a Float64 product reads a row >= N when N % 4 = 0)
// r5: N = 36, 40 likewise (pitch 38 / 42: 19, 21 odd) -- room for the persistent stream-pair tables next to two images per CU
__host__ __device__ inline int ld_for(int N) { return (kF64 && kWaves == 4 && (N == 36 || N == 40 || N == 44)) ? N + 2 : np_for(N) + 2; }
// columns actually stored per buffer: the K padding (up to the next multiple of 4) and the two riding
// columns N, N+1; MFMA B-operand reads of the remaining columns of the last tile (< Np) run past the buffer
// into whatever follows (finite or not, they only feed output columns that are never stored).
__host__ __device__ inline int cols_for(int N) {
#ifdef MOM_REAL_IS_FLOAT
  // Float32 build: every column of the padded buffers is stored (and zeroed): the Float32 strip chains run all 4 NT k-steps of
  // a row tile, so the strip rows >= N + 2 (multiplier columns >= N + 2) must be exact zeros, not whatever follows the buffer
  return np_for(N);
#else
  const int need = ((N + 3) / 4) * 4 > N + 2 ? ((N + 3) / 4) * 4 : N + 2;
  return need < np_for(N) ? need : np_for(N);
#endif
}
__host__ __device__ inline size_t mat_elems(int N) { return (size_t)ld_for(N) * cols_for(N); }
constexpr int kNumVec = 15 + 2 * kWaves;  // vectors carved from LDS (part = 2*kWaves vectors)
constexpr int kGenericBufs = 5;
// generic mode, N > 64: LDS staging tiles of wg_gemm_big behind the vectors: A panels [kBigStages][kBigKB][kBigLdA] and
// B panels [kBigStages][256][kBigLdB] reals; the same area holds the wave-private transposition patches of its epilogue
constexpr int kBigKB = 8;      // k extent of a panel (two MFMA k-steps between two barriers)
constexpr int kBigStages = 3;  // panel p is multiplied while p + 1 is complete in LDS (its first fragments are read before
                               // the barrier) and p + 2 is being written
constexpr int kBigRows = 128;  // rows of the output block of one pass (2 x 4 waves of 64 x 64)
constexpr int kBigLdA = 144;   // row pitch of an A panel: 128 + 16 (ds_read_b64 of 16 rows x 2 k conflict-free)
constexpr int kBigLdB = kBigKB + 2;  // k pitch of a B-panel column (ds_read_b64 of 2 k x 16 columns conflict-free)
constexpr int kBigCols = 256;  // largest operator the tiles are sized for
constexpr int kBigStA = kBigKB * kBigLdA;  // reals per A stage; a B stage holds np_for(N) columns of pitch kBigLdB
constexpr int kBigPatch = 36;  // row pitch of the epilogue patch (16 columns x 32 rows per wave)
__host__ __device__ inline int big_stage_b(int N) { return 16 * ((N + 15) / 16) * kBigLdB; }
// r4: k panels of 16 in TWO stages (gemm_big_pass16: half the barriers per product) wherever the LDS holds them -- every
// N <= 256 once the staging tiles are laid over `part` (the 2 kWaves scratch vectors of the mat-vecs, dead during a product; the
// first 16 reals stay: wg_sumsq_put writes part[wave] from inside an epilogue)
constexpr int kBig16KB = 16;
constexpr int kBig16LdB = kBig16KB + 2;
constexpr int kBig16MaxN = 256;
__host__ __device__ inline int big16_stage_doubles(int N) { return kBig16KB * kBigLdA + 16 * ((N + 15) / 16) * kBig16LdB; }
__host__ __device__ inline int big_tile_doubles(int N) {
  int st = kBigStages * (kBigStA + big_stage_b(N));
  const int pt = 8 * 16 * kBigPatch;
  if (N <= kBig16MaxN) st = 2 * big16_stage_doubles(N);  // (the K = 8 pass is then not used: MOM_NO_BIG16 builds keep its size)
#ifdef MOM_NO_BIG16
  st = kBigStages * (kBigStA + big_stage_b(N));
#endif
  return st > pt ? st : pt;
}
// Vector area: 15 vectors, the 32 series thresholds, np + 4 ints (ipiv, sh, bad) in whole pairs of reals (the Float32
// build needs one real per int), then `part` (2 kWaves vectors of scratch).  `part` comes last so that the
// register-resident doubling (mom_regdbl.hpp) can lay its two operand slots over it and the panel area behind.
__host__ __device__ inline size_t vec_ints_doubles(int N) {
  return 2 * ((sizeof(int) * (size_t)(np_for(N) + 4) + 2 * sizeof(real) - 1) / (2 * sizeof(real)));
}
__host__ __device__ inline size_t part_offset_doubles(int N) { return (size_t)(15 * np_for(N) + 32) + vec_ints_doubles(N); }
__host__ __device__ inline size_t vec_area_doubles(int N) { return part_offset_doubles(N) + (size_t)(2 * kWaves) * np_for(N); }
// start of the staging tiles of the panel GEMM (generic mode): over `part`, behind its first 16 reals
__host__ __device__ inline size_t big_tile_base_doubles(int N) { return part_offset_doubles(N) + 16; }
// LDS tail of k_layer: the staged per-(point, layer) scalars (mom_entry.hpp), kLayTab reals behind everything else
// (the register-resident doubling at N = 96 leaves 1 KB of the CU's 160 KB: the tail shrinks to what is left there -- 126
// doubles, still 25 layers of a two-basis scene per batch; the host refuses more than 64 bases, 67 reals per layer)
constexpr int kLayTab = 256;
constexpr size_t kLdsPerCU = 160 * 1024;
__host__ __device__ inline size_t lds_body_bytes(int N, bool lds_mats);
__host__ __device__ inline size_t lay_offset_reals(int N, bool lds_mats) { return (lds_body_bytes(N, lds_mats) + sizeof(real) - 1) / sizeof(real); }
__host__ __device__ inline int lay_cap_reals(int N, bool lds_mats) {
  const size_t used = lay_offset_reals(N, lds_mats) * sizeof(real);
  const size_t left = used < kLdsPerCU ? (kLdsPerCU - used) / sizeof(real) : 0;
  return left < (size_t)kLayTab ? (int)left : kLayTab;
}
__host__ __device__ inline size_t lds_bytes(int N, bool lds_mats) { return (lay_offset_reals(N, lds_mats) + lay_cap_reals(N, lds_mats)) * sizeof(real); }
// Strip images, Float64 (r5): the layer-independent tables of the elemental layer -- per PAIR OF STREAMS F1 = mu_j / (mu_i + mu_j),
// F2 = mu_j / (mu_i - mu_j), SI = 1 / mu_i + 1 / mu_j -- are built ONCE per (persistent) workgroup behind the tail above instead
// of once per layer in Q: a layer's table pass is then one exponential per pair, E = 1 - exp(-dtau SI), no division (the same
// expressions on the same operands as before: identical values).  ns: Stokes components per stream (1 if the streams are not
// regular); 0 if the per-layer tables do not apply either (3 Nq^2 reals must fit a matrix buffer, elemental_build).
__host__ __device__ inline int ptab_reals(int N, int ns) {
  const int Nq = N / (ns > 0 ? ns : 1), nt = 3 * Nq * Nq;
  if (nt > (int)mat_elems(N)) return 0;
  // 4-wave images live on two workgroups per CU: no tables where they would cost the second one (N = 44)
  if (kWaves == 4 && 2 * (lds_bytes(N, true) + (size_t)nt * sizeof(real)) + 2048 > kLdsPerCU) return 0;
  // any image: the tables must fit the CU next to the image itself (8-wave N = 60 with two Stokes components per stream --
  // the m = 0 sub-problem of a 30-stream IQU / IQUV scene -- would need 171 KB); the kernel then keeps the per-layer tables in Q
  if (lds_bytes(N, true) + (size_t)nt * sizeof(real) > kLdsPerCU) return 0;
  return nt;
}
__host__ __device__ inline size_t strip_lds_bytes(int N, int ns) { return lds_bytes(N, true) + (kF64 ? (size_t)ptab_reals(N, ns) * sizeof(real) : 0); }
__host__ __device__ inline size_t lds_body_bytes(int N, bool lds_mats) {
  size_t b = vec_area_doubles(N) * sizeof(real);
  if (lds_mats) b += 4 * mat_elems(N) * sizeof(real);
  else if (N > 64) {
    const size_t bt = (big_tile_base_doubles(N) + (size_t)big_tile_doubles(N)) * sizeof(real);
    if (bt > b) b = bt;
    if (sizeof(real) == 8 && kWaves == 8 && N <= 96) {  // register-resident doubling: 16 reals of part + two slots (rg_applies)
      const size_t rg = (part_offset_doubles(N) + 16 + 2 * (size_t)np_for(N) * ld_for(N)) * sizeof(real);
      if (rg > b) b = rg;
    }
  }
  return b;
}

template <bool LDSM>
__device__ __forceinline__ void make_ctx(Ctx &c, int N, int inv_mode, real *smem, real *gscratch) {
  c.N = N;
  c.Np = np_for(N);
  c.nc = cols_for(N);
  c.ld = ld_for(N);
  c.ldv = c.Np;
  c.fd.init(N);
  c.inv_mode = inv_mode;
  c.qpre = 0;
  c.slot = 0;
  c.ptab = nullptr;
  const size_t msz = mat_elems(N);
  real *p = smem;
  if (LDSM) {
    c.r = p; c.t = p + msz; c.P = p + 2 * msz; c.Q = p + 3 * msz; c.X = nullptr;
    p += 4 * msz;
  } else {
    c.r = gscratch; c.t = gscratch + msz; c.P = gscratch + 2 * msz; c.Q = gscratch + 3 * msz; c.X = gscratch + 4 * msz;
  }
  const int lv = c.ldv;
  c.jp = p; c.jm = p + lv; c.j1p = p + 2 * lv; c.j1m = p + 3 * lv; c.v1 = p + 4 * lv; c.v2 = p + 5 * lv;
  c.Jp = p + 6 * lv; c.Jm = p + 7 * lv; c.prow = p + 8 * lv; c.pcol = p + 9 * lv; c.rowk = p + 10 * lv;
  c.ei = p + 11 * lv; c.mu = p + 12 * lv; c.wt = p + 13 * lv; c.sg = p + 14 * lv;
  c.thr = p + 15 * lv;
  int *ip = reinterpret_cast<int *>(c.thr + 32);
  c.ipiv = ip; c.sh = ip + lv; c.bad = ip + lv + 1;
  c.part = p + part_offset_doubles(N);
  c.tabE = nullptr;   // elemental_build: the CURRENT Q / P (the generic mode rotates its buffer pointers between layers)
  c.tabZS = nullptr;
}

// zero the padding (rows/cols >= N) of the LDS matrix buffers; vectors fully
template <bool LDSM>
__device__ __forceinline__ void zero_padding(const Ctx &c) {
  const int N = c.N, Np = c.nc, ld = c.ld;
  // the padding is only ever read as a K index when N is not a multiple of the MFMA K step (4);
  // otherwise padded rows/columns only feed output rows/columns that are never stored.  Float32: the strip chains reach
  // every row of the last row tile (mom_strip.hpp, StripGeom::NKS), so the padding is always zeroed
  if (LDSM && (N % 4 != 0 || !kF64)) {
    const int padr = ld - N;
    for (int e = wg_tid(); e < padr * Np; e += kThreads) {
      const int j = e / padr, i = N + (e - j * padr);
      const int o = i + j * ld;
      c.r[o] = 0.0; c.t[o] = 0.0; c.P[o] = 0.0; c.Q[o] = 0.0;
    }
    const int padc = Np - N;
    for (int e = wg_tid(); e < padc * N; e += kThreads) {
      const int jj = e / N, i = e - jj * N;
      const int o = i + (N + jj) * ld;
      c.r[o] = 0.0; c.t[o] = 0.0; c.P[o] = 0.0; c.Q[o] = 0.0;
    }
  }
  if (wg_tid() < 32) c.thr[wg_tid()] = kNeumannThr2[wg_tid()];
}

// restore the zero padding of one buffer after it was used as scratch (only matters if N % 4 != 0)
__device__ __forceinline__ void rezero_padding(const Ctx &c, real *buf) {
  const int N = c.N, Np = c.nc, ld = c.ld;
  if (N % 4 == 0 && kF64) return;
  const int padr = ld - N;
  for (int e = wg_tid(); e < padr * Np; e += kThreads) {
    const int j = e / padr, i = N + (e - j * padr);
    buf[i + j * ld] = 0.0;
  }
  const int padc = Np - N;
  for (int e = wg_tid(); e < padc * N; e += kThreads) {
    const int jj = e / N, i = e - jj * N;
    buf[i + (N + jj) * ld] = 0.0;
  }
}

// element functor of a PADDED buffer (no bounds checks)
struct ElP {
  const real *p; int ld;
  __device__ __forceinline__ real operator()(int i, int j) const { return p[i + j * ld]; }
  template <int LD> __device__ __forceinline__ real at(int i, int j) const { return p[i + j * LD]; }
  __device__ __forceinline__ void launder() { asm volatile("" : "+v"(p)); }
};
// diag(sg) * padded buf * diag(sg)   (sg padded with anything finite)
struct ElSigP {
  const real *p; const real *sg; int ld;
  __device__ __forceinline__ real operator()(int i, int j) const { return sg[i] * sg[j] * p[i + j * ld]; }
  template <int LD> __device__ __forceinline__ real at(int i, int j) const { return sg[i] * sg[j] * p[i + j * LD]; }
  __device__ __forceinline__ void launder() { asm volatile("" : "+v"(p), "+v"(sg)); }
};
template <> struct lds_operand<ElP> { static constexpr bool value = true; };
template <> struct lds_operand<ElSigP> { static constexpr bool value = true; };
// element functor of an unpadded (global) array with bounds checks
struct El {
  const gdouble *p; int ld, N;
  __device__ __forceinline__ real operator()(int i, int j) const { return (i < N && j < N) ? p[i + j * ld] : 0.0; }
  // as a phase-matrix source of elemental_build: one term of weight 1
  __device__ __forceinline__ int terms() const { return 1; }
  __device__ __forceinline__ real weight(int) const { return 1.0; }
  __device__ __forceinline__ real basis(int, int i, int j) const { return p[i + j * ld]; }
};
struct ElZero { __device__ __forceinline__ real operator()(int, int) const { return 0.0; } };
struct ElEye {
  int N;
  __device__ __forceinline__ real operator()(int i, int j) const { return (i == j && i < N) ? 1.0 : 0.0; }
};

// y = M x ; all threads; ends with barrier
template <class FM>
__device__ __forceinline__ void wg_matvec(const Ctx &c, FM M, const real *x, real *y) {
  const int N = c.N, lane = wg_lane(), wave = wg_wave();
  const int chunk = (N + kWaves - 1) / kWaves;
  const int k0 = wave * chunk, k1 = min(N, k0 + chunk);
  for (int i = lane; i < N; i += 64) {
    real s = 0.0;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) s += M(i, k) * x[k];
    c.part[wave * c.ldv + i] = s;
  }
  __syncthreads();
  for (int i = wg_tid(); i < N; i += kThreads) {
    real s = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += c.part[w * c.ldv + i];
    y[i] = s;
  }
  __syncthreads();
}

// Generic mode: dst = I + src (NEG: I - src) over a padded slab buffer, flat 16-byte pairs with eight in flight per lane
// (the padding rows are copied along; dst may be src).  Needs a barrier after.
template <bool NEG>
__device__ __forceinline__ void slab_eye_plus(const Ctx &c, real *dst, const real *src) {
  typedef real r2 __attribute__((ext_vector_type(2)));
  const int N = c.N, ld = c.ld, tot2 = (ld * N) >> 1;  // ld is even
  for (int e0 = wg_tid(); e0 < tot2; e0 += 8 * kThreads) {
    r2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kThreads;
      if (e < tot2) v[u] = *(const r2 *)(src + 2 * e);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kThreads;
      if (e < tot2) {
        r2 w = NEG ? -v[u] : v[u];
        // element 2 e is (i, j) with i + j ld = 2 e: the diagonal i = j sits at j (ld + 1)
        const int j = (2 * e) / ld, i = 2 * e - j * ld;
        if (i == j) w.x += 1.0;
        if (i + 1 == j) w.y += 1.0;
        *(r2 *)(dst + 2 * e) = w;
      }
    }
  }
}

// Generic mode: composite block (global, pitch ld_s) -> slab buffer (pitch ld_d) as 16-byte row pairs, eight columns in
// flight per lane (wg_copy_mat moves 8-byte elements four at a time and divides per element: 3.6 % of the C4 run for the
// five copies of an interaction).  N odd: element copy.
template <bool LDSM, class PS>
__device__ __forceinline__ void copy_to_slab(const Ctx &c, PS src, int ld_s, real *dst, int ld_d) {
  const int N = c.N;
  if (LDSM || (N & 1) || (ld_s & 1) || (ld_d & 1)) {
    wg_copy_mat(N, c.fd, src, ld_s, dst, ld_d);
    return;
  }
  typedef real r2 __attribute__((ext_vector_type(2)));
  const real *sp = (const real *)src;
  const int lane = wg_lane(), wave = wg_wave(), hp = N >> 1;
  for (int j0 = wave; j0 < N; j0 += 8 * kWaves)
    for (int ip = lane; ip < hp; ip += 64) {
      r2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + kWaves * u;
        if (j < N) v[u] = *(const r2 *)(sp + 2 * ip + (size_t)j * ld_s);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + kWaves * u;
        if (j < N) *(r2 *)(dst + 2 * ip + (size_t)j * ld_d) = v[u];
      }
    }
}

// Generic mode: M is a padded buffer of the GLOBAL slab (N > 64: 512 KB at N = 256).  The element loop above keeps four
// 8-byte loads in flight per lane and is latency-bound there (two passes over r and Q per doubling step: 7.7 % of the C4
// run); here a lane owns two row pairs and walks its wave's column chunk eight columns at a time: sixteen 16-byte loads
// in flight.  NV = 1: y1 = M x1; NV = 2: also y2 = M x2 in the same pass.  y may alias x.  Ends with a barrier.
template <int NV>
__device__ __forceinline__ void wg_matvec_slab(const Ctx &c, const real *M, const real *x1, const real *x2, real *y1, real *y2) {
  typedef real r2 __attribute__((ext_vector_type(2)));
  const int N = c.N, ld = c.ld, lane = wg_lane(), wave = wg_wave();
  const int chunk = (N + kWaves - 1) / kWaves;
  const int k0 = wave * chunk, k1 = min(N, k0 + chunk);
  for (int ib = 0; ib < N; ib += 256) {
    int i0 = ib + 2 * lane, i1 = i0 + 128;
    const int c0 = i0 < c.Np - 2 ? i0 : c.Np - 2, c1 = i1 < c.Np - 2 ? i1 : c.Np - 2;  // rows < ld = Np + 2 exist
    r2 s10 = {0.0, 0.0}, s11 = {0.0, 0.0}, s20 = {0.0, 0.0}, s21 = {0.0, 0.0};
    for (int k = k0; k < k1; k += 8) {
      r2 m0[8], m1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = k + u < k1 ? k + u : k1 - 1;
        m0[u] = *(const r2 *)(M + c0 + (size_t)kk * ld);
        m1[u] = *(const r2 *)(M + c1 + (size_t)kk * ld);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = k + u < k1;
        const real a = in ? x1[k + u] : 0.0;
        s10 += m0[u] * a;
        s11 += m1[u] * a;
        if (NV == 2) {
          const real b = in ? x2[k + u] : 0.0;
          s20 += m0[u] * b;
          s21 += m1[u] * b;
        }
      }
    }
    real *p1 = c.part + wave * c.ldv, *p2 = c.part + (kWaves + wave) * c.ldv;
    if (i0 < N) { p1[i0] = s10.x; if (NV == 2) p2[i0] = s20.x; }
    if (i0 + 1 < N) { p1[i0 + 1] = s10.y; if (NV == 2) p2[i0 + 1] = s20.y; }
    if (i1 < N) { p1[i1] = s11.x; if (NV == 2) p2[i1] = s21.x; }
    if (i1 + 1 < N) { p1[i1 + 1] = s11.y; if (NV == 2) p2[i1 + 1] = s21.y; }
  }
  __syncthreads();
  for (int i = wg_tid(); i < N; i += kThreads) {
    real s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      s1 += c.part[w * c.ldv + i];
      if (NV == 2) s2 += c.part[(kWaves + w) * c.ldv + i];
    }
    y1[i] = s1;
    if (NV == 2) y2[i] = s2;
  }
  __syncthreads();
}

// dst(i,j) <- f(i, j, sum_k A(i,k) B(k,j), dst_old(i,j)); dst may be an operand of A/B.
// LDS mode: true in place (SYNC); generic mode: written to the spare buffer, then swapped.
template <bool LDSM, class FA, class FB, class FV>
__device__ __forceinline__ void gemm_to(Ctx &c, real *&dst, FA A, FB B, FV f) {
  const int N = c.N, ld = c.ld;
  real *d = dst;
  if (LDSM) {
    wg_gemm<true>(N, A, B, [=](int i, int j, real v) { d[i + j * ld] = f(i, j, v, d[i + j * ld]); });
  } else {
    real *s = c.X;
    wg_gemm<false, !LDSM>(N, A, B, [=](int i, int j, real v) { s[i + j * ld] = f(i, j, v, d[i + j * ld]); });
    c.X = d;
    dst = s;
  }
  __syncthreads();
}

// smallest p in 1..32 with beta2 <= thr[p-1] (thr: the table above, copied to LDS by the prologue),
// 1000 if none (or NaN): binary search, 5 dependent LDS reads.
__device__ __forceinline__ int neumann_terms(const real *thr, real beta2) {
  if (!(beta2 <= thr[31])) return 1000;
  int lo = 0, hi = 31;  // invariant: beta2 <= thr[hi]
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int mid = (lo + hi) >> 1;
    if (beta2 <= thr[mid]) hi = mid; else lo = mid + 1;
  }
  return hi + 1;
}

// the same for p <= 12 without memory: thresholds as immediates (the strip chains' range); 1000 beyond or NaN
__device__ __forceinline__ int neumann_terms_12(real beta2) {
#ifdef MOM_REAL_IS_FLOAT
  if (!(beta2 <= 4.761229038e-02f)) return 1000;
  int q = 1;
  q += beta2 > 0.0f; q += beta2 > 1.489934232e-08f; q += beta2 > 6.045524421e-06f; q += beta2 > 1.213959655e-04f;
  q += beta2 > 7.320204349e-04f; q += beta2 > 2.419753539e-03f; q += beta2 > 5.676200029e-03f; q += beta2 > 1.075029447e-02f;
  q += beta2 > 1.765854019e-02f; q += beta2 > 2.625958398e-02f; q += beta2 > 3.632882835e-02f;
  return q;
#endif
  if (!(beta2 <= 1.53988783074545245e-03)) return 1000;
  int p = 1;
  p += beta2 > 0.0; p += beta2 > 1.38777877561156685e-17; p += beta2 > 5.77492213356056750e-12;
  p += beta2 > 3.72517661162420568e-09; p += beta2 > 1.80656771560518035e-07; p += beta2 > 2.40186660760962690e-06;
  p += beta2 > 1.52417448931310540e-05; p += beta2 > 6.09157135028591602e-05; p += beta2 > 1.78874927371965362e-04;
  p += beta2 > 4.23309807394842467e-04; p += beta2 > 8.56292603484697687e-04;
  return p;
}

// lane-partial sum of squares -> part[wave] (call before the barrier that follows the GEMM);
// read back with wg_sumsq_get after that barrier.
__device__ __forceinline__ void wg_sumsq_put(const Ctx &c, real ss) {
  ss = wave_sum(ss);
  if (wg_lane() == 0) c.part[wg_wave()] = ss;
}
__device__ __forceinline__ real wg_sumsq_get(const Ctx &c) {
  real b2 = 0.0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) b2 += c.part[w];
  return b2;
}

// composite-layer pointers of one spectral point (column-major, ld = N)
struct CompPtrs {
  gdouble *R_mp, *R_pm, *T_pp, *T_mm, *J0p, *J0m;
  int ld;  // row pitch of the four matrix blocks (column-major, N columns): N in the operator-level arrays,
           // comp_pitch(N) in the scene-level composite state
};
// The scene-level composite blocks use a row pitch of a whole number of 128-byte lines: the strip chains store
// 16-real (128-byte) column segments straight from the accumulators, and with the natural pitch N three
// quarters of them would straddle two cache lines (partial-line HBM writes).
__host__ __device__ inline int comp_pitch(int N) { return 16 * ((N + 15) / 16); }

// ---------------------------------------------------------------------------------------
// C(i,j) <- epi(i, j, sum_k A(i,k) B(k,j)) for 64 < N <= 256 (generic mode): the operands live in global memory
// (per-workgroup scratch slab or the composite layer); each k panel of 8 is staged once through LDS and feeds 8 waves
// x (<= 4 x 4) MFMA tiles (64 x 64 outputs per wave, 128 x 256 per pass).  Pipeline per panel p: the global loads of
// panel p + 3 are issued, the first k-step is multiplied from fragments read before the last barrier, panel p + 2 goes
// from registers to LDS and the first fragments of panel p + 1 are read, the second k-step is multiplied, ONE barrier.
// Measured on per-workgroup slabs at N = 256 (scratch micro-benchmark, 256 workgroups): 153 us per product against
// 185 us for the two-stage version with guarded 8-byte loads and accumulator-layout stores (109 us = MFMA peak).
//  * ElP operands (padded slab buffers) are loaded with unconditional 16-byte loads from clamped addresses, k >= N
//    zeroed by a select; other functors keep the guarded element loads.
//  * No per-wave guard around the MFMAs: a wave whose share of the pass is shorter than the register block multiplies
//    rows / columns that the epilogue discards (a guard that depends on the wave index makes every MFMA an
//    exec-masked branch); only whole rows of tiles beyond TMr are skipped, by a workgroup-uniform scalar branch.
//  * Epilogue: the accumulators (row = lq + 4 r within a tile: 32-byte pieces in 16 different columns per store) go
//    through a wave-private LDS patch and reach epi() as row pairs, 64 consecutive rows of two columns per wave
//    instruction -- the element-order stores cost 35 us of a 180 us product, the transposed ones 12.
// All threads must call; C must not alias A or B; ends WITHOUT a barrier after the epilogue (callers add theirs).
// ---------------------------------------------------------------------------------------
// operands that are padded slab buffers: ElP, and ElSigP = diag(sg) buffer diag(sg) (the added layer's r+- / t-- in the
// interaction: the signs are applied when the panel goes from registers to LDS)
template <class F> struct is_elp { static constexpr bool value = false; };
template <> struct is_elp<ElP> { static constexpr bool value = true; };
template <> struct is_elp<ElSigP> { static constexpr bool value = true; };
template <class F> struct has_sig { static constexpr bool value = false; };
template <> struct has_sig<ElSigP> { static constexpr bool value = true; };

template <int TN, class FA, class FB, class FE>
__device__ __attribute__((noinline)) void gemm_big_pass(int N, int NC, FA A, FB B, FE epi, int row0, int tcnt, int TMr, int TNr) {
  static_assert(kBigKB == 8 && kThreads == 512, "thread -> panel element mapping below");
  constexpr int TM = 4;
  // the arguments of a noinline function arrive in vector registers: made scalar again, the "workgroup-uniform" branches
  // below are s_cbranch instead of exec-masked regions and the panel addresses come from scalar bases
  // (profiles/r04_C4_ab.txt: one product per workgroup 3-6 % faster at N = 128 ... 256)
#ifndef MOM_BIG_VECTOR_ARGS
  N = __builtin_amdgcn_readfirstlane(N); NC = __builtin_amdgcn_readfirstlane(NC); row0 = __builtin_amdgcn_readfirstlane(row0);
  tcnt = __builtin_amdgcn_readfirstlane(tcnt); TMr = __builtin_amdgcn_readfirstlane(TMr); TNr = __builtin_amdgcn_readfirstlane(TNr);
#endif
  typedef real r2 __attribute__((ext_vector_type(2)));
  real *tA = mom_smem + big_tile_base_doubles(N);
  real *tB = tA + kBigStages * kBigStA;
  const int tid = wg_tid(), lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int wr = (wave >> 2) & 1, wc = wave & 3;  // 2 x 4 wave grid (8-wave build only, see wg_gemm_nc)
  const int P = (N + kBigKB - 1) / kBigKB;
  const int Np = np_for(N), ntc = (NC + 15) >> 4, StB = big_stage_b(N);
  const bool two_b = Np > 128;  // the second B column of a thread (+ 128) exists (workgroup-uniform)
  // panel element of this thread: A rows (2 lane, 2 lane + 1) of k = wave; B k = (2 kq, 2 kq + 1) of columns tid / 4, + 128
  const int kq2 = 2 * (tid & 3), cb0 = tid >> 2, cb1 = cb0 + 128;
  const int ia = row0 + 2 * lane;
  r2 ga[2], gb0[2], gb1[2];
  const real *pa = nullptr, *pb0 = nullptr, *pb1 = nullptr;
  if constexpr (is_elp<FA>::value) pa = A.p + (ia < Np - 2 ? ia : Np - 2);
  if constexpr (is_elp<FB>::value) {
    pb0 = B.p + kq2 + (size_t)(cb0 < NC - 1 ? cb0 : NC - 1) * B.ld;
    pb1 = B.p + kq2 + (size_t)(cb1 < NC - 1 ? cb1 : NC - 1) * B.ld;
  }
  gb1[0] = gb1[1] = (r2){0.0, 0.0};
  // ElSigP: the signs of this thread's fixed coordinates (rows of its A pair, columns of its B pairs)
  real sra0 = 1.0, sra1 = 1.0, scb0 = 1.0, scb1 = 1.0;
  if constexpr (has_sig<FA>::value) {
    const int ic = ia < Np - 2 ? ia : Np - 2;
    sra0 = A.sg[ic < N ? ic : N - 1];
    sra1 = A.sg[ic + 1 < N ? ic + 1 : N - 1];
  }
  if constexpr (has_sig<FB>::value) {
    scb0 = B.sg[cb0 < N ? cb0 : N - 1];
    scb1 = B.sg[cb1 < N ? cb1 : N - 1];
  }
  r4 acc[TM][TN];
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = (r4){0.0, 0.0, 0.0, 0.0};
  auto fetch = [&](int p, int set) {
    const int k0 = p * kBigKB;
    {
      const int k = k0 + wave;
      if constexpr (is_elp<FA>::value) {
        ga[set] = *(const r2 *)(pa + (size_t)(k < N ? k : N - 1) * A.ld);  // k >= N: zeroed by the stash
      } else {
        ga[set].x = (ia < N && k < N) ? A(ia, k) : 0.0;
        ga[set].y = (ia + 1 < N && k < N) ? A(ia + 1, k) : 0.0;
      }
    }
    {
      const int k = k0 + kq2;
      if constexpr (is_elp<FB>::value) {
        gb0[set] = *(const r2 *)(pb0 + k0);  // k >= N: zeroed by the stash (a select here would wait for the load)
        if (two_b) gb1[set] = *(const r2 *)(pb1 + k0);
        (void)k;
      } else {
        gb0[set].x = (k < N && cb0 < NC) ? B(k, cb0) : 0.0;
        gb0[set].y = (k + 1 < N && cb0 < NC) ? B(k + 1, cb0) : 0.0;
        if (two_b) {
          gb1[set].x = (k < N && cb1 < NC) ? B(k, cb1) : 0.0;
          gb1[set].y = (k + 1 < N && cb1 < NC) ? B(k + 1, cb1) : 0.0;
        }
      }
    }
  };
  const int wa = wave * kBigLdA + 2 * lane, wb = cb0 * kBigLdB + kq2;
  auto stash = [&](int p, int set) {
    const int st = p % kBigStages, k0 = p * kBigKB;
    r2 va = ga[set], v0 = gb0[set], v1 = gb1[set];
    if constexpr (has_sig<FA>::value) {
      const int k = k0 + wave;
      const real sk = A.sg[k < N ? k : N - 1];
      va.x *= sra0 * sk;
      va.y *= sra1 * sk;
    }
    if constexpr (has_sig<FB>::value) {
      const int k = k0 + kq2;
      const real s0 = B.sg[k < N ? k : N - 1], s1 = B.sg[k + 1 < N ? k + 1 : N - 1];
      v0.x *= s0 * scb0; v0.y *= s1 * scb0;
      v1.x *= s0 * scb1; v1.y *= s1 * scb1;
    }
    if (k0 + kBigKB > N) {  // last panel of an operator whose edge is not a multiple of 8 (uniform branch)
      if constexpr (is_elp<FA>::value) {
        if (k0 + wave >= N) va = (r2){0.0, 0.0};
      }
      if constexpr (is_elp<FB>::value) {
        if (k0 + kq2 >= N) { v0.x = 0.0; v1.x = 0.0; }
        if (k0 + kq2 + 1 >= N) { v0.y = 0.0; v1.y = 0.0; }
      }
    }
    *(r2 *)(tA + st * kBigStA + wa) = va;
    if (cb0 < Np) *(r2 *)(tB + st * StB + wb) = v0;
    if (cb1 < Np) *(r2 *)(tB + st * StB + wb + 128 * kBigLdB) = v1;
  };
  const int fa = 16 * TMr * wr + lr + lq * kBigLdA, fb = lq + (16 * TNr * wc + lr) * kBigLdB;
  real a0[TM], b0[TN], a1[TM], b1[TN];
  auto frags = [&](int p, int ks, real (&a)[TM], real (&b)[TN]) {
    const int st = p % kBigStages;
    const real *sa = tA + st * kBigStA + fa + 4 * ks * kBigLdA, *sb = tB + st * StB + fb + 4 * ks;
#pragma unroll
    for (int t = 0; t < TM; ++t) a[t] = sa[16 * t];
#pragma unroll
    for (int t = 0; t < TN; ++t) b[t] = sb[16 * t * kBigLdB];
  };
  auto mfmas = [&](real (&a)[TM], real (&b)[TN]) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
      if (ti < 2 || ti < TMr) {  // workgroup-uniform (a function of N): a scalar branch per row of tiles
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = mma16(a[ti], b[tj], acc[ti][tj]);
      }
  };
  __syncthreads();  // the previous pass (or the caller) is done with the tiles
  fetch(0, 0);
  if (P > 1) fetch(1, 1);
  stash(0, 0);
  if (P > 2) fetch(2, 0);
  if (P > 1) stash(1, 1);
  __syncthreads();
  frags(0, 0, a0, b0);
  // panels are walked two per trip so that the register-set index is a compile-time constant
  auto panel = [&](int p, auto setc) {
    constexpr int SET = decltype(setc)::value;  // set holding panel p + 2
    frags(p, 1, a1, b1);
    if (p + 3 < P) fetch(p + 3, 1 - SET);  // -> the set panel p + 1 came from, free since its stash
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (p + 2 < P) stash(p + 2, SET);
    if (p + 1 < P) frags(p + 1, 0, a0, b0);  // stage complete since the last barrier; a0 / b0 are free (in-order issue)
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  for (int p = 0; p < P; p += 2) {
    panel(p, std::integral_constant<int, 0>{});
    if (p + 1 < P) panel(p + 1, std::integral_constant<int, 1>{});
  }
  // epilogue through the wave-private patch [16 columns][32 rows, pitch kBigPatch], two rounds of two row tiles per
  // column tile (the panel area is free: every wave is past the barrier of the last panel; the next pass / caller
  // starts with a barrier)
  real *tw = tA + wave * (16 * kBigPatch);
  const int nvt = tcnt - TMr * wr < TMr ? tcnt - TMr * wr : TMr;  // valid row tiles of this wave (may be <= 0)
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    if (!(tj < TNr && TNr * wc + tj < ntc)) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (2 * h >= nvt) continue;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) tw[lr * kBigPatch + 16 * tl + cd_row(lq, r)] = acc[2 * h + tl][tj][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int cl = 4 * it + (lane >> 4), rp = 2 * (lane & 15);
        const r2 v = *(const r2 *)(tw + cl * kBigPatch + rp);
        const int col = 16 * (TNr * wc + tj) + cl, rb = 32 * h + rp, rw = row0 + 16 * TMr * wr + rb;
        if (rb < 16 * nvt && col < NC) {
          if (rw + 1 < N) {
            // an epilogue may offer a row-pair form epi(i, j, v_i, v_i+1) (i even) to use 16-byte accesses
            if constexpr (std::is_invocable_v<FE, int, int, real, real>) {
              epi(rw, col, v.x, v.y);
            } else {
              epi(rw, col, v.x);
              epi(rw + 1, col, v.y);
            }
          } else if (rw < N) {
            epi(rw, col, v.x);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------
// r4: the same pass with k panels of 16 in two LDS stages, for N <= kBig16MaxN (the three-stage K = 8 form does not leave the
// LDS for more).  Per panel: the next panel goes from registers to the other stage, the loads of the one after are issued, FOUR
// k-steps are multiplied (fragments of k-step ks + 1 requested under the MFMAs of ks), one barrier.  A mid-size pass pays its
// per-panel cost (barrier, exposed LDS / global latency: ~2 000 cycles, tools/gemm_big_bench.hip) for 16-32 MFMAs per wave in
// the K = 8 form; here for 32-64.  Panel elements per thread: A units (k, row pair of the pass's 128 rows), B units (column,
// k pair); a unit index past the end repeats the last unit (the same value stored twice): no lane-dependent branch.
// ---------------------------------------------------------------------------------------
template <int TN, bool GUARD, class FA, class FB, class FE>
__device__ __attribute__((noinline)) void gemm_big_pass16(int N, int NC, FA A, FB B, FE epi, int row0, int tcnt, int TMr, int TNr) {
  static_assert(kThreads == 512, "thread -> panel element mapping below");
  constexpr int TM = 4, KB = kBig16KB, LdB = kBig16LdB, LdA = kBigLdA, MAXU = (TN == 4) ? 4 : 3;  // B units: 8 Np / 512
  N = __builtin_amdgcn_readfirstlane(N); NC = __builtin_amdgcn_readfirstlane(NC); row0 = __builtin_amdgcn_readfirstlane(row0);
  tcnt = __builtin_amdgcn_readfirstlane(tcnt); TMr = __builtin_amdgcn_readfirstlane(TMr); TNr = __builtin_amdgcn_readfirstlane(TNr);
  typedef real r2 __attribute__((ext_vector_type(2)));
  const int tid = wg_tid(), lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the tile guards below are s_cbranch, not exec masks
  const int wr = (wave >> 2) & 1, wc = wave & 3;  // 2 x 4 wave grid
  const int Np = np_for(N), ntc = (NC + 15) >> 4;
  // Column tiles of this wave: the ntc tiles are shared out as evenly as they go (shares differ by one), and the second wave
  // row takes the shares in REVERSE order -- wave (wr, wc) sits on SIMD wc, so a SIMD's two waves then carry a long and a short
  // share (9 column tiles: 3 + 2 on every SIMD instead of 3 + 3 on three of them and none on the fourth).  Tiles a wave does not
  // own are skipped by scalar branches (wave-uniform), rows likewise.
  const int cbase = ntc >> 2, crem = ntc & 3, ci = (GUARD && wr) ? 3 - wc : wc;  // (full blocks: both wave rows in the same order --
  // the two waves of a column block then store adjacent row ranges of the same columns, which the N = 256 epilogue is 2.5 % faster with)
  const int cstart = ci * cbase + (ci < crem ? ci : crem), nvc = cbase + (ci < crem ? 1 : 0);
  constexpr int StA = KB * LdA;
  const int St = StA + Np * LdB;
  real *tS = mom_smem + big_tile_base_doubles(N);
  const int P = (N + KB - 1) / KB;
  // ---- panel elements of this thread
  // A: 16 k x 64 row pairs (128 rows of the pass) = two units per thread
  const int nUB = (KB / 2) * Np;
  const int ub = (nUB + kThreads - 1) / kThreads;  // <= MAXU (uniform)
  auto unitA = [&](int j, int &k, int &i) {
    const int u = tid + kThreads * j;          // < 16 x 64 for j < 2
    k = u >> 6;
    i = row0 + 2 * (u & 63);
  };
  auto unitB = [&](int j, int &k, int &c) {
    const int u = tid + kThreads * j, uB = u < nUB ? u : nUB - 1;
    c = uB >> 3;
    k = 2 * (uB & 7);
  };
  int la[2], lb[MAXU];
  unsigned goa[2], gob[MAXU];
  real sa0[2], sa1[2], sb[MAXU];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int ak, ai;
    unitA(j, ak, ai);
    la[j] = ak * LdA + (ai - row0);
    goa[j] = 0;
    if constexpr (is_elp<FA>::value) goa[j] = (unsigned)((ai < Np - 2 ? ai : Np - 2) + ak * A.ld);
    sa0[j] = sa1[j] = 1.0;
    if constexpr (has_sig<FA>::value) {
      const int ic = ai < Np - 2 ? ai : Np - 2;
      sa0[j] = A.sg[ic < N ? ic : N - 1];
      sa1[j] = A.sg[ic + 1 < N ? ic + 1 : N - 1];
    }
  }
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    int bk, bc;
    unitB(j, bk, bc);
    lb[j] = StA + bc * LdB + bk;
    gob[j] = 0;
    if constexpr (is_elp<FB>::value) gob[j] = (unsigned)(bk + (bc < NC - 1 ? bc : NC - 1) * B.ld);
    sb[j] = 1.0;
    if constexpr (has_sig<FB>::value) sb[j] = B.sg[bc < N ? bc : N - 1];
  }
  r2 ga[2], gb[MAXU];
  auto fetch = [&](int p) {
    const int k0 = p * KB;
    const bool tail = k0 + KB > N;  // last panel of an operator whose edge is not a multiple of 16 (uniform)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (is_elp<FA>::value) {
        int kc = k0;
        if (tail) {
          int ak, ai;
          unitA(j, ak, ai);
          kc = (k0 + ak < N ? k0 + ak : N - 1) - ak;
        }
        ga[j] = *(const r2 *)(A.p + (size_t)kc * A.ld + goa[j]);
      } else {
        int ak, i;
        unitA(j, ak, i);
        const int k = k0 + ak;
        ga[j].x = (i < N && k < N) ? A(i, k) : 0.0;
        ga[j].y = (i + 1 < N && k < N) ? A(i + 1, k) : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      if (j < ub) {
        if constexpr (is_elp<FB>::value) {
          gb[j] = *(const r2 *)(B.p + k0 + gob[j]);  // (k >= N: inside the padded column; zeroed by the stash)
        } else {
          int bk, c;
          unitB(j, bk, c);
          const int k = k0 + bk;
          gb[j].x = (k < N && c < NC) ? B(k, c) : 0.0;
          gb[j].y = (k + 1 < N && c < NC) ? B(k + 1, c) : 0.0;
        }
      }
    }
  };
  auto stash = [&](int p) {
    real *st = tS + (p & 1) * St;
    const int k0 = p * KB;
    const bool tail = k0 + KB > N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r2 v = ga[j];
      if constexpr (has_sig<FA>::value) {
        const int k = k0 + ((tid + kThreads * j) >> 6);
        const real sk = A.sg[k < N ? k : N - 1];
        v.x *= sa0[j] * sk;
        v.y *= sa1[j] * sk;
      }
      if constexpr (is_elp<FA>::value) {
        if (tail && k0 + ((tid + kThreads * j) >> 6) >= N) v = (r2){0.0, 0.0};
      }
      *(r2 *)(st + la[j]) = v;
    }
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      if (j < ub) {
        r2 v = gb[j];
        int bk = 0, bc = 0;
        if (has_sig<FB>::value || tail) unitB(j, bk, bc);
        if constexpr (has_sig<FB>::value) {
          const int k = k0 + bk;
          v.x *= B.sg[k < N ? k : N - 1] * sb[j];
          v.y *= B.sg[k + 1 < N ? k + 1 : N - 1] * sb[j];
        }
        if constexpr (is_elp<FB>::value) {
          if (tail) {
            if (k0 + bk >= N) v.x = 0.0;
            if (k0 + bk + 1 >= N) v.y = 0.0;
          }
        }
        *(r2 *)(st + lb[j]) = v;
      }
    }
  };
  r4 acc[TM][TN];
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = (r4){0.0, 0.0, 0.0, 0.0};
  const int nvt = tcnt - TMr * wr < TMr ? tcnt - TMr * wr : TMr;  // valid row tiles of this wave (may be <= 0)
  const int fa = 16 * TMr * wr + lr + lq * LdA, fb = StA + lq + (16 * cstart + lr) * LdB;
  real a0[TM], b0[TN], a1[TM], b1[TN];
  auto frags = [&](const real *st, int ks, real (&a)[TM], real (&b)[TN]) {
    const real *sa = st + fa + 4 * ks * LdA, *sb_ = st + fb + 4 * ks;
#pragma unroll
    for (int t = 0; t < TM; ++t) a[t] = sa[16 * t];
#pragma unroll
    for (int t = 0; t < TN; ++t) b[t] = sb_[16 * t * LdB];
  };
  // GUARD (template parameter: two separately compiled functions -- both loops in one function cost either of them 5-8 %):
  // tiles this wave does not own are skipped by scalar branches; without it every wave multiplies the full TM x TN block,
  // which costs nothing when all blocks are full (N = 128, 256: the branches themselves cost ~2 %)
  auto mfmas = [&](real (&a)[TM], real (&b)[TN]) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
      if (GUARD ? ti < nvt : (ti < 2 || ti < TMr)) {  // wave- / workgroup-uniform: scalar branches
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          if (!GUARD || tj < nvc) acc[ti][tj] = mma16(a[ti], b[tj], acc[ti][tj]);
      }
  };
  __syncthreads();  // the previous pass (or the caller) is done with the stages
  fetch(0);
  stash(0);
  if (P > 1) fetch(1);
  __syncthreads();
  for (int p = 0; p < P; ++p) {
    const real *st = tS + (p & 1) * St;
    frags(st, 0, a0, b0);
#ifndef BIG16_NOFETCH
    if (p + 1 < P) stash(p + 1);       // stage (p + 1) & 1: free since the barrier that ended panel p - 1
    if (p + 2 < P) fetch(p + 2);
#endif
    frags(st, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    frags(st, 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    frags(st, 3, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  // epilogue through the wave-private patch, as in gemm_big_pass
  real *tw = tS + wave * (16 * kBigPatch);
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    if (!(tj < nvc)) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (2 * h >= nvt) continue;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) tw[lr * kBigPatch + 16 * tl + cd_row(lq, r)] = acc[2 * h + tl][tj][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int cl = 4 * it + (lane >> 4), rp = 2 * (lane & 15);
        const r2 v = *(const r2 *)(tw + cl * kBigPatch + rp);
        const int col = 16 * (cstart + tj) + cl, rb = 32 * h + rp, rw = row0 + 16 * TMr * wr + rb;
        if (rb < 16 * nvt && col < NC) {
          if (rw + 1 < N) {
            if constexpr (std::is_invocable_v<FE, int, int, real, real>) {
              epi(rw, col, v.x, v.y);
            } else {
              epi(rw, col, v.x);
              epi(rw + 1, col, v.y);
            }
          } else if (rw < N) {
            epi(rw, col, v.x);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <class FA, class FB, class FE>
__device__ void wg_gemm_big(int N, int NC, FA A, FB B, FE epi) {
  // Wave tiling fitted to the operator: a wave owns TMr x TNr MFMA tiles (each <= 4), TNr = ceil(column tiles / 4) and
  // TMr = ceil(row tiles of this pass / 2), the row tiles split evenly over ceil(row tiles / 8) passes.  N = 256 keeps
  // the 4 x 4 register block of two passes; N = 66 (5 x 5 tiles) runs 3 x 2 per wave instead of one wave carrying 16.
  // The register block is compiled for 4 x TN tiles, TN = 2, 3, 4; rows of tiles beyond TMr are skipped by a scalar branch.
  const int ntr = (N + 15) >> 4, ntc = (NC + 15) >> 4;
  const int TNr = (ntc + 3) >> 2;
  const int npass = (ntr + 7) >> 3;
  int t0 = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int tcnt = (ntr - t0 + (npass - pass) - 1) / (npass - pass);  // row tiles of this pass
    const int TMr = (tcnt + 1) >> 1;
    const int row0 = 16 * t0;
    t0 += tcnt;
#ifndef MOM_NO_BIG16
    if (N <= kBig16MaxN) {
      // full register blocks on every wave (column tiles a multiple of 4, an even number of row tiles): the unguarded image
      const bool full = (ntc & 3) == 0 && tcnt == 2 * TMr;
      if (full) {
        if (TNr <= 2) gemm_big_pass16<2, false>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
        else if (TNr == 3) gemm_big_pass16<3, false>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
        else gemm_big_pass16<4, false>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
      } else {
        if (TNr <= 2) gemm_big_pass16<2, true>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
        else if (TNr == 3) gemm_big_pass16<3, true>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
        else gemm_big_pass16<4, false>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);  // N > 192: the unguarded image (13-16 column tiles: at most one of 16 blocks short)
      }
      continue;
    }
#endif
    if (TNr <= 2) gemm_big_pass<2>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
    else if (TNr == 3) gemm_big_pass<3>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
    else gemm_big_pass<4>(N, NC, A, B, epi, row0, tcnt, TMr, TNr);
  }
}

}  // namespace MOM_NS
#include "mom_strip.hpp"
#include "mom_regdbl.hpp"
namespace MOM_NS {

// ---------------------------------------------------------------------------------------
// Ob <- T (I - B)^-1 with B in Bb (destroyed) and beta2 = ||B||_F^2.  T: element functor usable
// as MFMA A operand.  See the header comment for the series bound.  Ends with a barrier.
// ---------------------------------------------------------------------------------------
template <bool LDSM, class FT>
__device__ __forceinline__ void times_inv(Ctx &c, FT T, real *&Bb, real *&Ob, real beta2) {
  const int N = c.N, ld = c.ld, NN = N * N;
  int p = __builtin_amdgcn_readfirstlane((c.inv_mode == 1) ? 1000 : neumann_terms(c.thr, beta2));  // uniform
  if constexpr (!LDSM && kF64) {
    // Generic mode: the pivoted Gauss-Jordan works in the global slab and costs ~50 products of a step at N = 256, so the
    // series is carried further than the table (beta <= 0.29): the smallest p with beta^p / (1 - beta) <= 2^-56 from the
    // norm itself, up to beta = 0.9 (p <= 512: nine squarings, 19 products).  Same bound as the table's, same products.
    if (p > 32 && c.inv_mode != 1 && beta2 < 0.81) {
      const real beta = sqrt(beta2);
      p = (int)ceil((38.816242111356935 - log(1.0 - beta)) / -log(beta));
      if (p < 33) p = 33;
    }
  }
  MOM_STAMP(6);
  if (p <= 4) {
    // Horner: A_1 = T, A_{k+1} = T + A_k B
    real *o = Ob;
    if (p == 1) {
      for (int e = wg_tid(); e < NN; e += kThreads) {
        int i, j;
        c.fd.split(e, i, j);
        o[i + j * ld] = T(i, j);
      }
      __syncthreads();
    } else {
      wg_gemm<false, !LDSM>(N, T, ElP{Bb, ld}, [=](int i, int j, real v) { o[i + j * ld] = T(i, j) + v; });
      __syncthreads();
      for (int k = 3; k <= p; ++k)
        gemm_to<LDSM>(c, Ob, ElP{Ob, ld}, ElP{Bb, ld}, [=](int i, int j, real v, real) { return T(i, j) + v; });
    }
  } else if (p <= 512) {
    // G = (I + B)(I + B^2)(I + B^4)... ; Ob <- T G
    if constexpr (!LDSM) {
      slab_eye_plus<false>(c, Ob, Bb);
      __syncthreads();
    } else {
      real *o = Ob, *b = Bb;
      for (int e = wg_tid(); e < NN; e += kThreads) {
        int i, j;
        c.fd.split(e, i, j);
        o[i + j * ld] = ((i == j) ? 1.0 : 0.0) + b[i + j * ld];
      }
      __syncthreads();
    }
    for (int terms = 2; terms < p; terms *= 2) {
      gemm_to<LDSM>(c, Bb, ElP{Bb, ld}, ElP{Bb, ld}, [=](int, int, real v, real) { return v; });
      gemm_to<LDSM>(c, Ob, ElP{Ob, ld}, ElP{Bb, ld}, [=](int, int, real v, real old) { return old + v; });
    }
    gemm_to<LDSM>(c, Ob, T, ElP{Ob, ld}, [=](int, int, real v, real) { return v; });
  } else {
    real *b = Bb, *o = Ob;
    if constexpr (!LDSM) {
      slab_eye_plus<true>(c, b, b);
    } else {
      for (int e = wg_tid(); e < NN; e += kThreads) {
        int i, j;
        c.fd.split(e, i, j);
        b[i + j * ld] = ((i == j) ? 1.0 : 0.0) - b[i + j * ld];
      }
    }
    __syncthreads();
    if (N <= 64) wg_inverse_reg(N, b, ld, c.part, c.prow, c.ipiv, c.bad);  // part: >= 128 doubles (2*kWaves*ldv)
    else wg_inverse(N, c.fd, b, ld, c.prow, c.pcol, c.rowk, c.ipiv, c.sh, c.bad);
    wg_gemm<false, !LDSM>(N, T, ElP{b, ld}, [=](int i, int j, real v) { o[i + j * ld] = v; });
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// stream constants -> LDS (call after zero_padding + barrier; needs a barrier after)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_streams(const Ctx &c, const DevStreams &q) {
  for (int i = wg_tid(); i < c.Np; i += kThreads) {
    c.mu[i] = (i < c.N) ? q.mu[i] : 1.0;
    c.wt[i] = (i < c.N) ? q.wt[i] : 0.0;
    c.sg[i] = (i < c.N) ? q.sg[i] : 1.0;
  }
  if (wg_tid() == 0) *c.bad = 0;
}

template <bool LDSM>
__device__ __forceinline__ void wg_prologue(Ctx &c, const DevStreams &q, int N, real *smem, real *gscratch) {
  make_ctx<LDSM>(c, N, q.inv_mode, smem, gscratch);   // N: q.N, or the compile-time size of a strip image
  zero_padding<LDSM>(c);
  __syncthreads();
  load_streams(c, q);
  __syncthreads();
}
template <bool LDSM>
__device__ __forceinline__ void wg_prologue(Ctx &c, const DevStreams &q, real *smem, real *gscratch) {
  wg_prologue<LDSM>(c, q, q.N, smem, gscratch);
}

// ---------------------------------------------------------------------------------------
// elemental! into c.r (r-+), c.t (t++), c.jp, c.jm.  Zpp(i,j), Zmp(i,j): phase-matrix
// element functors for this spectral point.  Ends with a barrier.
// ---------------------------------------------------------------------------------------
template <class FZP, class FZM>
__device__ __forceinline__ void elemental_build(const Ctx &c, const DevStreams &q, int m, int nd, real tau_sum,
                                                real dtau, real varpi, FZP Zpp, FZM Zmp) {
  const int N = c.N, ld = c.ld, n = q.nS;
  const real winv = (m == 0) ? 0.5 : 0.25;   // wt / wdiv with wdiv = 2 or 4: a power of two, the product is the same value
  const real wct02 = (m == 0) ? 0.5 : 0.25;
  // exp(-dtau/mu_i) per stream and 1 - exp(-dtau (1/mu_i + 1/mu_j)) per PAIR OF STREAMS: the Stokes
  // components of a stream share mu, so the nS^2-fold repeated exponentials of get_elem_rt!
  // (elemental.jl:176) are evaluated once (same expression, same operands: identical values).
  const int ns = q.regular ? n : 1;
  const int Nq = N / ns;
  FastDiv fs;
  fs.init(ns);
  FastDiv fq;
  fq.init(Nq);
  // per stream pair (iq, jq): E = 1 - exp(..), F1 = mu_j/(mu_i+mu_j), F2 = mu_j/(mu_i-mu_j); the Q buffer is free here
  // (strip images: F1, F2 and SI = 1/mu_i + 1/mu_j come from the workgroup's persistent tables, c.ptab)
  const bool pt = c.ptab != nullptr;
  real *E = c.tabE ? c.tabE : c.Q, *F1q = E + Nq * Nq, *F2q = E + 2 * Nq * Nq;
  const real *F1 = pt ? c.ptab : F1q, *F2 = pt ? c.ptab + Nq * Nq : F2q, *SIp = pt ? c.ptab + 2 * Nq * Nq : nullptr;
  // two-term phase matrices (Rayleigh + one aerosol type): the 32 basis loads of this thread's first 8 elements go
  // out before the tables are built and are consumed after them
  // Element enumeration: lane = row (in blocks of 64), wave = column (strided by kWaves).  A thread's elements share
  // the row, so its per-row constants are loaded once, and everything per column is wave-uniform; the phase-matrix
  // loads stay coalesced (consecutive lanes = consecutive rows of one column).  Slot s -> (row block, column slot).
  // (4-wave build: the operators there have N <= 48, most lanes of a row block would idle -- plain linear order.)
  constexpr bool kRowLanes = kWaves >= 8;
  const int lane_e = wg_lane(), wave_e = wg_wave(), tid_e = wg_tid();
  const int CS = (N + kWaves - 1) / kWaves;
  const int slots = kRowLanes ? ((N + 63) >> 6) * CS : (N * N + kThreads - 1) / kThreads;
  const bool pre = Zpp.terms() == 2;
  real bp0[8], bm0[8], bp1[8], bm1[8];
  int ii[8], jj[8];
  bool ok[8];
  // r6 (quad-block image: ONE wave per workgroup, nobody to hide a load behind): the two-term path keeps the coordinates of the
  // batch it is computing apart from those of the batch it has already requested
  constexpr bool kPipeZ = (kWaves == 1);
  int ci[8], cj[8];
  bool cok[8];
  auto coords = [&](int s0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u;
      if constexpr (kRowLanes) {
        const int rb = s / CS;
        ii[u] = lane_e + 64 * rb;
        jj[u] = wave_e + kWaves * (s - rb * CS);
        ok[u] = s < slots && ii[u] < N && jj[u] < N;
      } else {
        const int e = tid_e + s * kThreads;
        ok[u] = e < N * N;
        c.fd.split(ok[u] ? e : 0, ii[u], jj[u]);
      }
    }
  };
  auto issue_z2 = [&](int s0) {
    coords(s0);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ok[u]) {
        bp0[u] = Zpp.basis(0, ii[u], jj[u]);
        bm0[u] = Zmp.basis(0, ii[u], jj[u]);
        bp1[u] = Zpp.basis(1, ii[u], jj[u]);
        bm1[u] = Zmp.basis(1, ii[u], jj[u]);
      }
  };
  if (pre) issue_z2(0);
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.ei[i] = exp(-dtau / c.mu[i]);
    c.v1[i] = c.wt[i] * winv;   // wct
    c.v2[i] = dtau / c.mu[i];
  }
  // the tables pay (and fit the buffer) when several Stokes components share a stream; scalar problems with
  // large N evaluate the same expressions per element instead
  const bool tab = 3 * Nq * Nq <= (int)mat_elems(N);
  if (tab && pt) {
    for (int e = wg_tid(); e < Nq * Nq; e += kThreads) E[e] = 1 - exp(-dtau * SIp[e]);
  } else if (tab)
    for (int e = wg_tid(); e < Nq * Nq; e += kThreads) {
      int iq, jq;
      fq.split(e, iq, jq);
      const real mui = c.mu[iq * ns], muj = c.mu[jq * ns];
      E[e] = 1 - exp(-dtau * ((1 / mui) + (1 / muj)));
      F1q[e] = muj / (mui + muj);
      F2q[e] = muj / (mui - muj);
    }
  __syncthreads();
  MOM_STAMP(46);
  const int i_start = n * (q.imu0 - 1), i_end = n * q.imu0;
  real *ZS = c.tabZS ? c.tabZS : c.P;  // mixed Z++ / Z-+ of the sun-block columns, [N x nS] each (P is free here)
  for (int s0 = 0; s0 < slots; s0 += 8) {
    real zp[8], zm[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      zp[u] = 0.0;
      zm[u] = 0.0;
    }
    // Z = sum_k w_k Z_k, accumulated in k order
    if (pre) {
      if (!kPipeZ && s0 != 0) issue_z2(s0);
      const real wp0 = Zpp.weight(0), wm0 = Zmp.weight(0), wp1 = Zpp.weight(1), wm1 = Zmp.weight(1);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ok[u]) {
          zp[u] += wp0 * bp0[u];
          zm[u] += wm0 * bm0[u];
          zp[u] += wp1 * bp1[u];
          zm[u] += wm1 * bm1[u];
        }
      if constexpr (kPipeZ) {   // one-wave workgroups: the next batch's 32 basis loads travel under this batch's element math
#pragma unroll
        for (int u = 0; u < 8; ++u) { ci[u] = ii[u]; cj[u] = jj[u]; cok[u] = ok[u]; }
        if (s0 + 8 < slots) issue_z2(s0 + 8);
      }
    } else {
      // the (run-time) sum over scatterer types is the OUTER loop: the 16 basis loads of a term are in flight together
      coords(s0);
      for (int k = 0; k < Zpp.terms(); ++k) {
        const real wp = Zpp.weight(k), wm = Zmp.weight(k);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (ok[u]) {
            zp[u] += wp * Zpp.basis(k, ii[u], jj[u]);
            zm[u] += wm * Zmp.basis(k, ii[u], jj[u]);
          }
        }
      }
    }
    MOM_STAMP(49);
    // the table / no-table choice is hoisted out of the element loop (as a compile-time flag of the body), so the
    // table path carries neither the exponential nor the divisions of the other one
    auto element_math = [&](auto tabc) {
      constexpr bool TAB = decltype(tabc)::value;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool pz = kPipeZ && pre;
        if (pz ? cok[u] : ok[u]) {
          const int i = pz ? ci[u] : ii[u], j = pz ? cj[u] : jj[u];
          if (j >= i_start && j < i_end) {
            ZS[i + (j - i_start) * N] = zp[u];
            ZS[i + (n + j - i_start) * N] = zm[u];
          }
          const real mui = c.mu[i], muj = c.mu[j];
          const real wj = c.v1[j];
          int iq, jq, dummy;
          fs.split(i, dummy, iq);
          fs.split(j, dummy, jq);
          const int pq = iq + jq * Nq;
          real rr, tt;
          if (wj > 1.e-8) {
            real Epq, F1pq;
            if constexpr (TAB) {
              Epq = E[pq];
              F1pq = F1[pq];
            } else {
              Epq = 1 - exp(-dtau * ((1 / mui) + (1 / muj)));
              F1pq = muj / (mui + muj);
            }
            rr = varpi * zm[u] * F1pq * wj * Epq;
            if (mui == muj) {
              if (i == j) {
                tt = c.ei[i] * (1 + varpi * zp[u] * c.v2[i] * c.v1[i]);
              } else {
                tt = 0.0;
              }
            } else {
              real F2pq;
              if constexpr (TAB) F2pq = F2[pq];
              else F2pq = muj / (mui - muj);
              tt = varpi * zp[u] * F2pq * wj * (c.ei[i] - c.ei[j]);
            }
          } else {
            rr = 0.0;
            tt = (i == j) ? c.ei[i] : 0.0;
          }
          if (nd >= 1) rr *= c.sg[i];  // apply_D_elemental!, elemental.jl:265-269
          c.r[i + j * ld] = rr;
          c.t[i + j * ld] = tt;
        }
      }
    };
    if (tab) element_math(std::true_type{});
    else element_math(std::false_type{});
    MOM_STAMP(57);
  }
  __syncthreads();
  MOM_STAMP(47);
  const real mus = c.mu[i_start];
  const real att = exp(-tau_sum / mus);
  for (int i = wg_tid(); i < N; i += kThreads) {
    real zp = 0.0, zm = 0.0;
    for (int k = 0; k < n; ++k) {
      zp += ZS[i + k * N] * q.I0[k];
      zm += ZS[i + (n + k) * N] * q.I0[k];
    }
    const real mui = c.mu[i];
    real jp, jm;
    int iq, sq, dummy;
    fs.split(i, dummy, iq);
    fs.split(i_start, dummy, sq);
    if (i >= i_start && i < i_end)
      jp = wct02 * varpi * zp * (dtau / mui) * c.ei[i];
    else
      jp = wct02 * varpi * zp * (mus / (mui - mus)) * (c.ei[i] - c.ei[i_start]);
    jm = wct02 * varpi * zm * (mus / (mui + mus)) * (tab ? E[iq + sq * Nq] : 1 - exp(-dtau * ((1 / mui) + (1 / mus))));
    jp *= att;
    jm *= att;
    if (nd >= 1) jm = q.D[i % n] * jm;  // elemental.jl:249-251
    c.jp[i] = jp;
    c.jm[i] = jm;
  }
  __syncthreads();
  MOM_STAMP(48);
  // P and Q served as table space (linear: the sun-block Z columns land in P's padding rows): their K padding must read
  // as zero again -- always in the Float32 build, whose strip chains run every k-step of the last row tile (r4: without
  // this the chains' riding rows met the stale table entries from the third series term on and a thick layer lost
  // 3 x the accuracy of the general path)
  if (N % 4 != 0 || !kF64) {
    rezero_padding(c, c.P);
    rezero_padding(c, c.Q);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// doubling_helper!: nd doublings of (r, t, jp, jm) held in the context, then the D signs.
// expk: this point's exp(-dtau/mu0) (returned squared nd times).  Ends with a barrier.
// ---------------------------------------------------------------------------------------
template <bool LDSM, int KS = 0>
__device__ __forceinline__ real doubling_run(Ctx &c, int nd, real expk, const CompPtrs *pre = nullptr) {
  const int N = c.N, ld = c.ld;
  if (nd == 0) return expk;
  // "ride": the source vectors travel as columns N, N+1 of the B operand r (buffer padding), so
  // r j and Q (..) come out of the MFMA products for free.  Needs two spare columns in the last
  // column tile and no K padding (N % 4 == 0); otherwise: separate mat-vec passes.
  const bool ride = (N % 4 == 0) && (c.nc - N >= 2);
  int it0 = 0;
  if constexpr (!LDSM && kF64 && kWaves == 8) {
    // 64 < N <= 96: the operators of the loop live in registers (mom_regdbl.hpp); what it cannot do (forced pivoting, a
    // series beyond 512 terms) comes back here with r, t, j0+- of the step it stopped at
    if (rg_applies(N) && c.inv_mode == 0) {
      it0 = (np_for(N) == 80) ? rg_doubling<5>(c, nd, &expk) : rg_doubling<6>(c, nd, &expk);
    }
  }
  if (ride && it0 < nd) {
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.r[i + N * ld] = c.jp[i];
      c.r[i + (N + 1) * ld] = c.jm[i];
    }
    __syncthreads();
  }
  for (int it = it0; it < nd; ++it) {
    bool strip_ok = false;
#ifndef MOM_NO_DBL_STRIP  // (diagnostic builds: the general doubling step inside a strip image)
    if constexpr (LDSM && KS > 0) strip_ok = ride && c.inv_mode == 0 && N == 4 * KS;
#endif
#ifdef MOM_DIAG_CHAIN_LT  // (diagnostic builds: chains only in the first / only after the first so many steps)
    strip_ok = strip_ok && it < MOM_DIAG_CHAIN_LT;
#endif
#ifdef MOM_DIAG_CHAIN_GE
    strip_ok = strip_ok && it >= MOM_DIAG_CHAIN_GE;
#endif
    // r r on strips where the general product has no straight-line schedule (4-wave build); the 8-wave build's
    // 8 x 2-tile product is faster than 4 strip waves
    if (strip_ok && kWaves == 4) {
      if constexpr (LDSM && KS > 0) doubling_rr_strip<KS>(c);
      __syncthreads();
    } else {
      real *r = c.r, *P = c.P;
      real ss = 0.0;
      MOM_STAMP(0);
      // P = r r (+ r j0+, r j0- riding) ; Q = t (I - r r)^-1  (tt⁺⁺_gp_refl)   (doubling.jl:44-48)
      wg_gemm_nc<false, !LDSM>(N, ride ? N + 2 : N, ElP{r, ld}, ElP{r, ld}, [=, &ss](int i, int j, real v) {
        P[i + j * ld] = v;
        if (j < N) ss += v * v;
      });
      wg_sumsq_put(c, ss);
      __syncthreads();
      MOM_STAMP(1);
    }
    MOM_STAMP(70);
    const real beta2 = wg_sumsq_get(c);
    if constexpr (LDSM && KS > 0) {
      // strip-chained step (mom_strip.hpp): the series, A r, (A r) t and A t in one barrier-free MFMA stream
      if (strip_ok) {
        int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(beta2));  // workgroup-uniform: scalar loop control in the chains
#ifdef MOM_DIAG_PPLUS  // (diagnostic builds: one series term more than the bound asks for)
        p += MOM_DIAG_PPLUS;
#endif
#ifdef MOM_DIAG_PMAX  // (diagnostic builds: longer series go to the general path)
        if (p > MOM_DIAG_PMAX) p = 1000;
#endif
        if (p <= kStripMaxP) {
#ifdef MOM_QPREFETCH
          doubling_step_strip<KS>(c, p, expk, (it == nd - 1) ? pre : nullptr);
#else
          doubling_step_strip<KS>(c, p, expk);
#endif
          expk = expk * expk;
          continue;
        }
      }
    }
    if (ride) {
      // w1 = j1- + r j0+ ; w2 = j0+ + r j1-   with j1± = j0± expk   (:51-60)
      real *r = c.r, *P = c.P;
      for (int i = wg_tid(); i < N; i += kThreads) {
        const real rjp = P[i + N * ld], rjm = P[i + (N + 1) * ld];
        r[i + N * ld] = c.jm[i] * expk + rjp;
        r[i + (N + 1) * ld] = c.jp[i] + expk * rjm;
      }
    }
    times_inv<LDSM>(c, ElP{c.t, ld}, c.P, c.Q, beta2);
    MOM_STAMP(2);
    real *r = c.r, *t = c.t, *P = c.P, *Q = c.Q;
    if (!ride) {
      // j1± = j0± expk                                       (:51,:54)
      for (int i = wg_tid(); i < N; i += kThreads) {
        c.j1p[i] = c.jp[i] * expk;
        c.j1m[i] = c.jm[i] * expk;
      }
      __syncthreads();
      // v1 = r j0+ ; v2 = r j1-
      if constexpr (LDSM) wg_matvec2(N, c.ldv, ElP{r, ld}, c.jp, c.j1m, c.v1, c.v2, c.part);
      else wg_matvec_slab<2>(c, r, c.jp, c.j1m, c.v1, c.v2);
      for (int i = wg_tid(); i < N; i += kThreads) {
        c.v1[i] = c.j1m[i] + c.v1[i];  // j1- + r j0+
        c.v2[i] = c.jp[i] + c.v2[i];   // j0+ (old) + r j1-
      }
      __syncthreads();
      if constexpr (LDSM) wg_matvec2(N, c.ldv, ElP{Q, ld}, c.v1, c.v2, c.v1, c.v2, c.part);
      else wg_matvec_slab<2>(c, Q, c.v1, c.v2, c.v1, c.v2);
      for (int i = wg_tid(); i < N; i += kThreads) {
        c.jm[i] = c.jm[i] + c.v1[i];   // :57
        c.jp[i] = c.j1p[i] + c.v2[i];  // :60
      }
    }
    MOM_STAMP(3);
    // P = Q r  (+ Q w1, Q w2 riding)
    wg_gemm_nc<false, !LDSM>(N, ride ? N + 2 : N, ElP{Q, ld}, ElP{r, ld}, [=](int i, int j, real v) { P[i + j * ld] = v; });
    __syncthreads();
    MOM_STAMP(4);
    if (ride) {
      // j0- += Q w1 (:57) ; j0+ = j1+ + Q w2 (:60); refresh the riding columns for the next step
      for (int i = wg_tid(); i < N; i += kThreads) {
        const real jm = c.jm[i] + P[i + N * ld];
        const real jp = c.jp[i] * expk + P[i + (N + 1) * ld];
        c.jm[i] = jm;
        c.jp[i] = jp;
        r[i + N * ld] = jp;
        r[i + (N + 1) * ld] = jm;
      }
    }
    expk = expk * expk;  // :61
    // r = r + P t (:64) ; t = Q t (:67)
    if (LDSM) {
      wg_gemm2<true>(N, ElP{P, ld}, ElP{Q, ld}, ElP{t, ld},
                     [=](int i, int j, real v) { r[i + j * ld] = r[i + j * ld] + v; },
                     [=](int i, int j, real v) { t[i + j * ld] = v; });
    } else {
      real *X = c.X;
      wg_gemm2<false, !LDSM>(N, ElP{P, ld}, ElP{Q, ld}, ElP{t, ld},
                      [=](int i, int j, real v) { r[i + j * ld] = r[i + j * ld] + v; },
                      [=](int i, int j, real v) { X[i + j * ld] = v; });
      c.X = t;
      c.t = X;
    }
    __syncthreads();
    MOM_STAMP(5);
  }
  // apply_D! (doubling.jl:93-110) and apply_D_SFI! (:112-118): r-+ rows and j0- scaled by sg
  {
    real *r = c.r;
    for (int e = wg_tid(); e < N * N; e += kThreads) {
      int i, j;
      c.fd.split(e, i, j);
      r[i + j * ld] *= c.sg[i];
    }
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.jm[i] *= c.sg[i];
      if (ride) { r[i + N * ld] = 0.0; r[i + (N + 1) * ld] = 0.0; c.P[i + N * ld] = 0.0; c.P[i + (N + 1) * ld] = 0.0; }
    }
  }
  __syncthreads();
  return expk;
}

// ---------------------------------------------------------------------------------------
// interaction_helper!: composite (global) <- composite (+) added.  Added r-+ in c.r, t++ in
// c.t, j0+ in c.jp, j0- in c.jm; added r+- and t-- as element functors.  Ends with barrier.
// ---------------------------------------------------------------------------------------
// IFACE: compile-time interface code 0..3 (the code of the other three cases is not generated: the
// per-layer kernels are launched with the layer's code as a template argument, which keeps the
// instruction footprint of the common 11 case small), or -1 to select at run time.
template <bool LDSM, int IFACE, int KS = 0, class FRPM, class FTMM>
__device__ __forceinline__ void interaction_core(Ctx &c, int iface_rt, const CompPtrs &g, FRPM rpm, FTMM tmm) {
  const int iface = (IFACE >= 0) ? IFACE : iface_rt;
  const int N = c.N, ld = c.ld, cl = g.ld;
  real *r = c.r, *t = c.t;
  if constexpr (LDSM && KS > 0 && std::is_same<FRPM, ElSigP>::value && std::is_same<FTMM, ElSigP>::value) {
    // ScatteringInterface_11 with r+- = D r-+ D, t-- = D t++ D of the layer held in c.r, c.t: two strip chains
    if ((IFACE < 0 || IFACE == 3) && iface == 3 && c.inv_mode == 0 && N == 4 * KS && rpm.p == c.r && tmm.p == c.t) {
#ifndef MOM_NO_INT_STRIP  // (diagnostic builds: the general interaction inside a strip image)
      if (interaction_strip<KS>(c, g)) return;
#endif
    }
  }
  if constexpr (!LDSM && kF64 && kWaves == 8 && std::is_same<FRPM, ElSigP>::value && std::is_same<FTMM, ElSigP>::value) {
    // 64 < N <= 96, interface 11, r+- / t-- the sign conjugates of the layer in c.r, c.t: register-resident operators
    // (mom_regdbl.hpp); returns false before it has stored anything when an inverse needs the pivoted form
    if ((IFACE < 0 || IFACE == 3) && iface == 3 && c.inv_mode == 0 && rg_applies(N) && rpm.p == c.r && tmm.p == c.t) {
      if ((np_for(N) == 80) ? rg_interaction<5>(c, g) : rg_interaction<6>(c, g)) return;
    }
  }
  // composite sources -> LDS
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.Jp[i] = g.J0p[i];
    c.Jm[i] = g.J0m[i];
  }
  __syncthreads();
  if ((IFACE < 0 || IFACE == 0) && iface == 0) {
    // J0+ = j0+ + t++ J0+ ; J0- = J0- + T-- j0-            (interaction.jl:16-17)
    wg_matvec(c, ElP{t, ld}, c.Jp, c.v1);
    wg_matvec(c, El{g.T_mm, cl, N}, c.jm, c.v2);
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.Jp[i] = c.jp[i] + c.v1[i];
      c.Jm[i] = c.Jm[i] + c.v2[i];
    }
    // T-- = t-- T-- ; T++ = t++ T++                          (:20-21)
    copy_to_slab<LDSM>(c, g.T_mm, cl, c.P, ld);
    copy_to_slab<LDSM>(c, g.T_pp, cl, c.Q, ld);
    __syncthreads();
    gdouble *Tmm = g.T_mm, *Tpp = g.T_pp;
    wg_gemm<false, !LDSM>(N, tmm, ElP{c.P, ld}, [=](int i, int j, real v) { Tmm[i + j * cl] = v; });
    wg_gemm<false, !LDSM>(N, ElP{t, ld}, ElP{c.Q, ld}, [=](int i, int j, real v) { Tpp[i + j * cl] = v; });
  } else if ((IFACE < 0 || IFACE == 1) && iface == 1) {
    copy_to_slab<LDSM>(c, g.T_mm, cl, c.P, ld);  // P = T--
    // J0- = J0- + T-- (r-+ J0+ + j0-) ; J0+ = j0+ + t++ J0+  (:36-37)
    wg_matvec(c, ElP{r, ld}, c.Jp, c.v1);
    for (int i = wg_tid(); i < N; i += kThreads) c.v1[i] = c.v1[i] + c.jm[i];
    __syncthreads();
    wg_matvec(c, ElP{c.P, ld}, c.v1, c.v2);
    wg_matvec(c, ElP{t, ld}, c.Jp, c.v1);
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.Jm[i] = c.Jm[i] + c.v2[i];
      c.Jp[i] = c.jp[i] + c.v1[i];
    }
    // R-+ = (T-- r-+) T++ ; R+- = r+- ; T++ = t++ T++ ; T-- = T-- t--   (:40-43)
    real *Q = c.Q, *P = c.P;
    wg_gemm<false, !LDSM>(N, ElP{P, ld}, ElP{r, ld}, [=](int i, int j, real v) { Q[i + j * ld] = v; });
    __syncthreads();
    gdouble *Rmp = g.R_mp, *Rpm = g.R_pm, *Tpp = g.T_pp, *Tmm = g.T_mm;
    wg_gemm<false, !LDSM>(N, ElP{Q, ld}, El{g.T_pp, cl, N}, [=](int i, int j, real v) { Rmp[i + j * cl] = v; });
    __syncthreads();
    copy_to_slab<LDSM>(c, g.T_pp, cl, Q, ld);
    __syncthreads();
    wg_gemm<false, !LDSM>(N, ElP{t, ld}, ElP{Q, ld}, [=](int i, int j, real v) { Tpp[i + j * cl] = v; });
    wg_gemm<false, !LDSM>(N, ElP{P, ld}, tmm, [=](int i, int j, real v) { Tmm[i + j * cl] = v; });
    for (int e = wg_tid(); e < N * N; e += kThreads) {
      int i, j;
      c.fd.split(e, i, j);
      Rpm[i + j * cl] = rpm(i, j);
    }
  } else if ((IFACE < 0 || IFACE == 2) && iface == 2) {
    real *P = c.P, *Q = c.Q;
    copy_to_slab<LDSM>(c, g.R_pm, cl, P, ld);  // P = R+-
    copy_to_slab<LDSM>(c, g.T_mm, cl, Q, ld);  // Q = T--
    __syncthreads();
    // J0+ = j0+ + t++ (J0+ + R+- j0-) ; J0- = J0- + T-- j0-   (:58-59)
    wg_matvec(c, ElP{P, ld}, c.jm, c.v1);
    for (int i = wg_tid(); i < N; i += kThreads) c.v1[i] = c.Jp[i] + c.v1[i];
    __syncthreads();
    wg_matvec(c, ElP{t, ld}, c.v1, c.v2);
    wg_matvec(c, ElP{Q, ld}, c.jm, c.v1);
    for (int i = wg_tid(); i < N; i += kThreads) {
      c.Jp[i] = c.jp[i] + c.v2[i];
      c.Jm[i] = c.Jm[i] + c.v1[i];
    }
    // T++ = t++ T++ ; T-- = T-- t-- ; R+- = (t++ R+-) t--       (:62-64)
    gdouble *Tpp = g.T_pp, *Tmm = g.T_mm, *Rpm = g.R_pm;
    wg_gemm<false, !LDSM>(N, ElP{Q, ld}, tmm, [=](int i, int j, real v) { Tmm[i + j * cl] = v; });
    __syncthreads();
    copy_to_slab<LDSM>(c, g.T_pp, cl, Q, ld);
    __syncthreads();
    wg_gemm<false, !LDSM>(N, ElP{t, ld}, ElP{Q, ld}, [=](int i, int j, real v) { Tpp[i + j * cl] = v; });
    __syncthreads();
    wg_gemm<false, !LDSM>(N, ElP{t, ld}, ElP{P, ld}, [=](int i, int j, real v) { Q[i + j * ld] = v; });
    __syncthreads();
    wg_gemm<false, !LDSM>(N, ElP{Q, ld}, tmm, [=](int i, int j, real v) { Rpm[i + j * cl] = v; });
  } else if ((IFACE < 0 || IFACE == 3) && iface == 3) {
    // ---- ScatteringInterface_11 (interaction.jl:69-117)
    // The four mat-vec products ride as column N of the B operands when the buffers have a spare
    // column in the last tile and no K padding (see doubling_run).
    const bool ride = (N % 4 == 0) && (c.nc - N >= 1);
    MOM_STAMP(10);
    copy_to_slab<LDSM>(c, g.R_pm, cl, c.P, ld);  // P = R+-
    if (ride)
      for (int i = wg_tid(); i < N; i += kThreads) c.P[i + N * ld] = c.Jp[i];
    __syncthreads();
    MOM_STAMP(11);
    real beta2;
    {
      real *P = c.P, *Q = c.Q;
      real ss = 0.0;
      // Q = r-+ R+- (+ r-+ J0+) ;  P = T01 = T-- (I - r-+ R+-)^-1            (:81-87)
      wg_gemm_nc<false, !LDSM>(N, ride ? N + 1 : N, ElP{r, ld}, ElP{P, ld}, [=, &ss](int i, int j, real v) {
        Q[i + j * ld] = v;
        if (j < N) ss += v * v;
      });
      wg_sumsq_put(c, ss);
      __syncthreads();
      beta2 = wg_sumsq_get(c);
      if (ride)  // v1 = r-+ J0+ + j0-  -> column N of r (B operand of T01 r-+ below)
        for (int i = wg_tid(); i < N; i += kThreads) r[i + N * ld] = Q[i + N * ld] + c.jm[i];
    }
    MOM_STAMP(12);
    times_inv<LDSM>(c, El{g.T_mm, cl, N}, c.Q, c.P, beta2);
    MOM_STAMP(13);
    if (!ride) {
      // J0- = J0- + T01 (r-+ J0+ + j0-)                          (:90)
      if constexpr (LDSM) wg_matvec(c, ElP{r, ld}, c.Jp, c.v1);
      else wg_matvec_slab<1>(c, r, c.Jp, nullptr, c.v1, nullptr);
      for (int i = wg_tid(); i < N; i += kThreads) c.v1[i] = c.v1[i] + c.jm[i];
      __syncthreads();
      if constexpr (LDSM) wg_matvec(c, ElP{c.P, ld}, c.v1, c.v2);
      else wg_matvec_slab<1>(c, c.P, c.v1, nullptr, c.v2, nullptr);
      for (int i = wg_tid(); i < N; i += kThreads) c.Jm[i] = c.Jm[i] + c.v2[i];
    }
    MOM_STAMP(14);
    {
      real *P = c.P, *Q = c.Q;
      gdouble *Tmm = g.T_mm, *Rmp = g.R_mp;
      // T-- = T01 t--                                           (:96)
      wg_gemm<false, !LDSM>(N, ElP{P, ld}, tmm, [=](int i, int j, real v) { Tmm[i + j * cl] = v; });
      // Q = T01 r-+ (+ T01 v1)
      wg_gemm_nc<false, !LDSM>(N, ride ? N + 1 : N, ElP{P, ld}, ElP{r, ld}, [=](int i, int j, real v) { Q[i + j * ld] = v; });
      __syncthreads();
      MOM_STAMP(15);
      if (ride)  // J0- = J0- + T01 v1 (:90); next rider: j0- for R+- j0-
        for (int i = wg_tid(); i < N; i += kThreads) {
          c.Jm[i] = c.Jm[i] + Q[i + N * ld];
          r[i + N * ld] = c.jm[i];
        }
      copy_to_slab<LDSM>(c, g.T_pp, cl, P, ld);  // P = T++ (old)
      __syncthreads();
      MOM_STAMP(16);
      // R-+ = R-+ + (T01 r-+) T++                              (:93)
      wg_gemm<false, !LDSM>(N, ElP{Q, ld}, ElP{P, ld}, [=](int i, int j, real v) { Rmp[i + j * cl] = Rmp[i + j * cl] + v; });
      __syncthreads();
      MOM_STAMP(17);
      copy_to_slab<LDSM>(c, g.R_pm, cl, Q, ld);  // Q = R+- (old)
      __syncthreads();
      MOM_STAMP(18);
    }
    if (!ride) {
      // w = J0+ + R+- j0-  (kept in j1p; j1p/j1m are free outside doubling)
      if constexpr (LDSM) wg_matvec(c, ElP{c.Q, ld}, c.jm, c.v1);
      else wg_matvec_slab<1>(c, c.Q, c.jm, nullptr, c.v1, nullptr);
      for (int i = wg_tid(); i < N; i += kThreads) c.j1p[i] = c.Jp[i] + c.v1[i];
    }
    {
      real *P = c.P, *Q = c.Q;
      real ss = 0.0;
      // P = R+- r-+ (+ R+- j0-) ; Q = T21 = t++ (I - R+- r-+)^-1            (:104-107)
      wg_gemm_nc<false, !LDSM>(N, ride ? N + 1 : N, ElP{Q, ld}, ElP{r, ld}, [=, &ss](int i, int j, real v) {
        P[i + j * ld] = v;
        if (j < N) ss += v * v;
      });
      wg_sumsq_put(c, ss);
      __syncthreads();
      beta2 = wg_sumsq_get(c);
      if (ride)
        for (int i = wg_tid(); i < N; i += kThreads) c.j1p[i] = c.Jp[i] + P[i + N * ld];
    }
    MOM_STAMP(19);
    times_inv<LDSM>(c, ElP{t, ld}, c.P, c.Q, beta2);
    MOM_STAMP(20);
    if (!ride) {
      // J0+ = j0+ + T21 (J0+ + R+- j0-)                          (:110)
      if constexpr (LDSM) wg_matvec(c, ElP{c.Q, ld}, c.j1p, c.v2);
      else wg_matvec_slab<1>(c, c.Q, c.j1p, nullptr, c.v2, nullptr);
      for (int i = wg_tid(); i < N; i += kThreads) c.Jp[i] = c.jp[i] + c.v2[i];
    }
    MOM_STAMP(21);
    copy_to_slab<LDSM>(c, g.R_pm, cl, c.P, ld);  // P = R+- (old)
    if (ride)
      for (int i = wg_tid(); i < N; i += kThreads) c.P[i + N * ld] = c.j1p[i];
    __syncthreads();
    MOM_STAMP(22);
    {
      // P = T21 R+- (+ T21 w)
      real *d = c.P, *Qb = c.Q;
      if (LDSM) {
        wg_gemm_nc<true>(N, ride ? N + 1 : N, ElP{Qb, ld}, ElP{d, ld}, [=](int i, int j, real v) { d[i + j * ld] = v; });
      } else {
        real *sp = c.X;
        wg_gemm_nc<false, !LDSM>(N, ride ? N + 1 : N, ElP{Qb, ld}, ElP{d, ld}, [=](int i, int j, real v) { sp[i + j * ld] = v; });
        c.X = d;
        c.P = sp;
      }
      __syncthreads();
    }
    MOM_STAMP(23);
    if (ride)  // J0+ = j0+ + T21 w (:110)
      for (int i = wg_tid(); i < N; i += kThreads) c.Jp[i] = c.jp[i] + c.P[i + N * ld];
    {
      real *P = c.P, *Q = c.Q;
      gdouble *Rpm = g.R_pm, *Tpp = g.T_pp;
      // R+- = r+- + (T21 R+-) t--                              (:116)
      wg_gemm<false, !LDSM>(N, ElP{P, ld}, tmm, [=](int i, int j, real v) { Rpm[i + j * cl] = rpm(i, j) + v; });
      __syncthreads();
      MOM_STAMP(24);
      if (ride)
        for (int i = wg_tid(); i < N; i += kThreads) { P[i + N * ld] = 0.0; r[i + N * ld] = 0.0; Q[i + N * ld] = 0.0; }
      copy_to_slab<LDSM>(c, g.T_pp, cl, P, ld);  // P = T++ (old)
      __syncthreads();
      MOM_STAMP(25);
      // T++ = T21 T++                                          (:113)
      wg_gemm<false, !LDSM>(N, ElP{Q, ld}, ElP{P, ld}, [=](int i, int j, real v) { Tpp[i + j * cl] = v; });
    }
  }
  __syncthreads();
  MOM_STAMP(26);
  for (int i = wg_tid(); i < N; i += kThreads) {
    g.J0p[i] = c.Jp[i];
    g.J0m[i] = c.Jm[i];
  }
  __syncthreads();
}

// composite_dst <- composite_src for this workgroup's unit (multi-sensor: the running top slab frozen at a sensor level)
__device__ __forceinline__ void copy_composite(const Ctx &c, const CompPtrs &s, const CompPtrs &d) {
  const int N = c.N;
  for (int e = wg_tid(); e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    const int o = i + j * s.ld;
    d.R_mp[o] = s.R_mp[o];
    d.R_pm[o] = s.R_pm[o];
    d.T_pp[o] = s.T_pp[o];
    d.T_mm[o] = s.T_mm[o];
  }
  for (int i = wg_tid(); i < N; i += kThreads) {
    d.J0p[i] = s.J0p[i];
    d.J0m[i] = s.J0m[i];
  }
}

// composite <- added (rt_kernel.jl:227-230) from the context
__device__ __forceinline__ void store_added_as_composite(const Ctx &c, const CompPtrs &g) {
  const int N = c.N, ld = c.ld;
  for (int e = wg_tid(); e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    const real rv = c.r[i + j * ld], tv = c.t[i + j * ld], s = c.sg[i] * c.sg[j];
    const int o = i + j * g.ld;
    __builtin_nontemporal_store(rv, g.R_mp + o);  // streaming: see MOM_NT_STORE in mom_strip.hpp
    __builtin_nontemporal_store(s * rv, g.R_pm + o);
    __builtin_nontemporal_store(tv, g.T_pp + o);
    __builtin_nontemporal_store(s * tv, g.T_mm + o);
  }
  for (int i = wg_tid(); i < N; i += kThreads) {
    g.J0p[i] = c.jp[i];
    g.J0m[i] = c.jm[i];
  }
}

}  // namespace MOM_NS
