// mom_wave.hip -- operators of edge 4 < N <= 32: ONE SPECTRAL POINT PER WAVEFRONT, all operators in registers, every
// product on the FP64 matrix cores, no barrier.
//
// A 16 x 16 tile in the C/D layout of v_mfma_f64_16x16x4_f64 (lane l, register r: row (l >> 4) + 4 r, column l & 15) is
// four doubles per lane.  Feeding two such tiles U, V to the four k-steps of one 16 x 16 x 16 product -- register s of U
// as the A operand, register s of V as the B operand -- gives
//        TN(U, V) = U^T V        in the same layout,
// because the A operand is read as A[row = l & 15][k = l >> 4]: a tile in C-layout IS the A operand of its transpose
// (mom_strip.hpp uses the B-operand half of this observation).  An operator of edge N <= 16 NT is NT x NT such tiles; a
// product X Y is TN(X^T, Y), and X^T is one pass of the tiles through a wave-private LDS slice (4 ds_write + 4 ds_read
// per tile, about a sixth of a product at NT = 2).  So a wave runs the whole adding algorithm -- elemental, doublings,
// interaction, surface, post-processing, for all Fourier moments and layers of its spectral point -- as a sequence of
// register products with the reference's own operation count (doubling step: r r, the series, t G, r t, A (r t), A t;
// interaction: 10 products + 2 inverses), the source vectors travelling as columns 0 (J+) and 1 (J-) of a tile column.
// LDS only serves the transposes, the rare pivoted inverse (series too long) and the final gather of the view rows.
// A lane-per-point layout (mom_small.hip) stops at N = 4; the workgroup-per-unit kernels (mom_kernels.hpp) are
// barrier-bound below N ~ 32 (a 32^3 product is 0.2 us of MFMA per wave) and use a fraction of their tiles.
//
// NT = 1 (N <= 16): two waves per SIMD; NT = 2 (N <= 32): one wave per SIMD (about 400 VGPRs).  k-steps are templated
// on KS = ceil(N / 4): rows >= 4 KS of every tile are zero padding and are skipped.
// Scope: ScatteringInterface_11 on every layer after the first (the host falls back to the general kernels otherwise),
// all surface kinds of mom_scene_set_surface, up to 256 (view, Stokes) outputs per point, Float64.  Reference semantics and file:line as in mom_kernels.hpp / mom_small.hip.
#include <hip/hip_runtime.h>

#include "mom_host.hpp"

// The same source builds the Float32 kernels (-DMOMW_FLOAT: namespace momwf, v_mfma_f32_16x16x4, MomWaveSweepArgsF).  The f32
// accumulator layout puts row 4 lq + r (not lq + 4 r) into register r; tiles stay valid MFMA operands of U^T V because both
// operands map the contraction index the same way, but every k-step then touches rows below N, so none is skipped.
#ifdef MOMW_FLOAT
#define MOMW_NS momwf
typedef float real;
#else
#define MOMW_NS momw
typedef double real;
#endif

namespace MOMW_NS {

typedef real r4 __attribute__((ext_vector_type(4)));
#ifdef MOMW_FLOAT
constexpr bool kF32 = true;
__device__ __forceinline__ r4 mma(real a, real b, r4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int absbits(real v) { return __float_as_int(fabsf(v)); }
#else
constexpr bool kF32 = false;
__device__ __forceinline__ r4 mma(real a, real b, r4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int absbits(real v) { return __double2hiint(fabs(v)); }
#endif
// row of accumulator register r within a 16 x 16 tile
__device__ __forceinline__ constexpr int crow(int lq, int r) { return kF32 ? 4 * lq + r : lq + 4 * r; }

template <int NT>
struct Mat {
  r4 t[NT][NT];  // t[bi][bj]: rows 16 bi .., columns 16 bj ..
};
template <int NT>
struct Vec {
  r4 t[NT];  // row block bi; column 0 = the "+" vector, column 1 = the "-" vector (lanes l & 15 == 0 / 1)
};

// k-steps of row block tk
template <int KS>
__device__ __forceinline__ constexpr int ksteps(int tk) { return kF32 ? 4 : ((KS - 4 * tk) >= 4 ? 4 : ((KS - 4 * tk) > 0 ? (KS - 4 * tk) : 0)); }

template <int NT, int KS>
__device__ __forceinline__ Mat<NT> TNacc(const Mat<NT> &U, const Mat<NT> &V, Mat<NT> acc) {  // acc + U^T V
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < ksteps<KS>(tk)) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int tj = 0; tj < NT; ++tj)
            acc.t[ti][tj] = mma(U.t[tk][ti][s], V.t[tk][tj][s], acc.t[ti][tj]);
      }
  return acc;
}
template <int NT>
__device__ __forceinline__ Mat<NT> zeros() {
  Mat<NT> Z;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) Z.t[a][b] = (r4){0.0, 0.0, 0.0, 0.0};
  return Z;
}
template <int NT, int KS>
__device__ __forceinline__ Mat<NT> TN(const Mat<NT> &U, const Mat<NT> &V) {
  return TNacc<NT, KS>(U, V, zeros<NT>());
}
template <int NT, int KS>
__device__ __forceinline__ Vec<NT> TNv(const Mat<NT> &U, const Vec<NT> &v) {  // U^T v
  Vec<NT> o;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) o.t[ti] = (r4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < ksteps<KS>(tk)) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
          o.t[ti] = mma(U.t[tk][ti][s], v.t[tk][s], o.t[ti]);
      }
  return o;
}

// Cross-lane moves without the LDS crossbar (ds_bpermute behind __shfl_xor: ~100 cycles each in a dependent chain):
// DPP within a row of 16 lanes, gfx950's permlane swaps across rows / halves.
template <int CTRL>
__device__ __forceinline__ real dpp_mov(real v) {
#ifdef MOMW_FLOAT
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
#else
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#endif
}
__device__ __forceinline__ real swap_rows16(real v) {  // lane ^ 16
#ifdef MOMW_FLOAT
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(((threadIdx.x >> 4) & 1) ? a[0] : a[1]);
#else
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const bool odd = (threadIdx.x >> 4) & 1;
  return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
#endif
}
__device__ __forceinline__ real swap_halves32(real v) {  // lane ^ 32
#ifdef MOMW_FLOAT
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(((threadIdx.x >> 5) & 1) ? a[0] : a[1]);
#else
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const bool up = (threadIdx.x >> 5) & 1;
  return __hiloint2double(up ? b[0] : b[1], up ? a[0] : a[1]);
#endif
}
// (the Float64 2 x 2-tile images keep the __shfl_xor forms: with the DPP ones the compiler's AGPR rewrite pass crashes
// on k_wsweep<2, 7> under -amdgpu-mfma-vgpr-form, as it does on k_wsweep<2, 8> anyway)
template <int NT>
__device__ __forceinline__ real wave_sum(real v) {
  if constexpr (NT > 1 && !kF32) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
  }
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_mov<0x124>(v);  // row_ror 4
  v += dpp_mov<0x128>(v);  // row_ror 8: every lane of a row now holds the row's sum
  v += swap_rows16(v);
  v += swap_halves32(v);
  return v;
}
template <int NT>
__device__ __forceinline__ Vec<NT> swap01(const Vec<NT> &v) {  // exchange columns 0 <-> 1
  Vec<NT> o;
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[b][r] = (NT > 1 && !kF32) ? __shfl_xor(v.t[b][r], 1) : dpp_mov<0xB1>(v.t[b][r]);  // lane ^ 1
  return o;
}

// series length for (I - B)^-1 from beta^2 = ||B||_F^2: the rule of mom_kernels.hpp (kNeumannThr2: tail <= 2^-56, the
// first-order term always kept); 1000 = use the pivoted inverse
#ifdef MOMW_FLOAT
// Float32: beta^p / (1 - beta) <= 2^-26 (mom_kernels.hpp, MOM_REAL_IS_FLOAT)
__device__ const real kThr2[32] = {
    0.000000000e+00f, 1.489934232e-08f, 6.045524421e-06f, 1.213959655e-04f,
    7.320204349e-04f, 2.419753539e-03f, 5.676200029e-03f, 1.075029447e-02f,
    1.765854019e-02f, 2.625958398e-02f, 3.632882835e-02f, 4.761229038e-02f,
    5.985942527e-02f, 7.284045577e-02f, 8.635379794e-02f, 1.002277241e-01f,
    1.143189633e-01f, 1.285098733e-01f, 1.427051184e-01f, 1.568283537e-01f,
    1.708191621e-01f, 1.846303449e-01f, 1.982255889e-01f, 2.115774909e-01f,
    2.246659062e-01f, 2.374765770e-01f, 2.500000000e-01f, 2.622304965e-01f,
    2.741654486e-01f, 2.858046753e-01f, 2.971499218e-01f, 3.082044441e-01f};
#else
__device__ const real kThr2[32] = {
    0.0, 1.38777877561156685e-17, 5.77492213356056750e-12, 3.72517661162420568e-09,
    1.80656771560518035e-07, 2.40186660760962690e-06, 1.52417448931310540e-05, 6.09157135028591602e-05,
    1.78874927371965362e-04, 4.23309807394842467e-04, 8.56292603484697687e-04, 1.53988783074545245e-03,
    2.52966413875025916e-03, 3.87056942792606993e-03, 5.59509450064154569e-03, 7.72318484591632843e-03,
    1.02632763666295982e-02, 1.32139270473634555e-02, 1.65656642989196294e-02, 2.03028052759899880e-02,
    2.44051136922726897e-02, 2.88492300492497432e-02, 3.36098586497813809e-02, 3.86607216385354419e-02,
    4.39753040322414940e-02, 4.95274191527264318e-02, 5.52916244610413068e-02, 6.12435157546498479e-02,
    6.73599244364155580e-02, 7.36190389296255826e-02, 8.00004677634075928e-02, 8.64852586225294262e-02};
#endif
// thr = kThr2[lane & 31], held in a register for the whole kernel: ONE compare + ballot instead of 31 compares with
// 64-bit literals (a product of two one-tile operators is only four MFMAs: the scan cost as much as a product)
__device__ __forceinline__ int series_terms(real beta2, real thr) {
  const unsigned m = (unsigned)__ballot(beta2 > thr);  // lanes 0..31
  if (!(beta2 <= beta2) || (m >> 31)) return 1000;     // NaN, or beyond the table
  return 1 + __popc(m & 0x7fffffffu);
}

using WArgs = ::MomWaveSweepArgsT<real>;  // mom_host.hpp: the one definition shared with momcore.hip / momcore_f32.hip

// per-lane coordinates; row / column quantities are read from the block's LDS table tab = mu[32] | wt[32] | sg[32]
// (padding entries: mu = 1, wt = 0, sg = 1)
struct Lay {
  int lr, lq, N, nS;
  const real *tab;
  real thr;    // kThr2[lane & 31] (series_terms)
  real *xp;  // wave-private LDS slice: transposes / pivoted inverse / output gather
  int *ipiv;
  __device__ __forceinline__ int row(int bi, int r) const { return 16 * bi + crow(lq, r); }
  __device__ __forceinline__ int col(int bj) const { return 16 * bj + lr; }
  __device__ __forceinline__ real mu(int i) const { return tab[i]; }
  __device__ __forceinline__ real wt(int i) const { return tab[32 + i]; }
  __device__ __forceinline__ real sg(int i) const { return tab[64 + i]; }
};

template <int NT>
__device__ __forceinline__ Mat<NT> ident(const Lay &L) {
  Mat<NT> I;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) I.t[a][b][r] = (a == b && crow(L.lq, r) == L.lr && L.col(b) < L.N) ? 1.0 : 0.0;
  return I;
}
template <int NT>
__device__ __forceinline__ Mat<NT> add(const Mat<NT> &A, const Mat<NT> &B) {
  Mat<NT> C;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) C.t[a][b] = A.t[a][b] + B.t[a][b];
  return C;
}
template <int NT>
__device__ __forceinline__ Mat<NT> scaleD(const Lay &L, const Mat<NT> &X) {  // D X D, D = Diagonal(sg)
  Mat<NT> Y;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y.t[a][b][r] = (L.sg(L.row(a, r)) * L.sg(L.col(b))) * X.t[a][b][r];
  return Y;
}

constexpr int kTileLd = 17;                   // pitch of a 16 x 16 tile in the LDS slice
constexpr int kTileDoubles = 16 * kTileLd;    // 272
template <int NT>
constexpr int slice_doubles() {
  return (NT * NT * kTileDoubles > (16 * NT) * (16 * NT + 1) ? NT * NT * kTileDoubles : (16 * NT) * (16 * NT + 1)) + 3 * 16 * NT;
}

// X^T through the wave's LDS slice: tile (a, b) of the result is the transpose of tile (b, a)
template <int NT>
__device__ __forceinline__ Mat<NT> transpose(const Lay &L, const Mat<NT> &X) {
  real *buf = L.xp;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf[(a * NT + b) * kTileDoubles + crow(L.lq, r) * kTileLd + L.lr] = X.t[a][b][r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Mat<NT> Y;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y.t[a][b][r] = buf[(b * NT + a) * kTileDoubles + L.lr * kTileLd + crow(L.lq, r)];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return Y;
}

// (I - B)^-1 by Gauss-Jordan elimination with implicit partial pivoting, one matrix row per lane through the wave's LDS
// slice (the register-resident scheme of wg_inverse_reg, mom_device.hpp, for a single wave)
template <int NT>
__device__ __noinline__ Mat<NT> inverse_gj(Mat<NT> B, int N, real *lds, int *ipiv, int *bad_out) {
  constexpr int NP = 16 * NT, LDM = NP + 1;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  int bad = 0;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * a + crow(lq, r), j = 16 * b + lr;
        lds[i * LDM + j] = ((i == j) ? 1.0 : 0.0) - B.t[a][b][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  real v[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) v[c] = (lane < N && c < N) ? lds[lane * LDM + c] : ((lane == c) ? 1.0 : 0.0);
  bool used = false;
  int myk = lane;  // rows >= N keep their identity row
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (k < N) {
      const int ah = (!used && lane < N) ? absbits(v[k]) : -1;
      int mh = ah;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mh = max(mh, __shfl_xor(mh, off));
      const unsigned long long mk = __ballot(ah == mh);
      const int pl = __ffsll((long long)mk) - 1;
      const real piv = __shfl(v[k], pl);
      if (!(fabs(piv) > 0.0) && !bad) bad = k + 1;
      const real d = 1.0 / piv, f = v[k];
      const bool isp = (lane == pl);
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        const real prow = __shfl(v[c], pl) * d;
        v[c] = isp ? prow : (v[c] - f * prow);
      }
      v[k] = isp ? d : (-f * d);
      if (isp) { used = true; myk = k; }
      if (lane == 0) ipiv[k] = pl;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // inv(A)[k][p_j] = S[p_k][j]: lane (row p_k, pivot of step myk) writes row myk with permuted columns
  if (lane < N) {
#pragma unroll
    for (int c = 0; c < NP; ++c)
      if (c < N) lds[myk * LDM + ipiv[c]] = v[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Mat<NT> G;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * a + crow(lq, r), j = 16 * b + lr;
        G.t[a][b][r] = (i < N && j < N) ? lds[i * LDM + j] : 0.0;
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (bad) *bad_out = bad;
  return G;
}

// (I - B)^-1: truncated Neumann series by Horner, G <- I + B G = I + TN(B^T, G); beyond 32 terms (or MOM_OPT_INVERSE = 1)
// the pivoted inverse
template <int NT, int KS>
__device__ __forceinline__ Mat<NT> inv_one_minus(const Lay &L, const Mat<NT> &B, int inv_mode, int &bad) {
  real ss = 0.0;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) ss += B.t[a][b][r] * B.t[a][b][r];
  const real beta2 = wave_sum<NT>(ss);
  const int p = (inv_mode == 1) ? 1000 : series_terms(beta2, L.thr);
  const Mat<NT> I = ident<NT>(L);
  if (p <= 32) {
    if (p == 1) return I;
    Mat<NT> G = add<NT>(I, B);
    if (p > 2) {
      const Mat<NT> Bt = transpose<NT>(L, B);
      for (int k = 2; k < p; ++k) G = TNacc<NT, KS>(Bt, G, I);
    }
    return G;
  }
  int b = 0;
  const Mat<NT> G = inverse_gj<NT>(B, L.N, L.xp, L.ipiv, &b);
  if (b && !bad) bad = b;
  return G;
}

// composite state in natural (c-) form; Jv: column 0 J0+, column 1 J0-
template <int NT>
struct Comp {
  Mat<NT> Rmp, Rpm, Tpp, Tmm;
  Vec<NT> Jv;
};

// ScatteringInterface_11 (interaction.jl:69-117).  Added layer: r-+ (r), t++ (t), sources jv; r+- = D r-+ D and
// t-- = D t++ D are formed where they are used (SURF: the surface layer has r+- = 0, t = I).  Returns r^T (the surface
// caller needs it for interaction_hdrf!).
template <int NT, int KS, bool SURF>
__device__ __forceinline__ Mat<NT> interact11(const Lay &L, Comp<NT> &C, const Mat<NT> &r, const Mat<NT> &t, const Vec<NT> &jv,
                                              int inv_mode, int &bad) {
  const Mat<NT> rT = transpose<NT>(L, r);
  // --- T01 = T-- (I - r-+ R+-)^-1                                                   (:81-87)
  const Mat<NT> G1 = inv_one_minus<NT, KS>(L, TN<NT, KS>(rT, C.Rpm), inv_mode, bad);
  const Mat<NT> T01t = TN<NT, KS>(G1, transpose<NT>(L, C.Tmm));  // (T-- G1)^T = G1^T T--^T
  // J0- = J0- + T01 (r-+ J0+ + j0-)                                                  (:90)  [old J0+]
  const Vec<NT> V1 = TNv<NT, KS>(rT, C.Jv);  // column 0: r-+ J0+
  const Vec<NT> jsw = swap01<NT>(jv);        // column 0: j0-
  Vec<NT> X1;
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) X1.t[b][q] = (L.lr == 0) ? V1.t[b][q] + jsw.t[b][q] : 0.0;
  const Vec<NT> TX1s = swap01<NT>(TNv<NT, KS>(T01t, X1));  // column 1: T01 (...)
  // R-+ = R-+ + T01 r-+ T++                                                          (:93)
  C.Rmp = TNacc<NT, KS>(T01t, TN<NT, KS>(rT, C.Tpp), C.Rmp);
  // T-- = T01 t--                                                                    (:96)
  const Mat<NT> tmm = SURF ? ident<NT>(L) : scaleD<NT>(L, t);
  C.Tmm = TN<NT, KS>(T01t, tmm);
  // --- T21 = t++ (I - R+- r-+)^-1                                                   (:104-107)  [old R+-]
  const Mat<NT> RpmT = transpose<NT>(L, C.Rpm);
  const Mat<NT> G2 = inv_one_minus<NT, KS>(L, TN<NT, KS>(RpmT, r), inv_mode, bad);
  const Mat<NT> T21t = TN<NT, KS>(G2, SURF ? ident<NT>(L) : transpose<NT>(L, t));
  // J0+ = j0+ + T21 (J0+ + R+- j0-)                                                  (:110)
  const Vec<NT> V2s = swap01<NT>(TNv<NT, KS>(RpmT, jv));  // column 0: R+- j0-
  Vec<NT> X2;
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) X2.t[b][q] = (L.lr == 0) ? C.Jv.t[b][q] + V2s.t[b][q] : 0.0;
  const Vec<NT> TX2 = TNv<NT, KS>(T21t, X2);  // column 0: T21 (...)
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      C.Jv.t[b][q] = (L.lr == 0) ? jv.t[b][q] + TX2.t[b][q] : ((L.lr == 1) ? C.Jv.t[b][q] + TX1s.t[b][q] : 0.0);
  // T++ = T21 T++                                                                    (:113)
  C.Tpp = TN<NT, KS>(T21t, C.Tpp);
  // R+- = r+- + T21 R+- t--                                                          (:116)
  if (SURF) C.Rpm = TN<NT, KS>(T21t, C.Rpm);
  else C.Rpm = TNacc<NT, KS>(T21t, TN<NT, KS>(RpmT, tmm), scaleD<NT>(L, r));
  return rT;
}

#ifndef MOMW_OCC1
#define MOMW_OCC1 2  // waves per SIMD of the NT = 1 images
#endif

// PK > 1 (NT = 1 only): PK spectral points per wavefront, their operators as the diagonal blocks of ONE 16 x 16 tile (block b
// = rows / columns [b Nb, (b + 1) Nb) of the packed edge N = PK Nb <= 16).  U^T V of block-diagonal tiles is block-diagonal
// and a packed vector is the concatenation of the points' vectors, so every product, transpose, series and pivoted inverse
// of the sweep runs unchanged on the packed operator; only the elemental layer, the surface layer and the outputs know
// about the blocks (per-block tau, varpi, weights, expk; entries outside the diagonal blocks are exact zeros).  The
// series length comes from the Frobenius norm of the whole tile (>= every block's: never fewer terms than a point alone
// would take).  N = 5: three points per wave, N = 6..8: two -- the tile of a single point is at most a quarter full there.
template <int PK>
struct Blk {
  real v[PK];
  __device__ __forceinline__ real at(int b) const {
    if constexpr (PK == 1) return v[0];
    else if constexpr (PK == 2) return b == 0 ? v[0] : v[1];
    else return b == 0 ? v[0] : (b == 1 ? v[1] : v[2]);
  }
};

template <int NT, int KS, int PK = 1>
__global__ void __launch_bounds__(256, (NT == 1 ? MOMW_OCC1 : 1)) k_wsweep(WArgs a) {
  static_assert(PK == 1 || NT == 1, "packing is for one-tile operators");
  __shared__ real s_lds[4][slice_doubles<NT>()];
  __shared__ int s_piv[4][16 * NT];
  __shared__ real s_tab[96];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * PK;  // first spectral point of this wave
  const int Nb = a.N, N = PK * Nb, nS = a.nS, S = a.S, K = a.K;  // Nb: edge of one point's operators; N: packed edge
  if (threadIdx.x < 32) {
    const int i = threadIdx.x, il = i % Nb;
    s_tab[i] = i < N ? a.mu[il] : 1.0;
    s_tab[32 + i] = i < N ? a.wt[il] : 0.0;
    s_tab[64 + i] = i < N ? a.sg[il] : 1.0;
  }
  __syncthreads();
  if (n0 >= S) return;
  Lay L;
  L.lr = lane & 15; L.lq = lane >> 4; L.N = N; L.nS = nS; L.tab = s_tab; L.xp = s_lds[wave]; L.ipiv = s_piv[wave]; L.thr = kThr2[lane & 31];
  real *post = L.xp + slice_doubles<NT>() - 3 * 16 * NT;  // J0+ | J0- | hdr_J0-, 16 NT each
  const int i_start = nS * (a.imu0 - 1), i_end = nS * a.imu0;
  const real mus = a.mu[i_start];
  int bad = 0;
  // block coordinates of this lane: column block cb / local column jl, and per accumulator register q the row block
  // rb[q] / local row il[q] (PK = 1: block 0, local = global)
  int cb = 0, jl = L.lr, rb[4 * NT], il[4 * NT], np[PK];
#pragma unroll
  for (int b = 0; b < PK; ++b) np[b] = min(n0 + b, S - 1);  // tail: the last wave repeats a point, its outputs are not stored
  if constexpr (PK > 1) {
    cb = min(L.lr / Nb, PK - 1);
    jl = L.lr - cb * Nb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = L.row(0, q);
      rb[q] = min(i / Nb, PK - 1);
      il[q] = i - rb[q] * Nb;
    }
  }
  // outputs: lane x (+ 64 per pass) handles (view v = x / nS, component k = x % nS); the sums over the Fourier moments
  // are kept in the output arrays themselves (one read-modify-write per moment: no registers held across the sweep)
  const int nout = a.nVza * nS;

  for (int m = 0; m < a.M; ++m) {
    const real wdiv = (m == 0) ? 2.0 : 4.0, wct02 = (m == 0) ? 0.5 : 0.25;
    const real *Zp_m = a.Zpp + (size_t)Nb * Nb * K * m, *Zm_m = a.Zmp + (size_t)Nb * Nb * K * m;
    Comp<NT> C;
    for (int z = 0; z < a.Nz; ++z) {
      const int nd = a.nd[z];
      Blk<PK> tau_b, varpi_b, dtau_b, expk_b, att_b, es_b;
      size_t ob[PK];
#pragma unroll
      for (int b = 0; b < PK; ++b) {
        ob[b] = np[b] + (size_t)S * z;
        tau_b.v[b] = a.tau[ob[b]];
        varpi_b.v[b] = a.varpi[ob[b]];
        dtau_b.v[b] = ldexp(tau_b.v[b], -nd);
        expk_b.v[b] = exp(-dtau_b.v[b] / a.mu0);
        att_b.v[b] = exp(-a.tau_sum[ob[b]] / mus);
        es_b.v[b] = exp(-dtau_b.v[b] / mus);
      }
      // ------------------------------------------------ elemental! (elemental.jl:164-253)
      Mat<NT> r, t;
      Vec<NT> jv;
      {
        // matrix entries belong to the lane's COLUMN block (a diagonal-block entry has row block == column block)
        const real dtau = dtau_b.at(cb), varpi = varpi_b.at(cb);
        const size_t o = (PK == 1) ? ob[0] : (cb == 0 ? ob[0] : (cb == 1 ? ob[PK > 1 ? 1 : 0] : ob[PK - 1]));
        // one register row (of every tile of a row block) per iteration, NOT unrolled over q: the live set of one
        // iteration is what the register budget affords next to the composite tiles
#pragma unroll
        for (int bi = 0; bi < NT; ++bi) {
#pragma unroll 1
          for (int q = 0; q < 4; ++q) {
            const int ip = L.row(bi, q);                       // packed row
            const int i = (PK == 1) ? ip : il[q];              // row inside its block
            const int rbq = (PK == 1) ? 0 : rb[q];
            const bool rok = ip < N;
            const real mui = L.mu(ip), wir = L.wt(ip) / wdiv;
            const real er = exp(-dtau / mui);
#pragma unroll
            for (int bj = 0; bj < NT; ++bj) {
              const int jp_ = L.col(bj);
              const int j = (PK == 1) ? jp_ : jl;
              const bool ok = rok && jp_ < N && rbq == cb;
              const real muj = L.mu(jp_), wjc = L.wt(jp_) / wdiv;
              real zp = 0.0, zm = 0.0;
              if (ok)
                for (int k = 0; k < K; ++k) {
                  const real w = a.zw[k + (size_t)K * o];
                  const size_t b = (size_t)Nb * Nb * k + i + (size_t)Nb * j;
                  zp += w * Zp_m[b];
                  zm += w * Zm_m[b];
                }
              real rij, tij;
              if (wjc > 1.e-8) {
                rij = varpi * zm * (muj / (mui + muj)) * wjc * (1 - exp(-dtau * ((1 / mui) + (1 / muj))));
                if (mui == muj) tij = (i == j) ? er * (1 + varpi * zp * (dtau / mui) * wir) : 0.0;
                else tij = varpi * zp * (muj / (mui - muj)) * wjc * (er - exp(-dtau / muj));
              } else {
                rij = 0.0;
                tij = (i == j) ? er : 0.0;
              }
              if (nd >= 1) rij *= L.sg(ip);  // apply_D_elemental!: rows of r-+ (elemental.jl:265-269)
              r.t[bi][bj][q] = ok ? rij : 0.0;
              t.t[bi][bj][q] = ok ? tij : 0.0;
            }
            // source rows: Z I0 over the sun's Stokes block (lanes of columns 0 and 1)             (elemental.jl:224-251)
            // -- a vector entry belongs to its ROW block
            real jx = 0.0;
            if (rok && L.lr < 2) {
              const real dtv = dtau_b.at(rbq), vpv = varpi_b.at(rbq), erv = (PK == 1) ? er : exp(-dtv / mui);
              const size_t ov = (PK == 1) ? ob[0] : (rbq == 0 ? ob[0] : (rbq == 1 ? ob[PK > 1 ? 1 : 0] : ob[PK - 1]));
              const real *Zs = (L.lr == 0) ? Zp_m : Zm_m;
              real zI = 0.0;
              for (int ks = 0; ks < nS; ++ks)
                for (int k = 0; k < K; ++k)
                  zI += a.zw[k + (size_t)K * ov] * Zs[(size_t)Nb * Nb * k + i + (size_t)Nb * (i_start + ks)] * a.I0[ks];
              if (L.lr == 0) {
                if (i >= i_start && i < i_end) jx = wct02 * vpv * zI * (dtv / mui) * erv;
                else jx = wct02 * vpv * zI * (mus / (mui - mus)) * (erv - es_b.at(rbq));
                jx *= att_b.at(rbq);
              } else {
                jx = wct02 * vpv * zI * (mus / (mui + mus)) * (1 - exp(-dtv * ((1 / mui) + (1 / mus))));
                jx *= att_b.at(rbq);
                if (nd >= 1) jx = a.D[i % nS] * jx;
              }
            }
            jv.t[bi][q] = jx;
          }
        }
      }
      // ------------------------------------------------ doubling_helper! (doubling.jl:43-68)
      for (int it = 0; it < nd; ++it) {
        const Mat<NT> rT = transpose<NT>(L, r);
        const Mat<NT> G = inv_one_minus<NT, KS>(L, TN<NT, KS>(rT, r), a.inv_mode, bad);  // (I - r r)^-1      (:44-47)
        const Mat<NT> At = TN<NT, KS>(G, transpose<NT>(L, t));                            // (t G)^T           (:48)
        const Vec<NT> Us = swap01<NT>(TNv<NT, KS>(rT, jv));  // columns: r j0- | r j0+
        Vec<NT> Wv;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // column 0: w2 = j0+ + r j1-  ; column 1: w1 = j1- + r j0+   (:51-60)
            const real expk = expk_b.at((PK == 1) ? 0 : rb[q]);
            Wv.t[b][q] = (L.lr == 0) ? jv.t[b][q] + expk * Us.t[b][q] : ((L.lr == 1) ? jv.t[b][q] * expk + Us.t[b][q] : 0.0);
          }
        const Vec<NT> AW = TNv<NT, KS>(At, Wv);
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // j0+ = j1+ + A w2 (:60) ; j0- = j0- + A w1 (:57)
            const real expk = expk_b.at((PK == 1) ? 0 : rb[q]);
            jv.t[b][q] = (L.lr == 0) ? jv.t[b][q] * expk + AW.t[b][q] : ((L.lr == 1) ? jv.t[b][q] + AW.t[b][q] : 0.0);
          }
#pragma unroll
        for (int b = 0; b < PK; ++b) expk_b.v[b] = expk_b.v[b] * expk_b.v[b];   // :61
        r = TNacc<NT, KS>(At, TN<NT, KS>(rT, t), r);          // r + A (r t), old t                 (:64)
        t = TN<NT, KS>(At, t);                                // A t                               (:67)
      }
      if (nd >= 1) {  // apply_D! / apply_D_SFI! (doubling.jl:93-118): rows of r-+ and j0- scaled by sg
#pragma unroll
        for (int bi = 0; bi < NT; ++bi)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const real s = L.sg(L.row(bi, q));
#pragma unroll
            for (int bj = 0; bj < NT; ++bj) r.t[bi][bj][q] *= s;
            if (L.lr == 1) jv.t[bi][q] *= s;
          }
      }
      // ------------------------------------------------ composite <- added (rt_kernel.jl:227-230) or interaction!
      if (z == 0) {
        C.Rmp = r; C.Rpm = scaleD<NT>(L, r); C.Tpp = t; C.Tmm = scaleD<NT>(L, t); C.Jv = jv;
      } else {
        (void)interact11<NT, KS, false>(L, C, r, t, jv, a.inv_mode, bad);
      }
    }
    // ---------------------------------------------------- Lambertian surface (m = 0) + closing interaction (Q6)
    Vec<NT> hdrJ;
#pragma unroll
    for (int b = 0; b < NT; ++b) hdrJ.t[b] = (r4){0.0, 0.0, 0.0, 0.0};
    if (m == 0 || a.surf_kind == 1) {
      Blk<PK> rho_b, att_b;
#pragma unroll
      for (int b = 0; b < PK; ++b) {
        rho_b.v[b] = 2 * ((a.surf_kind == 2) ? a.albedo_spec[np[b]] : a.albedo);  // lambertian_surface.jl:37 / :97
        att_b.v[b] = exp(-a.tau_sum[np[b] + (size_t)S * a.Nz] / a.mu0);
      }
      const real *Rs = a.Rsurf + (size_t)Nb * Nb * m;  // kind 1: rho_m [N,N] (rpv_surface.jl:39-43)
      Mat<NT> rs;
      Vec<NT> jv;
#pragma unroll
      for (int bi = 0; bi < NT; ++bi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ip = L.row(bi, q);
          const int i = (PK == 1) ? ip : il[q];
          const int rbq = (PK == 1) ? 0 : rb[q];
          const real rho = rho_b.at(rbq), att = att_b.at(rbq);
#pragma unroll
          for (int bj = 0; bj < NT; ++bj) {
            const int jp_ = L.col(bj);
            const int j = (PK == 1) ? jp_ : jl;
            const bool in = ip < N && jp_ < N && rbq == cb;
            real v;
            if (a.surf_kind == 1) v = in ? Rs[i + (size_t)Nb * j] * (L.mu(jp_) * L.wt(jp_)) : 0.0;       // rpv_surface.jl:58-62
            else v = (in && (i % nS == 0) && (j % nS == 0)) ? rho * (L.mu(jp_) * L.wt(jp_)) : 0.0;      // r-+ = R_surf Diagonal(mu w)  (:41-43,:58)
            rs.t[bi][bj][q] = v;
          }
          const bool in_sun = (i >= i_start) && (i < i_end);
          real jp = (in_sun ? a.I0[i - i_start] : 0.0) * att;                           // :55
          real jm = (i % nS == 0) ? (a.mu0 * (rho * a.I0[0])) * att : 0.0;             // :56
          if (a.surf_kind == 1 && ip < N) {                                               // j0- = mu0 (R_surf I0N) e^(-tau/mu0)  (rpv_surface.jl:48-56)
            real rI = 0.0;
            for (int k = 0; k < nS; ++k) rI += Rs[i + (size_t)Nb * (i_start + k)] * a.I0[k];
            jm = (a.mu0 * rI) * att;
          }
          if (a.surf_kind == 2) {                                                         // lambertian_surface.jl:112-114
            jp = 0.0;
            jm = (i % nS == 0) ? (a.mu0 * a.I0[0]) * (rho * att) : 0.0;
          }
          jv.t[bi][q] = !(ip < N) ? 0.0 : (L.lr == 0 ? jp : (L.lr == 1 ? jm : 0.0));
        }
      const Mat<NT> rsT = interact11<NT, KS, true>(L, C, rs, rs, jv, a.inv_mode, bad);  // (t operand unused)
      // interaction_hdrf! (interaction_hdrf.jl:9-45): hdr_J0- = r-+_surf J0+ + j0-_surf  -> column 0
      const Vec<NT> rJ = TNv<NT, KS>(rsT, C.Jv);
      const Vec<NT> jsw = swap01<NT>(jv);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) hdrJ.t[b][q] = (L.lr == 0) ? rJ.t[b][q] + jsw.t[b][q] : 0.0;
      // BHR flux sums over the streams of each Stokes component (column-0 lanes hold hdr_J0- and J0+), m = 0 only
      for (int pb = 0; pb < PK; ++pb)
        for (int k = 0; k < (m == 0 ? nS : 0); ++k) {
          real up = 0.0, dw = 0.0;
#pragma unroll
          for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ip = L.row(b, q);
              const int i = (PK == 1) ? ip : il[q];
              const bool mine = (PK == 1) || rb[q] == pb;
              if (L.lr == 0 && ip < N && mine && (i % nS == k)) {
                up += hdrJ.t[b][q] * L.wt(ip) * L.mu(ip);
                dw += C.Jv.t[b][q] * L.wt(ip) * L.mu(ip);
              }
            }
          up = wave_sum<NT>(up);
          dw = wave_sum<NT>(dw);
          // + j0+_surf[i_start] mu[i_start]: the direct beam (interaction_hdrf.jl:30)
          const real direct = ((a.surf_kind == 2) ? 0.0 : a.I0[0] * att_b.at(pb)) * mus;  // j0+_surf[i_start] mu[i_start]
          if (lane == 0 && n0 + pb < S) {
            a.bhr_uw[k + (size_t)nS * (n0 + pb)] = up;
            a.bhr_dw[k + (size_t)nS * (n0 + pb)] = dw + direct;
          }
        }
    }
    // ---------------------------------------------------- postprocessing_vza! (+ hdrf) through the wave's LDS slice
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = L.row(b, q);
        if (L.lr == 0) { post[i] = C.Jv.t[b][q]; post[32 * NT + i] = hdrJ.t[b][q]; }
        if (L.lr == 1) post[16 * NT + i] = C.Jv.t[b][q];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int xx = lane; xx < nout * PK; xx += 64) {
      const int pb = (PK == 1) ? 0 : xx / nout, x = xx - pb * nout;
      if (n0 + pb >= S) continue;
      const int xv = x / nS, xk = x - xv * nS;
      const real weight = (m == 0) ? 0.5 : 1.0;
      const real cs = weight * ((xk < 2) ? a.cos_mphi[xv + a.nVza * m] : a.sin_mphi[xv + a.nVza * m]);
      const int row = pb * Nb + (a.node[xv] - 1) * nS + xk;
      const size_t idx = xv + (size_t)a.nVza * xk + (size_t)nout * (n0 + pb);  // [nVza, nStokes, S]
      const real tv = (a.surf_kind == 2 && m > 0) ? 0.0 : cs * post[row];  // Legendre surface: t = 0 for m > 0
      const real hv = (m == 0 || a.surf_kind == 1) ? cs * post[32 * NT + row] : 0.0;  // BRDF surfaces: hdr over all moments
      a.T[idx] = (m == 0) ? tv : a.T[idx] + tv;
      a.R[idx] = (m == 0) ? cs * post[16 * NT + row] : a.R[idx] + cs * post[16 * NT + row];
      a.hdr[idx] = (m == 0) ? hv : a.hdr[idx] + hv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (bad && lane == 0) atomicMax(a.info, bad);
}

}  // namespace MOMW_NS

// The MFMAs of these 4-wave kernels (512 registers per lane) must be in the VGPR form: with the accumulators in AGPRs
// v_mfma_f64_16x16x4 issues at half rate on gfx950 (tools/mfma_peak.hip).  The Makefile passes
// -mllvm -amdgpu-mfma-vgpr-form for this file; that option crashes the compiler on k_wsweep<2, 8>, which is therefore
// built as a second object (mom_wave8.o, -DMOMW_ONLY_KS8) without it.
#ifdef MOMW_FLOAT
#define MOMW_LAUNCH momwf_launch_sweep
#define MOMW_LAUNCH8 momwf_launch_sweep8
#else
#define MOMW_LAUNCH momw_launch_sweep
#define MOMW_LAUNCH8 momw_launch_sweep8
#endif
hipError_t MOMW_LAUNCH8(const void *args, hipStream_t st);
#ifdef MOMW_ONLY_KS8
hipError_t MOMW_LAUNCH8(const void *args, hipStream_t st) {
  const MOMW_NS::WArgs a = *reinterpret_cast<const MOMW_NS::WArgs *>(args);
  const dim3 grid((unsigned)((a.S + 3) / 4)), block(256);
  hipLaunchKernelGGL((MOMW_NS::k_wsweep<2, 8>), grid, block, 0, st, a);
  return hipGetLastError();
}
#else
hipError_t MOMW_LAUNCH(const void *args, hipStream_t st) {
  const MOMW_NS::WArgs a = *reinterpret_cast<const MOMW_NS::WArgs *>(args);
  const dim3 grid((unsigned)((a.S + 3) / 4)), block(256);
  // a.pad = points per wavefront (block-diagonal packing of small operators, k_wsweep's PK): 3 at N = 5, 2 at N = 6..8
  if (a.pad == 3 && a.N == 5) {
    hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 4, 3>), dim3((unsigned)((a.S + 11) / 12)), block, 0, st, a);
    return hipGetLastError();
  }
  if (a.pad == 2 && a.N >= 5 && a.N <= 8) {
    const dim3 g2((unsigned)((a.S + 7) / 8));
#ifndef MOMW_FLOAT  // (the Float32 accumulator layout allows no k-step to be skipped: KS = 4 throughout)
    if (2 * a.N <= 12) hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 3, 2>), g2, block, 0, st, a);
    else
#endif
      hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 4, 2>), g2, block, 0, st, a);
    return hipGetLastError();
  }
  switch ((a.N + 3) / 4) {
    case 2: hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 3>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((MOMW_NS::k_wsweep<1, 4>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((MOMW_NS::k_wsweep<2, 5>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((MOMW_NS::k_wsweep<2, 6>), grid, block, 0, st, a); break;
    case 7: hipLaunchKernelGGL((MOMW_NS::k_wsweep<2, 7>), grid, block, 0, st, a); break;
    case 8: return MOMW_LAUNCH8(args, st);
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#endif
