// mom_tile.hpp -- one-wavefront register tiles on the FP64 matrix cores (shared by the wave-per-unit kernels).
//
// A 16 x 16 tile in the C/D layout of v_mfma_f64_16x16x4_f64 (lane l, register r: row (l >> 4) + 4 r, column l & 15) is
// four doubles per lane.  Feeding register s of tile U as the A operand and register s of tile V as the B operand of the
// four k-steps of a 16 x 16 x 16 product gives TN(U, V) = U^T V in the same layout (the A operand is read as
// A[row = l & 15][k = l >> 4]: a C-layout tile IS the A operand of its transpose).  A matrix of edge N <= 16 NT is
// NT x NT tiles.  Naming used by the callers: X_c = X in C-layout, X_t = X^T in C-layout; then
//      (L R)_c = TN(L_t, R_c)          (L R)_t = TN(R_c, L_t)
// i.e. a product needs its left factor transposed and its right factor plain, and comes out in either orientation.
// A column-major [N, N] block in memory loads COALESCED into the _t form (lanes l & 15 run along a column) and stores
// coalesced from it; the _c form of the same block is one pass through a wave-private LDS slice (transpose()).
#pragma once
#include <hip/hip_runtime.h>

namespace momt {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NT>
struct Mat {
  d4 t[NT][NT];  // t[bi][bj]: rows 16 bi .., columns 16 bj ..
};
template <int NT>
struct Vec {
  d4 t[NT];  // row block bi; column c of the tile (lanes l & 15 == c) is vector number c
};

struct Geo {
  int lr, lq, N;
  double *xp;  // wave-private LDS slice (slice_doubles<NT>() doubles)
  int *ipiv;   // 16 NT ints
  __device__ __forceinline__ int row(int bi, int r) const { return 16 * bi + lq + 4 * r; }
  __device__ __forceinline__ int col(int bj) const { return 16 * bj + lr; }
};

constexpr int kTileLd = 17;                 // pitch of a 16 x 16 tile in the LDS slice
constexpr int kTileDoubles = 16 * kTileLd;  // 272
template <int NT>
constexpr int slice_doubles() {
  return (NT * NT * kTileDoubles > (16 * NT) * (16 * NT + 1) ? NT * NT * kTileDoubles : (16 * NT) * (16 * NT + 1));
}
template <int NT>
constexpr int slice_bytes() { return slice_doubles<NT>() * 8 + 16 * NT * 4; }

template <int NT>
__device__ __forceinline__ Mat<NT> zeros() {
  Mat<NT> Z;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) Z.t[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  return Z;
}
template <int NT>
__device__ __forceinline__ Vec<NT> vzeros() {
  Vec<NT> Z;
#pragma unroll
  for (int a = 0; a < NT; ++a) Z.t[a] = (d4){0.0, 0.0, 0.0, 0.0};
  return Z;
}

// acc + U^T V; in the 2 x 2-tile kernels k-steps whose rows are all >= N (zero padding) are skipped
template <int NT>
__device__ __forceinline__ Mat<NT> TNacc(const Geo &g, const Mat<NT> &U, const Mat<NT> &V, Mat<NT> acc) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (NT == 1 || 16 * tk + 4 * s < g.N) {  // one tile: all four k-steps, unguarded (a guard is a branch around every MFMA)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int tj = 0; tj < NT; ++tj)
            acc.t[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(U.t[tk][ti][s], V.t[tk][tj][s], acc.t[ti][tj], 0, 0, 0);
      }
  return acc;
}
template <int NT>
__device__ __forceinline__ Mat<NT> TN(const Geo &g, const Mat<NT> &U, const Mat<NT> &V) {
  return TNacc<NT>(g, U, V, zeros<NT>());
}
// acc + U^T v for the (up to 16) vectors held as tile columns
template <int NT>
__device__ __forceinline__ Vec<NT> TNvacc(const Geo &g, const Mat<NT> &U, const Vec<NT> &v, Vec<NT> o) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (NT == 1 || 16 * tk + 4 * s < g.N) {  // one tile: all four k-steps, unguarded (a guard is a branch around every MFMA)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
          o.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(U.t[tk][ti][s], v.t[tk][s], o.t[ti], 0, 0, 0);
      }
  return o;
}
template <int NT>
__device__ __forceinline__ Vec<NT> TNv(const Geo &g, const Mat<NT> &U, const Vec<NT> &v) {
  return TNvacc<NT>(g, U, v, vzeros<NT>());
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
__device__ __forceinline__ Vec<NT> vadd(const Vec<NT> &A, const Vec<NT> &B) {
  Vec<NT> C;
#pragma unroll
  for (int a = 0; a < NT; ++a) C.t[a] = A.t[a] + B.t[a];
  return C;
}
template <int NT>
__device__ __forceinline__ Vec<NT> vscale(const Vec<NT> &A, double s) {
  Vec<NT> C;
#pragma unroll
  for (int a = 0; a < NT; ++a) C.t[a] = A.t[a] * s;
  return C;
}
template <int NT>
__device__ __forceinline__ Vec<NT> swap01(const Vec<NT> &v) {  // exchange columns 0 <-> 1
  Vec<NT> o;
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[b][r] = __shfl_xor(v.t[b][r], 1);
  return o;
}

// Device blocks are stored at the pitch of the tiling, 16 NT, with ZERO padding (matrices 16 NT x 16 NT, vectors 16 NT):
// the loads below carry no edge masks (the padding reads as the zeros a mask would have put there), their addresses are
// one per-lane base plus immediates and every 16-lane row is one aligned 128-byte line (mom_rrs.hpp State::P).  Measured on
// the N = 16 RRS scene: 315 -> 275 ms per run.
// The stores write the whole tile as well: every stored quantity is a product / sum / sign flip of zero-padded operands or
// is built with its own i, j < N guards, so the padding stays zero by value (C5: 331 -> 313 ms per run) -- an invariant that
// mom_rrs_check_padding counts violations of and tests/test_gpu_rrs.py::test_rrs_zero_padding_invariant asserts for N = 15,
// 20, 27, 32 in both switch positions.  Round 3 kept masked stores in the 2 x 2-tile kernels because an unmasked build had
// raised a GPU memory fault there; in round 4 that build (-DMOMR_UNMASK2 at the time) passes all 106 RRS tests, the padding
// check and the NT = 2 stress scenes (profiles/r04_C5_ab.txt), so the fault belonged to an intermediate state of the r3
// layout change, not to the stores.  -DMOMR_MASK2 restores the masked form.
#ifdef MOMR_MASK2
template <int NT>
__device__ __forceinline__ constexpr bool mask_store() { return NT > 1; }
#else
template <int NT>
__device__ __forceinline__ constexpr bool mask_store() { return false; }
#endif
// column-major [N, N] block at pitch 16 NT -> X_t (coalesced) / X_c (strided)
template <int NT>
__device__ __forceinline__ Mat<NT> load_t(const Geo &g, const double *p) {
  Mat<NT> X;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ro = g.row(a, r), co = g.col(b);
        X.t[a][b][r] = p[co + (16 * NT) * ro];
      }
  return X;
}
template <int NT>
__device__ __forceinline__ Mat<NT> load_c(const Geo &g, const double *p) {
  Mat<NT> X;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ro = g.row(a, r), co = g.col(b);
        X.t[a][b][r] = p[ro + (16 * NT) * co];
      }
  return X;
}
template <int NT>
__device__ __forceinline__ void store_t(const Geo &g, double *p, const Mat<NT> &X) {
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ro = g.row(a, r), co = g.col(b);
        if (!mask_store<NT>() || (ro < g.N && co < g.N)) p[co + (16 * NT) * ro] = X.t[a][b][r];
      }
}
// two vectors [N] as columns 0 and 1 (nullptr: zeros)
template <int NT>
__device__ __forceinline__ Vec<NT> loadv2(const Geo &g, const double *p0, const double *p1) {
  Vec<NT> v;
  const double *p = (g.lr == 0) ? p0 : ((g.lr == 1) ? p1 : nullptr);
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ro = g.row(a, r);
      v.t[a][r] = (p != nullptr) ? p[ro] : 0.0;
    }
  return v;
}
template <int NT>
__device__ __forceinline__ void storev(const Geo &g, double *p, const Vec<NT> &v, int column) {
  if (g.lr != column) return;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ro = g.row(a, r);
      if (!mask_store<NT>() || ro < g.N) p[ro] = v.t[a][r];
    }
}

// ---- matrix-vector products on the vector ALU -------------------------------------------------------------------------
// A 16 x 16 x 4 MFMA spends a whole tile column per vector; for the few matrix-vector products of a kernel that is
// otherwise bound by the matrix pipe it is cheaper to multiply on the vector ALU and reduce across lanes:
//   RV ("row layout"):    lane (lq, lr) holds x[16 a + lq + 4 r] in t[a][r] -- the same value in all 16 lanes lr
//   CV ("column layout"): lane (lq, lr) holds y[16 b + lr] in c[b]          -- the same value in all 4 lane groups lq
// y = M x with M_t (C-layout tiles of M^T) takes x in row layout and returns y in column layout: 4 NT FMAs per lane and
// row block, then two cross-lane steps (xor 16, xor 32); c2r converts a column-layout vector to row layout (4 NT shuffles).
template <int NT>
struct CV {
  double c[NT];
};
template <int NT>
__device__ __forceinline__ Vec<NT> loadR(const Geo &g, const double *p) {
  Vec<NT> v;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ro = g.row(a, r);
      v.t[a][r] = p[ro];
    }
  return v;
}
template <int NT>
__device__ __forceinline__ CV<NT> loadC(const Geo &g, const double *p) {
  CV<NT> v;
#pragma unroll
  for (int b = 0; b < NT; ++b) v.c[b] = p[g.col(b)];
  return v;
}
template <int NT>
__device__ __forceinline__ CV<NT> czeros() {
  CV<NT> v;
#pragma unroll
  for (int b = 0; b < NT; ++b) v.c[b] = 0.0;
  return v;
}
template <int NT>
__device__ __forceinline__ void storeC(const Geo &g, double *p, const CV<NT> &v) {
  if (g.lq != 0) return;
#pragma unroll
  for (int b = 0; b < NT; ++b)
    if (!mask_store<NT>() || g.col(b) < g.N) p[g.col(b)] = v.c[b];
}
template <int NT>
__device__ __forceinline__ CV<NT> cadd(const CV<NT> &A, const CV<NT> &B) {
  CV<NT> C;
#pragma unroll
  for (int b = 0; b < NT; ++b) C.c[b] = A.c[b] + B.c[b];
  return C;
}
template <int NT>
__device__ __forceinline__ CV<NT> cscale(const CV<NT> &A, double s) {
  CV<NT> C;
#pragma unroll
  for (int b = 0; b < NT; ++b) C.c[b] = A.c[b] * s;
  return C;
}
// lane l <- lane l ^ 16 / l ^ 32 of a double on the vector ALU (gfx950 v_permlane16_swap / v_permlane32_swap: swapping a
// register with itself exchanges odd and even rows of 16 lanes / the two halves of the wave) -- an order of magnitude
// less latency than the LDS crossbar (ds_bpermute) behind __shfl_xor, which matters in the dependent reduction chains
__device__ __forceinline__ double lane_xor16(double v, bool odd_row) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(odd_row ? b[0] : b[1], odd_row ? a[0] : a[1]);
}
__device__ __forceinline__ double lane_xor32(double v, bool upper_half) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(upper_half ? b[0] : b[1], upper_half ? a[0] : a[1]);
}
// y = M x, M given as M_t
template <int NT>
__device__ __forceinline__ CV<NT> mv_t(const Geo &g, const Mat<NT> &M_t, const Vec<NT> &xR) {
  CV<NT> y;
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fma(M_t.t[a][b][r], xR.t[a][r], acc);
    acc += lane_xor16(acc, (g.lq & 1) != 0);
    acc += lane_xor32(acc, (g.lq & 2) != 0);
    y.c[b] = acc;
  }
  return y;
}
template <int NT>
__device__ __forceinline__ Vec<NT> c2r(const Geo &g, const CV<NT> &y) {
  Vec<NT> x;
  const int base = (g.lq << 4);
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) x.t[a][r] = __shfl(y.c[a], base | (g.lq + 4 * r));
  return x;
}

// X^T through the wave's LDS slice: tile (a, b) of the result is the transpose of tile (b, a)
template <int NT>
__device__ __forceinline__ Mat<NT> transpose(const Geo &g, const Mat<NT> &X) {
  double *buf = g.xp;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf[(a * NT + b) * kTileDoubles + (g.lq + 4 * r) * kTileLd + g.lr] = X.t[a][b][r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  Mat<NT> Y;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y.t[a][b][r] = buf[(b * NT + a) * kTileDoubles + g.lr * kTileLd + g.lq + 4 * r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return Y;
}

template <int NT>
__device__ __forceinline__ Mat<NT> ident(const Geo &g) {
  Mat<NT> I;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) I.t[a][b][r] = (a == b && g.lq + 4 * r == g.lr && g.col(b) < g.N) ? 1.0 : 0.0;
  return I;
}

// (I - B)^-1 for B given in either orientation (the result comes out in the same one): Gauss-Jordan elimination with
// implicit partial pivoting, one matrix row per lane through the wave's LDS slice.  *bad_out = 1 + the step of a zero pivot.
template <int NT>
__device__ __noinline__ Mat<NT> inv_one_minus(const Geo &g, Mat<NT> B, int *bad_out) {
  constexpr int NP = 16 * NT, LDM = NP + 1;
  const int N = g.N, lane = 16 * g.lq + g.lr, lr = g.lr, lq = g.lq;
  double *lds = g.xp;
  int *ipiv = g.ipiv;
  int bad = 0;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * a + lq + 4 * r, j = 16 * b + lr;
        lds[i * LDM + j] = ((i == j) ? 1.0 : 0.0) - B.t[a][b][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double v[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) v[c] = (lane < N && c < N) ? lds[lane * LDM + c] : ((lane == c) ? 1.0 : 0.0);
  bool used = false;
  int myk = lane;  // rows >= N keep their identity row
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (k < N) {
      const int ah = (!used && lane < N) ? __double2hiint(fabs(v[k])) : -1;
      int mh = ah;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mh = max(mh, __shfl_xor(mh, off));
      const unsigned long long mk = __ballot(ah == mh);
      const int pl = __ffsll((long long)mk) - 1;
      const double piv = __shfl(v[k], pl);
      if (!(fabs(piv) > 0.0) && !bad) bad = k + 1;
      const double d = 1.0 / piv, f = v[k];
      const bool isp = (lane == pl);
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        const double prow = __shfl(v[c], pl) * d;
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
        const int i = 16 * a + lq + 4 * r, j = 16 * b + lr;
        G.t[a][b][r] = (i < N && j < N) ? lds[i * LDM + j] : 0.0;
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (bad) *bad_out = bad;
  return G;
}

}  // namespace momt
