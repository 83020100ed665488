// mom_q4.hpp -- the QUAD-BLOCK image (r6): one WAVEFRONT owns one (spectral point, moment) unit of an N <= 40 problem and
// multiplies on v_mfma_f64_4x4x4_4b -- the m = 0 launch of the headline scene (N0 = 40) and every operator of edge 36 / 40.
//
// Why (profiles/r06_q4_ab.txt).  With 16 x 16 tiles an N = 40 operator is 3 x 3 tiles: a workgroup runs its strip chains on three
// waves, any number of co-resident workgroups puts 1/3 of the matrix work on the busiest SIMD (the 54 ms floor of the r5 lean
// image), and 48 of the 16 x 3 = 48 rows / columns a strip product touches are 17 % padding each way.  v_mfma_f64_4x4x4_4b
// issues at the SAME rate as the 16 x 16 x 4 instruction on gfx950 (tools/mfma_4x4_probe.hip: 74.8 TFLOP/s at one wave per
// SIMD) with a granule of 4: N = 40 is ten blocks exactly.  Its operand maps (same probe; block b = lane bits 2..3):
//     A[b][i][k] in lane 16 k + 4 b + i,   B[b][k][j] in lane 16 k + 4 b + j,   D[b][i][j] in lane 16 i + 4 b + j,
// the four blocks of an instruction are independent products D[b] = A[b] B[b] (cbsz / abid do nothing for this instruction).
// So D is again a B operand (i -> k), and an operator X lives in NB x NJ registers (NB = N / 4 block rows, NJ = ceil(NB / 4)
// groups of four block columns: element (4 K + k, 16 Jg + 4 b + j) in lane 16 k + 4 b + j of register (K, Jg)) -- 30 registers,
// 60 VGPRs, for N = 40: a whole operator per wave.  X' = M^T X needs A[b] = block (I, K) of M^T in ALL four lane groups: the
// lanes read M[4 K + k + (4 I + i) LD] from the LDS buffer of M, the same 16 addresses in every group (an LDS broadcast read,
// conflict-free at LD = 40), one ds_read_b64 per three MFMAs.  100 reads + 300 MFMAs per 40^3 product, no barrier, no other
// wave: tools/q4_probe.hip measures 5 295 cycles per product with four one-wave workgroups per CU (ideal 4 800) = 0.756 of the
// FP64 matrix peak in USEFUL flops, against 0.52 for three 4-wave strip workgroups per CU (3 products per 5 760 cycles).
//
// The algebra is mom_strip.hpp's / mom_lean.hpp's (chains on the transposed quantities, multipliers M^T read from the buffer
// of M), with the strip replaced by the whole operator; the source vectors ride as an eleventh block row (rows N .. N + 3 of the
// multiplier = up to four vectors kept next to the buffers), 30 more MFMAs where a product carries them.
// LDS: r, t, P at pitch N (3 x 12 800 B) + eight vectors of N = 40 960 B at N = 40 -- exactly a quarter of the CU's 160 KB, so
// FOUR units share a CU, one per SIMD.  Everything the lean image leaves to the full image is left here too (series beyond
// kStripMaxP terms, forced pivoting, interfaces other than 11, multi-target sweeps): resume[unit] tells the following launch of
// the full image where to pick the unit up.  Scope: Float64, MOM_WAVES = 1 build (namespace momq), sweep mode.
#pragma once
#include "mom_entry.hpp"

namespace MOM_NS {

static_assert(kWaves == 1, "mom_q4.hpp: one wavefront per workgroup (-DMOM_WAVES=1)");

__device__ __forceinline__ real mma4(real a, real b, real c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

#ifndef Q4_MUL_VARIANT
#define Q4_MUL_VARIANT 0
#endif
template <int KS>
struct Q4Geom {
  static constexpr int N = 4 * KS, NB = KS, NJ = (KS + 3) / 4, LD = N;
  static constexpr int CP = 16 * ((N + 15) / 16);   // row pitch of the scene-level composite blocks (comp_pitch)
};
__host__ __device__ inline size_t q4_lds_bytes(int N) { return ((size_t)3 * N * N + 8 * (size_t)N) * sizeof(real); }
// table space of the elemental layer inside P: E | F1 | F2 (3 Nq^2), the sun-block columns (2 ns N), the layer's scalars (3 + K)
__host__ __device__ inline bool q4_applies(int N, int ns, int K) {
  const int Nq = N / (ns > 0 ? ns : 1);
  return kF64 && N >= 20 && N <= 40 && N % 4 == 0 && ns >= 2 && N % ns == 0 && K <= 60 &&   // (3 + K layer scalars: one lane each)
         3 * Nq * Nq + 2 * ns * N + 3 + K <= N * N &&
         4 * q4_lds_bytes(N) <= kLdsPerCU;
}

// lane coordinates of the D / B layout
struct Q4Lane {
  int k, c16;   // row within a block row (lane >> 4); column within a column group, 4 b + j (lane & 15)
  __device__ __forceinline__ Q4Lane() {
    int l = wg_lane();
    asm volatile("" : "+v"(l));
    k = l >> 4;
    c16 = l & 15;
  }
};

template <int KS> using Q4Mat = real[Q4Geom<KS>::NB][Q4Geom<KS>::NJ];

template <int KS>
__device__ __forceinline__ void q4_zero(real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
#pragma unroll
  for (int K = 0; K < Q4Geom<KS>::NB; ++K)
#pragma unroll
    for (int J = 0; J < Q4Geom<KS>::NJ; ++J) W[K][J] = 0.0;
}
template <int KS>
__device__ __forceinline__ void q4_copy(real (&D)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], const real (&S)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
#pragma unroll
  for (int K = 0; K < Q4Geom<KS>::NB; ++K)
#pragma unroll
    for (int J = 0; J < Q4Geom<KS>::NJ; ++J) D[K][J] = S[K][J];
}

// W[row][col] = X[col][row] for the column-major LDS buffer X (pitch LD): the transposed operator; columns >= N read as zero
template <int KS>
__device__ __forceinline__ void q4_load_T(const real *X, const Q4Lane &q, real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
  using G = Q4Geom<KS>;
  const real *base = X + q.c16 + q.k * G::LD;
#pragma unroll
  for (int K = 0; K < G::NB; ++K)
#pragma unroll
    for (int J = 0; J < G::NJ; ++J) W[K][J] = (16 * J + q.c16 < G::N) ? base[16 * J + 4 * K * G::LD] : 0.0;
}
template <int KS>
__device__ __forceinline__ void q4_store_T(real *X, const Q4Lane &q, const real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
  using G = Q4Geom<KS>;
  real *base = X + q.c16 + q.k * G::LD;
#pragma unroll
  for (int K = 0; K < G::NB; ++K)
#pragma unroll
    for (int J = 0; J < G::NJ; ++J)
      if (16 * J + q.c16 < G::N) base[16 * J + 4 * K * G::LD] = W[K][J];
}
// the same on a composite block in global memory (pitch CP): 16 consecutive reals per (row, column group) = whole 128-byte lines
template <int KS>
__device__ __forceinline__ void q4_load_glb_T(const gdouble *__restrict__ X, const Q4Lane &q, real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
  using G = Q4Geom<KS>;
  const gdouble *base = X + q.c16 + q.k * G::CP;
#pragma unroll
  for (int K = 0; K < G::NB; ++K)
#pragma unroll
    for (int J = 0; J < G::NJ; ++J) W[K][J] = (16 * J + q.c16 < G::N) ? MOM_NT_LOAD(base + 16 * J + 4 * K * G::CP) : 0.0;
}
template <int KS>
__device__ __forceinline__ void q4_store_glb_T(gdouble *__restrict__ X, const Q4Lane &q, const real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
  using G = Q4Geom<KS>;
  gdouble *base = X + q.c16 + q.k * G::CP;
#pragma unroll
  for (int K = 0; K < G::NB; ++K)
#pragma unroll
    for (int J = 0; J < G::NJ; ++J)
      if (16 * J + q.c16 < G::N) MOM_NT_STORE(W[K][J], base + 16 * J + 4 * K * G::CP);
}

// acc += M^T X: acc(I, Jg) += sum_K A(I, K) X(K, Jg), A(I, K)[i][k] = M[4 K + k + (4 I + i) LD] in every lane group.
// NV > 0: the riding block row -- accR(Jg) += sum_K V(K) X(K, Jg) with V(K)[i][k] = ride[i * N + 4 K + k] for i < NV (vector i at
// ride + i N), zero rows above: row i of accR is v_i^T X.
// q4_mul_c: out = C0 + M^T X (the MFMA's C operand of the first k-block is C0: no copy of the initial value); q4_mul: acc += M^T X.
template <int KS, int NV = 0, bool ZERO = false>
__device__ __forceinline__ void q4_mul_c(const real *M, const real (&X)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ],
                                         const real (&C0)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ],
                                         real (&acc)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], const real *ride = nullptr,
                                         real (*accR)[Q4Geom<KS>::NJ] = nullptr) {
  using G = Q4Geom<KS>;
  int l = wg_lane();
  asm volatile("" : "+v"(l));   // keep the address arithmetic inside (mom_device.hpp item_straight)
  const int k = l >> 4, i = l & 3;
  const real *base = M + k + i * G::LD;
#if Q4_MUL_VARIANT == 2
  // the plain triple loop, scheduled by the compiler (tools/q4_probe.hip's form)
#pragma unroll
  for (int I = 0; I < G::NB; ++I)
#pragma unroll
    for (int K = 0; K < G::NB; ++K) {
      const real a = base[4 * K + 4 * I * G::LD];
#pragma unroll
      for (int J = 0; J < G::NJ; ++J) acc[I][J] = mma4(a, X[K][J], K == 0 ? (ZERO ? (real)0 : C0[I][J]) : acc[I][J]);
    }
  if constexpr (NV > 0) {
    const real *rb = ride + k + i * G::N;
#pragma unroll
    for (int K = 0; K < G::NB; ++K) {
      const real a = (i < NV) ? rb[4 * K] : 0.0;
#pragma unroll
      for (int J = 0; J < G::NJ; ++J) (*accR)[J] = mma4(a, X[K][J], (*accR)[J]);
    }
  }
  return;
#endif
  // software pipeline over PAIRS of block rows: the A fragments of the next pair are requested before the 60 MFMAs of the current
  // one (left to itself the compiler reads each fragment right in front of its MFMAs and waits for it: one exposed LDS latency per
  // six MFMAs -- the first version of this image ran its products at 9 .. 11 k cycles instead of 5.3 k); two rows at a time give six
  // independent accumulators per k-block (three leave the pipe waiting for its own result: s_nop between the groups)
  constexpr int NP = (G::NB + 1) / 2;          // row pairs (the last one may be a single row)
  real an[2][G::NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int K = 0; K < G::NB; ++K) an[h][K] = (h < G::NB) ? base[4 * K + 4 * h * G::LD] : 0.0;
#pragma unroll
  for (int Pp = 0; Pp < NP; ++Pp) {
    const int I0 = 2 * Pp;
    const bool two = I0 + 1 < G::NB;
    real ac[2][G::NB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int K = 0; K < G::NB; ++K) ac[h][K] = an[h][K];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int In = I0 + 2 + h;
      if (In < G::NB) {
#pragma unroll
        for (int K = 0; K < G::NB; ++K) an[h][K] = base[4 * K + 4 * In * G::LD];
      } else if (NV > 0 && In == G::NB) {   // the riding block row follows the last operator row
        const real *rb = ride + k + i * G::N;
#pragma unroll
        for (int K = 0; K < G::NB; ++K) an[h][K] = (i < NV) ? rb[4 * K] : 0.0;
      }
    }
#if Q4_MUL_VARIANT != 1
    __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of the MFMAs
#endif
#pragma unroll
    for (int K = 0; K < G::NB; ++K) {
#pragma unroll
      for (int J = 0; J < G::NJ; ++J) acc[I0][J] = mma4(ac[0][K], X[K][J], K == 0 ? (ZERO ? (real)0 : C0[I0][J]) : acc[I0][J]);
      if (two) {
#pragma unroll
        for (int J = 0; J < G::NJ; ++J) acc[I0 + 1][J] = mma4(ac[1][K], X[K][J], K == 0 ? (ZERO ? (real)0 : C0[I0 + 1][J]) : acc[I0 + 1][J]);
      } else if (NV > 0) {   // odd NB: the riding row pairs with the last operator row
#pragma unroll
        for (int J = 0; J < G::NJ; ++J) (*accR)[J] = mma4(ac[1][K], X[K][J], (*accR)[J]);
      }
    }
#if Q4_MUL_VARIANT != 1
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  if constexpr (NV > 0 && (G::NB % 2) == 0) {   // even NB: the riding row is requested into an[0] by the last pair
#pragma unroll
    for (int K = 0; K < G::NB; ++K) {
#pragma unroll
      for (int J = 0; J < G::NJ; ++J) (*accR)[J] = mma4(an[0][K], X[K][J], (*accR)[J]);
    }
  }
}

template <int KS, int NV = 0>
__device__ __forceinline__ void q4_mul(const real *M, const real (&X)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ],
                                       real (&acc)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], const real *ride = nullptr,
                                       real (*accR)[Q4Geom<KS>::NJ] = nullptr) {
  q4_mul_c<KS, NV>(M, X, acc, acc, ride, accR);
}
// acc = M^T X (the first k-block's C operand is the constant zero: no zeroing of the accumulators)
template <int KS, int NV = 0>
__device__ __forceinline__ void q4_mul_z(const real *M, const real (&X)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ],
                                         real (&acc)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], const real *ride = nullptr,
                                         real (*accR)[Q4Geom<KS>::NJ] = nullptr) {
  q4_mul_c<KS, NV, true>(M, X, acc, acc, ride, accR);
}

// Y = sum_{k < p} (M^T)^k C0 by Horner, Y <- C0 + M^T Y starting from Y = C0: p - 1 products at ONE product site (a second site
// for alternating registers, i.e. no copy at all, measured slower: 128 KB of code and 105 spilled registers, 60.9 -> 64.5 ms on the
// C2 m = 0 launch); the initial value is the first k-block's C operand, so a round costs one copy of the running value, not two
#ifndef Q4_HORNER_SITES
#define Q4_HORNER_SITES 1
#endif
template <int KS>
__device__ __forceinline__ void q4_horner(const real *M, const real (&C0)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], int p,
                                          real (&Y)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ]) {
  q4_copy<KS>(Y, C0);
  MOM_STAMP(82);
#if Q4_HORNER_SITES == 2
  // two product sites on alternating registers: one copy per series (odd p - 1) instead of one per round
  int n = p - 1;
#pragma nounroll
  for (; n >= 2; n -= 2) {
    real Z[Q4Geom<KS>::NB][Q4Geom<KS>::NJ];
    q4_mul_c<KS>(M, Y, C0, Z);
    q4_mul_c<KS>(M, Z, C0, Y);
  }
  if (n == 1) {
    real acc[Q4Geom<KS>::NB][Q4Geom<KS>::NJ];
    q4_mul_c<KS>(M, Y, C0, acc);
    q4_copy<KS>(Y, acc);
  }
#else
#pragma nounroll
  for (int kk = 1; kk < p; ++kk) {
    real acc[Q4Geom<KS>::NB][Q4Geom<KS>::NJ];
    MOM_STAMP(74);
    q4_mul_c<KS>(M, Y, C0, acc);
#ifdef MOM_DIAG_STAMPS
    if (kk == 1) { MOM_STAMP(75); } else { MOM_STAMP(77); }   // first round of a series (code fetched from L2?) against the later ones
    if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) mom_diag_acc[kk == 1 ? 78 : 79] += 1;
#endif
    q4_copy<KS>(Y, acc);
    MOM_STAMP(76);
  }
#endif
}

// bit K of the mask: sg[4 K + k] < 0 (this lane's row of block row K)
template <int KS>
__device__ __forceinline__ unsigned q4_sign_mask(const real *sg, const Q4Lane &q) {
  unsigned m = 0;
#pragma unroll
  for (int K = 0; K < Q4Geom<KS>::NB; ++K)
    if (sg[4 * K + q.k] < 0.0) m |= 1u << K;
  return m;
}
template <int KS>
__device__ __forceinline__ void q4_flip(real (&W)[Q4Geom<KS>::NB][Q4Geom<KS>::NJ], unsigned mask) {
#pragma unroll
  for (int K = 0; K < Q4Geom<KS>::NB; ++K)
#pragma unroll
    for (int J = 0; J < Q4Geom<KS>::NJ; ++J) {
      const real v = W[K][J];
      W[K][J] = ((mask >> K) & 1u) ? -v : v;
    }
}
// one-wave LDS fence: every LDS write of this wave is visible to its later reads in program order; the compiler only has to keep
// the order (its alias analysis cannot see through the laundered lane coordinates)
__device__ __forceinline__ void q4_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_s_barrier(); }

// sum over the 64 lanes by DPP row operations (the __shfl_xor ladder of wave_sum is six LDS round trips, each exposed with one wave
// per SIMD): shift-adds within the rows of 16, then the row totals through readlane
__device__ __forceinline__ real q4_wave_sum(real v) {
  union { real d; int w[2]; } a, b;
#define Q4_DPP_ADD(ctrl)                                                                      \
  a.d = v;                                                                                    \
  b.w[0] = __builtin_amdgcn_update_dpp(0, a.w[0], ctrl, 0xf, 0xf, false);                     \
  b.w[1] = __builtin_amdgcn_update_dpp(0, a.w[1], ctrl, 0xf, 0xf, false);                     \
  v += b.d;
  Q4_DPP_ADD(0x111)  // row_shr:1
  Q4_DPP_ADD(0x112)  // row_shr:2
  Q4_DPP_ADD(0x114)  // row_shr:4
  Q4_DPP_ADD(0x118)  // row_shr:8   -> lane 15 of every row holds the row total
#undef Q4_DPP_ADD
  a.d = v;
  real s = 0.0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    union { real d; int w[2]; } t;
    t.w[0] = __builtin_amdgcn_readlane(a.w[0], 16 * row + 15);
    t.w[1] = __builtin_amdgcn_readlane(a.w[1], 16 * row + 15);
    s += t.d;
  }
  return s;
}

__device__ __forceinline__ void make_ctx_q4(Ctx &c, int N, int inv_mode, real *smem) {
  c.N = N; c.Np = N; c.nc = N; c.ld = N; c.ldv = N;
  c.fd.init(N);
  c.inv_mode = inv_mode;
  c.qpre = 0; c.slot = 0; c.ptab = nullptr;
  const size_t msz = (size_t)N * N;
  c.r = smem; c.t = smem + msz; c.P = smem + 2 * msz; c.Q = nullptr; c.X = nullptr;
  real *p = smem + 3 * msz;
  c.mu = p; c.wt = p + N; c.sg = p + 2 * N; c.jp = p + 3 * N; c.jm = p + 4 * N; c.ei = p + 5 * N; c.v1 = p + 6 * N; c.v2 = p + 7 * N;
  c.j1p = c.j1m = c.Jp = c.Jm = c.prow = c.pcol = c.rowk = c.part = c.thr = nullptr;
  c.ipiv = c.sh = nullptr;
  c.bad = reinterpret_cast<int *>(c.P + msz - 2);   // (load_streams clears it; nothing of this image sets it)
  c.tabE = c.tabZS = nullptr;
}

// nd doubling steps (doubling.jl:43-68) on the quad-block layout; bail = true, nothing of the layer has left the wave, if a step
// needs the general path.  In / out: c.r, c.t (plain, pitch N), c.jp, c.jm.  After the elemental layer c.ei, c.v1, c.v2 are free:
// c.ei / c.v1 hold the riding vectors of the step (w1, w2 of doubling_step_strip).
template <int KS>
__device__ __forceinline__ real doubling_run_q4(Ctx &c, int nd, real expk, bool &bail) {
  using G = Q4Geom<KS>;
  constexpr int N = G::N, NB = G::NB, NJ = G::NJ;
  bail = false;
  if (nd == 0) return expk;
  const Q4Lane q;
  real *r = c.r, *t = c.t, *P = c.P;
  for (int it = 0; it < nd; ++it) {
    // ---- P = r r, the riding rows r j0+, r j0-, and beta^2 = ||r r||_F^2
    real rj[NJ];                 // riding rows of the product, lanes k == 0: (r j0+)[col] ; k == 1: (r j0-)[col]
    real ss = 0.0;
    real Rn[NB][NJ], T0[NB][NJ];   // r^T (multiplied by r^T first, the initial value of the new r^T later) and t^T, requested together
    q4_load_T<KS>(r, q, Rn);
    q4_load_T<KS>(t, q, T0);
    {
      real B[NB][NJ];
#pragma unroll
      for (int J = 0; J < NJ; ++J) rj[J] = 0.0;
      // the vectors j0+, j0- sit in c.jp, c.jm = consecutive vectors (make_ctx_q4): ride base c.jp, vector 0 = j0+, 1 = j0-
      q4_mul_z<KS, 2>(r, Rn, B, c.jp, &rj);
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int J = 0; J < NJ; ++J) ss += B[K][J] * B[K][J];   // columns >= N are exact zeros (r^T's are)
      q4_store_T<KS>(P, q, B);
    }
    MOM_STAMP(70);
    ss = q4_wave_sum(ss);
    const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(ss));
    if (p > kStripMaxP || c.inv_mode != 0) {
      bail = true;
      return expk;
    }
    MOM_STAMP(80);
    // riding vectors of the multiplier r for the (A r) product: w1 = j1- + r j0+, w2 = j0+ + r j1- (doubling.jl:51-60), j1 = j0 expk
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int col = 16 * J + q.c16;
      if (col < N) {
        if (q.k == 0) c.ei[col] = c.jm[col] * expk + rj[J];          // w1  (lanes k == 0 hold r j0+)
        if (q.k == 1) c.v1[col] = c.jp[col] + expk * rj[J];          // w2  (lanes k == 1 hold r j0-); c.v1 = c.ei + N: ride base c.ei
      }
    }
    q4_fence();
    MOM_STAMP(81);
    real Y[NB][NJ];
    // Y = A^T = (t (I - r r)^-1)^T by Horner: Y <- t^T + (r r)^T Y, p - 1 times starting from t^T (p >= 2 unless r r = 0)
    q4_horner<KS>(P, T0, p, Y);
    MOM_STAMP(71);
    real aw[NJ];   // lanes k == 0: (A w1)[col]; k == 1: (A w2)[col]
    real Tn[NB][NJ];
    {
      real Zt[NB][NJ];
#pragma unroll
      for (int J = 0; J < NJ; ++J) aw[J] = 0.0;
      q4_mul_z<KS, 2>(r, Y, Zt, c.ei, &aw);    // (A r)^T ; riding rows (A w1)^T, (A w2)^T
      q4_mul<KS>(t, Zt, Rn);                    // r^T + t^T (A r)^T      (:64)
    }
    q4_mul_z<KS>(t, Y, Tn);                       // t^T A^T                (:67)
    // last step: apply_D! (doubling.jl:93-110) and apply_D_SFI! (:112-118) ride on the write-back -- the rows of r-+ (columns of
    // its transpose held here) and j0- are scaled by sg
    const bool last = (it == nd - 1);
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int col = 16 * J + q.c16;
      const real sc = (last && col < N) ? c.sg[col] : 1.0;
#pragma unroll
      for (int K = 0; K < NB; ++K) Rn[K][J] = Rn[K][J] * sc;
      if (col < N) {
        if (q.k == 0) c.jm[col] = (c.jm[col] + aw[J]) * sc;           // j0- += A w1 (:57)
        if (q.k == 1) c.jp[col] = c.jp[col] * expk + aw[J];           // j0+ = j1+ + A w2 (:60)
      }
    }
    q4_store_T<KS>(r, q, Rn);
    q4_store_T<KS>(t, q, Tn);
    q4_fence();
    MOM_STAMP(72);
    expk = expk * expk;
  }
  MOM_STAMP(73);
  return expk;
}

// ScatteringInterface_11 (interaction.jl:69-117) in the algebra of interaction_strip / interaction_strip_lean.  Returns false,
// nothing stored, if the series is too long.  Riding vectors: c.jm rides with r (column "N" of r = j0-); the composite J0+ is
// fetched into c.v1 and rides with P = T++.
template <int KS>
__device__ __forceinline__ bool interaction_q4(Ctx &c, const CompPtrs &g) {
  using G = Q4Geom<KS>;
  constexpr int N = G::N, NB = G::NB, NJ = G::NJ, NN = N * N;
  const Q4Lane q;
  const int lane = wg_lane();
  real *r = c.r, *t = c.t, *P = c.P;
  // P = R+- (plain)
  {
    constexpr int U = (NN + 63) / 64;
    real vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = lane + 64 * u;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        vr[u] = MOM_NT_LOAD(g.R_pm + i + j * G::CP);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = lane + 64 * u;
      if (e < NN) P[e] = vr[u];
    }
  }
  q4_fence();
  MOM_STAMP(50);
  // B^T = R+-^T r^T and W0 = R+-^T t^T
  real W0[NB][NJ];
  real ss = 0.0;
  {
    real Bs[NB][NJ];
    {
      real rT[NB][NJ];
      q4_load_T<KS>(r, q, rT);
      q4_mul_z<KS>(P, rT, Bs);
    }
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int J = 0; J < NJ; ++J) ss += Bs[K][J] * Bs[K][J];
    {
      real tT[NB][NJ];
      q4_load_T<KS>(t, q, tT);
      q4_mul_z<KS>(P, tT, W0);
    }
    q4_fence();
    q4_store_T<KS>(P, q, Bs);   // P = B = r-+ R+-
  }
  q4_fence();
  ss = q4_wave_sum(ss);
  const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(ss));
  if (p > kStripMaxP) return false;
  MOM_STAMP(52);
  const unsigned mask = q4_sign_mask<KS>(c.sg, q);
  // Global loads are requested one stage early -- one wave per SIMD has nobody to hide a memory round trip behind: T-- goes out
  // before the X series, T++ (+ J0+) before the T01 series, R-+ before the first product of chain 1.
  real Y1[NB][NJ], Y2[NB][NJ];
  constexpr int UT = (NN + 63) / 64;
  real vt[UT], vj;
  {
    real T1[NB][NJ];
    q4_load_glb_T<KS>(g.T_mm, q, T1);
    __builtin_amdgcn_sched_barrier(0);
    q4_horner<KS>(P, W0, p, Y2);       // Y2 <- W0 + B^T Y2 : X^T, X = t++ R+- (I - B)^-1
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int e = lane + 64 * u;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        vt[u] = MOM_NT_LOAD(g.T_pp + i + j * G::CP);
      }
    }
    vj = (lane < N) ? g.J0p[lane] : 0.0;
    __builtin_amdgcn_sched_barrier(0);
    q4_horner<KS>(P, T1, p, Y1);       // Y1 <- T--^T + B^T Y1 : T01^T
  }
  q4_fence();
  MOM_STAMP(55);
  // P = T++ (plain); c.v1 = J0+ (rides with P)
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int e = lane + 64 * u;
    if (e < NN) P[e] = vt[u];
  }
  if (lane < N) c.v1[lane] = vj;
  real Radd[NB][NJ];
  q4_load_glb_T<KS>(g.R_mp, q, Radd);
  q4_fence();
  MOM_STAMP(56);
  // ---- chain 1: T-- = T01 t--, R-+ += (T01 r-+) T++, J0- += T01 (r-+ J0+ + j0-)                (:90-96)
  {
    {
      real Yf[NB][NJ], o[NB][NJ];
      q4_copy<KS>(Yf, Y1);
      q4_flip<KS>(Yf, mask);
      q4_mul_z<KS>(t, Yf, o);
      q4_flip<KS>(o, mask);
      q4_store_glb_T<KS>(g.T_mm, q, o);
    }
    real V[NB][NJ], Vr[NJ], Rr[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) { Vr[J] = 0.0; Rr[J] = 0.0; }
    q4_mul_z<KS, 1>(r, Y1, V, c.jm, &Vr);                 // (T01 r-+)^T ; riding row: (T01 j0-)^T
    q4_mul<KS, 1>(P, V, Radd, c.v1, &Rr);               // R-+^T + T++^T (T01 r)^T ; riding row: (T01 r J0+)^T
    q4_store_glb_T<KS>(g.R_mp, q, Radd);
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int col = 16 * J + q.c16;
      if (col < N && q.k == 0) g.J0m[col] = g.J0m[col] + (Rr[J] + Vr[J]);
    }
  }
  MOM_STAMP(58);
  // ---- chain 2: T21 = t++ + X r-+, T++ = T21 T++, J0+ = j0+ + T21 (J0+ + R+- j0-), R+- = r+- + X t--   (:110-116)
  {
    real T21[NB][NJ], Tr[NJ], Or[NJ];
    q4_load_T<KS>(t, q, T21);
#pragma unroll
    for (int J = 0; J < NJ; ++J) { Tr[J] = 0.0; Or[J] = 0.0; }
    q4_mul<KS, 1>(r, Y2, T21, c.jm, &Tr);               // riding row: (X j0-)^T = (T21 R+- j0-)^T
    {
      real o[NB][NJ];
      q4_mul_z<KS, 1>(P, T21, o, c.v1, &Or);              // (T21 T++)^T ; riding row: (T21 J0+)^T
      q4_store_glb_T<KS>(g.T_pp, q, o);
    }
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int col = 16 * J + q.c16;
      if (col < N && q.k == 0) g.J0p[col] = c.jp[col] + (Or[J] + Tr[J]);
    }
    real acc[NB][NJ];
    q4_load_T<KS>(r, q, acc);
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
      const int col = 16 * J + q.c16;
      const real sc = (col < N) ? c.sg[col] : 1.0;
#pragma unroll
      for (int K = 0; K < NB; ++K) acc[K][J] = acc[K][J] * sc;
    }
    q4_flip<KS>(Y2, mask);
    q4_mul<KS>(t, Y2, acc);
    q4_flip<KS>(acc, mask);
    q4_store_glb_T<KS>(g.R_pm, q, acc);
  }
  q4_fence();
  MOM_STAMP(59);
  return true;
}

// One launch walks all layers of every unit (sweep mode only), one unit per wavefront.
// KS >= 9: one wave per SIMD (512 registers: an operator is 54 / 60 of them); smaller edges leave the register budget, and with it
// the number of units per SIMD, to the compiler (an operator of edge 20 .. 32 is 10 .. 16 registers)
#if MOM_STRIP_KS >= 9
#define MOM_Q4_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))
#else
#define MOM_Q4_ATTR
#endif
template <int KS>
__global__ void __launch_bounds__(64) MOM_Q4_ATTR k_layer_q4(const LayerArgs a) {
  using G = Q4Geom<KS>;
  constexpr int N = G::N;
  const size_t total = (size_t)a.S * a.M;
  Ctx c;
  make_ctx_q4(c, N, a.q.inv_mode, mom_smem);
  const int ns = a.q.regular ? a.q.nS : 1, Nq = N / ns;
  c.tabE = c.P;
  c.tabZS = c.P + 3 * Nq * Nq;
  real *lay = c.tabZS + 2 * ns * N;   // the layer's scalars: tau, varpi, tau_sum, K weights (dead once the elemental layer is built)
  const int nz = a.Nz_sweep;
  const int lane = wg_lane();
#ifdef MOM_DIAG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) mom_diag_last = mom_diag_now();
#endif
  load_streams(c, a.q);
  q4_fence();
  for (size_t pt = blockIdx.x; pt < total; pt += gridDim.x) {
    const int n = (int)(pt % a.S), mrel = (int)(pt / a.S), m = a.m_first + mrel;
    const size_t NNs = (size_t)N * N;
    CompPtrs g = comp_ptrs(a.comp, N, comp_pitch(N), pt);
    int done = nz;
    for (int z = 0; z < nz; ++z) {
      const int LW = 3 + a.K;
      if (lane < LW) {
        const size_t o = (size_t)n + (size_t)a.S * z;
        lay[lane] = (lane == 0) ? as_global(a.tau)[o] : (lane == 1) ? as_global(a.varpi)[o] : (lane == 2) ? as_global(a.tau_sum)[o]
                                                                                                          : as_global(a.zw)[(size_t)a.K * o + (lane - 3)];
      }
      q4_fence();
      MOM_STAMP(43);
      const int nd = a.nd_z[z];
      const bool first = (z == 0) && (a.first != 0);
      const real tau = lay[0], varpi = lay[1], tau_sum = lay[2];
      const real dtau = ldexp(tau, -nd);         // τ ./ 2^ndoubl   (rt_kernel.jl:244)
      real expk = exp(-dtau / a.q.mu0);          // init_layer      (rt_kernel.jl:273)
      ZMix zpp{as_global(a.Zpp) + NNs * a.K * mrel, lay + 3, a.K, N};
      ZMix zmp{as_global(a.Zmp) + NNs * a.K * mrel, lay + 3, a.K, N};
      elemental_build(c, a.q, m, nd, tau_sum, dtau, varpi, zpp, zmp);
      MOM_STAMP(41);
      bool bail;
      expk = doubling_run_q4<KS>(c, nd, expk, bail);
      if (!bail) {
        if (first) {
          store_added_as_composite(c, g);
          q4_fence();
          MOM_STAMP(42);
        } else {
          bail = !interaction_q4<KS>(c, g);
        }
      }
      if (bail) {  // the full image redoes this layer and finishes the unit
        done = z;
        break;
      }
    }
    if (lane == 0) a.resume[pt] = done;
  }
}

}  // namespace MOM_NS
