// mom_strip.hpp -- strip-chained products: the barrier-free inner chains of doubling and interaction.
// (Included from the middle of mom_kernels.hpp; LDS mode, N = 4 KS with two spare columns in the last tile.)
//
// The C/D layout of v_mfma_f64_16x16x4_f64 (lane l, register r: row (l>>4)+4r, column l&15) is exactly the
// B-operand layout of the same 16x16 block for its four k-steps (k = (l>>4)+4s, column l&15).  So a wave
// that owns a full Np-row x 16-column strip of a matrix X in accumulator registers can multiply it from
// the LEFT, X' = M^T X, again and again without X ever leaving its registers: no LDS round trip, no
// barrier, and column strips never interact.  Every product of the algorithm is a right-multiplication
// chain on a running quantity (A <- t + A B, then A r, (A r) t, A t; T01 <- T-- + T01 B, T01 t--, ...), so
// the chains run on the TRANSPOSED quantities: a strip holds X^T[:, c0..c0+15] = rows c0.. of X, and the
// left multiplier M^T is read from the LDS buffer of M with the conflict-free pattern M[k + row*ld].
//
// Rows >= N of a strip never enter a contraction (K = N exactly, N % 4 == 0).  They carry the source
// vectors: column N (N+1) of a multiplier's LDS buffer is row N (N+1) of M^T, so row N of the product is
// v^T X, i.e. the mat-vec products of doubling.jl:51-60 / interaction.jl:90,110 come out of the same MFMA
// stream ("riding rows", the transposed twin of the riding columns of doubling_run).
//
// Geometry per KS = N/4: NT = ceil(N/16) row tiles = column strips, LD = 16 NT + 2 (= ld_for(N)).  Strip s is
// owned by wave s of a group of four waves; the 8-wave build has two groups (two chains run concurrently in
// the interaction), the 4-wave build one (two workgroups per CU provide the overlap instead).
#pragma once

namespace MOM_NS {

#ifndef MOM_STRIP_MAXP
#define MOM_STRIP_MAXP 12
#endif
static_assert(MOM_STRIP_MAXP <= 12, "neumann_terms_12 covers p <= 12");
constexpr int kStripMaxP = MOM_STRIP_MAXP;  // series terms up to which the Horner chain beats squaring through LDS
constexpr int kStripGroups = kWaves / 4;
// k-steps whose A fragments the scheduler may have in flight at once (a scheduling barrier every kStripChunk steps):
// without it it hoists most of a product's 60 LDS reads to the front, 120 VGPRs that the fused kernel does not have
#ifndef MOM_STRIP_CHUNK
#define MOM_STRIP_CHUNK 4
#endif
constexpr int kStripChunk = MOM_STRIP_CHUNK;

// composite blocks are touched once per launch: streaming (non-temporal) accesses keep them from evicting the
// phase-matrix bases, which every unit re-reads, from L2
#ifndef MOM_NO_NT
#define MOM_NT_LOAD(p) __builtin_nontemporal_load(p)
#define MOM_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define MOM_NT_LOAD(p) (*(p))
#define MOM_NT_STORE(v, p) (*(p) = (v))
#endif

template <int KS>
struct StripGeom {
  static constexpr int N = 4 * KS;
  static constexpr int NT = (N + 15) / 16;
  static constexpr int LD = (kF64 && kWaves == 4 && (N == 36 || N == 40 || N == 44)) ? N + 2 : 16 * NT + 2;  // = ld_for(N)
  static constexpr int CP = 16 * NT;               // row pitch of the scene-level composite blocks (comp_pitch)
  // The riding rows N and N + 1 of a strip in its accumulators (tile RT).  Float64 layout (row = 16 rt + 4 r + lq): both in
  // register KS & 3, row N in the lanes lq == 0, row N + 1 in the lanes lq == 1.  Float32 layout (row = 16 rt + 4 lq + r):
  // both in the lanes lq == (N mod 16) / 4, row N in register 0, row N + 1 in register 1.
  static constexpr int RT = kF64 ? (KS >> 2) : (N / 16);
  static constexpr int RR0 = kF64 ? (KS & 3) : 0, RR1 = kF64 ? (KS & 3) : 1;
  static constexpr int LQ0 = kF64 ? 0 : ((N % 16) / 4), LQ1 = kF64 ? 1 : ((N % 16) / 4);
  // k-steps of a strip product.  Float64: the k-steps beyond N are skipped.  Float32: register r of tile rt as a B operand
  // supplies k = 16 rt + 4 lq + r -- a permutation of the contraction index that the A fragment follows -- so a k-step of
  // the last row tile reaches rows >= N; they contribute nothing because the LDS buffers keep rows N .. 16 NT - 1 at zero
  // (zero_padding), and all 4 NT k-steps run.
  static constexpr int NKS = kF64 ? KS : 4 * NT;
  static_assert(kF64 || (N % 16) != 0, "riding rows need a partly filled last row tile");
};
// contraction index of k-step ks relative to the lane's base (Float64: + lq, Float32: + 4 lq, added in the base pointer)
__device__ __forceinline__ constexpr int strip_kofs(int ks) { return kF64 ? 4 * ks : 16 * (ks >> 2) + (ks & 3); }
// row of register r of row tile rt relative to the lane's base row (Float64: lq, Float32: 4 lq)
__device__ __forceinline__ constexpr int strip_rofs(int rt, int r) { return 16 * rt + (kF64 ? 4 * r : r); }
__device__ __forceinline__ int strip_lq_base(int lq) { return kF64 ? lq : 4 * lq; }
// does (tile rt, register r) of this lane hold a row < N ?
template <int KS>
__device__ __forceinline__ bool strip_rowok(int rt, int r, int lq) {
  if constexpr (kF64) return 4 * rt + r < KS;
  else return 16 * rt + 4 * lq + r < 4 * KS;
}

// Which column strip (0 .. 3) a wave owns.  Default: wave & 3.  -DMOM_SIMD_AWARE (experiment, profiles/r05_mid_ab.txt; 4-wave
// build): from the SIMD the wave sits on and the workgroup's slot on the CU (HW_REG_HW_ID: simd_id bits 5:4, tg_id bits 19:16), so
// that the idle fourth wave (NT = 3) of co-resident workgroups falls on different SIMDs.  c.slot is set (and validated: the four
// waves of the workgroup must sit on four distinct SIMDs, else wave & 3) once in the prologue.
#define MOM_GETREG(id, off, size) ((((size)-1) << 11) | ((off) << 6) | (id))
__device__ __forceinline__ int strip_slot(const Ctx &c, int wave) {
#ifdef MOM_SIMD_AWARE
  if (kWaves == 4) return __builtin_amdgcn_readfirstlane(c.slot);
#endif
  return wave & 3;
}
__device__ __forceinline__ void strip_slot_init(Ctx &c) {
#ifdef MOM_SIMD_AWARE
  if (kWaves == 4) {
    const int wave = wg_wave();
    const unsigned simd = __builtin_amdgcn_s_getreg(MOM_GETREG(4, 4, 2)), tg = __builtin_amdgcn_s_getreg(MOM_GETREG(4, 16, 4));
    const int mine = (int)((simd + tg) & 3u);
    if (wg_lane() == 0) c.ipiv[wave] = mine;
    __syncthreads();
    unsigned seen = 0;
    for (int w = 0; w < 4; ++w) seen |= 1u << c.ipiv[w];
    c.slot = (seen == 15u) ? mine : wave;
    __syncthreads();
  }
#endif
}

__device__ __forceinline__ r4 mfma_f64(real a, real b, r4 c) {
  return mma16(a, b, c);
}

// acc[rt] += sum_k M[k + row*LD] B[k][col], row = 16 rt + (l & 15): left-multiplication of the strip B by M^T
template <int KS>
__device__ __forceinline__ void strip_mul(const real *M, int lr, int lq, const r4 (&B)[StripGeom<KS>::NT],
                                          r4 (&acc)[StripGeom<KS>::NT]) {
  constexpr int NT = StripGeom<KS>::NT, LD = StripGeom<KS>::LD;
  asm volatile("" : "+v"(lr), "+v"(lq));  // keep the address arithmetic inside (see item_straight)
  const real *base = M + strip_lq_base(lq) + lr * LD;
#pragma unroll
  for (int ks = 0; ks < StripGeom<KS>::NKS; ++ks) {
    real a[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) a[rt] = base[strip_kofs(ks) + 16 * rt * LD];
#ifdef MOM_DIAG_MASKB  // (diagnostic builds: strip rows >= N never reach the product -- a no-op while the multiplier's rows >= N are zero)
    const real b = strip_rowok<KS>(ks >> 2, ks & 3, lq) ? B[ks >> 2][ks & 3] : (real)0;
#else
    const real b = B[ks >> 2][ks & 3];
#endif
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) acc[rt] = mfma_f64(a[rt], b, acc[rt]);
    if ((ks + 1) % kStripChunk == 0) __builtin_amdgcn_sched_barrier(0);  // cap the A fragments in flight
  }
}

// two strips through the same multiplier: acc1 += M^T B1, acc2 += M^T B2 (A fragments loaded once)
template <int KS>
__device__ __forceinline__ void strip_mul2(const real *M, int lr, int lq, const r4 (&B1)[StripGeom<KS>::NT],
                                           r4 (&acc1)[StripGeom<KS>::NT], const r4 (&B2)[StripGeom<KS>::NT],
                                           r4 (&acc2)[StripGeom<KS>::NT]) {
  constexpr int NT = StripGeom<KS>::NT, LD = StripGeom<KS>::LD;
  asm volatile("" : "+v"(lr), "+v"(lq));
  const real *base = M + strip_lq_base(lq) + lr * LD;
#pragma unroll
  for (int ks = 0; ks < StripGeom<KS>::NKS; ++ks) {
    real a[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) a[rt] = base[strip_kofs(ks) + 16 * rt * LD];
    const real b1 = B1[ks >> 2][ks & 3], b2 = B2[ks >> 2][ks & 3];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      acc1[rt] = mfma_f64(a[rt], b1, acc1[rt]);
      acc2[rt] = mfma_f64(a[rt], b2, acc2[rt]);
    }
    if ((ks + 1) % kStripChunk == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT>
__device__ __forceinline__ void strip_zero(r4 (&W)[NT]) {
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) W[rt] = (r4){0.0, 0.0, 0.0, 0.0};
}
template <int NT>
__device__ __forceinline__ void strip_copy(r4 (&D)[NT], const r4 (&S)[NT]) {
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) D[rt] = S[rt];
}

// W[row][col] = X[col][row] for X column-major (LDS buffer: leading dimension LD; global composite block:
// N); rows >= N read as zero.  col = c0 + lr may run past N - 1 (garbage columns, never stored; an LDS
// buffer has LD > 16 NT - 1 rows, a global block is guarded with colok).
template <int KS>
__device__ __forceinline__ void strip_load_lds(const real *X, int lr, int lq, int c0, r4 (&W)[StripGeom<KS>::NT]) {
  asm volatile("" : "+v"(lr), "+v"(lq));
  constexpr int NT = StripGeom<KS>::NT, LD = StripGeom<KS>::LD;
  const real *base = X + c0 + lr + strip_lq_base(lq) * LD;
#pragma unroll
  for (int rt = 0; rt < NT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) W[rt][r] = strip_rowok<KS>(rt, r, lq) ? base[strip_rofs(rt, r) * LD] : 0.0;
}
template <int KS>
__device__ __forceinline__ void strip_store_lds(real *X, int lr, int lq, int c0, bool colok,
                                                const r4 (&W)[StripGeom<KS>::NT]) {
  asm volatile("" : "+v"(lr), "+v"(lq));
  constexpr int NT = StripGeom<KS>::NT, LD = StripGeom<KS>::LD;
  real *base = X + c0 + lr + strip_lq_base(lq) * LD;
  if (colok) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (strip_rowok<KS>(rt, r, lq)) base[strip_rofs(rt, r) * LD] = W[rt][r];
  }
}
template <int KS>
__device__ __forceinline__ void strip_load_glb(const gdouble *__restrict__ X, int lr, int lq, int c0, bool colok,
                                               r4 (&W)[StripGeom<KS>::NT]) {
  asm volatile("" : "+v"(lr), "+v"(lq));
  constexpr int NT = StripGeom<KS>::NT, CP = StripGeom<KS>::CP;
  const gdouble *base = X + c0 + lr + strip_lq_base(lq) * CP;
#pragma unroll
  for (int rt = 0; rt < NT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) W[rt][r] = (strip_rowok<KS>(rt, r, lq) && colok) ? MOM_NT_LOAD(base + strip_rofs(rt, r) * CP) : 0.0;
}
template <int KS>
__device__ __forceinline__ void strip_store_glb(gdouble *__restrict__ X, int lr, int lq, int c0, bool colok,
                                                const r4 (&W)[StripGeom<KS>::NT]) {
  asm volatile("" : "+v"(lr), "+v"(lq));
  constexpr int NT = StripGeom<KS>::NT, CP = StripGeom<KS>::CP;
  gdouble *base = X + c0 + lr + strip_lq_base(lq) * CP;
  if (colok) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (strip_rowok<KS>(rt, r, lq)) MOM_NT_STORE(W[rt][r], base + strip_rofs(rt, r) * CP);
  }
}

// bit 4 rt + r of the mask: sg[row] < 0 for row = 16 rt + 4 r + lq (this lane's strip rows)
__device__ __forceinline__ unsigned strip_sign_mask(const real *sg, int lq, int N) {
  unsigned m = 0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int row = 16 * (b >> 2) + cd_row(lq, b & 3);   // bit 4 rt + r <-> (tile rt, register r)
    if (row < N && sg[row] < 0.0) m |= 1u << b;
  }
  return m;
}
// W <- diag(sg) W  (row signs)
template <int NT>
__device__ __forceinline__ void strip_flip(r4 (&W)[NT], unsigned mask) {
#pragma unroll
  for (int rt = 0; rt < NT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real v = W[rt][r];
      W[rt][r] = ((mask >> (4 * rt + r)) & 1u) ? -v : v;
    }
}

// ---------------------------------------------------------------------------------------
// P = r r with the riding columns N, N+1 = r j0+, r j0- (columns N, N+1 of c.r hold j0+, j0-), and the
// wave-partial ||r r||_F^2 for wg_sumsq_get: the strip form of the first product of a doubling step.
// (P^T strip = r^T (r^T strip); the stored layout is the one the general path produces.)  Needs a barrier after.
// ---------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void doubling_rr_strip(const Ctx &c) {
  using G = StripGeom<KS>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD;
  const int lane = wg_lane(), wave = wg_wave(), lr = lane & 15, lq = lane >> 4;
  const int slot = strip_slot(c, wave);
  const int c0 = 16 * slot, col = c0 + lr;
  const bool active = (wave >> 2) == 0 && slot < NT, colok = col < N;
  real ss = 0.0;
  if (active) {
    r4 W[NT], B[NT];
    strip_load_lds<KS>(c.r, lr, lq, c0, W);
    strip_zero(B);
    strip_mul<KS>(c.r, lr, lq, W, B);
    strip_store_lds<KS>(c.P, lr, lq, c0, colok, B);
    if (colok) {
      if (lq == G::LQ0) c.P[col + N * LD] = B[G::RT][G::RR0];        // (r j0+)[col]
      if (lq == G::LQ1) c.P[col + (N + 1) * LD] = B[G::RT][G::RR1];  // (r j0-)[col]
#pragma unroll
      for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (strip_rowok<KS>(rt, r, lq)) ss += B[rt][r] * B[rt][r];
    }
  }
  wg_sumsq_put(c, ss);
}

// ---------------------------------------------------------------------------------------
// One doubling step (doubling.jl:44-67) given P = r r (+ riding columns r j0+, r j0-) and p series terms.
// The waves of group 0 own the column strips; the others only keep the barriers.  In: c.r, c.t, c.P, c.jp,
// c.jm; out: c.r, c.t, c.jp, c.jm and the riding columns N, N+1 of c.r (= new j0+, j0-) for the next step.
// ---------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void doubling_step_strip(Ctx &c, int p, real expk, const CompPtrs *pre = nullptr) {
  using G = StripGeom<KS>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD;
  const int lane = wg_lane(), wave = wg_wave(), lr = lane & 15, lq = lane >> 4;
  const int slot = strip_slot(c, wave);
  const int c0 = 16 * slot;
  const bool active = (wave >> 2) == 0 && slot < NT, colok = c0 + lr < N;
  real *r = c.r, *t = c.t;
  const real *P = c.P;
  r4 Rn[NT], Tn[NT];
  real aw = 0.0, aw2 = 0.0;  // (A w1)[col] in the lanes lq == LQ0, (A w2)[col] in the lanes lq == LQ1
#ifdef MOM_CHAIN_PRIO
  if (active) __builtin_amdgcn_s_setprio(MOM_CHAIN_PRIO);  // experiment (profiles/r04_C2_ab.txt): static priority of the chain waves
#endif
  if (active) {
    // riding rows of the multiplier r^T: w1 = j1- + r j0+, w2 = j0+ + r j1-  (doubling.jl:51-60); every strip wave
    // writes the same values, so each reads back its own writes in order
    if (lane < N) {
      r[lane + N * LD] = c.jm[lane] * expk + P[lane + N * LD];
      r[lane + (N + 1) * LD] = c.jp[lane] + expk * P[lane + (N + 1) * LD];
    }
    r4 T0[NT], Y[NT];
    strip_load_lds<KS>(t, lr, lq, c0, T0);
    strip_copy(Y, T0);
    // Y = A^T = (t (I - r r)^-1)^T by Horner: Y <- t^T + (r r)^T Y
#pragma nounroll
    for (int k = 1; k < p; ++k) {
      r4 acc[NT];
      strip_copy(acc, T0);
      strip_mul<KS>(P, lr, lq, Y, acc);
      strip_copy(Y, acc);
    }
    r4 Zt[NT];
    strip_zero(Zt);
    strip_mul<KS>(r, lr, lq, Y, Zt);  // (A r)^T ; rows N, N+1: (A w1)^T, (A w2)^T
    aw = Zt[G::RT][G::RR0];   // riding row N
    aw2 = Zt[G::RT][G::RR1];  // riding row N + 1
    // two single-strip products rather than one pass over t with shared A fragments (strip_mul2): the real product
    // needs 32 more live VGPRs, which the fused kernel pays for with spills inside the chain
    strip_load_lds<KS>(r, lr, lq, c0, Rn);
    strip_mul<KS>(t, lr, lq, Zt, Rn);  // r^T + t^T (A r)^T      (:64)
    strip_zero(Tn);
    strip_mul<KS>(t, lr, lq, Y, Tn);   // t^T A^T                (:67)
  }
#ifdef MOM_QPREFETCH
  // experiment (profiles/r04_C2_ab.txt): the waves that idle during the chains of the LAST doubling step fetch the coming
  // interaction's T++ block (+ J0+ as its riding column) into Q, which no strip step uses
  // (4-wave build: the idle waves are the ones without a strip, slot >= NT -- on a SIMD of their own, no chain wave to disturb)
  constexpr int kIdle0 = (kWaves == 8) ? 4 : NT;
#if MOM_QPREFETCH == 2
  // r5 variant: the idle waves REQUEST T++ during the chains and keep it in registers; the LDS stores into Q happen after the
  // chain barrier, next to the strip waves' write-back (r4's variant stored during the chains and lost 1.3 % to the LDS /
  // issue slots it took from the chain wave of the same SIMD)
  constexpr int kPreTH = (kIdle0 < kWaves) ? 64 * (kWaves - kIdle0) : 64, kPreU = (N * N + kPreTH - 1) / kPreTH;
  real pvt[kPreU], pvj = 0.0;
  const bool prefetcher = pre != nullptr && kIdle0 < kWaves && (kWaves == 8 ? (wave >> 2) == 1 : slot >= NT);
  const int ptid = (kWaves == 8) ? wg_tid() - 256 : 64 * (slot - NT) + lane;
  if (prefetcher) {
    constexpr int TH = kPreTH;
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int e = ptid + u * TH;
      if (e < N * N) {
        int i, j;
        c.fd.split(e, i, j);
        pvt[u] = MOM_NT_LOAD(pre->T_pp + i + j * G::CP);
      }
    }
    if (ptid < N) pvj = pre->J0p[ptid];
  }
  if (false) {
#else
  if (pre != nullptr && kIdle0 < kWaves && (kWaves == 8 ? (wave >> 2) == 1 : slot >= NT)) {
#endif
    constexpr int NN = N * N, U = 8, TH = 64 * (kWaves - kIdle0);
    real *Q = c.Q;
    const int tid4 = (kWaves == 8) ? wg_tid() - 256 : 64 * (slot - NT) + lane;
    for (int e0 = tid4; e0 < NN; e0 += U * TH) {
      real vt[U];
      int o[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * TH;
        if (e < NN) {
          int i, j;
          c.fd.split(e, i, j);
          vt[u] = MOM_NT_LOAD(pre->T_pp + i + j * G::CP);
          o[u] = i + j * LD;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u * TH < NN) Q[o[u]] = vt[u];
    }
    for (int i = tid4; i < N; i += TH) Q[i + N * LD] = pre->J0p[i];
  }
  if (pre != nullptr && kIdle0 < kWaves) c.qpre = 1;
#endif
#ifdef MOM_CHAIN_PRIO
  if (active) __builtin_amdgcn_s_setprio(0);
#endif
  MOM_STAMP(71);
  __syncthreads();
  MOM_STAMP(72);
#if defined(MOM_QPREFETCH) && MOM_QPREFETCH == 2
  if (prefetcher) {
    constexpr int TH = kPreTH;
    real *Q = c.Q;
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int e = ptid + u * TH;
      if (e < N * N) {
        int i, j;
        c.fd.split(e, i, j);
        Q[i + j * LD] = pvt[u];
      }
    }
    if (ptid < N) Q[ptid + N * LD] = pvj;
  }
#endif
  if (active) {
    strip_store_lds<KS>(r, lr, lq, c0, colok, Rn);
    strip_store_lds<KS>(t, lr, lq, c0, colok, Tn);
    const int col = c0 + lr;
    if (colok && lq == G::LQ0) {  // j0- += A w1 (:57)
      const real jm = c.jm[col] + aw;
      c.jm[col] = jm;
      r[col + (N + 1) * LD] = jm;
    }
    if (colok && lq == G::LQ1) {  // j0+ = j1+ + A w2 (:60)
      const real jp = c.jp[col] * expk + aw2;
      c.jp[col] = jp;
      r[col + N * LD] = jp;
    }
  }
  __syncthreads();
  MOM_STAMP(73);
}

// ---------------------------------------------------------------------------------------
// ScatteringInterface_11 (interaction.jl:69-117) as two strip chains: chain 1 = T01 (T--, R-+, J0-), chain 2 =
// T21 (T++, R+-, J0+).  8-wave build: waves 0..3 run chain 1 while waves 4..7 run chain 2; 4-wave build: the
// strip waves run one after the other.  B = r-+ R+- is formed once through LDS (exact Frobenius norm for the
// series length); the second inverse is expressed through the first,
//   T21 R+- = t++ R+- (I - B)^-1 =: X,   T21 = t++ + X r-+        (push-through identity),
// so both chains iterate with the same multiplier B^T.  Added layer in c.r (r-+), c.t (t++), c.jp, c.jm with
// r+- = D r-+ D, t-- = D t++ D.  Returns false (nothing stored yet) if the series is too long: the caller
// then runs the general path.  Ends with a barrier.
// ---------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ bool interaction_strip(Ctx &c, const CompPtrs &g) {
  using G = StripGeom<KS>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD;
  const int lane = wg_lane(), wave = wg_wave(), lr = lane & 15, lq = lane >> 4;
  const int slot = strip_slot(c, wave);
  const int grp = wave >> 2, c0 = 16 * slot, col = c0 + lr;
  const bool strip = slot < NT, colok = col < N;
  const bool do1 = strip && grp == 0, do2 = strip && grp == kStripGroups - 1;
  real *r = c.r, *t = c.t, *P = c.P, *Q = c.Q;
  r4 T1[NT], W0[NT];  // chain 1: T--^T strip ; chain 2: W0 = R+-^T t++^T
  const bool qpre = c.qpre != 0;  // MOM_QPREFETCH: Q = T++ (+ J0+) was fetched during the last doubling step
  c.qpre = 0;
  if (do1) strip_load_glb<KS>(g.T_mm, lr, lq, c0, colok, T1);
  // P = R+-, Q = T++ ; riding rows: column N of Q = J0+, column N of r = j0-.  All global loads of both blocks are
  // issued before the first LDS store: one exposed HBM latency instead of one per batch.
  {
    constexpr int NN = N * N, U = 8;
    for (int e0 = wg_tid(); e0 < NN; e0 += U * kThreads) {
      real vr[U], vt[U];
      int o[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        if (e < NN) {
          int i, j;
          c.fd.split(e, i, j);
          vr[u] = MOM_NT_LOAD(g.R_pm + i + j * G::CP);
          if (!qpre) vt[u] = MOM_NT_LOAD(g.T_pp + i + j * G::CP);
          o[u] = i + j * LD;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u * kThreads < NN) {
          P[o[u]] = vr[u];
          if (!qpre) Q[o[u]] = vt[u];
        }
    }
  }
  for (int i = wg_tid(); i < N; i += kThreads) {
    if (!qpre) Q[i + N * LD] = g.J0p[i];
    r[i + N * LD] = c.jm[i];
  }
  MOM_STAMP(50);
  __syncthreads();
  MOM_STAMP(51);
  MOM_STAMP4(91);
  // B = r-+ R+- on strips (B^T strip = R+-^T (r-+^T strip)) by the chain-1 waves while the chain-2 waves form W0;
  // both read R+- in P, which B then replaces (after the barrier), with ||B||_F^2 for the series length
  r4 Bs[NT];
  real ss = 0.0;
  if (do1) {
    r4 rT[NT];
    strip_load_lds<KS>(r, lr, lq, c0, rT);
    strip_zero(Bs);
    strip_mul<KS>(P, lr, lq, rT, Bs);
    if (colok) {
#pragma unroll
      for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (strip_rowok<KS>(rt, rr, lq)) ss += Bs[rt][rr] * Bs[rt][rr];
    }
  }
  if (do2) {
    r4 tT[NT];
    strip_load_lds<KS>(t, lr, lq, c0, tT);
    strip_zero(W0);
    strip_mul<KS>(P, lr, lq, tT, W0);
  }
  MOM_STAMP4(92);
  MOM_STAMP(52);
  __syncthreads();
  if (do1) strip_store_lds<KS>(P, lr, lq, c0, colok, Bs);
  wg_sumsq_put(c, ss);
  __syncthreads();
  MOM_STAMP(53);
  MOM_STAMP4(93);
  const real beta2 = wg_sumsq_get(c);
  const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(beta2));  // workgroup-uniform: scalar loop control
  if (p > kStripMaxP) return false;
  const unsigned mask = strip_sign_mask(c.sg, lq, N);
  if (do1) {
    r4 Y[NT];
    strip_copy(Y, T1);
#pragma nounroll
    for (int k = 1; k < p; ++k) {  // Y <- T--^T + B^T Y : T01^T
      r4 acc[NT];
      strip_copy(acc, T1);
      strip_mul<KS>(P, lr, lq, Y, acc);
      strip_copy(Y, acc);
    }
    MOM_STAMP(54);
    r4 Radd[NT];
    strip_load_glb<KS>(g.R_mp, lr, lq, c0, colok, Radd);
    const real j0m = (colok && lq == G::LQ0) ? g.J0m[col] : 0.0;
    // T-- = T01 t--  ->  (t--)^T T01^T = D t^T D Y                                   (:96)
    {
      r4 Yf[NT], o[NT];
      strip_copy(Yf, Y);
      strip_flip(Yf, mask);
      strip_zero(o);
      strip_mul<KS>(t, lr, lq, Yf, o);
      strip_flip(o, mask);
      strip_store_glb<KS>(g.T_mm, lr, lq, c0, colok, o);
    }
    // V = (T01 r-+)^T = r-+^T Y ; row N: (T01 j0-)^T
    r4 V[NT];
    strip_zero(V);
    strip_mul<KS>(r, lr, lq, Y, V);
    // R-+ = R-+ + (T01 r-+) T++  ->  R-+^T + T++^T V ; row N: (T01 r-+ J0+)^T        (:93)
    strip_mul<KS>(Q, lr, lq, V, Radd);
    strip_store_glb<KS>(g.R_mp, lr, lq, c0, colok, Radd);
    // J0- = J0- + T01 (r-+ J0+ + j0-)                                                (:90)
    if (colok && lq == G::LQ0) g.J0m[col] = j0m + (Radd[G::RT][G::RR0] + V[G::RT][G::RR0]);
    MOM_STAMP(55);
  }
  if (do2) {
    r4 Y[NT];
    strip_copy(Y, W0);
#pragma nounroll
    for (int k = 1; k < p; ++k) {  // Y <- W0 + B^T Y : X^T
      r4 acc[NT];
      strip_copy(acc, W0);
      strip_mul<KS>(P, lr, lq, Y, acc);
      strip_copy(Y, acc);
    }
    MOM_STAMP4(94);
    if (kWaves == 4) MOM_STAMP(58);  // (4-wave build: the same waves run chain 2 after chain 1)
    // T21^T = t++^T + r-+^T X^T ; row N: (X j0-)^T = (T21 R+- j0-)^T
    r4 T21[NT];
    strip_load_lds<KS>(t, lr, lq, c0, T21);
    strip_mul<KS>(r, lr, lq, Y, T21);
    // T++ = T21 T++  ->  T++^T T21^T ; row N: (T21 J0+)^T                            (:113)
    r4 o[NT];
    strip_zero(o);
    strip_mul<KS>(Q, lr, lq, T21, o);
    strip_store_glb<KS>(g.T_pp, lr, lq, c0, colok, o);
    // J0+ = j0+ + T21 (J0+ + R+- j0-)                                                (:110)
    if (colok && lq == G::LQ0) g.J0p[col] = c.jp[col] + (o[G::RT][G::RR0] + T21[G::RT][G::RR0]);
    // R+- = r+- + X t--  ->  r+-^T + D t^T D X^T = D (D r+-^T + t^T D X^T), D r+-^T[row][col] = sg[col] r-+[col][row]   (:116)
    r4 acc[NT];
    strip_load_lds<KS>(r, lr, lq, c0, acc);
    const real sc = colok ? c.sg[col] : 1.0;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) acc[rt] = acc[rt] * sc;
    strip_flip(Y, mask);
    strip_mul<KS>(t, lr, lq, Y, acc);
    strip_flip(acc, mask);
    strip_store_glb<KS>(g.R_pm, lr, lq, c0, colok, acc);
    MOM_STAMP4(95);
    if (kWaves == 4) MOM_STAMP(59);
  }
  __syncthreads();
  MOM_STAMP(56);
  MOM_STAMP4(96);
  return true;
}

}  // namespace MOM_NS
