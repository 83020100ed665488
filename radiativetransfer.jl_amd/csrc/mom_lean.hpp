// mom_lean.hpp -- the LEAN 4-wave strip image (r5): THREE operator buffers instead of four and a 168-register budget, so that
// THREE workgroups share a CU (N = 36, 40: the m = 0 launch of the headline scene, N0 = 40).
//
// Why (profiles/r05_mid_ab.txt): with 3 x 3 tiles a 4-wave workgroup runs its chains on three waves; two co-resident workgroups
// load the SIMDs (2,1,1,2) and the MFMA pipes of the two shared SIMDs are 71 % busy -- the 29 % are the elemental layer, the
// composite loads and the barriers of the workgroups, which only MORE independent units per CU can cover.  A third workgroup
// needs <= 53.3 KB of LDS and <= 168 VGPRs.  The full image (mom_entry.hpp k_layer) holds r, t, P, Q (4 x 14.1 KB at pitch 42)
// + 9.3 KB of vectors and carries the general doubling / interaction code (pivoted inverse, long series, the interface cases
// 00 / 01 / 10) that needs all four buffers and 253 registers.  This image:
//   * buffers r, t, P only (42.3 KB at N = 40) + vectors with `part` cut to its 16 norm slots + a 128-real layer tail: 48.5 KB;
//   * the elemental layer's tables (E, F1, F2 and the sun-block columns) all in P (elemental_build takes them from Ctx);
//   * doubling: the strip-chained step only (mom_strip.hpp);
//   * interaction (interface 11): interaction_strip's two chains on the same three waves, with T++ arriving in P AFTER the two
//     Horner loops have finished with B -- fetched from global memory into registers by the IDLE fourth wave while the strip waves
//     iterate (the one thing that wave can do for free: it has no other live state), two barriers more than the full image;
//   * anything else -- a series beyond kStripMaxP terms, a forced pivoted inverse -- is not computed here: the workgroup records
//     the layer in resume[unit] and leaves the unit; nothing of that layer has reached the composite state in global memory, so a
//     second launch of the FULL image (LayerArgs::resume) redoes the layer and carries the unit to the end.  A unit that
//     completes records Nz.  (C2: no unit ever leaves.)
// Scope: Float64, 4-wave build, sweep mode, single composite (no multi-target), interface 11 on every layer after the first.
#pragma once
#include "mom_entry.hpp"

namespace MOM_NS {

constexpr int kLeanLay = 128;  // reals of the layer-scalar tail
// The SIX-wave flavour (-DMOM_WAVES=6, namespace mom6; "lean6"): the doubling chains run on HALF-STRIPS -- the two waves 2 s, 2 s + 1
// of strip s split the contraction (k-steps 0 .. KH - 1 and KH .. KS - 1), each multiplies its half of the running strip into
// partial sums of the whole strip (15 MFMAs instead of 30 at N = 40) and the two exchange, through a 3 KB LDS slot per wave, the
// partial sums of the rows the OTHER one owns.  Six chain waves per unit, two units per CU: three chains on every SIMD -- the
// 54 ms floor of wave-owned strips (profiles/r05_mid_ab.txt: 1/3 of the MFMA time on the busiest SIMD) becomes 1/4.
// MEASURED (same file): 104.5 ms against the four-wave lean image's 73 ms on the headline's m = 0 launch -- a workgroup's six waves
// sit (2,2,1,1) on the SIMDs, so ITS product still takes two half-strip times on two of them (1 920 cycles, what a full strip
// takes); the gain has to come from the second workgroup filling the other SIMDs in phase, and the two workgroup barriers per
// product (partial sums must meet) cost more than those 25 %.  Kept as MOM_OPT_LEAN = 2 (tested), not the default.
constexpr bool kLean6 = (kWaves == 6);
constexpr int kHalfSlots = 6;   // k-step slots of a half-strip: up to 5 k-steps + the riding rows
__host__ __device__ inline size_t lean_vec_reals(int N) { return part_offset_doubles(N) + 16; }
__host__ __device__ inline size_t lean_exch_reals() { return kLean6 ? (size_t)kWaves * kHalfSlots * 64 : 0; }
__host__ __device__ inline size_t lean_lds_bytes(int N) {
  return (3 * mat_elems(N) + lean_vec_reals(N) + lean_exch_reals() + kLeanLay) * sizeof(real);
}
// does the image apply to operators of edge N with ns Stokes components per stream?  (the tables must fit P; three workgroups
// of four waves or two of six per CU)
__host__ __device__ inline bool lean_applies(int N, int ns) {
  const int Nq = N / (ns > 0 ? ns : 1);
  const int per_cu = kLean6 ? 2 : 3;
  return kF64 && (kWaves == 4 || kLean6) && (N == 36 || N == 40) && 3 * Nq * Nq + 2 * ns * N <= (int)mat_elems(N) &&
         per_cu * (lean_lds_bytes(N) + 1024) <= kLdsPerCU;
}

__device__ __forceinline__ void make_ctx_lean(Ctx &c, int N, int inv_mode, real *smem) {
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
  c.r = smem; c.t = smem + msz; c.P = smem + 2 * msz; c.Q = nullptr; c.X = nullptr;
  real *p = smem + 3 * msz;
  const int lv = c.ldv;
  c.jp = p; c.jm = p + lv; c.j1p = p + 2 * lv; c.j1m = p + 3 * lv; c.v1 = p + 4 * lv; c.v2 = p + 5 * lv;
  c.Jp = p + 6 * lv; c.Jm = p + 7 * lv; c.prow = p + 8 * lv; c.pcol = p + 9 * lv; c.rowk = p + 10 * lv;
  c.ei = p + 11 * lv; c.mu = p + 12 * lv; c.wt = p + 13 * lv; c.sg = p + 14 * lv;
  c.thr = p + 15 * lv;
  int *ip = reinterpret_cast<int *>(c.thr + 32);
  c.ipiv = ip; c.sh = ip + lv; c.bad = ip + lv + 1;
  c.part = p + part_offset_doubles(N);
}

// nd strip-chained doubling steps; bail = true (nothing of the layer has left the workgroup) if a step needs the general path
template <int KS>
__device__ __forceinline__ real doubling_run_lean(Ctx &c, int nd, real expk, bool &bail) {
  const int N = c.N, ld = c.ld;
  bail = false;
  if (nd == 0) return expk;
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.r[i + N * ld] = c.jp[i];
    c.r[i + (N + 1) * ld] = c.jm[i];
  }
  __syncthreads();
  for (int it = 0; it < nd; ++it) {
    doubling_rr_strip<KS>(c);
    __syncthreads();
    const real beta2 = wg_sumsq_get(c);
    const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(beta2));
    if (p > kStripMaxP || c.inv_mode != 0) {
      bail = true;
      return expk;
    }
    doubling_step_strip<KS>(c, p, expk);
    expk = expk * expk;
  }
  // apply_D! (doubling.jl:93-110) and apply_D_SFI! (:112-118): r-+ rows and j0- scaled by sg
  real *r = c.r;
  for (int e = wg_tid(); e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    r[i + j * ld] *= c.sg[i];
  }
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.jm[i] *= c.sg[i];
    r[i + N * ld] = 0.0; r[i + (N + 1) * ld] = 0.0; c.P[i + N * ld] = 0.0; c.P[i + (N + 1) * ld] = 0.0;
  }
  __syncthreads();
  return expk;
}

// ---------------------------------------------------------------------------------------------------------------------
// lean6: half-strip chains.  Wave w = 2 s + h (s = column strip, h = half).  A half-strip holds, per lane, the rows of the
// running strip that are the B operand of ITS k-steps: slot j <-> k-step ks0 + j, i.e. register (ks & 3) of row tile (ks >> 2)
// of the full strip's accumulator layout; half 1 also owns the slot of the riding rows N, N + 1 (k-step index KS).
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct HalfGeom {
  static constexpr int KH = (KS + 1) / 2;           // k-steps of half 0 (half 1: KS - KH)
  static_assert(KH <= kHalfSlots - 1 && KS - KH + 1 <= kHalfSlots, "slots of a half-strip");
};
// Every function below takes the half H as a TEMPLATE argument: the slots of a half map to accumulator registers
// acc[ks >> 2][ks & 3] with ks = ks0(H) + j, and a register index must be a compile-time constant (with H as a run-time value
// the first version of this image indexed its accumulators through scratch and ran 2.3 x slower than the four-wave one).
template <int KS, int H> struct HalfOf {
  static constexpr int KH = HalfGeom<KS>::KH;
  static constexpr int ks0 = H ? KH : 0;               // first k-step
  static constexpr int nk = H ? KS - KH : KH;          // k-steps (slots that are B operands)
  static constexpr int ns = H ? KS - KH + 1 : KH;      // owned slots (half 1: + the riding rows, k-step index KS)
};

// x[j] = X[col][row] (the transposed strip of the column-major LDS buffer X), rows of my k-steps; other slots read 0
template <int KS, int H>
__device__ __forceinline__ void half_load_lds(const real *X, int lr, int lq, int c0, real (&x)[kHalfSlots]) {
  constexpr int LD = StripGeom<KS>::LD;
  using O = HalfOf<KS, H>;
  const real *base = X + c0 + lr + lq * LD;
#pragma unroll
  for (int j = 0; j < kHalfSlots; ++j) x[j] = (j < O::nk) ? base[4 * (O::ks0 + j) * LD] : 0.0;
}
template <int KS, int H>
__device__ __forceinline__ void half_store_lds(real *X, int lr, int lq, int c0, bool colok, const real (&x)[kHalfSlots]) {
  constexpr int LD = StripGeom<KS>::LD;
  using O = HalfOf<KS, H>;
  real *base = X + c0 + lr + lq * LD;
  if (colok) {
#pragma unroll
    for (int j = 0; j < kHalfSlots; ++j)
      if (j < O::nk) base[4 * (O::ks0 + j) * LD] = x[j];
  }
}
// acc[rt] += sum over MY k-steps of M^T fragments x my slots
template <int KS, int H>
__device__ __forceinline__ void half_mul(const real *M, int lr, int lq, const real (&x)[kHalfSlots], r4 (&acc)[StripGeom<KS>::NT]) {
  constexpr int NT = StripGeom<KS>::NT, LD = StripGeom<KS>::LD;
  using O = HalfOf<KS, H>;
  asm volatile("" : "+v"(lr), "+v"(lq));
  const real *base = M + lq + lr * LD + 4 * O::ks0;
#pragma unroll
  for (int j = 0; j < O::nk; ++j) {
    real a[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) a[rt] = base[4 * j + 16 * rt * LD];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) acc[rt] = mma16(a[rt], x[j], acc[rt]);
  }
}
// partial sums of the whole strip -> the summed values of MY slots: write the partner's slots, barrier, read mine, add.
// All six waves call it together (the two barriers of a call are the only synchronisation of a product).
template <int KS, int H>
__device__ __forceinline__ void half_reduce(real *exch, int wave, int lane, const r4 (&acc)[StripGeom<KS>::NT], real (&out)[kHalfSlots]) {
  using O = HalfOf<KS, H>;
  using P = HalfOf<KS, 1 - H>;
  real *mine = exch + (size_t)wave * kHalfSlots * 64 + lane, *theirs = exch + (size_t)(wave ^ 1) * kHalfSlots * 64 + lane;
  __syncthreads();   // every wave has read what the previous exchange left in the slots
#pragma unroll
  for (int j = 0; j < P::ns; ++j) {
    constexpr int dummy = 0; (void)dummy;
    const int ks = P::ks0 + j;
    mine[j * 64] = acc[ks >> 2][ks & 3];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kHalfSlots; ++j) {
    const int ks = O::ks0 + j;
    out[j] = (j < O::ns) ? acc[(j < O::ns ? ks : 0) >> 2][(j < O::ns ? ks : 0) & 3] + theirs[j * 64] : 0.0;
  }
}

// one doubling step (doubling.jl:44-67) on the half-strip of this wave; returns false before anything is changed if the series is
// too long.  r, t, P, jp, jm as in doubling_step_strip.
template <int KS, int H>
__device__ __forceinline__ bool doubling_step_half(Ctx &c, real *exch, real expk) {
  using G = StripGeom<KS>;
  using O = HalfOf<KS, H>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD, RS = O::ns - 1;  // RS: the riding slot (half 1 only)
  const int lane = wg_lane(), wave = wg_wave(), lr = lane & 15, lq = lane >> 4;
  const int c0 = 16 * (wave >> 1), col = c0 + lr;
  const bool colok = col < N;
  real *r = c.r, *t = c.t, *P = c.P;
  // ---- P = r r (+ riding columns r j0+, r j0-) and ||r r||_F^2
  real rh[kHalfSlots], ph[kHalfSlots];
  half_load_lds<KS, H>(r, lr, lq, c0, rh);
  {
    r4 acc[NT];
    strip_zero(acc);
    half_mul<KS, H>(r, lr, lq, rh, acc);
    half_reduce<KS, H>(exch, wave, lane, acc, ph);
  }
  real ss = 0.0;
  half_store_lds<KS, H>(P, lr, lq, c0, colok, ph);
  if (colok) {
    if (H) {  // the riding rows: (r j0+)[col] in the lanes lq == 0, (r j0-)[col] in lq == 1
      if (lq == G::LQ0) P[col + N * LD] = ph[RS];
      if (lq == G::LQ1) P[col + (N + 1) * LD] = ph[RS];
    }
#pragma unroll
    for (int j = 0; j < O::nk; ++j) ss += ph[j] * ph[j];
  }
  wg_sumsq_put(c, ss);
  __syncthreads();
  const real beta2 = wg_sumsq_get(c);
  const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(beta2));
  if (p > kStripMaxP || c.inv_mode != 0) return false;
  // ---- riding rows of the multiplier r^T: w1 = j1- + r j0+, w2 = j0+ + r j1-  (doubling.jl:51-60); every wave writes the same values
  if (lane < N) {
    r[lane + N * LD] = c.jm[lane] * expk + P[lane + N * LD];
    r[lane + (N + 1) * LD] = c.jp[lane] + expk * P[lane + (N + 1) * LD];
  }
  real t0[kHalfSlots], y[kHalfSlots];
  half_load_lds<KS, H>(t, lr, lq, c0, t0);
#pragma unroll
  for (int j = 0; j < kHalfSlots; ++j) y[j] = t0[j];
  // Y = A^T = (t (I - r r)^-1)^T by Horner: Y <- t^T + (r r)^T Y
#pragma nounroll
  for (int k = 1; k < p; ++k) {
    r4 acc[NT];
    strip_zero(acc);
    half_mul<KS, H>(P, lr, lq, y, acc);
    real yn[kHalfSlots];
    half_reduce<KS, H>(exch, wave, lane, acc, yn);
#pragma unroll
    for (int j = 0; j < kHalfSlots; ++j) y[j] = t0[j] + yn[j];
  }
  // (A r)^T ; riding rows: (A w1)^T, (A w2)^T
  real zt[kHalfSlots];
  {
    r4 acc[NT];
    strip_zero(acc);
    half_mul<KS, H>(r, lr, lq, y, acc);
    half_reduce<KS, H>(exch, wave, lane, acc, zt);
  }
  const real aw = zt[RS];   // half 1: (A w1)[col] in the lanes lq == 0, (A w2)[col] in lq == 1
  // r^T + t^T (A r)^T (:64) and t^T A^T (:67)
  real rn[kHalfSlots], tn[kHalfSlots];
  {
    r4 acc[NT];
    strip_zero(acc);
    half_mul<KS, H>(t, lr, lq, zt, acc);   // (only the k-step slots of zt are operands)
    half_reduce<KS, H>(exch, wave, lane, acc, rn);
#pragma unroll
    for (int j = 0; j < kHalfSlots; ++j) rn[j] = rh[j] + rn[j];
  }
  {
    r4 acc[NT];
    strip_zero(acc);
    half_mul<KS, H>(t, lr, lq, y, acc);
    half_reduce<KS, H>(exch, wave, lane, acc, tn);
  }
  __syncthreads();   // every wave is done reading r, t, P of this step
  half_store_lds<KS, H>(r, lr, lq, c0, colok, rn);
  half_store_lds<KS, H>(t, lr, lq, c0, colok, tn);
  if (H && colok && lq == G::LQ0) {  // j0- += A w1 (:57)
    const real jm = c.jm[col] + aw;
    c.jm[col] = jm;
    r[col + (N + 1) * LD] = jm;
  }
  if (H && colok && lq == G::LQ1) {  // j0+ = j1+ + A w2 (:60)
    const real jp = c.jp[col] * expk + aw;
    c.jp[col] = jp;
    r[col + N * LD] = jp;
  }
  __syncthreads();
  return true;
}

// nd doubling steps on half-strips; same contract as doubling_run_lean
template <int KS>
__device__ __forceinline__ real doubling_run_half(Ctx &c, real *exch, int nd, real expk, bool &bail) {
  constexpr int N = 4 * KS, LD = StripGeom<KS>::LD;
  bail = false;
  if (nd == 0) return expk;
  const int h = wg_wave() & 1;
  real *r = c.r, *P = c.P;
  for (int i = wg_tid(); i < N; i += kThreads) {
    r[i + N * LD] = c.jp[i];
    r[i + (N + 1) * LD] = c.jm[i];
  }
  __syncthreads();
  for (int it = 0; it < nd; ++it) {
    // (the two halves run the same barrier sequence; the series length is workgroup-uniform)
    const bool ok = h ? doubling_step_half<KS, 1>(c, exch, expk) : doubling_step_half<KS, 0>(c, exch, expk);
    if (!ok) {
      bail = true;
      return expk;
    }
    expk = expk * expk;
  }
  // apply_D! (doubling.jl:93-110) and apply_D_SFI! (:112-118): r-+ rows and j0- scaled by sg
  for (int e = wg_tid(); e < N * N; e += kThreads) {
    int i, j;
    c.fd.split(e, i, j);
    r[i + j * LD] *= c.sg[i];
  }
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.jm[i] *= c.sg[i];
    r[i + N * LD] = 0.0; r[i + (N + 1) * LD] = 0.0; P[i + N * LD] = 0.0; P[i + (N + 1) * LD] = 0.0;
  }
  __syncthreads();
  return expk;
}

// ScatteringInterface_11 on three buffers (see interaction_strip for the algebra).  Returns false, nothing stored, if the series
// is too long.  Ends with a barrier.
template <int KS>
__device__ __forceinline__ bool interaction_strip_lean(Ctx &c, const CompPtrs &g) {
  using G = StripGeom<KS>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD, NN = N * N;
  static_assert(NT == 3 && kWaves > NT, "lean images: 3 x 3 tiles, at least one wave without a strip");
  const int lane = wg_lane(), wave = wg_wave(), lr = lane & 15, lq = lane >> 4;
  const int c0 = 16 * wave, col = c0 + lr;
  const bool strip = wave < NT, colok = col < N;
  real *r = c.r, *t = c.t, *P = c.P;
  // P = R+- ; riding row: column N of r = j0-
  {
    constexpr int U = (NN + kThreads - 1) / kThreads;
    real vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = wg_tid() + u * kThreads;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        vr[u] = MOM_NT_LOAD(g.R_pm + i + j * G::CP);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = wg_tid() + u * kThreads;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        P[i + j * LD] = vr[u];
      }
    }
  }
  for (int i = wg_tid(); i < N; i += kThreads) r[i + N * LD] = c.jm[i];
  __syncthreads();
  // B = r-+ R+- and W0 = R+-^T t++^T on strips, both from P = R+-
  r4 Bs[NT], W0[NT];
  real ss = 0.0;
  if (strip) {
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
    r4 tT[NT];
    strip_load_lds<KS>(t, lr, lq, c0, tT);
    strip_zero(W0);
    strip_mul<KS>(P, lr, lq, tT, W0);
  }
  __syncthreads();
  if (strip) strip_store_lds<KS>(P, lr, lq, c0, colok, Bs);
  wg_sumsq_put(c, ss);
  __syncthreads();
  const real beta2 = wg_sumsq_get(c);
  const int p = __builtin_amdgcn_readfirstlane(neumann_terms_12(beta2));
  if (p > kStripMaxP) return false;
  const unsigned mask = strip_sign_mask(c.sg, lq, N);
  // the two Horner loops (multiplier B^T in P) on the strip waves; the idle wave fetches T++ (+ J0+) into registers meanwhile
  r4 Y1[NT], Y2[NT];
  constexpr int IL = 64 * (kWaves - NT), UT = (NN + IL - 1) / IL;  // lanes without a strip (4-wave build: one wave; lean6: three)
  const int il = wg_tid() - 64 * NT;
  real vt[UT], vj = 0.0;
  if (strip) {
    strip_copy(Y2, W0);
#pragma nounroll
    for (int k = 1; k < p; ++k) {  // Y2 <- W0 + B^T Y2 : X^T
      r4 acc[NT];
      strip_copy(acc, W0);
      strip_mul<KS>(P, lr, lq, Y2, acc);
      strip_copy(Y2, acc);
    }
    r4 T1[NT];
    strip_load_glb<KS>(g.T_mm, lr, lq, c0, colok, T1);
    strip_copy(Y1, T1);
#pragma nounroll
    for (int k = 1; k < p; ++k) {  // Y1 <- T--^T + B^T Y1 : T01^T
      r4 acc[NT];
      strip_copy(acc, T1);
      strip_mul<KS>(P, lr, lq, Y1, acc);
      strip_copy(Y1, acc);
    }
  } else {
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int e = il + u * IL;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        vt[u] = MOM_NT_LOAD(g.T_pp + i + j * G::CP);
      }
    }
    if (il < N) vj = g.J0p[il];
  }
  __syncthreads();   // every strip wave is done with B
  if (!strip) {      // P = T++ ; riding row: column N = J0+
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int e = il + u * IL;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        P[i + j * LD] = vt[u];
      }
    }
    if (il < N) P[il + N * LD] = vj;
  }
  __syncthreads();
  if (strip) {
    // ---- chain 1: T-- = T01 t--, R-+ += (T01 r-+) T++, J0- += T01 (r-+ J0+ + j0-)                (:90-96)
    {
      r4 Radd[NT];
      strip_load_glb<KS>(g.R_mp, lr, lq, c0, colok, Radd);
      const real j0m = (colok && lq == G::LQ0) ? g.J0m[col] : 0.0;
      {
        r4 Yf[NT], o[NT];
        strip_copy(Yf, Y1);
        strip_flip(Yf, mask);
        strip_zero(o);
        strip_mul<KS>(t, lr, lq, Yf, o);
        strip_flip(o, mask);
        strip_store_glb<KS>(g.T_mm, lr, lq, c0, colok, o);
      }
      r4 V[NT];
      strip_zero(V);
      strip_mul<KS>(r, lr, lq, Y1, V);
      strip_mul<KS>(P, lr, lq, V, Radd);
      strip_store_glb<KS>(g.R_mp, lr, lq, c0, colok, Radd);
      if (colok && lq == G::LQ0) g.J0m[col] = j0m + (Radd[G::RT][G::RR0] + V[G::RT][G::RR0]);
    }
    // ---- chain 2: T21 = t++ + X r-+, T++ = T21 T++, J0+ = j0+ + T21 (J0+ + R+- j0-), R+- = r+- + X t--   (:110-116)
    {
      r4 T21[NT];
      strip_load_lds<KS>(t, lr, lq, c0, T21);
      strip_mul<KS>(r, lr, lq, Y2, T21);
      r4 o[NT];
      strip_zero(o);
      strip_mul<KS>(P, lr, lq, T21, o);
      strip_store_glb<KS>(g.T_pp, lr, lq, c0, colok, o);
      if (colok && lq == G::LQ0) g.J0p[col] = c.jp[col] + (o[G::RT][G::RR0] + T21[G::RT][G::RR0]);
      r4 acc[NT];
      strip_load_lds<KS>(r, lr, lq, c0, acc);
      const real sc = colok ? c.sg[col] : 1.0;
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) acc[rt] = acc[rt] * sc;
      strip_flip(Y2, mask);
      strip_mul<KS>(t, lr, lq, Y2, acc);
      strip_flip(acc, mask);
      strip_store_glb<KS>(g.R_pm, lr, lq, c0, colok, acc);
    }
  }
  __syncthreads();
  return true;
}

#ifndef MOM_LEAN_WAVES
#define MOM_LEAN_WAVES 3
#endif
// One launch walks all layers of every unit (sweep mode only); see the header of this file for what it does not do.
template <int KS>
__global__ void __launch_bounds__(kThreads, MOM_LEAN_WAVES) k_layer_lean(const LayerArgs a) {
  // (the argument block is never written: a store to it -- k_layer's `a.q.N = 4 KS` -- makes the compiler keep a private copy of
  // the whole 3.3 KB struct per LANE in scratch, 215 KB per wavefront, which the third workgroup per CU has to find room for)
  constexpr int N = 4 * KS;
  const size_t total = (size_t)a.S * a.M;
  Ctx c;
  make_ctx_lean(c, N, a.q.inv_mode, mom_smem);
  // the elemental layer's tables: E, F1, F2 and behind them the sun-block columns, all in P
  {
    const int ns = a.q.regular ? a.q.nS : 1, Nq = N / ns;
    c.tabE = c.P;
    c.tabZS = c.P + 3 * Nq * Nq;
  }
  zero_padding<true>(c);
  __syncthreads();
  load_streams(c, a.q);
  __syncthreads();
  real *exch = mom_smem + 3 * mat_elems(N) + lean_vec_reals(N);   // lean6: the half-strip exchange slots
  real *lay = exch + lean_exch_reals();
  const int nz = a.Nz_sweep;
  const int LW = 3 + a.K, LZ = kLeanLay / LW;
  for (size_t pt = blockIdx.x; pt < total; pt += gridDim.x) {
    const int n = (int)(pt % a.S), mrel = (int)(pt / a.S), m = a.m_first + mrel;
    const size_t NNs = (size_t)N * N;
    CompPtrs g = comp_ptrs(a.comp, N, comp_pitch(N), pt);
    int done = nz;
    for (int z = 0; z < nz; ++z) {
      const int zl = z % LZ;
      if (zl == 0) {
        const int cnt = ((nz - z < LZ) ? nz - z : LZ) * LW;
        for (int i = wg_tid(); i < cnt; i += kThreads) {
          const int zz = i / LW, k = i - zz * LW;
          const size_t o = (size_t)n + (size_t)a.S * (z + zz);
          lay[i] = (k == 0) ? as_global(a.tau)[o] : (k == 1) ? as_global(a.varpi)[o] : (k == 2) ? as_global(a.tau_sum)[o]
                                                                                                : as_global(a.zw)[(size_t)a.K * o + (k - 3)];
        }
        __syncthreads();
      }
      const int nd = a.nd_z[z];
      const bool first = (z == 0) && (a.first != 0);
      const real *ls = lay + zl * LW;
      const real tau = ls[0], varpi = ls[1], tau_sum = ls[2];
      const real dtau = ldexp(tau, -nd);         // τ ./ 2^ndoubl   (rt_kernel.jl:244)
      real expk = exp(-dtau / a.q.mu0);          // init_layer      (rt_kernel.jl:273)
      ZMix zpp{as_global(a.Zpp) + NNs * a.K * mrel, ls + 3, a.K, N};
      ZMix zmp{as_global(a.Zmp) + NNs * a.K * mrel, ls + 3, a.K, N};
      elemental_build(c, a.q, m, nd, tau_sum, dtau, varpi, zpp, zmp);
      bool bail;
      if constexpr (kLean6) expk = doubling_run_half<KS>(c, exch, nd, expk, bail);
      else expk = doubling_run_lean<KS>(c, nd, expk, bail);
      if (!bail) {
        if (first) {
          store_added_as_composite(c, g);
          __syncthreads();
        } else {
          bail = !interaction_strip_lean<KS>(c, g);
        }
      }
      if (bail) {  // workgroup-uniform: the full image redoes this layer and finishes the unit
        done = z;
        __syncthreads();
        break;
      }
    }
    if (wg_tid() == 0) a.resume[pt] = done;
  }
  if (wg_tid() == 0 && *c.bad) atomicMax(a.info, *c.bad);
}

}  // namespace MOM_NS
