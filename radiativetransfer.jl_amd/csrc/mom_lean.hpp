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
__host__ __device__ inline size_t lean_vec_reals(int N) { return part_offset_doubles(N) + 16; }
__host__ __device__ inline size_t lean_lds_bytes(int N) { return (3 * mat_elems(N) + lean_vec_reals(N) + kLeanLay) * sizeof(real); }
// does the image apply to operators of edge N with ns Stokes components per stream?  (the tables must fit P)
__host__ __device__ inline bool lean_applies(int N, int ns) {
  const int Nq = N / (ns > 0 ? ns : 1);
  return kF64 && kWaves == 4 && (N == 36 || N == 40) && 3 * Nq * Nq + 2 * ns * N <= (int)mat_elems(N) &&
         3 * lean_lds_bytes(N) + 3 * 1024 <= kLdsPerCU;
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

// ScatteringInterface_11 on three buffers (see interaction_strip for the algebra).  Returns false, nothing stored, if the series
// is too long.  Ends with a barrier.
template <int KS>
__device__ __forceinline__ bool interaction_strip_lean(Ctx &c, const CompPtrs &g) {
  using G = StripGeom<KS>;
  constexpr int N = G::N, NT = G::NT, LD = G::LD, NN = N * N;
  static_assert(kWaves == 4 && NT == 3, "lean image: 4-wave build, 3 x 3 tiles");
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
  constexpr int UT = (NN + 63) / 64;
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
      const int e = lane + u * 64;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        vt[u] = MOM_NT_LOAD(g.T_pp + i + j * G::CP);
      }
    }
    if (lane < N) vj = g.J0p[lane];
  }
  __syncthreads();   // every strip wave is done with B
  if (!strip) {      // P = T++ ; riding row: column N = J0+
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int e = lane + u * 64;
      if (e < NN) {
        int i, j;
        c.fd.split(e, i, j);
        P[i + j * LD] = vt[u];
      }
    }
    if (lane < N) P[lane + N * LD] = vj;
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
  real *lay = mom_smem + 3 * mat_elems(N) + lean_vec_reals(N);
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
      expk = doubling_run_lean<KS>(c, nd, expk, bail);
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
