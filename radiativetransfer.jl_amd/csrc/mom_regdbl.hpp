// mom_regdbl.hpp -- the doubling loop of the operator edges 64 < N <= 96 with REGISTER-RESIDENT operators
// (doubling_helper!, CoreKernel/doubling.jl:13-79; generic 8-wave Float64 build only).
//
// Above N = 64 four operators no longer fit the 160 KB of LDS, and the general path keeps them in a per-workgroup global
// slab: every product streams its operands through LDS panels (wg_gemm_big) on a 2 x 4 wave grid that leaves 25-48 % of
// its MFMA slots empty on the 5 x 5 / 6 x 6 tile grids of these sizes, with a barrier every 8 k.  But the register file
// of a CU (512 KB) holds what LDS cannot: here r, t and the two work matrices P, Q live in the accumulator (C/D) layout
// of the FP64 MFMA, their 16 x 16 tiles dealt round-robin over the 8 waves (tile q = i + NT j -> wave q mod 8: 25 or 36
// tiles, 6.25 / 9 per SIMD -- no empty slots), i.e. 4 x 5 tiles x 4 doubles = 160 VGPRs per lane at N = 96.  LDS is the
// exchange: a product C = A B first STAGES both operands (from the registers of the waves that own their tiles) as two
// whole column-major matrices (2 x 96 x 98 x 8 B = 150.5 KB at N = 96), then every wave multiplies its own tiles over the
// full K without a barrier, one A and one B fragment read per MFMA (25 % of the LDS read rate).  The source vectors
// (doubling.jl:51-60) are mat-vecs against the staged r and Q.  No global traffic inside the loop; r and t come from the
// slab once (elemental_build wrote them there) and go back once.
//
// One doubling step = 5 + (p - 2) products (the truncated Neumann series of times_inv, same term count p as the general
// path: Horner for p <= 4, repeated squaring up to 512 terms); anything else (forced pivoting, p beyond the series)
// hands the remaining steps back to the general path.
#pragma once

namespace MOM_NS {

constexpr int kRgMaxN = 96;
__host__ __device__ inline bool rg_applies(int N) { return kF64 && kWaves == 8 && N > 64 && N <= kRgMaxN; }
// LDS image of the register-resident doubling: the vector area up to and including 16 reals of `part`, then two slots
__host__ __device__ inline size_t rg_slot_doubles(int N) { return (size_t)np_for(N) * ld_for(N); }

// NTS tiles of a matrix in the registers of one wave: tile s is tile q = wave + 8 s of the NT x NT grid (q = i + NT j)
template <int NTS>
struct RgMat {
  r4 t[NTS];
};

template <int NTS>
struct RgGeom {
  const real *pa[NTS];  // A-fragment address of tile s in slot 0: (16 i + lr) + lq ld
  const real *pb[NTS];  // B-fragment address of tile s in slot 0: lq + (16 j + lr) ld
  int oc[NTS];          // C-layout offset of register 0 of tile s: (16 i + lq) + (16 j + lr) ld   (+ 4 r per register)
  int ti[NTS], tj[NTS];
};

template <int NT, int NTS>
__device__ __forceinline__ void rg_geom(RgGeom<NTS> &g, const real *S0, int ld) {
  const int lane = wg_lane(), lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(wg_wave());
#pragma unroll
  for (int s = 0; s < NTS; ++s) {
    const int q = wave + 8 * s;
    const int i = q % NT, j = q / NT;
    g.ti[s] = i; g.tj[s] = j;
    g.pa[s] = S0 + (16 * i + lr + lq * ld);
    g.pb[s] = S0 + (lq + (16 * j + lr) * ld);
    g.oc[s] = 16 * i + lq + (16 * j + lr) * ld;
  }
}

// registers -> slot (whole tiles, zero padding included).  Needs a barrier before (readers of the slot) and after.
template <int NTS>
__device__ __forceinline__ void rg_stage(real *L, const RgGeom<NTS> &g, const RgMat<NTS> &X) {
#pragma unroll
  for (int s = 0; s < NTS; ++s) {
    real *p = L + g.oc[s];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * r] = X.t[s][r];
  }
}

// slot -> registers; entries with row >= N or column >= N are taken as zero (the slab's padding is not)
template <int NTS>
__device__ __forceinline__ void rg_fetch(RgMat<NTS> &X, const real *L, const RgGeom<NTS> &g, int N) {
  const int lane = wg_lane(), lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int s = 0; s < NTS; ++s) {
    const real *p = L + g.oc[s];
    const bool colok = 16 * g.tj[s] + lr < N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real v = p[4 * r];
      X.t[s][r] = (colok && 16 * g.ti[s] + lq + 4 * r < N) ? v : 0.0;
    }
  }
}

// C (+)= A B with A in slot SA (0 / 1), B in slot SB, KS k-steps of 4; slot 1 sits `so` reals behind slot 0.  One A and one
// B fragment read per MFMA; the two waves of a SIMD alternate between their reads and their MFMAs by themselves.
template <int NTS, bool ACC>
__device__ __forceinline__ void rg_mm(RgMat<NTS> &C, int SA, int SB, const RgGeom<NTS> &g, int so, int ld, int KS) {
  if (!ACC) {
#pragma unroll
    for (int s = 0; s < NTS; ++s) C.t[s] = (r4){0.0, 0.0, 0.0, 0.0};
  }
  const int offa = SA ? so : 0, offb = SB ? so : 0, ld4 = 4 * ld;
#pragma unroll 2
  for (int kk = 0; kk < KS; ++kk) {
    real a[NTS], b[NTS];
#pragma unroll
    for (int s = 0; s < NTS; ++s) {
      a[s] = g.pa[s][offa + kk * ld4];
      b[s] = g.pb[s][offb + 4 * kk];
    }
#pragma unroll
    for (int s = 0; s < NTS; ++s) C.t[s] = mma16(a[s], b[s], C.t[s]);
  }
}

template <int NTS>
__device__ __forceinline__ real rg_sumsq(const RgMat<NTS> &X) {
  real ss = 0.0;
#pragma unroll
  for (int s = 0; s < NTS; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += X.t[s][r] * X.t[s][r];
  return ss;
}

// Q = T (I - B)^-1 by the truncated Neumann series of p terms (times_inv of mom_kernels.hpp: Horner for p <= 4, repeated
// squaring up to 512 terms); B in P (destroyed), T in registers.  Both slots must be free on entry (a barrier since their
// last readers); ends with a product reading both.
template <int NTS>
__device__ __forceinline__ void rg_series(RgMat<NTS> &Q, RgMat<NTS> &P, const RgMat<NTS> &T, int p, const RgGeom<NTS> &g, real *S0,
                                          real *S1, int so, int ld, int KS, int N) {
  if (p <= 4) {
    Q = T;
    if (p >= 2) {
      rg_stage(S1, g, P);
      rg_stage(S0, g, T);
      __syncthreads();
      rg_mm<NTS, true>(Q, 0, 1, g, so, ld, KS);    // T + T B
      for (int k = 3; k <= p; ++k) {
        __syncthreads();
        rg_stage(S0, g, Q);
        __syncthreads();
        Q = T;
        rg_mm<NTS, true>(Q, 0, 1, g, so, ld, KS);  // T + Q B
      }
    }
  } else {
    // G = (I + B)(I + B^2)(I + B^4) ... in Q; then Q = T G
    const int lane = wg_lane(), lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int s = 0; s < NTS; ++s) {
      Q.t[s] = P.t[s];
      if (g.ti[s] == g.tj[s]) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (lq + 4 * rr == lr && 16 * g.ti[s] + lr < N) Q.t[s][rr] += 1.0;
      }
    }
    for (int terms = 2; terms < p; terms *= 2) {
      rg_stage(S0, g, P);
      __syncthreads();
      rg_mm<NTS, false>(P, 0, 0, g, so, ld, KS);   // B <- B B
      __syncthreads();
      rg_stage(S0, g, Q);
      rg_stage(S1, g, P);
      __syncthreads();
      rg_mm<NTS, true>(Q, 0, 1, g, so, ld, KS);    // G <- G + G B
      __syncthreads();
    }
    rg_stage(S0, g, T);
    rg_stage(S1, g, Q);
    __syncthreads();
    rg_mm<NTS, false>(Q, 0, 1, g, so, ld, KS);     // Q = T G
  }
}

// series length for ||B||_F^2 = beta2, as times_inv chooses it in generic mode; > 512: pivoted inverse (not done here)
__device__ __forceinline__ int rg_terms(const Ctx &c, real beta2) {
  int p = neumann_terms(c.thr, beta2);
  if (p > 32 && beta2 < 0.81) {
    const real beta = sqrt(beta2);
    p = (int)ceil((38.816242111356935 - log(1.0 - beta)) / -log(beta));
    if (p < 33) p = 33;
  }
  return p;
}

typedef real rg_r2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rg_r2 rg_ld2(const real *p) { return *(const rg_r2 *)p; }
__device__ __forceinline__ rg_r2 rg_ld2(const gdouble *p) { return *(const __attribute__((address_space(1))) rg_r2 *)p; }
__device__ __forceinline__ void rg_st2(real *p, rg_r2 v) { *(rg_r2 *)p = v; }
__device__ __forceinline__ void rg_st2(gdouble *p, rg_r2 v) { *(__attribute__((address_space(1))) rg_r2 *)p = v; }

// N x N block of a column-major array (pitch ps: a composite block or a slab buffer) -> slot, zero padding up to Np x Np
// written along; SIG: diag(sg) block diag(sg) (the added layer's r+- / t--).  16-byte accesses (ps, ld even).
template <int NT, bool SIG, class PS>
__device__ __forceinline__ void rg_load(real *L, PS src, int ps, const real *sg, int N, int ld) {
  typedef real r2 __attribute__((ext_vector_type(2)));
  constexpr int Np = 16 * NT, H = Np / 2;
  for (int e = wg_tid(); e < H * Np; e += kThreads) {
    const int j = e / H, i = 2 * (e - j * H);
    r2 v = {0.0, 0.0};
    if (j < N) {
      if (i + 1 < N) v = rg_ld2(src + i + (size_t)j * ps);
      else if (i < N) v.x = src[i + (size_t)j * ps];
      if (SIG) { const real sj = sg[j]; v.x *= sg[i] * sj; v.y *= sg[i + 1] * sj; }
    }
    *(r2 *)(L + i + j * ld) = v;
  }
}

// slot -> N x N block of a column-major global array (pitch ps)
template <int NT, class PD>
__device__ __forceinline__ void rg_unload(PD dst, int ps, const real *L, int N, int ld) {
  typedef real r2 __attribute__((ext_vector_type(2)));
  constexpr int Np = 16 * NT, H = Np / 2;
  for (int e = wg_tid(); e < H * N; e += kThreads) {
    const int j = e / H, i = 2 * (e - j * H);
    const r2 v = *(const r2 *)(L + i + j * ld);
    if (i + 1 < N) rg_st2(dst + i + (size_t)j * ps, v);
    else if (i < N) dst[i + (size_t)j * ps] = v.x;
  }
}

// N x N block of a column-major array -> registers (accumulator layout), zero outside; SIG as above
template <int NTS, bool SIG, class PS>
__device__ __forceinline__ void rg_fetch_global(RgMat<NTS> &X, PS src, int ps, const real *sg, const RgGeom<NTS> &g, int N) {
  const int lane = wg_lane(), lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int s = 0; s < NTS; ++s) {
    const int col = 16 * g.tj[s] + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * g.ti[s] + lq + 4 * r;
      real v = 0.0;
      if (row < N && col < N) {
        v = src[row + (size_t)col * ps];
        if (SIG) v *= sg[row] * sg[col];
      }
      X.t[s][r] = v;
    }
  }
}

// y1[i] = sa1 a1[i] + sx1 sum_k M1[i,k] x1[k] (threads 0..127) and y2[i] = sa2 a2[i] + sx2 sum_k M2[i,k] x2[k] (threads
// 128..255); M1, M2 column-major in slots (y2 == nullptr: only the first).  No barrier inside.
__device__ __forceinline__ void rg_matvec2(const real *M1, const real *M2, int ld, int N, const real *x1, real sx1, const real *x2,
                                           real sx2, const real *a1, real sa1, const real *a2, real sa2, real *y1, real *y2) {
  const int tid = wg_tid();
  if (tid >= 256) return;
  const int i = tid & 127;
  if (i >= N) return;
  const bool second = tid >= 128;
  if (second && y2 == nullptr) return;
  const real *x = second ? x2 : x1;
  const real *M = second ? M2 : M1;
  const real sx = second ? sx2 : sx1;
  real s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 0;
  for (; k + 4 <= N; k += 4) {
    s0 += M[i + (k + 0) * ld] * x[k + 0];
    s1 += M[i + (k + 1) * ld] * x[k + 1];
    s2 += M[i + (k + 2) * ld] * x[k + 2];
    s3 += M[i + (k + 3) * ld] * x[k + 3];
  }
  for (; k < N; ++k) s0 += M[i + k * ld] * x[k];
  const real sum = ((s0 + s1) + (s2 + s3)) * sx;
  if (second) y2[i] = a2[i] * sa2 + sum;
  else y1[i] = a1[i] * sa1 + sum;
}

// flat copy of the first N columns between a slab buffer and a slot (same pitch ld, ld even)
__device__ __forceinline__ void rg_copy(real *dst, const real *src, int cnt) {
  typedef real r2 __attribute__((ext_vector_type(2)));
  const int n2 = cnt >> 1;
  for (int e0 = wg_tid(); e0 < n2; e0 += 4 * kThreads) {
    r2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * kThreads < n2) v[u] = *(const r2 *)(src + 2 * (e0 + u * kThreads));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * kThreads < n2) *(r2 *)(dst + 2 * (e0 + u * kThreads)) = v[u];
  }
}

// nd doubling steps on (c.r, c.t, c.jp, c.jm) of the slab / vector area; returns the number of steps done (the caller's
// general loop takes the rest) and leaves r, t in the slab again.  expk is updated.  Ends with a barrier.  NTS = the tile
// count of the calling wave: the two bodies of a workgroup (NTS = ceil(T / 8) and one less) execute the same barriers.
template <int NT, int NTS>
__device__ __forceinline__ int rg_body(Ctx &c, int nd, real *expk_io) {
  const int N = __builtin_amdgcn_readfirstlane(c.N), ld = ld_for(N), KS = (N + 3) >> 2;  // uniform: scalar loop control
  const int so = (int)rg_slot_doubles(N);
  real *S0 = mom_smem + part_offset_doubles(N) + 16, *S1 = S0 + so;  // spelled from mom_smem: LDS addressing
  RgGeom<NTS> g;
  rg_geom<NT, NTS>(g, S0, ld);
  RgMat<NTS> r, t, P, Q;
  real expk = *expk_io;
  // slab -> slots -> registers
  __syncthreads();
  rg_load<NT, false>(S0, (const real *)c.r, ld, nullptr, N, ld);
  rg_load<NT, false>(S1, (const real *)c.t, ld, nullptr, N, ld);
  __syncthreads();
  rg_fetch(r, S0, g, N);
  rg_fetch(t, S1, g, N);
  int it = 0;
  for (; it < nd; ++it) {
    __syncthreads();                  // everyone is done with both slots
    rg_stage(S0, g, r);
    __syncthreads();
    // w1 = j1- + r j0+ ; w2 = j0+ + r j1-   with j1± = j0± expk   (doubling.jl:51-60)
    rg_matvec2(S0, S0, ld, N, c.jp, 1.0, c.jm, expk, c.jm, expk, c.jp, 1.0, c.v1, c.v2);
    rg_mm<NTS, false>(P, 0, 0, g, so, ld, KS);       // P = r r   (:44)
    wg_sumsq_put(c, rg_sumsq(P));
    __syncthreads();
    const real beta2 = wg_sumsq_get(c);
    const int p = rg_terms(c, beta2);
    if (p > 512) break;               // pivoted inverse: the general path (r, t, j are still those of step `it`)
    rg_series<NTS>(Q, P, t, p, g, S0, S1, so, ld, KS, N);   // Q = t (I - P)^-1   (:47-48)
    // P = Q r   (the reference's (tt gp) r, :64) ; j0- += Q w1 (:57) ; j0+ = j1+ + Q w2 (:60)
    __syncthreads();
    rg_stage(S0, g, Q);
    rg_stage(S1, g, r);
    __syncthreads();
    rg_matvec2(S0, S0, ld, N, c.v1, 1.0, c.v2, 1.0, c.jm, 1.0, c.jp, expk, c.jm, c.jp);
    rg_mm<NTS, false>(P, 0, 1, g, so, ld, KS);
    // t <- Q t (:67) ; r <- r + P t (:64)
    __syncthreads();
    rg_stage(S1, g, t);
    __syncthreads();
    rg_mm<NTS, false>(t, 0, 1, g, so, ld, KS);       // the staged copy keeps the old t for the next product
    __syncthreads();
    rg_stage(S0, g, P);
    __syncthreads();
    rg_mm<NTS, true>(r, 0, 1, g, so, ld, KS);
    expk = expk * expk;               // :61
  }
  // registers -> slots -> slab
  __syncthreads();
  rg_stage(S0, g, r);
  rg_stage(S1, g, t);
  __syncthreads();
  rg_copy(c.r, S0, N * ld);
  rg_copy(c.t, S1, N * ld);
  __syncthreads();
  *expk_io = expk;
  return it;
}

template <int NT>
__device__ __attribute__((noinline)) int rg_doubling(Ctx &c, int nd, real *expk_io) {
  constexpr int T = NT * NT, TS = (T + 7) / 8, full = (T % 8 == 0) ? 8 : T % 8;  // waves < full own TS tiles, the others TS - 1
  const int wave = __builtin_amdgcn_readfirstlane(wg_wave());
  if (wave < full) return rg_body<NT, TS>(c, nd, expk_io);
  return rg_body<NT, TS - 1>(c, nd, expk_io);
}

// ScatteringInterface_11 (interaction_helper!, CoreKernel/interaction.jl:69-117) of the composite block g with the added
// layer in the slab (c.r = r-+, c.t = t++, r+- = S r-+ S, t-- = S t++ S with S = diag(sg); c.jp, c.jm) -- the same
// register-resident scheme: every operand is staged into a slot (composite blocks straight from global memory, zero
// padding written along), results go back through a slot as 16-byte rows.  Returns false BEFORE anything is stored if one
// of the two inverses needs more than the series (the general path then does the whole interaction).  All threads; ends
// with a barrier.
template <int NT, int NTS>
__device__ __forceinline__ bool rg_interaction_body(Ctx &c, const CompPtrs &cp) {
  const int N = __builtin_amdgcn_readfirstlane(c.N), ld = ld_for(N), KS = (N + 3) >> 2, cl = cp.ld;
  const int so = (int)rg_slot_doubles(N);
  real *S0 = mom_smem + part_offset_doubles(N) + 16, *S1 = S0 + so;
  const real *sr = c.r, *st = c.t;
  const gdouble *gRmp = cp.R_mp, *gRpm = cp.R_pm, *gTpp = cp.T_pp, *gTmm = cp.T_mm;
  RgGeom<NTS> g;
  rg_geom<NT, NTS>(g, S0, ld);
  RgMat<NTS> A, B, C, D;
  // composite sources -> LDS
  for (int i = wg_tid(); i < N; i += kThreads) {
    c.Jp[i] = cp.J0p[i];
    c.Jm[i] = cp.J0m[i];
  }
  __syncthreads();
  // B1 = r-+ R+- (:81) and B2 = R+- r-+ (:104) from one staging; v1 = j0- + r-+ J0+ ; w = J0+ + R+- j0-
  rg_load<NT, false>(S0, sr, ld, nullptr, N, ld);
  rg_load<NT, false>(S1, gRpm, cl, nullptr, N, ld);
  __syncthreads();
  rg_matvec2(S0, S1, ld, N, c.Jp, 1.0, c.jm, 1.0, c.jm, 1.0, c.Jp, 1.0, c.v1, c.j1p);
  rg_mm<NTS, false>(A, 0, 1, g, so, ld, KS);       // A = B1
  rg_mm<NTS, false>(D, 1, 0, g, so, ld, KS);       // D = B2 (kept until the second inverse)
  wg_sumsq_put(c, rg_sumsq(A));
  __syncthreads();
  const real b1 = wg_sumsq_get(c);
  __syncthreads();
  wg_sumsq_put(c, rg_sumsq(D));
  __syncthreads();
  const real b2 = wg_sumsq_get(c);
  const int p1 = rg_terms(c, b1), p2 = rg_terms(c, b2);
  if (p1 > 512 || p2 > 512) return false;
  // T01 = T-- (I - B1)^-1   (:83-87)
  rg_load<NT, false>(S0, gTmm, cl, nullptr, N, ld);
  __syncthreads();
  rg_fetch(B, S0, g, N);                           // B = T--
  __syncthreads();
  rg_series<NTS>(C, A, B, p1, g, S0, S1, so, ld, KS, N);   // C = T01
  // J0- += T01 v1 (:90) ; T-- = T01 t-- (:96)
  __syncthreads();
  rg_stage(S0, g, C);
  rg_load<NT, true>(S1, st, ld, c.sg, N, ld);
  __syncthreads();
  rg_matvec2(S0, S0, ld, N, c.v1, 1.0, nullptr, 1.0, c.Jm, 1.0, nullptr, 1.0, c.Jm, nullptr);
  rg_mm<NTS, false>(A, 0, 1, g, so, ld, KS);       // A = new T--
  __syncthreads();
  rg_stage(S1, g, A);
  __syncthreads();
  rg_unload<NT>(cp.T_mm, cl, S1, N, ld);
  // X = T01 r-+ ; R-+ += X T++   (:93)
  __syncthreads();
  rg_load<NT, false>(S1, sr, ld, nullptr, N, ld);
  __syncthreads();
  rg_mm<NTS, false>(A, 0, 1, g, so, ld, KS);       // A = T01 r-+
  __syncthreads();
  rg_stage(S0, g, A);
  rg_load<NT, false>(S1, gTpp, cl, nullptr, N, ld);
  rg_fetch_global<NTS, false>(B, gRmp, cl, nullptr, g, N);
  __syncthreads();
  rg_mm<NTS, true>(B, 0, 1, g, so, ld, KS);        // B = new R-+
  __syncthreads();
  rg_stage(S0, g, B);
  __syncthreads();
  rg_unload<NT>(cp.R_mp, cl, S0, N, ld);
  // T21 = t++ (I - B2)^-1   (:105-107)
  __syncthreads();
  rg_load<NT, false>(S0, st, ld, nullptr, N, ld);
  __syncthreads();
  rg_fetch(B, S0, g, N);                           // B = t++
  __syncthreads();
  rg_series<NTS>(C, D, B, p2, g, S0, S1, so, ld, KS, N);   // C = T21
  // J0+ = j0+ + T21 w (:110) ; Y = T21 R+- ; T++ = T21 T++ (:113)
  __syncthreads();
  rg_stage(S0, g, C);
  rg_load<NT, false>(S1, gRpm, cl, nullptr, N, ld);
  __syncthreads();
  rg_matvec2(S0, S0, ld, N, c.j1p, 1.0, nullptr, 1.0, c.jp, 1.0, nullptr, 1.0, c.Jp, nullptr);
  rg_mm<NTS, false>(A, 0, 1, g, so, ld, KS);       // A = T21 R+-
  __syncthreads();
  rg_load<NT, false>(S1, gTpp, cl, nullptr, N, ld);
  __syncthreads();
  rg_mm<NTS, false>(D, 0, 1, g, so, ld, KS);       // D = new T++
  // R+- = r+- + Y t--   (:116)
  __syncthreads();
  rg_stage(S0, g, A);
  rg_load<NT, true>(S1, st, ld, c.sg, N, ld);
  rg_fetch_global<NTS, true>(B, sr, ld, c.sg, g, N);       // B = r+- = S r-+ S
  __syncthreads();
  rg_mm<NTS, true>(B, 0, 1, g, so, ld, KS);        // B = new R+-
  __syncthreads();
  rg_stage(S0, g, D);
  rg_stage(S1, g, B);
  __syncthreads();
  rg_unload<NT>(cp.T_pp, cl, S0, N, ld);
  rg_unload<NT>(cp.R_pm, cl, S1, N, ld);
  for (int i = wg_tid(); i < N; i += kThreads) {
    cp.J0p[i] = c.Jp[i];
    cp.J0m[i] = c.Jm[i];
  }
  __syncthreads();
  return true;
}

template <int NT>
__device__ __attribute__((noinline)) bool rg_interaction(Ctx &c, const CompPtrs &cp) {
  constexpr int T = NT * NT, TS = (T + 7) / 8, full = (T % 8 == 0) ? 8 : T % 8;
  const int wave = __builtin_amdgcn_readfirstlane(wg_wave());
  if (wave < full) return rg_interaction_body<NT, TS>(c, cp);
  return rg_interaction_body<NT, TS - 1>(c, cp);
}

}  // namespace MOM_NS
