// mom_rrs_wg.hpp -- the RRS pair kernels above N = 16 as ONE WORKGROUP PER PAIR (included by mom_rrs.hip inside its
// namespace, big-tile object only: 2 x 2 tiles for 16 < N <= 32, 3 x 3 for 32 < N <= 48, 4 x 4 for 48 < N <= 64).
//   doubling_helper!(::RRS)            CoreKernel/doubling_inelastic.jl:61-89 (sources), :98-125 (operators)
//   interaction_helper!(::RRS, 11)     CoreKernel/interaction_inelastic.jl:249-335
//
// Why.  The wave-per-pair bodies (dbl_pair_body / int_pair_body) keep every operator of a pair as NT x NT register tiles of one
// wave: at NT = 3 / 4 the dozen live operators of a pair are 1 700 ... 3 000 registers per lane against 512, and the kernels
// ran from scratch memory (r4: 0.02-0.04 of the HBM roofline; profiles/r05_rrs_wg_ab.txt); at NT = 2 they fit, at one wave
// per SIMD.  Here a pair belongs to a workgroup of NT
// waves and every operator is split into COLUMN STRIPS: wave w owns tile column w (NT tiles = 4 NT registers per lane).  With
// TN(U, V) = U^T V of mom_tile.hpp, column strip w of a product needs the whole left factor U and only strip w of V:
//   * V strips stay in the registers of their wave (they are loaded from global memory as strips, or are the wave's own
//     result of an earlier product);
//   * U is read fragment by fragment from an LDS copy of the matrix, laid out in 16 x 16 tiles of pitch 17 so that BOTH
//     orientations are (nearly) conflict-free -- the transposes the one-wave bodies make through LDS (a_c, b_c, bn_c) become a
//     different index expression of the same copy, and W = b^T + Y, V = bn^T + Y are formed fragment-wise from two copies;
//   * computed left factors (X, Y, bn) are published strip-wise into LDS between two barriers; left factors that come from
//     global memory (r[n0]^T, (G t)[n0] ...) are loaded as strips -- one per wave, coalesced -- and published the same way.
// Four LDS matrices (a, b / bn, X / Y, the rotating global one) = 35 KB (NT = 2: four workgroups per CU) / 78 KB (NT = 3: two) /
// 139 KB (NT = 4: one).
// The source vectors go through the vector ALU as in the one-wave bodies: a wave multiplies its strip with the full vector
// (row layout) and owns 16 entries of the result; full vectors are exchanged through four small LDS buffers.
// The order of the products and of every accumulation is that of dbl_pair_body / int_pair_body: the results are bitwise those
// of the one-wave kernels up to the scheduling of independent roundings (tests/test_gpu_rrs.py::
// test_rrs_workgroup_and_wave_kernels_agree compares both at 1e-13; MOM_RRS_WG=0 selects the old ones, MOM_RRS_WG2=0 for N <= 32 only).
#pragma once
#ifndef MOMR_WG3_WPE
#define MOMR_WG3_WPE 2   // waves per SIMD the 3-wave image is compiled for (2: two workgroups per CU)
#endif

template <int NT>
struct Strip {
  d4 t[NT];  // t[a]: tile (a, w) of the matrix, w = the wave's column
};

template <int NT>
constexpr int wg_mat_doubles() { return NT * NT * kTileDoubles; }
template <int NT>
constexpr size_t wg_lds_bytes(int nmat) { return ((size_t)nmat * wg_mat_doubles<NT>() + 8 * 16 * NT) * 8; }  // + 4 exchange buffers + 4 source vectors

template <int NT>
__device__ __forceinline__ Strip<NT> szeros() {
  Strip<NT> Z;
#pragma unroll
  for (int a = 0; a < NT; ++a) Z.t[a] = (d4){0.0, 0.0, 0.0, 0.0};
  return Z;
}
template <int NT>
__device__ __forceinline__ Strip<NT> sadd(const Strip<NT> &A, const Strip<NT> &B) {
  Strip<NT> C;
#pragma unroll
  for (int a = 0; a < NT; ++a) C.t[a] = A.t[a] + B.t[a];
  return C;
}
// column strip w of the _t form of a column-major block at pitch 16 NT (load_t restricted to b = w): 128-byte rows
template <int NT>
__device__ __forceinline__ Strip<NT> sload(const Geo &g, int w, const double *p) {
  Strip<NT> X;
  const double *q = p + 16 * w + g.lr + (16 * NT) * g.lq;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) X.t[a][r] = q[(16 * NT) * (16 * a + 4 * r)];
  return X;
}
template <int NT>
__device__ __forceinline__ void sstore(const Geo &g, int w, double *p, const Strip<NT> &X) {
  double *q = p + 16 * w + g.lr + (16 * NT) * g.lq;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[(16 * NT) * (16 * a + 4 * r)] = X.t[a][r];
}
// f(i = column index, j = row index, value) over the strip (map_t restricted to b = w)
template <int NT, class F>
__device__ __forceinline__ void smap(const Geo &g, int w, Strip<NT> &X, F f) {
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) X.t[a][r] = f(16 * w + g.lr, g.row(a, r), X.t[a][r]);
}
// strip -> LDS matrix (tile (a, w) at (a NT + w) kTileDoubles, element (row, col) of the tile at row kTileLd + col)
template <int NT>
__device__ __forceinline__ void spublish(const Geo &g, int w, double *M, const Strip<NT> &X) {
  double *q = M + w * kTileDoubles + g.lq * kTileLd + g.lr;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[a * NT * kTileDoubles + 4 * r * kTileLd] = X.t[a][r];
}
template <int NT>
__device__ __forceinline__ Strip<NT> sread(const Geo &g, int w, const double *M) {  // the wave's own strip back from LDS
  Strip<NT> X;
  const double *q = M + w * kTileDoubles + g.lq * kTileLd + g.lr;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) X.t[a][r] = q[a * NT * kTileDoubles + 4 * r * kTileLd];
  return X;
}
// register s of tile (tk, ti) of the LDS matrix M (TR = false) or of its transpose (TR = true) in the C / A-operand layout
template <int NT, bool TR>
__device__ __forceinline__ double ufrag(const Geo &g, const double *M, int tk, int ti, int s) {
  return TR ? M[(ti * NT + tk) * kTileDoubles + g.lr * kTileLd + g.lq + 4 * s]
            : M[(tk * NT + ti) * kTileDoubles + (g.lq + 4 * s) * kTileLd + g.lr];
}
// acc + (column strip of) U^T V;  U = M (TR = false) or M^T (TR = true) from LDS.  k-steps in the zero padding are skipped.
// NG ("no guard"): ALL k-steps are executed, the ones in the zero padding included (they add zeros), without a branch per step --
// the doubling pair kernel uses it when at most one k-step of a product lies in the padding (N = 32, 48, 60, 64 ...): the straight
// code gains 4-6 % there (profiles/r05_rrs_wg_ab.txt (13)); with more padding, or in the interaction kernel, the guards win.
template <int NT, bool TR, bool NG = false>
__device__ __forceinline__ Strip<NT> sTNacc(const Geo &g, const double *M, const Strip<NT> &V, Strip<NT> acc) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (NG || 16 * tk + 4 * s < g.N) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
          acc.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(ufrag<NT, TR>(g, M, tk, ti, s), V.t[tk][s], acc.t[ti], 0, 0, 0);
      }
  return acc;
}
// The same products with a JOB per k-step: job(tk, s) is called once per k-step, ahead of its MFMAs -- the callers use it to issue
// ONE load instruction of a strip they need later (StripReq), so that the 4 NT loads of a strip go out between the products'
// MFMAs instead of as a burst in front of them.  (A burst stalls the wave at issue: phase stamps with the next pair's 67 loads
// requested in front of the last two products put 4 400 cycles on that section, 66 per load -- the CU takes a strip set at L2 -> L1
// bandwidth, and a wave blocked in the memory queue issues no MFMAs.  The k-step guards are branches, so the compiler's scheduler
// cannot interleave across them by itself.)
template <int NT, bool TR, bool NG, class J>
__device__ __forceinline__ Strip<NT> sTNacc_job(const Geo &g, const double *M, const Strip<NT> &V, Strip<NT> acc, J job) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      job(tk, s);
      if (NG || 16 * tk + 4 * s < g.N) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
          acc.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(ufrag<NT, TR>(g, M, tk, ti, s), V.t[tk][s], acc.t[ti], 0, 0, 0);
      }
    }
  return acc;
}
template <int NT, bool NG, class J>
__device__ __forceinline__ Strip<NT> sTNacc_sum_job(const Geo &g, const double *M1, const double *M2, const Strip<NT> &V, Strip<NT> acc, J job) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      job(tk, s);
      if (NG || 16 * tk + 4 * s < g.N) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const double u = ufrag<NT, true>(g, M1, tk, ti, s) + ufrag<NT, false>(g, M2, tk, ti, s);
          acc.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, V.t[tk][s], acc.t[ti], 0, 0, 0);
        }
      }
    }
  return acc;
}
// a strip requested one element row per call: step(a, r) issues the load of X.t[a][r] (sload spread over 4 NT calls)
template <int NT>
struct StripReq {
  Strip<NT> X;
  const double *q;
  __device__ __forceinline__ void init(const Geo &g, int w, const double *p) { q = p + 16 * w + g.lr + (16 * NT) * g.lq; }
  __device__ __forceinline__ void step(int a, int r) { X.t[a][r] = q[(16 * NT) * (16 * a + 4 * r)]; }
};
// ... the same for a strip of the _c form (sloadT spread over 4 NT calls)
template <int NT>
struct StripReqT {
  Strip<NT> X;
  const double *q;
  __device__ __forceinline__ void init(const Geo &g, int w, const double *p) { q = p + g.lq + (size_t)(16 * NT) * (16 * w + g.lr); }
  __device__ __forceinline__ void step(int a, int r) { X.t[a][r] = q[16 * a + 4 * r]; }
};
// ... with U = M1^T + M2 formed fragment-wise (W = b^T + Y, V = bn^T + Y of the doubling step)
template <int NT, bool NG = false>
__device__ __forceinline__ Strip<NT> sTNacc_sum(const Geo &g, const double *M1, const double *M2, const Strip<NT> &V, Strip<NT> acc) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (NG || 16 * tk + 4 * s < g.N) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const double u = ufrag<NT, true>(g, M1, tk, ti, s) + ufrag<NT, false>(g, M2, tk, ti, s);
          acc.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, V.t[tk][s], acc.t[ti], 0, 0, 0);
        }
      }
  return acc;
}
// entries [16 w, 16 w + 16) of M x (column layout: lane lr holds entry 16 w + lr), M given by strip w of M_t, x in row layout
template <int NT>
__device__ __forceinline__ double smv(const Geo &g, const Strip<NT> &M, const Vec<NT> &xR) {
  double acc = 0.0;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = fma(M.t[a][r], xR.t[a][r], acc);
  acc += lane_xor16(acc, (g.lq & 1) != 0);
  acc += lane_xor32(acc, (g.lq & 2) != 0);
  return acc;
}
// the wave's 16 entries of a vector -> LDS buffer; the full vector in row layout <- LDS buffer (after a barrier)
__device__ __forceinline__ void vput(const Geo &g, int w, double *buf, double y) {
  if (g.lq == 0) buf[16 * w + g.lr] = y;
}
template <int NT>
__device__ __forceinline__ Vec<NT> vget(const Geo &g, const double *buf) {
  Vec<NT> x;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) x.t[a][r] = buf[g.row(a, r)];
  return x;
}
__device__ __forceinline__ void wg_sync() { __syncthreads(); }
// (diagnostic builds, tools/phase_stamps_rrs_wg.py: the barrier time of the doubling kernel as its own section, id 26)
#ifdef MOMR_DIAG_STAMPS
#define WG_SYNC_ST() do { wg_sync(); MOMR_STAMP_NW(26); } while (0)
#define MOMR_STAMP_INIT()                                                                \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == (gridDim.x >> 1)) {                            \
      unsigned long long n__;                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");      \
      momr_diag_last = n__;                                                              \
    }                                                                                    \
  } while (0)
#else
#define WG_SYNC_ST() wg_sync()
#define MOMR_STAMP_INIT()
#endif

// The inelastic elemental layer of one pair, strip w (ie_elem_tile restricted to tile column w; same expressions, same
// order): ier-+ and iet++ strips of the _t form, the wave's 16 entries of ieJ0+ / ieJ0-.
template <int NT>
__device__ __forceinline__ void ie_elem_strip(const Geo &g, int w, const KArgs &a, int n1, int dn, int n0, Strip<NT> &a_s, Strip<NT> &b_s,
                                              double &Jp, double &Jm) {
#pragma clang fp contract(off)
  const int N = a.N, n = a.nS;
  const double scl = (double)(1ull << a.sh);
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const double d1 = a.tau[n1] / scl, d0 = a.tau[n0] / scl, ratio = d1 / d0;
  const double pre = a.varpiR[dn] * a.varpi[n0] * a.fscatt[n0];
  const double fs0 = a.fscatt[n0], vR = a.varpiR[dn], v0 = a.varpi[n0];
  CV<NT> e0C, muC;
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) {
    const int i = g.col(tb);
    const double mu = (i < N) ? a.mu[i] : 1.0;
    muC.c[tb] = mu;
    e0C.c[tb] = exp(-d0 / mu);
  }
  const Vec<NT> e0R = c2r<NT>(g, e0C), muR = c2r<NT>(g, muC);
  const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
  const int base = (g.lq << 4);
  double e0s = 0.0, mus = 1.0;
#pragma unroll
  for (int tb = 0; tb < NT; ++tb)
    if ((i_start >> 4) == tb) {
      e0s = __shfl(e0C.c[tb], base | (i_start & 15));
      mus = __shfl(muC.c[tb], base | (i_start & 15));
    }
  // the wave's own column: i = 16 w + lr
  const int i = 16 * w + g.lr;
  const double mui = (i < N) ? a.mu[i] : 1.0;
  const double e1 = exp(-d1 / mui), e0i = exp(-d0 / mui);
#pragma unroll
  for (int ta = 0; ta < NT; ++ta)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = g.row(ta, rr);
      double r = 0.0, t = 0.0;
      if (i < N && j < N) {
        const double muj = muR.t[ta][rr], wj = a.wt[j] / wdiv;
        if (wj > 1.e-8) {
          const double e0 = e0R.t[ta][rr];
          r = fs0 * vR * v0 * a.Zr_mp[i + (size_t)N * j] * (1 / ((mui / muj) + ratio)) * (1 - e1 * e0) * wj;
          if (mui == muj) {
            if (i == j) {
              const double wi = a.wt[i] / wdiv;
              if (fabs(d0 - d1) > 1.e-6) t = pre * a.Zr_pp[i + (size_t)N * i] * wi * (e0i - e1) / (1 - ratio);
              else t = pre * a.Zr_pp[i + (size_t)N * i] * wi * (1 - e0i);
            }
          } else {
            t = pre * a.Zr_pp[i + (size_t)N * j] * (1 / ((mui / muj) - ratio)) * wj * (e1 - e0);
          }
        }
        if (scomp(i, n, a.strict_idx) > 2) r = -r;  // apply_D_elemental_RRS!, ndoubl >= 1
      }
      a_s.t[ta][rr] = r;
      b_s.t[ta][rr] = t;
    }
  const double att = exp(-a.tau_sum[n0] / mus);
  double jp = 0.0, jm = 0.0;
  if (i < N) {
    double zpI = 0.0, zmI = 0.0;
    for (int ii = i_start; ii < i_end; ++ii) {
      zpI += a.Zr_pp[i + (size_t)N * ii] * a.I0[ii - i_start];
      zmI += a.Zr_mp[i + (size_t)N * ii] * a.I0[ii - i_start];
    }
    if (i >= i_start && i < i_end) {
      if (fabs(d0 - d1) > 1.e-6) jp = (e0i - e1) / (ratio - 1) * pre * zpI * wct02;
      else jp = wct02 * pre * zpI * (1 - e0s);
    } else {
      jp = wct02 * pre * zpI * (1 / ((mui / mus) - ratio)) * (e1 - e0s);
    }
    jm = wct02 * pre * zmI * (1 / ((mui / mus) + ratio)) * (1 - e1 * e0s);
    jp *= att;
    jm = a.D[i % n] * (jm * att);
  }
  Jp = jp;
  Jm = jm;
}

// ---------------------------------------------------------------------------------------------------------------------
// doubling step, pair kernel, one workgroup of NT waves per pair (dbl_pair_body in strips)
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, bool FUSE, int MODE, bool NG>
__device__ __forceinline__ void dbl_pair_wg(const KArgs &a) {
  constexpr bool STRICT = (MODE == 2), fuseD = (MODE == 1);
  constexpr int MD = wg_mat_doubles<NT>();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Geo g;
  g.lr = lane & 15;
  g.lq = lane >> 4;
  g.N = a.N;
  g.xp = nullptr;
  g.ipiv = nullptr;
  double *S_a = reinterpret_cast<double *>(rrs_smem), *S_b = S_a + MD, *S_x = S_b + MD, *S_g = S_x + MD, *vb = S_g + MD;
  double *vb0 = vb, *vb1 = vb + 16 * NT, *vb2 = vb + 32 * NT, *vb3 = vb + 48 * NT, *vsv = vb + 64 * NT;
  const int n = a.nS, cw = 16 * w + g.lr;  // the lane's column of the strip / entry of a column-layout vector
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  const size_t span = (size_t)(a.n1_hi - a.n1_lo), npairs = span * a.nR;
  auto sgn_i = [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; };
  auto sgn_ij = [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; };
  // (Four forms of prefetching the NEXT pair's first strips were measured and dropped: profiles/r05_rrs_wg_ab.txt (4), (7).)
  MOMR_STAMP_INIT();
  for (size_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const int n1 = a.n1_lo + (int)(p % span), dn = (int)(p / span);
    const int n0 = n1 + a.off[dn];
    const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
    if (n0 < 0 || n0 >= a.S) {  // as in dbl_pair_body
      if (FUSE) {
        sstore<NT>(g, w, a.ie_a[R_MP] + o4, szeros<NT>());
        sstore<NT>(g, w, a.ie_a[T_PP] + o4, szeros<NT>());
        const double jm = a.D[cw % n] * a.ie_a[J0M][o3 + cw];
        if (g.lq == 0) a.ie_a[J0M][o3 + cw] = jm;
      }
      if (fuseD) {
        Strip<NT> an = FUSE ? szeros<NT>() : sload<NT>(g, w, a.ie_a[R_MP] + o4);
        Strip<NT> bn = FUSE ? szeros<NT>() : sload<NT>(g, w, a.ie_a[T_PP] + o4);
        if (n > 1) {
          smap<NT>(g, w, an, sgn_i);
          sstore<NT>(g, w, a.ie_a[R_MP] + o4, an);
          if (g.lq == 0) {  // one lane per entry reads and rewrites it (the FUSE store above came from the same lane)
            const double jm = a.ie_a[J0M][o3 + cw];
            if (scomp(cw, n, a.strict_idx) > 2) a.ie_a[J0M][o3 + cw] = -jm;
          }
          smap<NT>(g, w, an, sgn_ij);
          smap<NT>(g, w, bn, sgn_ij);
        }
        if (!a.derive_pm) {
          sstore<NT>(g, w, a.ie_a[R_PM] + o4, an);
          sstore<NT>(g, w, a.ie_a[T_MM] + o4, bn);
        }
      }
      continue;
    }
    MOMR_STAMP_NW(20);  // loop head, off-grid pairs
    const size_t m1 = NN * n1, m0 = NN * n0, v0 = VS * n0;
    // At the top only what the first product needs is requested as a burst -- the pair's ier-+ block and r[n0]^T (+ the source
    // entries); iet++ and r[n1] follow row by row between the MFMAs of the first product (StripReq), like every later strip.
    // (NT = 3, at its 256-register budget, keeps all four in the burst: the late form measured 2 % slower there.)
    constexpr bool LATE = (NT != 3);
    Strip<NT> a_s, b_s;
    double Jp = 0.0, Jm = 0.0;
    StripReq<NT> b_q, r1_q;
    if (!FUSE) {
      a_s = sload<NT>(g, w, a.ie_a[R_MP] + o4);
      Jp = a.ie_a[J0P][o3 + cw];
      Jm = a.ie_a[J0M][o3 + cw];
      if (LATE) b_q.init(g, w, a.ie_a[T_PP] + o4);
      else b_s = sload<NT>(g, w, a.ie_a[T_PP] + o4);
    }
    const Strip<NT> r0_s = sload<NT>(g, w, a.sm[SM_RT] + m0);
    if (LATE) r1_q.init(g, w, a.a_cur[R_MP] + m1);
    else r1_q.X = sload<NT>(g, w, a.a_cur[R_MP] + m1);
    const double e1 = a.expk_cur[n1];
    // Register economy: a strip that has a copy in LDS is READ BACK from there where it is needed again (a, b, bn), and the
    // strips from global memory are requested one or two products before their first use, not at the top.
    // The four source vectors of n0 (j1-, j0+, tmp1, tmp2) take ONE load instruction per wave -- lane (lq, lr) fetches entry
    // 16 w + lr of vector lq -- and reach their row layout through LDS (as 16 NT-entry row-layout loads each they were 64 NT
    // of the ~100 NT load instructions of a pair, against a 63-deep in-order counter of outstanding memory operations).
    const double *jp0 = STRICT ? a.jpseq + v0 + VS * a.S * dn : a.a_cur[J0P] + v0;
    // (NT = 3 keeps the row-layout loads: at its 256-register budget the LDS form spills more than it saves,
    // profiles/r05_rrs_wg_ab.txt)
    constexpr bool VLDS = (NT != 3);
    const double *vsrc = (g.lq == 0) ? a.sv[SV_J1M] + v0 : ((g.lq == 1) ? jp0 : ((g.lq == 2) ? a.sv[SV_TMP1] + v0 : a.sv[SV_TMP2] + v0));
    double vch = 0.0;
    Vec<NT> j1m0, jp0R, tm1, tm2;
    if (VLDS) {
      vch = vsrc[cw];
    } else {
      j1m0 = loadR<NT>(g, a.sv[SV_J1M] + v0); jp0R = loadR<NT>(g, jp0);
      tm1 = loadR<NT>(g, a.sv[SV_TMP1] + v0); tm2 = loadR<NT>(g, a.sv[SV_TMP2] + v0);
    }
    if (FUSE) ie_elem_strip<NT>(g, w, a, n1, dn, n0, a_s, b_s, Jp, Jm);
    const double J1p = Jp * e1, J1m = Jm * e1;  // ieJ1+, ieJ1-   :52-56
    MOMR_STAMP_NW(21);  // first loads issued, (fused elemental), wait for ieJ0+-
    WG_SYNC_ST();  // the previous pair has finished with the LDS matrices
    spublish<NT>(g, w, S_a, a_s);
    spublish<NT>(g, w, S_g, r0_s);
    if (!LATE || FUSE) spublish<NT>(g, w, S_b, b_s);  // (LATE: when it has arrived, with X)
    vput(g, w, vb0, J1m);
    if (VLDS) vsv[16 * NT * g.lq + cw] = vch;
    // the products of the pair's own blocks with the source vectors of n0                                         :61-89
    double a_j1m, a_jp, b1 = 0.0, b2 = 0.0;
    auto own_mv = [&]() {
      a_j1m = smv<NT>(g, a_s, j1m0);  // ier j1-[n0]
      a_jp = smv<NT>(g, a_s, jp0R);   // ier j0+[n0]
      if (!LATE || FUSE) {
        b1 = smv<NT>(g, b_s, tm1);    // iet++ tmp1
        b2 = STRICT ? smv<NT>(g, sload<NT>(g, w, a.ie_a[T_MM] + o4), tm2) : smv<NT>(g, b_s, tm2);  // D5: iet-- as the array holds it
      }
    };
    if (!VLDS) own_mv();
    Strip<NT> bmm_s;
    if (STRICT && LATE && !FUSE) bmm_s = sload<NT>(g, w, a.ie_a[T_MM] + o4);  // D5: iet-- as the array holds it
    StripReq<NT> gt0_q, ttgp1_q, gr0_q, t0_q, ttgpr1_q;  // requested row by row between the MFMAs of the products below
    gt0_q.init(g, w, a.sm[SM_GT] + m0); ttgp1_q.init(g, w, a.sm[SM_TTGP] + m1); gr0_q.init(g, w, a.sm[SM_GR] + m0);
    t0_q.init(g, w, a.sm[SM_TT] + m0); ttgpr1_q.init(g, w, a.sm[SM_TTGPR] + m1);
    MOMR_STAMP_NW(23);  // publish a, b, r0 (waits for the pair's blocks)
    WG_SYNC_ST();
    if (VLDS) {
      j1m0 = vget<NT>(g, vsv); jp0R = vget<NT>(g, vsv + 16 * NT);
      tm1 = vget<NT>(g, vsv + 32 * NT); tm2 = vget<NT>(g, vsv + 48 * NT);
      own_mv();
    }
    // X = ier r0 + r1 ier
    Strip<NT> X_s = sTNacc_job<NT, false, NG>(g, S_g, a_s, szeros<NT>(), [&](int tk, int s) {                          // U = r0_c
      if (LATE) r1_q.step(tk, s);
      if (LATE && !FUSE) b_q.step(tk, s);
    });
    const Strip<NT> &r1_s = r1_q.X;
    if (LATE && !FUSE) b_s = b_q.X;
    X_s = sTNacc_job<NT, true, NG>(g, S_a, r1_s, X_s, [&](int tk, int s) {                                            // U = a_c = (a_t)^T
      gt0_q.step(tk, s);
      ttgp1_q.step(tk, s);
    });
    const Strip<NT> &gt0_s = gt0_q.X, &ttgp1_s = ttgp1_q.X;
    const double X1 = smv<NT>(g, X_s, tm1), X2 = smv<NT>(g, X_s, tm2);
    if (LATE && !FUSE) {
      b1 = smv<NT>(g, b_s, tm1);                                                     // iet++ tmp1
      b2 = STRICT ? smv<NT>(g, bmm_s, tm2) : smv<NT>(g, b_s, tm2);
    }
    MOMR_STAMP_NW(25);  // 4 + 2 mat-vecs, X: 2 products
    WG_SYNC_ST();  // r0 has been read by everybody
    spublish<NT>(g, w, S_x, X_s);
    spublish<NT>(g, w, S_g, gt0_s);
    if (LATE && !FUSE) spublish<NT>(g, w, S_b, b_s);
    MOMR_STAMP_NW(27);  // publications
    // ---- sources
    {
      const double uu = (Jp + smv<NT>(g, r1_s, vget<NT>(g, vb0))) + (a_j1m + X1);
      vput(g, w, vb1, uu);
      WG_SYNC_ST();  // (also: X and (G t)[n0] are in LDS)
      const double Jpn = (J1p + smv<NT>(g, ttgp1_s, vget<NT>(g, vb1))) + b1;  // new ieJ0+
      vput(g, w, vb2, Jpn);
      WG_SYNC_ST();
      const double rv2 = smv<NT>(g, r1_s, vget<NT>(g, vb2));                  // r1 ieJ0+(new)
      const double u2 = (J1m + rv2) + (a_jp + X2);
      vput(g, w, vb3, u2);
      WG_SYNC_ST();
      double Jmn = (Jm + smv<NT>(g, ttgp1_s, vget<NT>(g, vb3))) + b2;         // new ieJ0-
      if (fuseD && n > 1 && scomp(cw, n, a.strict_idx) > 2) Jmn = -Jmn;
      if (g.lq == 0) {
        a.ie_a[J0P][o3 + cw] = Jpn;
        a.ie_a[J0M][o3 + cw] = Jmn;
      }
    }
    MOMR_STAMP_NW(24);  // source chain: 4 mat-vecs, 3 exchanges (their barriers: 26)
    // ---- operators                                                                                              :98-125
    const Strip<NT> Y_s = sTNacc_job<NT, false, NG>(g, S_x, sread<NT>(g, w, S_g), szeros<NT>(),  // Y_c = X G t[n0]   (U = X_t)
                                                [&](int tk, int s) { gr0_q.step(tk, s); });   // (G r)[n0]: consumed three products later
    const Strip<NT> &gr0_s = gr0_q.X;
    MOMR_STAMP_NW(28);  // Y: 1 product
    WG_SYNC_ST();                                                              // X has been read
    spublish<NT>(g, w, S_x, Y_s);
    MOMR_STAMP_NW(27);
    WG_SYNC_ST();
    Strip<NT> bn_s = sTNacc_sum_job<NT, NG>(g, S_b, S_x, ttgp1_s, szeros<NT>(),   // tG (iet + Y)            (U = W_c = b_c + Y_c)
                                        [&](int tk, int s) { t0_q.step(tk, s); });
    bn_s = sTNacc_job<NT, false, NG>(g, S_g, sread<NT>(g, w, S_b), bn_s,           // + iet G t[n0]           (U = (G t)[n0]_c)
                                 [&](int tk, int s) { ttgpr1_q.step(tk, s); });
    const Strip<NT> &t0_s = t0_q.X, &ttgpr1_s = ttgpr1_q.X;
    MOMR_STAMP_NW(29);  // iet: 2 products
    WG_SYNC_ST();                                                              // b and (G t)[n0] have been read
    spublish<NT>(g, w, S_b, bn_s);
    spublish<NT>(g, w, S_g, gr0_s);
    MOMR_STAMP_NW(27);
    WG_SYNC_ST();
    // (the new iet++ is final: its stores go out row by row between the MFMAs of this product, not in the burst at the end)
    double *bn_st = a.ie_a[T_PP] + o4 + 16 * w + g.lr + (16 * NT) * g.lq;
    Strip<NT> Q_s = sTNacc_job<NT, false, NG>(g, S_g, bn_s, szeros<NT>(),      // iet(new) G r[n0]        (U = (G r)[n0]_c)
                                              [&](int tk, int s) { bn_st[(16 * NT) * (16 * tk + 4 * s)] = bn_s.t[tk][s]; });
    Q_s = sTNacc<NT, true, NG>(g, S_a, ttgp1_s, Q_s);                       // + tG ier                (U = a_c)
    MOMR_STAMP_NW(30);  // Q: 2 products
    WG_SYNC_ST();                                                              // (G r)[n0] has been read
    spublish<NT>(g, w, S_g, t0_s);
    MOMR_STAMP_NW(27);
    WG_SYNC_ST();
    Strip<NT> an_s = sTNacc_sum<NT, NG>(g, S_b, S_x, ttgpr1_s, szeros<NT>());  // (iet(new) + Y) ...      (U = V_c = bn_c + Y_c)
    an_s = sTNacc<NT, false, NG>(g, S_g, Q_s, an_s);                           // + t[n0]-side product    (U = t0_c)
    an_s = sadd<NT>(sread<NT>(g, w, S_a), an_s);
    MOMR_STAMP_NW(31);  // ier: 2 products, read-back of a
    if (fuseD) {  // apply_D_matrix_IE!, corrected indexing (D2)
      if (n > 1) smap<NT>(g, w, an_s, sgn_i);
      Strip<NT> apm = an_s, bmm = sread<NT>(g, w, S_b);
      if (n > 1) {
        smap<NT>(g, w, apm, sgn_ij);
        smap<NT>(g, w, bmm, sgn_ij);
      }
      if (!a.derive_pm) {
        sstore<NT>(g, w, a.ie_a[R_PM] + o4, apm);
        sstore<NT>(g, w, a.ie_a[T_MM] + o4, bmm);
      }
    }
    sstore<NT>(g, w, a.ie_a[R_MP] + o4, an_s);
    MOMR_STAMP_NW(32);  // D signs, operator stores issued
  }
}

// column strip w of the _c form (the transpose of what sload returns) straight from global memory: lane groups read 32-byte
// pieces of 16 lines per instruction, the 4 NT instructions of a strip consume every line of the strip's 8 NT x 128 B region
template <int NT>
__device__ __forceinline__ Strip<NT> sloadT(const Geo &g, int w, const double *p) {
  Strip<NT> X;
  const double *q = p + g.lq + (size_t)(16 * NT) * (16 * w + g.lr);
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) X.t[a][r] = q[16 * a + 4 * r];
  return X;
}

// ---------------------------------------------------------------------------------------------------------------------
// interaction, ScatteringInterface_11, pair kernel, one workgroup of NT waves per pair (int_pair_body, iface == 3, in strips).
// Four LDS matrices P, Q, R, T take the twelve left factors of the eighteen products in turn (in brackets: the slot):
//   M1 = a Rpm0 + r1 E [a: P, r1: Q]      N1 = a Tpp0 + r1 C [P, Q]        A = ieT-- + M1 T01 [M1: P]
//   ieR-+ += N1 T01 [N1: Q] + G1RT0 A [T]     ieT-- = bm^T T01 [bm: R, read transposed] + G1T0 A [P]
//   M2 = E R0 [E: Q] + Rpm1 a^T [Rpm1: R]     N2 = E TMM0 [Q] + Rpm1 bm^T [R]     B = iet++ + M2 T21 [M2: P]
//   ieR+- = ier+- + N2 T21 [N2: Q] + G2RT0 B [T]     ieT++ = C^T T21 [C: R, read transposed] + G2T0 B [P]
// (written with the operand names of int_pair_body; every product in its order).  Right factors that are transposes of
// blocks in memory (E^T, C^T, a^T, bm^T) are loaded as transposed strips (sloadT).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, bool SURF, bool DERIVE>
__device__ __forceinline__ void int_pair_wg(const KArgs &a) {
  constexpr int MD = wg_mat_doubles<NT>();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Geo g;
  g.lr = lane & 15;
  g.lq = lane >> 4;
  g.N = a.N;
  g.xp = nullptr;
  g.ipiv = nullptr;
  double *S_p = reinterpret_cast<double *>(rrs_smem), *S_q = S_p + MD, *S_r = S_q + MD, *S_t = S_r + MD, *vb = S_t + MD;
  double *vb0 = vb, *vb1 = vb + 16 * NT;
  const int n = a.nS, cw = 16 * w + g.lr;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  const size_t span = (size_t)(a.n1_hi - a.n1_lo), npairs = span * a.nR;
  auto sgn_ij = [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; };
  auto flip = [&](Strip<NT> X) {  // ier+- = sgn (.) ier-+, iet-- = sgn (.) iet++ (corrected D2); sgn is symmetric in (i, j)
    if (n > 1) smap<NT>(g, w, X, sgn_ij);
    return X;
  };
  for (size_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const int n1 = a.n1_lo + (int)(p % span), dn = (int)(p / span);
    const int n0 = n1 + a.off[dn];
    if (n0 < 0 || n0 >= a.S) continue;
    const size_t u = (size_t)n1 + (size_t)a.S * dn, o4 = NN * u, o3 = VS * u;
    const size_t m1 = NN * n1, m0 = NN * n0, v0 = VS * n0;
    const double *pa = a.ie_a[R_MP] + o4, *pbm = a.ie_a[DERIVE ? T_PP : T_MM] + o4;
    // ---- stage 0: a -> P, r1 -> Q.  Only these two strips and R+-[n0] are requested as a burst; the other six strips of the first
    // four products (and bm, G1RT0 for the second LDS round) follow row by row between the MFMAs of the products before their use
    const Strip<NT> a_s = SURF ? szeros<NT>() : sload<NT>(g, w, pa);
    const Strip<NT> r1_s = sload<NT>(g, w, a.x[R_MP] + m1);
    const Strip<NT> Rpm0_s = sload<NT>(g, w, a.sm[SI_RPM] + m0);
    // (4 x 4 tiles only: -4 % there, nothing at 2 x 2, and 3 x 3 at its 256-register budget spills and loses 7 %:
    // profiles/r05_rrs_wg_ab.txt (16); the other tile counts request the same strips in two bursts)
    constexpr bool LATE = (NT == 4);
    StripReq<NT> Tpp0_q, T01_q, A_q, bm_q, g1rt_q;
    StripReqT<NT> Ec_q, Cc_q;
    if (LATE) {
      Tpp0_q.init(g, w, a.sm[SI_TPP] + m0); T01_q.init(g, w, a.sm[SI_T01] + m1); A_q.init(g, w, a.ie_c[C_T_MM] + o4);
      bm_q.init(g, w, pbm); g1rt_q.init(g, w, a.sm[SI_G1RT] + m0);
      Ec_q.init(g, w, a.ie_c[C_R_PM] + o4); Cc_q.init(g, w, a.ie_c[C_T_PP] + o4);
    } else {
      if (!SURF) bm_q.X = sload<NT>(g, w, pbm);
      g1rt_q.X = sload<NT>(g, w, a.sm[SI_G1RT] + m0);
      Tpp0_q.X = sload<NT>(g, w, a.sm[SI_TPP] + m0);
    }
    const double Jap = SURF ? 0.0 : a.ie_a[J0P][o3 + cw], Jam = SURF ? 0.0 : a.ie_a[J0M][o3 + cw];
    const double Jcp = a.ie_c[C_J0P][o3 + cw], Jcm = a.ie_c[C_J0M][o3 + cw];
    const double v1 = smv<NT>(g, a_s, loadR<NT>(g, a.c_cur[C_J0P] + v0));   // ier J0+[n0]
    const double v2 = smv<NT>(g, r1_s, loadR<NT>(g, a.ie_c[C_J0P] + o3));   // r ieJ0+
    const double uu1 = (v1 + v2) + Jam;
    wg_sync();  // the previous pair has finished with the LDS matrices
    spublish<NT>(g, w, S_p, a_s);
    spublish<NT>(g, w, S_q, r1_s);
    vput(g, w, vb0, uu1);
    if (!LATE) {
      spublish<NT>(g, w, S_r, (DERIVE && !SURF) ? flip(bm_q.X) : (SURF ? szeros<NT>() : bm_q.X));
      spublish<NT>(g, w, S_t, g1rt_q.X);
      Ec_q.X = sloadT<NT>(g, w, a.ie_c[C_R_PM] + o4);
      Cc_q.X = sloadT<NT>(g, w, a.ie_c[C_T_PP] + o4);
    }
    wg_sync();
    // A = T01 (ier R+-[n0] + r ieR+-) + ieT--                                                                     :252-262
    Strip<NT> M1_s = sTNacc_job<NT, false, false>(g, S_p, Rpm0_s, szeros<NT>(), [&](int tk, int s) { if (LATE) { Tpp0_q.step(tk, s); Ec_q.step(tk, s); } });
    Strip<NT> N1_s = sTNacc_job<NT, false, false>(g, S_p, Tpp0_q.X, szeros<NT>(), [&](int tk, int s) { if (LATE) { Cc_q.step(tk, s); T01_q.step(tk, s); } });
    if (!LATE) T01_q.X = sload<NT>(g, w, a.sm[SI_T01] + m1);
    M1_s = sTNacc_job<NT, false, false>(g, S_q, Ec_q.X, M1_s, [&](int tk, int s) { if (LATE) { A_q.step(tk, s); if (!SURF) bm_q.step(tk, s); } });
    if (!LATE) A_q.X = sload<NT>(g, w, a.ie_c[C_T_MM] + o4);
    N1_s = sTNacc_job<NT, false, false>(g, S_q, Cc_q.X, N1_s, [&](int tk, int s) { if (LATE) g1rt_q.step(tk, s); });
    const Strip<NT> &T01_s = T01_q.X;
    Strip<NT> A_s = A_q.X;
    wg_sync();  // a and r have been read
    spublish<NT>(g, w, S_p, M1_s);
    spublish<NT>(g, w, S_q, N1_s);
    if (LATE) {
      spublish<NT>(g, w, S_r, (DERIVE && !SURF) ? flip(bm_q.X) : (SURF ? szeros<NT>() : bm_q.X));
      spublish<NT>(g, w, S_t, g1rt_q.X);
    }
    wg_sync();
    A_s = sTNacc<NT, false>(g, S_p, T01_s, A_s);
    // ieJ0- += T01 (ier J0+[n0] + r ieJ0+ + ieJ0-(added)) + A G1 (j0-[n0] + r[n0] J0+[n0])                         :251-264
    {
      const double wv = smv<NT>(g, T01_s, vget<NT>(g, vb0)) + smv<NT>(g, A_s, loadR<NT>(g, a.sv[SVI_G1V] + v0));
      if (g.lq == 0) a.ie_c[C_J0M][o3 + cw] = Jcm + wv;
    }
    // ieR-+ += T01 (ier T++[n0] + r ieT++) + A G1 r[n0] T++[n0];  ieT-- = T01 iet-- + A G1 t--[n0]                  :271-284
    Strip<NT> Rm_s = sload<NT>(g, w, a.ie_c[C_R_MP] + o4);
    const Strip<NT> g1t_s = sload<NT>(g, w, a.sm[SI_G1T] + m0);
    Rm_s = sTNacc<NT, false>(g, S_q, T01_s, Rm_s);
    Rm_s = sTNacc<NT, false>(g, S_t, A_s, Rm_s);
    double *Rm_st = a.ie_c[C_R_MP] + o4 + 16 * w + g.lr + (16 * NT) * g.lq;  // (LATE: stored row by row inside the next product)
    if (!LATE) sstore<NT>(g, w, a.ie_c[C_R_MP] + o4, Rm_s);
    // the operands of the second half (E as a left factor, R+-[n1], G2 R+-[n0] t--[n0])
    const Strip<NT> E_s = sload<NT>(g, w, a.ie_c[C_R_PM] + o4), Rpm1_s = sload<NT>(g, w, a.c_cur[C_R_PM] + m1);
    const Strip<NT> g2rt_s = sload<NT>(g, w, a.sm[SI_G2RT] + m0);
    Strip<NT> F_s = sTNacc_job<NT, true, false>(g, S_r, T01_s, szeros<NT>(),  // U = bm_c = (bm_t)^T
                                                [&](int tk, int s) { if (LATE) Rm_st[(16 * NT) * (16 * tk + 4 * s)] = Rm_s.t[tk][s]; });
    wg_sync();  // M1, N1, bm, G1RT0 have been read
    spublish<NT>(g, w, S_p, g1t_s);
    spublish<NT>(g, w, S_q, E_s);
    spublish<NT>(g, w, S_r, Rpm1_s);
    spublish<NT>(g, w, S_t, g2rt_s);
    {
      const double v3 = smv<NT>(g, E_s, loadR<NT>(g, a.x[J0M] + v0));                              // ieR+- j0-[n0]
      const double v4 = SURF ? 0.0 : smv<NT>(g, Rpm1_s, loadR<NT>(g, a.ie_a[J0M] + o3));           // R+- ieJ0-(added)
      vput(g, w, vb1, (Jcp + v3) + v4);
    }
    const Strip<NT> R0_s = sload<NT>(g, w, a.sm[SI_R] + m0), Tmm0_s = sload<NT>(g, w, a.sm[SI_TMM] + m0);
    wg_sync();
    F_s = sTNacc<NT, false>(g, S_p, A_s, F_s);
    sstore<NT>(g, w, a.ie_c[C_T_MM] + o4, F_s);
    // B = T21 (ieR+- r[n0] + R+- ier) + iet++                                                                     :302-310
    Strip<NT> ac_s = SURF ? szeros<NT>() : sloadT<NT>(g, w, pa);
    Strip<NT> bmc_s = SURF ? szeros<NT>() : sloadT<NT>(g, w, pbm);
    if (DERIVE && !SURF) bmc_s = flip(bmc_s);
    Strip<NT> M2_s = sTNacc<NT, false>(g, S_q, R0_s, szeros<NT>());
    Strip<NT> N2_s = sTNacc<NT, false>(g, S_q, Tmm0_s, szeros<NT>());
    const Strip<NT> T21_s = sload<NT>(g, w, a.sm[SI_T21] + m1);
    M2_s = sTNacc<NT, false>(g, S_r, ac_s, M2_s);
    N2_s = sTNacc<NT, false>(g, S_r, bmc_s, N2_s);
    Strip<NT> B_s = SURF ? szeros<NT>() : sload<NT>(g, w, a.ie_a[T_PP] + o4);
    const Strip<NT> C_s = sload<NT>(g, w, a.ie_c[C_T_PP] + o4);
    wg_sync();  // G1T0, E, R+-[n1] have been read
    spublish<NT>(g, w, S_p, M2_s);
    spublish<NT>(g, w, S_q, N2_s);
    spublish<NT>(g, w, S_r, C_s);
    wg_sync();
    B_s = sTNacc<NT, false>(g, S_p, T21_s, B_s);
    // ieJ0+ = ieJ0+(added) + T21 (ieJ0+ + ieR+- j0-[n0] + R+- ieJ0-(added)) + B G2 (J0+[n0] + R+-[n0] j0-[n0])      :301-312
    {
      const double wv = smv<NT>(g, T21_s, vget<NT>(g, vb1)) + smv<NT>(g, B_s, loadR<NT>(g, a.sv[SVI_G2V] + v0));
      if (g.lq == 0) a.ie_c[C_J0P][o3 + cw] = Jap + wv;
    }
    // ieT++ = T21 ieT++ + B G2 T++[n0];  ieR+- = ier+- + T21 (ieR+- t--[n0] + R+- iet--) + B G2 R+-[n0] t--[n0]     :320-334
    Strip<NT> En_s = SURF ? szeros<NT>() : (DERIVE ? flip(sload<NT>(g, w, pa)) : sload<NT>(g, w, a.ie_a[R_PM] + o4));
    const Strip<NT> g2t_s = sload<NT>(g, w, a.sm[SI_G2T] + m0);
    En_s = sTNacc<NT, false>(g, S_q, T21_s, En_s);
    En_s = sTNacc<NT, false>(g, S_t, B_s, En_s);
    sstore<NT>(g, w, a.ie_c[C_R_PM] + o4, En_s);
    Strip<NT> Cn_s = sTNacc<NT, true>(g, S_r, T21_s, szeros<NT>());  // U = C_c = (C_t)^T
    wg_sync();  // M2 has been read
    spublish<NT>(g, w, S_p, g2t_s);
    wg_sync();
    Cn_s = sTNacc<NT, false>(g, S_p, B_s, Cn_s);
    sstore<NT>(g, w, a.ie_c[C_T_PP] + o4, Cn_s);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// doubling step, POINT kernel above N = 32 as one workgroup of NT waves per spectral point (k_dbl_point in strips): the
// wave-per-point form keeps r, t, their transposes, G and four products as NT x NT register tiles -- 1 000+ registers per lane
// at NT = 3 / 4, i.e. scratch memory -- and a launch has only S waves.  Here r, t, G = (I - r r)^-1 and (t G)^T live in four LDS
// matrices, every product is computed in column strips (transposed factors are the transposed index expression of the same
// copy), the Gauss-Jordan inverse runs on all NT waves (wg_inv_one_minus: a wave holds 16 columns of every matrix row), and the
// 2-column source products are split by row blocks with the full vectors exchanged through LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ Strip<NT> sreadT(const Geo &g, int w, const double *M) {  // strip w of M^T from the LDS copy of M
  Strip<NT> X;
  const double *q = M + w * NT * kTileDoubles + g.lr * kTileLd + g.lq;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) X.t[a][r] = q[a * kTileDoubles + 4 * r];
  return X;
}
template <int NT>
__device__ __forceinline__ Mat<NT> mread(const Geo &g, const double *M) {
  Mat<NT> X;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) X.t[a][b][r] = M[(a * NT + b) * kTileDoubles + (g.lq + 4 * r) * kTileLd + g.lr];
  return X;
}
template <int NT>
__device__ __forceinline__ void mpublish(const Geo &g, double *M, const Mat<NT> &X) {
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) M[(a * NT + b) * kTileDoubles + (g.lq + 4 * r) * kTileLd + g.lr] = X.t[a][b][r];
}
// row block w of U^T v (TNv of mom_tile.hpp restricted to ti = w), U = M or M^T from LDS, v = the 2 (.. 16) vectors held as tile columns
template <int NT, bool TR>
__device__ __forceinline__ d4 sTNv(const Geo &g, int w, const double *M, const Vec<NT> &v) {
  d4 o = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (16 * tk + 4 * s < g.N) o = __builtin_amdgcn_mfma_f64_16x16x4f64(ufrag<NT, TR>(g, M, tk, w, s), v.t[tk][s], o, 0, 0, 0);
  return o;
}
template <int NT>
__device__ __forceinline__ d4 vpick(const Vec<NT> &v, int w) {  // v.t[w] without indexing the register array
  d4 o = v.t[0];
#pragma unroll
  for (int a = 1; a < NT; ++a)
    if (a == w) o = v.t[a];
  return o;
}
// row block w of a vector tile -> LDS; all row blocks <- LDS (16 x 16 tiles of pitch kTileLd)
__device__ __forceinline__ void vxput(const Geo &g, int w, double *buf, const d4 &o) {
#pragma unroll
  for (int r = 0; r < 4; ++r) buf[w * kTileDoubles + (g.lq + 4 * r) * kTileLd + g.lr] = o[r];
}
template <int NT>
__device__ __forceinline__ Vec<NT> vxget(const Geo &g, const double *buf) {
  Vec<NT> v;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) v.t[a][r] = buf[a * kTileDoubles + (g.lq + 4 * r) * kTileLd + g.lr];
  return v;
}
constexpr int kWgInvDoubles = 32 + 128 + 2 + 2;  // ipiv (64 ints), two multiplier columns, two reciprocal pivots, two pivot rows (ints)
template <int NT>
constexpr size_t wg_point_lds_bytes() { return ((size_t)4 * wg_mat_doubles<NT>() + 2 * NT * kTileDoubles + kWgInvDoubles) * 8; }

// (I - B)^-1 IN PLACE in the LDS matrix S (tile layout; B on entry), by all NT waves of the workgroup: Gauss-Jordan elimination
// with implicit partial pivoting as inv_one_minus of mom_tile.hpp (one matrix row per lane, the same operations on every element
// in the same order: the same result), but wave w holds only columns [16 w, 16 w + 16) of its row -- 16 registers instead of
// 16 NT, nothing in scratch.  Per step the wave that owns column k finds the pivot row and publishes the multiplier column, the
// pivot's lane and reciprocal through LDS (double-buffered: ONE barrier per step); every wave then updates its 16 columns.
// *bad_out = 1 + the step of a zero pivot.  `aux`: kWgInvDoubles doubles of LDS.
template <int NT>
__device__ __forceinline__ void wg_inv_one_minus(const Geo &g, int w, double *S, double *aux, int *bad_out) {
  const int N = g.N, lane = 16 * g.lq + g.lr;
  int *ipiv = reinterpret_cast<int *>(aux);
  double *colbuf = aux + 32, *dbuf = aux + 160;
  int *plbuf = reinterpret_cast<int *>(aux + 162);
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = 16 * w + c;
    const double id = (lane == col) ? 1.0 : 0.0;
    v[c] = (lane < N && col < N) ? id - S[((lane >> 4) * NT + w) * kTileDoubles + (lane & 15) * kTileLd + c] : id;
  }
  bool used = false;
  int myk = lane, bad = 0, par = 0;
#pragma unroll
  for (int kb = 0; kb < NT; ++kb)
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      const int k = 16 * kb + kc;
      if (k < N) {
        if (w == kb) {  // the owner of column k
          const int ah = (!used && lane < N) ? __double2hiint(fabs(v[kc])) : -1;
          int mh = ah;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) mh = max(mh, __shfl_xor(mh, off));
          const unsigned long long mk = __ballot(ah == mh);
          const int pl = __ffsll((long long)mk) - 1;
          const double piv = __shfl(v[kc], pl);
          if (!(fabs(piv) > 0.0) && !bad) bad = k + 1;
          colbuf[64 * par + lane] = v[kc];
          if (lane == 0) {
            dbuf[par] = 1.0 / piv;
            plbuf[par] = pl;
          }
        }
        wg_sync();
        const int pl = plbuf[par];
        const double d = dbuf[par], f = colbuf[64 * par + lane];
        const bool isp = (lane == pl);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double prow = __shfl(v[c], pl) * d;
          v[c] = isp ? prow : (v[c] - f * prow);
        }
        if (w == kb) v[kc] = isp ? d : (-f * d);
        if (isp) { used = true; myk = k; }
        if (threadIdx.x == 0) ipiv[k] = pl;
        par ^= 1;
      }
    }
  wg_sync();  // all reads of S (at the top) and all ipiv entries are done
  spublish<NT>(g, w, S, szeros<NT>());
  wg_sync();
  // inv(A)[k][p_j] = S[p_k][j]: lane (row p_k, pivot of step myk) writes row myk with permuted columns
  if (lane < N) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = 16 * w + c;
      if (col < N) {
        const int pc = ipiv[col];
        S[((myk >> 4) * NT + (pc >> 4)) * kTileDoubles + (myk & 15) * kTileLd + (pc & 15)] = v[c];
      }
    }
  }
  wg_sync();
  if (bad) *bad_out = bad;
}

template <int NT>
__device__ __forceinline__ void dbl_point_wg(const KArgs &a) {
  constexpr int MD = wg_mat_doubles<NT>();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *S_r = reinterpret_cast<double *>(rrs_smem), *S_t = S_r + MD, *S_G = S_t + MD, *S_x = S_G + MD;
  double *vx = S_x + MD, *vx2 = vx + NT * kTileDoubles;
  Geo g;
  g.lr = lane & 15;
  g.lq = lane >> 4;
  g.N = a.N;
  g.xp = nullptr;
  g.ipiv = nullptr;
  double *inv_aux = vx2 + NT * kTileDoubles;                  // small buffers of the workgroup inverse
  const int n = a.nS;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  int bad = 0;
  auto sgn_i = [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; };
  auto sgn_ij = [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; };
  // rows [16 w, 16 w + 16) of the vectors in columns 0 / 1 of a row block -> two arrays
  auto store_rows = [&](double *p0, double *p1, const d4 &o) {
    double *p = (g.lr == 0) ? p0 : ((g.lr == 1) ? p1 : nullptr);
    if (p != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[16 * w + g.lq + 4 * r] = o[r];
    }
  };
  for (int pt = blockIdx.x; pt < a.S; pt += gridDim.x) {
    const size_t om = NN * pt, ov = VS * pt;
    const Strip<NT> r_s = sload<NT>(g, w, a.a_cur[R_MP] + om), t_s = sload<NT>(g, w, a.a_cur[T_PP] + om);
    double e = a.expk_cur[pt];
    Vec<NT> J = loadv2<NT>(g, a.a_cur[J0P] + ov, a.a_cur[J0M] + ov);      // (j0+ | j0-)
    wg_sync();  // the previous point has finished with the LDS matrices
    spublish<NT>(g, w, S_r, r_s);
    spublish<NT>(g, w, S_t, t_s);
    wg_sync();
    const Strip<NT> rc_s = sreadT<NT>(g, w, S_r), tc_s = sreadT<NT>(g, w, S_t);
    sstore<NT>(g, w, a.sm[SM_RT] + om, rc_s);
    sstore<NT>(g, w, a.sm[SM_TT] + om, tc_s);
    spublish<NT>(g, w, S_G, sTNacc<NT, false>(g, S_r, rc_s, szeros<NT>()));  // r r (as TN(r_t, r_c))
    wg_sync();
    wg_inv_one_minus<NT>(g, w, S_G, inv_aux, &bad);                           // (I - r r)^-1, in place             :47
    const Strip<NT> ttgp_s = sTNacc<NT, false>(g, S_G, t_s, szeros<NT>());    // (t G)^T = TN(G_c, t_t)            :48
    const Strip<NT> ttgpr_s = sTNacc<NT, true>(g, S_r, ttgp_s, szeros<NT>()); // (t G r)^T = TN(r_c, .)
    sstore<NT>(g, w, a.sm[SM_GT] + om, sTNacc<NT, true>(g, S_G, tc_s, szeros<NT>()));  // G t, row-major: TN(G_t, t_c)
    sstore<NT>(g, w, a.sm[SM_GR] + om, sTNacc<NT, true>(g, S_G, rc_s, szeros<NT>()));  // G r, row-major
    sstore<NT>(g, w, a.sm[SM_TTGP] + om, ttgp_s);
    sstore<NT>(g, w, a.sm[SM_TTGPR] + om, ttgpr_s);
    spublish<NT>(g, w, S_x, ttgp_s);  // (the inverse's workspace is free again)
    const Vec<NT> J1 = vscale<NT>(J, e);                                      // (j1+ | j1-)                       :51-55
    store_rows(a.sv[SV_J1P] + ov, a.sv[SV_J1M] + ov, vpick<NT>(J1, w));
    auto mix = [&](const Vec<NT> &Jc) {
      Vec<NT> m;  // (j0+ | j1-)
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) m.t[ta] = (g.lr == 0) ? Jc.t[ta] : J1.t[ta];
      return m;
    };
    // s = (j0+ + r j1- | j1- + r j0+): every wave its row block, the full tile through LDS
    auto s_of = [&](const Vec<NT> &Jc) {
      const Vec<NT> mx = mix(Jc);
      vxput(g, w, vx, vpick<NT>(mx, w) + sTNv<NT, false>(g, w, S_r, swap01<NT>(mx)));  // + r (j1- | j0+)
      wg_sync();
      return vxget<NT>(g, vx);
    };
    {
      const Vec<NT> s = s_of(J);  // (the barrier inside also publishes (t G)^T)
      const d4 tmp = sTNv<NT, true>(g, w, S_G, s);                            // tmp = G s = (tmp1 | tmp2)          :58-59
      store_rows(a.sv[SV_TMP1] + ov, a.sv[SV_TMP2] + ov, tmp);
      // elastic source update: once (corrected) or nRaman times with expk squared every time (strict, D1)        :90-95
      const int reps = a.strict_rrs ? a.nR : 1;
      Vec<NT> sk = s;
      for (int k = 0; k < reps; ++k) {
        if (a.strict_rrs) store_rows(a.jpseq + ov + VS * a.S * k, nullptr, vpick<NT>(J, w));
        if (k > 0) {
          wg_sync();  // everybody has read the previous s
          sk = s_of(J);
        }
        const d4 q = sTNv<NT, false>(g, w, S_x, sk);                         // (tG (j0+ + r j1-) | tG (j1- + r j0+))
        const d4 j1b = vpick<NT>(J1, w), jb = vpick<NT>(J, w);
        d4 jn;
#pragma unroll
        for (int r = 0; r < 4; ++r) jn[r] = ((g.lr == 0) ? j1b[r] : jb[r]) + q[r];
        if (k > 0) wg_sync();  // ... and the previous J
        vxput(g, w, vx2, jn);
        wg_sync();
        J = vxget<NT>(g, vx2);
        e = e * e;
      }
    }
    // r <- r + (tG r) t,  t <- tG t                                                                             :128-131
    Strip<NT> rn_s = sadd<NT>(sread<NT>(g, w, S_r), sTNacc<NT, true>(g, S_t, ttgpr_s, szeros<NT>()));  // U = t_c
    Strip<NT> tn_s = sTNacc<NT, true>(g, S_t, ttgp_s, szeros<NT>());
    d4 jout = vpick<NT>(J, w);
    if (a.last) {  // apply_D_matrix! (doubling.jl:93-134) and apply_D_matrix_SFI! (:112-144)
      if (n > 1) {
        smap<NT>(g, w, rn_s, sgn_i);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (g.lr == 1 && scomp(16 * w + g.lq + 4 * r, n, a.strict_idx) > 2) jout[r] = -jout[r];
      }
      Strip<NT> rpm = rn_s, tmm = tn_s;
      if (n > 1) {
        smap<NT>(g, w, rpm, sgn_ij);
        smap<NT>(g, w, tmm, sgn_ij);
      }
      sstore<NT>(g, w, a.a_nxt[R_PM] + om, rpm);
      sstore<NT>(g, w, a.a_nxt[T_MM] + om, tmm);
    }
    sstore<NT>(g, w, a.a_nxt[R_MP] + om, rn_s);
    sstore<NT>(g, w, a.a_nxt[T_PP] + om, tn_s);
    store_rows(a.a_nxt[J0P] + ov, a.a_nxt[J0M] + ov, jout);
    if (threadIdx.x == 0) a.expk_nxt[pt] = e;
  }
  if (bad && lane == 0) atomicMax(a.info, bad);
}

// ---------------------------------------------------------------------------------------------------------------------
// interaction, ScatteringInterface_11, POINT kernel above N = 32 as one workgroup of NT waves per spectral point (k_int_point,
// iface == 3, in strips; interaction_inelastic.jl:244-340 elastic part).  Four LDS matrices in turn:
//   A: r | rT | Rt        B: R+-[comp]        C: (I - r R+-) -> G1 -> T01, (I - R+- r) -> G2 -> T21        D: t-- | T++[comp]
// Transposed right factors are transposed strips from global memory (sloadT) or from an LDS copy (sreadT).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void int_point_wg(const KArgs &a) {
  constexpr int MD = wg_mat_doubles<NT>();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *S_a = reinterpret_cast<double *>(rrs_smem), *S_b = S_a + MD, *S_c = S_b + MD, *S_d = S_c + MD;
  double *vx = S_d + MD, *vx2 = vx + NT * kTileDoubles, *inv_aux = vx2 + NT * kTileDoubles;
  Geo g;
  g.lr = lane & 15;
  g.lq = lane >> 4;
  g.N = a.N;
  g.xp = nullptr;
  g.ipiv = nullptr;
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  int bad = 0;
  auto store_rows = [&](double *p0, double *p1, const d4 &o) {  // columns 0 / 1 of a row block -> two arrays (nullptr: none)
    double *p = (g.lr == 0) ? p0 : ((g.lr == 1) ? p1 : nullptr);
    if (p != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[16 * w + g.lq + 4 * r] = o[r];
    }
  };
  auto swap01b = [&](const d4 &v) {  // exchange columns 0 <-> 1 of a row block
    d4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __shfl_xor(v[r], 1);
    return o;
  };
  for (int pt = blockIdx.x; pt < a.S; pt += gridDim.x) {
    const size_t om = NN * pt, ov = VS * pt;
    const Strip<NT> r_s = sload<NT>(g, w, a.x[R_MP] + om), Rpm_s = sload<NT>(g, w, a.c_cur[C_R_PM] + om);
    const Strip<NT> tmm_s = sload<NT>(g, w, a.x[T_MM] + om);
    const Vec<NT> ja = loadv2<NT>(g, a.x[J0P] + ov, a.x[J0M] + ov);                    // (j0+ | j0-) added
    const Vec<NT> Jc = loadv2<NT>(g, a.c_cur[C_J0P] + ov, a.c_cur[C_J0M] + ov);        // (J0+ | J0-) composite
    wg_sync();  // the previous point has finished with the LDS matrices
    spublish<NT>(g, w, S_a, r_s);
    spublish<NT>(g, w, S_b, Rpm_s);
    spublish<NT>(g, w, S_d, tmm_s);
    wg_sync();
    const Strip<NT> r_c = sreadT<NT>(g, w, S_a), Rpm_c = sreadT<NT>(g, w, S_b), tmm_c = sreadT<NT>(g, w, S_d);
    const Strip<NT> Tpp_c = sloadT<NT>(g, w, a.c_cur[C_T_PP] + om);
    sstore<NT>(g, w, a.sm[SI_RPM] + om, Rpm_c);
    sstore<NT>(g, w, a.sm[SI_TPP] + om, Tpp_c);
    sstore<NT>(g, w, a.sm[SI_R] + om, r_c);
    sstore<NT>(g, w, a.sm[SI_TMM] + om, tmm_c);
    spublish<NT>(g, w, S_c, sTNacc<NT, false>(g, S_a, Rpm_c, szeros<NT>()));          // r R+-  (as TN(r_t, Rpm_c))
    wg_sync();
    wg_inv_one_minus<NT>(g, w, S_c, inv_aux, &bad);                                    // G1 = (I - r R+-)^-1        :244
    const Strip<NT> T01_s = sTNacc<NT, false>(g, S_c, sload<NT>(g, w, a.c_cur[C_T_MM] + om), szeros<NT>());  // (T-- G1)^T  :247
    const Strip<NT> rT_s = sTNacc<NT, false>(g, S_a, Tpp_c, szeros<NT>());             // r T++ (as TN(r_t, Tpp_c))
    sstore<NT>(g, w, a.sm[SI_T01] + om, T01_s);
    sstore<NT>(g, w, a.sm[SI_G1RT] + om, sTNacc<NT, true>(g, S_c, rT_s, szeros<NT>()));   // G1 r T++
    sstore<NT>(g, w, a.sm[SI_G1T] + om, sTNacc<NT, true>(g, S_c, tmm_c, szeros<NT>()));   // G1 t--
    // s1 = r J0+ + j0-  (column 0): every wave its row block, the full tile through LDS
    vxput(g, w, vx, sTNv<NT, false>(g, w, S_a, Jc) + swap01b(vpick<NT>(ja, w)));
    wg_sync();
    const Vec<NT> s1 = vxget<NT>(g, vx);
    store_rows(a.sv[SVI_G1V] + ov, nullptr, sTNv<NT, true>(g, w, S_c, s1));            // G1 (j0- + r J0+)
    wg_sync();  // r, G1 have been read
    spublish<NT>(g, w, S_c, T01_s);
    spublish<NT>(g, w, S_a, rT_s);
    wg_sync();
    const d4 dJm = sTNv<NT, false>(g, w, S_c, s1);                                     // column 0                  :267
    const Strip<NT> Rmp_n = sadd<NT>(sload<NT>(g, w, a.c_cur[C_R_MP] + om), sTNacc<NT, false>(g, S_a, T01_s, szeros<NT>()));  // :288
    const Strip<NT> Tmm_n = sTNacc<NT, true>(g, S_d, T01_s, szeros<NT>());             // U = t--_c                 :290
    sstore<NT>(g, w, a.c_nxt[C_R_MP] + om, Rmp_n);
    sstore<NT>(g, w, a.c_nxt[C_T_MM] + om, Tmm_n);
    const Strip<NT> B2_s = sTNacc<NT, false>(g, S_b, r_c, szeros<NT>());               // R+- r (as TN(Rpm_t, r_c))
    const Strip<NT> Rt_s = sTNacc<NT, false>(g, S_b, tmm_c, szeros<NT>());             // R+- t--
    wg_sync();  // T01, rT, t-- have been read
    spublish<NT>(g, w, S_c, B2_s);
    spublish<NT>(g, w, S_d, sload<NT>(g, w, a.c_cur[C_T_PP] + om));                    // T++[comp]: left factor of the last product but one
    wg_sync();
    wg_inv_one_minus<NT>(g, w, S_c, inv_aux, &bad);                                    // G2 = (I - R+- r)^-1        :295
    const Strip<NT> T21_s = sTNacc<NT, false>(g, S_c, sload<NT>(g, w, a.x[T_PP] + om), szeros<NT>());  //            :297
    sstore<NT>(g, w, a.sm[SI_T21] + om, T21_s);
    sstore<NT>(g, w, a.sm[SI_G2T] + om, sTNacc<NT, true>(g, S_c, Tpp_c, szeros<NT>()));
    sstore<NT>(g, w, a.sm[SI_G2RT] + om, sTNacc<NT, true>(g, S_c, Rt_s, szeros<NT>()));
    // s2 = J0+ + R+- j0-  (column 0)
    vxput(g, w, vx2, vpick<NT>(Jc, w) + swap01b(sTNv<NT, false>(g, w, S_b, ja)));
    wg_sync();
    const Vec<NT> s2 = vxget<NT>(g, vx2);
    store_rows(a.sv[SVI_G2V] + ov, nullptr, sTNv<NT, true>(g, w, S_c, s2));
    wg_sync();  // G2 has been read
    spublish<NT>(g, w, S_c, T21_s);
    spublish<NT>(g, w, S_a, Rt_s);
    wg_sync();
    const d4 dJp = sTNv<NT, false>(g, w, S_c, s2);                                     //                           :315
    const d4 dJm1 = swap01b(dJm), jab = vpick<NT>(ja, w), Jcb = vpick<NT>(Jc, w);
    d4 Jn;
#pragma unroll
    for (int r = 0; r < 4; ++r) Jn[r] = (g.lr == 0) ? (jab[r] + dJp[r]) : (Jcb[r] + dJm1[r]);
    store_rows(a.c_nxt[C_J0P] + ov, a.c_nxt[C_J0M] + ov, Jn);
    sstore<NT>(g, w, a.c_nxt[C_T_PP] + om, sTNacc<NT, true>(g, S_d, T21_s, szeros<NT>()));   // U = T++_c          :338
    sstore<NT>(g, w, a.c_nxt[C_R_PM] + om,
               sadd<NT>(sload<NT>(g, w, a.x[R_PM] + om), sTNacc<NT, false>(g, S_a, T21_s, szeros<NT>())));  //      :340
  }
  if (bad && lane == 0) atomicMax(a.info, bad);
}

#define MOMR_WG_WPE(NT_) (NT_ == 2 ? 2 : (NT_ == 3 ? MOMR_WG3_WPE : 1))
#define MOMR_WG_ATTR(NT_) __launch_bounds__(64 * NT_) __attribute__((amdgpu_waves_per_eu(MOMR_WG_WPE(NT_), MOMR_WG_WPE(NT_))))
template <bool FUSE, int MODE, bool NG>
__global__ void MOMR_WG_ATTR(2) k_dbl_pair_wg2(KArgs a) { dbl_pair_wg<2, FUSE, MODE, NG>(a); }
template <bool FUSE, int MODE, bool NG>
__global__ void MOMR_WG_ATTR(3) k_dbl_pair_wg3(KArgs a) { dbl_pair_wg<3, FUSE, MODE, NG>(a); }
template <bool FUSE, int MODE, bool NG>
__global__ void MOMR_WG_ATTR(4) k_dbl_pair_wg4(KArgs a) { dbl_pair_wg<4, FUSE, MODE, NG>(a); }
template <bool SURF, bool DERIVE>
__global__ void MOMR_WG_ATTR(3) k_int_pair_wg3(KArgs a) { int_pair_wg<3, SURF, DERIVE>(a); }
template <bool SURF, bool DERIVE>
__global__ void MOMR_WG_ATTR(4) k_int_pair_wg4(KArgs a) { int_pair_wg<4, SURF, DERIVE>(a); }
template <bool SURF, bool DERIVE>
__global__ void MOMR_WG_ATTR(2) k_int_pair_wg2(KArgs a) { int_pair_wg<2, SURF, DERIVE>(a); }
__global__ void __launch_bounds__(128) k_dbl_point_wg2(KArgs a) { dbl_point_wg<2>(a); }
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 1))) k_dbl_point_wg3(KArgs a) { dbl_point_wg<3>(a); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_dbl_point_wg4(KArgs a) { dbl_point_wg<4>(a); }
__global__ void __launch_bounds__(128) k_int_point_wg2(KArgs a) { int_point_wg<2>(a); }
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 1))) k_int_point_wg3(KArgs a) { int_point_wg<3>(a); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_int_point_wg4(KArgs a) { int_point_wg<4>(a); }
