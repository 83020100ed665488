// mom_rrs_wg.hpp -- the RRS pair kernels above N = 32 as ONE WORKGROUP PER PAIR (included by mom_rrs.hip inside its
// namespace, big-tile object only: 3 x 3 tiles for 32 < N <= 48, 4 x 4 for 48 < N <= 64).
//   doubling_helper!(::RRS)            CoreKernel/doubling_inelastic.jl:61-89 (sources), :98-125 (operators)
//   interaction_helper!(::RRS, 11)     CoreKernel/interaction_inelastic.jl:249-335
//
// Why.  The wave-per-pair bodies (dbl_pair_body / int_pair_body) keep every operator of a pair as NT x NT register tiles of one
// wave: at NT = 3 / 4 the dozen live operators of a pair are 1 700 ... 3 000 registers per lane against 512, and the kernels
// ran from scratch memory (profiles/r05_rrs_nt.txt: 0.02-0.04 of the HBM roofline).  Here a pair belongs to a workgroup of NT
// waves and every operator is split into COLUMN STRIPS: wave w owns tile column w (NT tiles = 4 NT registers per lane).  With
// TN(U, V) = U^T V of mom_tile.hpp, column strip w of a product needs the whole left factor U and only strip w of V:
//   * V strips stay in the registers of their wave (they are loaded from global memory as strips, or are the wave's own
//     result of an earlier product);
//   * U is read fragment by fragment from an LDS copy of the matrix, laid out in 16 x 16 tiles of pitch 17 so that BOTH
//     orientations are (nearly) conflict-free -- the transposes the one-wave bodies make through LDS (a_c, b_c, bn_c) become a
//     different index expression of the same copy, and W = b^T + Y, V = bn^T + Y are formed fragment-wise from two copies;
//   * computed left factors (X, Y, bn) are published strip-wise into LDS between two barriers; left factors that come from
//     global memory (r[n0]^T, (G t)[n0] ...) are loaded as strips -- one per wave, coalesced -- and published the same way.
// Four LDS matrices (a, b / bn, X / Y, the rotating global one) = 78 KB (NT = 3: two workgroups per CU) / 139 KB (NT = 4).
// The source vectors go through the vector ALU as in the one-wave bodies: a wave multiplies its strip with the full vector
// (row layout) and owns 16 entries of the result; full vectors are exchanged through four small LDS buffers.
// The order of the products and of every accumulation is that of dbl_pair_body / int_pair_body: the results are bitwise those
// of the one-wave kernels (tests/test_gpu_rrs.py compares both, MOM_RRS_WG=0 selects the old ones).
#pragma once

template <int NT>
struct Strip {
  d4 t[NT];  // t[a]: tile (a, w) of the matrix, w = the wave's column
};

template <int NT>
constexpr int wg_mat_doubles() { return NT * NT * kTileDoubles; }
template <int NT>
constexpr size_t wg_lds_bytes(int nmat) { return ((size_t)nmat * wg_mat_doubles<NT>() + 4 * 16 * NT) * 8; }

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
// register s of tile (tk, ti) of the LDS matrix M (TR = false) or of its transpose (TR = true) in the C / A-operand layout
template <int NT, bool TR>
__device__ __forceinline__ double ufrag(const Geo &g, const double *M, int tk, int ti, int s) {
  return TR ? M[(ti * NT + tk) * kTileDoubles + g.lr * kTileLd + g.lq + 4 * s]
            : M[(tk * NT + ti) * kTileDoubles + (g.lq + 4 * s) * kTileLd + g.lr];
}
// acc + (column strip of) U^T V;  U = M (TR = false) or M^T (TR = true) from LDS.  k-steps in the zero padding are skipped.
template <int NT, bool TR>
__device__ __forceinline__ Strip<NT> sTNacc(const Geo &g, const double *M, const Strip<NT> &V, Strip<NT> acc) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (16 * tk + 4 * s < g.N) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
          acc.t[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(ufrag<NT, TR>(g, M, tk, ti, s), V.t[tk][s], acc.t[ti], 0, 0, 0);
      }
  return acc;
}
// ... with U = M1^T + M2 formed fragment-wise (W = b^T + Y, V = bn^T + Y of the doubling step)
template <int NT>
__device__ __forceinline__ Strip<NT> sTNacc_sum(const Geo &g, const double *M1, const double *M2, const Strip<NT> &V, Strip<NT> acc) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (16 * tk + 4 * s < g.N) {
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
template <int NT, bool FUSE, int MODE>
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
  double *vb0 = vb, *vb1 = vb + 16 * NT, *vb2 = vb + 32 * NT, *vb3 = vb + 48 * NT;
  const int n = a.nS, cw = 16 * w + g.lr;  // the lane's column of the strip / entry of a column-layout vector
  const size_t NN = (size_t)a.P * a.P, VS = a.P;
  const size_t span = (size_t)(a.n1_hi - a.n1_lo), npairs = span * a.nR;
  auto sgn_i = [&](int i, int, double v) { return scomp(i, n, a.strict_idx) > 2 ? -v : v; };
  auto sgn_ij = [&](int i, int j, double v) { return dsgn(scomp(i, n, a.strict_idx), scomp(j, n, a.strict_idx)) * v; };
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
    const size_t m1 = NN * n1, m0 = NN * n0, v0 = VS * n0;
    Strip<NT> a_s, b_s;
    double Jp, Jm;
    if (!FUSE) {
      a_s = sload<NT>(g, w, a.ie_a[R_MP] + o4);
      b_s = sload<NT>(g, w, a.ie_a[T_PP] + o4);
      Jp = a.ie_a[J0P][o3 + cw];
      Jm = a.ie_a[J0M][o3 + cw];
    }
    const Strip<NT> r0_s = sload<NT>(g, w, a.sm[SM_RT] + m0), r1_s = sload<NT>(g, w, a.a_cur[R_MP] + m1);
    const Strip<NT> gt0_s = sload<NT>(g, w, a.sm[SM_GT] + m0), ttgp1_s = sload<NT>(g, w, a.sm[SM_TTGP] + m1);
    const double e1 = a.expk_cur[n1];
    const double *jp0 = STRICT ? a.jpseq + v0 + VS * a.S * dn : a.a_cur[J0P] + v0;
    const Vec<NT> j1m0 = loadR<NT>(g, a.sv[SV_J1M] + v0), jp0R = loadR<NT>(g, jp0);
    const Vec<NT> tm1 = loadR<NT>(g, a.sv[SV_TMP1] + v0), tm2 = loadR<NT>(g, a.sv[SV_TMP2] + v0);
    if (FUSE) ie_elem_strip<NT>(g, w, a, n1, dn, n0, a_s, b_s, Jp, Jm);
    const double J1p = Jp * e1, J1m = Jm * e1;  // ieJ1+, ieJ1-   :52-56
    wg_sync();  // the previous pair has finished with the LDS matrices
    spublish<NT>(g, w, S_a, a_s);
    spublish<NT>(g, w, S_b, b_s);
    spublish<NT>(g, w, S_g, r0_s);
    vput(g, w, vb0, J1m);
    wg_sync();
    // X = ier r0 + r1 ier
    Strip<NT> X_s = sTNacc<NT, false>(g, S_g, a_s, szeros<NT>());  // U = r0_c
    X_s = sTNacc<NT, true>(g, S_a, r1_s, X_s);                     // U = a_c = (a_t)^T
    wg_sync();  // r0 has been read by everybody
    spublish<NT>(g, w, S_x, X_s);
    spublish<NT>(g, w, S_g, gt0_s);
    // the late operands: requested here, consumed after the next two products
    const Strip<NT> gr0_s = sload<NT>(g, w, a.sm[SM_GR] + m0), t0_s = sload<NT>(g, w, a.sm[SM_TT] + m0);
    const Strip<NT> ttgpr1_s = sload<NT>(g, w, a.sm[SM_TTGPR] + m1);
    // ---- sources                                                                                               :61-89
    {
      const double a_j1m = smv<NT>(g, a_s, j1m0);  // ier j1-[n0]
      const double a_jp = smv<NT>(g, a_s, jp0R);   // ier j0+[n0]
      const double X1 = smv<NT>(g, X_s, tm1), X2 = smv<NT>(g, X_s, tm2);
      const double b1 = smv<NT>(g, b_s, tm1);      // iet++ tmp1
      const double b2 = STRICT ? smv<NT>(g, sload<NT>(g, w, a.ie_a[T_MM] + o4), tm2)  // D5: iet-- as the array holds it
                               : smv<NT>(g, b_s, tm2);
      const double uu = (Jp + smv<NT>(g, r1_s, vget<NT>(g, vb0))) + (a_j1m + X1);
      vput(g, w, vb1, uu);
      wg_sync();  // (also: X and (G t)[n0] are in LDS)
      const double Jpn = (J1p + smv<NT>(g, ttgp1_s, vget<NT>(g, vb1))) + b1;  // new ieJ0+
      vput(g, w, vb2, Jpn);
      wg_sync();
      const double rv2 = smv<NT>(g, r1_s, vget<NT>(g, vb2));                  // r1 ieJ0+(new)
      const double u2 = (J1m + rv2) + (a_jp + X2);
      vput(g, w, vb3, u2);
      wg_sync();
      double Jmn = (Jm + smv<NT>(g, ttgp1_s, vget<NT>(g, vb3))) + b2;         // new ieJ0-
      if (fuseD && n > 1 && scomp(cw, n, a.strict_idx) > 2) Jmn = -Jmn;
      if (g.lq == 0) {
        a.ie_a[J0P][o3 + cw] = Jpn;
        a.ie_a[J0M][o3 + cw] = Jmn;
      }
    }
    // ---- operators                                                                                              :98-125
    const Strip<NT> Y_s = sTNacc<NT, false>(g, S_x, gt0_s, szeros<NT>());  // Y_c = X G t[n0]          (U = X_t)
    wg_sync();                                                              // X has been read
    spublish<NT>(g, w, S_x, Y_s);
    wg_sync();
    Strip<NT> bn_s = sTNacc_sum<NT>(g, S_b, S_x, ttgp1_s, szeros<NT>());    // tG (iet + Y)            (U = W_c = b_c + Y_c)
    bn_s = sTNacc<NT, false>(g, S_g, b_s, bn_s);                            // + iet G t[n0]           (U = (G t)[n0]_c)
    wg_sync();                                                              // b and (G t)[n0] have been read
    spublish<NT>(g, w, S_b, bn_s);
    spublish<NT>(g, w, S_g, gr0_s);
    wg_sync();
    Strip<NT> Q_s = sTNacc<NT, false>(g, S_g, bn_s, szeros<NT>());          // iet(new) G r[n0]        (U = (G r)[n0]_c)
    Q_s = sTNacc<NT, true>(g, S_a, ttgp1_s, Q_s);                           // + tG ier                (U = a_c)
    wg_sync();                                                              // (G r)[n0] has been read
    spublish<NT>(g, w, S_g, t0_s);
    wg_sync();
    Strip<NT> an_s = sTNacc_sum<NT>(g, S_b, S_x, ttgpr1_s, szeros<NT>());   // (iet(new) + Y) ...      (U = V_c = bn_c + Y_c)
    an_s = sTNacc<NT, false>(g, S_g, Q_s, an_s);                            // + t[n0]-side product    (U = t0_c)
    an_s = sadd<NT>(a_s, an_s);
    if (fuseD) {  // apply_D_matrix_IE!, corrected indexing (D2)
      if (n > 1) smap<NT>(g, w, an_s, sgn_i);
      Strip<NT> apm = an_s, bmm = bn_s;
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
    sstore<NT>(g, w, a.ie_a[T_PP] + o4, bn_s);
  }
}

#define MOMR_WG_ATTR(NT_) __launch_bounds__(64 * NT_) __attribute__((amdgpu_waves_per_eu(1, (NT_ == 3 ? 2 : 1))))
template <bool FUSE, int MODE>
__global__ void MOMR_WG_ATTR(3) k_dbl_pair_wg3(KArgs a) { dbl_pair_wg<3, FUSE, MODE>(a); }
template <bool FUSE, int MODE>
__global__ void MOMR_WG_ATTR(4) k_dbl_pair_wg4(KArgs a) { dbl_pair_wg<4, FUSE, MODE>(a); }
