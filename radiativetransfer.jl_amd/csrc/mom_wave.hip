// mom_wave.hip -- operators of edge 4 < N <= 16: ONE SPECTRAL POINT PER WAVEFRONT, all operators in registers, every
// product on the FP64 matrix cores without touching LDS.
//
// A 16 x 16 tile in the C/D layout of v_mfma_f64_16x16x4_f64 (lane l, register r: row (l >> 4) + 4 r, column l & 15) is
// four doubles per lane.  Feeding two such tiles U, V to the four k-steps of one 16 x 16 x 16 product -- register s of U
// as the A operand, register s of V as the B operand -- gives
//        TN(U, V) = U^T V        in the same layout,
// because the A operand is read as A[row = l & 15][k = l >> 4]: a tile in C-layout IS the A operand of its transpose
// (mom_strip.hpp uses the B-operand half of this observation).  So a wave that keeps, for every operator X it needs as a
// LEFT factor, the tile of X^T ("t-form") next to or instead of the tile of X ("c-form"), runs the whole adding
// algorithm as a sequence of 4-MFMA products on registers: X Y = TN(X_t, Y_c), (X Y)^T = TN(Y_c, X_t).  The elemental
// layer is evaluated directly in both forms (element (i,j) and (j,i) share their exponentials), the doubling recursion
// and the interaction are closed under TN with the forms listed at the functions below, source vectors travel as
// columns 0 (J+) and 1 (J-) of a tile.  No LDS, no barrier: LDS only serves the rare pivoted inverse (series too long)
// and the final gather of the view rows.  A lane-per-point layout (mom_small.hip) stops at N = 4, the workgroup-per-
// point kernels (mom_kernels.hpp) use 1/16 ... 1/4 of their tiles and a whole CU per 2 units at these sizes.
//
// One wave walks all Fourier moments and layers of its spectral point (like momsm::k_sweep), two waves per SIMD.
// Scope: ScatteringInterface_11 on every layer after the first (the host falls back to the general kernels otherwise),
// LambertianSurfaceScalar, Float64.  Reference semantics and file:line as in mom_kernels.hpp / mom_small.hip.
#include <hip/hip_runtime.h>

#include "mom_host.hpp"

#ifndef MOMW_OCC
#define MOMW_OCC 2  // waves per SIMD the kernel is built for
#endif

namespace momw {

typedef double d4 __attribute__((ext_vector_type(4)));

// KS = ceil(N / 4) k-steps: rows >= 4 KS of every tile are zero padding
template <int KS>
__device__ __forceinline__ d4 TN(d4 U, d4 V) {  // U^T V
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[s], V[s], acc, 0, 0, 0);
  return acc;
}
template <int KS>
__device__ __forceinline__ d4 TNacc(d4 U, d4 V, d4 acc) {  // acc + U^T V
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[s], V[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ d4 shfl_xor1(d4 v) {  // exchange columns 0 <-> 1 (2 <-> 3, ...)
  d4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = __shfl_xor(v[r], 1);
  return o;
}

// series length for (I - B)^-1 from beta^2 = ||B||_F^2: the rule of mom_kernels.hpp (kNeumannThr2: tail <= 2^-56, the
// first-order term always kept); 1000 = use the pivoted inverse
__device__ const double kThr2[32] = {
    0.0, 1.38777877561156685e-17, 5.77492213356056750e-12, 3.72517661162420568e-09,
    1.80656771560518035e-07, 2.40186660760962690e-06, 1.52417448931310540e-05, 6.09157135028591602e-05,
    1.78874927371965362e-04, 4.23309807394842467e-04, 8.56292603484697687e-04, 1.53988783074545245e-03,
    2.52966413875025916e-03, 3.87056942792606993e-03, 5.59509450064154569e-03, 7.72318484591632843e-03,
    1.02632763666295982e-02, 1.32139270473634555e-02, 1.65656642989196294e-02, 2.03028052759899880e-02,
    2.44051136922726897e-02, 2.88492300492497432e-02, 3.36098586497813809e-02, 3.86607216385354419e-02,
    4.39753040322414940e-02, 4.95274191527264318e-02, 5.52916244610413068e-02, 6.12435157546498479e-02,
    6.73599244364155580e-02, 7.36190389296255826e-02, 8.00004677634075928e-02, 8.64852586225294262e-02};
__device__ __forceinline__ int series_terms(double beta2) {
  if (!(beta2 <= kThr2[31])) return 1000;
  int p = 1;
#pragma unroll
  for (int k = 0; k < 31; ++k) p += (beta2 > kThr2[k]) ? 1 : 0;
  return p;
}

struct WArgs {
  int N, S, M, K, Nz, nVza, nS, imu0, inv_mode, pad;
  double mu0, albedo;
  double I0[4], D[4];
  const double *mu, *wt, *sg;           // [N]
  const double *Zpp, *Zmp;              // [N,N,K,M]
  const int *nd;                        // [Nz]
  const int *node;                      // [nVza]
  const double *cos_mphi, *sin_mphi;    // [nVza,M]
  const double *tau, *varpi, *zw, *tau_sum;  // [S,Nz], [S,Nz], [K,S,Nz], [S,Nz+1]
  double *R, *T, *hdr, *bhr_uw, *bhr_dw;
  int *info;
};

// per-lane constants of the tile layout; the row quantities (row = lq + 4 r) are read from the block's LDS table
// tab = mu[16] | wt[16] | sg[16] (padding rows: mu = 1, wt = 0, sg = 1) when needed instead of living in 24 VGPRs
struct Lay {
  int lr, lq, N, nS;
  bool cok;                // column < N
  double muc, wc, sgc;     // column quantities (column = lr)
  const double *tab;
  __device__ __forceinline__ bool rok(int r) const { return lq + 4 * r < N; }
  __device__ __forceinline__ double mur(int r) const { return tab[lq + 4 * r]; }
  __device__ __forceinline__ double wr(int r) const { return tab[16 + lq + 4 * r]; }
  __device__ __forceinline__ double sgr(int r) const { return tab[32 + lq + 4 * r]; }
};
__device__ __forceinline__ d4 ident(const Lay &L) {
  d4 I;
#pragma unroll
  for (int r = 0; r < 4; ++r) I[r] = (L.lq + 4 * r == L.lr && L.cok) ? 1.0 : 0.0;
  return I;
}
__device__ __forceinline__ d4 scaleD(const Lay &L, d4 X) {  // D X D, D = Diagonal(sg)
  d4 Y;
#pragma unroll
  for (int r = 0; r < 4; ++r) Y[r] = (L.sgr(r) * L.sgc) * X[r];
  return Y;
}

// (I - B)^-1 for the tile pair (B in c-form Bc, B^T in c-form Bt): Horner G <- I + B G = I + TN(Bt, G); beyond 32
// terms (or MOM_OPT_INVERSE = 1) Gauss-Jordan with implicit partial pivoting in a row-per-lane layout through the
// wave's LDS slice (the register-resident scheme of wg_inverse_reg, mom_device.hpp, for a single wave).
__device__ __noinline__ d4 inverse_gj(d4 Bc, d4 Ic, int N, double *lds, int *ipiv, int *bad_out) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  int bad = 0;
  // A = I - B, row-major with pitch 17 in the wave's LDS slice
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[(lq + 4 * r) * 17 + lr] = Ic[r] - Bc[r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = (lane < N && c < N) ? lds[lane * 17 + c] : ((lane < 16 && lane == c) ? 1.0 : 0.0);
  bool used = false;
  int myk = lane;  // rows >= N keep their identity row
#pragma unroll
  for (int k = 0; k < 16; ++k) {
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
      for (int c = 0; c < 16; ++c) {
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
    for (int c = 0; c < 16; ++c)
      if (c < N) lds[myk * 17 + ipiv[c]] = v[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  d4 G;
#pragma unroll
  for (int r = 0; r < 4; ++r) G[r] = (lq + 4 * r < N && lr < N) ? lds[(lq + 4 * r) * 17 + lr] : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (bad) *bad_out = bad;
  return G;
}

template <int KS>
__device__ __forceinline__ d4 inv_one_minus(const Lay &L, d4 Bc, d4 Bt, int inv_mode, double *lds, int *ipiv, int &bad) {
  double ss = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) ss += Bc[r] * Bc[r];
  const double beta2 = wave_sum(ss);
  const int p = (inv_mode == 1) ? 1000 : series_terms(beta2);
  const d4 Ic = ident(L);
  if (p <= 32) {
    d4 G = Ic;
    for (int k = 1; k < p; ++k) G = TNacc<KS>(Bt, G, Ic);
    return G;
  }
  int b = 0;
  const d4 G = inverse_gj(Bc, Ic, L.N, lds, ipiv, &b);
  if (b && !bad) bad = b;
  return G;
}

// ScatteringInterface_11 (interaction.jl:69-117) on tiles.  Added layer: r-+ (rc, rt), t++ (tc, tt), sources jv (column 0
// j0+, column 1 j0-); r+- = D r-+ D and t-- = D t++ D are formed where they are used (SURF: the surface layer has
// r+- = 0, t = I).  Composite state kept in exactly the forms the next interaction consumes: R-+ (c), R+- (c and t),
// T++ (c), T-- (t), Jv (column 0 J0+, column 1 J0-).
struct Comp {
  d4 Rmp_c, Rpm_c, Rpm_t, Tpp_c, Tmm_t, Jv;
};
template <int KS, bool SURF>
__device__ __forceinline__ void interact11(const Lay &L, Comp &C, d4 rc, d4 rt, d4 tc, d4 tt, d4 jv, int inv_mode,
                                           double *lds, int *ipiv, int &bad) {
  // --- T01 = T-- (I - r-+ R+-)^-1                                                   (:81-87)
  const d4 B1 = TN<KS>(rt, C.Rpm_c), B1t = TN<KS>(C.Rpm_c, rt);
  const d4 G1 = inv_one_minus<KS>(L, B1, B1t, inv_mode, lds, ipiv, bad);
  const d4 T01t = TN<KS>(G1, C.Tmm_t);  // (T-- G1)^T = G1^T T--^T
  // J0- = J0- + T01 (r-+ J0+ + j0-)                                                  (:90)  [old J0+]
  const d4 V1 = TN<KS>(rt, C.Jv);       // column 0: r-+ J0+
  const d4 jsw = shfl_xor1(jv);         // column 0: j0-
  d4 X1;
#pragma unroll
  for (int r = 0; r < 4; ++r) X1[r] = (L.lr == 0) ? V1[r] + jsw[r] : 0.0;
  const d4 TX1 = TN<KS>(T01t, X1);      // column 0: T01 (...)
  const d4 TX1s = shfl_xor1(TX1);
  // R-+ = R-+ + T01 r-+ T++                                                          (:93)
  const d4 rT = TN<KS>(rt, C.Tpp_c);
  C.Rmp_c = TNacc<KS>(T01t, rT, C.Rmp_c);
  // T-- = T01 t--   (kept transposed: t--^T T01^T)                                   (:96)
  C.Tmm_t = SURF ? T01t : TN<KS>(scaleD(L, tc), T01t);
  // --- T21 = t++ (I - R+- r-+)^-1                                                   (:104-107)  [old R+-]
  const d4 B2 = TN<KS>(C.Rpm_t, rc), B2t = TN<KS>(rc, C.Rpm_t);
  const d4 G2 = inv_one_minus<KS>(L, B2, B2t, inv_mode, lds, ipiv, bad);
  const d4 T21t = SURF ? TN<KS>(G2, ident(L)) : TN<KS>(G2, tt);
  // J0+ = j0+ + T21 (J0+ + R+- j0-)                                                  (:110)
  const d4 V2 = TN<KS>(C.Rpm_t, jv);    // column 1: R+- j0-
  const d4 V2s = shfl_xor1(V2);
  d4 X2;
#pragma unroll
  for (int r = 0; r < 4; ++r) X2[r] = (L.lr == 0) ? C.Jv[r] + V2s[r] : 0.0;
  const d4 TX2 = TN<KS>(T21t, X2);      // column 0: T21 (...)
#pragma unroll
  for (int r = 0; r < 4; ++r)
    C.Jv[r] = (L.lr == 0) ? jv[r] + TX2[r] : ((L.lr == 1) ? C.Jv[r] + TX1s[r] : 0.0);
  // T++ = T21 T++                                                                    (:113)
  C.Tpp_c = TN<KS>(T21t, C.Tpp_c);
  // R+- = r+- + T21 R+- t--   (both forms)                                           (:116)
  d4 Y, zero = {0.0, 0.0, 0.0, 0.0};
  if (SURF) {  // t-- = I: Y = R+-
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[r] = C.Rpm_c[r];
  } else {
    Y = TN<KS>(C.Rpm_t, scaleD(L, tc));
  }
  const d4 Rpm_c_new = TNacc<KS>(T21t, Y, SURF ? zero : scaleD(L, rc));
  const d4 Rpm_t_new = TNacc<KS>(Y, T21t, SURF ? zero : scaleD(L, rt));
  C.Rpm_c = Rpm_c_new; C.Rpm_t = Rpm_t_new;
}

template <int KS>
__global__ void __launch_bounds__(256, MOMW_OCC) k_wsweep(WArgs a) {
  __shared__ double s_lds[4][16 * 17 + 48];
  __shared__ int s_piv[4][16];
  __shared__ double s_tab[48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;  // spectral point of this wave
  const int N = a.N, nS = a.nS, S = a.S, K = a.K;
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    s_tab[i] = i < N ? a.mu[i] : 1.0;
    s_tab[16 + i] = i < N ? a.wt[i] : 0.0;
    s_tab[32 + i] = i < N ? a.sg[i] : 1.0;
  }
  __syncthreads();
  if (n >= S) return;
  double *lds = s_lds[wave];
  int *ipiv = s_piv[wave];
  Lay L;
  L.lr = lane & 15; L.lq = lane >> 4; L.N = N; L.nS = nS; L.tab = s_tab;
  L.cok = L.lr < N;
  L.muc = s_tab[L.lr]; L.wc = s_tab[16 + L.lr]; L.sgc = s_tab[32 + L.lr];
  const int i_start = nS * (a.imu0 - 1), i_end = nS * a.imu0;
  const double mus = a.mu[i_start];
  int bad = 0;
  // accumulators of the outputs: lane x < nVza * nS handles (view v = x / nS, component k = x % nS)
  double accR = 0.0, accT = 0.0, accH = 0.0;
  const int xv = lane / nS, xk = lane - xv * nS;
  const bool xok = lane < a.nVza * nS;

  for (int m = 0; m < a.M; ++m) {
    const double wdiv = (m == 0) ? 2.0 : 4.0, wct02 = (m == 0) ? 0.5 : 0.25;
    const double *Zp_m = a.Zpp + (size_t)N * N * K * m, *Zm_m = a.Zmp + (size_t)N * N * K * m;
    Comp C;
    for (int z = 0; z < a.Nz; ++z) {
      const int nd = a.nd[z];
      const size_t o = n + (size_t)S * z;
      const double tau = a.tau[o], varpi = a.varpi[o], tau_sum = a.tau_sum[o];
      const double dtau = ldexp(tau, -nd);
      double expk = exp(-dtau / a.mu0);
      // ------------------------------------------------ elemental! in both forms (elemental.jl:164-253)
      d4 rc, rt, tc, tt, jv;
      {
        const double wjc = L.wc / wdiv;              // w'_j of the column stream
        const double ec = exp(-dtau / L.muc);        // exp(-dtau / mu_col)
        const double att = exp(-tau_sum / mus), es = exp(-dtau / mus);
        // one row (register) per iteration, NOT unrolled: the live set of one iteration is what the register budget of
        // two waves per SIMD affords next to the composite tiles
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          const int i = L.lq + 4 * r, j = L.lr;
          const bool rok = i < N, ok = rok && L.cok;
          double zpij = 0.0, zmij = 0.0, zpji = 0.0, zmji = 0.0;
          if (ok)
            for (int k = 0; k < K; ++k) {
              const double w = a.zw[k + (size_t)K * o];
              const size_t b = (size_t)N * N * k;
              zpij += w * Zp_m[b + i + N * j]; zmij += w * Zm_m[b + i + N * j];
              zpji += w * Zp_m[b + j + N * i]; zmji += w * Zm_m[b + j + N * i];
            }
          const double mui = L.mur(r), muj = L.muc, wir = L.wr(r) / wdiv;
          const double er = exp(-dtau / mui);
          const double E = 1 - exp(-dtau * ((1 / mui) + (1 / muj)));  // symmetric in (i, j)
          double rij, tij, rji, tji;
          // element (i, j): column stream j
          if (wjc > 1.e-8) {
            rij = varpi * zmij * (muj / (mui + muj)) * wjc * E;
            if (mui == muj) tij = (i == j) ? er * (1 + varpi * zpij * (dtau / mui) * wir) : 0.0;
            else tij = varpi * zpij * (muj / (mui - muj)) * wjc * (er - ec);
          } else {
            rij = 0.0;
            tij = (i == j) ? er : 0.0;
          }
          // element (j, i): column stream i (the reference's expression with i and j exchanged)
          if (wir > 1.e-8) {
            rji = varpi * zmji * (mui / (muj + mui)) * wir * E;
            if (muj == mui) tji = (i == j) ? ec * (1 + varpi * zpji * (dtau / muj) * wjc) : 0.0;
            else tji = varpi * zpji * (mui / (muj - mui)) * wir * (ec - er);
          } else {
            rji = 0.0;
            tji = (i == j) ? ec : 0.0;
          }
          if (nd >= 1) { rij *= L.sgr(r); rji *= L.sgc; }  // apply_D_elemental!: rows of r-+ (elemental.jl:265-269)
          rc[r] = ok ? rij : 0.0; tc[r] = ok ? tij : 0.0;
          rt[r] = ok ? rji : 0.0; tt[r] = ok ? tji : 0.0;
          // source rows: Z I0 over the sun's Stokes block (lanes of columns 0 and 1)             (elemental.jl:224-251)
          double jx = 0.0;
          if (rok && L.lr < 2) {
            const double *Zs = (L.lr == 0) ? Zp_m : Zm_m;
            double zI = 0.0;
            for (int ks = 0; ks < nS; ++ks)
              for (int k = 0; k < K; ++k)
                zI += a.zw[k + (size_t)K * o] * Zs[(size_t)N * N * k + i + (size_t)N * (i_start + ks)] * a.I0[ks];
            if (L.lr == 0) {
              if (i >= i_start && i < i_end) jx = wct02 * varpi * zI * (dtau / mui) * er;
              else jx = wct02 * varpi * zI * (mus / (mui - mus)) * (er - es);
              jx *= att;
            } else {
              jx = wct02 * varpi * zI * (mus / (mui + mus)) * (1 - exp(-dtau * ((1 / mui) + (1 / mus))));
              jx *= att;
              if (nd >= 1) jx = a.D[i % nS] * jx;
            }
          }
          jv[r] = jx;
        }
      }
      // ------------------------------------------------ doubling_helper! (doubling.jl:43-68)
      // forms: rc, rt, tc, tt; per step B = r r, B^T, G = (I - B)^-1, A^T = G^T t^T, W = r t, then
      // r <- r + A W (c and t), t <- A t (c and t); sources through the vector tile
      for (int it = 0; it < nd; ++it) {
        const d4 B = TN<KS>(rt, rc), Bt = TN<KS>(rc, rt);
        const d4 G = inv_one_minus<KS>(L, B, Bt, a.inv_mode, lds, ipiv, bad);
        const d4 At = TN<KS>(G, tt);
        const d4 U = TN<KS>(rt, jv);          // columns: r j0+ | r j0-
        const d4 Us = shfl_xor1(U);
        d4 Wv;
#pragma unroll
        for (int r = 0; r < 4; ++r)       // column 0: w2 = j0+ + r j1-  ; column 1: w1 = j1- + r j0+   (:51-60)
          Wv[r] = (L.lr == 0) ? jv[r] + expk * Us[r] : ((L.lr == 1) ? jv[r] * expk + Us[r] : 0.0);
        const d4 AW = TN<KS>(At, Wv);
#pragma unroll
        for (int r = 0; r < 4; ++r)       // j0+ = j1+ + A w2 (:60) ; j0- = j0- + A w1 (:57)
          jv[r] = (L.lr == 0) ? jv[r] * expk + AW[r] : ((L.lr == 1) ? jv[r] + AW[r] : 0.0);
        expk = expk * expk;               // :61
        const d4 W = TN<KS>(rt, tc);          // r t (old t)
        const d4 rc_n = TNacc<KS>(At, W, rc), rt_n = TNacc<KS>(W, At, rt);   // r + A (r t)      (:64)
        const d4 tc_n = TN<KS>(At, tc), tt_n = TN<KS>(tc, At);               // A t             (:67)
        rc = rc_n; rt = rt_n; tc = tc_n; tt = tt_n;
      }
      if (nd >= 1) {  // apply_D! / apply_D_SFI! (doubling.jl:93-118): rows of r-+ and j0- scaled by sg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          rc[r] *= L.sgr(r);
          rt[r] *= L.sgc;
          if (L.lr == 1) jv[r] *= L.sgr(r);
        }
      }
      // ------------------------------------------------ composite <- added (rt_kernel.jl:227-230) or interaction!
      if (z == 0) {
        C.Rmp_c = rc; C.Rpm_c = scaleD(L, rc); C.Rpm_t = scaleD(L, rt); C.Tpp_c = tc; C.Tmm_t = scaleD(L, tt); C.Jv = jv;
      } else {
        interact11<KS, false>(L, C, rc, rt, tc, tt, jv, a.inv_mode, lds, ipiv, bad);
      }
    }
    // ---------------------------------------------------- Lambertian surface (m = 0) + closing interaction (Q6)
    d4 hdrJ = {0.0, 0.0, 0.0, 0.0};
    if (m == 0) {
      const double rho = 2 * a.albedo;
      const double att = exp(-a.tau_sum[n + (size_t)S * a.Nz] / a.mu0);
      d4 rs_c, rs_t, jv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = L.lq + 4 * r, j = L.lr;
        const bool ok = L.rok(r) && L.cok, ii = (i % nS == 0) && (j % nS == 0);
        rs_c[r] = (ok && ii) ? rho * (L.muc * L.wc) : 0.0;         // r-+ = R_surf Diagonal(mu w)  (:41-43,:58)
        rs_t[r] = (ok && ii) ? rho * (L.mur(r) * L.wr(r)) : 0.0;
        const bool in_sun = (i >= i_start) && (i < i_end);
        const double jp = (in_sun ? a.I0[i - i_start] : 0.0) * att;                     // :55
        const double jm = (i % nS == 0) ? (a.mu0 * (rho * a.I0[0])) * att : 0.0;       // :56
        jv[r] = !L.rok(r) ? 0.0 : (L.lr == 0 ? jp : (L.lr == 1 ? jm : 0.0));
      }
      interact11<KS, true>(L, C, rs_c, rs_t, rs_c, rs_c, jv, a.inv_mode, lds, ipiv, bad);  // (t operands unused)
      // interaction_hdrf! (interaction_hdrf.jl:9-45): hdr_J0- = r-+_surf J0+ + j0-_surf  -> column 0
      const d4 rJ = TN<KS>(rs_t, C.Jv);
      const d4 jsw = shfl_xor1(jv);
#pragma unroll
      for (int r = 0; r < 4; ++r) hdrJ[r] = (L.lr == 0) ? rJ[r] + jsw[r] : 0.0;
      // BHR flux sums over the streams of each Stokes component (column-0 lanes hold hdr_J0- and J0+)
      for (int k = 0; k < nS; ++k) {
        double up = 0.0, dw = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = L.lq + 4 * r;
          if (L.lr == 0 && L.rok(r) && (i % nS == k)) {
            up += hdrJ[r] * L.wr(r) * L.mur(r);
            dw += C.Jv[r] * L.wr(r) * L.mur(r);
          }
        }
        up = wave_sum(up);
        dw = wave_sum(dw);
        // + j0+_surf[i_start] mu[i_start]: the direct beam (interaction_hdrf.jl:30)
        const double direct = a.I0[0] * att * mus;
        if (lane == 0) {
          a.bhr_uw[k + (size_t)nS * n] = up;
          a.bhr_dw[k + (size_t)nS * n] = dw + direct;
        }
      }
    }
    // ---------------------------------------------------- postprocessing_vza! (+ hdrf) through the wave's LDS slice
    // layout: [0..15] J0+, [16..31] J0-, [32..47] hdr_J0-
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = L.lq + 4 * r;
      if (L.lr == 0) { lds[i] = C.Jv[r]; lds[32 + i] = hdrJ[r]; }
      if (L.lr == 1) lds[16 + i] = C.Jv[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (xok) {
      const double weight = (m == 0) ? 0.5 : 1.0;
      const double cs = weight * ((xk < 2) ? a.cos_mphi[xv + a.nVza * m] : a.sin_mphi[xv + a.nVza * m]);
      const int row = (a.node[xv] - 1) * nS + xk;
      accT += cs * lds[row];
      accR += cs * lds[16 + row];
      if (m == 0) accH = cs * lds[32 + row];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (xok) {
    const size_t idx = xv + (size_t)a.nVza * (xk + (size_t)nS * n);
    a.R[idx] = accR;
    a.T[idx] = accT;
    a.hdr[idx] = accH;
  }
  if (bad && lane == 0) atomicMax(a.info, bad);
}

}  // namespace momw

hipError_t momw_launch_sweep(const void *args, hipStream_t st) {
  const momw::WArgs a = *reinterpret_cast<const momw::WArgs *>(args);
  const dim3 grid((unsigned)((a.S + 3) / 4)), block(256);
  if (a.N <= 8) hipLaunchKernelGGL(momw::k_wsweep<2>, grid, block, 0, st, a);
  else if (a.N <= 12) hipLaunchKernelGGL(momw::k_wsweep<3>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(momw::k_wsweep<4>, grid, block, 0, st, a);
  return hipGetLastError();
}
