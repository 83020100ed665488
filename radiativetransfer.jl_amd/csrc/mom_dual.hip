// mom_dual.hip -- ForwardDiff.Dual through the elastic hot path: rt_run on Dual numbers (values + P partials).
//
// The reference differentiates rt_run by running it on ForwardDiff.Dual element types (rt_run.jl:89-96 allocates R, T,
// R_SFI, T_SFI in the Dual type; every layer operator is a Dual array; gpu_batched.jl:100-150 gives the two batched
// operators their Dual methods: C = A B, dC_i = A dB_i + dA_i B;  X = A^-1, dX_i = -X dA_i X).  Here the same run is the
// TANGENT-LINEAR sweep of the layer loop, written for the GPU as batched launches over HBM-resident "dual matrices":
//
//   DM  [1 + P][U][N x N]   component 0 = the value, component c = the partial c (column-major N x N per unit, no padding)
//   DV  [1 + P][U][N]       the same for source vectors;   U = units (spectral points) of the current chunk
//
// Every statement of rt_kernel!/elemental!/doubling_helper!/interaction_helper! (file:line at each step below) becomes one
// launch that carries all components: products by k_dgemm (64 x 64 tiles on v_mfma_f64_16x16x4, the Dual product rule
// applied per component: c = 0: A0 B0, c > 0: Ac B0 + A0 Bc), inverses by k_dinv (pivoted Gauss-Jordan in LDS, value only)
// followed by two products per partial ((A G)_c = (A_c + (A G)_0 W_c) G0), matrix-vector statements by k_dmatvec, the elemental
// layer by k_delemental (analytic derivatives of get_elem_rt! / get_elem_rt_SFI!).  The operators of value AND partials do not
// fit one CU's LDS (5 live operators x (1 + P) x 29 KB per unit at N = 60), which is why this path streams them instead of using
// the fused LDS-resident images of the value run (mom_kernels.hpp, mom_q4.hpp).  Measured (profiles/r06_dual_ab.txt): neither
// HBM (1.5-2.5 TB/s per product launch) nor the matrix pipe (0.4-0.56 busy) is at its roof; half of k_dgemm's time at N = 60 is
// the fixed cost of an item (start-up, first loads, epilogue), which is why the kernel is small (17.7 KB, five per CU).
//
// Units are independent: a scene larger than the workspace budget is processed in chunks of units.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "mom_host.hpp"

namespace momd {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 64, KC = 16, LDB = KC + 2;   // the 64 x 64 tile, K chunks of 16: 17.7 KB of LDS per workgroup

// ---------------------------------------------------------------------------------------------------------------------
// C_c = alpha * prod_c + beta * E_c + eye * (c == 0) I          mode 0: Dual product rule, 1: A0 Bc, 2: Ac B0
// ---------------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  int N, U, c0, nc, mode, tiles_i, tiles_j;
  const double *A, *B, *E;
  double *C;
  double alpha, beta, eye;
  // riding vectors (mode 0 only): x[q] are columns N + q of the right operand, so y[q] = add[q] + (A x[q]) comes out of the same
  // product with the same Dual rule -- at N = 60 in columns 60, 61 of the 64-wide tile, which the padding computes anyway.  These
  // are the matrix-vector statements of doubling.jl:57-60 and interaction.jl:82,100 (the source-vector updates).
  int nq;
  const double *x[2], *add[2];
  double *y[2];
};

#ifdef MOMD_STAMPS
// diagnostic build only (tools/dgemm_stamps.py; never part of the shipped library): s_memtime deltas of the sections of a (term, K chunk)
// phase, wave 0 of one workgroup in the middle of every launch
__device__ unsigned long long momd_stamp_acc[8];
#define MOMD_STAMP(id)                                                          \
  do {                                                                          \
    if (stamp_on) {                                                             \
      __builtin_amdgcn_sched_barrier(0);                                        \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
      if (threadIdx.x == 0) atomicAdd(&momd_stamp_acc[id], now_ - stamp_last);  \
      stamp_last = now_;                                                        \
      __builtin_amdgcn_sched_barrier(0);                                        \
    }                                                                           \
  } while (0)
extern "C" void momd_stamps_read(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(momd_stamp_acc), sizeof(momd_stamp_acc));
  if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(momd_stamp_acc), z, sizeof(z)); }
}
#else
#define MOMD_STAMP(id)
#endif
// NW wavefronts per workgroup = a tile of 16 NW x 16 NW (NW = 4: the 64 x 64 tile; 5, 6: ONE tile for operator edges up to 80 / 96, where
// 2 x 2 tiles of 64 would spend three quarters of their work on padding).  Wave w owns columns 16 w .. 16 w + 15 and NW row blocks.
template <bool VEC2, bool RIDE, int NW>
__global__ void __launch_bounds__(64 * NW) k_dgemm(GemmArgs a) {
  constexpr int TMw = 16 * NW, TNw = 16 * NW, LDAw = TMw + 2, NTH = 64 * NW;
  static_assert((TMw / 2) * KC % NTH == 0 && TNw * (KC / 2) % NTH == 0, "staging pieces per thread");
  constexpr int NQw = (TMw / 2) * KC / NTH;   // 16-byte pieces of a staged tile per thread (A and B alike)
#ifdef MOMD_STAMPS
  const bool stamp_on = (blockIdx.x == gridDim.x / 2) && (threadIdx.x < 64);
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
  __shared__ __attribute__((aligned(16))) double smem[KC * LDAw + TNw * LDB];
  double *As = smem, *Bs = smem + KC * LDAw;   // As[i + k LDA], Bs[k + j LDB]; the epilogue reuses the space as Ct[i + j LDA]
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int N = a.N;
  const size_t NN = (size_t)N * N, CS = (size_t)a.U * NN;
  const int ntile = a.tiles_i * a.tiles_j, nb = ntile * a.nc;
  // workgroups b and b + 8 share an XCD (MI355X_MICROARCH.md, dispatch): all tiles and components of one unit go to ONE XCD, next
  // to each other in time, so the value operands A0 / B0 every component needs are read from HBM once and from that L2 afterwards
  const int slot = blockIdx.x >> 3, inner = slot % nb, unit = (slot / nb) * 8 + (blockIdx.x & 7);
  if (unit >= a.U) return;
  const int tile = inner % ntile, c = a.c0 + inner / ntile;
  const int i0 = (tile % a.tiles_i) * TMw, j0 = (tile / a.tiles_i) * TNw;
  const size_t uo = (size_t)unit * NN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lq = lane >> 4, lr = lane & 15;
  d4 acc[NW];
#pragma unroll
  for (int tb = 0; tb < NW; ++tb) acc[tb] = d4{0.0, 0.0, 0.0, 0.0};
  int nterms = 1;
  const double *A0 = a.A + uo, *B0 = a.B + uo, *Ac = a.A + c * CS + uo, *Bc = a.B + c * CS + uo;
  const int nq = RIDE ? a.nq : 0;
  const size_t vo0 = (size_t)unit * N, voc = ((size_t)c * a.U + unit) * N;
  // term 0: (TA0, TB0); term 1 (Dual rule, c > 0): (A0, Bc).  Selected by value, not through indexed arrays (those went to scratch).
  const double *TA0, *TB0;
  if (a.mode == 0) {
    if (c == 0) { TA0 = A0; TB0 = B0; }
    else { nterms = 2; TA0 = Ac; TB0 = B0; }
  } else if (a.mode == 1) { TA0 = A0; TB0 = Bc; }
  else { TA0 = Ac; TB0 = B0; }
  const double *x0p = RIDE ? a.x[0] : nullptr, *x1p = (RIDE && nq > 1) ? a.x[1] : nullptr;
  const int nch = (N + KC - 1) / KC, nph = nterms * nch;
  // the next (term, K chunk) is fetched into registers while the matrix cores work on the current one
  d2 ra[NQw], rb[NQw];
  double sa[2 * NQw], sb[2 * NQw];
  auto gload = [&](int ph) {
    const bool t1 = ph >= nch;
    const double *Ag = t1 ? A0 : TA0, *Bg = t1 ? Bc : TB0;
    const size_t xo = t1 ? voc : vo0;   // the riding vectors carry the component of this term's right operand
    const int k0 = (t1 ? ph - nch : ph) * KC;
    if (VEC2) {  // N even: pairs along the contiguous axis are 16-byte aligned and never straddle the edge
#pragma unroll
      for (int q = 0; q < NQw; ++q) {
        const int e = t + NTH * q;
        { const int i = 2 * (e % (TMw / 2)), k = e / (TMw / 2), gi = i0 + i, gk = k0 + k;
          ra[q] = (gi < N && gk < N) ? *(const d2 *)(Ag + gi + (size_t)gk * N) : d2{0.0, 0.0}; }
        { const int k = 2 * (e % (KC / 2)), j = e / (KC / 2), gk = k0 + k, gj = j0 + j;
          d2 v = d2{0.0, 0.0};
          if (gk < N) {
            if (gj < N) v = *(const d2 *)(Bg + gk + (size_t)gj * N);
            else if (RIDE && gj - N < nq) v = *(const d2 *)((gj == N ? x0p : x1p) + xo + gk);
          }
          rb[q] = v; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2 * NQw; ++q) {
        const int e = t + NTH * q;
        { const int i = e % TMw, k = e / TMw, gi = i0 + i, gk = k0 + k;
          sa[q] = (gi < N && gk < N) ? Ag[gi + (size_t)gk * N] : 0.0; }
        { const int k = e % KC, j = e / KC, gk = k0 + k, gj = j0 + j;
          double v = 0.0;
          if (gk < N) {
            if (gj < N) v = Bg[gk + (size_t)gj * N];
            else if (RIDE && gj - N < nq) v = (gj == N ? x0p : x1p)[xo + gk];
          }
          sb[q] = v; }
      }
    }
  };
  auto sstore = [&]() {
    if (VEC2) {
#pragma unroll
      for (int q = 0; q < NQw; ++q) {
        const int e = t + NTH * q;
        *(d2 *)(As + 2 * (e % (TMw / 2)) + (e / (TMw / 2)) * LDAw) = ra[q];
        *(d2 *)(Bs + 2 * (e % (KC / 2)) + (e / (KC / 2)) * LDB) = rb[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2 * NQw; ++q) {
        const int e = t + NTH * q;
        As[(e % TMw) + (e / TMw) * LDAw] = sa[q];
        Bs[(e % KC) + (e / KC) * LDB] = sb[q];
      }
    }
  };
  gload(0);
  MOMD_STAMP(0);   // start-up: argument loads, index arithmetic, the first loads going out
  for (int ph = 0; ph < nph; ++ph) {
    __syncthreads();
    MOMD_STAMP(1); // barrier at the top of a phase
    sstore();
    MOMD_STAMP(2); // wait for this phase's global loads + the LDS stores
    __syncthreads();
    MOMD_STAMP(3); // barrier after the stores
    if (ph + 1 < nph) gload(ph + 1);
    MOMD_STAMP(4); // issue of the next phase's loads
    const int kmax = min(KC, N - (ph >= nch ? ph - nch : ph) * KC);
    // the product transposed on the matrix core (rows of the MFMA tile = columns j of C, columns = rows i): the 16 lanes of a
    // quarter-wave then hold 16 consecutive rows of one column of C, so the stores of the epilogue are 128-byte segments
    for (int kk = 0; kk < kmax; kk += 4) {
      const double bv = Bs[(kk + lq) + (16 * w + lr) * LDB];
#pragma unroll
      for (int tb = 0; tb < NW; ++tb) {
        const double av = As[(16 * tb + lr) + (kk + lq) * LDAw];
        acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc[tb], 0, 0, 0);
      }
    }
    MOMD_STAMP(5); // the chunk's k-steps: LDS operand reads + MFMAs
  }
  MOMD_STAMP(6);   // drain
  // Epilogue through LDS: the accumulators hold 16-row runs of scattered columns (128-byte pieces that straddle the 128-byte lines of a
  // 60-row column); written from there, the stores were the largest single cost of the kernel (the run without this kernel's MFMAs
  // is 9 % shorter, without its global loads 17 %; profiles/r06_dual_ab.txt).  Staged as the tile Ct[i + j LDA], the result leaves in
  // 16-byte pieces along the contiguous axis -- whole lines -- and the E operand is read the same way.
  double *Ct = smem;   // 32 columns at a time: 32 x LDAw doubles fit the staging space (NW = 4: 2112 of 2208)
  static_assert(32 * LDAw <= KC * LDAw + TNw * LDB, "epilogue pass does not fit the staging space");
  constexpr int NPASS = (TNw + 31) / 32;
  double *C = a.C + c * CS + uo;
  const double *E = a.E ? a.E + c * CS + uo : nullptr;
  for (int half = 0; half < NPASS; ++half) {
    __syncthreads();
    if ((w >> 1) == half) {
#pragma unroll
      for (int tb = 0; tb < NW; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(16 * tb + lr) + (16 * (w & 1) + lq + 4 * r) * LDAw] = acc[tb][r];
    }
    __syncthreads();
    if (VEC2) {
      const int i = 2 * (t % (TMw / 2)), gi = i0 + i;
      for (int jl = t / (TMw / 2); jl < 32; jl += NTH / (TMw / 2)) {
        const int gj = j0 + 32 * half + jl;
        if (gi < N && gj < N) {
          const size_t o = gi + (size_t)gj * N;
          d2 v = *(const d2 *)(Ct + i + jl * LDAw);
          v = v * a.alpha;
          if (E) v += a.beta * *(const d2 *)(E + o);
          if (c == 0) { if (gi == gj) v[0] += a.eye; if (gi + 1 == gj) v[1] += a.eye; }
          *(d2 *)(C + o) = v;
        } else if (RIDE && gi < N && gj >= N && gj - N < nq) {
          const double *ad = (gj == N) ? a.add[0] : a.add[1];
          double *y = (gj == N) ? a.y[0] : a.y[1];
          d2 v = *(const d2 *)(Ct + i + jl * LDAw);
          if (ad) v += *(const d2 *)(ad + voc + gi);
          *(d2 *)(y + voc + gi) = v;
        }
      }
    } else {
      const int i = t % TMw, gi = i0 + i;
      for (int jl = t / TMw; jl < 32; jl += NTH / TMw) {
        const int gj = j0 + 32 * half + jl;
        if (gi < N && gj < N) {
          const size_t o = gi + (size_t)gj * N;
          double v = a.alpha * Ct[i + jl * LDAw];
          if (E) v += a.beta * E[o];
          if (c == 0 && gi == gj) v += a.eye;
          C[o] = v;
        } else if (RIDE && gi < N && gj >= N && gj - N < nq) {
          const double *ad = (gj == N) ? a.add[0] : a.add[1];
          double *y = (gj == N) ? a.y[0] : a.y[1];
          y[voc + gi] = (ad ? ad[voc + gi] : 0.0) + Ct[i + jl * LDAw];
        }
      }
    }
  }
}

// The same product for operators of edge N + nq <= 16 NT (shipped: NT = 1): ONE WAVEFRONT per (unit, component), operands read straight
// from global memory in the layout of the MFMA (a 30 x 30 operator is 7 KB: cache-resident), no LDS, no barrier.  The 64 x 64
// workgroup tile above costs the same for every edge up to 64 (N = 30: 4.5 x the MFMA work, two staging phases per term).
template <int NT, bool RIDE>
__global__ void __launch_bounds__(256) k_dgemm_w(GemmArgs a) {
  const int N = a.N;
  const size_t NN = (size_t)N * N, CS = (size_t)a.U * NN;
  const int lane = threadIdx.x & 63, lq = lane >> 4, lr = lane & 15;
  const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // the four waves of a workgroup: neighbouring components of a unit
  const size_t unit = item / a.nc;
  if (unit >= (size_t)a.U) return;
  const int c = a.c0 + (int)(item % a.nc);
  const size_t uo = unit * NN;
  const double *A0 = a.A + uo, *B0 = a.B + uo, *Ac = a.A + c * CS + uo, *Bc = a.B + c * CS + uo;
  const int nq = RIDE ? a.nq : 0;
  const size_t vo0 = unit * N, voc = ((size_t)c * a.U + unit) * N;
  int nterms = 1;
  const double *TA0, *TB0;
  if (a.mode == 0) {
    if (c == 0) { TA0 = A0; TB0 = B0; }
    else { nterms = 2; TA0 = Ac; TB0 = B0; }
  } else if (a.mode == 1) { TA0 = A0; TB0 = Bc; }
  else { TA0 = Ac; TB0 = B0; }
  d4 acc[NT][NT];
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[tj][ti] = d4{0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < nterms; ++t) {
    const double *Ag = t ? A0 : TA0, *Bg = t ? Bc : TB0;
    const size_t xo = t ? voc : vo0;
#pragma unroll 4
    for (int kk = 0; kk < N; kk += 4) {
      const int k = kk + lq;
      double bv[NT], av[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int col = 16 * tj + lr;
        double v = 0.0;
        if (k < N) {
          if (col < N) v = Bg[k + (size_t)col * N];
          else if (RIDE && col - N < nq) v = (col == N ? a.x[0] : a.x[1])[xo + k];
        }
        bv[tj] = v;
      }
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int row = 16 * ti + lr;
        av[ti] = (k < N && row < N) ? Ag[row + (size_t)k * N] : 0.0;
      }
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[tj], av[ti], acc[tj][ti], 0, 0, 0);
    }
  }
  double *C = a.C + c * CS + uo;
  const double *E = a.E ? a.E + c * CS + uo : nullptr;
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = 16 * tj + lq + 4 * r, gi = 16 * ti + lr;
        if (gi < N && gj < N) {
          const size_t o = gi + (size_t)gj * N;
          double v = a.alpha * acc[tj][ti][r];
          if (E) v += a.beta * E[o];
          if (c == 0 && gi == gj) v += a.eye;
          C[o] = v;
        } else if (RIDE && gi < N && gj - N < nq) {
          const double *ad = (gj == N) ? a.add[0] : a.add[1];
          double *y = (gj == N) ? a.y[0] : a.y[1];
          y[voc + gi] = (ad ? ad[voc + gi] : 0.0) + acc[tj][ti][r];
        }
      }
}

// ---------------------------------------------------------------------------------------------------------------------
// G0 = inv(eye I + s W0): batch_inv! (gpu_batched.jl:61-82) on the VALUE component, pivoted Gauss-Jordan in LDS, one
// workgroup per unit.  The partials follow as products (gpu_batched.jl:129-150).
// ---------------------------------------------------------------------------------------------------------------------
struct InvArgs {
  int N, U;
  const double *W;
  double *G;
  double eye, s;
  int *info;
};

// The matrix lives in REGISTERS, rows across the lanes: thread (lane l, wave w) owns A[l + 64 ib][w + 4 s].  Per elimination
// step only the pivot row (16 NBK values per wave) and column k (one value per lane) pass through LDS; everything that is
// not the rank-one update -- the pivot search of the next column (a wave reduction), its reciprocal, the rewrite of column
// k -- is done by the ONE wave that owns that column, under a wave-uniform branch, so the other three waves do not issue
// it.  Lane k gets the coefficient val - 1, which turns the common update R - c * (row k * pinv) into row k * pinv for
// the pivot row itself.  Rows are exchanged (lanes p and k, through LDS) only when partial pivoting asks for it.
// Two barriers per step.  (The first forms kept the matrix in LDS, 600 us per launch of 2000 units of N = 60, then in
// registers with rows across the waves, 300 us: profiles/r06_dual_ab.txt.)
#define MOMD_ROW_CASES(F)                                                                                          \
  F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17) F(18) F(19) F(20) F(21) \
  F(22) F(23) F(24) F(25) F(26) F(27) F(28) F(29) F(30) F(31)
template <int NBK>
__global__ void __launch_bounds__(256) k_dinv(InvArgs a) {
  constexpr int RPT = 16 * NBK;   // columns per thread
  extern __shared__ double sm[];
  const int N = a.N, t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // no staging matrix: with rows across the lanes, column j of the operator is 64 consecutive lanes = one coalesced access, both
  // ways; the 2 KB of LDS left (pivot row, column k, bookkeeping) let the register count, not the LDS, set the workgroups per CU
  double *rowk = sm, *rowx = rowk + 4 * RPT, *colk = rowx + 4 * RPT, *pv = colk + 64 * NBK;  // pv: val, pinv
  int *piv = (int *)(pv + 2), *pcur = piv + 64 * NBK, *dst = pcur + 2;
  const size_t NN = (size_t)N * N, uo = (size_t)blockIdx.x * NN;
  // thread (lane, w) owns A[lane + 64 ib][w + 4 s], s < 16 NBK, ib < NBK.  The registers are FOUR arrays of 16 doubles -- [s < 16 | s >= 16]
  // x [ib 0 | ib 1] -- because a wave-uniform dynamic index into 16 doubles is a v_movrel while one 64-double array went to scratch
  // (N = 80: 3.8 ms per launch of 1000 units, 65 % of the Dual run).  `each` runs a body over the halves with static indices.
  double A0[16], A1[16], B0[16], B1[16];
  auto each = [&](auto &&f) {
    f(A0, A1, 0);
    if constexpr (NBK > 1) f(B0, B1, 16);
  };
  // column `scol` of this wave, scol wave-uniform: f(entry of row block 0, entry of row block 1).  One register array of 16 doubles takes a
  // dynamic index as a v_movrel (NBK = 1); with four of them the compiler keeps them in scratch, so there the entries are fetched and
  // put back through a scalar switch over statically indexed registers (cases of two moves)
#define MOMD_C16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
  auto with_col = [&](int scol, bool writeback, auto &&f) {
    if constexpr (NBK == 1) {
      f(A0[scol & 15], A1[0]);
    } else {
      double e0 = 0.0, e1 = 0.0;
      switch (scol) {
#define MOMD_G(S) case S: e0 = A0[S]; e1 = A1[S]; break;
        MOMD_C16(MOMD_G)
#undef MOMD_G
#define MOMD_G(S) case 16 + S: e0 = B0[S]; e1 = B1[S]; break;
        MOMD_C16(MOMD_G)
#undef MOMD_G
        default: break;
      }
      f(e0, e1);
      if (writeback) {
        switch (scol) {
#define MOMD_G(S) case S: A0[S] = e0; A1[S] = e1; break;
          MOMD_C16(MOMD_G)
#undef MOMD_G
#define MOMD_G(S) case 16 + S: B0[S] = e0; B1[S] = e1; break;
          MOMD_C16(MOMD_G)
#undef MOMD_G
          default: break;
        }
      }
    }
  };
#undef MOMD_C16
  each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int j = w + 4 * (sbase + s2);
      r0[s2] = (lane < N && j < N) ? a.s * a.W[uo + lane + (size_t)j * N] + (lane == j ? a.eye : 0.0) : 0.0;
      if constexpr (NBK > 1) r1[s2] = (lane + 64 < N && j < N) ? a.s * a.W[uo + lane + 64 + (size_t)j * N] + (lane + 64 == j ? a.eye : 0.0) : 0.0;
    }
  });
  // pivot of column kc among rows >= kc: executed by the wave that owns the column; leaves p, val, 1 / val in LDS.  The
  // maximum goes down the rows of 16 by DPP shifts and across them by readlane (a __shfl ladder is six exposed LDS round
  // trips on the critical path of all four waves); the row that holds it comes from a ballot.
  auto search = [&](int kc) {
    with_col(kc >> 2, false, [&](double &e0, double &e1) {
      double best = -1.0, val = 0.0;
      int bi = kc;
      if (lane >= kc && lane < N) { best = fabs(e0); val = e0; bi = lane; }
      if constexpr (NBK > 1) {
        const int i = lane + 64;
        if (i >= kc && i < N && fabs(e1) > best) { best = fabs(e1); val = e1; bi = i; }
      }
      double m = best;
      union { double d; int w2[2]; } ua, ub;
#define MOMD_DPP_MAX(ctrl)                                                    \
  ua.d = m;                                                                   \
  ub.w2[0] = __builtin_amdgcn_update_dpp(ua.w2[0], ua.w2[0], ctrl, 0xf, 0xf, false); \
  ub.w2[1] = __builtin_amdgcn_update_dpp(ua.w2[1], ua.w2[1], ctrl, 0xf, 0xf, false); \
  m = fmax(m, ub.d);
      MOMD_DPP_MAX(0x111) MOMD_DPP_MAX(0x112) MOMD_DPP_MAX(0x114) MOMD_DPP_MAX(0x118)   // row_shr 1, 2, 4, 8: lane 15 of each row
#undef MOMD_DPP_MAX
      ua.d = m;
      double mx = -1.0;
#pragma unroll
      for (int row = 0; row < 4; ++row) {
        union { double d; int w2[2]; } tt;
        tt.w2[0] = __builtin_amdgcn_readlane(ua.w2[0], 16 * row + 15);
        tt.w2[1] = __builtin_amdgcn_readlane(ua.w2[1], 16 * row + 15);
        mx = fmax(mx, tt.d);
      }
      const unsigned long long hit = __ballot(best == mx);
      const int src_lane = __builtin_ctzll(hit);          // the lowest lane holding the maximum
      union { double d; int w2[2]; } uv, ur;
      uv.d = val;
      ur.w2[0] = __builtin_amdgcn_readlane(uv.w2[0], src_lane);
      ur.w2[1] = __builtin_amdgcn_readlane(uv.w2[1], src_lane);
      const int prow = __builtin_amdgcn_readlane(bi, src_lane);
      if (lane == 0) {
        double x = __builtin_amdgcn_rcp(ur.d);             // 1 / val to the last bit or two: two Newton steps on v_rcp_f64
        x = x * (2.0 - ur.d * x);
        x = x * (2.0 - ur.d * x);
        pv[0] = ur.d; pv[1] = x; pcur[0] = prow; pcur[1] = (mx > 0.0) ? 0 : 1;
      }
    });
  };
  if (w == 0) search(0);
  __syncthreads();
  bool bad = false;
  for (int k = 0; k < N; ++k) {
    const int p = __builtin_amdgcn_readfirstlane(pcur[0]);
    bad = bad || (pcur[1] != 0);
    const double val = pv[0], pinv = pv[1];
    const int kb = k >> 6, kl = k & 63, wk = k & 3, sk = k >> 2;
    if (t == 0) piv[k] = p;
    if (p != k) {  // (block-uniform) exchange rows p and k: every wave hands its 16 NBK entries of both rows through LDS
      const int pb = p >> 6, pl = p & 63;
      each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          if (lane == pl) rowx[w * RPT + sbase + s2] = (NBK > 1 && pb == 1) ? r1[s2] : r0[s2];
          if (lane == kl) rowk[w * RPT + sbase + s2] = (NBK > 1 && kb == 1) ? r1[s2] : r0[s2];
        }
      });
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are complete (same-wave exchange, no barrier needed)
      __builtin_amdgcn_wave_barrier();
      each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          if (lane == pl) { const double v = rowk[w * RPT + sbase + s2]; if (NBK > 1 && pb == 1) r1[s2] = v; else r0[s2] = v; }
          if (lane == kl) { const double v = rowx[w * RPT + sbase + s2]; if (NBK > 1 && kb == 1) r1[s2] = v; else r0[s2] = v; }
        }
      });
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    // stage the scaled pivot row (each wave its own columns) and column k (its owner wave)
    if (lane == kl)
      each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) rowk[w * RPT + sbase + s2] = ((NBK > 1 && kb == 1) ? r1[s2] : r0[s2]) * pinv;
      });
    if (w == wk)
      with_col(sk, false, [&](double &e0, double &e1) {
        colk[lane] = e0;
        if constexpr (NBK > 1) colk[lane + 64] = e1;
      });
    __syncthreads();
    double c[2], c0[2];
#pragma unroll
    for (int ib = 0; ib < NBK; ++ib) {
      c0[ib] = colk[lane + 64 * ib];
      c[ib] = (ib == kb && lane == kl) ? val - 1.0 : c0[ib];
    }
    each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const double rk = rowk[w * RPT + sbase + s2];
        r0[s2] = r0[s2] - c[0] * rk;
        if constexpr (NBK > 1) r1[s2] = r1[s2] - c[1] * rk;
      }
    });
    if (w == wk)   // column k of the inverse in progress: pinv on the pivot row, -c_i pinv elsewhere
      with_col(sk, true, [&](double &e0, double &e1) {
        e0 = (kb == 0 && lane == kl) ? pinv : -c0[0] * pinv;
        if constexpr (NBK > 1) e1 = (kb == 1 && lane == kl) ? pinv : -c0[1] * pinv;
      });
    if (k + 1 < N && w == ((k + 1) & 3)) search(k + 1);
    __syncthreads();
  }
  // undo the row exchanges as column exchanges in reverse order: dst[j] = final position of the working column j, followed
  // through the exchanges by one thread per column (the loads of piv[] do not depend on the position: no serial LDS chain)
  for (int j = t; j < N; j += 256) {
    int pos = j;
    for (int k = N - 1; k >= 0; --k) {
      const int p = piv[k];
      pos = (pos == k) ? p : ((pos == p) ? k : pos);
    }
    dst[j] = pos;
  }
  __syncthreads();
  each([&](double (&r0)[16], double (&r1)[16], int sbase) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int j = w + 4 * (sbase + s2);
      if (j < N) {
        if (lane < N) a.G[uo + lane + (size_t)dst[j] * N] = r0[s2];
        if (NBK > 1 && lane + 64 < N) a.G[uo + lane + 64 + (size_t)dst[j] * N] = r1[s2];
      }
    }
  });
  if (t == 0 && bad) atomicMax(a.info, 1);
}

// The inverse for N <= NC (16, 32): ONE WAVEFRONT per unit, lane i holds row i in registers.  Pivot search down the lanes (DPP
// shifts, readlane, ballot), the pivot row broadcast by readlane (scalar operands of the update), no LDS traffic and no barrier in
// the elimination; rows are exchanged (through LDS) only when the pivot is not on the diagonal.
template <int NC>
__global__ void __launch_bounds__(256) k_dinv_w(InvArgs a) {
  __shared__ double xch[4][2][NC];
  __shared__ int perm[4][2][NC];   // [wave][piv | dst][.]
  const int N = a.N, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t unit = (size_t)blockIdx.x * 4 + w;
  if (unit >= (size_t)a.U) return;
  const size_t NN = (size_t)N * N, uo = unit * NN;
  // the row in halves of 16 registers, each its own array: a wave-uniform dynamic index into 16 doubles is a v_movrel, one
  // 32-element array went to scratch.  `each` runs a body over both halves with static indices.
  constexpr int NH = NC / 16;
  double Ra[16], Rb[16];
  auto each = [&](auto &&f) {
    f(Ra, 0);
    if constexpr (NH > 1) f(Rb, 16);
  };
  each([&](double (&r)[16], int base) {
#pragma unroll
    for (int j2 = 0; j2 < 16; ++j2) {
      const int j = base + j2;
      r[j2] = (lane < N && j < N) ? a.s * a.W[uo + lane + (size_t)j * N] + (lane == j ? a.eye : 0.0) : 0.0;
    }
  });
  // column k of every lane's row, k wave-uniform: a scalar switch over statically indexed registers (cases of one move each)
#define MOMD_C16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
  auto col_get = [&](int k) {
    double v = 0.0;
    switch (k) {
#define MOMD_G(S) case S: v = Ra[S]; break;
      MOMD_C16(MOMD_G)
#undef MOMD_G
#define MOMD_G(S) case 16 + S: if constexpr (NH > 1) v = Rb[S]; break;
      MOMD_C16(MOMD_G)
#undef MOMD_G
      default: break;
    }
    return v;
  };
  auto col_set = [&](int k, double v) {
    switch (k) {
#define MOMD_G(S) case S: Ra[S] = v; break;
      MOMD_C16(MOMD_G)
#undef MOMD_G
#define MOMD_G(S) case 16 + S: if constexpr (NH > 1) Rb[S] = v; break;
      MOMD_C16(MOMD_G)
#undef MOMD_G
      default: break;
    }
  };
#undef MOMD_C16
  bool bad = false;
  union U2 { double d; int w2[2]; };
  auto lane_get = [&](double v, int src) {
    U2 u, r;
    u.d = v;
    r.w2[0] = __builtin_amdgcn_readlane(u.w2[0], src);
    r.w2[1] = __builtin_amdgcn_readlane(u.w2[1], src);
    return r.d;
  };
  for (int kv = 0; kv < N; ++kv) {
    const int k = __builtin_amdgcn_readfirstlane(kv);
    const double colv = col_get(k);
    double best = (lane >= k && lane < N) ? fabs(colv) : -1.0;
    double m = best;
    U2 ua, ub;
#define MOMD_DPP_MAX(ctrl)                                                    \
  ua.d = m;                                                                   \
  ub.w2[0] = __builtin_amdgcn_update_dpp(ua.w2[0], ua.w2[0], ctrl, 0xf, 0xf, false); \
  ub.w2[1] = __builtin_amdgcn_update_dpp(ua.w2[1], ua.w2[1], ctrl, 0xf, 0xf, false); \
  m = fmax(m, ub.d);
    MOMD_DPP_MAX(0x111) MOMD_DPP_MAX(0x112) MOMD_DPP_MAX(0x114) MOMD_DPP_MAX(0x118)
#undef MOMD_DPP_MAX
    double mx = lane_get(m, 15);
    if (NC > 16) mx = fmax(mx, lane_get(m, 31));
    const int p = __builtin_ctzll(__ballot(best == mx));
    if (!(mx > 0.0)) bad = true;
    if (lane == 0) perm[w][0][k] = p;
    if (p != k) {   // exchange rows p and k (lanes p and k) through LDS
      each([&](double (&r)[16], int base) {
        if (lane == p) {
#pragma unroll
          for (int j2 = 0; j2 < 16; ++j2) xch[w][0][base + j2] = r[j2];
        }
        if (lane == k) {
#pragma unroll
          for (int j2 = 0; j2 < 16; ++j2) xch[w][1][base + j2] = r[j2];
        }
      });
      __builtin_amdgcn_wave_barrier();
      each([&](double (&r)[16], int base) {
        if (lane == p) {
#pragma unroll
          for (int j2 = 0; j2 < 16; ++j2) r[j2] = xch[w][1][base + j2];
        }
        if (lane == k) {
#pragma unroll
          for (int j2 = 0; j2 < 16; ++j2) r[j2] = xch[w][0][base + j2];
        }
      });
      __builtin_amdgcn_wave_barrier();
    }
    const double c0 = col_get(k);           // column k after the exchange
    const double val = lane_get(c0, k);
    double pinv = __builtin_amdgcn_rcp(val);
    pinv = pinv * (2.0 - val * pinv);
    pinv = pinv * (2.0 - val * pinv);
    const double c = (lane == k) ? val - 1.0 : c0;   // lane k: R_k - (val - 1) (R_k pinv) = R_k pinv
    each([&](double (&r)[16], int base) {
#pragma unroll
      for (int j2 = 0; j2 < 16; ++j2)
        if (base + j2 < N) r[j2] = r[j2] - c * (lane_get(r[j2], k) * pinv);
    });
    col_set(k, (lane == k) ? pinv : -c0 * pinv);
  }
  // undo the row exchanges as column exchanges in reverse order (one lane per column follows its position)
  __builtin_amdgcn_wave_barrier();
  int pos = lane;
  for (int k = N - 1; k >= 0; --k) {
    const int p = perm[w][0][k];
    pos = (pos == k) ? p : ((pos == p) ? k : pos);
  }
  if (lane < N) perm[w][1][lane] = pos;
  __builtin_amdgcn_wave_barrier();
  each([&](double (&r)[16], int base) {
#pragma unroll
    for (int j2 = 0; j2 < 16; ++j2)
      if (base + j2 < N && lane < N) a.G[uo + lane + (size_t)perm[w][1][base + j2] * N] = r[j2];
  });
  if (bad && lane == 0) atomicMax(a.info, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// y_c = add_c + (M x)_c with the Dual product rule, for up to two (x, add, y) sets sharing M
// ---------------------------------------------------------------------------------------------------------------------
struct MvArgs {
  int N, U, P, nq;
  const double *M;
  const double *x[2], *add[2];
  double *y[2];
};

__global__ void __launch_bounds__(256) k_dmatvec(MvArgs a) {
  __shared__ double part[2][4][64];
  // all components of a unit on one XCD (see k_dgemm): M0 comes from HBM once
  const int nb = a.P + 1, slot = blockIdx.x >> 3, c = slot % nb, unit = (slot / nb) * 8 + (blockIdx.x & 7);
  if (unit >= a.U) return;
  const int N = a.N, t = threadIdx.x, g = t >> 6, il = t & 63, nq = a.nq;
  const size_t NN = (size_t)N * N, u = unit, vo0 = u * N, voc = ((size_t)c * a.U + u) * N;
  const double *M0 = a.M + u * NN, *Mc = a.M + ((size_t)c * a.U + u) * NN;
  for (int rb = 0; rb < N; rb += 64) {
    const int i = rb + il;
    double s0 = 0.0, s1 = 0.0;
    if (i < N) {
      // the matrix is read once for both vector sets
      if (c == 0) {
        for (int j = g; j < N; j += 4) {
          const double m0 = M0[i + (size_t)j * N];
          s0 += m0 * a.x[0][vo0 + j];
          if (nq > 1) s1 += m0 * a.x[1][vo0 + j];
        }
      } else {
        for (int j = g; j < N; j += 4) {
          const double m0 = M0[i + (size_t)j * N], mc = Mc[i + (size_t)j * N];
          s0 += mc * a.x[0][vo0 + j] + m0 * a.x[0][voc + j];
          if (nq > 1) s1 += mc * a.x[1][vo0 + j] + m0 * a.x[1][voc + j];
        }
      }
    }
    part[0][g][il] = s0;
    part[1][g][il] = s1;
    __syncthreads();
    if (g < nq && i < N) {
      const double *ad = a.add[g];
      a.y[g][voc + i] = (ad ? ad[voc + i] : 0.0) + ((part[g][0][il] + part[g][1][il]) + (part[g][2][il] + part[g][3][il]));
    }
    __syncthreads();
  }
}

// j1+ = j0+ e, j1- = j0- e (doubling.jl:51-54) on Duals, then expk .= expk.^2 (:62); one block per unit
struct ScaleArgs {
  int N, U, P;
  const double *jp, *jm;
  double *j1p, *j1m, *e;
};
__global__ void k_dscale(ScaleArgs a) {
  const size_t u = blockIdx.x, US = (size_t)a.U;
  const double e0 = a.e[u];
  for (int idx = threadIdx.x; idx < a.N * (a.P + 1); idx += blockDim.x) {
    const int c = idx / a.N, i = idx - c * a.N;
    const size_t o = (c * US + u) * a.N + i, o0 = u * a.N + i;
    if (c == 0) { a.j1p[o] = a.jp[o] * e0; a.j1m[o] = a.jm[o] * e0; }
    else {
      const double ec = a.e[c * US + u];
      a.j1p[o] = a.jp[o] * e0 + a.jp[o0] * ec;
      a.j1m[o] = a.jm[o] * e0 + a.jm[o0] * ec;
    }
  }
  __syncthreads();
  if (threadIdx.x <= a.P) {
    const int c = threadIdx.x;
    a.e[c * US + u] = (c == 0) ? e0 * e0 : 2.0 * e0 * a.e[c * US + u];
  }
}

__device__ __forceinline__ int stokes_comp(int i, int n, int strict) { return strict ? ((i + 1) % n) : (i % n) + 1; }

// apply_D! / apply_D_SFI! after the doublings (doubling.jl:93-118): r-+ rows of U, V change sign, r+- = D r-+ D,
// t-- = D t++ D, j0- rows of U, V change sign -- the same linear map on every component
struct SignArgs {
  int N, nS, U, P, strict;
  double *r_mp, *t_pp, *r_pm, *t_mm, *j0m;
};
__global__ void k_dsign(SignArgs a) {
  const size_t NN = (size_t)a.N * a.N, total = NN * a.U * (a.P + 1);
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e % a.N), j = (int)((e / a.N) % a.N);
  if (a.nS == 1) {
    a.r_pm[e] = a.r_mp[e];
    a.t_mm[e] = a.t_pp[e];
    return;
  }
  const int ci = stokes_comp(i, a.nS, a.strict), cj = stokes_comp(j, a.nS, a.strict);
  const double sg = (((ci <= 2) && (cj <= 2)) || ((ci > 2) && (cj > 2))) ? 1.0 : -1.0;
  double r = a.r_mp[e];
  if (ci > 2) r = -r;
  a.r_mp[e] = r;
  a.r_pm[e] = sg * r;
  a.t_mm[e] = sg * a.t_pp[e];
  if (j == 0 && ci > 2) {
    const size_t o = (e / NN) * a.N + i;
    a.j0m[o] = -a.j0m[o];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// elemental! (elemental.jl:109-162): get_elem_rt! (:164-207), get_elem_rt_SFI! (:209-253), apply_D_elemental! (:255-274)
// on Dual tau, varpi, Z: one thread per (unit, i, j) forms the value and, per partial, the exact derivative of the same
// expression (what a Dual carries through exp, * and /).  Also expk = exp(-dtau/mu0) (rt_kernel.jl:196).
// ---------------------------------------------------------------------------------------------------------------------
struct ElArgs {
  int N, nS, U, P, K, S, Nz, m, iz, nd, strict, imu0;
  size_t u0;
  double mu0;
  const double *mu, *wt;
  double I0[4], D[4];
  const double *tau, *varpi, *zw, *tau_sum;       // [S,Nz], [S,Nz], [K,S,Nz], [S,Nz+1]
  const double *dtau, *dvarpi, *dzw, *dtau_sum;   // the same shapes with P as the slowest axis (nullptr: zero)
  const double *Zpp, *Zmp, *dZpp, *dZmp;          // moment m: [N,N,K]; partials [N,N,K] at stride dZ_stride per partial
  size_t dZ_stride;
  double *r_mp, *t_pp, *r_pm, *t_mm, *j0p, *j0m, *e;
};

__global__ void __launch_bounds__(256) k_delemental(ElArgs a) {
#pragma clang fp contract(off)
  const int N = a.N, n = a.nS, P = a.P, K = a.K;
  const size_t NN = (size_t)N * N;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NN * a.U) return;
  const int i = (int)(idx % N), j = (int)((idx / N) % N);
  const size_t u = idx / NN, s = a.u0 + u, CS = (size_t)a.U * NN, VS = (size_t)a.U * N;
  const size_t SZ = (size_t)a.S * a.Nz, lz = s + (size_t)a.S * a.iz;
  const double sc = ldexp(1.0, -a.nd);
  const double d = a.tau[lz] * sc, w = a.varpi[lz];
  const double wdiv = (a.m == 0) ? 2.0 : 4.0, wct02 = (a.m == 0) ? 0.5 : 0.25;
  const double mui = a.mu[i], muj = a.mu[j], wj = a.wt[j] / wdiv, wi = a.wt[i] / wdiv;
  const size_t zo = i + (size_t)N * j;
  double Zp = 0.0, Zm = 0.0;
  for (int k = 0; k < K; ++k) {
    const double zk = a.zw[k + (size_t)K * lz];
    Zp += zk * a.Zpp[zo + NN * k];
    Zm += zk * a.Zmp[zo + NN * k];
  }
  // value-side factors: r = w Zm Ar(d), t = w Zp At(d) (off-diagonal) ...
  const double ei = exp(-d / mui), ej = exp(-d / muj);
  const bool live = wj > 1.e-8, diag = (i == j), same = (mui == muj);
  double Ar = 0.0, dAr = 0.0, At = 0.0, dAt = 0.0;
  if (live) {
    const double si = (1.0 / mui) + (1.0 / muj), E = exp(-d * si), F1 = muj / (mui + muj);
    Ar = F1 * wj * (1.0 - E);
    dAr = F1 * wj * si * E;
    if (!same) {
      const double F2 = muj / (mui - muj);
      At = F2 * wj * (ei - ej);
      dAt = F2 * wj * (-ei / mui + ej / muj);
    }
  }
  // the VALUE in the reference's own association (elemental.jl:180-200), no FMA contraction: a one-ulp difference here is amplified
  // by every later doubling (tests/helpers.py stokes_rtol), so the value follows the text literally like the fused kernels do
  double r0 = 0.0, t0 = 0.0;
  {
    if (live) {
      r0 = w * Zm * (muj / (mui + muj)) * wj * (1.0 - exp(-d * ((1.0 / mui) + (1.0 / muj))));
      if (!same) t0 = w * Zp * (muj / (mui - muj)) * wj * (ei - ej);
      else if (diag) t0 = ei * (1.0 + w * Zp * (d / mui) * wi);
    } else if (diag) t0 = ei;
  }
  const int ci = stokes_comp(i, n, a.strict), cj = stokes_comp(j, n, a.strict);
  const double sg = (((ci <= 2) && (cj <= 2)) || ((ci > 2) && (cj > 2))) ? 1.0 : -1.0;
  const double rsign = (a.nd >= 1 && ci > 2) ? -1.0 : 1.0;
  const size_t o = u * NN + zo;
  a.r_mp[o] = rsign * r0;
  a.t_pp[o] = t0;
  if (a.nd < 1) { a.r_pm[o] = sg * r0; a.t_mm[o] = sg * t0; }
  for (int p = 0; p < P; ++p) {
    const double dd = a.dtau ? a.dtau[lz + SZ * p] * sc : 0.0, dw = a.dvarpi ? a.dvarpi[lz + SZ * p] : 0.0;
    double dZp = 0.0, dZm = 0.0;
    for (int k = 0; k < K; ++k) {
      if (a.dzw) {
        const double dzk = a.dzw[k + (size_t)K * (lz + SZ * p)];
        dZp += dzk * a.Zpp[zo + NN * k];
        dZm += dzk * a.Zmp[zo + NN * k];
      }
      if (a.dZpp) {
        const double zk = a.zw[k + (size_t)K * lz];
        dZp += zk * a.dZpp[zo + NN * k + a.dZ_stride * p];
        dZm += zk * a.dZmp[zo + NN * k + a.dZ_stride * p];
      }
    }
    double r1 = 0.0, t1 = 0.0;
    if (live) {
      r1 = (dw * Zm + w * dZm) * Ar + w * Zm * dAr * dd;
      if (!same) t1 = (dw * Zp + w * dZp) * At + w * Zp * dAt * dd;
      else if (diag)
        t1 = -(dd / mui) * ei * (1.0 + w * Zp * (d / mui) * wi) + ei * (wi / mui) * ((dw * Zp + w * dZp) * d + w * Zp * dd);
    } else if (diag) t1 = -(dd / mui) * ei;
    const size_t oc = (size_t)(p + 1) * CS + o;
    a.r_mp[oc] = rsign * r1;
    a.t_pp[oc] = t1;
    if (a.nd < 1) { a.r_pm[oc] = sg * r1; a.t_mm[oc] = sg * t1; }
  }
  if (j != 0) return;
  // source vectors (get_elem_rt_SFI!, elemental.jl:209-253): one thread per (unit, i)
  const int i_start = n * (a.imu0 - 1), i_end = n * a.imu0;
  const double mus = a.mu[i_start];
  const bool insun = (i >= i_start && i < i_end);
  double gp, dgp, gm, dgm;  // j+ = wct02 w ZpI0 gp(d) att, j- likewise
  if (insun) {
    gp = (d / mui) * ei;
    dgp = (1.0 / mui) * ei * (1.0 - d / mui);
  } else {
    const double F = mus / (mui - mus), es = exp(-d / mus);
    gp = F * (ei - es);
    dgp = F * (-ei / mui + es / mus);
  }
  {
    const double Fm = mus / (mui + mus), sm_ = (1.0 / mui) + (1.0 / mus), E = exp(-d * sm_);
    gm = Fm * (1.0 - E);
    dgm = Fm * sm_ * E;
  }
  const size_t ts = s + (size_t)a.S * a.iz, TS = (size_t)a.S * (a.Nz + 1);
  const double att = exp(-a.tau_sum[ts] / mus);
  double ZpI = 0.0, ZmI = 0.0;
  for (int kk = 0; kk < n; ++kk) {
    double zp = 0.0, zm = 0.0;
    const size_t zc = i + (size_t)N * (i_start + kk);
    for (int k = 0; k < K; ++k) {
      const double zk = a.zw[k + (size_t)K * lz];
      zp += zk * a.Zpp[zc + NN * k];
      zm += zk * a.Zmp[zc + NN * k];
    }
    ZpI += zp * a.I0[kk];
    ZmI += zm * a.I0[kk];
  }
  const double Dm = (a.nd >= 1) ? a.D[i % n] : 1.0;
  const size_t ov = u * N + i;
  {
    // elemental.jl:226-246 in the reference's association
    double jp, jm;
    if (insun) jp = wct02 * w * ZpI * (d / mui) * exp(-d / mui);
    else jp = wct02 * w * ZpI * (mus / (mui - mus)) * (exp(-d / mui) - exp(-d / mus));
    jm = wct02 * w * ZmI * (mus / (mui + mus)) * (1.0 - exp(-d * ((1.0 / mui) + (1.0 / mus))));
    a.j0p[ov] = jp * att;
    a.j0m[ov] = Dm * (jm * att);
  }
  const double e0 = exp(-d / a.mu0);
  if (i == 0) a.e[u] = e0;
  for (int p = 0; p < P; ++p) {
    const double dd = a.dtau ? a.dtau[lz + SZ * p] * sc : 0.0, dw = a.dvarpi ? a.dvarpi[lz + SZ * p] : 0.0;
    const double datt = a.dtau_sum ? -(a.dtau_sum[ts + TS * p] / mus) * att : 0.0;
    double dZpI = 0.0, dZmI = 0.0;
    for (int kk = 0; kk < n; ++kk) {
      double zp = 0.0, zm = 0.0;
      const size_t zc = i + (size_t)N * (i_start + kk);
      for (int k = 0; k < K; ++k) {
        if (a.dzw) {
          const double dzk = a.dzw[k + (size_t)K * (lz + SZ * p)];
          zp += dzk * a.Zpp[zc + NN * k];
          zm += dzk * a.Zmp[zc + NN * k];
        }
        if (a.dZpp) {
          const double zk = a.zw[k + (size_t)K * lz];
          zp += zk * a.dZpp[zc + NN * k + a.dZ_stride * p];
          zm += zk * a.dZmp[zc + NN * k + a.dZ_stride * p];
        }
      }
      dZpI += zp * a.I0[kk];
      dZmI += zm * a.I0[kk];
    }
    const size_t oc = (size_t)(p + 1) * VS + ov;
    a.j0p[oc] = wct02 * (((dw * ZpI + w * dZpI) * gp + w * ZpI * dgp * dd) * att + w * ZpI * gp * datt);
    a.j0m[oc] = Dm * (wct02 * (((dw * ZmI + w * dZmI) * gm + w * ZmI * dgm * dd) * att + w * ZmI * gm * datt));
    if (i == 0) a.e[(size_t)(p + 1) * a.U + u] = -(dd / a.mu0) * e0;
  }
}

// d tau_sum[s, z + 1] = d tau_sum[s, z] + d tau[s, z] (compEffectiveLayerProperties.jl:108 on Duals)
__global__ void k_dtausum(int S, int Nz, int P, const double *dtau, double *dts) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)S * P) return;
  const size_t s = idx % S, p = idx / S;
  double acc = 0.0;
  dts[s + (size_t)S * (Nz + 1) * p] = 0.0;
  for (int z = 0; z < Nz; ++z) {
    acc += dtau[s + (size_t)S * z + (size_t)S * Nz * p];
    dts[s + (size_t)S * (z + 1) + (size_t)S * (Nz + 1) * p] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// create_surface_layer! on Duals: LambertianSurfaceScalar (lambertian_surface.jl:20-75), BRDF matrices (rpv_surface.jl:
// 20-66), LambertianSurfaceLegendre (lambertian_surface.jl:77-138) -- written into the added-layer arrays
// ---------------------------------------------------------------------------------------------------------------------
struct SurfArgs {
  int N, nS, U, P, S, Nz, m, kind, imu0;
  size_t u0;
  double mu0, albedo;
  double I0[4];
  const double *mu, *wt;
  const double *dalbedo;                 // [P] device (kind 0)
  const double *Rsurf, *dRsurf;          // moment m: [N,N]; partial p at + dR_stride p (kind 1)
  size_t dR_stride;
  const double *alb_spec, *dalb_spec;    // [S], [S,P] (kind 2)
  const double *tau_sum, *dtau_sum;      // [S,Nz+1](,P)
  double *r_mp, *t_pp, *r_pm, *t_mm, *j0p, *j0m;
};

__global__ void __launch_bounds__(256) k_dsurface(SurfArgs a) {
  const int N = a.N, n = a.nS, P = a.P;
  const size_t NN = (size_t)N * N;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NN * a.U) return;
  const int i = (int)(idx % N), j = (int)((idx / N) % N);
  const size_t u = idx / NN, s = a.u0 + u, CS = (size_t)a.U * NN, VS = (size_t)a.U * N;
  const size_t ts = s + (size_t)a.S * a.Nz, TS = (size_t)a.S * (a.Nz + 1);
  const bool lamb0 = (i % n == 0) && (j % n == 0);
  const bool active = (a.kind == 1) || (a.m == 0);
  const double eyev = (a.kind == 2 && a.m > 0) ? 0.0 : (i == j ? 1.0 : 0.0);  // Legendre: t = 0 for m > 0 (:131-132)
  const double mw = a.mu[j] * a.wt[j];
  const int i_start = n * (a.imu0 - 1);
  const double att = exp(-a.tau_sum[ts] / a.mu0);
  for (int c = 0; c <= P; ++c) {
    const int p = c - 1;
    // R_surf[i, j] of this component
    double R = 0.0;
    if (active) {
      if (a.kind == 1) R = (c == 0) ? a.Rsurf[i + (size_t)N * j] : (a.dRsurf ? a.dRsurf[i + (size_t)N * j + a.dR_stride * p] : 0.0);
      else if (a.kind == 0) R = lamb0 ? 2.0 * ((c == 0) ? a.albedo : (a.dalbedo ? a.dalbedo[p] : 0.0)) : 0.0;
      else R = lamb0 ? 2.0 * ((c == 0) ? a.alb_spec[s] : (a.dalb_spec ? a.dalb_spec[s + (size_t)a.S * p] : 0.0)) : 0.0;
    }
    const size_t o = c * CS + idx;
    a.r_mp[o] = R * mw;
    a.r_pm[o] = 0.0;
    a.t_pp[o] = (c == 0) ? eyev : 0.0;
    a.t_mm[o] = (c == 0) ? eyev : 0.0;
    if (j == 0) {
      double RI = 0.0, RI0 = 0.0;  // (R_surf I0N)[i] of this component and of the value
      double I0i = 0.0;
      if (active) {
        for (int kk = 0; kk < n; ++kk) {
          const int jj = i_start + kk;
          const bool l0 = (i % n == 0) && (jj % n == 0);
          double Rv, Rc;
          if (a.kind == 1) {
            Rv = a.Rsurf[i + (size_t)N * jj];
            Rc = (c == 0) ? Rv : (a.dRsurf ? a.dRsurf[i + (size_t)N * jj + a.dR_stride * p] : 0.0);
          } else if (a.kind == 0) {
            Rv = l0 ? 2.0 * a.albedo : 0.0;
            Rc = (c == 0) ? Rv : (l0 && a.dalbedo ? 2.0 * a.dalbedo[p] : 0.0);
          } else {
            Rv = l0 ? 2.0 * a.alb_spec[s] : 0.0;
            Rc = (c == 0) ? Rv : (l0 && a.dalb_spec ? 2.0 * a.dalb_spec[s + (size_t)a.S * p] : 0.0);
          }
          RI += Rc * a.I0[kk];
          RI0 += Rv * a.I0[kk];
        }
        if (i >= i_start && i < i_start + n && a.kind != 2) I0i = a.I0[i - i_start];  // Legendre: j0+ = 0 (:112)
      }
      const double datt = (c > 0 && a.dtau_sum) ? -(a.dtau_sum[ts + TS * p] / a.mu0) * att : 0.0;
      const size_t ov = c * VS + u * N + i;
      if (c == 0) {
        a.j0p[ov] = I0i * att;
        a.j0m[ov] = a.mu0 * RI * att;
      } else {
        a.j0p[ov] = I0i * datt;
        a.j0m[ov] = a.mu0 * (RI * att + RI0 * datt);
      }
    }
  }
}

// postprocessing_vza! (postprocessing_vza.jl:9-60, SFI branch) on Duals: R_SFI += bigCS J0-, T_SFI += bigCS J0+
struct PostArgs {
  int N, nS, U, P, S, nVza, m, M;
  size_t u0;
  const int *node;
  const double *cos_mphi, *sin_mphi, *J0p, *J0m, *hdrJ;   // hdrJ: r-+_surf J0+ + j0-_surf (interaction_hdrf!, interaction_hdrf.jl:9-45)
  double *R, *T, *dR, *dT;  // [nVza,nS,S], [nVza,nS,S,P]
  double *hdr, *dhdr;       // postprocessing_vza_hdrf! (postprocessing_vza.jl:63-93)
};
__global__ void k_dpost(PostArgs a) {
  const size_t per = (size_t)a.nVza * a.nS, total = per * a.U * (a.P + 1);
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int v = (int)(idx % a.nVza), k = (int)((idx / a.nVza) % a.nS);
  const size_t u = (idx / per) % a.U, c = idx / (per * a.U);
  const double weight = (a.m == 0) ? 0.5 : 1.0;
  const double cs = weight * ((k < 2) ? a.cos_mphi[v + (size_t)a.nVza * a.m] : a.sin_mphi[v + (size_t)a.nVza * a.m]);
  const size_t row = (size_t)(a.node[v] - 1) * a.nS + k, o = (c * a.U + u) * a.N + row;
  const size_t out = v + (size_t)a.nVza * (k + (size_t)a.nS * (a.u0 + u));
  double *R = (c == 0) ? a.R : a.dR + per * a.S * (c - 1), *T = (c == 0) ? a.T : a.dT + per * a.S * (c - 1);
  double *H = (c == 0) ? a.hdr : a.dhdr + per * a.S * (c - 1);
  if (a.m == 0) { R[out] = cs * a.J0m[o]; T[out] = cs * a.J0p[o]; H[out] = cs * a.hdrJ[o]; }
  else { R[out] += cs * a.J0m[o]; T[out] += cs * a.J0p[o]; H[out] += cs * a.hdrJ[o]; }
}

// the up- and down-welling flux sums of the BHR (interaction_hdrf.jl:27-41, m = 0) on Duals: one thread per (unit, component, Stokes i)
struct BhrArgs {
  int N, nS, U, P, S, imu0;
  size_t u0;
  const double *mu, *wt, *hdrJ, *J0p, *surf_j0p;
  double *uw, *dw, *duw, *ddw;   // [nS,S], [nS,S,P]
};
__global__ void k_dbhr(BhrArgs a) {
  const size_t total = (size_t)a.nS * a.U * (a.P + 1);
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)(idx % a.nS);
  const size_t u = (idx / a.nS) % a.U, c = idx / ((size_t)a.nS * a.U);
  const size_t o = (c * a.U + u) * a.N;
  double up = 0.0, dn = 0.0;
  for (int r = i; r < a.N; r += a.nS) {
    const double wq = a.wt[r] * a.mu[r];
    up += a.hdrJ[o + r] * wq;
    dn += a.J0p[o + r] * wq;
  }
  const int i0 = a.nS * (a.imu0 - 1);
  dn += a.surf_j0p[o + i0] * a.mu[i0];   // the direct beam: component 0 of the sun block for every i, as written (:38-40)
  const size_t out = i + (size_t)a.nS * (a.u0 + u);
  if (c == 0) { a.uw[out] = up; a.dw[out] = dn; }
  else { a.duw[out + (size_t)a.nS * a.S * (c - 1)] = up; a.ddw[out + (size_t)a.nS * a.S * (c - 1)] = dn; }
}

}  // namespace momd

// =====================================================================================================================
// host side
// =====================================================================================================================
#define DCHK(x)                                                                                        \
  do {                                                                                                 \
    const hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) {                                                                            \
      if (err) *err = std::string(#x) + ": " + hipGetErrorString(e_);                                  \
      return 2;                                                                                        \
    }                                                                                                  \
  } while (0)

namespace {
struct Layer { double *r_mp, *r_pm, *t_pp, *t_mm, *jp, *jm; };   // added layer (DMs / DVs)
struct Comp { double *R_mp, *R_pm, *T_pp, *T_mm, *Jp, *Jm; };    // composite layer
}  // namespace

size_t momd_bytes_per_unit(int N, int P) {
  return ((size_t)14 * N * N + (size_t)14 * N + 1) * (P + 1) * sizeof(double);
}

int momd_run(const MomDualScene &sc, std::string *err) {
  using namespace momd;
  const int N = sc.N, P = sc.P, Nz = sc.Nz;
  const size_t NN = (size_t)N * N;
  if (N > 128) {
    if (err) *err = "mom_rt_run_dual: operator edge N > 128 is not supported (the inverse runs in one CU's LDS)";
    return 1;
  }
  hipStream_t st = sc.stream;
  // chunk of units that fits the workspace
  const size_t per_unit = momd_bytes_per_unit(N, P);
  size_t Uc = std::min<size_t>(std::min<size_t>((size_t)sc.S, 65535), std::max<size_t>(1, sc.work_budget / per_unit));  // grid.y <= 65535
  const size_t need = Uc * per_unit + 256;
  if (*sc.work_cap < need) {
    if (*sc.work) DCHK(hipFree(*sc.work));
    *sc.work = nullptr;
    *sc.work_cap = 0;
    DCHK(hipMalloc(sc.work, need));
    *sc.work_cap = need;
  }
  double *dts = nullptr;  // d tau_sum [S, Nz+1, P]
  if (P > 0 && sc.dtau) {
    dts = sc.dtau_sum_buf;
    const size_t n = (size_t)sc.S * P;
    hipLaunchKernelGGL(k_dtausum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sc.S, Nz, P, sc.dtau, dts);
  }
  const int nbk = (N <= 64) ? 1 : 2;
  const size_t inv_lds = ((size_t)3 * 64 * nbk + 2) * sizeof(double) + (size_t)(3 * 64 * nbk + 2) * sizeof(int);
  DCHK(mom_allow_lds((const void *)k_dinv<1>, inv_lds));
  DCHK(mom_allow_lds((const void *)k_dinv<2>, inv_lds));

  for (size_t u0 = 0; u0 < (size_t)sc.S; u0 += Uc) {
    const int U = (int)std::min<size_t>(Uc, (size_t)sc.S - u0);
    const size_t MS = (size_t)(P + 1) * U * NN, VS = (size_t)(P + 1) * U * N;
    double *base = (double *)*sc.work;
    auto mat = [&]() { double *p = base; base += MS; return p; };
    auto vec = [&]() { double *p = base; base += VS; return p; };
    Layer ad{mat(), mat(), mat(), mat(), vec(), vec()};
    Comp co{mat(), mat(), mat(), mat(), vec(), vec()};
    double *W = mat(), *G = mat(), *TG = mat(), *X = mat(), *Y = mat(), *t2 = mat();
    double *j1p = vec(), *j1m = vec(), *va = vec(), *vb = vec(), *vn1 = vec(), *vn2 = vec();
    double *ek = base;  // [(1+P)][U]
    base += (size_t)(P + 1) * U;

    const int tiles = (N + TM - 1) / TM;
    struct Ride { int nq; const double *x[2], *add[2]; double *y[2]; };
    const Ride no_ride{0, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    auto gemm_r = [&](double *C, const double *A, const double *B, int mode, int c0, int nc, double alpha, const double *E,
                      double beta, double eye, const Ride &rd) {
      if (nc <= 0) return;
      const int tiles_j = (N + rd.nq + TN - 1) / TN;
      GemmArgs g{N, U, c0, nc, mode, tiles, tiles_j, A, B, E, C, alpha, beta, eye, rd.nq, {rd.x[0], rd.x[1]}, {rd.add[0], rd.add[1]},
                 {rd.y[0], rd.y[1]}};
      if (N + rd.nq <= 16) {   // one wavefront per (unit, component), operands straight from global memory (k_dgemm_w)
        g.tiles_i = 1; g.tiles_j = 1;
        const dim3 gw((unsigned)(((size_t)U * nc + 3) / 4));
        if (rd.nq) hipLaunchKernelGGL((k_dgemm_w<1, true>), gw, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_dgemm_w<1, false>), gw, dim3(256), 0, st, g);
        return;
      }
      if (N + rd.nq <= 96) {
        // ONE workgroup tile of 16 nw x 16 nw, nw wavefronts: 32 / 48 (faster than wavefront tiles of that size: N = 30 / 42: 461 / 953 ->
        // 351 / 673 ms), 64, and 80 / 96 instead of 2 x 2 tiles of 64 (N = 72 / 80 / 96: 500 / 579 / 684 -> 362 / 452 / 656 ms)
        const int nw = (N + rd.nq + 15) / 16;
        g.tiles_i = 1; g.tiles_j = 1;
        const dim3 grid1((unsigned)(8 * nc * ((U + 7) / 8)));
#define MOMD_GO(V2, RD)                                                                                      \
  do {                                                                                                       \
    switch (nw) {                                                                                            \
      case 2: hipLaunchKernelGGL((k_dgemm<V2, RD, 2>), grid1, dim3(128), 0, st, g); break;                   \
      case 3: hipLaunchKernelGGL((k_dgemm<V2, RD, 3>), grid1, dim3(192), 0, st, g); break;                   \
      case 4: hipLaunchKernelGGL((k_dgemm<V2, RD, 4>), grid1, dim3(256), 0, st, g); break;                   \
      case 5: hipLaunchKernelGGL((k_dgemm<V2, RD, 5>), grid1, dim3(320), 0, st, g); break;                   \
      default: hipLaunchKernelGGL((k_dgemm<V2, RD, 6>), grid1, dim3(384), 0, st, g); break;                  \
    }                                                                                                        \
  } while (0)
        if (N % 2 == 0) { if (rd.nq) MOMD_GO(true, true); else MOMD_GO(true, false); }
        else { if (rd.nq) MOMD_GO(false, true); else MOMD_GO(false, false); }
#undef MOMD_GO
        return;
      }
      const dim3 grid((unsigned)(8 * tiles * tiles_j * nc * ((U + 7) / 8)));
      if (N % 2 == 0) {
        if (rd.nq) hipLaunchKernelGGL((k_dgemm<true, true, 4>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_dgemm<true, false, 4>), grid, dim3(256), 0, st, g);
      } else {
        if (rd.nq) hipLaunchKernelGGL((k_dgemm<false, true, 4>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_dgemm<false, false, 4>), grid, dim3(256), 0, st, g);
      }
    };
    auto gemm = [&](double *C, const double *A, const double *B, int mode, int c0, int nc, double alpha, const double *E,
                    double beta, double eye) { gemm_r(C, A, B, mode, c0, nc, alpha, E, beta, eye, no_ride); };
    // C = A B on Duals with y[q] = add[q] + A x[q] riding along
    auto dual_ride = [&](double *C, const double *A, const double *B, const double *E, const Ride &rd) {
      gemm_r(C, A, B, 0, 0, P + 1, 1.0, E, E ? 1.0 : 0.0, 0.0, rd);
    };
    auto dual = [&](double *C, const double *A, const double *B) { gemm(C, A, B, 0, 0, P + 1, 1.0, nullptr, 0.0, 0.0); };
    auto dual_add = [&](double *C, const double *A, const double *B, const double *E) { gemm(C, A, B, 0, 0, P + 1, 1.0, E, 1.0, 0.0); };
    // OUT = A (I - W)^-1 on Duals (batch_inv! + batched_mul, gpu_batched.jl:100-150): G0 = (I - W0)^-1 by Gauss-Jordan, OUT0 = A0 G0,
    // and with dG = G dW G the partials are OUT_c = A_c G0 + A0 G0 W_c G0 = (A_c + OUT0 W_c) G0: two products per partial
    auto times_inv = [&](double *OUT, const double *A, const double *Wm) {
      InvArgs ia{N, U, Wm, G, 1.0, -1.0, sc.info};
      if (N <= 16) hipLaunchKernelGGL(k_dinv_w<16>, dim3((unsigned)((U + 3) / 4)), dim3(256), 0, st, ia);
      else if (N <= 32) hipLaunchKernelGGL(k_dinv_w<32>, dim3((unsigned)((U + 3) / 4)), dim3(256), 0, st, ia);
      else if (nbk == 1) hipLaunchKernelGGL(k_dinv<1>, dim3((unsigned)U), dim3(256), inv_lds, st, ia);
      else hipLaunchKernelGGL(k_dinv<2>, dim3((unsigned)U), dim3(256), inv_lds, st, ia);
      gemm(OUT, A, G, 0, 0, 1, 1.0, nullptr, 0.0, 0.0);
      gemm(Y, OUT, Wm, 1, 1, P, 1.0, A, 1.0, 0.0);
      gemm(OUT, Y, G, 2, 1, P, 1.0, nullptr, 0.0, 0.0);
    };
    auto matvec = [&](const double *M, int nq, const double *x0, const double *a0, double *y0, const double *x1, const double *a1,
                      double *y1) {
      MvArgs mv{N, U, P, nq, M, {x0, x1}, {a0, a1}, {y0, y1}};
      hipLaunchKernelGGL(k_dmatvec, dim3((unsigned)(8 * (P + 1) * ((U + 7) / 8))), dim3(256), 0, st, mv);
    };
    const unsigned eblocks = (unsigned)((NN * U + 255) / 256);

    // interaction_helper! for the four interfaces (interaction.jl:8-22, 27-43, 49-64, 69-117) on Duals
    auto interaction = [&](int iface) {
      if (iface == 0) {
        matvec(ad.t_pp, 1, co.Jp, ad.jp, vn1, nullptr, nullptr, nullptr);     // J0+ = j0+ + t++ J0+
        matvec(co.T_mm, 1, ad.jm, co.Jm, vn2, nullptr, nullptr, nullptr);     // J0- = J0- + T-- j0-
        std::swap(co.Jp, vn1); std::swap(co.Jm, vn2);
        dual(t2, ad.t_mm, co.T_mm); std::swap(co.T_mm, t2);                   // T-- = t-- T--
        dual(t2, ad.t_pp, co.T_pp); std::swap(co.T_pp, t2);                   // T++ = t++ T++
      } else if (iface == 1) {
        matvec(ad.r_mp, 1, co.Jp, ad.jm, va, nullptr, nullptr, nullptr);      // r-+ J0+ + j0-
        matvec(co.T_mm, 1, va, co.Jm, vn2, nullptr, nullptr, nullptr);        // J0- += T-- (.)
        matvec(ad.t_pp, 1, co.Jp, ad.jp, vn1, nullptr, nullptr, nullptr);     // J0+ = j0+ + t++ J0+
        std::swap(co.Jp, vn1); std::swap(co.Jm, vn2);
        dual(X, co.T_mm, ad.r_mp); dual(co.R_mp, X, co.T_pp);                 // R-+ = (T-- r-+) T++
        (void)hipMemcpyAsync(co.R_pm, ad.r_pm, MS * sizeof(double), hipMemcpyDeviceToDevice, st);  // R+- = r+-
        dual(t2, ad.t_pp, co.T_pp); std::swap(co.T_pp, t2);                   // T++ = t++ T++
        dual(t2, co.T_mm, ad.t_mm); std::swap(co.T_mm, t2);                   // T-- = T-- t--
      } else if (iface == 2) {
        matvec(co.R_pm, 1, ad.jm, co.Jp, va, nullptr, nullptr, nullptr);      // J0+ + R+- j0-
        matvec(ad.t_pp, 1, va, ad.jp, vn1, nullptr, nullptr, nullptr);        // J0+ = j0+ + t++ (.)
        matvec(co.T_mm, 1, ad.jm, co.Jm, vn2, nullptr, nullptr, nullptr);     // J0- += T-- j0-
        std::swap(co.Jp, vn1); std::swap(co.Jm, vn2);
        dual(X, ad.t_pp, co.R_pm); dual(co.R_pm, X, ad.t_mm);                 // R+- = (t++ R+-) t--
        dual(t2, ad.t_pp, co.T_pp); std::swap(co.T_pp, t2);
        dual(t2, co.T_mm, ad.t_mm); std::swap(co.T_mm, t2);
      } else {
        dual_ride(W, ad.r_mp, co.R_pm, nullptr, Ride{1, {co.Jp, nullptr}, {ad.jm, nullptr}, {va, nullptr}});  // r-+ R+- (:76-79); riding: r-+ J0+ + j0-
        times_inv(TG, co.T_mm, W);                                            // T01_inv = T-- (I - r-+ R+-)^-1
        dual_ride(X, TG, ad.r_mp, nullptr, Ride{1, {va, nullptr}, {co.Jm, nullptr}, {vn2, nullptr}});       // riding: J0- += T01_inv (.) (:82)
        dual_add(co.R_mp, X, co.T_pp, co.R_mp);                               // R-+ += (T01_inv r-+) T++      (:86)
        dual(co.T_mm, TG, ad.t_mm);                                           // T-- = T01_inv t--             (:89)
        dual_ride(W, co.R_pm, ad.r_mp, nullptr, Ride{1, {ad.jm, nullptr}, {co.Jp, nullptr}, {va, nullptr}});  // R+- r-+ (:93); riding: J0+ + R+- j0-
        times_inv(TG, ad.t_pp, W);                                            // T21_inv = t++ (I - R+- r-+)^-1
        dual_ride(t2, TG, co.T_pp, nullptr, Ride{1, {va, nullptr}, {ad.jp, nullptr}, {vn1, nullptr}});      // T++ = T21_inv T++ (:104); riding: J0+ = j0+ + T21_inv (.) (:100)
        std::swap(co.T_pp, t2);
        std::swap(co.Jp, vn1); std::swap(co.Jm, vn2);
        dual(X, TG, co.R_pm); dual_add(co.R_pm, X, ad.t_mm, ad.r_pm);         // R+- = r+- + (T21_inv R+-) t-- (:107)
      }
    };

    for (int m = 0; m < sc.M; ++m) {
      const size_t zoff = NN * sc.K * m;
      for (int iz = 0; iz < Nz; ++iz) {
        const int nd = sc.nd[iz];
        ElArgs ea{};
        ea.N = N; ea.nS = sc.nS; ea.U = U; ea.P = P; ea.K = sc.K; ea.S = sc.S; ea.Nz = Nz; ea.m = m; ea.iz = iz; ea.nd = nd;
        ea.strict = sc.strict; ea.imu0 = sc.imu0; ea.u0 = u0; ea.mu0 = sc.mu0; ea.mu = sc.mu; ea.wt = sc.wt;
        for (int k = 0; k < 4; ++k) { ea.I0[k] = sc.I0[k]; ea.D[k] = sc.D[k]; }
        ea.tau = sc.tau; ea.varpi = sc.varpi; ea.zw = sc.zw; ea.tau_sum = sc.tau_sum;
        ea.dtau = sc.dtau; ea.dvarpi = sc.dvarpi; ea.dzw = sc.dzw; ea.dtau_sum = dts;
        ea.Zpp = sc.Zpp + zoff; ea.Zmp = sc.Zmp + zoff;
        ea.dZpp = sc.dZpp ? sc.dZpp + zoff : nullptr; ea.dZmp = sc.dZmp ? sc.dZmp + zoff : nullptr;
        ea.dZ_stride = NN * sc.K * sc.M;
        ea.r_mp = ad.r_mp; ea.t_pp = ad.t_pp; ea.r_pm = ad.r_pm; ea.t_mm = ad.t_mm; ea.j0p = ad.jp; ea.j0m = ad.jm; ea.e = ek;
        hipLaunchKernelGGL(k_delemental, dim3(eblocks), dim3(256), 0, st, ea);
        // doubling_helper! (doubling.jl:43-68) on Duals
        for (int it = 0; it < nd; ++it) {
          ScaleArgs sa{N, U, P, ad.jp, ad.jm, j1p, j1m, ek};
          hipLaunchKernelGGL(k_dscale, dim3((unsigned)U), dim3(128), 0, st, sa);  // j1+-, expk^2            (:51-54, :62)
          dual_ride(W, ad.r_mp, ad.r_mp, nullptr, Ride{2, {ad.jp, j1m}, {j1m, ad.jp}, {va, vb}});  // r-+ r-+ (:46); riding: j1- + r-+ j0+, j0+ + r-+ j1-
          times_inv(TG, ad.t_pp, W);                                          // tt++_gp_refl = t++ (I - r-+ r-+)^-1 (:46-47)
          dual_ride(X, TG, ad.r_mp, nullptr, Ride{2, {va, vb}, {ad.jm, j1p}, {vn2, vn1}});  // TG r-+; riding: j0- += TG (.), j0+ = j1+ + TG (.) (:57-60)
          std::swap(ad.jm, vn2); std::swap(ad.jp, vn1);
          dual_add(ad.r_mp, X, ad.t_pp, ad.r_mp);                             // r-+ += (TG r-+) t++            (:65)
          dual(t2, TG, ad.t_pp); std::swap(ad.t_pp, t2);                      // t++ = TG t++                   (:68)
        }
        if (nd > 0) {
          SignArgs sg{N, sc.nS, U, P, sc.strict, ad.r_mp, ad.t_pp, ad.r_pm, ad.t_mm, ad.jm};
          hipLaunchKernelGGL(k_dsign, dim3((unsigned)((MS + 255) / 256)), dim3(256), 0, st, sg);
        }
        if (iz == 0) {  // rt_kernel.jl:213-220: the first layer IS the composite layer
          std::swap(co.R_mp, ad.r_mp); std::swap(co.R_pm, ad.r_pm); std::swap(co.T_pp, ad.t_pp); std::swap(co.T_mm, ad.t_mm);
          std::swap(co.Jp, ad.jp); std::swap(co.Jm, ad.jm);
        } else {
          interaction(sc.iface[iz]);
        }
      }
      SurfArgs sa{};
      sa.N = N; sa.nS = sc.nS; sa.U = U; sa.P = P; sa.S = sc.S; sa.Nz = Nz; sa.m = m; sa.kind = sc.surf_kind; sa.imu0 = sc.imu0;
      sa.u0 = u0; sa.mu0 = sc.mu0; sa.albedo = sc.albedo;
      for (int k = 0; k < 4; ++k) sa.I0[k] = sc.I0[k];
      sa.mu = sc.mu; sa.wt = sc.wt; sa.dalbedo = sc.dalbedo;
      sa.Rsurf = sc.Rsurf ? sc.Rsurf + NN * m : nullptr; sa.dRsurf = sc.dRsurf ? sc.dRsurf + NN * m : nullptr;
      sa.dR_stride = NN * sc.M;
      sa.alb_spec = sc.albedo_spec; sa.dalb_spec = sc.dalbedo_spec; sa.tau_sum = sc.tau_sum; sa.dtau_sum = dts;
      sa.r_mp = ad.r_mp; sa.t_pp = ad.t_pp; sa.r_pm = ad.r_pm; sa.t_mm = ad.t_mm; sa.j0p = ad.jp; sa.j0m = ad.jm;
      hipLaunchKernelGGL(k_dsurface, dim3(eblocks), dim3(256), 0, st, sa);
      interaction(sc.iface[Nz - 1]);  // rt_run.jl:198-200: the LAST layer's interface code
      matvec(ad.r_mp, 1, co.Jp, ad.jm, va, nullptr, nullptr, nullptr);   // hdr_J0- = r-+_surf J0+ + j0-_surf (interaction_hdrf.jl:20-24)
      if (m == 0) {
        BhrArgs ba{N, sc.nS, U, P, sc.S, sc.imu0, u0, sc.mu, sc.wt, va, co.Jp, ad.jp, sc.bhr_uw, sc.bhr_dw, sc.dbhr_uw, sc.dbhr_dw};
        const size_t nb_ = (size_t)sc.nS * U * (P + 1);
        hipLaunchKernelGGL(k_dbhr, dim3((unsigned)((nb_ + 255) / 256)), dim3(256), 0, st, ba);
      }
      PostArgs pa{N, sc.nS, U, P, sc.S, sc.nVza, m, sc.M, u0, sc.node, sc.cos_mphi, sc.sin_mphi, co.Jp, co.Jm, va, sc.R, sc.T, sc.dR, sc.dT,
                  sc.hdr, sc.dhdr};
      const size_t tot = (size_t)sc.nVza * sc.nS * U * (P + 1);
      hipLaunchKernelGGL(k_dpost, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, pa);
    }
  }
  DCHK(hipGetLastError());
  return 0;
}
