// mom_device.hpp -- workgroup-level building blocks of the MI355X Matrix-Operator core.
//
// Execution model: ONE WORKGROUP (kWaves wavefronts of 64 lanes) owns ONE spectral point of
// ONE Fourier moment.  Its N x N operators (N = nStokes*Nquad) live in LDS (N <= 64, "LDS
// mode") or in a per-workgroup global scratch slab (N > 64, "generic mode").  All dense
// products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: 16x16 output tile per wave,
// K step 4, 64 cycles per instruction per SIMD -> 77 TFLOP/s measured on MI355X).
//
// Buffers are column-major with leading dimension ld = Np + 2 and Np = 16*ceil(N/16) columns;
// rows/columns >= N are ZERO and stay zero (every store is guarded), so the MFMA operand loads
// need no bounds checks.  ld % 32 in {2, 18} keeps the B-operand ds_read_b64 conflict-free
// and the A-operand reads 2-way.
//
// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md section 3): lane l holds
//   A[row = l & 15][k = l >> 4],  B[k = l >> 4][col = l & 15],
//   C/D reg r: [row = (l >> 4) + 4 r][col = l & 15].
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

// The whole device library is compiled once per workgroup shape: MOM_WAVES wavefronts per workgroup inside
// namespace MOM_NS (momcore.hip: 8 waves, namespace mom; momcore_w4.hip: 4 waves, namespace mom4).
#ifndef MOM_NS
#define MOM_NS mom
#endif

namespace MOM_NS {

// The device library is compiled once per scalar type as well: MOM_REAL = double (default; every build but one) or
// float (momcore_f32.hip, namespace momf: the reference's float_type = Float32, parameters_from_yaml.jl:160).
#ifndef MOM_REAL
#define MOM_REAL double
#endif
typedef MOM_REAL real;
typedef real r4 __attribute__((ext_vector_type(4)));
constexpr bool kF64 = sizeof(real) == 8;

// one 16 x 16 x 4 MFMA step in the build's scalar type.  C/D layout: f64: row = (lane >> 4) + 4 reg; f32: row =
// 4 (lane >> 4) + reg (cdna_hip_programming.md section 3); column = lane & 15 in both.
typedef double mom_d4 __attribute__((ext_vector_type(4)));
typedef float mom_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mom_d4 mma16(double a, double b, mom_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ mom_f4 mma16(float a, float b, mom_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int cd_row(int lq, int r) { return kF64 ? lq + 4 * r : 4 * lq + r; }

extern __shared__ real mom_smem[];  // the workgroup's dynamic LDS image (carved by make_ctx)

#ifndef MOM_WAVES
#define MOM_WAVES 8
#endif
constexpr int kWaves = MOM_WAVES;
constexpr int kThreads = 64 * kWaves;
#ifdef MOM_TJ
constexpr int kTJ = MOM_TJ;  // 4-wave build: 3 tiles per item (operators up to 48 x 48: one item per wave)
#else
constexpr int kTJ = 16 / kWaves;  // column tiles per work item: 8 waves x 2 tiles cover N <= 64
#endif

// Global-memory pointers with their address space spelled out.  Pointers that reach a kernel inside an argument
// struct are generic to the compiler: it would emit FLAT loads/stores, which count on BOTH memory counters, so
// every LDS wait behind a composite store in flight would also wait for that store.  Through these types the
// accesses become global_load / global_store (vmcnt only).
typedef __attribute__((address_space(1))) real gdouble;  // (the name predates the float build)
__device__ __forceinline__ gdouble *as_global(real *p) { return (gdouble *)p; }
__device__ __forceinline__ const gdouble *as_global(const real *p) { return (const gdouble *)p; }

// Thread coordinates through an opaque asm: the address arithmetic derived from them is recomputed where it is
// used (a few integer ops) instead of being hoisted out of the per-unit loop of the fused kernels, kept live
// across its thousands of instructions and spilled -- a scratch reload costs an `s_waitcnt vmcnt(0)`, i.e. it
// also waits for every global store in flight.
__device__ __forceinline__ int wg_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ int wg_lane() { return wg_tid() & 63; }
__device__ __forceinline__ int wg_wave() { return __builtin_amdgcn_readfirstlane(wg_tid() >> 6); }

// e -> (i = e % N, j = e / N) without an integer division (valid for e*N < 2^32)
struct FastDiv {
  unsigned magic;
  int N;
  __device__ __forceinline__ void init(int n) { N = n; magic = (unsigned)((0x100000000ull + (unsigned)n - 1u) / (unsigned)n); }
  __device__ __forceinline__ void split(int e, int &i, int &j) const {
    j = (N == 1) ? e : (int)__umulhi((unsigned)e, magic);  // n = 1: the magic number does not fit 32 bits
    i = e - j * N;
  }
};

// ---------------------------------------------------------------------------------------
// Work item of one wave: NA products sharing the B operand over a 16 x (16*kTJ) output strip.
// Two K schedules:
//  * straight (57 <= N <= 64, i.e. 15 or 16 MFMA k-steps, LDS operands): branch-free, fully
//    unrolled stream, so the LDS reads run ahead of the MFMAs (one exposed LDS latency per product
//    instead of one per chunk) -- measured 2.16 us vs 2.82 us per 60x60 product;
//  * chunked: 4 k-steps of loads, then their MFMAs (any N; generic mode).
// ---------------------------------------------------------------------------------------
// A-operand bundles: NA functors behind one interface (u selects the product)
template <class FA>
struct One {
  FA a;
  __device__ __forceinline__ real operator()(int, int i, int k) const { return a(i, k); }
  template <int LD> __device__ __forceinline__ real at(int, int i, int k) const { return a.template at<LD>(i, k); }
  __device__ __forceinline__ void launder() { a.launder(); }
};
template <class FA1, class FA2>
struct Two {
  FA1 a1;
  FA2 a2;
  __device__ __forceinline__ real operator()(int u, int i, int k) const { return u == 0 ? a1(i, k) : a2(i, k); }
  template <int LD> __device__ __forceinline__ real at(int u, int i, int k) const {
    return u == 0 ? a1.template at<LD>(i, k) : a2.template at<LD>(i, k);
  }
  __device__ __forceinline__ void launder() { a1.launder(); a2.launder(); }
};

// branch-free, fully unrolled item with a COMPILE-TIME leading dimension: every LDS read is
// base + immediate offset, and the (laundered) bases cannot be hoisted out of the caller's loops,
// so neither address arithmetic nor long live ranges cost registers.
template <int NA, int KS, int LD, class FAs, class FB>
__device__ __forceinline__ void item_straight(int row, int lq, int col0, FAs A, FB B, r4 (&acc)[NA][kTJ]) {
  // launder the lane coordinates (NOT the pointers: an asm operand would turn the LDS pointers
  // into generic ones and the reads into flat loads)
  asm volatile("" : "+v"(row), "+v"(lq), "+v"(col0));
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 4 * ks + lq;
    real a[NA], b[kTJ];
#pragma unroll
    for (int u = 0; u < NA; ++u) a[u] = A.template at<LD>(u, row, k);
#pragma unroll
    for (int t = 0; t < kTJ; ++t) b[t] = B.template at<LD>(k, col0 + 16 * t);
#pragma unroll
    for (int t = 0; t < kTJ; ++t)
#pragma unroll
      for (int u = 0; u < NA; ++u) acc[u][t] = mma16(a[u], b[t], acc[u][t]);
  }
}

// nt: number of column tiles of this item that lie inside the padded matrix (wave-uniform); tiles beyond it
// are neither loaded nor multiplied
template <int NA, int C, bool PRED, class FAs, class FB>
__device__ __forceinline__ void kchunk(int ks0, int row, int lq, int col0, int nt, const FAs &A, FB B,
                                       r4 (&acc)[NA][kTJ]) {
  real a[NA][C], b[C][kTJ];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = 4 * (ks0 + c) + lq;
#pragma unroll
    for (int u = 0; u < NA; ++u) a[u][c] = A(u, row, k);
#pragma unroll
    for (int t = 0; t < kTJ; ++t)
      if (!PRED || t < nt) b[c][t] = B(k, col0 + 16 * t);
  }
#pragma unroll
  for (int t = 0; t < kTJ; ++t)
    if (!PRED || t < nt) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int u = 0; u < NA; ++u)
          acc[u][t] = mma16(a[u][c], b[c][t], acc[u][t]);
    }
}

// ---------------------------------------------------------------------------------------
// C_u(i,j) <- epi(u, i, j, sum_k A(u,i,k) B(k,j)), u < NA, for i < N, j < NC.
// A(u,i,k), B(k,j): element functors valid on the whole padded index range (zero where the K
// range is padded).  NC = N: plain products; NC in (N, Np]: columns N..NC-1 of the B operand
// (kept in the buffer's padding) ride along -- vectors get multiplied by A for free.
// SYNC: all waves finish reading their operands before any wave stores (an output may alias an
// operand).  Requires at most one work item per wave, i.e. N <= 64.
// ---------------------------------------------------------------------------------------
// operand functors that read LDS with cheap addressing opt in to the fast16 schedule
template <class F> struct lds_operand { static constexpr bool value = false; };

template <int NA, bool SYNC, bool FAST, class FAs, class FB, class FE>
__device__ __forceinline__ void wg_gemm_n(int N, int NC, FAs A, FB B, FE epi) {
  const int lane = wg_lane(), wave = wg_wave();
  const int lr = lane & 15, lq = lane >> 4;
  const int Tn = (N + 15) >> 4;
  const int Cg = (Tn + kTJ - 1) / kTJ;
  const int items = Tn * Cg;
  const int ksteps = (N + 3) >> 2;
  for (int item = wave; item < (SYNC ? kWaves : items); item += kWaves) {
    const bool have = item < items;
    const int ti = item % Tn, cg = item / Tn;
    r4 acc[NA][kTJ];
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
      for (int t = 0; t < kTJ; ++t) acc[u][t] = (r4){0.0, 0.0, 0.0, 0.0};
    if (have) {
      const int row = 16 * ti + lr, col0 = 16 * cg * kTJ + lr;
      bool done = false;
#ifndef MOM_NO_STRAIGHT
      if constexpr (FAST) {
        if (ksteps == 15) { item_straight<NA, 15, 66>(row, lq, col0, A, B, acc); done = true; }
        else if (ksteps == 16) { item_straight<NA, 16, 66>(row, lq, col0, A, B, acc); done = true; }
      }
#endif
      if (!done) {
        const int nt = min(kTJ, Tn - cg * kTJ);
        constexpr int CH = (NA * kTJ > 4) ? 2 : 4;  // k-steps per chunk, bounded by the register budget
        auto run = [&](auto pred) {
          constexpr bool PR = decltype(pred)::value;
          int ks = 0;
          for (; ks + 4 <= ksteps; ks += 4) {
            if constexpr (CH == 4) {
              kchunk<NA, 4, PR>(ks, row, lq, col0, nt, A, B, acc);
            } else {
              kchunk<NA, 2, PR>(ks, row, lq, col0, nt, A, B, acc);
              kchunk<NA, 2, PR>(ks + 2, row, lq, col0, nt, A, B, acc);
            }
          }
          const int rem = ksteps - ks;
          if (rem == 3) kchunk<NA, 3, PR>(ks, row, lq, col0, nt, A, B, acc);
          else if (rem == 2) kchunk<NA, 2, PR>(ks, row, lq, col0, nt, A, B, acc);
          else if (rem == 1) kchunk<NA, 1, PR>(ks, row, lq, col0, nt, A, B, acc);
        };
        if (nt == kTJ) run(std::false_type{});  // all column tiles of the item exist: no predicates
        else run(std::true_type{});
      }
    }
    if (SYNC) __syncthreads();
    if (have) {
#pragma unroll
      for (int t = 0; t < kTJ; ++t) {
        const int col = 16 * (cg * kTJ + t) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rw = 16 * ti + cd_row(lq, r);
          if (rw < N && col < NC) {
#pragma unroll
            for (int u = 0; u < NA; ++u) epi(u, rw, col, acc[u][t][r]);
          }
        }
      }
    }
  }
}

// Large operators (generic mode, N > 64): operands streamed through LDS in k panels, see wg_gemm_big in mom_kernels.hpp.
// BIG is true only in the generic-mode (LDSM = false) instantiations, so the LDS-resident kernels carry none of it.
template <class FA, class FB, class FE>
__device__ void wg_gemm_big(int N, int NC, FA A, FB B, FE epi);

template <bool SYNC, bool BIG = false, class FA, class FB, class FE>
__device__ __forceinline__ void wg_gemm_nc(int N, int NC, FA A, FB B, FE epi) {
  if constexpr (BIG && !SYNC && kWaves == 8) {
    if (N > 64 && N <= 256) {
      wg_gemm_big(N, NC, A, B, epi);
      return;
    }
  }
  wg_gemm_n<1, SYNC, lds_operand<FA>::value && lds_operand<FB>::value>(
      N, NC, One<FA>{A}, B, [=](int, int i, int j, real v) { epi(i, j, v); });
}
template <bool SYNC, bool BIG = false, class FA, class FB, class FE>
__device__ __forceinline__ void wg_gemm(int N, FA A, FB B, FE epi) {
  wg_gemm_nc<SYNC, BIG>(N, N, A, B, epi);
}
// Two products sharing the B operand: C1 = A1*B, C2 = A2*B (r += (A r) t and t = A t).
template <bool SYNC, bool BIG = false, class FA1, class FA2, class FB, class FE1, class FE2>
__device__ __forceinline__ void wg_gemm2(int N, FA1 A1, FA2 A2, FB B, FE1 epi1, FE2 epi2) {
  if constexpr (BIG && !SYNC && kWaves == 8) {
    if (N > 64 && N <= 256) {  // two passes over B: the two accumulator sets of the shared-operand form do not fit next to 64 x 64 wave tiles
      wg_gemm_big(N, N, A1, B, epi1);
      wg_gemm_big(N, N, A2, B, epi2);
      return;
    }
  }
  wg_gemm_n<2, SYNC, lds_operand<FA1>::value && lds_operand<FA2>::value && lds_operand<FB>::value>(
      N, N, Two<FA1, FA2>{A1, A2}, B,
      [=](int u, int i, int j, real v) { if (u == 0) epi1(i, j, v); else epi2(i, j, v); });
}

// ---------------------------------------------------------------------------------------
// y1 = M x1, y2 = M x2 (one pass over M).  M(i,k) functor; x1,x2,y1,y2 in LDS; part: LDS
// scratch of 2*kWaves*ldv doubles.  All threads must call.  y may alias x.
// ---------------------------------------------------------------------------------------
template <class FM>
__device__ __forceinline__ void wg_matvec2(int N, int ldv, FM M, const real *x1, const real *x2, real *y1,
                                           real *y2, real *part) {
  const int lane = wg_lane(), wave = wg_wave();
  const int chunk = (N + kWaves - 1) / kWaves;
  const int k0 = wave * chunk, k1 = min(N, k0 + chunk);
  for (int i = lane; i < N; i += 64) {
    real s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
      const real m = M(i, k);
      s1 += m * x1[k];
      s2 += m * x2[k];
    }
    part[wave * ldv + i] = s1;
    part[(kWaves + wave) * ldv + i] = s2;
  }
  __syncthreads();
  for (int i = wg_tid(); i < N; i += kThreads) {
    real s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { s1 += part[w * ldv + i]; s2 += part[(kWaves + w) * ldv + i]; }
    y1[i] = s1;
    y2[i] = s2;
  }
  __syncthreads();
}

// max over the wave (all lanes get it)
__device__ __forceinline__ real wave_max(real v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ real wave_sum(real v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------
// In-place inverse of the N x N matrix a (column-major, ld) by Gauss-Jordan elimination
// with partial (row) pivoting -- the arithmetic counterpart of the reference's batch_inv!
// (gpu_batched.jl:36-87: getrf + getri).  prow/pcol/rowk: LDS vectors (>= N), ipiv: LDS ints
// (>= N), sh: LDS int.  *bad (LDS int) is set nonzero if a zero pivot was met.
// All threads must call; ends with a barrier.  3 barriers per elimination step.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_inverse(int N, const FastDiv &fd, real *a, int ld, real *prow, real *pcol,
                                           real *rowk, int *ipiv, int *sh, int *bad) {
  const int tid = wg_tid(), lane = tid & 63, wave = tid >> 6;
  const int NN = N * N;
  for (int k = 0; k < N; ++k) {
    // (1) pivot search on column k, rows k..N-1 (first maximum, like idamax)
    if (wave == 0) {
      real best = -1.0;
      int bi = N;
      for (int i = k + lane; i < N; i += 64) {
        const real v = fabs(a[i + k * ld]);
        if (v > best) { best = v; bi = i; }
      }
      const real wm = wave_max(best);
      // lowest row index attaining the maximum
      int cand = (best == wm) ? bi : N;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
      if (lane == 0) {
        sh[0] = cand;
        ipiv[k] = cand;
        if (!(wm > 0.0)) *bad = k + 1;
      }
    }
    __syncthreads();
    const int p = sh[0];
    const real d = 1.0 / a[p + k * ld];
    // (2) scaled pivot row -> prow, old row k -> rowk, pivot column (as after the swap) -> pcol
    for (int j = tid; j < 2 * N; j += kThreads) {
      if (j < N) {
        rowk[j] = a[k + j * ld];
        prow[j] = (j == k) ? d : a[p + j * ld] * d;
      } else {
        const int i = j - N;
        pcol[i] = (i == k) ? 0.0 : ((i == p) ? a[k + k * ld] : a[i + k * ld]);
      }
    }
    __syncthreads();
    // (3) rank-1 update of every row but k (row p takes the old row k: the interchange);
    //     row k <- scaled pivot row
    for (int e0 = tid; e0 < NN; e0 += 4 * kThreads) {
      real av[4], fv[4], pv[4];
      int ii[4], jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * kThreads;
        if (e < NN) {
          fd.split(e, ii[u], jj[u]);
          av[u] = a[ii[u] + jj[u] * ld];
          fv[u] = pcol[ii[u]];
          pv[u] = prow[jj[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * kThreads;
        if (e < NN) {
          const int i = ii[u], j = jj[u];
          real v;
          if (i == k) {
            v = pv[u];
          } else {
            const real aij = (i == p) ? rowk[j] : av[u];
            v = (j == k) ? (-fv[u] * d) : (aij - fv[u] * pv[u]);
          }
          a[i + j * ld] = v;
        }
      }
    }
    __syncthreads();
  }
  // undo the row interchanges: columns swapped in reverse order
  for (int k = N - 1; k >= 0; --k) {
    const int p = ipiv[k];
    if (p != k) {
      for (int i = tid; i < N; i += kThreads) {
        const real x = a[i + k * ld], y = a[i + p * ld];
        a[i + k * ld] = y;
        a[i + p * ld] = x;
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Register-resident Gauss-Jordan inverse for N <= 64 (one matrix row per lane).
// Wave w holds columns [CW*w, CW*w + CW) of the matrix, lane = row: CW = 64/kWaves registers per
// lane.  Partial pivoting is IMPLICIT: the pivot of step k is the largest |a[i][k]| among the rows
// not yet used as pivots; rows are never moved, the permutation is undone when the result is
// written back (inv(A)[k][p_j] = S[p_k][j]).  Per elimination step: one wave-level max + ballot in
// the wave that owns column k, ONE workgroup barrier (pivot column / index / reciprocal travel
// through double-buffered LDS slots), CW readlane broadcasts of the pivot row and CW FMAs per lane.
// a: column-major buffer (ld) holding the matrix on entry and the inverse on exit.
// pcol: LDS, >= 2*64 doubles; shd: LDS, >= 4 doubles; ipiv: LDS ints >= 64; bad: LDS int.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_f64(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// sign/exponent/leading mantissa bits of |v| as an ordered integer (the pivot search key of wg_inverse_reg)
__device__ __forceinline__ int abs_key(double v) { return __double2hiint(fabs(v)); }
__device__ __forceinline__ int abs_key(float v) { return __float_as_int(fabsf(v)); }

__device__ __forceinline__ void wg_inverse_reg(int N, real *a, int ld, real *pcol, real *shd, int *ipiv,
                                               int *bad) {
  constexpr int CW = 64 / kWaves;
  const int lane = wg_lane(), wave = wg_wave();
  real v[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int col = CW * wave + c;
    v[c] = (lane < N && col < N) ? a[lane + col * ld] : 0.0;
  }
  bool used = false;
  int myk = 0;
  const int npanel = (N + CW - 1) / CW;
  for (int kp = 0; kp < npanel; ++kp) {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int k = CW * kp + c;
      if (k < N) {  // wave-uniform
        const int slot = k & 1;
        if (wave == kp) {
          // pivot = first row whose |a| has the largest HIGH WORD (sign/exponent/20 mantissa bits):
          // within 2^-20 of the true maximum, which is all partial pivoting needs, and a 32-bit
          // integer wave reduction instead of a 64-bit one
          const int ah = (!used && lane < N) ? abs_key(v[c]) : -1;
          int mh = ah;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) mh = max(mh, __shfl_xor(mh, off));
          unsigned long long mk = __ballot(ah == mh);
          int p = __ffsll((long long)mk) - 1;
          const real m = fabs(readlane_f64(v[c], p));
          const real piv = readlane_f64(v[c], p);
          const real d = 1.0 / piv;
          pcol[slot * 64 + lane] = v[c];
          if (lane == 0) {
            shd[2 * slot] = d;
            shd[2 * slot + 1] = (real)p;
            ipiv[k] = p;
            if (!(m > 0.0)) *bad = k + 1;
          }
        }
        __syncthreads();
        const real d = shd[2 * slot];
        const int p = __builtin_amdgcn_readfirstlane((int)shd[2 * slot + 1]);
        const real f = pcol[slot * 64 + lane];
        const bool isp = (lane == p);
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
          const real prow = readlane_f64(v[cc], p) * d;
          v[cc] = isp ? prow : (v[cc] - f * prow);
        }
        if (wave == kp) v[c] = isp ? d : (-f * d);
        if (isp) { used = true; myk = k; }
      }
    }
  }
  __syncthreads();  // ipiv complete; everyone is done reading the matrix buffer's old contents
  if (lane < N) {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int col = CW * wave + c;
      if (col < N) a[myk + ipiv[col] * ld] = v[c];
    }
  }
  __syncthreads();
}

// copy an N x N column-major block src(ld_s) -> dst(ld_d), all threads; loads batched by 4
template <class PS, class PD>  // pointer types: LDS/generic real* or global gdouble*
__device__ __forceinline__ void wg_copy_mat(int N, const FastDiv &fd, PS src, int ld_s, PD dst, int ld_d) {
  const int NN = N * N;
  for (int e0 = wg_tid(); e0 < NN; e0 += 4 * kThreads) {
    real v[4];
    int o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * kThreads;
      if (e < NN) {
        int i, j;
        fd.split(e, i, j);
        v[u] = src[i + j * ld_s];
        o[u] = i + j * ld_d;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * kThreads < NN) dst[o[u]] = v[u];
  }
}

}  // namespace MOM_NS
