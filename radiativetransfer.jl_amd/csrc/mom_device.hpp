// mom_device.hpp -- workgroup-level building blocks of the MI355X Matrix-Operator core.
//
// Execution model: ONE WORKGROUP (4 wavefronts of 64 lanes) owns ONE spectral point of ONE
// Fourier moment.  Its N x N operators (N = nStokes*Nquad) live in LDS (N <= 64, "LDS mode")
// or in a per-workgroup global scratch slab (N > 64, "generic mode"); all dense products run
// on the FP64 matrix cores (v_mfma_f64_16x16x4_f64, 16x16 output tile per wave, K step 4);
// (I - R r)^-1 is a pivoted Gauss-Jordan inverse held in the same memory (later rounds:
// blocked).  Matrices are column-major with leading dimension ld (ld % 32 in {2,30} keeps the
// MFMA B-operand reads bank-conflict free and the A-operand reads 2-way).
//
// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md section 3): lane l holds
//   A[row = l & 15][k = l >> 4],  B[k = l >> 4][col = l & 15],
//   C/D reg r: [row = (l >> 4) + 4 r][col = l & 15].
#pragma once
#include <hip/hip_runtime.h>

namespace mom {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTJ = 4;  // column tiles processed together by one wave (shares the A operand)

// ---------------------------------------------------------------------------------------
// C(i,j) <- epi(i, j, sum_k A(i,k) B(k,j)) for an N x N product.
// A(i,k), B(k,j): element functors (any memory space; must return 0 outside [0,N)).
// SYNC: all waves finish reading their operands before any wave stores (allows the output
// to alias an operand).  Requires at most one work item per wave, i.e. N <= 64.
// ---------------------------------------------------------------------------------------
template <bool SYNC, class FA, class FB, class FE>
__device__ __forceinline__ void wg_gemm(int N, FA A, FB B, FE epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int Tn = (N + 15) >> 4;
  const int Cg = (Tn + kTJ - 1) / kTJ;
  const int items = Tn * Cg;
  const int ksteps = (N + 3) >> 2;
  for (int item = wave; item < (SYNC ? kWaves : items); item += kWaves) {
    const bool have = item < items;
    const int ti = item % Tn, cg = item / Tn;
    d4 acc[kTJ];
#pragma unroll
    for (int t = 0; t < kTJ; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (have) {
      const int row = 16 * ti + lr;
#pragma unroll 4
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = 4 * ks + lq;
        const double a = A(row, k);
#pragma unroll
        for (int t = 0; t < kTJ; ++t) {
          const double b = B(k, 16 * (cg * kTJ + t) + lr);
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    if (SYNC) __syncthreads();
    if (have) {
#pragma unroll
      for (int t = 0; t < kTJ; ++t) {
        const int col = 16 * (cg * kTJ + t) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rw = 16 * ti + lq + 4 * r;
          if (rw < N && col < N) epi(rw, col, acc[t][r]);
        }
      }
    }
  }
}

// Two products sharing the B operand: C1 = A1*B, C2 = A2*B (e.g. r += (A r) t and t = A t).
template <bool SYNC, class FA1, class FA2, class FB, class FE1, class FE2>
__device__ __forceinline__ void wg_gemm2(int N, FA1 A1, FA2 A2, FB B, FE1 epi1, FE2 epi2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int Tn = (N + 15) >> 4;
  const int Cg = (Tn + kTJ - 1) / kTJ;
  const int items = Tn * Cg;
  const int ksteps = (N + 3) >> 2;
  for (int item = wave; item < (SYNC ? kWaves : items); item += kWaves) {
    const bool have = item < items;
    const int ti = item % Tn, cg = item / Tn;
    d4 acc1[kTJ], acc2[kTJ];
#pragma unroll
    for (int t = 0; t < kTJ; ++t) { acc1[t] = (d4){0.0, 0.0, 0.0, 0.0}; acc2[t] = (d4){0.0, 0.0, 0.0, 0.0}; }
    if (have) {
      const int row = 16 * ti + lr;
#pragma unroll 2
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = 4 * ks + lq;
        const double a1 = A1(row, k);
        const double a2 = A2(row, k);
#pragma unroll
        for (int t = 0; t < kTJ; ++t) {
          const double b = B(k, 16 * (cg * kTJ + t) + lr);
          acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1[t], 0, 0, 0);
          acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc2[t], 0, 0, 0);
        }
      }
    }
    if (SYNC) __syncthreads();
    if (have) {
#pragma unroll
      for (int t = 0; t < kTJ; ++t) {
        const int col = 16 * (cg * kTJ + t) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rw = 16 * ti + lq + 4 * r;
          if (rw < N && col < N) { epi1(rw, col, acc1[t][r]); epi2(rw, col, acc2[t][r]); }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// y1 = M x1, y2 = M x2 (one pass over M).  M(i,k) functor; x1,x2,y1,y2 in LDS; part: LDS
// scratch of 8*ldv doubles.  All threads must call.  y may alias x.
// ---------------------------------------------------------------------------------------
template <class FM>
__device__ __forceinline__ void wg_matvec2(int N, int ldv, FM M, const double *x1, const double *x2, double *y1,
                                           double *y2, double *part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = (N + kWaves - 1) / kWaves;
  const int k0 = wave * chunk, k1 = min(N, k0 + chunk);
  for (int i = lane; i < N; i += 64) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = k0; k < k1; ++k) {
      const double m = M(i, k);
      s1 += m * x1[k];
      s2 += m * x2[k];
    }
    part[wave * ldv + i] = s1;
    part[(kWaves + wave) * ldv + i] = s2;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += kThreads) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { s1 += part[w * ldv + i]; s2 += part[(kWaves + w) * ldv + i]; }
    y1[i] = s1;
    y2[i] = s2;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// In-place inverse of the N x N matrix a (column-major, ld) by Gauss-Jordan elimination
// with partial (row) pivoting -- the arithmetic counterpart of the reference's batch_inv!
// (gpu_batched.jl:36-87: getrf + getri).  prow/pcol/rowk: LDS vectors (>= N), ipiv: LDS ints
// (>= N), sh: LDS int.  *bad (LDS int) is set nonzero if a zero pivot was met.
// All threads must call; ends with a barrier.  3 barriers per elimination step.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_inverse(int N, double *a, int ld, double *prow, double *pcol, double *rowk,
                                           int *ipiv, int *sh, int *bad) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < N; ++k) {
    // (1) pivot search on column k, rows k..N-1 (first maximum, like idamax)
    if (wave == 0) {
      double best = -1.0;
      int bi = k;
      for (int i = k + lane; i < N; i += 64) {
        const double v = fabs(a[i + k * ld]);
        if (v > best) { best = v; bi = i; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) {
        sh[0] = bi;
        ipiv[k] = bi;
        if (!(best > 0.0)) *bad = k + 1;
      }
    }
    __syncthreads();
    const int p = sh[0];
    const double d = 1.0 / a[p + k * ld];
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
    for (int e = tid; e < N * N; e += kThreads) {
      const int j = e / N, i = e - j * N;
      double v;
      if (i == k) {
        v = prow[j];
      } else {
        const double f = pcol[i];
        const double aij = (i == p) ? rowk[j] : a[i + j * ld];
        v = (j == k) ? (-f * d) : (aij - f * prow[j]);
      }
      a[i + j * ld] = v;
    }
    __syncthreads();
  }
  // undo the row interchanges: columns swapped in reverse order
  for (int k = N - 1; k >= 0; --k) {
    const int p = ipiv[k];
    if (p != k) {
      for (int i = tid; i < N; i += kThreads) {
        const double x = a[i + k * ld], y = a[i + p * ld];
        a[i + k * ld] = y;
        a[i + p * ld] = x;
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

// copy N x N column-major block src(ld_s) -> dst(ld_d), all threads
__device__ __forceinline__ void wg_copy_mat(int N, const double *__restrict__ src, int ld_s, double *__restrict__ dst,
                                            int ld_d) {
  for (int e = threadIdx.x; e < N * N; e += kThreads) {
    const int j = e / N, i = e - j * N;
    dst[i + j * ld_d] = src[i + j * ld_s];
  }
}

}  // namespace mom
