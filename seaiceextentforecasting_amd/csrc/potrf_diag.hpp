// Diagonal-block kernel of the blocked Cholesky (north/June1st.py:265 np.linalg.cholesky -> dpotrf):
// factor one 128x128 SPD block in LDS, write L11 back, then invert it in place and write inv(L11) to the
// workspace so that the panel solve  L21 = A21 inv(L11)^T  is a plain MFMA GEMM.
//
// One 256-thread workgroup (the step is a latency chain, not throughput work).  The block lives in LDS
// ([128][130] doubles: pitch 130 keeps the 8-byte MFMA fragment reads conflict free).  It is processed in
// 16-column steps:
//   B1  16x16 diagonal factorisation by ONE wave, rows in registers, pivots broadcast with v_readlane
//       (wavefront shuffles, no LDS round trips, no barriers inside the step);
//   B2  16-wide triangular solve of the rows below, one thread per row, L11 broadcast from LDS;
//   B3  rank-16 update of the remaining lower tiles on v_mfma_f64_16x16x4_f64.
// The inverse is formed with the in-place blocked lower-triangular recurrence (LAPACK dtrtri shape,
// last block column first), again with MFMA for the block products.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"

namespace sigp {

constexpr int DB = 128;   // diagonal block size
constexpr int DP = 130;   // LDS pitch (doubles)
constexpr int DIAG_LDS_BYTES = (DB * DP + DB + 8 * 16 * 16) * (int)sizeof(double);

__device__ inline double readlane_f64(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// A: the 128x128 block inside the big matrix (row stride lda); Linv: [128][128] row-major workspace;
// info: device word, first failing 1-based global pivot index (0 = none yet); pivot_base: global index of row 0.
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ A, long lda, double* __restrict__ Linv,
                                                         int* __restrict__ info, int pivot_base) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* S = smem;                 // [128][130]
  double* dinv = smem + DB * DP;    // [128] reciprocals of the diagonal of L
  double* XD = dinv + DB;           // [8][16][16] inverses of the 16x16 diagonal blocks

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;

  // ---- load the block (full 128x128; only the lower triangle is meaningful) ----
  for (int idx = tid; idx < DB * (DB / 2); idx += 256) {
    int row = idx >> 6, cp = (idx & 63) * 2;
    *(d2*)(S + row * DP + cp) = *(const d2*)(A + (long)row * lda + cp);
  }
  __syncthreads();

  // ---- factorisation, 8 steps of 16 columns ----
  for (int jb = 0; jb < 8; ++jb) {
    const int o = jb * 16;
    if (wave == 0) {
      // B1: 16x16 Cholesky; lane i (mod 16) holds row i
      double r[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) r[c] = S[(o + lr) * DP + o + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double dj = readlane_f64(r[j], j);
        if (!(dj > 0.0)) {   // non-positive or NaN pivot: LAPACK info = index of the failing pivot
          if (lane == 0 && *info == 0) *info = pivot_base + o + j + 1;
          dj = 1.0;
        }
        const double s = sqrt(dj);
        const double inv = 1.0 / s;
        const double lij = (lr == j) ? s : r[j] * inv;
        r[j] = lij;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
          const double lcj = readlane_f64(lij, c);
          r[c] -= lij * lcj;
        }
        if (lane == 0) dinv[o + j] = inv;
      }
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[(o + lr) * DP + o + c] = (c <= lr) ? r[c] : 0.0;
      }
    }
    __syncthreads();
    // B2: rows below: x L11^T = a   (one thread per row)
    const int nrows = DB - (o + 16);
    if (tid < nrows) {
      const int row = o + 16 + tid;
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = S[row * DP + o + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x[j] *= dinv[o + j];
#pragma unroll
        for (int p = j + 1; p < 16; ++p) x[p] -= x[j] * S[(o + p) * DP + o + j];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) S[row * DP + o + c] = x[c];
    }
    __syncthreads();
    // B3: rank-16 update of the lower tiles of the trailing (7-jb)x(7-jb) block grid
    const int nb = 7 - jb;
    const int nt = nb * (nb + 1) / 2;
    for (int t = wave; t < nt; t += 4) {
      int ti = 0, rem = t;
      while (rem > ti) { rem -= ti + 1; ++ti; }   // row ti has ti+1 tiles
      const int tj = rem;
      const int rowb = o + 16 + ti * 16, colb = o + 16 + tj * 16;
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = S[(rowb + lq + 4 * r) * DP + colb + lr];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double a = -S[(rowb + lr) * DP + o + kk * 4 + lq];
        const double b = S[(colb + lr) * DP + o + kk * 4 + lq];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(rowb + lq + 4 * r) * DP + colb + lr] = acc[r];
    }
    __syncthreads();
  }

  // ---- write L11 back (lower, zeros above the diagonal) ----
  for (int idx = tid; idx < DB * (DB / 2); idx += 256) {
    int row = idx >> 6, cp = (idx & 63) * 2;
    d2 v = *(const d2*)(S + row * DP + cp);
    if (cp > row) v.x = 0.0;
    if (cp + 1 > row) v.y = 0.0;
    *(d2*)(A + (long)row * lda + cp) = v;
  }

  // ---- inverse: 16x16 diagonal-block inverses (columns of inv(L_bb) by forward substitution) ----
  if (wave < 2) {
    const int blk = wave * 4 + lq, c = lr, o = blk * 16;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] *= dinv[o + i];
#pragma unroll
      for (int p = i + 1; p < 16; ++p) x[p] -= S[(o + p) * DP + o + i] * x[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) XD[(blk * 16 + i) * 16 + c] = x[i];
  }
  __syncthreads();

  // ---- blocked in-place inverse, last block column first:  X[ib][jb] = -(sum_p X[ib][p] L[p][jb]) X[jb][jb] ----
  for (int jb = 7; jb >= 0; --jb) {
    const int o = jb * 16;
    const int nb = 7 - jb;
    d4 t0 = d4{0, 0, 0, 0}, t1 = d4{0, 0, 0, 0};
    // each wave owns block rows ib = jb+1+wave and jb+1+wave+4 (if present)
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int q = wave + 4 * slot;
      if (q < nb) {
        const int ib = jb + 1 + q;
        d4 acc = d4{0, 0, 0, 0};
        for (int p = jb + 1; p <= ib; ++p) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double a = S[(ib * 16 + lr) * DP + p * 16 + kk * 4 + lq];    // X[ib][p] (row lr, k)
            const double b = S[(p * 16 + kk * 4 + lq) * DP + o + lr];           // L[p][jb] (k, col lr)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
          }
        }
        if (slot == 0) t0 = acc; else t1 = acc;
      }
    }
    __syncthreads();   // every read of the original L[.][jb] is done
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int q = wave + 4 * slot;
      if (q < nb) {
        const int ib = jb + 1 + q;
        d4 acc = slot == 0 ? t0 : t1;
        // park T in the (wave-private) destination block so it can be re-read in A-operand layout
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ib * 16 + lq + 4 * r) * DP + o + lr] = acc[r];
        d4 u = d4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double a = -S[(ib * 16 + lr) * DP + o + kk * 4 + lq];
          const double b = XD[(jb * 16 + kk * 4 + lq) * 16 + lr];
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, u, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ib * 16 + lq + 4 * r) * DP + o + lr] = u[r];
      }
    }
    // diagonal block of the inverse
    {
      const int i = tid >> 4, c = tid & 15;   // 256 threads = 16x16
      S[(o + i) * DP + o + c] = XD[(jb * 16 + i) * 16 + c];
    }
    __syncthreads();
  }

  // ---- write inv(L11) (lower, zeros above) ----
  for (int idx = tid; idx < DB * (DB / 2); idx += 256) {
    int row = idx >> 6, cp = (idx & 63) * 2;
    d2 v = *(const d2*)(S + row * DP + cp);
    if (cp > row) v.x = 0.0;
    if (cp + 1 > row) v.y = 0.0;
    *(d2*)(Linv + row * DB + cp) = v;
  }
}

}  // namespace sigp
