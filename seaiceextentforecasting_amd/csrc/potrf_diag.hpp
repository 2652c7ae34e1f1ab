// Diagonal-block kernel of the blocked Cholesky (north/June1st.py:265 np.linalg.cholesky -> dpotrf):
// factor one 128x128 SPD block in LDS, write L11 back, then invert it in place and write inv(L11) to the
// workspace so that the panel solve  L21 = A21 inv(L11)^T  is a plain MFMA GEMM.
//
// One 512-thread workgroup (the step is a latency chain, not throughput work).  The block lives in LDS
// ([128][130] doubles: pitch 130 keeps the 8-byte MFMA fragment reads conflict free).  It is processed in
// 16-column steps:
//   B1  16x16 diagonal factorisation by ONE wave, rows in registers, pivots broadcast with v_readlane
//       (wavefront shuffles, no LDS round trips, no barriers inside the step);
//   B2  16-wide triangular solve of the rows below, one thread per row, L11 broadcast from LDS;
//   B3  rank-16 update of the remaining lower tiles on v_mfma_f64_16x16x4_f64 -- with look-ahead: wave 0 takes the next
//       diagonal tile first and runs B1 of the next step while the other waves finish the update, so the serial pivot chain
//       of step jb+1 hides the bulk of B3 of step jb.
// Register budget: the kernel must stay at <= 128 VGPRs (121 now).  Its 8 waves then take 2 x 128 of a SIMD's 512
// registers and fit beside ONE resident wave of the trailing-update kernel (256 VGPRs); a 220-VGPR build (measured with a
// DPP row_newbcast pivot loop, 7 % faster on an idle GPU) has to wait for BOTH update workgroups of a CU to finish:
// 1.06 ms instead of 0.23 ms per launch in lockstep batches, -2 % fits/s.  Check `.vgpr_count` after touching it.
// The inverse is formed with the in-place blocked lower-triangular recurrence (LAPACK dtrtri shape,
// last block column first), again with MFMA for the block products.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"

namespace sigp {

constexpr int DB = 128;   // diagonal block size
constexpr int BP = 18;    // pitch (doubles) inside a 16x16 LDS block: conflict-free 8-byte MFMA fragment reads
constexpr int BSZ = 16 * BP;
// LDS holds only the 36 lower 16x16 blocks (block-packed) + the reciprocal diagonal: 84 KB, so the kernel can
// share a CU with one resident trailing-update workgroup (64 KB) instead of waiting for a whole CU to drain.
constexpr int DIAG_LDS_BYTES = (36 * BSZ + DB) * (int)sizeof(double);   // fp64 size; fp32 needs half
constexpr int DIAG_THREADS = 512;

__device__ inline double readlane_t(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ inline float readlane_t(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ inline double rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }
__device__ inline float rsq_seed(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline int dblk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BSZ; }   // bj <= bi

// A: the 128x128 block inside the big matrix (row stride lda); Linv: [128][128] row-major workspace whose
// strictly-upper part is zero (zeroed once at allocation, never written here);
// info: device word, first failing 1-based global pivot index (0 = none yet); pivot_base: global index of row 0.
// blockIdx.x = batch member: A += b*strideA, Linv += b*strideL, info += b.
template <typename T>
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(T* __restrict__ A, long lda, T* __restrict__ Linv,
                                                                  int* __restrict__ info, int pivot_base, int skip_in, long strideA,
                                                                  long strideL) {
  const int skip = skip_in & (DBG_MASK | 32);   // bits 1..16: phase ablations of tools/diag_bench.py (debug library only)
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v2_t v2_t;
  // latency chain on the critical path of every panel, co-resident with MFMA-saturating update waves: ask the
  // instruction arbiter for the highest wave priority (bit 32 of `skip` disables it, for A/B timing)
  if (!(skip & 32)) __builtin_amdgcn_s_setprio(3);
  A += blockIdx.x * strideA;
  Linv += blockIdx.x * strideL;
  info += blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* S = (T*)smem_raw;               // 36 blocks of [16][18]
  T* dinv = S + 36 * BSZ;            // [128] reciprocals of the diagonal of L

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;

  // ---- load the lower triangle of the block ----
#pragma unroll
  for (int it = 0; it < DB * (DB / 2) / DIAG_THREADS; ++it) {
    const int idx = tid + it * DIAG_THREADS;
    const int row = idx >> 6, cp = (idx & 63) * 2;
    if (cp <= row) *(v2_t*)(S + dblk(row >> 4, cp >> 4) + (row & 15) * BP + (cp & 15)) = *(const v2_t*)(A + (long)row * lda + cp);
  }
  __syncthreads();

  // ---- factorisation, 8 steps of 16 columns ----
  // B1: 16x16 Cholesky of the diagonal tile jb by ONE wave; lane i (mod 16) holds row i; pivots via v_readlane; 1/sqrt by
  // v_rsq_f64 + Goldschmidt
  auto factor16 = [&](int jb) {
    T* Sjj = S + dblk(jb, jb);
    T r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = Sjj[lr * BP + c];
    // Branch-free: the 16 pivots are ONE basic block, so the scheduler can run the tail of pivot j's column updates under the
    // rsq / Goldschmidt latency of pivot j+1 (a per-pivot `if` for the failure case cut the block and serialised them).
    int bad = 0;          // first non-positive / NaN pivot of this tile (1-based), uniform
    T myinv = (T)0;       // lane j keeps 1/L_jj
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      T dj = readlane_t(r[j], j);
      const bool neg = !(dj > (T)0);
      bad = (neg && bad == 0) ? j + 1 : bad;
      dj = neg ? (T)1 : dj;
      const T y0 = rsq_seed(dj);
      T g = dj * y0, hh = (T)0.5 * y0;
      T e = fma(-hh, g, (T)0.5);
      g = fma(g, e, g); hh = fma(hh, e, hh);
      e = fma(-hh, g, (T)0.5);
      g = fma(g, e, g); hh = fma(hh, e, hh);
      const T e2 = fma(-g, g, dj);
      const T s = fma(e2, hh, g);      // sqrt(dj)
      const T inv = hh + hh;           // 1/sqrt(dj)
      const T lij = (lr == j) ? s : r[j] * inv;
      r[j] = lij;
      myinv = (lr == j) ? inv : myinv;
#pragma unroll
      for (int c = j + 1; c < 16; ++c) {
        const T lcj = readlane_t(lij, c);
        r[c] = fma(-lij, lcj, r[c]);
      }
    }
    if (lane < 16) {
      dinv[jb * 16 + lr] = myinv;
#pragma unroll
      for (int c = 0; c < 16; ++c) Sjj[lr * BP + c] = (c <= lr) ? r[c] : (T)0;
    }
    // LAPACK info = index of the first failing pivot
    if (bad != 0 && lane == 0 && *info == 0) *info = pivot_base + jb * 16 + bad;
  };
  // one 16x16 tile of the rank-16 update: C(ti, tj) -= L(ti, jb) L(tj, jb)^T  (two of them interleaved so that one's MFMA
  // dependency chain hides behind the other's)
  auto update_pair = [&](int jb, int t, bool two, int t2) {
    int ti0 = 0, rem = t;
    while (rem > ti0) { rem -= ti0 + 1; ++ti0; }
    const int tj0 = rem;
    int ti1 = 0; rem = two ? t2 : t;
    while (rem > ti1) { rem -= ti1 + 1; ++ti1; }
    const int tj1 = rem;
    T* C0 = S + dblk(jb + 1 + ti0, jb + 1 + tj0);
    T* C1 = S + dblk(jb + 1 + ti1, jb + 1 + tj1);
    const T* A0 = S + dblk(jb + 1 + ti0, jb), *B0 = S + dblk(jb + 1 + tj0, jb);
    const T* A1 = S + dblk(jb + 1 + ti1, jb), *B1 = S + dblk(jb + 1 + tj1, jb);
    acc_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc0[r] = C0[N_::drow(lq, r) * BP + lr];
      acc1[r] = C1[N_::drow(lq, r) * BP + lr];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const T a0 = -A0[lr * BP + kk * 4 + lq];
      const T b0 = B0[lr * BP + kk * 4 + lq];
      const T a1 = -A1[lr * BP + kk * 4 + lq];
      const T b1 = B1[lr * BP + kk * 4 + lq];
      acc0 = N_::mfma(a0, b0, acc0);
      acc1 = N_::mfma(a1, b1, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C0[N_::drow(lq, r) * BP + lr] = acc0[r];
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) C1[N_::drow(lq, r) * BP + lr] = acc1[r];
    }
  };

  if (wave == 0 && !(skip & 1)) factor16(0);
  __syncthreads();
  for (int jb = 0; jb < 8; ++jb) {
    T* Sjj = S + dblk(jb, jb);
    // B2: rows below: x L11^T = a   (one thread per row)
    const int nrows = DB - (jb * 16 + 16);
    if (tid < nrows && !(skip & 2)) {
      const int row = jb * 16 + 16 + tid;
      T* Sr = S + dblk(row >> 4, jb) + (row & 15) * BP;
      T x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = Sr[c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x[j] *= dinv[jb * 16 + j];
#pragma unroll
        for (int p = j + 1; p < 16; ++p) x[p] = fma(-x[j], Sjj[p * BP + j], x[p]);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) Sr[c] = x[c];
    }
    __syncthreads();
    // B3 with look-ahead: wave 0 updates the NEXT diagonal tile (tile 0 of the trailing grid) and factors it straight away
    // (B1 of step jb + 1, the serial pivot chain) while waves 1..7 apply the rank-16 update to all the other lower tiles of
    // the trailing (7-jb)x(7-jb) block grid, two tiles per iteration.
    const int nb = 7 - jb;
    const int nt = nb * (nb + 1) / 2;
    if (wave == 0) {
      if (nt > 0 && !(skip & 4)) update_pair(jb, 0, false, 0);
      if (jb < 7 && !(skip & 1)) factor16(jb + 1);
    } else if (!(skip & 4)) {
      for (int t = wave; t < nt; t += 14) update_pair(jb, t, (t + 7) < nt, t + 7);
    }
    __syncthreads();
  }

  // ---- write L11 back (lower triangle only; the strictly-upper part of the block is never read) ----
#pragma unroll
  for (int it = 0; it < DB * (DB / 2) / DIAG_THREADS; ++it) {
    const int idx = tid + it * DIAG_THREADS;
    const int row = idx >> 6, cp = (idx & 63) * 2;
    if (cp <= row) {
      v2_t v = *(const v2_t*)(S + dblk(row >> 4, cp >> 4) + (row & 15) * BP + (cp & 15));
      if (cp + 1 > row) v.y = (T)0;
      *(v2_t*)(A + (long)row * lda + cp) = v;
    }
  }

  if (skip & 8) return;
  __syncthreads();   // the diagonal blocks are about to be overwritten by their inverses
  // ---- inverse: 16x16 diagonal-block inverses in place (columns of inv(L_bb) by forward substitution) ----
  if (wave < 2) {
    const int b = wave * 4 + lq, c = lr;
    T* Sbb = S + dblk(b, b);
    T x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == c) ? (T)1 : (T)0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] *= dinv[b * 16 + i];
#pragma unroll
      for (int p = i + 1; p < 16; ++p) x[p] = fma(-Sbb[p * BP + i], x[i], x[p]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) Sbb[i * BP + c] = x[i];
  }
  __syncthreads();

  // ---- blocked in-place inverse by recursive doubling:  [L11 0; L21 L22]^-1 = [X11 0; -X22 (L21 X11)  X22] ----
  // Level s (in 16x16 tiles, s = 1, 2, 4) joins the 8/(2s) pairs of finished s-tile inverse blocks: first P = L21 X11 (it
  // overwrites L21, whose copy in global memory was written back above), then X21 = -X22 P; the 4s tiles of a level are
  // dealt over the 8 waves (two per wave at s = 4).  3 levels x 4 barriers and dependency chains of at most 16 MFMAs,
  // against 7 column steps x 2 barriers with chains of up to 28 in the column-by-column recurrence this replaces (9.7 us).
  for (int sblk = 1; sblk < 8 && !(skip & 16); sblk *= 2) {
    const int ntile = 4 * sblk, s2 = sblk * sblk;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
      acc_t res[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) res[q][r] = (T)0;
        const int t = wave + 8 * q;
        if (t < ntile) {
          const int pair = t / s2, w = t - pair * s2, i = w / sblk, j = w - i * sblk;
          const int b0 = pair * 2 * sblk;
          // phase 0:  P(i, j)   = sum_{k = j}^{s-1} L21(i, k) X11(k, j)      (X11 lower triangular)
          // phase 1:  X21(i, j) = -sum_{k = 0}^{i}  X22(i, k) P(k, j)        (X22 lower triangular)
          const int k0 = phase == 0 ? j : 0, k1 = phase == 0 ? sblk - 1 : i;
          for (int k = k0; k <= k1; ++k) {
            const T* A_ = phase == 0 ? S + dblk(b0 + sblk + i, b0 + k) : S + dblk(b0 + sblk + i, b0 + sblk + k);
            const T* B_ = phase == 0 ? S + dblk(b0 + k, b0 + j) : S + dblk(b0 + sblk + k, b0 + j);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              T a_ = A_[lr * BP + kk * 4 + lq];           // (row lr, k)
              const T b_ = B_[(kk * 4 + lq) * BP + lr];   // (k, col lr)
              if (phase == 1) a_ = -a_;
              res[q] = N_::mfma(a_, b_, res[q]);
            }
          }
        }
      }
      __syncthreads();   // every read of the blocks about to be overwritten (L21, then P) is done
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = wave + 8 * q;
        if (t < ntile) {
          const int pair = t / s2, w = t - pair * s2, i = w / sblk, j = w - i * sblk;
          T* D = S + dblk(pair * 2 * sblk + sblk + i, pair * 2 * sblk + j);
#pragma unroll
          for (int r = 0; r < 4; ++r) D[N_::drow(lq, r) * BP + lr] = res[q][r];
        }
      }
      __syncthreads();
    }
  }

  // ---- write inv(L11) (lower triangle) ----
#pragma unroll
  for (int it = 0; it < DB * (DB / 2) / DIAG_THREADS; ++it) {
    const int idx = tid + it * DIAG_THREADS;
    const int row = idx >> 6, cp = (idx & 63) * 2;
    if (cp <= row) {
      v2_t v = *(const v2_t*)(S + dblk(row >> 4, cp >> 4) + (row & 15) * BP + (cp & 15));
      if (cp + 1 > row) v.y = (T)0;
      *(v2_t*)(Linv + row * DB + cp) = v;
    }
  }
}

}  // namespace sigp
