// HBM-bound kernels around the factorisation: covariance build (K5), ride-along rows (K9), fused
// reductions (K8, K11, K12).  fp64 throughout.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"
#include "lane_ops.hpp"

namespace sigp {

enum { KID_NETDIFFUSION = 0, KID_RBF = 1, KID_MATERN52 = 2,
       KID_RBF_DLOGL = 17, KID_MATERN52_DLOGL = 18 };   // d k~ / d log(ell), for the exact MLII gradient

struct KParams {
  int kernel_id;
  int ds;           // data-set index of this batch member (X + ds*strideX, y + ds*stridey, Xs + ds*strideXs)
  double c_rbf;     // -0.5 / ell^2
  double inv_ell;   // 1 / ell
  double sn;        // sigma_n tilde (added on the diagonal)
};

// exp(x) for the covariance functions (x <= 0, finite): k = rint(x log2 e), r = x - k ln2 (Cody-Waite, ln2 split so that
// k ln2_hi is exact), degree-13 Taylor polynomial on |r| <= 0.347 (truncation 4e-18), v_ldexp_f64.  <= 2 ulp.  The library
// exp() costs the build more than its arithmetic: every one of its 64-bit polynomial coefficients is re-materialised with two
// v_mov_b32 per use (VALU issue slots; the build is VALU-bound: 2.84 ms without its stores, 2.31 ms of stores alone), and it
// carries overflow / NaN paths that cannot occur here.  The coefficients below sit in a __constant__ table: uniform scalar
// loads, used as the one scalar operand a v_fma_f64 may take.
struct ExpCoef { double l2e, ln2hi, ln2lo, c[14]; };
__constant__ ExpCoef kExpCoef = {1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                                 {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                  1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0}};
__device__ __forceinline__ double exp_cov(double x) {
  x = fmax(x, -746.0);               // exp underflows to 0 below this: keeps k in int range and r accurate for any length scale (l -> 0, x -> -inf)
  const double k = __builtin_rint(x * kExpCoef.l2e);
  double r = fma(-k, kExpCoef.ln2hi, x);
  r = fma(-k, kExpCoef.ln2lo, r);
  double p = kExpCoef.c[13];
#pragma unroll
  for (int i = 12; i >= 0; --i) p = fma(p, r, kExpCoef.c[i]);
  return ldexp(p, (int)k);
}

// The same function with 6 of the argument reduction's bits moved into a 64-entry table of 2^(j/64) (in LDS: one ds_read_b64 per call on the
// LDS port, which the build leaves idle): k = rint(64 x log2 e), r = x - k ln2 / 64 (|r| <= 0.0054), degree-5 polynomial (truncation 4e-17),
// exp(x) = 2^(k >> 6) tab[k & 63] p(r).  16 VALU instructions instead of 21: the covariance build is VALU-bound beside its stores.  The
// table is filled by exp_cov itself (exp_tab_fill, 64 threads, once per workgroup): <= 4 ulp in all.
__device__ __forceinline__ void exp_tab_fill(double* tab, int tid) {
  if (tid < 64) tab[tid] = exp_cov(-(double)((64 - tid) & 63) * (6.93147180559945286227e-01 / 64.0)) * (tid ? 2.0 : 1.0);   // 2^(tid/64) = 2 exp(-(64 - tid) ln2 / 64)
}
__device__ __forceinline__ double exp_cov_tab(double x, const double* tab) {
  x = fmax(x, -746.0);
  const double k = __builtin_rint(x * (64.0 * 1.44269504088896338700e+00));
  double r = fma(-k, kExpCoef.ln2hi * (1.0 / 64.0), x);
  r = fma(-k, kExpCoef.ln2lo * (1.0 / 64.0), r);
  const int ki = (int)k;
  const double t = tab[ki & 63];
  double p = kExpCoef.c[5];
#pragma unroll
  for (int i = 4; i >= 0; --i) p = fma(p, r, kExpCoef.c[i]);
  return ldexp(t * p, ki >> 6);
}

__device__ inline double cov_from_sq(const KParams& kp, double sq) {
  if (kp.kernel_id == KID_RBF) return exp_cov(kp.c_rbf * sq);
  if (kp.kernel_id == KID_RBF_DLOGL) return exp_cov(kp.c_rbf * sq) * sq * (kp.inv_ell * kp.inv_ell);   // k * |d|^2 / l^2
  const double s = sqrt(5.0 * sq) * kp.inv_ell;
  if (kp.kernel_id == KID_MATERN52_DLOGL) return (s * s * (1.0 / 3.0)) * (1.0 + s) * exp_cov(-s);
  return (1.0 + s + s * s * (1.0 / 3.0)) * exp_cov(-s);
}
__device__ inline double cov_from_sq_tab(const KParams& kp, double sq, const double* tab) {      // (the two covariance functions of a fit, table-driven exp)
  if (kp.kernel_id == KID_RBF) return exp_cov_tab(kp.c_rbf * sq, tab);
  if (kp.kernel_id == KID_MATERN52) {
    const double s = sqrt(5.0 * sq) * kp.inv_ell;
    return (1.0 + s + s * s * (1.0 / 3.0)) * exp_cov_tab(-s, tab);
  }
  return cov_from_sq(kp, sq);
}

// K5 (north/June1st.py:265 with an RBF / Matern-5/2 covariance in place of X Sigma X^T):
// K~ = k(X,X) + sn I in 64-row x 128-column tiles, identity on the padding rows/cols (i or j >= n).
// X is [n_pad][dp] row-major, zero padded.  A thread owns 8 rows x 4 consecutive columns (per feature: 8 broadcast
// LDS reads + two 16-byte reads for 32 elements), a wave writes two 1-KiB row segments per store.
// Lower-triangle build (flags bit 0 clear): a 1-D grid over exactly the tiles that touch the lower triangle
// (kbuild_tiles(n_pad) of them; tiles on the diagonal also fill the part of the upper triangle they cover, which
// nothing reads).  flags bit 0 set ("full", derivative matrices): 2-D grid (n_pad/128, n_pad/64), every tile, zeros
// instead of the identity on the padding.  blockIdx.z = batch member; hyper-parameters and data set from kps[z].
constexpr int KB_TM = 64;    // tile rows
constexpr int KB_TN = 128;   // tile columns
constexpr int KB_DC = 32;    // feature chunk staged in LDS
inline long kbuild_tiles(long n_pad) { const long T = n_pad / KB_TN; return T * (T + 1); }
template <typename TO>
struct alignas(4 * sizeof(TO)) KbOut4 { TO v[4]; };

template <typename TO, int DC = KB_DC>
__global__ __launch_bounds__(256) void kbuild_kernel(const double* __restrict__ X, long strideX, int dp, int d, int n,
                                                     TO* __restrict__ Mat, long strideM, long ld,
                                                     const KParams* __restrict__ kps, int flags_in, int colblk0 = 0,
                                                     double* __restrict__ Mat64 = nullptr, long stride64 = 0) {
  const int flags = flags_in & (DBG_MASK | 1 | 8 | 16);
  const int full = flags & 1;      // flag bits 2 / 4: timing ablations (no covariance function / no store)
  int bi, bj;                      // 64-row tile, 128-column tile
  long mcol = -1;                  // column of the tile in Mat when it is not the global one
  if (flags & 16) {                // "cyclic": ALL block columns one rank owns in a matrix sharded by block-cyclic panels, in one launch:
    const int W = colblk0 & 255, world = (colblk0 >> 8) & 255, rank = (colblk0 >> 16) & 255;   // local block column x = blockIdx.x of
    bi = blockIdx.y;                                                                           // Mat [rows][own columns] is global block
    bj = ((int)blockIdx.x / W * world + rank) * W + (int)blockIdx.x % W;                       // column (x / W world + rank) W + x % W
    if (bi * KB_TM + KB_TM <= bj * KB_TN) return;
    mcol = (long)blockIdx.x * KB_TN;
  } else if (flags & 8) {          // "panel": the column blocks colblk0 .. colblk0 + gridDim.x of the lower triangle only (one rank's
    bi = blockIdx.y; bj = colblk0 + blockIdx.x;   // share of a matrix sharded by block columns); Mat is the virtual origin of the
    if (bi * KB_TM + KB_TM <= bj * KB_TN) return; // local storage, so global (row, column) indexing lands in it
  } else if (full) {
    bi = blockIdx.y; bj = blockIdx.x;
  } else {                         // row-tile pair k = (2k, 2k+1) has column tiles 0..k
    const int u = blockIdx.x >> 1;
    int k = (int)((sqrt(8.0 * u + 1.0) - 1.0) * 0.5);
    while ((k + 1) * (k + 2) / 2 <= u) ++k;
    while (k * (k + 1) / 2 > u) --k;
    bj = u - k * (k + 1) / 2;
    bi = 2 * k + (blockIdx.x & 1);
  }
  const KParams kp = kps[blockIdx.z];
  X += kp.ds * strideX;
  Mat += blockIdx.z * strideM;
  __shared__ double Xi[KB_TM][DC + 1];                                    // DC = 8 for d <= 8: 12 KB instead of 50 KB of LDS, so the
  __shared__ __attribute__((aligned(16))) double XjT[DC][KB_TN + 2];      // CU holds enough workgroups to overlap one's exp() with another's stores
  __shared__ double etab[64];                                             // 2^(j/64) for exp_cov_tab (published by the staging loop's barriers)
  const int tid = threadIdx.x;
  exp_tab_fill(etab, tid);
  const int c4 = (tid & 31) * 4, rg = tid >> 5;
  double acc[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[s][c] = 0.0;
  for (int p0 = 0; p0 < d; p0 += DC) {
    const int pc = min(DC, d - p0);
    __syncthreads();
    // staging: lanes run along the rows, so the transposed image XjT[p][r] is written conflict-free and only the
    // pc live features are touched (the strided 8-byte global reads hit the same lines again for the next p)
    for (int idx = tid; idx < KB_TN * pc; idx += 256) {
      const int r = idx & (KB_TN - 1), p = idx >> 7;
      XjT[p][r] = X[(long)(bj * KB_TN + r) * dp + p0 + p];
    }
    for (int idx = tid; idx < KB_TM * pc; idx += 256) {
      const int r = idx & (KB_TM - 1), p = idx >> 6;
      Xi[r][p] = X[(long)(bi * KB_TM + r) * dp + p0 + p];
    }
    __syncthreads();
    for (int p = 0; p < pc; ++p) {
      const d2 b01 = *(const d2*)(&XjT[p][c4]);
      const d2 b23 = *(const d2*)(&XjT[p][c4 + 2]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double a = Xi[rg + 8 * s][p];
        const double d0 = a - b01.x, d1 = a - b01.y, d2_ = a - b23.x, d3 = a - b23.y;
        acc[s][0] += d0 * d0;
        acc[s][1] += d1 * d1;
        acc[s][2] += d2_ * d2_;
        acc[s][3] += d3 * d3;
      }
    }
  }
  // Tiles that neither touch the diagonal nor the padding (all but 2 T of the T (T + 1) tiles of a member) take a path without the per-element
  // index compares and selects: the build is VALU-bound beside its stores (tools/kbuild_bench.py), and those were ~6 of ~46 instructions
  // per element.  Same values.
  const bool interior = bi * KB_TM + KB_TM <= n && bj * KB_TN + KB_TN <= n && (bi * KB_TM + KB_TM <= bj * KB_TN || bi * KB_TM >= bj * KB_TN + KB_TN);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int gi = bi * KB_TM + rg + 8 * s, gj = bj * KB_TN + c4;
    KbOut4<TO> o;
    KbOut4<double> o64;
    if (interior) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double v = (flags & 2) ? acc[s][c] : cov_from_sq_tab(kp, acc[s][c], etab);
        o.v[c] = (TO)v; o64.v[c] = v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double v;
        if (gi >= n) v = (gi == gj + c && !full) ? 1.0 : 0.0;
        else v = (gj + c < n) ? ((flags & 2) ? acc[s][c] : cov_from_sq_tab(kp, acc[s][c], etab)) + (gi == gj + c ? kp.sn : 0.0) : 0.0;
        o.v[c] = (TO)v; o64.v[c] = v;
      }
    }
    if (!(flags & 4) || o.v[0] == (TO)12345.678) *(KbOut4<TO>*)(Mat + (long)gi * ld + (mcol >= 0 ? mcol + c4 : (long)gj)) = o;
    // the same tile in fp64 beside a lower-precision one (fp32 engine: what the refinement's residuals read), same [row][ld] layout
    if (Mat64 != nullptr) *(KbOut4<double>*)(Mat64 + blockIdx.z * stride64 + (long)gi * ld + gj) = o64;
  }
}

// The same tiles with the squared distances in GEMM form on the matrix pipe (SURVEY 8d counts the build that way):
//     |x_i - x_j|^2 = |x_i|^2 + |x_j|^2 - 2 x_i . x_j,      the inner products by v_mfma_f64_16x16x4_f64
// from LDS images of the tile's 64 + 128 rows of X (feature chunks of DC).  kbuild_kernel spends 2 d fp64 VALU instructions per element
// on the distance (d = 32: 64 of its ~95) while the matrix pipe idles; here the VALU keeps the covariance function and three adds, and
// the kernel is left with its stores.  Wave w owns tile rows 16 w .. 16 w + 15 and all eight 16-column tiles; the column operand is fed
// in the order sigma(i) = 4 (i mod 4) + i div 4, so that a lane's four accumulator registers (rows lq + 4 r of the MFMA result) are FOUR
// CONSECUTIVE COLUMNS 4 lq .. 4 lq + 3 of matrix row lr: one 32-byte (fp64) / 16-byte (fp32) store per lane and tile, a whole 128-byte
// line per matrix row and wave instruction.  Rounding: the Gram form loses relative accuracy for near-coincident points, not absolute:
// |error(sq)| <~ 4 eps max(|x_i|^2, |x_j|^2), i.e. <= 1e-15 d in K~ at the synthetic workloads' scales (K~ tolerance 1e-13); the diagonal
// is exact (sq = 0 by construction), sq is clamped at 0.
template <typename TO, int DC>
__global__ __launch_bounds__(256) void kbuild_mfma_kernel(const double* __restrict__ X, long strideX, int dp, int d, int n,
                                                          TO* __restrict__ Mat, long strideM, long ld,
                                                          const KParams* __restrict__ kps, int flags_in, int colblk0 = 0,
                                                          double* __restrict__ Mat64 = nullptr, long stride64 = 0) {
  const int flags = flags_in & (DBG_MASK | 1 | 8 | 16);
  const int full = flags & 1;
  int bi, bj;
  long mcol = -1;
  if (flags & 16) {                // (tile selection exactly as kbuild_kernel)
    const int W = colblk0 & 255, world = (colblk0 >> 8) & 255, rank = (colblk0 >> 16) & 255;
    bi = blockIdx.y;
    bj = ((int)blockIdx.x / W * world + rank) * W + (int)blockIdx.x % W;
    if (bi * KB_TM + KB_TM <= bj * KB_TN) return;
    mcol = (long)blockIdx.x * KB_TN;
  } else if (flags & 8) {
    bi = blockIdx.y; bj = colblk0 + blockIdx.x;
    if (bi * KB_TM + KB_TM <= bj * KB_TN) return;
  } else if (full) {
    bi = blockIdx.y; bj = blockIdx.x;
  } else {
    const int u = blockIdx.x >> 1;
    int k = (int)((sqrt(8.0 * u + 1.0) - 1.0) * 0.5);
    while ((k + 1) * (k + 2) / 2 <= u) ++k;
    while (k * (k + 1) / 2 > u) --k;
    bj = u - k * (k + 1) / 2;
    bi = 2 * k + (blockIdx.x & 1);
  }
  const KParams kp = kps[blockIdx.z];
  X += kp.ds * strideX;
  Mat += blockIdx.z * strideM;
  constexpr int LP = DC + 2;                                       // conflict-free 8-byte fragment reads (row = lane & 15, k = lane >> 4)
  __shared__ __attribute__((aligned(16))) double Xi[KB_TM * LP];
  __shared__ __attribute__((aligned(16))) double Xj[KB_TN * LP];
  __shared__ __attribute__((aligned(16))) double nI[KB_TM], nJ[KB_TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int dq = (d + 3) & ~3;                                     // (X is zero-padded to dp >= dq features)
  if (tid < KB_TM + KB_TN) {                                       // squared norms of the tile's rows, one thread per row
    const double* xr = X + (long)(tid < KB_TM ? bi * KB_TM + tid : bj * KB_TN + tid - KB_TM) * dp;
    double sacc = 0.0;
    for (int p = 0; p < dq; p += 2) { const d2 v = *(const d2*)(xr + p); sacc = fma(v.x, v.x, sacc); sacc = fma(v.y, v.y, sacc); }
    if (tid < KB_TM) nI[tid] = sacc; else nJ[tid - KB_TM] = sacc;
  }
  d4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
  const int sig = 4 * (lr & 3) + (lr >> 2);                        // the column fed as MFMA row lr
  for (int p0 = 0; p0 < dq; p0 += DC) {
    const int pc = min(DC, dq - p0);                               // multiple of 4
    __syncthreads();
    for (int idx = tid; idx < (KB_TM + KB_TN) * (pc >> 1); idx += 256) {      // 16-byte pieces, lanes along the features of a row
      const int row = idx / (pc >> 1), q = (idx - row * (pc >> 1)) * 2;
      const d2 v = *(const d2*)(X + (long)(row < KB_TM ? bi * KB_TM + row : bj * KB_TN + row - KB_TM) * dp + p0 + q);
      double* dst = row < KB_TM ? Xi + row * LP + q : Xj + (row - KB_TM) * LP + q;
      *(d2*)dst = v;
    }
    __syncthreads();
    const double* bI = Xi + (16 * wave + lr) * LP + lq;
    const double* aJ = Xj + sig * LP + lq;
    for (int kk = 0; kk < pc; kk += 4) {
      const double b = bI[kk];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aJ[16 * t * LP + kk], b, acc[t], 0, 0, 0);
    }
  }
  const int gi = bi * KB_TM + 16 * wave + lr;
  const double ni = nI[16 * wave + lr];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int gj = bj * KB_TN + 16 * t + 4 * lq;
    const d2 nj01 = *(const d2*)(nJ + 16 * t + 4 * lq), nj23 = *(const d2*)(nJ + 16 * t + 4 * lq + 2);
    const double njv[4] = {nj01.x, nj01.y, nj23.x, nj23.y};
    KbOut4<TO> o;
    KbOut4<double> o64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double sq = fmax(fma(-2.0, acc[t][c], ni + njv[c]), 0.0);
      if (gi == gj + c) sq = 0.0;
      double v;
      if (gi >= n) v = (gi == gj + c && !full) ? 1.0 : 0.0;
      else v = (gj + c < n) ? ((flags & 2) ? sq : cov_from_sq(kp, sq)) + (gi == gj + c ? kp.sn : 0.0) : 0.0;
      o.v[c] = (TO)v; o64.v[c] = v;
    }
    if (!(flags & 4) || o.v[0] == (TO)12345.678) *(KbOut4<TO>*)(Mat + (long)gi * ld + (mcol >= 0 ? mcol + 16 * t + 4 * lq : (long)gj)) = o;
    if (Mat64 != nullptr) *(KbOut4<double>*)(Mat64 + blockIdx.z * stride64 + (long)gi * ld + gj) = o64;
  }
}

// Ride-along block (128 rows x n_pad): row 0 = y (may be null -> zeros), rows 1..m = k~(xs_j, x_i)
// (north/June1st.py:272 KXXs^T in unit signal variance), remaining rows 0.  first_row lets predict()
// fill rows 0..m-1 with cross-covariances only (y == nullptr, first_row = 0).
template <typename TO>
__global__ __launch_bounds__(256) void ride_build_kernel(const double* __restrict__ X, long strideX, const double* __restrict__ Xs,
                                                         long strideXs, const double* __restrict__ y, long stridey, int dp, int d,
                                                         int n, int n_pad, int m, int first_row, TO* __restrict__ Z,
                                                         long strideZ, long ld, const KParams* __restrict__ kps, int compute_cov,
                                                         int i0 = 0, int icount = -1, int cyclic = 0) {
  int i = i0 + blockIdx.x * 256 + threadIdx.x;         // column (training point); [i0, i0 + icount): one rank's block columns
  const int r = blockIdx.y;                            // row of the ride block
  long zcol = -1;
  if (cyclic) {                                        // all own columns of a block-cyclic sharding at once (cyclic = W | world << 8 | rank << 16):
    const int W = cyclic & 255, world = (cyclic >> 8) & 255, rank = (cyclic >> 16) & 255;      // local column i of Z [128][icount] is global column ...
    if (i >= icount) return;
    zcol = i;
    const int x = i >> 7;
    i = (((x / W) * world + rank) * W + x % W) * 128 + (i & 127);
  }
  if (i >= n_pad || (!cyclic && icount >= 0 && i >= i0 + icount)) return;
  const KParams kp = kps[blockIdx.z];
  X += kp.ds * strideX;
  Xs += kp.ds * strideXs;
  if (y != nullptr) y += kp.ds * stridey;
  Z += blockIdx.z * strideZ;
  double v = 0.0;
  if (i < n) {
    if (y != nullptr && r == 0) {
      v = y[i];
    } else if (compute_cov && r >= first_row && r < first_row + m) {
      const double* xs = Xs + (long)(r - first_row) * dp;
      const double* xi = X + (long)i * dp;
      double sq = 0.0;
      for (int p = 0; p < d; ++p) { const double t = xs[p] - xi[p]; sq += t * t; }
      v = cov_from_sq(kp, sq);
    } else if (!compute_cov) {
      return;   // rows were produced by a GEMM (reference kernel); only row 0 / padding handled here
    }
  }
  Z[(long)r * ld + (zcol >= 0 ? zcol : (long)i)] = (TO)v;
}

// after a GEMM-form build (reference kernel): add sn on the diagonal, identity on the padding
__global__ void diag_fix_kernel(double* __restrict__ Mat, long ld, int n, int n_pad, double sn, int i0 = 0, int icount = -1) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad || (icount >= 0 && i >= i0 + icount)) return;
  if (i < n) Mat[(long)i * ld + i] += sn; else Mat[(long)i * ld + i] = 1.0;
}

// zero-padded copy  dst[rows_pad][dp] <- src[rows][d] (row stride lds)
__global__ void pad_copy_kernel(const double* __restrict__ src, long lds, int rows, int d, double* __restrict__ dst,
                                int rows_pad, int dp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows_pad * dp) return;
  const int r = (int)(idx / dp), p = (int)(idx % dp);
  dst[idx] = (r < rows && p < d) ? src[(long)r * lds + p] : 0.0;
}

__device__ inline double block_reduce_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;   // valid on thread 0
}

// K8/K11/K12 fused reductions over the solved ride rows W = [z ; v_1..v_m] (z = L~^-1 y, v = L~^-1 k~*):
//   res[r]       = W[r] . zrow      (r = 0: y^T A~ = n sigma_f, north/June1st.py:267;  r>0: fmean, :276)
//   res[128 + r] = W[r] . W[r]      (v^T v of north/June1st.py:277)
//   res[256]     = sum_i<n log L~_ii (north/June1st.py:246)
// grid = nrows + 1 blocks of 256 threads.
// blockIdx.y = batch member (strides sW, sZ, sM; res += 512*y).
template <typename TI>
__global__ __launch_bounds__(256) void epilogue_kernel(const TI* __restrict__ W, long ldw, const TI* __restrict__ zrow,
                                                       const TI* __restrict__ Mat, long ld, int n, int n_pad, int nrows,
                                                       double* __restrict__ res, long sW, long sZ, long sM) {
  __shared__ double sh[4];
  const int r = blockIdx.x;
  W += blockIdx.y * sW;
  zrow += blockIdx.y * sZ;
  if (Mat != nullptr) Mat += blockIdx.y * sM;
  res += blockIdx.y * 512;
  if (r < nrows) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_pad; i += 256) {
      const double w = (double)W[(long)r * ldw + i];
      a += w * (double)zrow[i];
      b += w * w;
    }
    a = block_reduce_sum(a, sh);
    b = block_reduce_sum(b, sh);
    if (threadIdx.x == 0) { res[r] = a; res[128 + r] = b; }
  } else if (Mat != nullptr) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += log((double)Mat[(long)i * ld + i]);
    a = block_reduce_sum(a, sh);
    if (threadIdx.x == 0) res[256] = a;
  }
}

// sum_i log L_ii over the diagonal entries [i0, i0 + icount) n (i < n) of a matrix given by its virtual origin (one rank's
// block columns of a factor sharded by columns); accumulates into *out (one block)
// cyclic = W | world << 8 | rank << 16: Mat [rows][ncol] holds the rank's own block columns side by side; one block walks them all
// in order (a fixed summation order), *out = the sum.
template <typename TI>
__global__ __launch_bounds__(256) void logdiag_cyclic_kernel(const TI* __restrict__ Mat, long ld, int n, int ncol, int cyclic, double* __restrict__ out) {
  __shared__ double sh[4];
  const int W = cyclic & 255, world = (cyclic >> 8) & 255, rank = (cyclic >> 16) & 255;
  double a = 0.0;
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const int x = c >> 7;
    const int i = (((x / W) * world + rank) * W + x % W) * 128 + (c & 127);     // global column = row of the diagonal entry
    if (i < n) a += log((double)Mat[(long)i * ld + c]);
  }
  a = block_reduce_sum(a, sh);
  if (threadIdx.x == 0) *out = a;
}

// copy a [rows][cols] block (device -> device) with different strides
__global__ void copy_block_kernel(const double* __restrict__ src, long lds, double* __restrict__ dst, long ldd, int rows,
                                  int cols) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx % cols);
  dst[(long)r * ldd + c] = src[(long)r * lds + c];
}

// pack a [rows][row_bytes] block into a contiguous buffer (sharded fit: a factored panel -> the broadcast buffer): 16-byte
// accesses, one 256-thread block per 4 rows (hipMemcpy2DAsync device-to-device runs a fraction of this rate on strided rows).
// row_bytes and both strides are multiples of 16.
__global__ __launch_bounds__(256) void pack_rows_kernel(const char* __restrict__ src, long src_stride, char* __restrict__ dst, long dst_stride, long rows, int row_bytes) {
  const int nv = row_bytes >> 4;
  for (long r = (long)blockIdx.x * 4; r < min(rows, (long)blockIdx.x * 4 + 4); ++r) {
    const uint4* s = (const uint4*)(src + r * src_stride);
    uint4* d = (uint4*)(dst + r * dst_stride);
    for (int v = threadIdx.x; v < nv; v += 256) d[v] = s[v];
  }
}

// panel_mode 1: diagonal blocks of the Mt workspace = the inverse diagonal blocks of the panel (grid: block j, member)
template <typename T>
__global__ void mt_diag_kernel(const T* __restrict__ dinv, long dinvStride, T* __restrict__ Mt, long ldm, long mtStride) {
  const int j = blockIdx.x;
  const T* src = dinv + (long)blockIdx.y * dinvStride + (long)j * 128 * 128;
  T* dst = Mt + (long)blockIdx.y * mtStride + (long)j * 128 * ldm + (long)j * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += blockDim.x) dst[(long)(e >> 7) * ldm + (e & 127)] = src[e];
}

// U[b][b] = dinv[b]^T for every 128-block b of the diagonal (one workgroup per block, through LDS so that both sides are
// coalesced): the leaves of the recursive triangular inversion in sigp_nlml_grad.
template <typename T>
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const T* __restrict__ dinv, T* __restrict__ U, long ld, long sD = 0, long sU = 0) {
  __shared__ T tile[32][33];
  const T* src = dinv + (long)blockIdx.y * sD + (long)blockIdx.x * 128 * 128;      // blockIdx.y = lockstep member
  T* dst = U + (long)blockIdx.y * sU + (long)blockIdx.x * 128 * (ld + 1);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int bi = 0; bi < 4; ++bi)
    for (int bj = 0; bj < 4; ++bj) {
      __syncthreads();
      for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(long)(bi * 32 + r) * 128 + bj * 32 + tx];
      __syncthreads();
      for (int r = ty; r < 32; r += 8) dst[(long)(bj * 32 + r) * ld + bi * 32 + tx] = tile[tx][r];
    }
}

// K13/K14 reductions (north/June1st.py:251-252).  Kinv = K~^-1 (lower triangle valid), D = a symmetric
// derivative matrix (full), a = A~ (K~^-1 y).  One block per row i < n:
//   part[4i+0] = Kinv_ii D_ii + 2 sum_{j<i} Kinv_ij D_ij   (row i's share of tr(K~^-1 D), lower triangle only)
//   part[4i+1] = a_i * sum_j D_ij a_j
//   part[4i+2] = K~^-1_ii              part[4i+3] = a_i^2
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ Kinv, const double* __restrict__ D,
                                                          const double* __restrict__ a, long ld, int n, double* __restrict__ part) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double t = 0.0, q = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double dij = D[(long)i * ld + j];
    if (j <= i) t += (j < i ? 2.0 : 1.0) * Kinv[(long)i * ld + j] * dij;
    q += dij * a[j];
  }
  t = block_reduce_sum(t, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) {
    part[4 * i + 0] = t;
    part[4 * i + 1] = a[i] * q;
    part[4 * i + 2] = Kinv[(long)i * ld + i];
    part[4 * i + 3] = a[i] * a[i];
  }
}

// The same reductions for a lockstep group of RBF / Matern fits (sigp_nlml_grad_batch), the derivative matrix D = dK~/dlog l
// computed on the fly from X (never stored): blockIdx.y = member, one block per GR_ROWS rows, lower triangle only (D is
// symmetric with a zero diagonal for both kernels: a^T D a = 2 sum_{j<i} a_i D_ij a_j).  kps[member] carries the *_DLOGL
// covariance id and the data set.  part [member][rows][4] as above.
constexpr int GR_ROWS = 4;
__global__ __launch_bounds__(256) void grad_reduce_cov_kernel(const double* __restrict__ Kinv, long sK, long ld, const double* __restrict__ a, long sa,
                                                              const double* __restrict__ X, long strideX, int dp, int d, int n,
                                                              const KParams* __restrict__ kps, double* __restrict__ part, long spart) {
  __shared__ double Xi[GR_ROWS][65];
  __shared__ double sh[4];
  const KParams kp = kps[blockIdx.y];
  X += kp.ds * strideX;
  Kinv += (long)blockIdx.y * sK; a += (long)blockIdx.y * sa; part += (long)blockIdx.y * spart;
  const int i0 = blockIdx.x * GR_ROWS;
  for (int idx = threadIdx.x; idx < GR_ROWS * d; idx += 256) {
    const int r = idx / d, p = idx % d;
    Xi[r][p] = (i0 + r < n) ? X[(long)(i0 + r) * dp + p] : 0.0;
  }
  __syncthreads();
  double t[GR_ROWS], q[GR_ROWS];
#pragma unroll
  for (int r = 0; r < GR_ROWS; ++r) { t[r] = 0.0; q[r] = 0.0; }
  const int jmax = min(n, i0 + GR_ROWS);                   // j < i for the last row of the block
  for (int j = threadIdx.x; j < jmax; j += 256) {
    double sq[GR_ROWS];
#pragma unroll
    for (int r = 0; r < GR_ROWS; ++r) sq[r] = 0.0;
    const double* xj = X + (long)j * dp;
    for (int p = 0; p < d; ++p) {
      const double v = xj[p];
#pragma unroll
      for (int r = 0; r < GR_ROWS; ++r) { const double u = Xi[r][p] - v; sq[r] = fma(u, u, sq[r]); }
    }
    const double aj = a[j];
#pragma unroll
    for (int r = 0; r < GR_ROWS; ++r) {
      const int i = i0 + r;
      if (j < i && i < n) {
        const double dij = cov_from_sq(kp, sq[r]);
        t[r] = fma(2.0 * Kinv[(long)i * ld + j], dij, t[r]);
        q[r] = fma(dij, aj, q[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < GR_ROWS; ++r) {
    const double ts = block_reduce_sum(t[r], sh);
    const double qs = block_reduce_sum(q[r], sh);
    const int i = i0 + r;
    if (threadIdx.x == 0 && i < n) {
      const double ai = a[i];
      part[4 * (long)i + 0] = ts;                            // row i's share of tr(K~^-1 D)   (D_ii = 0)
      part[4 * (long)i + 1] = 2.0 * ai * qs;                 // row i's share of a^T D a
      part[4 * (long)i + 2] = Kinv[(long)i * ld + i];
      part[4 * (long)i + 3] = ai * ai;
    }
  }
}
// out [member][4] = column sums of part [member][n][4], one block per member, rows added in a fixed order
__global__ __launch_bounds__(256) void grad_sum_kernel(const double* __restrict__ part, long spart, int n, double* __restrict__ out) {
  __shared__ double sh[4];
  part += (long)blockIdx.x * spart;
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < 4; ++c) s4[c] += part[4 * (long)i + c];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double v = block_reduce_sum(s4[c], sh);
    if (threadIdx.x == 0) out[4 * blockIdx.x + c] = v;
  }
}

// ---- block triangular solves with a handful of right-hand sides -------------------------------------------------------
// (fp32 engine: x = L^-T L^-1 r of every refinement step, north/June1st.py:266 in preconditioner form.)  The factor is cut
// into big column blocks of TS_BS whose diagonal blocks carry explicit inverses (trtri_levels with span = TS_BS/128), so a
// solve is, per big block, ONE skinny product with the inverse diagonal block and ONE skinny update of everything that
// remains -- 2 launches per 2048 columns, each streaming its part of L exactly once at HBM speed -- instead of two tile-GEMM
// launches per 128 columns that move 128-row tiles for <= 4 useful rows.
constexpr int TS_BS = 2048;
constexpr int TS_RHS = 4;

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef f4 type; static constexpr int N = 4; };
template <> struct Vec16<double> { typedef d2 type; static constexpr int N = 2; };

// "row-dot":  Zout[r][j] (-)= sum_k Zin[r][k] * Mx[j*ld + k]   for the matrix rows j < nrows, one WAVE per row (the row is
// contiguous in k: 16-byte loads, 1 KiB per wave-instruction).  kmode 0: k in [0, K);  1: k in [0, j] (rows of a lower-
// triangular block);  2: k in [j, K) (rows of an upper-triangular block).  Elements outside the range are SELECTED away,
// never multiplied (the other triangle of an inverse block is uninitialised memory).
// Zadd (may be null; same leading dimension as Zin): the product is taken with Zin + Zadd (the sharded forward solve's
// right-hand side minus the contributions all-reduced over the ranks).  pm_pw > 0: Zout is PANEL-MAJOR [panel][TS_RHS][pm_pw]
// (entry of global row g = j_off + j of right-hand side r at ((g / pm_pw) TS_RHS + r) pm_pw + g % pm_pw), the layout in which
// one panel's piece of every right-hand side is one contiguous message.
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ Mx, long ld, int nrows, int K, int kmode,
                                                     const T* __restrict__ Zin, long ldzin, T* __restrict__ Zout, long ldzout,
                                                     int nrhs, int sub, const T* __restrict__ Zadd = nullptr, int pm_pw = 0, int j_off = 0,
                                                     long sM = 0, long sZi = 0, long sZo = 0) {
  typedef typename Vec16<T>::type V;
  constexpr int NV = Vec16<T>::N;
  Mx += (long)blockIdx.y * sM; Zin += (long)blockIdx.y * sZi; Zout += (long)blockIdx.y * sZo;      // blockIdx.y = lockstep member
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= nrows) return;
  const int klo = kmode == 2 ? j : 0, khi = kmode == 1 ? j + 1 : K;       // [klo, khi)
  const T* row = Mx + (long)j * ld;
  T acc[TS_RHS];
#pragma unroll
  for (int r = 0; r < TS_RHS; ++r) acc[r] = (T)0;
  const int step = 64 * NV;
  for (int k0 = (klo / step) * step; k0 < khi; k0 += step) {
    const int k = k0 + lane * NV;
    if (k >= khi || k + NV <= klo) continue;
    const V m = *(const V*)(row + k);
#pragma unroll
    for (int r = 0; r < TS_RHS; ++r) {
      if (r < nrhs) {
        V z = *(const V*)(Zin + (long)r * ldzin + k);
        if (Zadd != nullptr) {
          const V z2 = *(const V*)(Zadd + (long)r * ldzin + k);
#pragma unroll
          for (int e = 0; e < NV; ++e) z[e] += z2[e];
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const bool in = (k + e >= klo) && (k + e < khi);
          acc[r] += in ? m[e] * z[e] : (T)0;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < TS_RHS; ++r) {
    T v = acc[r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && r < nrhs) {
      const int g = j_off + j;
      T* o = pm_pw > 0 ? Zout + ((long)(g / pm_pw) * TS_RHS + r) * pm_pw + g % pm_pw : Zout + (long)r * ldzout + j;
      *o = sub ? *o - v : v;
    }
  }
}

// "column-dot":  Zout[r][j] -= sum_{k < K} Zin[r][k] * Mx[k*ld + j]   for the columns j < ncols (the rows of Mx are
// contiguous in j).  A workgroup owns 16 x (16 B) consecutive columns and splits k over its 16 thread rows; the 16 partial
// sums meet in LDS in a fixed order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void coldot_kernel(const T* __restrict__ Mx, long ld, int ncols, int K,
                                                     const T* __restrict__ Zin, long ldzin, T* __restrict__ Zout, long ldzout, int nrhs) {
  typedef typename Vec16<T>::type V;
  constexpr int NV = Vec16<T>::N;
  constexpr int CW = 16 * NV;                                 // columns per workgroup
  __shared__ T part[16][TS_RHS][CW + 1];
  const int cl = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int j = blockIdx.x * CW + cl * NV;
  T acc[TS_RHS][NV];
#pragma unroll
  for (int r = 0; r < TS_RHS; ++r)
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[r][e] = (T)0;
  if (j < ncols) {
    const int kper = (K + 15) / 16;
    const int k1 = min(K, (kg + 1) * kper);
    // eight rows in flight per thread (one 16-byte load each + the right-hand sides' entries): the loop was one dependent load per iteration --
    // 2.3 TB/s on the fp32 refinement's backward solves, the factor streamed at a fraction of what HBM gives.  Same k order per thread: same bits.
    constexpr int U = 8;
    int k = kg * kper;
    for (; k + U <= k1; k += U) {
      V m[U];
      T z[U][TS_RHS];
#pragma unroll
      for (int u = 0; u < U; ++u) m[u] = *(const V*)(Mx + (long)(k + u) * ld + j);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < TS_RHS; ++r) z[u][r] = r < nrhs ? Zin[(long)r * ldzin + k + u] : (T)0;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < TS_RHS; ++r)
#pragma unroll
          for (int e = 0; e < NV; ++e) acc[r][e] = fma(z[u][r], m[u][e], acc[r][e]);
    }
    for (; k < k1; ++k) {
      const V m = *(const V*)(Mx + (long)k * ld + j);
#pragma unroll
      for (int r = 0; r < TS_RHS; ++r) {
        if (r < nrhs) {
          const T z = Zin[(long)r * ldzin + k];
#pragma unroll
          for (int e = 0; e < NV; ++e) acc[r][e] = fma(z, m[e], acc[r][e]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < TS_RHS; ++r)
#pragma unroll
    for (int e = 0; e < NV; ++e) part[kg][r][cl * NV + e] = acc[r][e];
  __syncthreads();
  for (int idx = threadIdx.x; idx < nrhs * CW; idx += 256) {
    const int r = idx / CW, c = idx % CW;
    const int jc = blockIdx.x * CW + c;
    if (jc >= ncols) continue;
    T v = (T)0;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[g][r][c];
    Zout[(long)r * ldzout + jc] -= v;
  }
}

// dst big diagonal block b = (src big diagonal block b)^T, blocks of `bs` (last one `n - b*bs`) on the diagonal of [n][ld]
template <typename T>
__global__ __launch_bounds__(256) void transpose_diag_blocks_kernel(const T* __restrict__ src, T* __restrict__ dst, long ld, int n, int bs) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z, sz = min(bs, n - b * bs);
  const int ti = blockIdx.y * 32, tj = blockIdx.x * 32;
  if (ti >= sz || tj >= sz) return;
  const long o = (long)b * bs * (ld + 1);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = src[o + (long)(ti + r) * ld + tj + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) dst[o + (long)(tj + r) * ld + ti + tx] = tile[tx][r];
}

// ---- ComplexNetworks tau() on the device (behaviour of ComplexNetworks.py:31-47) ---------------------------
// Row standardisation  z_i = (x_i - mean_i) / ||x_i - mean_i||  so that the correlation matrix is the plain product Z Z^T
// (one MFMA GEMM).  Z is [n_pad][k_pad], zero padded.  A series with a NaN or zero variance gives a NaN row, as np.corrcoef does.
__global__ void corr_standardise_kernel(const double* __restrict__ X, long ldx, int n, int t, double* __restrict__ Z, int n_pad, int k_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  double* z = Z + (long)i * k_pad;
  if (i >= n) { for (int k = 0; k < k_pad; ++k) z[k] = 0.0; return; }
  const double* x = X + (long)i * ldx;
  double m = 0.0;
  for (int k = 0; k < t; ++k) m += x[k];
  m /= (double)t;
  double ss = 0.0;
  for (int k = 0; k < t; ++k) { const double c = x[k] - m; ss = fma(c, c, ss); }
  const double inv = 1.0 / sqrt(ss);
  for (int k = 0; k < k_pad; ++k) z[k] = k < t ? (x[k] - m) * inv : 0.0;
}

// In place on R [n][ld]: clip to [-1, 1] (as np.corrcoef), NaN on the diagonal (a cell is not its own neighbour), and the
// fused threshold reduction: per block, the sum and the count of the off-diagonal entries with r >= 0 and r > r_crit
// (<=> one-sided t-test p-value below the significance level, t = r sqrt(dof / (1 - r^2)) being monotone in r).
__global__ __launch_bounds__(256) void corr_threshold_kernel(double* __restrict__ R, long ld, int n, double r_crit, double* __restrict__ part) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double s = 0.0, c = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    double r = R[(long)i * ld + j];
    r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);        // NaN passes through both comparisons
    if (j == i) r = __builtin_nan("");
    R[(long)i * ld + j] = r;
    if (r >= 0.0 && r > r_crit) { s += r; c += 1.0; }
  }
  s = block_reduce_sum(s, sh);
  c = block_reduce_sum(c, sh);
  if (threadIdx.x == 0) { part[2 * i] = s; part[2 * i + 1] = c; }
}

// ---- ComplexNetworks intra_links(): per-area anomaly series (behaviour of ComplexNetworks.py:298-309) -------------------
// out[a][t] = sum over the pixels p with label[p] == a of data[p][t] * weight[p], NaN products counted as 0, pixels added in
// ascending p (row-major) order -- the same additions in the same order as the host restatement, so results are bit-identical.
// One thread per (area, time step); the label scan is shared by the T threads of an area through L1/L2.
__global__ void area_sums_kernel(const double* __restrict__ data, const double* __restrict__ weight, const int* __restrict__ label, int P, int T,
                                 int A, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (t >= T || a >= A) return;
  double acc = 0.0;
  for (int p = 0; p < P; ++p) {
    if (label[p] != a) continue;
    const double v = data[(long)p * T + t] * weight[p];
    acc += (v != v) ? 0.0 : v;
  }
  out[(long)a * T + t] = acc;
}

// ---- per-pixel linear detrending, every cut-off year in one launch (behaviour of north/June1st.py:179-194 and
// north/retrospective_forecasts/June1st_retro.py:178-195: scipy.stats.linregress per pixel in a Python double loop) --------
// data [P][T]; cut c uses the first ncut[c] time steps.  dt_out: cut c's detrended series at dt_off[c] + p*ncut[c]; trend_out
// [ncuts][P][2] = slope, intercept.  One thread per (pixel, cut).  A pixel with any NaN inside the window comes out all-NaN
// (linregress propagates it) -- plain IEEE arithmetic does the same.
__global__ void detrend_kernel(const double* __restrict__ data, int P, int T, const int* __restrict__ ncut, const long* __restrict__ dt_off,
                               int ncuts, double* __restrict__ dt_out, double* __restrict__ trend_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (p >= P || c >= ncuts) return;
  const int n = ncut[c];
  const double* y = data + (long)p * T;
  const double tm = 0.5 * (double)(n - 1);
  double ym = 0.0;
  for (int k = 0; k < n; ++k) ym += y[k];
  ym /= (double)n;
  double sxx = 0.0, sxy = 0.0;
  for (int k = 0; k < n; ++k) { const double dt = (double)k - tm; sxx += dt * dt; sxy += dt * (y[k] - ym); }
  const double slope = (sxy / (double)n) / (sxx / (double)n);
  const double icpt = ym - slope * tm;
  double* o = dt_out + dt_off[c] + (long)p * n;
  for (int k = 0; k < n; ++k) o[k] = y[k] - (slope * (double)k + icpt);
  trend_out[((long)c * P + p) * 2] = slope;
  trend_out[((long)c * P + p) * 2 + 1] = icpt;
}

// ---- fp32 factor + fp64 iterative refinement (BASELINE configs[4]) -----------------------------------
// fp64 residual of the refinement, with the covariance recomputed on the fly (no n x n fp64 matrix is ever stored):
//   Rout[r][i] = Bq[r][i] - sum_j ( k~(x_i, x_j) + sn [i==j] ) Xq[r][j]        r < nrhs <= 4, i < n
// Bq row r is: r == 0 -> y, r >= 1 -> k~(xs_{r-1}, .) (recomputed).  One thread per i (x_i in registers, DREG
// features), x_j tiles of 64 rows broadcast from LDS.  The j range is cut into gridDim.y chunks of jlen columns
// (krefine_residual_kernel writes the partial sums, krefine_finish_kernel adds them in chunk order and subtracts from Bq): one
// thread per row alone is 128 workgroups at n = 32768 -- half the CUs idle, one wave per SIMD -- and took 18.3 ms per residual.
template <int DREG>
__global__ __launch_bounds__(256) void krefine_residual_kernel(const double* __restrict__ X, int dp, int n, int nrhs,
                                                               const double* __restrict__ Xq, long ldq, double* __restrict__ part,
                                                               long ldp, int jlen, KParams kp, int i0 = 0, int i1 = -1) {
  __shared__ double Xj[64][DREG + 1];
  __shared__ double Q[4][64];
  const int iend = i1 < 0 ? n : i1;                    // rows [i0, iend): one rank's share of a residual sharded by rows
  const int i = i0 + blockIdx.x * 256 + threadIdx.x;
  double xi[DREG];
#pragma unroll
  for (int p = 0; p < DREG; ++p) xi[p] = (i < iend) ? X[(long)i * dp + p] : 0.0;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int jbeg = blockIdx.y * jlen, jend = min(n, jbeg + jlen);
  for (int j0 = jbeg; j0 < jend; j0 += 64) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * DREG; idx += 256) {
      const int jj = idx / DREG, p = idx % DREG;
      Xj[jj][p] = (j0 + jj < jend) ? X[(long)(j0 + jj) * dp + p] : 0.0;
    }
    {
      const int r = threadIdx.x >> 6, jj = threadIdx.x & 63;
      Q[r][jj] = (r < nrhs && j0 + jj < jend) ? Xq[(long)r * ldq + j0 + jj] : 0.0;
    }
    __syncthreads();
    const int jn = min(64, jend - j0);
    for (int jj = 0; jj < jn; ++jj) {
      double sq = 0.0;
#pragma unroll
      for (int p = 0; p < DREG; ++p) { const double t = xi[p] - Xj[jj][p]; sq = fma(t, t, sq); }
      double kv = cov_from_sq(kp, sq);
      if (j0 + jj == i) kv += kp.sn;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fma(kv, Q[r][jj], acc[r]);
    }
  }
  if (i >= iend) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) part[((long)blockIdx.y * 4 + r) * ldp + i] = acc[r];
}
// The same partial sums from a STORED fp64 K~ -- the lower triangle the covariance build wrote beside the fp32 matrix (kbuild_kernel's
// Mat64; the diagonal 128-blocks are complete) -- in two passes over it, both coalesced:
//   columns: thread i walks column i downwards, sum_{j >= i} K~[j][i] x[j] (= the upper part of row i by symmetry, and the diagonal);
//            a wavefront reads 64 consecutive doubles of row j.  Rows in gridDim.y chunks of jlen, partial (chunk, r, i) as above.
//   rows:    one wavefront per row i, lanes along it, sum_{j < i} K~[i][j] x[j]; one more partial, index `chunk_out`.
// One 8-byte load + nrhs FMAs per element: HBM-bound (8 n^2 bytes per residual) where the on-the-fly kernel is bound by its d
// subtract-multiplies + one exp per element (5.3 ms at n = 32768, d = 32).
__global__ __launch_bounds__(256) void kres_lower_cols_kernel(const double* __restrict__ Kq, long ldk, int n, int nrhs,
                                                              const double* __restrict__ Xq, long ldq, double* __restrict__ part,
                                                              long ldp, int jlen) {
  __shared__ double Q[4][256];
  const int ib = blockIdx.x * 256, i = ib + threadIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int jbeg = blockIdx.y * jlen, jend = min(n, jbeg + jlen);
  if (jend > ib) {
    const double* col = Kq + (i < n ? i : 0);
    for (int j0 = max(jbeg, ib & ~255); j0 < jend; j0 += 256) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) Q[r][threadIdx.x] = (r < nrhs && j0 + (int)threadIdx.x < jend) ? Xq[(long)r * ldq + j0 + threadIdx.x] : 0.0;
      __syncthreads();
      const int jn = min(256, jend - j0);
      if (j0 >= ib + 256) {            // wholly below this block's diagonal: every row counts for every thread
        int jj = 0;
        for (; jj + 8 <= jn; jj += 8) {
          double kv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) kv[u] = col[(long)(j0 + jj + u) * ldk];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(kv[u], Q[r][jj + u], acc[r]);
        }
        for (; jj < jn; ++jj) {
          const double kv = col[(long)(j0 + jj) * ldk];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fma(kv, Q[r][jj], acc[r]);
        }
      } else {
        for (int jj = 0; jj < jn; ++jj) {
          if (j0 + jj < i || i >= n) continue;
          const double kv = col[(long)(j0 + jj) * ldk];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fma(kv, Q[r][jj], acc[r]);
        }
      }
    }
  }
  if (i >= n) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) part[((long)blockIdx.y * 4 + r) * ldp + i] = acc[r];
}
__global__ __launch_bounds__(256) void kres_lower_rows_kernel(const double* __restrict__ Kq, long ldk, int n, int nrhs,
                                                              const double* __restrict__ Xq, long ldq, double* __restrict__ part,
                                                              long ldp, int chunk_out) {
  constexpr int JT = 1024;                             // columns per staged piece of x: 16 loads in flight per lane between two barriers (was 4)
  __shared__ double Q[4][JT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;                 // this wavefront's row
  const int iend = min(n, (int)blockIdx.x * 4 + 4);    // the workgroup's rows are [4 blockIdx.x, iend): columns below iend - 1 matter to someone
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double* row = Kq + (long)(i < n ? i : 0) * ldk;
  for (int j0 = 0; j0 < iend - 1; j0 += JT) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < JT / 256; ++q) {
        const int j = j0 + q * 256 + (int)threadIdx.x;
        Q[r][q * 256 + threadIdx.x] = (r < nrhs && j < n) ? Xq[(long)r * ldq + j] : 0.0;
      }
    __syncthreads();
    if (i < n) {
      double kv[JT / 64];
#pragma unroll
      for (int u = 0; u < JT / 64; ++u) {
        const int j = j0 + lane + 64 * u;
        kv[u] = j < i ? row[j] : 0.0;                  // (same terms in the same order as before: u ascending within a piece, pieces ascending)
      }
#pragma unroll
      for (int u = 0; u < JT / 64; ++u) {
        if (j0 + lane + 64 * u < i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fma(kv[u], Q[r][lane + 64 * u], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
  if (i < n && lane == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[((long)chunk_out * 4 + r) * ldp + i] = acc[r];
  }
}
// The same residual in ONE pass over the stored lower triangle (the two kernels above read it twice: 8 n^2 bytes per residual, this
// one 4 n^2).  Workgroup (S, c) owns rows [256 S, +256) x columns [c jlen, +jlen) of the triangle and walks it in 64 x 64 tiles
// staged in LDS (the next tile's 32 KB already in flight in registers); every tile below the diagonal is used twice from LDS,
//   row sums     lane = row,    wave w the tile's columns 16 w ..: racc[row tile][r] += K[i][j] x[r][j]   (kept across the walk)
//   column sums  lane = column, wave w the tile's rows 16 w ..:    cacc[r]           += K[i][j] x[r][i]   (kept across the 4 row tiles)
// (a diagonal 64-tile is stored whole: row sums only), x through scalar loads (wave-uniform addresses).  No atomics -- every partial
// has its own slot and krefine_finish_kernel adds them in slot order:
//   slot c,        rows of S:             the row sums of (S, c)                       c <  nJ
//   slot nJ + S,   columns of chunk c:    the column sums of (S, c)                    S <  nS
// Entry i of a residual is the sum of slots c <= (256 S(i) + 255) / jlen and nJ + S, S >= S(i): exactly the slots written for it.
// jlen is a multiple of 64; x beyond n is zero and the padded rows / columns of K~ are stored (identity), so no masks.
constexpr int KSYM_LP = 65;
template <int NR>                // right-hand sides carried (1, 2 or 4 >= nrhs)
__global__ __launch_bounds__(256, 3) void kres_sym_kernel(const double* __restrict__ Kq, long ldk, int n_pad, int nrhs,
                                                       const double* __restrict__ Xq, long ldq, double* __restrict__ part,
                                                       long ldp, int jlen, int nJ) {
  const int S = blockIdx.x, c = blockIdx.y;
  if ((long)c * jlen > (long)S * 256 + 255) return;
  __shared__ double T[64 * KSYM_LP];
  __shared__ double red[4][4][64];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int It0 = 4 * S, nIt = min(4, n_pad / 64 - It0);          // row tiles It0 .. It0 + nIt - 1
  const int Jt0 = c * (jlen / 64), Jt1 = min(min((c + 1) * (jlen / 64), It0 + nIt), n_pad / 64);
  const int lrow = t >> 5, lcp = (t & 31) * 2;                    // the thread's piece of a tile: rows lrow + 8 k, columns lcp, lcp + 1
  const int Ilast = n_pad / 64 - 1;
  d2 pre[8];
  double pxJ[NR], pxI[NR];                                        // x of the tile's columns / rows, lane l = entry l: fetched with the tile
  // every (row tile, J) of the band is walked -- a tile right of the diagonal (the band's diagonal 256-block only) or below the
  // matrix is fetched from a valid address and zeroed on its way into LDS: straight-line code, 0.6 % more bytes at n = 32768
  auto issue = [&](int I, int J) {
    const int Ic = min(I, Ilast), Jc = min(J, Ic);
    const double* src = Kq + ((long)Ic * 64 + lrow) * ldk + (long)Jc * 64 + lcp;
#pragma unroll
    for (int k = 0; k < 8; ++k) pre[k] = *(const d2*)(src + (long)(8 * k) * ldk);
#pragma unroll
    for (int r = 0; r < NR; ++r) { pxJ[r] = Xq[(long)r * ldq + (long)J * 64 + lane]; pxI[r] = Xq[(long)r * ldq + (long)Ic * 64 + lane]; }
  };
  double racc[4][NR], cacc[NR];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < NR; ++r) racc[a][r] = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) cacc[r] = 0.0;
  const int ntile = 4 * (Jt1 - Jt0);                              // the walk: J ascending, the band's four row tiles under each J
  if (ntile > 0) issue(It0, Jt0);
#pragma unroll 1
  for (int q = 0; q < ntile; ++q) {
    const int a = q & 3, J = Jt0 + (q >> 2), I = It0 + a;
    const bool live = a < nIt && J <= I;                          // (uniform)
    __syncthreads();                                              // the last tile's readers are done
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      T[(lrow + 8 * k) * KSYM_LP + lcp] = live ? pre[k].x : 0.0;
      T[(lrow + 8 * k) * KSYM_LP + lcp + 1] = live ? pre[k].y : 0.0;
    }
    double xJ[NR], xI[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { xJ[r] = pxJ[r]; xI[r] = pxI[r]; }
    __syncthreads();
    if (q + 1 < ntile) issue(a < 3 ? I + 1 : It0, a < 3 ? J : J + 1);     // the next tile of the walk
    {
      double rt[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) rt[r] = 0.0;
      const double* tr = T + lane * KSYM_LP + 16 * w;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const double kv = tr[jj];
#pragma unroll
        for (int r = 0; r < NR; ++r) rt[r] = fma(kv, readlane_t(xJ[r], 16 * w + jj), rt[r]);
      }
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)                              // (registers, not an indexed array: + 0.0 is exact)
#pragma unroll
        for (int r = 0; r < NR; ++r) racc[aa][r] += (aa == a) ? rt[r] : 0.0;
    }
    if (J < I) {                                                  // (a diagonal tile is stored whole: its row sums are everything)
      const double* tc = T + (16 * w) * KSYM_LP + lane;
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) {
        const double kv = tc[ii * KSYM_LP];
#pragma unroll
        for (int r = 0; r < NR; ++r) cacc[r] = fma(kv, readlane_t(xI[r], 16 * w + ii), cacc[r]);
      }
    }
    if (a == 3) {
      // the column sums of tile column J over this workgroup's rows: waves in order, one slot entry each
#pragma unroll
      for (int r = 0; r < NR; ++r) { red[w][r][lane] = cacc[r]; cacc[r] = 0.0; }
      __syncthreads();
      const int r = t >> 6;
      if (r < nrhs) part[((long)(nJ + S) * 4 + r) * ldp + (long)J * 64 + lane] = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    }
  }
  // the row sums: four waves' shares, through the tile's LDS
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < NR; ++r) T[((w * 4 + a) * 4 + r) * 64 + lane] = racc[a][r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * 256 + t, a = idx >> 8, r = (idx >> 6) & 3;   // (a, r, lane)
    if (a < nIt && r < nrhs) {
      const double v = ((T[((0 * 4 + a) * 4 + r) * 64 + lane] + T[((1 * 4 + a) * 4 + r) * 64 + lane]) + T[((2 * 4 + a) * 4 + r) * 64 + lane]) +
                       T[((3 * 4 + a) * 4 + r) * 64 + lane];
      part[((long)c * 4 + r) * ldp + (long)(It0 + a) * 64 + lane] = v;
    }
  }
}
template <int DREG>
__global__ __launch_bounds__(256) void krefine_finish_kernel(const double* __restrict__ X, const double* __restrict__ Xs,
                                                             const double* __restrict__ y, int dp, int n, int nrhs,
                                                             const double* __restrict__ part, long ldp, int nchunk,
                                                             double* __restrict__ Rout, long ldr, KParams kp, int i0 = 0, int i1 = -1,
                                                             int sym_jlen = 0, int sym_nS = 0) {
  const int i = i0 + blockIdx.x * 256 + threadIdx.x;
  if (i >= (i1 < 0 ? n : i1)) return;
  for (int r = blockIdx.y; r < nrhs; r += gridDim.y) {          // (gridDim.y = 1: every right-hand side in turn)
    double b;
    if (r == 0) {
      b = y[i];
    } else {
      const double* xs = Xs + (long)(r - 1) * dp;
      const double* xi = X + (long)i * dp;
      double sq = 0.0;
#pragma unroll
      for (int p = 0; p < DREG; ++p) { const double t = xs[p] - xi[p]; sq = fma(t, t, sq); }
      b = cov_from_sq(kp, sq);
    }
    double a = 0.0;
    if (sym_nS > 0) {            // kres_sym_kernel's slots for entry i (nchunk = nJ)
      const int Si = i >> 8, cmax = min(nchunk - 1, (int)(((long)Si * 256 + 255) / sym_jlen));
      for (int c = 0; c <= cmax; ++c) a += part[((long)c * 4 + r) * ldp + i];
      for (int S = Si; S < sym_nS; ++S) a += part[((long)(nchunk + S) * 4 + r) * ldp + i];
    } else {
      for (int c = 0; c < nchunk; ++c) a += part[((long)c * 4 + r) * ldp + i];
    }
    Rout[(long)r * ldr + i] = b - a;
  }
}

// out[r] = max_i |R[r][i]| (r < nrhs), out[nrhs] = max_i |y[i]|: what the refinement's stopping test and its reported residual need --
// 8 doubles to the host instead of the residual rows (pageable copies of (1 + m) n doubles cost 0.8 ms per test at n = 32768)
__global__ __launch_bounds__(1024) void refine_norms_kernel(const double* __restrict__ R, long ldr, int n, int nrhs,
                                                            const double* __restrict__ y, double* __restrict__ out) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  const double* src = b < nrhs ? R + (long)b * ldr : y;
  // max |.| that PROPAGATES NaN (fmax drops it): a non-finite residual -- a member whose factor was poisoned by a failed pivot -- must fail
  // the refinement's stopping test and show up in refine_resid, not read as 0
  auto nmax = [](double a, double v) { return (v != v || a != a) ? __builtin_nan("") : fmax(a, v); };
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) m = nmax(m, fabs(src[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = nmax(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = nmax(m, sh[k]);
    out[b] = m;
  }
}

// out[k] = sum_b part[b][k], k < 8: krefine_dots_kernel's block partials, added in block order (one wave)
__global__ __launch_bounds__(64) void sum_parts8_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  if (threadIdx.x >= 8) return;
  double a = 0.0;
  for (int b = 0; b < nblk; ++b) a += part[(long)b * 8 + threadIdx.x];
  out[threadIdx.x] = a;
}

// dst[r][i] (TO) <- src[r][i] (TI), rows x cols, different strides; zero beyond cols_valid
template <typename TI, typename TO>
__global__ void convert_rows_kernel(const TI* __restrict__ src, long lds, TO* __restrict__ dst, long ldd, int rows, int cols,
                                    int cols_valid) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx % cols);
  dst[(long)r * ldd + c] = (c < cols_valid) ? (TO)src[(long)r * lds + c] : (TO)0;
}

// Xacc[r][i] += (double) D[r][i]
template <typename TI>
__global__ void accumulate_rows_kernel(const TI* __restrict__ D, long ldd, double* __restrict__ Xacc, long ldx, int rows, int cols) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx % cols);
  Xacc[(long)r * ldx + c] += (double)D[(long)r * ldd + c];
}

// final fp64 dots of the refined solutions: out[r] = B_r . Xq_0 (r=0: y.alpha~; r>=1: k*_r . alpha~), out[4+r] = B_r . Xq_r
// (k*_r . w_r), out[8] = max_r ||R_r||_inf proxy is computed on the host from the residual rows
template <int DREG>
__global__ __launch_bounds__(256) void krefine_dots_kernel(const double* __restrict__ X, const double* __restrict__ Xs,
                                                           const double* __restrict__ y, int dp, int n, int nrhs,
                                                           const double* __restrict__ Xq, long ldq, double* __restrict__ part,
                                                           KParams kp) {
  __shared__ double sh[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double xi[DREG];
#pragma unroll
  for (int p = 0; p < DREG; ++p) xi[p] = (i < n) ? X[(long)i * dp + p] : 0.0;
  for (int r = 0; r < nrhs; ++r) {
    double b = 0.0;
    if (i < n) {
      if (r == 0) b = y[i];
      else {
        const double* xs = Xs + (long)(r - 1) * dp;
        double sq = 0.0;
#pragma unroll
        for (int p = 0; p < DREG; ++p) { const double t = xs[p] - xi[p]; sq = fma(t, t, sq); }
        b = cov_from_sq(kp, sq);
      }
    }
    double v0 = (i < n) ? b * Xq[i] : 0.0;
    double v1 = (i < n) ? b * Xq[(long)r * ldq + i] : 0.0;
    v0 = block_reduce_sum(v0, sh);
    v1 = block_reduce_sum(v1, sh);
    if (threadIdx.x == 0) { part[(long)blockIdx.x * 8 + r] = v0; part[(long)blockIdx.x * 8 + 4 + r] = v1; }
  }
}

// mean of new test points against the refined alpha~ (fp64): out[r] = sum_i k~(xs_r, x_i) alpha_i ; one block per point
__global__ __launch_bounds__(256) void cross_mean_kernel(const double* __restrict__ X, const double* __restrict__ Xs, int dp, int d, int n,
                                                         const double* __restrict__ alpha, double* __restrict__ out, KParams kp) {
  __shared__ double sh[4];
  const double* xs = Xs + (long)blockIdx.x * dp;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double* xi = X + (long)i * dp;
    double sq = 0.0;
    for (int p = 0; p < d; ++p) { const double t = xs[p] - xi[p]; sq = fma(t, t, sq); }
    a = fma(cov_from_sq(kp, sq), alpha[i], a);
  }
  a = block_reduce_sum(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

// pivot info of one rank as a double for the MIN all-reduce over the ranks (0 = none -> a huge value)
__global__ void info_to_double_kernel(const int* __restrict__ info, double* __restrict__ out) { out[0] = info[0] != 0 ? (double)info[0] : 1e18; }

// ---- panel-major work vectors of the sharded triangular solves (sigp_dist_fit fp32 refinement, sigp_dist_predict) ------------------------------
// dst [panel][TS_RHS][pw] (float) <- src [rows][lds] (double), zero beyond `cols_valid` and for right-hand sides >= rows
template <typename T>
__global__ void pm_from_rows_kernel(const double* __restrict__ src, long lds, T* __restrict__ dst, int rows, int n_pad, int cols_valid, int pw) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)TS_RHS * n_pad) return;
  const int r = (int)(idx / n_pad), i = (int)(idx % n_pad);
  dst[((long)(i / pw) * TS_RHS + r) * pw + i % pw] = (r < rows && i < cols_valid) ? (T)src[(long)r * lds + i] : (T)0;
}
// Xacc [rows][ldx] (double) (+)= src panel-major (float)
template <typename T>
__global__ void pm_to_rows_kernel(const T* __restrict__ src, double* __restrict__ Xacc, long ldx, int rows, int n_pad, int pw, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * n_pad) return;
  const int r = (int)(idx / n_pad), i = (int)(idx % n_pad);
  const double v = (double)src[((long)(i / pw) * TS_RHS + r) * pw + i % pw];
  double* o = Xacc + (long)r * ldx + i;
  *o = accumulate ? *o + v : v;
}

// sigp_dist_predict: cross-covariances of up to TS_RHS test points with every training point, panel-major:
// dst[(i / pw) TS_RHS + r][i % pw] = k~(xs_r, x_i) for i < n, r < nrhs; 0 otherwise
template <typename T>
__global__ void cross_cov_pm_kernel(const double* __restrict__ X, const double* __restrict__ Xs, int dp, int d, int n, int n_pad, int nrhs, T* __restrict__ dst, int pw, KParams kp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)TS_RHS * n_pad) return;
  const int r = (int)(idx / n_pad), i = (int)(idx % n_pad);
  double v = 0.0;
  if (r < nrhs && i < n) {
    const double* xs = Xs + (long)r * dp;
    const double* xi = X + (long)i * dp;
    double sq = 0.0;
    for (int p = 0; p < d; ++p) { const double t = xs[p] - xi[p]; sq = fma(t, t, sq); }
    v = cov_from_sq(kp, sq);
  }
  dst[((long)(i / pw) * TS_RHS + r) * pw + i % pw] = (T)v;
}
// out[r] = sum over this rank's columns of Z[r][c]^2 (r < nrhs; one block per right-hand side, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void rowsq_kernel(const T* __restrict__ Z, long ldz, int ncol, double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (int c = threadIdx.x; c < ncol; c += 256) { const double z = (double)Z[(long)blockIdx.x * ldz + c]; a += z * z; }
  a = block_reduce_sum(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

}  // namespace sigp
