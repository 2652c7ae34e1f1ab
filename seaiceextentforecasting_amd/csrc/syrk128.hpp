// Trailing-update kernel v2 (the >99 %-of-flops kernel of the Cholesky, north/June1st.py:265):
//   C[128x128 tile] -= A[128 x K] * B[128 x K]^T      fp64, v_mfma_f64_16x16x4_f64
// Differences from the generic gemm_mfma_kernel:
//   * A/B K-slices go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction): no staging
//     VGPRs, no ds_write, no VALU in the staging path;
//   * LDS image is [row][8 x 16-B k-pair slots] with slot' = slot ^ ((row>>1)&7) (XOR swizzle applied on the
//     SOURCE address of the DMA and on the fragment read): ds_read_b128 fragment reads are bank-conflict free
//     with no padding (64 KiB per workgroup -> 2 workgroups per CU);
//   * one 16-B read feeds two MFMA k-steps (lane lq holds k = 2*lq, 2*lq+1 of an 8-wide k-group; A and B use
//     the same k permutation, so the sum over k is unchanged);
//   * the sign is folded into the accumulator (acc = -C; acc += A B^T; C = -acc) so A needs no negation.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"

namespace sigp {

constexpr int SY_T = 128;                      // tile
constexpr int SY_LDS_BYTES = 2 * 2 * SY_T * KT * (int)sizeof(double);   // 2 buffers x (A,B) x 128 rows x 16 k

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// SET = false:  C -= A B^T   (trailing / inner updates)
// SET = true :  C  = A B^T   (panel solve L21 = A21 inv(L11)^T, in place: a tile reads only its own rows of A = C)
template <bool SET>
__global__ __launch_bounds__(256, 2) void syrk128_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                       // [2][128][16]
  double* Bs = smem + 2 * SY_T * KT;       // [2][128][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;

  int bi, bj;
  if (!gemm_tile_coords(g, blockIdx.x, bi, bj)) return;
  const long bz = blockIdx.y;
  const double* Ag = g.A + bz * g.sA + (long)bi * SY_T * g.lda;
  const double* Bg = g.B + bz * g.sB + (long)bj * SY_T * g.ldb;
  double* Cg = g.C + bz * g.sC + ((long)bi * SY_T + wm * 64) * g.ldc + (long)bj * SY_T + wn * 64;

  unsigned long long st_c0 = 0, st_r0 = 0;
  if (g.stamp) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = SET ? 0.0 : -Cg[(long)(i * 16 + lq + 4 * r) * g.ldc + j * 16 + lr];

  // DMA coordinates: per wave-instruction 8 rows x 128 B; lane -> (row = lane>>3, slot' = lane&7)
  const int drow = wave * 8 + (lane >> 3);                 // + 32*p
  const int dks = ((lane & 7) ^ ((drow >> 1) & 7)) * 2;    // source k offset (doubles) of this lane's 16 B
  const double* Asrc = Ag + (long)drow * g.lda + dks;
  const double* Bsrc = Bg + (long)drow * g.ldb + dks;
  const long a32 = 32 * g.lda, b32 = 32 * g.ldb;

#define SY_ISSUE(k0, buf)                                                                                     \
  {                                                                                                           \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                           \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Asrc + p * a32 + (k0)),                                    \
                                       (lds_ptr_t)(As + ((buf) * SY_T + p * 32 + wave * 8) * KT), 16, 0, 0);  \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bsrc + p * b32 + (k0)),                                    \
                                       (lds_ptr_t)(Bs + ((buf) * SY_T + p * 32 + wave * 8) * KT), 16, 0, 0);  \
    }                                                                                                         \
  }

  // fragment read offsets (doubles) inside a row for k-group h = 0 / 1
  const int x = (lr >> 1) & 7;
  const int fo0 = ((lq ^ x) & 7) * 2, fo1 = (((4 + lq) ^ x) & 7) * 2;
  const int arow0 = (wm * 64 + lr) * KT, brow0 = (wn * 64 + lr) * KT;

  // Software pipeline (2 LDS buffers, one barrier per K-slice, placed MID-slice):
  //   read group-1 fragments of slice s | MFMA group 0 | barrier (slice s+1 landed, slice s fully read)
  //   | DMA slice s+2 into the buffer just freed, read group-0 fragments of slice s+1 | MFMA group 1
  // so every fragment read and every DMA has a 32-MFMA group (2048 cycles) to land behind.
  const int nst = g.K / KT;
  d2 a0[4], b0[4], a1[4], b1[4];
  SY_ISSUE(0, 0);
  __syncthreads();
  if (nst > 1) SY_ISSUE(KT, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const d2*)(As + arow0 + i * 16 * KT + fo0);
#pragma unroll
  for (int j = 0; j < 4; ++j) b0[j] = *(const d2*)(Bs + brow0 + j * 16 * KT + fo0);
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    const double* Ab = As + buf * SY_T * KT + arow0;
    const double* Bb = Bs + buf * SY_T * KT + brow0;
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = *(const d2*)(Ab + i * 16 * KT + fo1);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = *(const d2*)(Bb + j * 16 * KT + fo1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nst && !(g.dbg & 1)) SY_ISSUE((s + 2) * KT, buf);
    if (s + 1 < nst) {
      const double* An = As + (buf ^ 1) * SY_T * KT + arow0;
      const double* Bn = Bs + (buf ^ 1) * SY_T * KT + brow0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = *(const d2*)(An + i * 16 * KT + fo0);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = *(const d2*)(Bn + j * 16 * KT + fo0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i][e], b1[j][e], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SY_ISSUE

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cg[(long)(i * 16 + lq + 4 * r) * g.ldc + j * 16 + lr] = SET ? acc[i][j][r] : -acc[i][j][r];
  if (g.stamp && tid == 0) {
    g.stamp[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = __builtin_amdgcn_s_memtime() - st_c0;
    g.stamp[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
}

}  // namespace sigp
