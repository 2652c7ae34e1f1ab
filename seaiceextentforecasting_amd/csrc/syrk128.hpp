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
#include <type_traits>

#include "gemm_mfma.hpp"

namespace sigp {

constexpr int SY_T = 128;                      // tile
constexpr int SY_SLICE_BYTES = 128;            // K-slice per row: 16 doubles / 32 floats
constexpr int SY_LDS_BYTES = 2 * 2 * SY_T * SY_SLICE_BYTES;   // 2 buffers x (A,B) x 128 rows x 128 B = 64 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// SET = false:  C -= A B^T   (trailing / inner updates)
// SET = true :  C  = A B^T   (panel solve L21 = A21 inv(L11)^T, in place: a tile reads only its own rows of A = C)
// T = double: v_mfma_f64_16x16x4_f64, one 16-byte fragment read feeds 2 k-steps; T = float: v_mfma_f32_16x16x4_f32,
// one read feeds 4 k-steps (same LDS / DMA bytes per MFMA-cycle, so the pipeline balance is the same).
// One 128x128 tile:  C (-)= A[128 x K] B[128 x K]^T.  Ag / Bg point at the tile's first A / B row, Cw at THIS WAVE's
// 64x64 quadrant of the C tile.  All 256 threads of the workgroup call it together (it contains barriers); As / Bs are
// the workgroup's LDS staging buffers.  On return every wave has issued its C stores (not yet waited for).
// TRI (panel_strip_kernel only): the LAST 128 columns of B are a lower-triangular block (B[c][k] = 0 for k > c: an inverse diagonal
// block), so a 16-column tile of C needs only the k-groups up to its last column.  The wave's four 16-column tiles are then INTERLEAVED
// (wave wn owns tiles wn, wn + 2, wn + 4, wn + 6 of the 128 columns instead of four consecutive ones: Cw points at column 16 wn) so that
// both column waves of a row pair skip about the same share, and slice s' = 0 .. 7 of that K block issues the MFMAs of its tiles
// jj >= ceil((s' - wn) / 2) only: 16 / 20 of 32 tile-slices for wn = 0 / 1.  The skipped products are exact zeros: same bits.
// DG (diagonal tiles of a symmetric update, C = C - A A^T with the same rows on both sides; only the lower half of such a tile is ever
// read): BOTH the wave's four 16-row tiles and its four 16-column tiles are interleaved (row tiles wm, wm + 2, ..; column tiles wn, wn + 2, ..:
// Cw points at row 16 wm, column 16 wn) and the wave loads, multiplies and stores only the 16 x 16 pairs on or below the diagonal:
// DG = 1 (waves 0, 2, 3) pairs j <= i, 10 of 16; DG = 2 (wave 1: row tiles even, column tiles odd) pairs j < i, 6 of 16.  36 of 64 pairs
// per tile; the pairs above the diagonal are left as they are in memory.  The kept pairs see the same products in the same order.
// RD (tiles of the ride-along block row when at most 16 of its 128 rows are in use -- y and up to 15 test points; the rest are zero rows):
// only the first 16-row tile takes part.  RD = 1 (the row waves wm = 0): pairs i = 0, and of A only rows 0 .. 15 are staged; RD = 2 (wm = 1):
// no pairs, no A rows -- the wave stages its share of B and keeps the barriers.  Rows 16 .. 127 of C are neither read nor written.
template <typename T, bool SET, int ABL = 0, bool TRI = false, int WNC = 0, int DG = 0, int RD = 0>
__device__ __forceinline__ void syrk128_tile(const T* Ag, long lda, const T* Bg, long ldb, T* Cw,
                                             long ldc, int K, int dbg_in, T* As, T* Bs, bool stamp_in, unsigned long long& ph0, unsigned long long& ph1,
                                             int tid_in = -1) {
  static_assert(!TRI || (SET && sizeof(T) == 8), "TRI: fp64 strip solves only (C is not read)");
  static_assert(!DG || (!TRI && !SET), "DG: diagonal tiles of the trailing update");
  constexpr int IS = DG ? 32 : 16;                  // rows between a wave's consecutive 16-row tiles
  constexpr int JS = (TRI || DG) ? 32 : 16;         // columns between a wave's consecutive 16-column tiles
  static_assert(!RD || (!TRI && !SET && !DG), "RD: ride-row tiles of the trailing update");
#define SY_NEED(i, j) (RD ? (RD == 1 && (i) == 0) : (DG == 0 || (DG == 1 ? (j) <= (i) : (j) < (i))))
  const int dbg = dbg_in & DBG_MASK;            // ablation bits: debug library only
  const bool stamp = stamp_in && DBG_MASK != 0;
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v16_t v16_t;
  constexpr int KTe = N_::KT, NE = N_::NE;
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar
  const int wm = wave >> 1, wn = TRI ? WNC : (wave & 1);      // TRI: the caller instantiates the tile once per column wave (WNC)
  const int lr = lane & 15, lq = lane >> 4;
  // fp32: the MFMA is issued with its operands swapped, D' = B_j A_i^T = (A_i B_j^T)^T, so that a lane's four accumulator registers are
  // four CONSECUTIVE COLUMNS of one row of C (row lr, columns 4 lq .. 4 lq + 3: v_mfma_f32_16x16x4_f32 puts row 4 lq + r, column lr of
  // ITS product in register r) -- one 16-byte load and one 16-byte store per accumulator instead of four 4-byte ones (a quarter of the
  // vector-memory instructions of the prologue and the epilogue, which are issue-bound).  The products and their order are the same:
  // results are bit-identical to the untransposed form.  (fp64 puts rows lq + 4 r in a lane: nothing contiguous either way.)
  constexpr bool TRANSPOSED_ACC = sizeof(T) == 4;

  acc_t acc[4][4];
  if (SET || (dbg & 8)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
  } else if (TRANSPOSED_ACC) {   // fp32: one 16-byte load per accumulator (see TRANSPOSED_ACC below)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!SY_NEED(i, j)) continue;
        const acc_t c = *(const acc_t*)(Cw + (long)(i * IS + lr) * ldc + j * JS + 4 * lq);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = -c[r];
      }
  } else if (DG || !(dbg & 128)) {   // accumulator-layout loads straight from global: 64 8-byte loads per lane
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (SY_NEED(i, j))
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = -Cw[(long)(i * IS + N_::drow(lq, r)) * ldc + j * JS + lr];
  } else {
    // (dbg 128) acc = -C through LDS: the C tile comes in by LDS-DMA (16 bytes per lane, one wave-instruction per 1 KiB)
    // in two halves of 64 rows -- the A/B staging buffers are idle before the K loop -- and each wave picks its
    // accumulator fragments out of LDS
    constexpr int CPR = SY_T / NE;                 // 16-byte chunks per C row (64 fp64 / 32 fp32)
    constexpr int RPI = 64 / CPR;                  // rows per wave-instruction (1 / 2)
    T* Cs = As;                                    // [64][128] image (fp64: all 64 KiB of As+Bs)
    const T* Ctile = Cw - (long)(wm * 64) * ldc - wn * 64;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh) __syncthreads();                     // fragment reads of the first half are done
#pragma unroll
      for (int q = 0; q < 16 / RPI; ++q) {
        const int R0 = wave * 16 + q * RPI;        // first row of this instruction inside the half
        const int R = R0 + (lane / CPR), ch = lane % CPR;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ctile + (long)(hh * 64 + R) * ldc + ((ch ^ N_::cswz(R)) * NE)),
                                         (lds_ptr_t)(Cs + R0 * SY_T), 16, 0, 0);
      }
      __syncthreads();
      if (wm == hh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int R = i * 16 + N_::drow(lq, r), col = wn * 64 + j * 16 + lr;
              acc[i][j][r] = -Cs[R * SY_T + (((col / NE) ^ N_::cswz(R)) * NE) + (col % NE)];
            }
      }
    }
    __syncthreads();                               // LDS is free for the A/B pipeline
  }

  // DMA coordinates: per wave-instruction 8 rows x 128 B; lane -> (row = lane>>3, slot' = lane&7)
  const int drow_ = wave * 8 + (lane >> 3);                  // + 32*p
  const int dks = ((lane & 7) ^ ((drow_ >> 1) & 7)) * NE;    // source k offset (elements) of this lane's 16 B
  const T* Asrc = Ag + (long)drow_ * lda + dks;
  const T* Bsrc = Bg + (long)drow_ * ldb + dks;
  const long a32 = 32 * lda, b32 = 32 * ldb;
#define SY_ISSUE(k0, buf)                                                                                     \
  {                                                                                                           \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                           \
      if (RD == 0 || (RD == 1 && p == 0))                                                                     \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Asrc + p * a32 + (k0)),                                  \
                                         (lds_ptr_t)(As + ((buf) * SY_T + p * 32 + wave * 8) * KTe), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bsrc + p * b32 + (k0)),                                    \
                                       (lds_ptr_t)(Bs + ((buf) * SY_T + p * 32 + wave * 8) * KTe), 16, 0, 0); \
    }                                                                                                         \
  }

  // fragment read offsets (elements) inside a row for k-group h = 0 / 1
  const int x = (lr >> 1) & 7;
  const int fo0 = ((lq ^ x) & 7) * NE, fo1 = (((4 + lq) ^ x) & 7) * NE;
  const int arow0 = ((DG ? wm * 16 : wm * 64) + lr) * KTe, brow0 = ((TRI || DG ? wn * 16 : wn * 64) + lr) * KTe;

  // Software pipeline (2 LDS buffers, one barrier per K-slice, placed MID-slice):
  //   MFMA group 0 with the 8 group-1 fragment reads of slice s spread through it | barrier (slice s+1 landed, slice s fully read)
  //   | MFMA group 1 with the 8 DMA instructions of slice s+2 (into the buffer just freed) and the 8 group-0 fragment reads of
  //   slice s+1 spread through it
  // so every fragment read and every DMA has most of an MFMA group to land behind AND costs the wave no issue time of its own:
  // as bursts between the groups the loads cost the fp32 kernel 14 % and the fp64 kernel 4-5 % (A/B with the DMA switched off).
  const int nst = K / KTe;
  v16_t a0[4], b0[4], a1[4], b1[4];
  SY_ISSUE(0, 0);
  __syncthreads();
  if (stamp) ph0 = __builtin_amdgcn_s_memtime();   // C tile and first K-slice have landed
  if (nst > 1) SY_ISSUE(KTe, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(As + arow0 + i * IS * KTe + fo0);
#pragma unroll
  for (int j = 0; j < 4; ++j) b0[j] = *(const v16_t*)(Bs + brow0 + j * JS * KTe + fo0);
  // The loads sit INSIDE the MFMA groups (sched_group_barrier recipes: one load, G/8 MFMAs, ...).  The DMA is unconditional inside the
  // steady-state loop (a branch would cut the scheduling region); the last two slices run without it.  ABL (debug library only):
  // bit 0 no in-loop DMA, bit 1 no in-loop fragment reads, bit 2 no MFMA -- timing ablations as template arguments, so that the product kernel
  // and the debug library's ABL = 0 kernel are the same code.
  auto slice = [&](int s, auto with_dma, auto jmin_c) __attribute__((always_inline)) {
    constexpr int JMIN = decltype(jmin_c)::value;       // first 16-column tile of the wave this slice multiplies (0 unless TRI)
    constexpr int NM = NE * (RD == 1 ? 4 : RD == 2 ? 0 : DG == 1 ? 10 : DG == 2 ? 6 : 4 * (4 - JMIN));   // MFMAs per group
    constexpr int NP = NM < 8 ? NM : 8;                 // ... of which this many lead one load each
    const int buf = s & 1;
    const T* Ab = As + buf * SY_T * KTe + arow0;
    const T* Bb = Bs + buf * SY_T * KTe + brow0;
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = *(const v16_t*)(Ab + i * IS * KTe + fo1);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = *(const v16_t*)(Bb + j * JS * KTe + fo1);
    if (!(ABL & 4))
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = JMIN; j < 4; ++j)
          if (SY_NEED(i, j)) acc[i][j] = TRANSPOSED_ACC ? N_::mfma(b0[j][e], a0[i][e], acc[i][j]) : N_::mfma(a0[i][e], b0[j][e], acc[i][j]);
#pragma unroll
    for (int q = 0; q < NP; ++q) {                      // reads early, one per MFMA (the first MFMA leads: its operands' wait -- lgkmcnt(0) --
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // would otherwise wait for the first of these reads as well), then the rest of the group
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (NP < 8) __builtin_amdgcn_sched_group_barrier(0x100, 8 - NP, 0);
    if (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (decltype(with_dma)::value && !(ABL & 1)) SY_ISSUE((s + 2) * KTe, buf);
    if ((decltype(with_dma)::value || s + 1 < nst) && !(ABL & 2)) {
      const T* An = As + (buf ^ 1) * SY_T * KTe + arow0;
      const T* Bn = Bs + (buf ^ 1) * SY_T * KTe + brow0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(An + i * IS * KTe + fo0);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = *(const v16_t*)(Bn + j * JS * KTe + fo0);
    }
    if (!(ABL & 4))
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = JMIN; j < 4; ++j)
          if (SY_NEED(i, j)) acc[i][j] = TRANSPOSED_ACC ? N_::mfma(b1[j][e], a1[i][e], acc[i][j]) : N_::mfma(a1[i][e], b1[j][e], acc[i][j]);
    if (decltype(with_dma)::value && NM >= 16) {
      // the DMA first (it is waited for one group + one barrier later: every MFMA it is issued behind comes off its lead), then the reads
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      }
      if (NM > 16) __builtin_amdgcn_sched_group_barrier(0x008, NM - 16, 1);
    } else if (decltype(with_dma)::value && NM >= 8) {   // short groups (the six-pair diagonal wave): the DMA behind the first MFMAs, then the reads
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 1);
      if (NM > 8) __builtin_amdgcn_sched_group_barrier(0x008, NM - 8, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    typedef std::integral_constant<int, 0> J0_;
    int s = 0;
    if constexpr (TRI) {
      const int s_tri = nst - SY_T / KTe;             // first slice of the triangular K block (K is a multiple of 128 here)
      for (; s < s_tri; ++s) slice(s, std::true_type{}, J0_{});
      // the eight slices of the triangular block, straight-line, with the first tile of each a compile-time constant: ceil((s' - wn) / 2).
      // (A run-time switch inside a loop, or both column waves' sequences behind one branch, cost the register allocator ~1 KB of scratch.)
#define SY_TRI_STEP(SP) \
  slice(s_tri + SP, std::integral_constant<bool, (SP + 2 < 8)>{}, std::integral_constant<int, ((SP - WNC + 1) >> 1)>{});
      SY_TRI_STEP(0) SY_TRI_STEP(1) SY_TRI_STEP(2) SY_TRI_STEP(3) SY_TRI_STEP(4) SY_TRI_STEP(5) SY_TRI_STEP(6) SY_TRI_STEP(7)
#undef SY_TRI_STEP
    } else {
      for (; s + 2 < nst; ++s) slice(s, std::true_type{}, J0_{});
      for (; s < nst; ++s) slice(s, std::false_type{}, J0_{});
    }
  }
#undef SY_ISSUE
  if (stamp) ph1 = __builtin_amdgcn_s_memtime();   // K loop issued

  if (!(dbg & 16) || acc[0][0][0] == (T)12345.678) {   // dbg 16: timing ablation without the C store
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!SY_NEED(i, j)) continue;
        if (TRANSPOSED_ACC) {
          acc_t c;
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = SET ? acc[i][j][r] : -acc[i][j][r];
          *(acc_t*)(Cw + (long)(i * IS + lr) * ldc + j * JS + 4 * lq) = c;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) Cw[(long)(i * IS + N_::drow(lq, r)) * ldc + j * JS + lr] = SET ? acc[i][j][r] : -acc[i][j][r];
        }
      }
  }
#undef SY_NEED
}

// DIAG_SKIP: the launch is a symmetric update (lower tile space, same operand on both sides) and its diagonal tiles take the DG form of
// the tile -- 36 of the 64 16 x 16 pairs (launch_syrk128_t picks the instantiation); the tiles of block row g.ride_bi1 - 1 (the ride-along
// block, when at most 16 of its rows are in use) take the RD form -- 8 of 64 pairs.
template <typename T, bool SET, bool PERSIST = false, int ABL = 0, bool DIAG_SKIP = false>
__global__ __launch_bounds__(256, 2) void syrk128_kernel(GemmArgsT<T> g) {
  static_assert(!DIAG_SKIP || (!SET && !PERSIST && ABL == 0), "DIAG_SKIP: the product trailing update only");
  constexpr int KTe = Num<T>::KT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = (T*)smem_raw;                 // [2][128][KT]
  T* Bs = As + 2 * SY_T * KTe;          // [2][128][KT]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  unsigned long long* const stamp = DBG_MASK ? g.stamp : nullptr;
  // Persistent form (g.ntile > 0): the grid is a fixed number of resident workgroups, FEWER than the chip's 2-per-CU slots,
  // each walking the flattened (member, tile) list with a stride of the grid size.  The slots left open are what the
  // panel stream's latency-chain kernels (diagonal blocks, top-block solves) start in without waiting for an update
  // workgroup to retire.  Tiles cost the same, so the static deal is balanced to within one tile.
  const int total = PERSIST ? g.ntile * max(1, g.batch) : 1;
  for (int t = PERSIST ? (int)blockIdx.x : 0; t < total; t += PERSIST ? (int)gridDim.x : 1) {
    int bi, bj;
    long bz;
    if (PERSIST) {
      bz = t / g.ntile;
      if (!gemm_tile_coords(g, t - (int)bz * g.ntile, bi, bj)) continue;
    } else if (g.patch < 0) {
      if (!gemm_tile_coords_chunked(g, blockIdx.x, bi, bj, bz)) return;
    } else {
      bz = blockIdx.y;
      if (!gemm_tile_coords(g, blockIdx.x, bi, bj)) return;
    }
    const long zz = (long)blockIdx.z;
    const T* Ag = g.A + bz * g.sA + zz * g.zA + (long)bi * SY_T * g.lda;
    const T* Bg = g.B + bz * g.sB + zz * g.zB + (long)bj * SY_T * g.ldb;
    int K = g.K;
    if (g.ktri == 1) { const int ks = bi * SY_T; Ag += ks; Bg += ks; K -= ks; }   // upper-triangular operand rows: nothing left of the diagonal block
    T* Cw = g.C + bz * g.sC + zz * g.zC + ((long)bi * SY_T + wm * 64) * g.ldc + (long)bj * SY_T + wn * 64;
    // diagonal tile of a symmetric update (same rows of the same matrix on both sides): only its lower half is ever read
    const bool diag = DIAG_SKIP && g.lower && bi == bj && g.A == g.B && g.sA == g.sB && g.zA == g.zB && g.lda == g.ldb;
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (stamp) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long ph0 = 0, ph1 = 0;
    int tid_t = tid;
    if (PERSIST || DIAG_SKIP) asm volatile("" : "+v"(tid_t));   // opaque per tile: keeps the lane-dependent address arithmetic of the tile inside the
                                                   // loop (hoisted, it costs 18 VGPRs that do not exist: the kernel would spill to scratch)
    const bool ride = DIAG_SKIP && g.ride_bi1 > 0 && bi == g.ride_bi1 - 1;      // ride-along block row with at most 16 rows in use
    if (DIAG_SKIP && ride) {
      if (wm == 0) syrk128_tile<T, SET, ABL, false, 0, 0, DIAG_SKIP ? 1 : 0>(Ag, g.lda, Bg, g.ldb, Cw, g.ldc, K, g.dbg, As, Bs, stamp != nullptr, ph0, ph1, tid_t);
      else syrk128_tile<T, SET, ABL, false, 0, 0, DIAG_SKIP ? 2 : 0>(Ag, g.lda, Bg, g.ldb, Cw, g.ldc, K, g.dbg, As, Bs, stamp != nullptr, ph0, ph1, tid_t);
    } else if (DIAG_SKIP && diag) {
      T* Cd = g.C + bz * g.sC + zz * g.zC + ((long)bi * SY_T + wm * 16) * g.ldc + (long)bj * SY_T + wn * 16;
      if (wave == 1) syrk128_tile<T, SET, ABL, false, 0, DIAG_SKIP ? 2 : 0>(Ag, g.lda, Bg, g.ldb, Cd, g.ldc, K, g.dbg, As, Bs, stamp != nullptr, ph0, ph1, tid_t);
      else syrk128_tile<T, SET, ABL, false, 0, DIAG_SKIP ? 1 : 0>(Ag, g.lda, Bg, g.ldb, Cd, g.ldc, K, g.dbg, As, Bs, stamp != nullptr, ph0, ph1, tid_t);
    } else {
      syrk128_tile<T, SET, ABL>(Ag, g.lda, Bg, g.ldb, Cw, g.ldc, K, g.dbg, As, Bs, stamp != nullptr, ph0, ph1, tid_t);
    }
    if (stamp && (g.dbg & 256)) {   // phase breakdown: wait for the C stores, then {prologue, loop, epilogue} cycles of wave 0
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 0) {
        unsigned long long* o = stamp + 2 * ((size_t)gridDim.x * gridDim.y) + 3 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = ph0 - st_c0; o[1] = ph1 - ph0; o[2] = __builtin_amdgcn_s_memtime() - ph1;
      }
    }
    if (stamp && tid == 0) {
      stamp[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = __builtin_amdgcn_s_memtime() - st_c0;
      stamp[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = ((__builtin_amdgcn_s_memrealtime() - st_r0) << 8) | (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf);
    }
    if (PERSIST) __syncthreads();   // every wave is done with the LDS buffers before the next tile's first DMA lands in them
  }
}

#ifdef SIGP_DEBUG_TOOLS
}  // namespace sigp
#include "debug/syrk_tile_experiments.hpp"   // rejected tile shapes, libsigp_debug.so only
namespace sigp {
#endif


// Panel solve by row strips (panel_mode 1).  The top Wp x Wp block of a panel is already factored and
//   Mt = [ inv(L_00)                                   ]      (block row j: -inv(L_jj) L_jk for k < j, inv(L_jj) at k = j)
//        [ -inv(L_11) L_10   inv(L_11)                 ]
//        [ ...                                         ]
// is given.  Block forward substitution of one 128-row strip below the top block,
//   X_j = (A_j - sum_{k<j} X_k L_jk^T) inv(L_jj)^T  =  [X_0 .. X_{j-1} | A_j] [Mt_j0 .. Mt_jj]^T ,
// is then ONE tile product per column block j with K = 128 (j+1), whose A operand is the strip itself as it stands
// (columns < j already solved, column j still the input).  One workgroup walks j = 0 .. Wp-1 over its strip, so the
// strip is read and written once from HBM and every flop of the panel's lower rows runs in the LDS-DMA MFMA loop.
template <typename T>
struct StripArgsT {
  T* M; long ld; long sM;          // matrix, row stride, member stride
  const T* Mt; long ldm; long sMt; // Mt workspace [Wp*128][ldm] per member
  int rb0;                         // first row block of the strips (blockIdx.x = strip)
  int J;                           // first column block of the panel
  int Wp;                          // panel width in column blocks
};

// TRI_ (default for fp64): the triangular last K block of every column product skips its zero tile-slices (syrk128_tile's TRI form).
template <typename T, bool TRI_ = (sizeof(T) == 8)>
__global__ __launch_bounds__(256, 2) void panel_strip_kernel(StripArgsT<T> g) {
  constexpr int KTe = Num<T>::KT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = (T*)smem_raw;
  T* Bs = As + 2 * SY_T * KTe;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const long bz = blockIdx.y;
  unsigned long long ph0 = 0, ph1 = 0;
  T* strip = g.M + bz * g.sM + (long)(g.rb0 + blockIdx.x) * SY_T * g.ld + (long)g.J * SY_T;
  const T* Mt = g.Mt + bz * g.sMt;
  for (int j = 0; j < g.Wp; ++j) {
    constexpr bool TRI = TRI_ && sizeof(T) == 8;   // (the engine strip-solves fp64 panels only; the fp32 instantiation keeps the plain tile)
    T* Cq = strip + (long)(wm * 64) * g.ld + (long)j * SY_T + wn * (TRI ? 16 : 64);
    const T* Mj = Mt + (long)j * SY_T * g.ldm;
    int tid_t = threadIdx.x;
    asm volatile("" : "+v"(tid_t));             // opaque per column: the tile's lane-dependent addresses are formed inside the loop (hoisted, they spill)
    if (!TRI || wn == 0) syrk128_tile<T, true, 0, TRI, 0>(strip, g.ld, Mj, g.ldm, Cq, g.ld, SY_T * (j + 1), 0, As, Bs, false, ph0, ph1, tid_t);
    else syrk128_tile<T, true, 0, TRI, 1>(strip, g.ld, Mj, g.ldm, Cq, g.ld, SY_T * (j + 1), 0, As, Bs, false, ph0, ph1, tid_t);   // (same barriers on both paths)
    // X_j is the A operand of the next column.  Producer and consumer are the same workgroup (one CU, one L1/L2
    // path), so workgroup scope is enough -- but the wait for this wave's stores has to be spelled out: a workgroup-
    // scope fence on gfx950 (not in threadgroup-split mode) does not include vmcnt(0), and whether the compiler's own
    // waitcnt pass puts one in front of the barrier depends on the surrounding control flow (it did for the plain tile
    // and did not behind the two column-wave instantiations: the next column's LDS-DMA then read rows of X_j that had not
    // landed -- wrong factors in large lockstep groups only).  Then the barrier: everyone's stores are out and everyone's
    // fragment reads of this tile are done before the next tile's DMA refills the LDS buffers.  (An agent-scope fence
    // here writes back / invalidates the XCD's L2 on gfx950 and cost 10 % of the whole fit.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

}  // namespace sigp
