// Diagonal-block kernel of the blocked Cholesky (north/June1st.py:265 np.linalg.cholesky -> dpotrf):
// factor one 128x128 SPD block in LDS, write L11 back, and form inv(L11) in the workspace so that the panel solve
// L21 = A21 inv(L11)^T  is a plain MFMA GEMM.
//
// One 512-thread workgroup: the step is a latency chain (128 dependent pivots), not throughput work, so the kernel is built
// around keeping everything else off that chain.  The block lives in LDS as 36 lower 16x16 tiles (pitch 18: conflict-free
// 8-byte MFMA fragment reads) and is processed in 16-column steps ("slots"); in every slot the eight waves have fixed roles:
//   all waves first bring the NEXT block column up to date (rank-16 update, one tile per wave, one barrier); then
//   pivot waves (wave 0, and wave 1 / 2 while there are more than three row tiles below): the 16 pivots of that column with the
//       rows in registers -- lanes 0..15 hold the diagonal tile, lanes 16..63 three tiles of the rows below, so the row solve
//       x L11^T = a is the same instruction stream as the pivot loop and costs nothing extra.  Pivots are LDL^T-style
//       (reciprocal of the pivot + unscaled columns; the 16 inverse square roots are taken once, lane-parallel, after the loop):
//       the dependent chain per pivot is v_rcp_f64 + 2 Newton steps instead of v_rsq_f64 + 2 Goldschmidt steps;
//   MFMA waves (the others): the rank-16 update of the remaining trailing tiles (v_mfma_f64_16x16x4_f64), then row s of the
//       INVERSE by bordering:  X(s,j) = -X(s,s) sum_{k=j}^{s-1} L(s,k) X(k,j)  -- the rows above are complete, the sums need
//       no data of this slot, and only the last product waits (LDS flag) for
//   wave 7: writes row s of L back to global memory, inverts the 16x16 diagonal tile (forward substitution, one lane per
//       column), raises the flag, and streams the finished row s-1 of the inverse to the workspace.
// The inverse is therefore complete one short tail (last tile inverse + one product) after the last pivot, instead of a
// separate phase of 3 levels x 4 barriers; results that overwrite tiles other waves still read are held in registers over the
// slot's closing barrier and written before the next slot's opening barrier.
// Register budget: the kernel must stay at <= 128 VGPRs.  Its 8 waves then take 2 x 128 of a SIMD's 512 registers and fit
// beside ONE resident wave of the trailing-update kernel (256 VGPRs); a 220-VGPR build has to wait for BOTH update workgroups
// of a CU to finish: 1.06 ms instead of 0.23 ms per launch in lockstep batches.  Check `.vgpr_count` (make resources).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"
#include "lane_ops.hpp"

namespace sigp {

constexpr int DB = 128;   // diagonal block size
constexpr int BP = 18;    // pitch (elements) inside a 16x16 LDS tile: conflict-free 8-byte MFMA fragment reads
constexpr int BSZ = 16 * BP;
constexpr int DIAG_XT = 1;   // wave 0's copy of the factored diagonal tile (moved into the block one barrier later)
// LDS: the 36 lower tiles + that copy + the reciprocal diagonal + 8 flag words: 86 KB in fp64, so the kernel
// shares a CU with one resident trailing-update workgroup (64 KB) instead of waiting for a whole CU to drain.
template <typename T> constexpr int diag_lds_bytes() { return ((36 + DIAG_XT) * BSZ + DB) * (int)sizeof(T) + 64; }
constexpr int DIAG_LDS_BYTES = diag_lds_bytes<double>();
constexpr int DIAG_THREADS = 512;

#ifdef SIGP_DEBUG_TOOLS
// tools/diag_phases.py: wall-clock stamps (s_memrealtime, 10 ns ticks) of the kernel's phases, written by wave 0 when bit 64 of `flags` is set
__device__ unsigned long long g_diag_stamp[64];
#define DIAG_STAMP(k) do { if ((flags & 64) && threadIdx.x == 0) { g_diag_stamp[(k)] = __builtin_amdgcn_s_memrealtime(); if ((k) == 0 || (k) == 44) g_diag_stamp[48 + ((k) != 0)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define DIAG_STAMP(k) do { } while (0)
#endif

__device__ inline double rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }
__device__ inline float rsq_seed(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double rcp_seed(double x) { return __builtin_amdgcn_rcp(x); }
__device__ inline float rcp_seed(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline int dblk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BSZ; }   // bj <= bi

// A: the 128x128 block inside the big matrix (row stride lda); Linv: [128][128] row-major workspace whose
// strictly-upper part is zero (zeroed once at allocation, never written here);
// info: device word, first failing 1-based global pivot index (0 = none yet); pivot_base: global index of row 0.
// blockIdx.x = batch member: A += b*strideA, Linv += b*strideL, info += b.
template <typename T>
__device__ __forceinline__ void potrf_diag_body(T* __restrict__ A, long lda, T* __restrict__ Linv, int* __restrict__ info, int pivot_base,
                                                int flags, char* smem_raw) {
  const int skip = flags & DBG_MASK;   // timing ablations of tools/diag_bench.py (debug library only): 1 no pivot loop, 2 no MFMA-wave
                                       // work, 4 no wave-7 work, 8 no pivot-wave update, 16 no global loads/stores of the tiles
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v2_t v2_t;
  T* S = (T*)smem_raw;               // 36 tiles of [16][18]
  T* XT = S + 36 * BSZ;              // wave 0: the factored diagonal tile until commit_diag
  T* dinv = XT + DIAG_XT * BSZ;      // [128] reciprocals of the diagonal of L
  volatile int* flag = (volatile int*)(dinv + DB);   // flag[b] = 1: tile (b, b) holds its inverse

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DIAG_STAMP(0);
  int lane = tid & 63;
  int lr = lane & 15, lq = lane >> 4;
  // latency chain on the critical path of every panel, co-resident with MFMA-saturating update waves: ask the instruction
  // arbiter for a high wave priority, the pivot loop above the kernel's own MFMA waves (bit 32 of `flags` disables it: A/B timing)
  if (!(flags & 32)) {
    if (wave < 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
  }

  // ---- load the lower triangle of the block ----
  // All sixteen 16-byte loads of a thread are issued before the first one is waited for (a load and its LDS store inside one predicated
  // iteration were sixteen dependent global round trips: 7.5 of the kernel's 37 us, tools/diag_phases.py).  Lanes right of the diagonal
  // load nothing: their addresses are clamped onto the diagonal pair and the values dropped.
  {
    constexpr int NIT = DB * (DB / 2) / DIAG_THREADS;
    v2_t ldv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * DIAG_THREADS;
      const int row = idx >> 6, cp = min((idx & 63) * 2, row & ~1);
      ldv[it] = (skip & 16) ? v2_t{(T)0, (T)0} : *(const v2_t*)(A + (long)row * lda + cp);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * DIAG_THREADS;
      const int row = idx >> 6, cp = (idx & 63) * 2;
      if (cp <= row && !(skip & 16)) *(v2_t*)(S + dblk(row >> 4, cp >> 4) + (row & 15) * BP + (cp & 15)) = ldv[it];
    }
  }
  if (tid < 8) flag[tid] = 0;
  __syncthreads();
  DIAG_STAMP(1);

  // one tile of a rank-16 step:  Cd = Cs - A_ B_^T  (two tiles interleaved: one's MFMA chain hides behind the other's)
  auto upd2 = [&](const T* Cs0, T* Cd0, const T* A0, const T* B0, bool two, const T* Cs1, T* Cd1, const T* A1, const T* B1) {
    acc_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc0[r] = Cs0[N_::drow(lq, r) * BP + lr];
      acc1[r] = Cs1[N_::drow(lq, r) * BP + lr];
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
    for (int r = 0; r < 4; ++r) Cd0[N_::drow(lq, r) * BP + lr] = acc0[r];
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Cd1[N_::drow(lq, r) * BP + lr] = acc1[r];
    }
  };

  // ---- pivot wave: block column jb of the rows this wave holds (lanes 0..15 the diagonal tile, from `dsrc`; lanes 16q..16q+15
  // row tile jb + 3w + q).  The row tiles below are written back in place, the diagonal
  // tile is left in wave 0's copy and moved into the block before the next slot's opening barrier (commit_diag).
  auto pivot_column = [&](int jb, int w) {
    T r[16];
    const int bt = jb + 3 * w + lq;
    const bool below = lq > 0 && bt <= 7;
    T* tile = S + dblk(below ? bt : jb, jb);        // lanes without a row tile walk the diagonal tile too (results unused)
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = tile[lr * BP + c];
    if (jb == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); DIAG_STAMP(50); }
    // Branch-free: the 16 pivots are ONE basic block, so the scheduler can run the tail of pivot j's column updates under the
    // reciprocal latency of pivot j+1.
    int bad = 0;          // first non-positive / NaN pivot of this tile (1-based), uniform
    T myd = (T)1;         // lane j (diagonal rows) keeps d_j
    if (!(skip & 1))
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const T dj = readlane_t(r[j], j);
      // a non-positive / NaN pivot is recorded, NOT replaced: its reciprocal poisons the rest of this member's factor (inf / NaN), which nobody
      // reads once `info` is set -- the compare-and-select that kept the factor finite sat on the dependent chain of every pivot
      const bool neg = !(dj > (T)0);
      bad = (neg && bad == 0) ? j + 1 : bad;
      T x = rcp_seed(dj);
      T e = fma(-dj, x, (T)1);
      x = fma(x, e, x);
      e = fma(-dj, x, (T)1);
      x = fma(x, e, x);                // 1 / d_j
      const T t = r[j] * x;            // u_ij / d_j
      // the pivot column's entries four at a time into four scalar register pairs, then their four FMAs: with one pair reused for every entry each
      // v_readlane pair / wait state / FMA triple was serialised on it (s_nop after every pair: ~20 cycles per entry, 355 per pivot -- tools/diag_phases.py)
#pragma unroll
      for (int c0 = j + 1; c0 < 16; c0 += 4) {
        T u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = (c0 + q < 16) ? readlane_t(r[j], c0 + q) : (T)0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c0 + q < 16) r[c0 + q] = fma(-t, u[q], r[c0 + q]);
      }
    }
    // lane j of the diagonal rows: d_j = its own r[j], untouched since pivot j-1 updated it (picked out once here: six instructions per pivot inside the loop)
#pragma unroll
    for (int j = 0; j < 16; ++j) myd = (lr == j) ? r[j] : myd;
    if (jb == 1) DIAG_STAMP(51);
    // 1 / sqrt(d) for the 16 pivots at once (lane j: d_j): v_rsq + two Goldschmidt steps
    const T y0 = rsq_seed(myd);
    T g = myd * y0, hh = (T)0.5 * y0;
    T e = fma(-hh, g, (T)0.5);
    g = fma(g, e, g); hh = fma(hh, e, hh);
    e = fma(-hh, g, (T)0.5);
    g = fma(g, e, g); hh = fma(hh, e, hh);
    const T e2 = fma(-g, g, myd);
    const T sq = fma(e2, hh, g);       // sqrt(d)
    const T rs = hh + hh;              // 1 / sqrt(d)
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {     // (four scalar register pairs in flight, as in the pivot loop)
      T sc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[q] = readlane_t(rs, j0 + q);
#pragma unroll
      for (int q = 0; q < 4; ++q) r[j0 + q] *= sc[q];
    }
    if (jb == 1) DIAG_STAMP(52);
    if (below) {
#pragma unroll
      for (int c = 0; c < 16; ++c) tile[lr * BP + c] = r[c];
    } else if (w == 0 && lq == 0) {    // the factored diagonal tile stays in wave 0's copy until commit_diag
      dinv[jb * 16 + lr] = rs;         // (first read by the tile inverse of the next slot)
      // the whole row as it stands, then the diagonal entry over it: nothing reads the tile right of its diagonal before wave 7 overwrites it with
      // the inverse (store_tile skips it, the tile inverse reads below the diagonal), and sixteen select pairs per lane were 0.3 us per column
#pragma unroll
      for (int c = 0; c < 16; ++c) XT[lr * BP + c] = r[c];
      XT[lr * BP + lr] = sq;
    }
    // LAPACK info = index of the first failing pivot
    if (w == 0 && bad != 0 && lane == 0 && *info == 0) *info = pivot_base + jb * 16 + bad;
    if (jb == 1) DIAG_STAMP(53);
  };
  auto commit_diag = [&](int jb) {     // wave 0: its copy of the factored diagonal tile -> the block
    if (wave == 0) {
      T* Sjj = S + dblk(jb, jb);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = (lane >> 3) + 8 * p, cp = (lane & 7) * 2;
        *(v2_t*)(Sjj + row * BP + cp) = *(const v2_t*)(XT + row * BP + cp);
      }
    }
  };
  // one 16x16 tile LDS -> global, lower part only when `diag` (rows of 16 elements = 8 pairs; 64 lanes x 2 passes)
  auto store_tile = [&](const T* tile, T* G, long ldg, bool diag) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (lane >> 3) + 8 * p, cp = (lane & 7) * 2;
      if (!diag || cp <= row) *(v2_t*)(G + (long)row * ldg + cp) = *(const v2_t*)(tile + row * BP + cp);
    }
  };

  if (wave < 3) pivot_column(0, wave);
  DIAG_STAMP(2);
  __syncthreads();
  DIAG_STAMP(3);

  acc_t pend;               // an MFMA wave's finished inverse tile X(s, pend_j)^T, written before the next opening barrier
  int pend_j = -1;
  for (int s = 0; s < 8; ++s) {
    // keep the lane-index arithmetic of the three roles inside the loop: hoisted out of it (per-element LDS offsets of every
    // branch kept live over the whole kernel) it costs 60 VGPRs
    asm volatile("" : "+v"(lane));
    lr = lane & 15; lq = lane >> 4;
    // ---- writes held back over the closing barrier: the diagonal tile of column s, row s-1 of the inverse ----
    commit_diag(s);
    if (pend_j >= 0) {
      T* D = S + dblk(s - 1, pend_j);
#pragma unroll
      for (int q = 0; q < 4; ++q) D[lr * BP + N_::drow(lq, q)] = pend[q];
      pend_j = -1;
    }
    __syncthreads();
    DIAG_STAMP(4 + 5 * s);
    const int jn = s + 1;                               // the block column the pivot waves work on in this slot
    // rank-16 update (column s) of block column jn, one tile per wave (the pivot loop needs all of them: with every SIMD's matrix
    // pipe on it the step is one 4-MFMA chain instead of up to four on the pivot wave's own)
    if (s < 7 && wave <= 7 - jn && !(skip & 8)) {
      T* Cn = S + dblk(jn + wave, jn);
      upd2(Cn, Cn, S + dblk(jn + wave, s), S + dblk(jn, s), false, Cn, Cn, S + dblk(jn + wave, s), S + dblk(jn, s));
    }
    DIAG_STAMP(5 + 5 * s);
    if (s < 7) __syncthreads();
    DIAG_STAMP(6 + 5 * s);
    const bool pivot_wave = s < 7 && (wave == 0 || (wave == 1 && s <= 2));
    if (pivot_wave) {
      const int w = wave;
      pivot_column(jn, w);
      DIAG_STAMP(7 + 5 * s);
    } else if (wave == 7) {
      // row s of L -> global; inverse of the diagonal tile in place; row s-1 of the inverse -> workspace
      if (!(skip & 16)) store_tile(S + dblk(s, s), A + (long)(s * 16) * lda + s * 16, lda, true);
      if (lane < 16 && !(skip & 4)) {
        const int c = lr;
        T* Sbb = S + dblk(s, s);
        T x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = (i == c) ? (T)1 : (T)0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          x[i] *= dinv[s * 16 + i];
#pragma unroll
          for (int p = i + 1; p < 16; ++p) x[p] = fma(-Sbb[p * BP + i], x[i], x[p]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) Sbb[i * BP + c] = x[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) flag[s] = 1;
      if (!(skip & 16)) {
      for (int k = 0; k < s; ++k) store_tile(S + dblk(s, k), A + (long)(s * 16) * lda + k * 16, lda, false);
      if (s > 0)
        for (int k = 0; k < s; ++k) store_tile(S + dblk(s - 1, k), Linv + (long)((s - 1) * 16) * DB + k * 16, DB, k == s - 1);
      if (s == 7) store_tile(S + dblk(7, 7), Linv + (long)(7 * 16) * DB + 7 * 16, DB, true);
      }
    } else if (!(skip & 2)) {
      // MFMA waves: index m among the nm of this slot
      const int nm = (s == 7) ? 7 : (s >= 3 ? 6 : 5);
      const int m = (s == 7) ? wave : (s >= 3 ? wave - 1 : wave - 2);
      // rank-16 update (column s) of the trailing tiles (i, k), s+2 <= k <= i <= 7, two per iteration
      const int nb = 6 - s;
      const int nt = nb > 0 ? nb * (nb + 1) / 2 : 0;
      auto tile_of = [&](int t, int& ti, int& tj) { ti = 0; int rem = t; while (rem > ti) { rem -= ti + 1; ++ti; } tj = rem; };
      for (int t = (m + nm - (s % nm)) % nm; t < nt; t += 2 * nm) {
        const bool two = t + nm < nt;
        int i0, k0, i1, k1;
        tile_of(t, i0, k0);
        tile_of(two ? t + nm : t, i1, k1);
        i0 += s + 2; k0 += s + 2; i1 += s + 2; k1 += s + 2;
        T* C0 = S + dblk(i0, k0);
        T* C1 = S + dblk(i1, k1);
        upd2(C0, C0, S + dblk(i0, s), S + dblk(k0, s), two, C1, C1, S + dblk(i1, s), S + dblk(k1, s));
      }
      // row s of the inverse: tile j = m (the longest sums go to the first waves)
      if (m < s) {
        const int j = m;
        acc_t q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (T)0;
        for (int k = j; k < s; ++k) {
          const T* A_ = S + dblk(s, k);
          const T* B_ = S + dblk(k, j);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) q = N_::mfma(A_[lr * BP + kk * 4 + lq], B_[(kk * 4 + lq) * BP + lr], q);
        }
        for (int spin = 0; flag[s] == 0 && spin < (1 << 20); ++spin) __builtin_amdgcn_s_sleep(1);   // (wave 7 raises it unconditionally)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // X(s,j)^T = -Q^T X(s,s)^T: the accumulator layout of Q is the A-operand layout of Q^T (k = the accumulator's row)
        const T* Xss = S + dblk(s, s);
        acc_t res;
#pragma unroll
        for (int i = 0; i < 4; ++i) res[i] = (T)0;
#pragma unroll
        for (int i = 0; i < 4; ++i) res = N_::mfma(q[i], -Xss[lr * BP + N_::drow(lq, i)], res);
        if (s < 7) {
          pend = res; pend_j = j;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) Linv[(long)(7 * 16 + lr) * DB + j * 16 + N_::drow(lq, i)] = res[i];
        }
      }
    }
    if (s < 7) __syncthreads();
    DIAG_STAMP(8 + 5 * s);
  }
  DIAG_STAMP(44);
}

template <typename T>
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(T* __restrict__ A, long lda, T* __restrict__ Linv,
                                                                  int* __restrict__ info, int pivot_base, int flags, long strideA,
                                                                  long strideL) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  potrf_diag_body<T>(A + blockIdx.x * strideA, lda, Linv + blockIdx.x * strideL, info + blockIdx.x, pivot_base, flags, smem_raw);
}

// The diagonal block of column c+1 AND, beside it, the rank-128 update of the panel's columns c+2.. with column c: one launch.
// On the panel stream the chain is  diagonal block -> column solve -> update of the next column -> diagonal block ...; the update
// of the panel's OTHER columns is most of a panel's flops but nothing on the chain needs it before the column solve that follows
// this launch.  As a launch of its own on the same stream it sits in the chain (30 us per column); on a second stream every
// hand-off is an inter-queue barrier packet (7-13 us each, measured).  As extra workgroups of the diagonal-block launch it costs
// nothing: workgroups 0..nb-1 factor (one per lockstep member), every other workgroup is two 256-thread 64x64-tile engines of
// the generic update (same arithmetic, same k order: bit-identical to separate launches) running in lockstep on equal K.
static_assert(2 * gemm_lds_bytes<double, 64, 64, false>() <= DIAG_LDS_BYTES, "fp64: the update engines live in the diagonal block's LDS allocation");
// With the fused chain link (chain_link.hpp) in front of it, the launch has two more jobs: the block row c+1 of column c, L[c+1, c],
// was left in a scratch block by the link (the link's other workgroups were still reading the unsolved rows) -- `Balt` != nullptr:
// the riding tiles of block column c+1 (bj < 2 in 64-tile units) read their B operand from it, and nb more workgroups copy it to
// its place in the matrix (`copy_dst`, row stride lda; nothing in this launch reads it there).
template <typename T>
__global__ __launch_bounds__(DIAG_THREADS, 4) void diag_update_kernel(T* __restrict__ A, long lda, T* __restrict__ Linv, int* __restrict__ info,
                                                                   int pivot_base, int flags, long strideA, long strideL, int nb,
                                                                   GemmArgsT<T> g, int ntile, int wgs, const T* __restrict__ Balt, long sBalt,
                                                                   T* __restrict__ copy_dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if ((int)blockIdx.x < nb) {
    potrf_diag_body<T>(A + blockIdx.x * strideA, lda, Linv + blockIdx.x * strideL, info + blockIdx.x, pivot_base, flags, smem_raw);
    return;
  }
  const int u = (int)blockIdx.x - nb;
  if (u >= nb * wgs) {                 // copy workgroups: scratch block of member z -> block (c+1, c) of its matrix
    typedef typename Num<T>::v16_t v16_t;
    constexpr int CPR = DB * (int)sizeof(T) / 16;          // 16-byte chunks per row
    const int z = u - nb * wgs;
    const T* src = Balt + (long)z * sBalt;
    T* dst = copy_dst + (long)z * strideA;
    for (int i = (int)threadIdx.x; i < DB * CPR; i += DIAG_THREADS) {
      const int row = i / CPR, ch = i % CPR;
      *(v16_t*)(dst + (long)row * lda + ch * (16 / (int)sizeof(T))) = *(const v16_t*)(src + (long)row * DB + ch * (16 / (int)sizeof(T)));
    }
    return;
  }
  const int bz = u / wgs, w = u - bz * wgs;
  const int e = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  int t = 2 * w + e;
  const bool own = t < ntile;
  if (!own) t = ntile - 1;          // odd tile count: the last workgroup's second engine shadows the first (no store)
  int bi = 0, bj = 0;
  gemm_tile_coords(g, t, bi, bj);
  GemmArgsT<T> gl = g;
  if (Balt != nullptr && bj < 2) {  // block column c+1: its rows of column c are still in the link's scratch block
    gl.B = Balt; gl.ldb = DB; gl.sB = sBalt;
  }
  gemm_tile_body<T, 64, 64, 2, 2, GEMM_SUB, false, 2>(gl, bi, bj, (long)bz, (int)threadIdx.x & 255, smem_raw + e * gemm_lds_bytes<T, 64, 64, false>(), own);
}

}  // namespace sigp
