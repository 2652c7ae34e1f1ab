// Diagonal-block kernel of the blocked Cholesky (north/June1st.py:265 np.linalg.cholesky -> dpotrf):
// factor one 128x128 SPD block in LDS, write L11 back, and form inv(L11) in the workspace so that the panel solve
// L21 = A21 inv(L11)^T  is a plain MFMA GEMM.
//
// One 512-thread workgroup: the step is a latency chain (128 dependent pivots), not throughput work, so the kernel is built
// around keeping everything else off that chain.  The block lives in LDS as 36 lower 16x16 tiles (pitch 18: conflict-free
// 8-byte MFMA fragment reads) and is processed in 16-column steps ("slots"); in every slot the eight waves have fixed roles, dealt by the
// SIMD each wave sits on (diag_logical_wave: on gfx950 a wave issuing fp64 MFMAs starves the other wave of its SIMD -- the fp64 vector ALU
// and the fp64 matrix pipe are the same 16 lanes, a v_mfma_f64_16x16x4_f64 holds them for 64 cycles):
//   all waves first bring the NEXT block column up to date with the column just factored (one tile per wave, one barrier); then
//   pivot wave A (and B on another SIMD while more than three row tiles remain): the 16 pivots of that column with the rows in registers --
//       every 16-lane row of the wave holds a replica of the diagonal tile and one tile of the rows below, so the rank-1 step of a pivot is
//       DPP row broadcasts (v_fmac_f64_dpp row_newbcast: one instruction per entry, nothing through scalar registers) and the row solve
//       x L11^T = a of a tile is the same instruction stream as the pivots.  Wave A's first row carries the IDENTITY as its "tile below":
//       its row solve is the inverse of the diagonal tile, for free.  Pivots are LDL^T-style (reciprocal of the pivot + unscaled columns;
//       the 16 inverse square roots are taken once, lane-parallel, after the loop); the loop is ONE hand-scheduled instruction stream in
//       which the independent entries of pivot j fill the latency shadows of pivot j + 1's reciprocal chain (PivotStream);
//   MFMA waves (on the SIMDs without a pivot wave): six chains of dependent MFMAs per slot -- the tiles of block column s + 2 catch up with the
//       columns factored so far (LEFT-looking: the work is spread over the slots instead of front-loaded), the sums of row s of the INVERSE
//       by bordering, X(s,j) = -X(s,s) sum_{k=j}^{s-1} L(s,k) X(k,j), and one term of the LAST inverse row accumulated ahead -- dealt by length
//       so that the two waves of a SIMD share its pipe evenly;
//   the store wave (beside pivot wave A: no MFMAs, few instructions): finished tiles of L and of the inverse -> global memory.
// Results that overwrite tiles other waves still read are held in registers / scratch tiles over the slot's closing barrier and written
// before the next slot's opening barrier.
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
constexpr int DIAG_XT = 5;   // scratch tiles beside the block: XL[2], XI, XB, IDT (potrf_diag_body)
// LDS: the 36 lower tiles + 5 scratch tiles + the SIMD table: 92 KB in fp64, so the kernel still shares a CU with one resident
// trailing-update workgroup (64 KB of the CU's 160) instead of waiting for a whole CU to drain.
template <typename T> constexpr int diag_lds_bytes() { return (36 + DIAG_XT) * BSZ * (int)sizeof(T) + 64; }
constexpr int DIAG_LDS_BYTES = diag_lds_bytes<double>();
constexpr int DIAG_THREADS = 512;

#ifdef SIGP_DEBUG_TOOLS
// tools/diag_phases.py: wall-clock stamps (s_memrealtime, 10 ns ticks) of the kernel's phases, written by wave 0 when bit 64 of `flags` is set
__device__ unsigned long long g_diag_stamp[64];
__device__ unsigned long long g_diag_stamp_role[3 * 64];     // the same clock from the other roles: [0] logical wave 7, [1] logical wave 2, [2] logical wave 1; 8 stamps per slot
#define DIAG_STAMP(k) do { if ((flags & 64) && threadIdx.x == 0) { g_diag_stamp[(k)] = __builtin_amdgcn_s_memrealtime(); if ((k) == 0 || (k) == 44) g_diag_stamp[48 + ((k) != 0)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define DIAG_STAMP_ROLE(k) do { if ((flags & 64) && lane == 0 && (wave == 7 || wave == 2 || wave == 1)) g_diag_stamp_role[(wave == 7 ? 0 : wave == 2 ? 64 : 128) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DIAG_STAMP(k) do { } while (0)
#define DIAG_STAMP_ROLE(k) do { } while (0)
#endif

__device__ inline double rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }
__device__ inline float rsq_seed(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double rcp_seed(double x) { return __builtin_amdgcn_rcp(x); }
__device__ inline float rcp_seed(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline int dblk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BSZ; }   // bj <= bi

// The 16 pivots of one tile column as ONE hand-scheduled instruction stream (lane_ops.hpp's ordered statements; every DPP control is an
// immediate, so the whole loop is unrolled at compile time).  d = the diagonal tile's rows (one replica per 16-lane row), r = the rows of
// the tile below; LDL^T-style: reciprocal of the pivot + unscaled columns.  The dependent chain of pivot J + 1 is
//   d[J+1] += bcast(d[J]) ntd_J  ->  d_{J+1} = bcast_{J+1}(d[J+1])  ->  v_rcp  ->  two Newton steps (4 dependent fmas)  ->  ntd_{J+1} = -d[J+1] / d_{J+1}
// -- eight dependent fp64 operations of ~16 cycles each on a wave that issues in order -- while pivot J still owes 2 (15 - J) - 1
// independent entries (the other columns of d and r): PIVOT_GAP[k] of them are placed in front of chain operation k, the rest behind the
// chain.  A non-positive / NaN pivot is not replaced: its reciprocal poisons the rest of this member's factor (inf / NaN), which nobody
// reads once `info` is set; the caller finds the first one from the diagonal afterwards.
constexpr int PIVOT_CHAIN = 8;
constexpr int PIVOT_GAP[PIVOT_CHAIN] = {2, 3, 5, 3, 3, 3, 3, 0};     // entries in front of: bcast, rcp, err, upd, err, upd, ntd, nt
// item at position p of the stream "entries of pivot J beside the chain of pivot J + 1": >= 0 entry number (2 (c - J - 1) + (0: d, 1: r)),
// -1 - k chain operation k, -100 end.  J = 14: pivot 15 has no entries, so its chain is not needed.
constexpr int pivot_sched_item(int J, int p) {
  const int n = 2 * (15 - J);
  const bool chain = J < 14;
  int f = 0, pos = 0;
  if (n > 0) { if (p == pos) return 0; f = 1; ++pos; }
  if (chain)
    for (int k = 0; k < PIVOT_CHAIN; ++k) {
      for (int q = 0; q < PIVOT_GAP[k] && f < n; ++q) { if (p == pos) return f; ++f; ++pos; }
      if (p == pos) return -1 - k;
      ++pos;
    }
  while (f < n) { if (p == pos) return f; ++f; ++pos; }
  return -100;
}
template <typename T> struct PivotRegs { T dj, x, e, ntd[2], nt[2]; };     // chain temporaries; ntd / nt alternate with the pivot's parity
template <typename T, int K, int J> __device__ __forceinline__ void pivot_chain_op(T (&d)[16], T (&r)[16], PivotRegs<T>& q) {
  // chain of pivot J (its d[J] is final)
  if constexpr (K == 0) o_bcast16<J>(q.dj, d[J]);
  else if constexpr (K == 1) o_rcp(q.x, q.dj);
  else if constexpr (K == 2 || K == 4) o_nr_err(q.e, q.dj, q.x);
  else if constexpr (K == 3 || K == 5) o_nr_upd(q.x, q.e);
  else if constexpr (K == 6) o_mul_neg(q.ntd[J & 1], d[J], q.x);
  else o_mul_neg(q.nt[J & 1], r[J], q.x);
}
template <typename T, int J, int P> struct PivotStream {
  static __device__ __forceinline__ void run(T (&d)[16], T (&r)[16], PivotRegs<T>& q) {
    constexpr int it = pivot_sched_item(J, P);
    if constexpr (it != -100) {
      if constexpr (it >= 0) {
        constexpr int C = J + 1 + it / 2;
        if constexpr ((it & 1) == 0) o_fmac_bcast16<C>(d[C], d[J], q.ntd[J & 1]);
        else o_fmac_bcast16<C>(r[C], d[J], q.nt[J & 1]);
      } else {
        pivot_chain_op<T, -1 - it, J + 1>(d, r, q);
      }
      PivotStream<T, J, P + 1>::run(d, r, q);
    }
  }
};
template <typename T, int J> struct PivotSteps {
  static __device__ __forceinline__ void run(T (&d)[16], T (&r)[16], PivotRegs<T>& q) {
    PivotStream<T, J, 0>::run(d, r, q);
    PivotSteps<T, J + 1>::run(d, r, q);
  }
};
template <typename T> struct PivotSteps<T, 15> {
  static __device__ __forceinline__ void run(T (&)[16], T (&)[16], PivotRegs<T>&) {}
};
template <typename T, int K> struct PivotHead {            // the chain of pivot 0, nothing to hide behind
  static __device__ __forceinline__ void run(T (&d)[16], T (&r)[16], PivotRegs<T>& q) {
    pivot_chain_op<T, K, 0>(d, r, q);
    PivotHead<T, K + 1>::run(d, r, q);
  }
};
template <typename T> struct PivotHead<T, PIVOT_CHAIN> {
  static __device__ __forceinline__ void run(T (&)[16], T (&)[16], PivotRegs<T>&) {}
};
template <typename T, int C> struct ScaleCols {         // r[c] (and d[c]) *= rs of lane c (column c's 1 / sqrt(d_c))
  static __device__ __forceinline__ void run(T (&r)[16], T (&d)[16], T rs, bool both) {
    const T b = bcast16_ready<C>(rs);
    r[C] *= b;
    if (both) d[C] *= b;
    ScaleCols<T, C + 1>::run(r, d, rs, both);
  }
};
template <typename T> struct ScaleCols<T, 16> {
  static __device__ __forceinline__ void run(T (&)[16], T (&)[16], T, bool) {}
};

// Which role a wavefront plays.  A SIMD that is issuing MFMAs starves the other wavefront it hosts: beside a wave issuing back-to-back
// v_mfma_f64_16x16x4_f64 a dependent v_fma_f64 chain runs at 75 cycles per operation instead of 8, independent ones at 25 instead of 5, and
// s_setprio does not change it; MFMAs on the OTHER three SIMDs cost a VALU wave nothing (tools/probes/valu_issue_probe.hip).  The eight
// waves of this workgroup sit two per SIMD, so the roles are dealt by SIMD: the wave that shares the pivot wave's SIMD gets the one role
// without MFMAs (the store wave: logical wave 7), the wave beside the second pivot wave is the MFMA wave that idles while that one
// pivots (logical wave 6), and the MFMA waves 2..5 are numbered so that (2, 3) and (4, 5) are SIMD pairs like (1, 6).  The placement is read from HW_ID at run time (tab[w] = SIMD of physical wave w; observed: 0 2 1 3 0 2 1 3) and
// any placement is handled -- a role map is a permutation whatever the table says, only the speed depends on it.
__device__ inline int diag_logical_wave(const volatile int* tab, int pw) {
  int sd[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) sd[w] = __builtin_amdgcn_readfirstlane(tab[w]) & 3;
  int mA = 7;                                   // shares the SIMD of physical wave 0 (the pivot wave); nobody does: the last wave
#pragma unroll
  for (int w = 7; w >= 1; --w) mA = (sd[w] == sd[0]) ? w : mA;
  unsigned used = 1u | (1u << mA);
  int B = -1, Bf = -1;                          // second pivot wave: first free wave on another SIMD (else: first free wave)
#pragma unroll
  for (int w = 7; w >= 1; --w) {
    const bool fr = !((used >> w) & 1u);
    Bf = fr ? w : Bf;
    B = (fr && sd[w] != sd[0]) ? w : B;
  }
  B = B < 0 ? Bf : B;
  used |= 1u << B;
  int sdB = 0;
#pragma unroll
  for (int w = 1; w < 8; ++w) sdB = (w == B) ? sd[w] : sdB;
  int mB = -1, mBf = -1;                        // shares the second pivot wave's SIMD (else: the last free wave)
#pragma unroll
  for (int w = 7; w >= 1; --w) {
    const bool fr = !((used >> w) & 1u);
    mB = (fr && sd[w] == sdB) ? w : mB;
  }
#pragma unroll
  for (int w = 1; w < 8; ++w) mBf = !((used >> w) & 1u) ? w : mBf;
  mB = mB < 0 ? mBf : mB;
  used |= 1u << mB;
  // the other four: logical (2, 3) and (4, 5) are SIMD pairs where the placement allows (the MFMA work of a slot is dealt by pairs)
  int p2 = -1, p3 = -1, p4 = -1, p5 = -1;
#pragma unroll
  for (int w = 7; w >= 1; --w) p2 = !((used >> w) & 1u) ? w : p2;
  used |= 1u << p2;
  int sd2 = 0;
#pragma unroll
  for (int w = 1; w < 8; ++w) sd2 = (w == p2) ? sd[w] : sd2;
  int p3f = -1;
#pragma unroll
  for (int w = 7; w >= 1; --w) {
    const bool fr = !((used >> w) & 1u);
    p3f = fr ? w : p3f;
    p3 = (fr && sd[w] == sd2) ? w : p3;
  }
  p3 = p3 < 0 ? p3f : p3;
  used |= 1u << p3;
#pragma unroll
  for (int w = 7; w >= 1; --w) p4 = !((used >> w) & 1u) ? w : p4;
  used |= 1u << p4;
#pragma unroll
  for (int w = 7; w >= 1; --w) p5 = !((used >> w) & 1u) ? w : p5;
  if (pw == 0) return 0;
  if (pw == mA) return 7;
  if (pw == B) return 1;
  if (pw == mB) return 6;
  return pw == p2 ? 2 : pw == p3 ? 3 : pw == p4 ? 4 : 5;
}

// A: the 128x128 block inside the big matrix (row stride lda); Linv: [128][128] row-major workspace whose
// strictly-upper part is zero (zeroed once at allocation, never written here);
// info: device word, first failing 1-based global pivot index (0 = none yet); pivot_base: global index of row 0.
// blockIdx.x = batch member: A += b*strideA, Linv += b*strideL, info += b.
template <typename T>
__device__ __forceinline__ void potrf_diag_body(T* __restrict__ A, long lda, T* __restrict__ Linv, int* __restrict__ info, int pivot_base,
                                                int flags, char* smem_raw) {
  const int skip = flags & DBG_MASK;   // timing ablations of tools/diag_bench.py (debug library only): 1 no pivot loop, 2 no MFMA-wave
                                       // work, 4 no store-wave work, 8 no column update, 16 no global loads/stores of the tiles
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v2_t v2_t;
  T* S = (T*)smem_raw;               // 36 tiles of [16][18]
  T* XL = S + 36 * BSZ;              // [2] the factored diagonal tile of column jb (parity jb & 1) for the store wave; pivot wave A's scratch
  T* XI = XL + 2 * BSZ;              // the inverse of that tile until commit_diag moves it into the block
  T* XB = XI + BSZ;                  // pivot wave B's scratch
  T* IDT = XB + BSZ;                 // a 16 x 16 identity: the "tile below" whose row solve is the diagonal tile's inverse (see pivot_column)
  volatile int* simd_tab = (volatile int*)(IDT + BSZ);   // [8] SIMD of each physical wave

  const int tid = threadIdx.x, pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DIAG_STAMP(0);
  int lane = tid & 63;
  int lr = lane & 15, lq = lane >> 4;
  if (lane == 0) simd_tab[pwave] = (flags & 128) ? (pwave & 3) : (int)((__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) >> 4) & 3);   // HW_ID.SIMD_ID (debug bit 128: pretend w mod 4)

  // ---- load the lower triangle of the block ----
  // All sixteen 16-byte loads of a thread are issued before the first one is waited for (a load and its LDS store inside one predicated
  // iteration were sixteen dependent global round trips: 7.5 of the kernel's 37 us, tools/diag_phases.py).  Lanes right of the diagonal
  // load nothing: their addresses are clamped onto the diagonal pair and the values dropped.  (Fetching block column 0 first and the
  // other 28 tiles beside the first pivot column is not faster: with fewer waves requesting, the tiles arrive later than those pivots end
  // -- 5.1 - 6.2 us against 4.7 -- and carried in flight on every wave they cost the pivot waves their registers.)
  {
    constexpr int NIT = DB * (DB / 2) / DIAG_THREADS;
    v2_t ldv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * DIAG_THREADS;
      const int row = idx >> 6, cp = min((idx & 63) * 2, row & ~1);
      ldv[it] = (skip & 16) ? v2_t{(T)0, (T)0} : *(const v2_t*)(A + (long)row * lda + cp);
    }
    if (tid < 128) *(v2_t*)(IDT + (tid >> 3) * BP + (tid & 7) * 2) = v2_t{(tid >> 3) == (tid & 7) * 2 ? (T)1 : (T)0, (tid >> 3) == (tid & 7) * 2 + 1 ? (T)1 : (T)0};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * DIAG_THREADS;
      const int row = idx >> 6, cp = (idx & 63) * 2;
      if (cp <= row && !(skip & 16)) *(v2_t*)(S + dblk(row >> 4, cp >> 4) + (row & 15) * BP + (cp & 15)) = ldv[it];
    }
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(diag_logical_wave(simd_tab, pwave));     // the role (see diag_logical_wave)
  // latency chain on the critical path of every panel, co-resident with MFMA-saturating update waves: ask the instruction
  // arbiter for a high wave priority, the pivot loop above the kernel's own MFMA waves (bit 32 of `flags` disables it: A/B timing)
  if (!(flags & 32)) {
    if (wave < 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
  }
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

  // ---- pivot wave w (0: A, 1: B): block column jb.  Every 16-lane row g of the wave holds, lane = tile row, registers = the 16 columns:
  //   d[] a REPLICA of the diagonal tile (jb, jb) and r[] the tile below it this row owns -- so the whole rank-1 step of a pivot is DPP row
  //   broadcasts of the replica's pivot column (lane_ops.hpp): one fused multiply-add per entry and register array, nothing through scalar
  //   registers.  The replicas of the rows (and of the second pivot wave) run the same operations on the same data and stay bit-identical.
  //   Wave A: row 0 owns the IDENTITY, rows 1..3 the tiles jb+1 .. jb+3; wave B (while jb <= 3): tiles jb+4 .. jb+7.  The row solve
  //   x L^T = a of a tile is the same instruction stream as the pivots; for a = I it leaves L^-T: the inverse of the diagonal tile costs
  //   nothing (it used to be a 16-step substitution on a wave of its own, with the MFMA waves' last products waiting for its flag).
  //   Tiles below are written back in place; wave A leaves the factored diagonal tile in XL[jb & 1] (the store wave sends it to global memory
  //   in the next slot) and the inverse in XI (moved over the block's tile before the next slot's opening barrier: commit_diag).
  auto pivot_column = [&](int jb, int w) {
    T d[16], r[16];
    const bool ident = w == 0 && lq == 0;
    const int bt = w == 0 ? jb + lq : jb + 4 + lq;
    const bool below = !ident && bt <= 7;
    const T* dtile = S + dblk(jb, jb);
    T* tile = ident ? IDT : S + dblk(below ? bt : jb, jb);     // rows without a tile walk the diagonal tile (results unused)
    T* scr = w == 0 ? XL + (jb & 1) * BSZ : XB;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const v2_t a = *(const v2_t*)(dtile + lr * BP + c);
      const v2_t b = *(const v2_t*)(tile + lr * BP + c);
      d[c] = a[0]; d[c + 1] = a[1]; r[c] = b[0]; r[c + 1] = b[1];
    }
    if (jb == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); DIAG_STAMP(50); }
    if (!(skip & 1)) {
      PivotRegs<T> q;
      PivotHead<T, 0>::run(d, r, q);
      PivotSteps<T, 0>::run(d, r, q);
    }
    if (jb == 1) DIAG_STAMP(51);
    // lane lr's own diagonal entry d[lr] (a register index that differs per lane): through this wave's scratch tile -- the first row of
    // 16 lanes writes its rows, every lane reads its diagonal element back (LDS operations of one wave execute in order)
    if (lq == 0) {
#pragma unroll
      for (int c = 0; c < 16; c += 2) *(v2_t*)(scr + lr * BP + c) = v2_t{d[c], d[c + 1]};
    }
    const T myd = scr[lr * BP + lr];
    // first non-positive / NaN pivot of this tile (1-based; pivots before it are what they would be anyway): LAPACK's info
    const int bad = __builtin_ffsll((long long)(__builtin_amdgcn_ballot_w64(!(myd > (T)0)) & 0xffffull));
    // 1 / sqrt(d) for the 16 pivots at once (lane j: d_j): v_rsq + two Goldschmidt steps
    const T y0 = rsq_seed(myd);
    T g = myd * y0, hh = (T)0.5 * y0;
    T e = fma(-hh, g, (T)0.5);
    g = fma(g, e, g); hh = fma(hh, e, hh);
    e = fma(-hh, g, (T)0.5);
    g = fma(g, e, g); hh = fma(hh, e, hh);
    const T e2 = fma(-g, g, myd);
    const T sq = fma(e2, hh, g);       // sqrt(d)
    const T rs = dpp_ready(hh + hh);   // 1 / sqrt(d)
    ScaleCols<T, 0>::run(r, d, rs, w == 0);
    if (jb == 1) DIAG_STAMP(52);
    if (below) {
#pragma unroll
      for (int c = 0; c < 16; c += 2) *(v2_t*)(tile + lr * BP + c) = v2_t{r[c], r[c + 1]};
    }
    if (ident) {
      // r = L^-T (row lr = column lr of X = L^-1; exact zeros right of ... left of the diagonal): X(c, lr) -> XI [c][lr]
#pragma unroll
      for (int c = 0; c < 16; ++c) XI[c * BP + lr] = r[c];
      // the factored diagonal tile, the whole row as it stands with the diagonal entry over it (store_tile skips the part right of the diagonal)
#pragma unroll
      for (int c = 0; c < 16; c += 2) *(v2_t*)(scr + lr * BP + c) = v2_t{d[c], d[c + 1]};
      scr[lr * BP + lr] = sq;
    }
    // LAPACK info = index of the first failing pivot
    if (w == 0 && bad != 0 && lane == 0 && *info == 0) *info = pivot_base + jb * 16 + bad;
    if (jb == 1) DIAG_STAMP(53);
  };
  auto commit_diag = [&](int jb) {     // wave 0: the inverse of the diagonal tile (jb, jb) -> the block (nobody needs L(jb, jb) from there any more)
    if (wave == 0) {
      T* Sjj = S + dblk(jb, jb);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = (lane >> 3) + 8 * p, cp = (lane & 7) * 2;
        *(v2_t*)(Sjj + row * BP + cp) = *(const v2_t*)(XI + row * BP + cp);
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
  // block row `brow`, tiles k = 0 .. ntile-1 (all full tiles) -> global rows at G: every LDS read of the row is in flight before the first
  // store waits for one (a read-then-store loop was up to 26 dependent LDS round trips per slot on this wave)
  auto store_row = [&](int brow, int ntile, T* G, long ldg) {
    v2_t buf[14];
    const int row = lane >> 3, cp = (lane & 7) * 2;
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k < ntile) {
        const T* tile = S + dblk(brow, k);
        buf[2 * k] = *(const v2_t*)(tile + row * BP + cp);
        buf[2 * k + 1] = *(const v2_t*)(tile + (row + 8) * BP + cp);
      }
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k < ntile) {
        *(v2_t*)(G + (long)row * ldg + k * 16 + cp) = buf[2 * k];
        *(v2_t*)(G + (long)(row + 8) * ldg + k * 16 + cp) = buf[2 * k + 1];
      }
  };

  DIAG_STAMP_ROLE(6);
  if (wave < 2) pivot_column(0, wave);
  DIAG_STAMP(2);
  DIAG_STAMP_ROLE(5);
  __syncthreads();
  DIAG_STAMP(3);

  acc_t pend;               // an MFMA wave's finished inverse tile X(s, pend_j)^T, written before the next opening barrier
  int pend_j = -1;
  // The LAST row of the inverse is accumulated ahead: logical wave w = 2 .. 5, 1, 6 owns the sum Q(7, jh) = sum_k L(7, k) X(k, jh),
  // jh = 0 .. 3, 4, 5, and adds the term k as soon as row k of the inverse is in the block (slot k + 1) -- in the last slot, where nothing
  // hides it, a chain of 8 MFMAs is left per tile instead of up to 32.  Terms are added in ascending k whenever: the same sums.
  const int jh = (wave >= 2 && wave <= 5) ? wave - 2 : wave == 1 ? 4 : wave == 6 ? 5 : -1;
  int q7_next = jh;
  acc_t q7;
#pragma unroll
  for (int i = 0; i < 4; ++i) q7[i] = (T)0;
  auto q7_terms = [&](int kend) {          // add the terms k = q7_next .. kend-1
    for (int k = q7_next; k < kend; ++k) {
      const T* A_ = S + dblk(7, k);
      const T* B_ = S + dblk(k, jh);
      T ca[4], cb[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { ca[kk] = A_[lr * BP + kk * 4 + lq]; cb[kk] = B_[(kk * 4 + lq) * BP + lr]; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) q7 = N_::mfma(ca[kk], cb[kk], q7);
    }
    q7_next = q7_next > kend ? q7_next : kend;
  };
  for (int s = 0; s < 8; ++s) {
    // keep the lane-index arithmetic of the three roles inside the loop: hoisted out of it (per-element LDS offsets of every
    // branch kept live over the whole kernel) it costs 60 VGPRs
    asm volatile("" : "+v"(lane));
    lr = lane & 15; lq = lane >> 4;
    // ---- writes held back over the closing barrier: the inverse of diagonal tile s, row s-1 of the inverse ----
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
    DIAG_STAMP_ROLE(8 * s + 0);
    const bool pivot_wave = s < 7 && (wave == 0 || (wave == 1 && jn <= 3));     // wave A: the identity + three tiles below; wave B: four more while there are any
    if (pivot_wave) {
      const int w = wave;
      pivot_column(jn, w);
      DIAG_STAMP(7 + 5 * s);
    } else if (wave == 7) {
      // the store wave (on the pivot wave's SIMD: no MFMAs, few instructions): the factored diagonal tile of column s, the finished
      // tiles of block row s + 1 of L and block row s - 1 of the inverse -> global
      if (!(skip & 16) && !(skip & 4)) {
        store_tile(XL + (s & 1) * BSZ, A + (long)(s * 16) * lda + s * 16, lda, true);
        if (s < 7) store_row(s + 1, s + 1, A + (long)((s + 1) * 16) * lda, lda);      // (block row s + 1 of L left of its diagonal tile is final since the pivots of column s)
        if (s > 0) {
          store_row(s - 1, s - 1, Linv + (long)((s - 1) * 16) * DB, DB);
          store_tile(S + dblk(s - 1, s - 1), Linv + (long)((s - 1) * 16) * DB + (s - 1) * 16, DB, true);
        }
        if (s == 7) store_tile(S + dblk(7, 7), Linv + (long)(7 * 16) * DB + 7 * 16, DB, true);
      }
      DIAG_STAMP_ROLE(8 * s + 3);
    } else if (!(skip & 2)) {
      // MFMA waves.  A slot's work is always SIX chains of dependent MFMAs (64 cycles each on a SIMD's only fp64 pipe): s sums of row s of the
      // inverse, X(s, j) = -X(s, s) sum_{k=j}^{s-1} L(s, k) X(k, j) (4 (s - j) + 4 MFMAs), and 6 - s tiles of the LEFT-looking trailing update --
      // block column s + 2 receives the columns 0 .. s it has not seen yet (column s + 1 follows in the next slot's column update), every
      // tile's columns in ascending order: the same operations in the same order as a right-looking sweep, but 24 / 40 / 48 / 48 / 40 / 24
      // MFMAs per slot instead of 84 / 60 / 40 / 24 / 12 / 4 (the early slots, where only four waves may issue MFMAs, were bound by them).
      // By length: rank 0 the sum j = 0, then the tiles (4 (s + 1) each), then the sums j = 1 ..; the two waves of a SIMD share its pipe, so
      // a pair takes ranks r and 5 - r: (1, 6) -> 0, 5; (2, 3) -> 1, 4; (4, 5) -> 2, 3.  While the second pivot wave works (jn <= 3) its
      // SIMD issues no MFMAs (diag_logical_wave): waves 2 .. 5 take ranks {0, 4}, {2}, {1, 5}, {3}.  Slot 7: seven sums, the longest on the
      // pivot wave's quiet SIMD.
      int r0 = -1, r1 = -1;
      if (s == 7) { r0 = wave == 0 ? 6 : jh; }                     // (the owners of Q(7, .) finish their tiles; the pivot wave takes j = 6)
      else if (s >= 3) { r0 = wave == 1 ? 0 : wave == 6 ? 5 : wave == 2 ? 1 : wave == 3 ? 4 : wave == 4 ? 2 : 3; }
      else if (wave >= 2 && wave <= 5) { r0 = wave == 2 ? 0 : wave == 3 ? 2 : wave == 4 ? 1 : 3; r1 = wave == 2 ? 4 : wave == 4 ? 5 : -1; }
      const int ntl = s <= 5 ? 6 - s : 0;                        // tiles (kc + t, kc), t = 0 .. ntl-1, of block column kc = s + 2
      for (int pass = 0; pass < 2; ++pass) {
        const int rk = pass == 0 ? r0 : r1;
        if (rk < 0) continue;
        // item of rank rk: a tile (t >= 0) or a sum (j >= 0)
        int t = -1, j = -1;
        if (s == 7) j = rk;
        else if (s == 0) t = rk;
        else if (rk == 0) j = 0;
        else if (rk <= ntl) t = rk - 1;
        else j = rk - ntl;
        if (t >= 0 && t < ntl) {
          const int kc = s + 2, i0 = kc + t;
          T* C0 = S + dblk(i0, kc);
          acc_t acc0;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc0[r] = C0[N_::drow(lq, r) * BP + lr];
          // (the operands of column c + 1 are in flight while the MFMAs of column c run)
          T ca0[4], cb[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { ca0[kk] = (S + dblk(i0, 0))[lr * BP + kk * 4 + lq]; cb[kk] = (S + dblk(kc, 0))[lr * BP + kk * 4 + lq]; }
          for (int c = 0; c <= s; ++c) {
            T na0[4], nb[4];
            const int cn = c < s ? c + 1 : c;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { na0[kk] = (S + dblk(i0, cn))[lr * BP + kk * 4 + lq]; nb[kk] = (S + dblk(kc, cn))[lr * BP + kk * 4 + lq]; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc0 = N_::mfma(-ca0[kk], cb[kk], acc0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { ca0[kk] = na0[kk]; cb[kk] = nb[kk]; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) C0[N_::drow(lq, r) * BP + lr] = acc0[r];
        }
        if (pass == 0) DIAG_STAMP_ROLE(8 * s + 1);
        if (s == 7 && j >= 0 && j == jh) {
          q7_terms(7);
          const T* Xss = S + dblk(7, 7);
          acc_t res;
#pragma unroll
          for (int i = 0; i < 4; ++i) res[i] = (T)0;
#pragma unroll
          for (int i = 0; i < 4; ++i) res = N_::mfma(q7[i], -Xss[lr * BP + N_::drow(lq, i)], res);
#pragma unroll
          for (int i = 0; i < 4; ++i) Linv[(long)(7 * 16 + lr) * DB + j * 16 + N_::drow(lq, i)] = res[i];
        } else if (j >= 0 && j < s) {
          acc_t q;
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = (T)0;
          T ca[4], cb[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { ca[kk] = (S + dblk(s, j))[lr * BP + kk * 4 + lq]; cb[kk] = (S + dblk(j, j))[(kk * 4 + lq) * BP + lr]; }
          for (int k = j; k < s; ++k) {
            T na[4], nb[4];
            const int kn = k + 1 < s ? k + 1 : k;                 // (next term's operands in flight under this term's MFMAs)
            const T* A_ = S + dblk(s, kn);
            const T* B_ = S + dblk(kn, j);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { na[kk] = A_[lr * BP + kk * 4 + lq]; nb[kk] = B_[(kk * 4 + lq) * BP + lr]; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) q = N_::mfma(ca[kk], cb[kk], q);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { ca[kk] = na[kk]; cb[kk] = nb[kk]; }
          }
          if (pass == 0) DIAG_STAMP_ROLE(8 * s + 2);
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
      if (s >= 3 && s < 7 && jh >= 0) q7_terms(s);      // the term(s) of Q(7, jh) that became available (k < s); the slots with four MFMA waves have none to spare
    }
    DIAG_STAMP_ROLE(8 * s + 4);
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
                                                                   T* __restrict__ copy_dst, int reps) {
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
  // A riding workgroup takes `reps` consecutive tile pairs (option ride_reps; 1 by default).  Beside the trailing update of a lockstep batch a
  // workgroup of this kernel gets onto a CU only when one of its two resident update workgroups retires (their 8 waves hold all of the CU's
  // vector registers): the 6 720 two-tile workgroups of a 160-member launch are admitted at ~1.5 per microsecond -- 4.4 ms per launch for
  // 0.5 ms of MFMA work.  Longer runs per admitted workgroup make the launch 3-4 x shorter -- and the update launches beside it as much
  // longer: the batch is bound by the sum of its MFMA work (docs/EXPERIMENTS.md), so the default stays 1.  Same tiles, same arithmetic.
  const int bz = u / wgs, w = u - bz * wgs;
  const int e = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
    const int t0 = 2 * (w * reps + rep);
    if (t0 >= ntile) break;           // (uniform)
    if (rep) __syncthreads();         // every wave is done with the engines' LDS images of the last pair
    int t = t0 + e;
    const bool own = t < ntile;
    if (!own) t = ntile - 1;          // odd tile count: the last pair's second engine shadows the first (no store)
    int bi = 0, bj = 0;
    gemm_tile_coords(g, t, bi, bj);
    GemmArgsT<T> gl = g;
    if (Balt != nullptr && bj < 2) {  // block column c+1: its rows of column c are still in the link's scratch block
      gl.B = Balt; gl.ldb = DB; gl.sB = sBalt;
    }
    int tl = (int)threadIdx.x & 255;
    asm volatile("" : "+v"(tl));      // (the tile body's per-lane offsets are recomputed per pair: hoisted out of this loop they cost the kernel 20 spilled registers)
    gemm_tile_body<T, 64, 64, 2, 2, GEMM_SUB, false, 2>(gl, bi, bj, (long)bz, tl, smem_raw + e * gemm_lds_bytes<T, 64, 64, false>(), own);
  }
}

}  // namespace sigp
