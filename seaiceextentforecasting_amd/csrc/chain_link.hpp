// The link between two diagonal blocks of the blocked Cholesky's latency chain (north/June1st.py:265 np.linalg.cholesky -> dpotrf).
//
// Inside a panel the chain was  diagonal block c -> column solve (all rows below) -> update of column c+1 (all rows) -> diagonal
// block c+1: two small launches between two diagonal blocks, each a global-memory round trip of its own (10-11 + 7.5-9.4 us for a
// single fit, whatever the rows they cover).  Of both, the next diagonal block only needs ONE 128x128 block each:
//     L[c+1, c] = A[c+1, c] inv(L_cc)^T            and            A[c+1, c+1] -= L[c+1, c] L[c+1, c]^T.
// chain_link_kernel does exactly that in one launch, and everything else of the two launches leaves the chain:
//   * workgroups 0..35 ("chain"): one 16x16 tile (ti, tj) of the lower triangle of block (c+1, c+1) each.  A workgroup solves the
//     two 16-row pieces of A[c+1, c] its tile needs against inv(L_cc) itself (redundantly: the block is tiny, the chain is what
//     counts; the zero half of the triangular inverse is skipped), keeps them in LDS and applies them to its tile.  The solved
//     rows of L[c+1, c] go to a SCRATCH block, not in place: the other workgroups of this launch still read the unsolved rows.
//     (The diagonal-block launch that follows copies the scratch block into the matrix; its riding update of column c+1 reads it
//     from the scratch.)
//   * the other workgroups ("ride"): the column solve of the rows BELOW block row c+1, in place, 16 rows per workgroup with the
//     same triangular-skipping K loop (a quarter of the matrix instructions per wave of the stand-alone solve launch's 32-row tiles:
//     the launch is over when its longest workgroup is).
// The update of column c+1's rows below its diagonal block rides in the next diagonal-block launch with the panel's other columns.
// Same operations on the same data in the same k order as the two launches it replaces: the factor is bit-identical (the products
// with the inverse's structural zeros, which this kernel skips, add exact zeros).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"

namespace sigp {

constexpr int LINK_NB = 128;           // block size (= the factorisation's column-block width)
constexpr int LINK_CHAIN_WGS = 36;     // 16x16 tiles of the lower triangle of a 128x128 block
constexpr int LINK_LP = 132;           // pitch (elements) of the [32][128] image of a workgroup's solved rows in LDS

template <typename T>
struct LinkArgsT {
  T* Acol; long ld;     // block (c+1, c) of the matrix: rows of block column c from block row c+1 down (row stride ld)
  const T* Linv;        // [128][128] row-major inverse of L_cc (strictly-upper part zero)
  T* Cdiag;             // block (c+1, c+1)
  T* scratch;           // [128][128] row-major: L[c+1, c]
  long sM, sL, sS;      // lockstep-member strides (elements) of the matrix, of the inverse blocks, of the scratch blocks
  int rows_ride;        // 128-row blocks below block row c+1 whose column solve rides in this launch
};

template <typename T> constexpr int link_lds_bytes() { return gemm_lds_bytes<T, 32, LINK_NB, false>(); }

// RIDE = false: chain workgroup t (tile (ti, tj) of block (c+1, c+1)).  RIDE = true: rows [16 t, 16 t + 16) below block row c+1, solved in place.
template <typename T, bool RIDE>
__device__ __forceinline__ void link_chain_body(const LinkArgsT<T>& a, int t, long bz, int tid, char* smem_raw) {
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v16_t v16_t;
  constexpr int KTe = N_::KT, LDPe = N_::LDP, NE = N_::NE;
  constexpr int NST = LINK_NB / KTe;   // K slices: 8 (fp64) / 4 (fp32)
  constexpr int PD = 4;                // slices in flight in registers
  static_assert(32 * LINK_LP * (int)sizeof(T) <= link_lds_bytes<T>(), "the image of the solved rows reuses the staging buffers");
  __builtin_amdgcn_s_setprio(RIDE ? 2 : 3);   // a link of the latency chain, usually beside MFMA-saturating update waves
  int ti = 0;
  if (!RIDE) { while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; }
  const int tj = RIDE ? 0 : t - ti * (ti + 1) / 2;
  if (RIDE) ti = LINK_NB / 16 + t;     // 16-row pieces counted from block row c+1
  const bool diag = RIDE || ti == tj;  // (one row piece only)
  const T* Ag = a.Acol + bz * a.sM;
  const T* Bg = a.Linv + bz * a.sL;
  T* Cg = a.Cdiag + bz * a.sM + (long)(16 * ti) * a.ld + 16 * tj;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  T* As = (T*)smem_raw;                // [2][32][LDP]: rows 0..15 = block rows 16 ti.., rows 16..31 = block rows 16 tj..
  T* Bs = As + 2 * 32 * LDPe;          // [2][128][LDP]: the inverse, [n][k]
  T* Ls = (T*)smem_raw;                // after the K loop: [32][LINK_LP] solved rows

  acc_t cacc;                          // wave 0: the tile of block (c+1, c+1), asked for first
#pragma unroll
  for (int r = 0; r < 4; ++r) cacc[r] = (T)0;
  if (!RIDE && wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) cacc[r] = Cg[(long)N_::drow(lq, r) * a.ld + lr];
  }

  const int arow = tid >> 3, acp = (tid & 7) * NE;
  const int grow = arow < 16 ? 16 * ti + arow : 16 * tj + (arow - 16);
  const T* Ar = Ag + (long)grow * a.ld + acp;
  const T* Br = Bg + (long)arow * LINK_NB + acp;
  v16_t va[PD], vb[PD][4];
#define LINK_GLOAD(k0, q)                                                                    \
  {                                                                                          \
    va[q] = *(const v16_t*)(Ar + (k0));                                                      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) vb[q][p] = *(const v16_t*)(Br + (long)(32 * p) * LINK_NB + (k0)); \
  }
#define LINK_SSTORE(buf, q)                                                                  \
  {                                                                                          \
    *(v16_t*)(As + ((buf) * 32 + arow) * LDPe + acp) = va[q];                                \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) *(v16_t*)(Bs + ((buf) * LINK_NB + arow + 32 * p) * LDPe + acp) = vb[q][p]; \
  }
  acc_t acc[2][2];                     // [row piece][column tile]: this wave's column tiles are nt = wave and nt = 7 - wave
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][c][r] = (T)0;
#pragma unroll
  for (int q = 0; q < PD; ++q)
    if (q < NST) LINK_GLOAD(q * KTe, q);
  LINK_SSTORE(0, 0);
  __syncthreads();
#pragma unroll
  for (int s0 = 0; s0 < NST; s0 += PD) {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      const int s = s0 + q;
      if (s < NST) {
        const int buf = s & 1;
        if (s + PD < NST) LINK_GLOAD((s + PD) * KTe, q);
        const T* Ab = As + (buf * 32 + lr) * LDPe + lq;
        const T* Bb = Bs + (buf * LINK_NB + lr) * LDPe + lq;
#pragma unroll
        for (int kk = 0; kk < KTe / 4; ++kk) {
          const int k = s * KTe + 4 * kk;
          const T a0 = Ab[kk * 4], a1 = Ab[16 * LDPe + kk * 4];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int nt = c == 0 ? wave : 7 - wave;
            if (k < 16 * (nt + 1)) {   // columns 16 nt .. of the inverse's transpose are zero from row 16 (nt + 1) on
              const T b = Bb[(16 * nt) * LDPe + kk * 4];
              acc[0][c] = N_::mfma(a0, b, acc[0][c]);
              if (!diag) acc[1][c] = N_::mfma(a1, b, acc[1][c]);
            }
          }
        }
        if (s + 1 < NST) LINK_SSTORE(buf ^ 1, (q + 1) % PD);
        __syncthreads();
      }
    }
  }
#undef LINK_GLOAD
#undef LINK_SSTORE
  if (RIDE) {                          // in place: nobody else reads these rows in this launch
    T* Og = a.Acol + bz * a.sM + (long)(16 * ti) * a.ld;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int nt = c == 0 ? wave : 7 - wave;
#pragma unroll
      for (int r = 0; r < 4; ++r) Og[(long)N_::drow(lq, r) * a.ld + 16 * nt + lr] = acc[0][c][r];
    }
    return;
  }
  // the tiles of block column 0 keep their row piece of L[c+1, c] for everyone after this launch
  if (tj == 0) {
    T* Sg = a.scratch + bz * a.sS + (long)(16 * ti) * LINK_NB;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int nt = c == 0 ? wave : 7 - wave;
#pragma unroll
      for (int r = 0; r < 4; ++r) Sg[(long)N_::drow(lq, r) * LINK_NB + 16 * nt + lr] = acc[0][c][r];
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int nt = c == 0 ? wave : 7 - wave;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Ls[N_::drow(lq, r) * LINK_LP + 16 * nt + lr] = acc[0][c][r];
      if (!diag) Ls[(16 + N_::drow(lq, r)) * LINK_LP + 16 * nt + lr] = acc[1][c][r];
    }
  }
  __syncthreads();
  if (wave == 0) {                     // tile (ti, tj) -= L_i L_j^T, k ascending (the k order of the update launch it replaces)
    const T* La = Ls + lr * LINK_LP + lq;
    const T* Lb = Ls + ((diag ? 0 : 16) + lr) * LINK_LP + lq;
#pragma unroll
    for (int k = 0; k < LINK_NB; k += 4) cacc = N_::mfma(-La[k], Lb[k], cacc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Cg[(long)N_::drow(lq, r) * a.ld + lr] = cacc[r];
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void chain_link_kernel(LinkArgsT<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = (int)blockIdx.x;
  if (b < LINK_CHAIN_WGS) link_chain_body<T, false>(a, b, (long)blockIdx.y, (int)threadIdx.x, smem_raw);
  else link_chain_body<T, true>(a, b - LINK_CHAIN_WGS, (long)blockIdx.y, (int)threadIdx.x, smem_raw);
}

}  // namespace sigp
