// Tile-shape experiments of the trailing update (128 x 256 and 128 x 64 workgroup tiles: bit-identical results, both measured slower,
// docs/EXPERIMENTS.md).  NOT product code: compiled into libsigp_debug.so only (-DSIGP_DEBUG_TOOLS), included by syrk128.hpp under that
// switch so that the launchers of sigp_debug.inc / tools/syrk_bench.py can reach them; the product kernel is csrc/syrk128.hpp.
#pragma once
#ifndef SIGP_DEBUG_TOOLS
#error "csrc/debug/ is for libsigp_debug.so only"
#endif

namespace sigp {

// ---- 128 x 256 workgroup tile (8 waves) -------------------------------------------------------------------------------------
// The same update, C[128 x 256] -= A[128 x K] B[256 x K]^T, by 8 waves (2 x 4 quadrants of 64 x 64): one A slice serves two
// 128-column blocks, so a K-slice moves 48 KiB global -> LDS for two tiles instead of 64 KiB (6 instead of 8 DMA instructions per
// wave and slice), at the same occupancy (8 waves per CU: one 96-KiB workgroup instead of two 64-KiB ones).  Same MFMA sequence per
// 16x16 accumulator, same k order: results bit-identical to syrk128_kernel.  Used for lower updates with an even number of
// 128-column blocks (the second block of a diagonal tile lies above the diagonal: computed and stored like the upper halves of
// syrk128_kernel's diagonal tiles -- nothing reads it).
constexpr int SYW_LDS_BYTES = 2 * (SY_T + 2 * SY_T) * SY_SLICE_BYTES;   // 2 buffers x (128 A rows + 256 B rows) x 128 B = 96 KiB
__host__ __device__ inline int syrk_wide_tiles(int r0, int r1, int c0, int c1) {   // lower: column tile j covers blocks c0+2j, c0+2j+1
  int n = 0;
  for (int c = c0; c < c1; c += 2) { const int lo = c > r0 ? c : r0; if (r1 > lo) n += r1 - lo; }
  return n;
}
template <typename T>
__global__ __launch_bounds__(512) void syrk_wide_kernel(GemmArgsT<T> g) {
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v16_t v16_t;
  constexpr int KTe = N_::KT, NE = N_::NE;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = (T*)smem_raw;                 // [2][128][KT]
  T* Bs = As + 2 * SY_T * KTe;          // [2][256][KT]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lq = lane >> 4;
  int bi = 0, bj = 0;
  {
    int rem = blockIdx.x; bool found = false;
    for (int c = g.c0; c < g.c1; c += 2) {
      const int lo = c > g.r0 ? c : g.r0, cnt = g.r1 - lo;
      if (cnt <= 0) continue;
      if (rem < cnt) { bj = c; bi = lo + rem; found = true; break; }
      rem -= cnt;
    }
    if (!found) return;
  }
  const long bz = blockIdx.y;
  const T* Ag = g.A + bz * g.sA + (long)bi * SY_T * g.lda;
  const T* Bg = g.B + bz * g.sB + (long)bj * SY_T * g.ldb;
  T* Cw = g.C + bz * g.sC + ((long)bi * SY_T + wm * 64) * g.ldc + (long)bj * SY_T + wn * 64;
  const int K = g.K;

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = -Cw[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr];

  // DMA coordinates: per wave-instruction 8 rows x 128 B; lane -> (row = lane>>3, slot' = lane&7); 64 rows per round of the 8 waves
  const int drow_ = wave * 8 + (lane >> 3);
  const int dks = ((lane & 7) ^ ((drow_ >> 1) & 7)) * NE;
  const T* Asrc = Ag + (long)drow_ * g.lda + dks;
  const T* Bsrc = Bg + (long)drow_ * g.ldb + dks;
  const long a64 = 64 * g.lda, b64 = 64 * g.ldb;
#define SYW_ISSUE(k0, buf)                                                                                        \
  {                                                                                                               \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                 \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Asrc + p * a64 + (k0)),                                        \
                                       (lds_ptr_t)(As + ((buf) * SY_T + p * 64 + wave * 8) * KTe), 16, 0, 0);     \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                 \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bsrc + p * b64 + (k0)),                                        \
                                       (lds_ptr_t)(Bs + ((buf) * 2 * SY_T + p * 64 + wave * 8) * KTe), 16, 0, 0); \
  }
  const int x = (lr >> 1) & 7;
  const int fo0 = ((lq ^ x) & 7) * NE, fo1 = (((4 + lq) ^ x) & 7) * NE;
  const int arow0 = (wm * 64 + lr) * KTe, brow0 = (wn * 64 + lr) * KTe;
  const int nst = K / KTe;
  v16_t a0[4], b0[4], a1[4], b1[4];
  SYW_ISSUE(0, 0);
  __syncthreads();
  if (nst > 1) SYW_ISSUE(KTe, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(As + arow0 + i * 16 * KTe + fo0);
#pragma unroll
  for (int j = 0; j < 4; ++j) b0[j] = *(const v16_t*)(Bs + brow0 + j * 16 * KTe + fo0);
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    const T* Ab = As + buf * SY_T * KTe + arow0;
    const T* Bb = Bs + buf * 2 * SY_T * KTe + brow0;
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = *(const v16_t*)(Ab + i * 16 * KTe + fo1);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = *(const v16_t*)(Bb + j * 16 * KTe + fo1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N_::mfma(a0[i][e], b0[j][e], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nst) SYW_ISSUE((s + 2) * KTe, buf);
    if (s + 1 < nst) {
      const T* An = As + (buf ^ 1) * SY_T * KTe + arow0;
      const T* Bn = Bs + (buf ^ 1) * 2 * SY_T * KTe + brow0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(An + i * 16 * KTe + fo0);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = *(const v16_t*)(Bn + j * 16 * KTe + fo0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N_::mfma(a1[i][e], b1[j][e], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SYW_ISSUE
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr] = -acc[i][j][r];
}

// ---- 128 x 64 workgroup tile (4 waves of 64 x 32), three workgroups per CU ---------------------------------------------------
// The opposite of the wide tile: half the tile's columns, so a workgroup needs 48 KiB of LDS and ~130 VGPRs per wave and THREE of
// them fit on a CU.  What the update kernel loses cycles to is not its operand stream but the moments when every resident
// workgroup of a CU is outside its K loop (C-tile load / store under memory load): a third workgroup covers more of them.  Costs
// 50 % more global -> LDS bytes and fragment reads per flop.  Same MFMA sequence per 16x16 accumulator and k order: bit-identical.
constexpr int SYN_LDS_BYTES = 2 * (SY_T + SY_T / 2) * SY_SLICE_BYTES;   // 2 buffers x (128 A rows + 64 B rows) x 128 B = 48 KiB
template <typename T>
__global__ __launch_bounds__(256, 3) void syrk_n64_kernel(GemmArgsT<T> g) {
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v16_t v16_t;
  constexpr int KTe = N_::KT, NE = N_::NE;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = (T*)smem_raw;                 // [2][128][KT]
  T* Bs = As + 2 * SY_T * KTe;          // [2][64][KT]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  int bi, bj;
  if (!gemm_tile_coords(g, (int)(blockIdx.x >> 1), bi, bj)) return;      // two workgroups per 128 x 128 tile: its column halves
  const int half = blockIdx.x & 1;
  const long bz = blockIdx.y;
  const T* Ag = g.A + bz * g.sA + (long)bi * SY_T * g.lda;
  const T* Bg = g.B + bz * g.sB + ((long)bj * SY_T + half * 64) * g.ldb;
  T* Cw = g.C + bz * g.sC + ((long)bi * SY_T + wm * 64) * g.ldc + (long)bj * SY_T + half * 64 + wn * 32;
  const int K = g.K;

  acc_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = -Cw[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr];

  const int drow_ = wave * 8 + (lane >> 3);                  // 32 rows per round of the 4 waves
  const int dks = ((lane & 7) ^ ((drow_ >> 1) & 7)) * NE;
  const T* Asrc = Ag + (long)drow_ * g.lda + dks;
  const T* Bsrc = Bg + (long)drow_ * g.ldb + dks;
  const long a32 = 32 * g.lda, b32 = 32 * g.ldb;
#define SYN_ISSUE(k0, buf)                                                                                     \
  {                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                              \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Asrc + p * a32 + (k0)),                                     \
                                       (lds_ptr_t)(As + ((buf) * SY_T + p * 32 + wave * 8) * KTe), 16, 0, 0);  \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                              \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bsrc + p * b32 + (k0)),                                     \
                                       (lds_ptr_t)(Bs + ((buf) * 64 + p * 32 + wave * 8) * KTe), 16, 0, 0);    \
  }
  const int x = (lr >> 1) & 7;
  const int fo0 = ((lq ^ x) & 7) * NE, fo1 = (((4 + lq) ^ x) & 7) * NE;
  const int arow0 = (wm * 64 + lr) * KTe, brow0 = (wn * 32 + lr) * KTe;
  const int nst = K / KTe;
  v16_t a0[4], b0[2], a1[4], b1[2];
  SYN_ISSUE(0, 0);
  __syncthreads();
  if (nst > 1) SYN_ISSUE(KTe, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(As + arow0 + i * 16 * KTe + fo0);
#pragma unroll
  for (int j = 0; j < 2; ++j) b0[j] = *(const v16_t*)(Bs + brow0 + j * 16 * KTe + fo0);
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    const T* Ab = As + buf * SY_T * KTe + arow0;
    const T* Bb = Bs + buf * 64 * KTe + brow0;
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = *(const v16_t*)(Ab + i * 16 * KTe + fo1);
#pragma unroll
    for (int j = 0; j < 2; ++j) b1[j] = *(const v16_t*)(Bb + j * 16 * KTe + fo1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = N_::mfma(a0[i][e], b0[j][e], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nst) SYN_ISSUE((s + 2) * KTe, buf);
    if (s + 1 < nst) {
      const T* An = As + (buf ^ 1) * SY_T * KTe + arow0;
      const T* Bn = Bs + (buf ^ 1) * 64 * KTe + brow0;
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = *(const v16_t*)(An + i * 16 * KTe + fo0);
#pragma unroll
      for (int j = 0; j < 2; ++j) b0[j] = *(const v16_t*)(Bn + j * 16 * KTe + fo0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = N_::mfma(a1[i][e], b1[j][e], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SYN_ISSUE
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr] = -acc[i][j][r];
}

}  // namespace sigp
