// fp64 MFMA tile GEMM for gfx950:  C (-)= A * B^T  (or A * B), the kernel behind
//   - the panel TRSM            L21 = A21 * inv(L11)^T        (north/June1st.py:265 np.linalg.cholesky)
//   - the inner / outer trailing updates  A22 -= L21 L21^T     (same call; >99 % of the flops)
//   - the forward / backward block solves of predict and alpha (north/June1st.py:266, 274)
//   - the reference kernel build  X Sigma X^T                  (north/June1st.py:265 multi_dot)
//
// Design (CDNA4): 256 threads = 4 waves; each wave owns a (TM/WM)x(TN/WN) register tile made of
// 16x16 v_mfma_f64_16x16x4_f64 accumulators.  A and B tiles are staged global -> VGPR -> LDS in
// K-slices of 16 with one barrier per slice (double buffered); the LDS images are [row][k] with a
// pitch of 18 doubles so that the per-lane 8-byte fragment reads (row = lane&15, k = lane>>4) hit 32
// distinct 8-byte bank slots per 32-lane group (conflict free).  Subtraction is folded into the A
// staging (A is negated once on its way to LDS) so the accumulator is initialised from C and written
// back once: C traffic is one read + one write per tile, whatever K is.
#pragma once
#include <hip/hip_runtime.h>

namespace sigp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Element-type traits.  A K-slice is always 128 bytes per row (16 doubles / 32 floats) and a staging access is
// always 16 bytes, so the byte geometry of every LDS image and DMA is the same for fp64 and fp32.
// MFMA 16x16x4: A/B one element per lane (row/col = lane&15, k = lane>>4); D rows differ:
//   f64: row = (lane>>4) + 4*reg      f32: row = 4*(lane>>4) + reg
template <typename T> struct Num;
template <> struct Num<double> {
  typedef d4 acc_t;  typedef d2 v16_t;  typedef d2 v2_t;
  static constexpr int NE = 2;        // elements per 16 bytes
  static constexpr int KT = 16;       // k-slice per LDS stage
  static constexpr int LDP = 18;      // LDS pitch (elements) of a [row][k] image: conflict-free 8-byte fragment reads
  __device__ static inline acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  __device__ static inline int drow(int lq, int r) { return lq + 4 * r; }
  // 16-byte-chunk swizzle of row R of a [rows][128] C image in LDS: a 32-lane ds_read_b64 pass sees lq = 0,1 (rows R, R+1)
  // x 16 consecutive doubles; shifting odd rows by 128 B makes the two rows cover all 64 banks once
  __device__ static inline int cswz(int R) { return (R & 1) << 3; }
};
template <> struct Num<float> {
  typedef f4 acc_t;  typedef f4 v16_t;  typedef f2 v2_t;
  static constexpr int NE = 4;
  static constexpr int KT = 32;
  static constexpr int LDP = 36;      // 144-byte rows (16-byte aligned); 4-byte fragment reads are 2-way conflicted
  __device__ static inline acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ static inline int drow(int lq, int r) { return 4 * lq + r; }
  // fp32: a 64-lane ds_read_b32 sees 4 rows (R, R+4, R+8, R+12) x 16 consecutive floats (64 B): shift by 64 B per row group
  __device__ static inline int cswz(int R) { return ((R >> 2) & 3) << 2; }
};

constexpr int KT = Num<double>::KT;   // (fp64 names kept for the fp64-only call sites)

// Timing-ablation switches (GemmArgsT::dbg / ::stamp, kbuild flags 2|4, potrf_diag skip bits) exist for tools/*_bench.py only:
// they are live in libsigp_debug.so (-DSIGP_DEBUG_TOOLS) and compile out of the product library.
#ifdef SIGP_DEBUG_TOOLS
constexpr int DBG_MASK = ~0;
#else
constexpr int DBG_MASK = 0;
#endif

enum { GEMM_SUB = 0, GEMM_SET = 1, GEMM_SETNEG = 2 };   // C -= A B^T | C = A B^T | C = -A B^T

template <typename T>
struct GemmArgsT {
  const T* A; long lda;        // A is M x K row-major (K contiguous)
  const T* B; long ldb;        // B is N x K row-major (BT=false) or K x N row-major (BT=true)
  T* C; long ldc;
  int K;                       // multiple of the K-slice (16 doubles / 32 floats)
  int batch;                   // grid.y (0 or 1 = single problem)
  long sA, sB, sC;             // batch strides (elements): blockIdx.y selects the batch member
  // tile space in units of (TM, TN): columns bj in [c0,c1), rows bi in [max(r0, lower ? bj : r0), r1)
  int r0, r1, c0, c1;
  int lower;
  unsigned long long* stamp;   // debug: per-workgroup {s_memtime, s_memrealtime} deltas over the kernel body (nullptr = off)
  int dbg;                     // timing ablations (debug only): 1 no global loads in loop, 2 no LDS stores, 4 no barrier
  int patch;                   // lower only: 0 = column-major walk, P>0 = PxP-tile patches per XCD (old, unbalanced), P<0 = XCD-chunked
                               // walk in |P|x|P| patches over the flattened (member, tile) list (gemm_tile_coords_chunked; needs ntile)
  int ntile;                   // persistent launches (grid.y == 1, grid.x = resident workgroups): tiles per batch member; workgroup w
                               // then walks the flattened (member, tile) list w, w + grid.x, ... (member-major).  0 = one tile per workgroup.
  int ktri;                    // triangular operands: 1 = k starts at TM*bi (rows of A -- and of B in a lower SYRK -- are zero left of
                               // their diagonal block: upper-triangular factors), 2 = k stops after TN*(bj+1) (BT image upper-triangular)
  int zcount; long zA, zB, zC; // a second batch dimension (grid.z, strides in elements): lockstep members x (pairs of a level | panels)
  int zshift;                  // lower, plain walk, batched: member z's trapezoid starts zshift*z rows lower (rows bi >= bj + zshift*z) -- the
                               // block-cyclic column panels one rank owns in a sharded factor, updated in ONE launch (grid.y = own panel)
  int ride_bi1;                // syrk128_kernel (symmetric updates): 1 + the block row that is the ride-along block with at most 16 rows in use
                               // (its tiles take the RD form: 8 of 64 sub-tile pairs); 0 = no such row.  ride_rows = the rows in use (accounting)
  int ride_rows;
};
typedef GemmArgsT<double> GemmArgs;

// number of tiles of the (possibly trapezoidal) tile space; shared by host and device
__host__ __device__ inline int gemm_tile_count(int r0, int r1, int c0, int c1, int lower) {
  if (r1 <= r0 || c1 <= c0) return 0;
  if (!lower) return (r1 - r0) * (c1 - c0);
  int n = 0;
  for (int c = c0; c < c1; ++c) { int lo = c > r0 ? c : r0; if (r1 > lo) n += r1 - lo; }
  return n;
}

// Lower (trapezoid) tile spaces are walked in 8x8-tile patches, one patch at a time per XCD: workgroups b and
// b+8 land on the same XCD (round-robin dispatch), so the 64 tiles an XCD runs concurrently share 8 A row-blocks
// and 8 B row-blocks and march through K together -- each K-slice is fetched into that XCD's L2 once and
// reused 8x (column-major walking had 64 distinct A blocks per XCD: every panel read missed L2).
// Placement only affects speed, never results.
__host__ __device__ inline int gemm_patch_count(int r1, int c0, int c1, int PATCH) {
  const int prmax = (r1 - c0 + PATCH - 1) / PATCH, pc = (c1 - c0 + PATCH - 1) / PATCH;
  int n = 0;
  for (int c = 0; c < pc && c < prmax; ++c) n += prmax - c;
  return n;
}
__host__ __device__ inline int gemm_grid_size(int r0, int r1, int c0, int c1, int lower, int patch) {
  if (r1 <= r0 || c1 <= c0) return 0;
  if (!lower) return (r1 - r0) * (c1 - c0);
  if (patch <= 0) return gemm_tile_count(r0, r1, c0, c1, lower);
  const int np = gemm_patch_count(r1, c0, c1, patch);
  return (np + 7) / 8 * 8 * patch * patch;
}

template <typename T>
__device__ inline bool gemm_tile_coords(const GemmArgsT<T>& g, int b, int& bi, int& bj) {
  // Triangular operands (ktri) make the tiles' K spans unequal (up to the whole K for one row / column block, one block for the
  // opposite one): the walk then starts with the longest tiles, so the short ones fill the slots the long ones leave instead of a
  // long one starting last (top level of trtri_levels at n = 8192: 1.9 ms -> about the sum of the spans / 512 slots).
  if (!g.lower) {
    const int nr = g.r1 - g.r0, nc = g.c1 - g.c0;
    if (g.ktri == 1) {                 // K shrinks with the row block: rows ascending
      bi = g.r0 + b / nc;
      bj = g.c0 + b % nc;
      return bi < g.r1;
    }
    if (g.ktri == 2) {                 // K grows with the column block: columns descending
      if (b >= nr * nc) return false;
      bj = g.c1 - 1 - b / nr;
      bi = g.r0 + b % nr;
      return true;
    }
    bj = g.c0 + b / nr;
    bi = g.r0 + b % nr;
    return bj < g.c1;
  }
  if (g.patch <= 0 && g.ktri == 1) {   // lower, K shrinks with the row block: row by row
    int rem = b;
    const int rfirst = g.c0 > g.r0 ? g.c0 : g.r0;
    for (int r = rfirst; r < g.r1; ++r) {
      const int cnt = (r < g.c1 ? r + 1 : g.c1) - g.c0;
      if (rem < cnt) { bi = r; bj = g.c0 + rem; return true; }
      rem -= cnt;
    }
    return false;
  }
  if (g.patch <= 0) {
    int rem = b;
    const int zoff = g.zshift * (int)blockIdx.y;
    for (int c = g.c0; c < g.c1; ++c) {
      int lo = c + zoff > g.r0 ? c + zoff : g.r0;
      int cnt = g.r1 - lo;
      if (cnt <= 0) continue;
      if (rem < cnt) { bj = c; bi = lo + rem; return true; }
      rem -= cnt;
    }
    return false;
  }
  // lower, patched: rows effectively start at the column (r0 <= c0 for every caller)
  const int PATCH = g.patch;
  // rotate the patch -> XCC assignment per batch member so that half-empty diagonal patches do not always land on
  // the same XCC
  const int xcd = (b + (int)blockIdx.y) & 7, s = b >> 3;
  int P = (s / (PATCH * PATCH)) * 8 + xcd;
  const int w = s % (PATCH * PATCH);
  const int prmax = (g.r1 - g.c0 + PATCH - 1) / PATCH, pcn = (g.c1 - g.c0 + PATCH - 1) / PATCH;
  int pc = 0, pr = -1;
  for (; pc < pcn && pc < prmax; ++pc) {
    const int cnt = prmax - pc;
    if (P < cnt) { pr = pc + P; break; }
    P -= cnt;
  }
  if (pr < 0) return false;
  bj = g.c0 + pc * PATCH + (w % PATCH);
  bi = g.c0 + pr * PATCH + (w / PATCH);
  return bj < g.c1 && bi < g.r1 && bi >= bj && bi >= g.r0;
}

// XCD-chunked walk of a lower (trapezoid) tile space, all batch members flattened into one 1-D grid (g.patch < 0, g.ntile =
// tiles per member).  Workgroup b runs on XCD b % 8 (round-robin dispatch) and the chip holds 512 of them (2 per CU), so
//     b = round*512 + slot*8 + xcd   ->   u = (round*8 + xcd)*64 + slot
// hands every XCD, per round, ONE chunk of 64 consecutive tiles of the patch-ordered tile list: an 8x8 patch of one member
// (8 A row-blocks x 8 B row-blocks marching through K together, each K-slice fetched into that XCD's L2 once and used 8
// times), instead of 64 tiles scattered over 60 row-blocks.  Every chunk but the last holds exactly 64 tiles, so the XCDs
// stay balanced (the older `patch` walk padded half-empty diagonal patches with idle workgroups and lost 30 %).
// Placement only affects speed, never results.
template <typename T>
__device__ inline bool gemm_tile_coords_chunked(const GemmArgsT<T>& g, unsigned b, int& bi, int& bj, long& bz) {
  const unsigned round = b >> 9, slot = (b & 511u) >> 3, xcd = b & 7u;
  const long u_all = ((long)round * 8 + xcd) * 64 + slot;
  const int nb = g.batch > 1 ? g.batch : 1;
  if (u_all >= (long)g.ntile * nb) return false;
  bz = u_all / g.ntile;
  int u = (int)(u_all - bz * g.ntile);
  const int P = -g.patch;
  const int pcols = (g.c1 - g.c0 + P - 1) / P, prows = (g.r1 - g.c0 + P - 1) / P;
  for (int pc = 0; pc < pcols; ++pc) {
    const int cj0 = g.c0 + pc * P, w = min(P, g.c1 - cj0);
    for (int pr = pc; pr < prows; ++pr) {
      const int ri0 = g.c0 + pr * P, hh = min(P, g.r1 - ri0);
      if (hh <= 0) break;
      if (pr > pc) {
        const int cnt = w * hh;
        if (u < cnt) { bi = ri0 + u / w; bj = cj0 + u % w; return true; }
        u -= cnt;
      } else {                                   // diagonal patch: row i of the patch has min(i + 1, w) tiles
        for (int i = 0; i < hh; ++i) {
          const int cnt = min(i + 1, w);
          if (u < cnt) { bi = ri0 + i; bj = cj0 + u; return true; }
          u -= cnt;
        }
      }
    }
  }
  return false;
}

template <typename T, int TM, int TN, bool BT>
constexpr int gemm_lds_bytes() {
  return (2 * TM * Num<T>::LDP + (BT ? 2 * Num<T>::KT * (TN + 16) : 2 * TN * Num<T>::LDP)) * (int)sizeof(T);
}

// One TM x TN tile by 256 threads (tid = 0..255; `smem_raw` = gemm_lds_bytes<T, TM, TN, BT>() bytes of LDS).  The barriers inside
// are workgroup-wide: a workgroup of two such 256-thread engines (diag_update_kernel) runs them in lockstep on equal K.
// store = false: everything but the final write of the tile (an engine without a tile of its own keeps the barrier count).
template <typename T, int TM, int TN, int WM, int WN, int MODE, bool BT, int PDX = 0>
__device__ __forceinline__ void gemm_tile_body(const GemmArgsT<T>& g, int bi, int bj, long bz, int tid, char* smem_raw, bool store) {
  static_assert(WM * WN == 4, "4 waves");
  typedef Num<T> N_;
  typedef typename N_::acc_t acc_t;
  typedef typename N_::v16_t v16_t;
  constexpr int KTe = N_::KT, LDPe = N_::LDP, NE = N_::NE;
  constexpr int WTM = TM / WM, WTN = TN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int PA = TM / 32;   // 16-byte loads per thread per stage for A (TM rows x 128 B)
  constexpr int PB = TN / 32;
  constexpr int BTP = TN + 16;  // pitch of the [k][n] image (BT)
  static_assert(FM >= 1 && FN >= 1 && PA >= 1 && PB >= 1, "tile too small");

  T* As = (T*)smem_raw;               // [2][TM][LDP]
  T* Bs = As + 2 * TM * LDPe;         // [2][TN][LDP]  or  [2][KT][BTP]

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;

  const long zz = (long)blockIdx.z;
  const T* Ag = g.A + bz * g.sA + zz * g.zA + (long)bi * TM * g.lda;
  const T* Bg = BT ? g.B + bz * g.sB + zz * g.zB + (long)bj * TN : g.B + bz * g.sB + zz * g.zB + (long)bj * TN * g.ldb;
  T* Cg = g.C + bz * g.sC + zz * g.zC + ((long)bi * TM + wm * WTM) * g.ldc + (long)bj * TN + wn * WTN;

  acc_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (MODE == GEMM_SUB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = Cg[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;
      }
    }

  // staging coordinates: 8 lanes x 16 B per 128-byte row slice
  const int arow = tid >> 3, acp = (tid & 7) * NE;
  constexpr int BT_TPR = TN / NE;                                  // threads per k-row of the [k][n] image
  const int bkr = tid / BT_TPR, bcp = (tid % BT_TPR) * NE;
  constexpr int BT_RPP = 256 / BT_TPR;                             // k-rows per pass

  // Register prefetch depth: small tiles (the latency-chain launches: panel solves, inner updates, block solves) keep PD K-slices
  // in flight in registers, so a slice costs max(MFMA time, global latency / PD) instead of a whole global-load latency; the LDS
  // image stays double-buffered (a slice goes to LDS one step before it is used).  The 128x128 tile has no registers to spare.
  constexpr int PD = PDX ? PDX : ((TM * TN <= 64 * 64) ? 4 : 1);
  v16_t va[PD][PA], vb[PD][PB];
#define SIGP_GLOAD(k0, q)                                                                      \
  {                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PA; ++p)                                             \
        va[q][p] = *(const v16_t*)(Ag + (long)(arow + 32 * p) * g.lda + (k0) + acp);           \
    if (BT) {                                                                                  \
      _Pragma("unroll") for (int p = 0; p < PB; ++p)                                           \
          vb[q][p] = *(const v16_t*)(Bg + (long)((k0) + bkr + BT_RPP * p) * g.ldb + bcp);      \
    } else {                                                                                   \
      _Pragma("unroll") for (int p = 0; p < PB; ++p)                                           \
          vb[q][p] = *(const v16_t*)(Bg + (long)(arow + 32 * p) * g.ldb + (k0) + acp);         \
    }                                                                                          \
  }
#define SIGP_SSTORE(buf, q)                                                                    \
  {                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                           \
      v16_t t = va[q][p];                                                                      \
      if (MODE != GEMM_SET) t = -t;                                                            \
      *(v16_t*)(As + ((buf) * TM + arow + 32 * p) * LDPe + acp) = t;                           \
    }                                                                                          \
    if (BT) {                                                                                  \
      _Pragma("unroll") for (int p = 0; p < PB; ++p)                                           \
          *(v16_t*)(Bs + ((buf) * KTe + bkr + BT_RPP * p) * BTP + bcp) = vb[q][p];             \
    } else {                                                                                   \
      _Pragma("unroll") for (int p = 0; p < PB; ++p)                                           \
          *(v16_t*)(Bs + ((buf) * TN + arow + 32 * p) * LDPe + acp) = vb[q][p];                \
    }                                                                                          \
  }

  int Kspan = g.K;
  if (g.ktri == 1) { const int ks = bi * TM; Ag += ks; Bg += BT ? (long)ks * g.ldb : (long)ks; Kspan -= ks; }
  if (g.ktri == 2) Kspan = min(Kspan, (bj + 1) * TN);
  const int nst = Kspan / KTe;
  const int dbg = g.dbg & DBG_MASK;
#pragma unroll
  for (int q = 0; q < PD; ++q)
    if (q < nst) SIGP_GLOAD(q * KTe, q);
  SIGP_SSTORE(0, 0);
  __syncthreads();
  // invariant at step s: LDS buffer s&1 holds slice s, register set (s+j) % PD holds slice s+j for 1 <= j < PD, set s % PD is free
  for (int s0 = 0; s0 < nst; s0 += PD) {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      const int s = s0 + q;
      if (s < nst) {
        const int buf = s & 1;
        if (s + PD < nst && !(dbg & 1)) SIGP_GLOAD((s + PD) * KTe, q);
        const T* Ab = As + (buf * TM + wm * WTM + lr) * LDPe + lq;
        const T* Bb = BT ? Bs + (buf * KTe + lq) * BTP + wn * WTN + lr : Bs + (buf * TN + wn * WTN + lr) * LDPe + lq;
#pragma unroll
        for (int kk = 0; kk < KTe / 4; ++kk) {
          T a[FM], b[FN];
#pragma unroll
          for (int i = 0; i < FM; ++i) a[i] = Ab[i * 16 * LDPe + kk * 4];
#pragma unroll
          for (int j = 0; j < FN; ++j) b[j] = BT ? Bb[kk * 4 * BTP + j * 16] : Bb[j * 16 * LDPe + kk * 4];
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = N_::mfma(a[i], b[j], acc[i][j]);
        }
        if (s + 1 < nst && !(dbg & 2)) SIGP_SSTORE(buf ^ 1, (q + 1) % PD);
        if (!(dbg & 4)) __syncthreads();
      }
    }
  }
#undef SIGP_GLOAD
#undef SIGP_SSTORE

  if (!store) return;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cg[(long)(i * 16 + N_::drow(lq, r)) * g.ldc + j * 16 + lr] = acc[i][j][r];
}

template <typename T, int TM, int TN, int WM, int WN, int MODE, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgsT<T> g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int bi, bj;
  if (!gemm_tile_coords(g, blockIdx.x, bi, bj)) return;
  gemm_tile_body<T, TM, TN, WM, WN, MODE, BT>(g, bi, bj, (long)blockIdx.y, (int)threadIdx.x, smem_raw, true);
}

}  // namespace sigp
