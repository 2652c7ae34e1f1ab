// libsigp.so -- C-ABI + host-side drivers of the MI355X-native GP engine (see include/sigp.h).
// Device code: gemm_mfma.hpp (fp64 MFMA tile GEMM), potrf_diag.hpp (diagonal block), kernels_misc.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/sigp.h"
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound at run time (dlopen), see sigp_dist.inc
#include "gemm_mfma.hpp"
#include "kernels_misc.hpp"
#include "potrf_diag.hpp"
#include "chain_link.hpp"
#include "smallgp.hpp"
#include "syrk128.hpp"

using namespace sigp;

namespace {

constexpr int NB = 128;         // column-block width of the factorisation (= diagonal block)
constexpr int RIDE = 128;       // rows of the ride-along block
constexpr int MAX_SLOTS = 16;

struct ProfEvent { hipEvent_t a, b; int kclass; double flops; int K; };

// One "slot" = everything one in-flight fit needs: the augmented matrix, inverse diagonal blocks, two
// streams (update / panel) and result buffers.  Slot 0 backs the single-fit API.
struct Slot {
  int capB = 0;               // batch capacity: fits factorised in lockstep by this slot
  long cap_npad = 0;
  long matStride = 0, dinvStride = 0;   // per-member strides (elements) for the current n_pad
  double* mat = nullptr;      // [capB][(n_pad + RIDE)][n_pad]: K~ -> L~ (lower) and the ride rows below it
  double* dinv = nullptr;     // [capB][T][128][128] inverses of the diagonal blocks (strictly-upper parts stay zero)
  double* res = nullptr;      // device [capB][512]: epilogue reductions
  double* res_host = nullptr; // pinned
  int* info = nullptr;        // device [capB]
  int* info_host = nullptr;   // pinned
  KParams* kps = nullptr;     // device [capB] per-member hyper-parameters / data-set index
  KParams* kps_host = nullptr;// pinned
  double* lk = nullptr;       // [capB][128][128] scratch blocks of the fused chain link (chain_link.hpp): L[c+1, c] between the link and the next diagonal block
  double* mt = nullptr; int cap_mt = 0;   // panel_mode 1: [cap_mt][MT_LD][MT_LD] pre-multiplied top blocks (see panel_strip_kernel)
  hipStream_t s_upd = nullptr, s_pan = nullptr;
  hipEvent_t ev_pan = nullptr, ev_la = nullptr, ev_done = nullptr;
  hipEvent_t ev_group = nullptr;   // recorded on s_upd when a lockstep group has finished (head pipelining of the next group)
  hipEvent_t ev_tail = nullptr;    // recorded on s_pan when a group enters its tail (pipeline_head = 3: the next group's head starts there)
};

}  // namespace

struct sigp_handle {
  int device = 0;
  int dtype = SIGP_F64;
  std::string err;
  // problem
  long n = 0, d = 0, dp = 0, n_pad = 0, m = 0;
  double* X = nullptr;  long cap_X = 0;       // [n_pad][dp]
  double* y = nullptr;  long cap_y = 0;       // [n_pad]
  double* Xs = nullptr; long cap_Xs = 0;      // [128][dp] ride-along test rows (row j = test point j)
  double* scratchZ = nullptr; long cap_Z = 0; // [128][n_pad] second ride block (predict / alpha); predict with many points: up to 16 of them
  KParams* pred_kps = nullptr;                // chunk parameters of a prediction group
  double* T = nullptr; long cap_T = 0;        // reference kernel: X Sigma~ [n_pad][dp]
  double* Sig = nullptr; long cap_Sig = 0;    // Sigma~ padded [dp][dp]
  double* XsA = nullptr;                      // [128][dp] ride rows shifted by one (row 0 = 0) for the GEMM-form build
  double* stage = nullptr; long cap_stage = 0;// generic host->device staging
  double* gU = nullptr; long cap_gU = 0;      // MLII gradient workspaces: L~^-T, -K~^-1, dK~  ([n_pad][n_pad] each)
  double* gK = nullptr; long cap_gK = 0;
  double* gD = nullptr; long cap_gD = 0;
  double* gPart = nullptr; long cap_gPart = 0;
  KParams* gKps = nullptr; int cap_gKps = 0;  // derivative-covariance parameters of a lockstep MLII group
  double* gSig = nullptr; long cap_gSig = 0;  // MLII gradient (reference kernel): M Sigma~ padded [dp][dp] and X (M Sigma~) [n_pad][dp] --
  double* gT = nullptr; long cap_gT = 0;      // separate from Sig / T, which sigp_predict reads after a fit
  // fp32 engine (dtype == SIGP_F32): fp32 factor + fp64 iterative refinement (BASELINE configs[4])
  float* fmat = nullptr; float* fdinv = nullptr; float* fZ = nullptr; long cap_f_npad = 0; int cap_f_G = 0;
  float* fU = nullptr; float* fV = nullptr;  // [n_pad][n_pad] each: inverse-transposes of the factor's 2048-column diagonal blocks (fU, upper) and their
                                             // transposes (fV, lower) for the block triangular solves of the refinement; only the diagonal big blocks are used
  float* fXw = nullptr;                      // [4][n_pad] second working row set of those solves
  double* xq = nullptr;      // [4][n_pad] refined solutions: row 0 alpha~ = K~^-1 y, rows 1..m  w_j = K~^-1 k~*_j
  double* rq = nullptr;      // [4][n_pad] fp64 residuals
  double* rpart = nullptr;   // [REFINE_CHUNKS][4][n_pad] partial sums of a residual (krefine_residual_kernel)
  double* kq = nullptr;      // [G][n_pad][n_pad] fp64 K~ (lower triangle) of the current fp32 fit(s): written by the covariance build beside the fp32 matrix,
  size_t cap_kq = 0;         // read by the refinement's residuals (kres_lower_*_kernel); nullptr / kq_members = 0: residuals recompute the covariance on the fly
  int kq_members = 0;
  double* fpart = nullptr;   // partial sums of the final dots
  int opt_refine_iters = 3;
  int opt_refine_stored = 1; // fp32 fits: the covariance build also writes K~ in fp64 (8 n^2 bytes per member, up to 40 GB) and the refinement's residuals read it instead of recomputing it
  int opt_refine_sym = 1;    // ... in ONE pass over the stored lower triangle (kres_sym_kernel); 0: two passes (kres_lower_cols / _rows)
  int opt_refine_tol_e = 12; // stop refining once every residual is below 10^-this (relative); 0 = always refine_iters steps
  double refine_resid = 0;   // ||y - K~ alpha~||_inf / ||y||_inf after the last refinement step
  std::vector<double> kss_unit;               // k~(xs,xs) per ride test point
  std::vector<double> fit_res;                // epilogue reductions of the last fit (host copy)
  Slot slots[MAX_SLOTS];
  int nslots = 0;
  // batch data (device resident)
  double* bX = nullptr; double* by = nullptr; double* bXs = nullptr;
  long b_count = 0, b_n = 0, b_d = 0, b_dp = 0, b_m = 0, b_npad = 0;
  // small-order batches of the reference kernel (one workgroup per fit): resident data sets + per-call problem list
  std::vector<SmallSet> sm_sets;              // host copy (validation, LDS sizing)
  SmallSet* sm_sets_dev = nullptr;
  double* sm_A = nullptr; long cap_sm_A = 0;
  double* sm_y = nullptr; long cap_sm_y = 0;
  double* sm_lam = nullptr; long cap_sm_lam = 0;
  double* sm_dlam = nullptr; long cap_sm_dlam = 0; long sm_lam_len = 0; bool sm_has_dlam = false;   // lam_mode 1: derivative weights
  SmallProb* sm_probs = nullptr; long cap_sm_probs = 0;
  double* sm_out = nullptr; long cap_sm_out = 0;   // [nprob][4 + 2*mstride]
  int sm_ch = 32, sm_mmax = 0, sm_nmax = 0; long sm_lds = 0;
  int opt_small_nt64 = 0;                     // measurement switch: one wavefront per fit at orders <= 64
  // one large fit sharded over the GPUs of a node (sigp_dist_fit): this rank's block columns only
  struct DistLocal {
    bool on = false;
    int W = 0, world = 1, rank = 0, P = 0;     // panel width in 128-blocks, ranks, number of panels
    std::vector<long> lcol;                    // per panel: first local column (elements), -1 = not this rank's
    long ncol = 0;                             // local columns = row stride of mat
    void* mat = nullptr; size_t cap = 0;       // [(n_pad + 128)][ncol] (fp64 or fp32): own panels side by side, ride rows at the bottom; cap in bytes
    void* dinv = nullptr; size_t cap_dinv = 0; // [T][128][128] inverse diagonal blocks (own blocks filled); bytes
  } dl;
  // the communicator of the sharded fit: the library's own RCCL communicator (sigp_dist_init), or a caller-supplied transport
  struct DistComm {
    int nranks = 1, rank = 0;
    int kind = 0;                              // 0 = none (one rank), 1 = library RCCL communicator, 2 = caller transport (sigp_dist_init_transport)
    void* comm = nullptr;                      // ncclComm_t
    sigp_transport tr{};
    hipStream_t s_comm = nullptr;              // the panel broadcasts run here, beside the update and panel streams
    void* hstage = nullptr; size_t cap_hstage = 0;   // pinned staging buffer of a host-pointer transport
    void* pbuf[2] = {nullptr, nullptr}; size_t cap_pbuf = 0;   // two row-major panel buffers in rotation (device)
    void* sbuf[2] = {nullptr, nullptr};                        // ... and two staging buffers of the streamed segments (same size)
    void* tbuf[2] = {nullptr, nullptr}; size_t cap_tbuf = 0;   // row-split exchange: a panel's top block [W 128][W 128] + its W inverse diagonal blocks, in rotation
    hipEvent_t ev_solve[2] = {nullptr, nullptr};               // ... this rank's row piece of the panel is solved (the all-gather waits for it)
    hipEvent_t ev_seg = nullptr;                               // a segment has arrived (next owner's panel stream waits for it)
    hipEvent_t ev_hot[2] = {nullptr, nullptr};                 // row-split exchange with the row-distributed first update: the panel's first rows below its top block have arrived, solved
    hipEvent_t ev_pack[2] = {nullptr, nullptr}, ev_bcast[2] = {nullptr, nullptr}, ev_read[2] = {nullptr, nullptr}, ev_first[2] = {nullptr, nullptr}, ev_mark = nullptr;
    std::vector<hipEvent_t> ev_t;              // timing events of the last fit (dist_stats)
    // sharded triangular solves of the fp32 refinement: inverses of the own panels' diagonal blocks, panel-major work vectors
    // (element type = the handle's: fp32 for the refinement of an fp32 fit, the handle's own for sigp_dist_predict)
    void* Uinv = nullptr; void* Vinv = nullptr; size_t cap_inv = 0;          // [own panel][PW][PW] each (upper: L_pp^-T, lower: L_pp^-1)
    void* invP = nullptr; size_t cap_invP = 0;                               // scratch of one inversion
    void* ft = nullptr; void* fx = nullptr; void* fr = nullptr; void* fc = nullptr; size_t cap_vec = 0; size_t cap_ft = 0;
    // sigp_dist_predict after a sharded fit: alpha~ = K~^-1 y replicated (fp64), results of a prediction call
    bool fit_ok = false, inv_ready = false, alpha_ready = false;
    double* alpha = nullptr; size_t cap_alpha = 0; double* pred = nullptr; size_t cap_pred = 0;
    long cap_ref_npad = 0;                     // order the fp64 refinement vectors (xq, rq, rpart, fpart) of a sharded fp32 fit are sized for
    double* dinfo = nullptr;                   // device scalar for the MIN all-reduce of the pivot info
    bool dead = false;                         // a collective failed / timed out (dist_wait): every later sharded call fails until sigp_dist_shutdown
    hipEvent_t ev_wait = nullptr;              // what dist_wait polls
    int* prog = nullptr;                       // pinned host word: last panel the update stream got through (progress_kernel)
    // statistics of the last sharded fit (sigp_get_stat "dist_*")
    double st_fit_ms = 0, st_factor_ms = 0, st_bcast_bytes = 0, st_comm_ms = 0, st_stall_ms = 0, st_replicated_ms = 0, st_solve_ms = 0;
    double st_collectives = 0, st_host_comm_ms = 0, st_enqueue_ms = 0;
    double st_link_bytes = 0, st_owner_ms = 0, st_split_panels = 0, st_link_panel_max = 0;   // (.. the most one rank puts on ONE link within one panel's exchange)   // bytes this rank sends to ONE peer per fit; device time of the owner-only work per fit; panels exchanged by row pieces
    std::vector<hipEvent_t> ev_own;                                  // (pairs of stamps around the owner-only work of this rank's panels)
  } dc;
  int opt_dist_la2d = 1;                       // row-split exchange: the NEXT panel's first update divided by rows as well (see shard_factor): 0 = the next owner applies it alone
  int opt_dist_split = -1;  /* -1 = by the number of ranks (on from four ranks) */                      // sharded fit, panel exchange by ROW PIECES: the owner factors only the panel's W x W top block and broadcasts it; every rank
                                               // solves 1/world of the rows below it (scattered to it) and an all-gather assembles the panel -- instead of one
                                               // rank solving all rows and broadcasting the panel.  Same arithmetic per row: bit-identical results.
  int opt_dist_seg = 2;                        // sharded fit: column blocks per streamed broadcast segment (>= panel width: the panel travels whole)
  long opt_dist_timeout_ms = 120000;           // deadline of every host-side wait of the sharded path (dist_wait); 0 = wait for ever
  int opt_dist_stats = 0;                      // time the broadcasts and the update stream's waits for them with HIP events
  int opt_owner_only = 0;                      // sigp_set_train does not allocate the full n x n slot matrix
  // state
  int kernel_id = -1;
  double ell = 0, sn_tilde = 0;
  bool built = false, factored = false, fitted = false;
  double sigma_f = 0, nlml = 0;
  KParams kp{};
  // options
  bool outer_set = false;  // outer_blocks was given: no automatic choice (potrf_core)
  int opt_outer = 8;       // outer panel width in 128-blocks (K = 1024 trailing updates; single fits: 8.7 vs 9.0 ms at n = 8192, 34.6 vs 42.1 ms at n = 16384 against 2)
  int opt_lookahead = 1;
  int opt_small_tiles = 320; // use 64x64 tiles when the 128-tile count is below this
  int opt_tiny_tiles = 256;  // use 32x32 tiles when the 64-tile count (x members) is at most this (latency-bound single fits)
  int opt_pan_priority = 1;  // panel streams at high priority
  int opt_panel_ll = 0;      // panels up to this width are factored left-looking (0 = binary recursion only)
  int opt_panel_mode = 2;    // rows below a panel's top block: 0 recursion (trsm + updates per 128 columns), 1 strip solve (panel_strip_kernel),
                             // 2 strips when there are at least opt_strip_min strips x members to fill the chip (lockstep batches)
  int opt_strip_min = 512;
  int opt_diag_tiles = 1;    // symmetric trailing updates: a diagonal tile multiplies the 36 of 64 16 x 16 pairs on or below its diagonal (0 = the whole tile; same lower halves)
  int opt_ride_tiles = 1;    // trailing updates: tiles of the ride-along block row multiply its first 16 rows only when no more are in use (y + <= 15 test points)
  int opt_strip_tri = 1;     // strip solves skip the zero tile-slices of the inverse diagonal blocks (0 = multiply the whole 128 x 128 block: same bits, 9 % more MFMAs)
  int opt_update_dbg = 0;    // debug library: ablation bits OR-ed into the trailing updates' dbg word (8 no C load, 16 no C store: timing only, results garbage)
  int opt_c_dma = 0;         // trailing update: bring the C tile in by LDS-DMA instead of 64 accumulator-layout loads per lane (A/B switch)
  int ncu = 256;              // compute units of the device
  int opt_ride_reps = 0;      // tile pairs per riding workgroup of diag_update_kernel: 0 = one (default), n = fixed, -1 = by the launch (about four riders per CU): measured neutral, see docs/EXPERIMENTS.md
  int opt_kbuild_mfma = 2;   // covariance build: squared distances in GEMM form on the matrix pipe (kbuild_mfma_kernel): 2 = from 16 features on (below, the 2 d
                             // VALU instructions per element are not what the build waits for: same time either way), 1 = always, 0 = never (VALU)
  int opt_diag_prio = 1;     // diagonal-block kernel raises its wave priority (s_setprio 3)
  int opt_schedule = 0;      // 0 right-looking outer panels (K = 128*outer per trailing update), 1 left-looking (K grows to n)
  int opt_trsm128 = 256;     // panel solve on 128-row tiles (LDS-DMA kernel) once rows_below*members reaches this
  int opt_n64_tiles = 0;     // trailing updates on 128 x 64 workgroup tiles, 3 workgroups per CU (syrk_n64_kernel): bit 0 fp64, bit 1 fp32
  int opt_wide_tiles = 0;    // trailing updates on 128 x 256 workgroup tiles (syrk_wide_kernel): bit 0 fp64, bit 1 fp32
  int opt_syrk_v2 = 1;       // trailing update on syrk128_kernel (LDS-DMA, swizzled) instead of the generic kernel
  int opt_patch = 0;         // tile walk of the lower updates: 0 column-major, P = PxP patches per XCD
  int opt_xcd_chunks = 0;    // > 0: trailing updates with >= 512 tiles walk their tiles in XCD-sized chunks of PxP patches (P = this value)
  int opt_first_on_panel = 1;        // right-looking + look-ahead: the update of the next panel's columns runs on the panel stream: 0 never,
                                     // 1 when that panel is a latency chain (not strip-solved: single fits, small groups), 2 always
  int opt_panel_chain = 15;          // latency-chain form of a panel (right-looking, column by column; only the next column's update is a launch of its
                                     // own, the other columns' update rides in the diagonal-block launch): bit 0 panels that are not strip-solved (single
                                     // fits, small groups), bit 1 the top block of strip-solved panels (lockstep batches); 0 = binary recursion;
                                     // bit 2: the chain's two small launches between two diagonal blocks (column solve, next column's update) are
                                     // ONE launch that does only what the next diagonal block needs (chain_link_kernel), the rest rides;
                                     // bit 3: the binary recursion's leaf pairs (two columns) take that form too
  int opt_chain_rows = 80;           // (see panel_any)
  int opt_link_rows = 256;           // fused chain link while rows-below x members stays under this (chain_panel)
  int opt_strips_after_update = 1;   // right-looking schedule with look-ahead: the next panel's strip solve waits for the rest of the trailing update
  int opt_head_gate = 16;    // pipeline_head = 3: a group's tail begins when at most this many block columns remain behind the panel just enqueued
  int opt_pipeline_head = 0; // lockstep batches with >= 2 groups in flight: only the next group's head (build + first panel) overlaps the current
                             // group (measured: 312 vs 318 fits/s with one group in flight, 323 with two unrestricted -- DESIGN section 7)
  int opt_update_late = 0;   // with update_wgs: only the outer updates of the last `update_late` panels run persistent (0 = all outer updates)
  bool persist_now = false;  // set by potrf_core around the outer trailing updates it wants persistent
  int opt_update_wgs = 0;    // > 0: trailing updates with more tiles than this run as a persistent grid of this many workgroups (slots left free
                             // for the panel stream's latency chain); 0: one workgroup per tile
  int opt_group = 8;         // fits factorised in lockstep per launch in the batch path
  int opt_host_timing = 0;   // print host enqueue time per batch_run (debug)
  int opt_reserve_cus = 0;   // CUs masked out of the update streams (0: none -- the 84 KB diagonal kernel fits beside an update workgroup, and a CU-masked stream measured 6 % slower)
  // profiling
  unsigned prof = 0;         // bit k set: bracket launches of kernel class k with HIP events
  std::vector<ProfEvent> pev;
  double p_ms[SIGP_KC_COUNT] = {0};
  int64_t p_n[SIGP_KC_COUNT] = {0};
  double p_flops[SIGP_KC_COUNT] = {0};
  double p_bytes[SIGP_KC_COUNT] = {0};
};

namespace {

int fail(sigp_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail(h, SIGP_HIP_ERROR, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

long round_up(long a, long b) { return (a + b - 1) / b * b; }

void dist_release(sigp_handle* h);   // sigp_shard.inc

int ensure(sigp_handle* h, double** p, long* cap, long need) {
  if (*cap >= need) return SIGP_OK;
  if (*p) HIPCHK(h, hipFree(*p));
  *p = nullptr; *cap = 0;
  HIPCHK(h, hipMalloc((void**)p, (size_t)need * sizeof(double)));
  *cap = need;
  return SIGP_OK;
}

// update stream: every CU except the last `reserve` ones (the 150 KB-LDS diagonal kernel of the panel stream
// otherwise waits for a whole CU to drain under the trailing update: measured 170 us instead of 55 us)
int make_update_stream(sigp_handle* h, hipStream_t* st) {
  int ncu = 0;
  HIPCHK(h, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));
  const int reserve = std::min(h->opt_reserve_cus, ncu / 2);
  if (reserve <= 0) {
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(h, hipStreamCreateWithPriority(st, hipStreamNonBlocking, lo));
    return SIGP_OK;
  }
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int i = 0; i < ncu - reserve; ++i) mask[i / 32] |= (1u << (i % 32));
  HIPCHK(h, hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask.data()));
  return SIGP_OK;
}

int slot_init(sigp_handle* h, Slot& s) {
  if (s.s_upd) return SIGP_OK;
  int lo = 0, hi = 0;
  HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
  int rc = make_update_stream(h, &s.s_upd);
  if (rc) return rc;
  HIPCHK(h, hipStreamCreateWithPriority(&s.s_pan, hipStreamNonBlocking, h->opt_pan_priority ? hi : lo));
  HIPCHK(h, hipEventCreateWithFlags(&s.ev_pan, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&s.ev_la, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&s.ev_group, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&s.ev_tail, hipEventDisableTiming));
  return SIGP_OK;
}

void slot_free_buffers(Slot& s) {
  if (s.mat) (void)hipFree(s.mat);
  if (s.dinv) (void)hipFree(s.dinv);
  if (s.res) (void)hipFree(s.res);
  if (s.res_host) (void)hipHostFree(s.res_host);
  if (s.info) (void)hipFree(s.info);
  if (s.info_host) (void)hipHostFree(s.info_host);
  if (s.kps) (void)hipFree(s.kps);
  if (s.kps_host) (void)hipHostFree(s.kps_host);
  if (s.mt) (void)hipFree(s.mt);
  if (s.lk) (void)hipFree(s.lk);
  s.mt = nullptr; s.cap_mt = 0; s.lk = nullptr;
  s.mat = s.dinv = s.res = s.res_host = nullptr; s.info = s.info_host = nullptr; s.kps = s.kps_host = nullptr;
  s.capB = 0; s.cap_npad = 0;
}

// make room for B lockstep members of padded order n_pad
int slot_reserve(sigp_handle* h, Slot& s, long n_pad, int B) {
  int rc = slot_init(h, s);
  if (rc) return rc;
  s.matStride = (n_pad + RIDE) * n_pad;
  s.dinvStride = (n_pad / NB) * NB * NB;
  if (s.cap_npad >= n_pad && s.capB >= B) return SIGP_OK;
  HIPCHK(h, hipDeviceSynchronize());
  const long np = std::max(n_pad, s.cap_npad);
  const int nb = std::max(B, s.capB);
  slot_free_buffers(s);
  HIPCHK(h, hipMalloc((void**)&s.mat, (size_t)nb * (np + RIDE) * np * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&s.dinv, (size_t)nb * (np / NB) * NB * NB * sizeof(double)));
  HIPCHK(h, hipMemset(s.dinv, 0, (size_t)nb * (np / NB) * NB * NB * sizeof(double)));   // strictly-upper parts stay zero
  HIPCHK(h, hipMalloc((void**)&s.res, (size_t)nb * 512 * sizeof(double)));
  HIPCHK(h, hipHostMalloc((void**)&s.res_host, (size_t)nb * 512 * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&s.info, (size_t)nb * sizeof(int)));
  HIPCHK(h, hipHostMalloc((void**)&s.info_host, (size_t)nb * sizeof(int)));
  HIPCHK(h, hipMalloc((void**)&s.kps, (size_t)nb * sizeof(KParams)));
  HIPCHK(h, hipHostMalloc((void**)&s.kps_host, (size_t)nb * sizeof(KParams)));
  HIPCHK(h, hipMalloc((void**)&s.lk, (size_t)nb * NB * NB * sizeof(double)));
  HIPCHK(h, hipDeviceSynchronize());
  s.cap_npad = np; s.capB = nb;
  return SIGP_OK;
}

void slot_free(Slot& s) {
  slot_free_buffers(s);
  if (s.ev_pan) (void)hipEventDestroy(s.ev_pan);
  if (s.ev_la) (void)hipEventDestroy(s.ev_la);
  if (s.ev_done) (void)hipEventDestroy(s.ev_done);
  if (s.ev_group) (void)hipEventDestroy(s.ev_group);
  if (s.ev_tail) (void)hipEventDestroy(s.ev_tail);
  if (s.s_upd) (void)hipStreamDestroy(s.s_upd);
  if (s.s_pan) (void)hipStreamDestroy(s.s_pan);
  s = Slot();
}

constexpr int MT_W = 16;             // widest panel (in 128-blocks) the strip solve supports
constexpr long MT_LD = MT_W * NB;    // row stride of a member's Mt workspace
int slot_ensure_mt(sigp_handle* h, Slot& s, int nb) {
  if (s.cap_mt >= nb) return SIGP_OK;
  HIPCHK(h, hipDeviceSynchronize());
  if (s.mt) HIPCHK(h, hipFree(s.mt));
  s.mt = nullptr; s.cap_mt = 0;
  HIPCHK(h, hipMalloc((void**)&s.mt, (size_t)nb * MT_LD * MT_LD * sizeof(double)));
  s.cap_mt = nb;
  return SIGP_OK;
}

// ---- profiling brackets -------------------------------------------------------------------------
struct ProfScope {
  sigp_handle* h; hipStream_t st; ProfEvent pe; bool on;
  ProfScope(sigp_handle* h_, hipStream_t st_, int kclass, double flops, double bytes, int K = 0) : h(h_), st(st_), on((h_->prof >> kclass) & 1u) {
    h->p_flops[kclass] += flops; h->p_bytes[kclass] += bytes; h->p_n[kclass] += 1;
    if (on) {
      pe.kclass = kclass; pe.flops = flops; pe.K = K;
      (void)hipEventCreate(&pe.a); (void)hipEventCreate(&pe.b);
      (void)hipEventRecord(pe.a, st);
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(pe.b, st); h->pev.push_back(pe); }
  }
};

void prof_drain(sigp_handle* h) {
  for (auto& pe : h->pev) {
    (void)hipEventSynchronize(pe.b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) h->p_ms[pe.kclass] += ms;
    if (h->opt_host_timing >= 2) fprintf(stderr, "[sigp-launch] class %d K %d gflop %.3f ms %.4f tflops %.2f\n", pe.kclass, pe.K, pe.flops * 1e-9, ms, pe.flops / (ms * 1e-3) * 1e-12);
    (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b);
  }
  h->pev.clear();
}

// hipFuncSetAttribute is per device: remember per (kernel instantiation, device) whether the dynamic-LDS limit was raised
constexpr int MAX_DEVICES = 64;
// (distinct handles may be driven from distinct threads: the flag is only set after the attribute call returned, and
// both happen under the mutex, so no thread can launch with > 64 KB of dynamic LDS before the limit is raised)
struct AttrOnce {
  std::mutex mu;
  bool done[MAX_DEVICES] = {false};
  hipError_t set(int dev, const void* fn, int lds_bytes) {
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < MAX_DEVICES && done[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess && dev >= 0 && dev < MAX_DEVICES) done[dev] = true;
    return e;
  }
};

// ---- GEMM launch ----------------------------------------------------------------------------------
template <typename T, int TM, int TN, int WM, int WN, int MODE, bool BT>
int launch_gemm_cfg(sigp_handle* h, hipStream_t st, const GemmArgsT<T>& g) {
  const int nt = gemm_grid_size(g.r0, g.r1, g.c0, g.c1, g.lower, g.patch);
  if (nt <= 0) return SIGP_OK;
  auto kern = gemm_mfma_kernel<T, TM, TN, WM, WN, MODE, BT>;
  constexpr int lds = gemm_lds_bytes<T, TM, TN, BT>();
  static AttrOnce attr;
  HIPCHK(h, attr.set(h->device, (const void*)kern, lds));
  hipLaunchKernelGGL(kern, dim3(nt, std::max(1, g.batch), std::max(1, g.zcount)), dim3(256), lds, st, g);
  HIPCHK(h, hipGetLastError());
  return SIGP_OK;
}
// fp64 shorthand used by the fp64-only call sites
template <int TM, int TN, int WM, int WN, int MODE, bool BT>
int launch_gemm_cfg(sigp_handle* h, hipStream_t st, const GemmArgs& g) { return launch_gemm_cfg<double, TM, TN, WM, WN, MODE, BT>(h, st, g); }

template <typename T, bool SET>
int launch_syrk128_t(sigp_handle* h, hipStream_t st, const GemmArgsT<T>& g, bool may_persist = false) {
  const int nt = gemm_grid_size(g.r0, g.r1, g.c0, g.c1, g.lower, g.patch);
  if (nt <= 0) return SIGP_OK;
  static AttrOnce attr;
  HIPCHK(h, attr.set(h->device, (const void*)syrk128_kernel<T, SET>, SY_LDS_BYTES));
#ifdef SIGP_DEBUG_TOOLS   // tile-walk experiments (persistent grid, XCD-chunked walk): measured slower, libsigp_debug.so only
  if (g.dbg & (3 | 512)) {  // timing ablations of the K loop (tools/syrk_bench.py): dbg 1 no in-loop DMA, 2 no in-loop fragment reads, 512 no MFMA
    auto go = [&](auto abl) -> int {
      constexpr int A = decltype(abl)::value;
      static AttrOnce attr_a;
      HIPCHK(h, attr_a.set(h->device, (const void*)syrk128_kernel<T, SET, false, A>, SY_LDS_BYTES));
      hipLaunchKernelGGL((syrk128_kernel<T, SET, false, A>), dim3(nt, std::max(1, g.batch), std::max(1, g.zcount)), dim3(256), SY_LDS_BYTES, st, g);
      HIPCHK(h, hipGetLastError());
      return SIGP_OK;
    };
    switch ((g.dbg & 3) | ((g.dbg & 512) ? 4 : 0)) {
      case 1: return go(std::integral_constant<int, 1>{});
      case 2: return go(std::integral_constant<int, 2>{});
      case 3: return go(std::integral_constant<int, 3>{});
      case 4: return go(std::integral_constant<int, 4>{});
      case 5: return go(std::integral_constant<int, 5>{});
      case 6: return go(std::integral_constant<int, 6>{});
      default: return go(std::integral_constant<int, 7>{});
    }
  }
  const long total = (long)nt * std::max(1, g.batch);
  if (may_persist && h->persist_now && h->opt_update_wgs > 0 && g.patch == 0 && total > h->opt_update_wgs) {
    GemmArgsT<T> gp = g;
    gp.ntile = nt;
    static AttrOnce attr_p;
    HIPCHK(h, attr_p.set(h->device, (const void*)syrk128_kernel<T, SET, true>, SY_LDS_BYTES));
    hipLaunchKernelGGL((syrk128_kernel<T, SET, true>), dim3(h->opt_update_wgs, 1), dim3(256), SY_LDS_BYTES, st, gp);
    HIPCHK(h, hipGetLastError());
    return SIGP_OK;
  }
  if (may_persist && h->opt_xcd_chunks > 0 && g.lower && g.patch == 0 && g.r0 <= g.c0 && total >= 512) {
    GemmArgsT<T> gc = g;                       // XCD-chunked walk: one flat grid over (member, tile), padded to whole rounds of 512
    gc.patch = -h->opt_xcd_chunks; gc.ntile = nt;
    hipLaunchKernelGGL((syrk128_kernel<T, SET>), dim3((unsigned)round_up(total, 512), 1), dim3(256), SY_LDS_BYTES, st, gc);
    HIPCHK(h, hipGetLastError());
    return SIGP_OK;
  }
#else
  (void)may_persist;
#endif
  if constexpr (!SET) {
    // symmetric update (lower tile space over one operand): its diagonal tiles multiply their lower half only (syrk128_tile's DG form)
    if (h->opt_diag_tiles && g.lower && g.A == g.B && g.lda == g.ldb && g.sA == g.sB && g.zA == g.zB && g.r0 <= g.c0) {
      static AttrOnce attr_d;
      HIPCHK(h, attr_d.set(h->device, (const void*)syrk128_kernel<T, false, false, 0, true>, SY_LDS_BYTES));
      hipLaunchKernelGGL((syrk128_kernel<T, false, false, 0, true>), dim3(nt, std::max(1, g.batch), std::max(1, g.zcount)), dim3(256), SY_LDS_BYTES, st, g);
      HIPCHK(h, hipGetLastError());
      return SIGP_OK;
    }
  }
  hipLaunchKernelGGL((syrk128_kernel<T, SET>), dim3(nt, std::max(1, g.batch), std::max(1, g.zcount)), dim3(256), SY_LDS_BYTES, st, g);
  HIPCHK(h, hipGetLastError());
  return SIGP_OK;
}
int launch_syrk128(sigp_handle* h, hipStream_t st, const GemmArgs& g) { return launch_syrk128_t<double, false>(h, st, g); }

// C[rows r0..r1, cols c0..c1 in 128-units] -= A B^T with the tile shape picked from the tile count
template <typename T>
int gemm_sub_auto(sigp_handle* h, hipStream_t st, GemmArgsT<T> g /* in 128-units */) {
  const int nb = std::max(1, g.batch);
  const double nt1 = gemm_tile_count(g.r0, g.r1, g.c0, g.c1, g.lower);
  const int nt = (int)nt1 * nb;
  if (nt <= 0) return SIGP_OK;
  // algorithmic work: a diagonal tile of a lower (SYRK-shaped) update only needs its lower half -- the strictly-upper NB(NB-1)/2 entries are
  // not counted as flops (syrk128_kernel multiplies 36 of the 64 16 x 16 pairs of such a tile, the other tile kernels all of them)
  int ndiag = 0;
  if (g.lower) for (int c = g.c0; c < g.c1; ++c) ndiag += (c >= g.r0 && c < g.r1);
  int nride = 0;             // ... and a tile of a ride-along block row with ride_rows <= 16 rows in use counts those rows only
  if (g.ride_bi1 > 0 && g.ride_bi1 - 1 >= g.r0 && g.ride_bi1 - 1 < g.r1) nride = g.c1 - g.c0;
  const double flops = nb * (nt1 * 2.0 * NB * NB - ndiag * (double)NB * (NB - 1) - nride * 2.0 * NB * (NB - g.ride_rows)) * g.K,
               bytes = nb * (nt1 * 2.0 * NB * NB - nride * 2.0 * NB * (NB - g.ride_rows)) * sizeof(T);
  if (nt >= h->opt_small_tiles && (h->opt_syrk_v2 || sizeof(T) == 4)) {
    ProfScope ps(h, st, SIGP_KC_SYRK128, flops, bytes, g.K);
#ifdef SIGP_DEBUG_TOOLS
    // 128 x 256 workgroup tiles (syrk_wide_kernel): lower updates over an even number of column blocks, plain walk
    const int wide = sizeof(T) == 4 ? (h->opt_wide_tiles & 2) : (h->opt_wide_tiles & 1);
    if (wide && g.lower && g.patch == 0 && g.ktri == 0 && ((g.c1 - g.c0) & 1) == 0 && !h->persist_now && h->opt_xcd_chunks == 0 &&
        nt >= 4 * h->opt_small_tiles) {
      static AttrOnce wattr;
      HIPCHK(h, wattr.set(h->device, (const void*)syrk_wide_kernel<T>, SYW_LDS_BYTES));
      hipLaunchKernelGGL(syrk_wide_kernel<T>, dim3((unsigned)syrk_wide_tiles(g.r0, g.r1, g.c0, g.c1), (unsigned)nb), dim3(512), SYW_LDS_BYTES, st, g);
      HIPCHK(h, hipGetLastError());
      return SIGP_OK;
    }
    // 128 x 64 workgroup tiles, three workgroups per CU (syrk_n64_kernel)
    const int n64 = sizeof(T) == 4 ? (h->opt_n64_tiles & 2) : (h->opt_n64_tiles & 1);
    if (n64 && g.patch == 0 && g.ktri == 0 && !h->persist_now && h->opt_xcd_chunks == 0) {
      static AttrOnce nattr;
      HIPCHK(h, nattr.set(h->device, (const void*)syrk_n64_kernel<T>, SYN_LDS_BYTES));
      const int ntl = gemm_grid_size(g.r0, g.r1, g.c0, g.c1, g.lower, 0);
      hipLaunchKernelGGL(syrk_n64_kernel<T>, dim3((unsigned)(2 * ntl), (unsigned)nb), dim3(256), SYN_LDS_BYTES, st, g);
      HIPCHK(h, hipGetLastError());
      return SIGP_OK;
    }
#endif
    if (h->opt_c_dma) g.dbg |= 128;
    if (DBG_MASK && h->opt_update_dbg) g.dbg |= h->opt_update_dbg & (8 | 16);
    return launch_syrk128_t<T, false>(h, st, g, true);   // tile-walk options (xcd_chunks, update_wgs when persist_now) apply here
  }
  if (nt >= h->opt_small_tiles) {
    ProfScope ps(h, st, SIGP_KC_SYRK128, flops, bytes);
    return launch_gemm_cfg<T, 128, 128, 2, 2, GEMM_SUB, false>(h, st, g);
  }
  ProfScope ps(h, st, SIGP_KC_UPDATE_SMALL, flops, bytes);
  if (nt * 4 <= h->opt_tiny_tiles && g.K <= 2 * NB) {   // so few 64-tiles that most SIMDs would idle: 32x32 tiles, a quarter of the MFMA chain per
                                                        // wave.  Short K only: at K = 1024 the 32x32 tiles' operand traffic (4 flop/B) is what the
                                                        // launch waits for (53-61 us for the 41-tile update before the last panel of n = 4096)
    g.r0 *= 4; g.r1 *= 4; g.c0 *= 4; g.c1 *= 4; g.zshift *= 4;
    return launch_gemm_cfg<T, 32, 32, 2, 2, GEMM_SUB, false>(h, st, g);
  }
  g.r0 *= 2; g.r1 *= 2; g.c0 *= 2; g.c1 *= 2; g.zshift *= 2;   // same region in 64-units
  return launch_gemm_cfg<T, 64, 64, 2, 2, GEMM_SUB, false>(h, st, g);
}

// ---- builds -----------------------------------------------------------------------------------------
KParams make_kparams(int kernel_id, double ell, double sn, int ds) {
  KParams kp;
  kp.kernel_id = kernel_id; kp.ds = ds;
  kp.c_rbf = -0.5 / (ell * ell);
  kp.inv_ell = 1.0 / ell;
  kp.sn = sn;
  return kp;
}

// push the first nb entries of s.kps_host to the device (stream-ordered; kps_host stays untouched until retire)
int upload_kparams(sigp_handle* h, Slot& s, int nb, hipStream_t st = nullptr) {
  HIPCHK(h, hipMemcpyAsync(s.kps, s.kps_host, (size_t)nb * sizeof(KParams), hipMemcpyHostToDevice, st ? st : s.s_upd));
  return SIGP_OK;
}

// covariance build of a tile set: squared distances in GEMM form on the matrix pipe (kbuild_mfma_kernel, option kbuild_mfma, default) or
// feature by feature on the VALU (kbuild_kernel)
template <typename TO>
void launch_kbuild(sigp_handle* h, dim3 grid, hipStream_t st, const double* X, long strideX, int dp, int d, int n, TO* Mat, long strideM, long ld,
                   const KParams* kps, int flags, int colblk0 = 0, double* Mat64 = nullptr, long stride64 = 0) {
  if (h->opt_kbuild_mfma == 1 || (h->opt_kbuild_mfma == 2 && d >= 16)) {
    if (d <= 8) hipLaunchKernelGGL((kbuild_mfma_kernel<TO, 8>), grid, dim3(256), 0, st, X, strideX, dp, d, n, Mat, strideM, ld, kps, flags, colblk0, Mat64, stride64);
    else hipLaunchKernelGGL((kbuild_mfma_kernel<TO, 32>), grid, dim3(256), 0, st, X, strideX, dp, d, n, Mat, strideM, ld, kps, flags, colblk0, Mat64, stride64);
    return;
  }
  if (d <= 8 && std::is_same<TO, double>::value) hipLaunchKernelGGL((kbuild_kernel<TO, 8>), grid, dim3(256), 0, st, X, strideX, dp, d, n, Mat, strideM, ld, kps, flags, colblk0, Mat64, stride64);
  else hipLaunchKernelGGL(kbuild_kernel<TO>, grid, dim3(256), 0, st, X, strideX, dp, d, n, Mat, strideM, ld, kps, flags, colblk0, Mat64, stride64);
}

// RBF / Matern build of K~ (lower) + ride rows for the nb lockstep members of slot s.  Member b uses data set
// kps[b].ds: X + ds*strideX, y + ds*stridey, Xs + ds*strideXs.
int build_cov(sigp_handle* h, Slot& s, int nb, const double* X, long strideX, const double* y, long stridey, const double* Xs,
              long strideXs, long n, long d, long dp, long n_pad, long m, hipStream_t stb = nullptr) {
  const long ld = n_pad;
  hipStream_t st = stb ? stb : s.s_upd;
  ProfScope ps(h, st, SIGP_KC_KBUILD, nb * ((double)n * n / 2 * (3.0 * d + 20)), nb * (8.0 * n * d + 4.0 * n * (n + 1)));
  dim3 grid((unsigned)kbuild_tiles(n_pad), 1, (unsigned)nb);
  launch_kbuild<double>(h, grid, st, X, strideX, (int)dp, (int)d, (int)n, s.mat, s.matStride, ld, s.kps, 0);
  HIPCHK(h, hipGetLastError());
  dim3 g2((unsigned)((n_pad + 255) / 256), RIDE, (unsigned)nb);
  hipLaunchKernelGGL(ride_build_kernel<double>, g2, dim3(256), 0, st, X, strideX, Xs, strideXs, y, stridey, (int)dp, (int)d, (int)n,
                     (int)n_pad, (int)m, 1, s.mat + n_pad * ld, s.matStride, ld, s.kps, 1);
  HIPCHK(h, hipGetLastError());
  return SIGP_OK;
}

// ---- a panel as a latency chain ---------------------------------------------------------------------------------------------
// Block columns [J0, J0 + Wp) of nb lockstep members (already up to date), right-looking and column by column, arranged around the
// chain  diagonal block c -> what diagonal block c+1 needs -> diagonal block c+1:
//   fused link (panel_chain bit 2, chain_link.hpp): ONE launch between two diagonal blocks -- its first 36 workgroups form
//     L[c+1, c] and apply it to block (c+1, c+1); the column solve of the rows below rides in the same launch, the update of every
//     other block (column c+1 below its diagonal block, the panel's columns c+2..) rides in the launch of diagonal block c+1;
//   otherwise (and for columns with so many rows below that their solve is chip-filling work for the LDS-DMA kernel): column solve
//     (all rows), update of column c+1 (all rows), diagonal block c+1 with the update of the columns c+2.. riding.
// Same k order per tile as the binary recursion either way: bit-identical factors.
// Mm = (virtual) origin of the storage the panel's columns live in, row stride ld, member stride matStride: a slot's square
// matrices, or one rank's block columns of a sharded factor (origin shifted so that GLOBAL block indices land in it).
// rlim = one past the last row block touched (R for a whole panel, J0 + Wp for the top block of a strip-solved panel).
// on_col (may be null): called once block column c is final in the matrix (the sharded fit streams it to the other ranks).
template <typename Real>
int chain_panel(sigp_handle* h, Slot& s, hipStream_t sp, Real* Mm, long ld, long matStride, Real* dinvp, long dinvStride, int nb, int J0, int Wp, int rlim,
                const std::function<int(int)>* on_col = nullptr) {
  constexpr int diag_lds = diag_lds_bytes<Real>();
  constexpr int du_lds = std::max(diag_lds, 2 * gemm_lds_bytes<Real, 64, 64, false>());   // (fp32: the two update engines need more than the block)
  static AttrOnce du_attr, d_attr, l_attr;
  HIPCHK(h, du_attr.set(h->device, (const void*)diag_update_kernel<Real>, du_lds));
  HIPCHK(h, d_attr.set(h->device, (const void*)potrf_diag_kernel<Real>, diag_lds));
  HIPCHK(h, l_attr.set(h->device, (const void*)chain_link_kernel<Real>, link_lds_bytes<Real>()));
  const int flags = h->opt_diag_prio ? 0 : 32;
  Real* lk = (Real*)s.lk;
  auto upd_args = [&](int kc, int ccol0, int c0, int c1) {   // columns ccol0 + [c0, c1) -= (column kc)(column kc)^T, rows from each column's diagonal block to rlim
    const long o = (long)ccol0 * NB;
    GemmArgsT<Real> g{};
    g.A = Mm + o * ld + (long)kc * NB; g.lda = ld;
    g.B = g.A; g.ldb = ld;
    g.C = Mm + o * ld + o; g.ldc = ld;
    g.batch = nb; g.sA = g.sB = g.sC = matStride;
    g.K = NB; g.r0 = 0; g.r1 = rlim - ccol0; g.c0 = c0; g.c1 = c1; g.lower = 1; g.patch = 0;
    return g;
  };
  // diagonal block of column c (factor + inverse); gu (64-tile units) = an update whose tiles ride in the launch; linked: the launch follows
  // a fused link of column c - 1 (B operand of block column c from the scratch block, which is also copied into the matrix)
  auto diag = [&](int c, const GemmArgsT<Real>* gu, bool linked) -> int {
    const int ntile = gu ? gemm_grid_size(gu->r0, gu->r1, gu->c0, gu->c1, gu->lower, 0) : 0;
    const double uflops = gu ? nb * (double)ntile * 2.0 * 64 * 64 * gu->K : 0.0;
    ProfScope ps(h, sp, SIGP_KC_DIAG, nb * 2.0 * NB * NB * NB / 3 + uflops, nb * 3.0 * NB * NB * 8 + nb * (double)ntile * 2.0 * 64 * 64 * sizeof(Real));
    Real* Ac = Mm + (long)c * NB * ld + (long)c * NB;
    if (ntile > 0 || linked) {
      // tile pairs per riding workgroup: in lockstep batches so many that about four riders per CU are left (see diag_update_kernel)
      const int pairs = (ntile + 1) / 2;
      const long want = h->opt_ride_reps > 0 ? h->opt_ride_reps : (h->opt_ride_reps < 0 ? ((long)nb * pairs + 4L * h->ncu - 1) / (4L * h->ncu) : 1);
      const int reps = (int)std::max<long>(1, std::min<long>(want, 32));
      const int wgs = (pairs + reps - 1) / reps;
      GemmArgsT<Real> g0{};
      hipLaunchKernelGGL(diag_update_kernel<Real>, dim3(nb + nb * wgs + (linked ? nb : 0)), dim3(DIAG_THREADS), du_lds, sp, Ac, ld, dinvp + (long)c * NB * NB, s.info,
                         c * NB, flags, matStride, dinvStride, nb, gu ? *gu : g0, ntile, wgs, linked ? (const Real*)lk : (const Real*)nullptr, (long)NB * NB,
                         linked ? Ac - NB : (Real*)nullptr, reps);
    } else {
      hipLaunchKernelGGL(potrf_diag_kernel<Real>, dim3(nb), dim3(DIAG_THREADS), diag_lds, sp, Ac, ld, dinvp + (long)c * NB * NB, s.info, c * NB, flags, matStride,
                         dinvStride);
    }
    HIPCHK(h, hipGetLastError());
    return SIGP_OK;
  };
  // rows below the diagonal block of column c:  L[c+1.., c] = A[c+1.., c] inv(L_cc)^T
  auto solve_column = [&](int c) -> int {
    const long o = (long)(c + 1) * NB;
    const int rows_below = rlim - (c + 1);
    if (rows_below <= 0) return SIGP_OK;
    GemmArgsT<Real> g{};
    g.A = Mm + o * ld + (long)c * NB; g.lda = ld;
    g.B = dinvp + (long)c * NB * NB; g.ldb = NB;
    g.C = Mm + o * ld + (long)c * NB; g.ldc = ld;
    g.batch = nb; g.sA = g.sC = matStride; g.sB = dinvStride;
    g.K = NB; g.r0 = 0; g.c0 = 0; g.c1 = 1; g.lower = 0;
    ProfScope ps(h, sp, SIGP_KC_TRSM, nb * 2.0 * rows_below * NB * NB * NB, nb * 2.0 * rows_below * NB * NB * 8);
    if (rows_below * nb >= h->opt_trsm128) {   // enough 128-row tiles to fill the chip: the LDS-DMA kernel
      g.r1 = rows_below;
      return launch_syrk128_t<Real, true>(h, sp, g);
    }
    g.r1 = rows_below * 4;                     // few rows: 32-row tiles for parallelism
    return launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SET, false>(h, sp, g);
  };
  int rc = diag(J0, nullptr, false);
  if (rc) return rc;
  for (int i = 0; i < Wp; ++i) {
    const int c = J0 + i;
    const int rows_below = rlim - (c + 1);
    const bool last = i + 1 >= Wp;
    // (the link's ride solves any number of rows, 16 per workgroup; beyond link_rows 128-row blocks x members the stand-alone LDS-DMA solve is the better kernel)
    const bool fused = !last && (h->opt_panel_chain & 4) != 0 && rows_below >= 1 && (long)rows_below * nb < h->opt_link_rows;
    if (!fused) {
      if ((rc = solve_column(c))) return rc;
      if (on_col && (rc = (*on_col)(c))) return rc;
      if (last) break;
      if ((rc = gemm_sub_auto(h, sp, upd_args(c, c + 1, 0, 1)))) return rc;
      const int rest = Wp - i - 2;                             // columns c+2 .. J0+Wp-1
      if (rest > 0) {
        GemmArgsT<Real> gu = upd_args(c, c + 1, 1, 1 + rest);
        gu.r0 *= 2; gu.r1 *= 2; gu.c0 *= 2; gu.c1 *= 2;
        rc = diag(c + 1, &gu, false);
      } else {
        rc = diag(c + 1, nullptr, false);
      }
      if (rc) return rc;
      continue;
    }
    {
      LinkArgsT<Real> a{};
      a.Acol = Mm + (long)(c + 1) * NB * ld + (long)c * NB; a.ld = ld;
      a.Linv = dinvp + (long)c * NB * NB;
      a.Cdiag = Mm + (long)(c + 1) * NB * ld + (long)(c + 1) * NB;
      a.scratch = lk;
      a.sM = matStride; a.sL = dinvStride; a.sS = (long)NB * NB;
      a.rows_ride = rows_below - 1;
      ProfScope ps(h, sp, SIGP_KC_TRSM, nb * (2.0 * rows_below * NB * NB * NB + (double)NB * NB * NB), nb * 2.0 * rows_below * NB * NB * 8);
      hipLaunchKernelGGL(chain_link_kernel<Real>, dim3((unsigned)(LINK_CHAIN_WGS + 8 * a.rows_ride), (unsigned)nb), dim3(256), link_lds_bytes<Real>(), sp, a);
      HIPCHK(h, hipGetLastError());
    }
    GemmArgsT<Real> gu = upd_args(c, c + 1, 0, Wp - i - 1);    // columns c+1 .. J0+Wp-1 from row block c+2 down (block (c+1, c+1) is done)
    gu.r0 = 2; gu.r1 *= 2; gu.c0 *= 2; gu.c1 *= 2;
    if ((rc = diag(c + 1, &gu, true))) return rc;
    if (on_col && (rc = (*on_col)(c))) return rc;              // (block row c+1 of column c reached the matrix in that launch)
  }
  return SIGP_OK;
}

// ---- blocked Cholesky of the nb lockstep members of slot s (each augmented with its ride rows) --------
// Every launch covers the same step of all nb factorisations (grid.y / grid.x = member), so launches stay
// GPU-filling as the trailing matrices shrink and the per-step latency chain is paid once per nb fits.
// block columns per outer panel
// (a single fit of at most 24 block columns, unless the caller chose: ONE panel -- no panel boundary (each is a trailing update in series with the
//  chain), and the right-looking rides of such a panel still fit beside its diagonal blocks: n = 2048 0.670 ms (16) vs 0.715 (8), n = 3072 1.082
//  (24) vs 1.126 (16) / 1.122 (8).  From there on the rides of the first columns outlast the diagonal block: n = 4096 1.653 (8) / 1.675 (16) / 1.755 (32);
//  tools/single_sweep.py)
// (fp32 fits from 192 block columns on: the fp32 update runs its K = 1024 tile in half the time of the fp64 one, so the tile's C read +
//  write weighs twice as much -- K = 2048 instead: n = 32768 102.95 vs 104.4 ms (12: 103.3, 24: 104.1, 32: 106.3); n = 16384 19.2 vs 19.05: not there)
inline int outer_width(const sigp_handle* h, int nb, int T, bool f32 = false) {
  if (!h->outer_set && f32 && T >= 192) return 16;             // (a lockstep group of 4 at n = 32768: 97.7 vs 98.6-99.0 ms per fit)
  return (!h->outer_set && nb == 1 && T <= 24 && (h->opt_panel_chain & 4)) ? std::max(1, T) : std::max(1, h->opt_outer);
}
template <typename Real>
int potrf_core(sigp_handle* h, Slot& s, Real* M, long matStride, Real* dinvp, long dinvStride, int nb, long n_pad, bool head_on_panel = false,
               int ride_rows = RIDE) {
  const long ld = n_pad;
  const int T = (int)(n_pad / NB);   // column blocks
  const int R = T + 1;               // row blocks including the ride block
  // ride_rows: rows of the ride-along block in use (y + the test points; the rest are zero rows).  Up to 16: the block row's tiles in the
  // trailing updates multiply their first 16-row sub-tile only (syrk128_tile's RD form)
  const bool ride16 = h->opt_ride_tiles && h->opt_diag_tiles && ride_rows <= 16;     // (the RD form lives in the kernel instantiation that has the DG form)
  const int W = outer_width(h, nb, T, std::is_same<Real, float>::value);
  constexpr int diag_lds = diag_lds_bytes<Real>();
  if (std::is_same<Real, double>::value && (h->opt_panel_mode == 1 || (h->opt_panel_mode == 2 && (long)R * nb >= h->opt_strip_min))) {
    int rcm = slot_ensure_mt(h, s, nb);
    if (rcm) return rcm;
  }
  static AttrOnce diag_attr;
  HIPCHK(h, diag_attr.set(h->device, (const void*)potrf_diag_kernel<Real>, diag_lds));
  const bool la = h->opt_lookahead != 0;
  hipStream_t sp = la ? s.s_pan : s.s_upd;   // panel stream
  hipStream_t su = s.s_upd;
  // head_on_panel (lockstep batches, head pipelining): the covariance build of this group was enqueued on the PANEL stream and
  // the update stream starts with a wait for the previous group, so the build and the first panel run while the previous
  // group is still updating; nothing of this group's head may then be ordered behind the update stream
  const bool head = head_on_panel && la;
  HIPCHK(h, hipMemsetAsync(s.info, 0, (size_t)nb * sizeof(int), head ? sp : su));
  if (la && !head) {   // panel stream starts after the build on the update stream
    HIPCHK(h, hipEventRecord(s.ev_la, su));
    HIPCHK(h, hipStreamWaitEvent(sp, s.ev_la, 0));
  }

  // lower-trapezoid update  C[cols ccol0.., rows >= col .. R) -= P P^T,  P = L[:, kcol0 .. kcol0+kw)
  auto update_args = [&](int kcol0, int kw, int ccol0, int c0, int c1, int rlim) -> GemmArgsT<Real> {
    const long o = (long)ccol0 * NB;
    GemmArgsT<Real> g{};
    g.A = M + o * ld + (long)kcol0 * NB; g.lda = ld;
    g.B = g.A; g.ldb = ld;
    g.C = M + o * ld + o; g.ldc = ld;
    g.batch = nb; g.sA = g.sB = g.sC = matStride;
    g.K = kw * NB; g.r0 = 0; g.r1 = rlim - ccol0; g.c0 = c0; g.c1 = c1; g.lower = 1; g.patch = h->opt_patch;
    if (ride16 && rlim == R) { g.ride_bi1 = R - ccol0; g.ride_rows = ride_rows; }     // (block row R - 1 of the matrix = row R - 1 - ccol0 of this tile space)
    return g;
  };
  auto update = [&](hipStream_t st, int kclass, int kcol0, int kw, int ccol0, int c0, int c1, int rlim) -> int {
    (void)kclass;
    return gemm_sub_auto(h, st, update_args(kcol0, kw, ccol0, c0, c1, rlim));
  };
  // diagonal block of column c (factor + inverse); `gu` (64-tile units) = an update whose tiles ride in the same launch
  auto diag_block = [&](int c, const GemmArgsT<Real>* gu) -> int {
    const int ntile = gu ? gemm_grid_size(gu->r0, gu->r1, gu->c0, gu->c1, gu->lower, 0) : 0;
    const double uflops = gu ? nb * (double)ntile * 2.0 * 64 * 64 * gu->K : 0.0;
    ProfScope ps(h, sp, SIGP_KC_DIAG, nb * 2.0 * NB * NB * NB / 3 + uflops, nb * 3.0 * NB * NB * 8 + nb * (double)ntile * 2.0 * 64 * 64 * sizeof(Real));
    Real* Ac = M + (long)c * NB * ld + (long)c * NB;
    const int flags = h->opt_diag_prio ? 0 : 32;
    if (ntile > 0) {
      static AttrOnce du_attr;
      constexpr int du_lds = std::max(diag_lds, 2 * gemm_lds_bytes<Real, 64, 64, false>());   // (fp32: the two update engines need more than the block)
      HIPCHK(h, du_attr.set(h->device, (const void*)diag_update_kernel<Real>, du_lds));
      const int pairs = (ntile + 1) / 2;
      const long want = h->opt_ride_reps > 0 ? h->opt_ride_reps : (h->opt_ride_reps < 0 ? ((long)nb * pairs + 4L * h->ncu - 1) / (4L * h->ncu) : 1);
      const int reps = (int)std::max<long>(1, std::min<long>(want, 32));
      const int wgs = (pairs + reps - 1) / reps;
      hipLaunchKernelGGL(diag_update_kernel<Real>, dim3(nb + nb * wgs), dim3(DIAG_THREADS), du_lds, sp, Ac, ld, dinvp + (long)c * NB * NB, s.info,
                         c * NB, flags, matStride, dinvStride, nb, *gu, ntile, wgs, (const Real*)nullptr, 0L, (Real*)nullptr, reps);
    } else {
      hipLaunchKernelGGL(potrf_diag_kernel<Real>, dim3(nb), dim3(DIAG_THREADS), diag_lds, sp, Ac, ld, dinvp + (long)c * NB * NB, s.info, c * NB,
                         flags, matStride, dinvStride);
    }
    HIPCHK(h, hipGetLastError());
    return SIGP_OK;
  };
  // rows below the diagonal block of column c:  L[c+1.., c] = A[c+1.., c] inv(L_cc)^T
  auto solve_column = [&](int c, int rlim) -> int {
    const long o = (long)(c + 1) * NB;
    const int rows_below = rlim - (c + 1);   // 128-row blocks below the diagonal block (ride block included when rlim = R)
    if (rows_below <= 0) return SIGP_OK;
    GemmArgsT<Real> g{};
    g.A = M + o * ld + (long)c * NB; g.lda = ld;
    g.B = dinvp + (long)c * NB * NB; g.ldb = NB;
    g.C = M + o * ld + (long)c * NB; g.ldc = ld;
    g.batch = nb; g.sA = g.sC = matStride; g.sB = dinvStride;
    g.K = NB; g.r0 = 0; g.c0 = 0; g.c1 = 1; g.lower = 0;
    ProfScope ps(h, sp, SIGP_KC_TRSM, nb * 2.0 * rows_below * NB * NB * NB, nb * 2.0 * rows_below * NB * NB * 8);
    if (rows_below * nb >= h->opt_trsm128) {   // enough 128-row tiles to fill the chip: the LDS-DMA kernel
      g.r1 = rows_below;
      return launch_syrk128_t<Real, true>(h, sp, g);
    }
    g.r1 = rows_below * 4;                     // few rows: 32-row tiles for parallelism
    return launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SET, false>(h, sp, g);
  };
  // factor block columns [J0, J0+Wp) (already up to date) by binary recursion: the left half, a rank-(half) update
  // of the right half's columns, then the right half.
  // rlim = one past the last row block the recursion touches: R for the whole panel, J0+Wp for its top block only
  std::function<int(int, int, int)> panel_rec = [&](int J0, int Wp, int rlim) -> int {
    if (Wp == 1) {
      int rc1 = diag_block(J0, nullptr);
      return rc1 ? rc1 : solve_column(J0, rlim);
    }
    if (h->opt_panel_ll && Wp <= h->opt_panel_ll) {
      // left-looking inside a (sub)panel: column block c is updated once with all earlier columns of the panel
      // (K = 128 (c-J0)), then factored: each panel column is read/written once and the average K doubles
      for (int i = 0; i < Wp; ++i) {
        int rc;
        if (i > 0 && (rc = update(sp, SIGP_KC_UPDATE_SMALL, J0, i, J0 + i, 0, 1, rlim))) return rc;
        if ((rc = panel_rec(J0 + i, 1, rlim))) return rc;
      }
      return SIGP_OK;
    }
    if (Wp == 2 && (h->opt_panel_chain & 8) && rlim - J0 >= 2)   // the recursion's leaf pairs through the fused link: D, link, D + riding update of the second column, solve
      return chain_panel<Real>(h, s, sp, M, ld, matStride, dinvp, dinvStride, nb, J0, 2, rlim);
    const int hw = Wp / 2;
    int rc = panel_rec(J0, hw, rlim);
    if (rc) return rc;
    if ((rc = update(sp, SIGP_KC_UPDATE_SMALL, J0, hw, J0 + hw, 0, Wp - hw, rlim))) return rc;
    return panel_rec(J0 + hw, Wp - hw, rlim);
  };
  // The same panel as a latency chain (chain_panel above): right-looking column by column, bit-identical to the recursion.
  auto panel_chain = [&](int J0, int Wp, int rlim) -> int {
    return chain_panel<Real>(h, s, sp, M, ld, matStride, dinvp, dinvStride, nb, J0, Wp, rlim);
  };
  // top = the top block of a strip-solved panel.  A whole panel takes the chain form only while its riding updates (K = 128, 64x64
  // tiles: 4 flop per operand byte) stay shorter than the diagonal block they ride beside: up to chain_rows (80) 128-row blocks x
  // members below the panel's first column (n = 32768 in fp32 is 4 % faster with the recursion's K = 256 / 512 updates)
  auto chain_form = [&](int J0, int Wp, int rlim, bool top) -> bool {
    return Wp > 2 && (top ? (h->opt_panel_chain & 2) != 0 : ((h->opt_panel_chain & 1) != 0 && (long)(rlim - J0) * nb <= h->opt_chain_rows));
  };
  auto panel_any = [&](int J0, int Wp, int rlim, bool top) -> int {
    return chain_form(J0, Wp, rlim, top) ? panel_chain(J0, Wp, rlim) : panel_rec(J0, Wp, rlim);
  };
  // factor block columns [J0, J0+Wp): panel_top = everything on the panel stream up to the strip solve (the whole panel when it
  // is not strip-solved); panel_strips = the Mt products + strip kernel for the rows below the top block (no-op otherwise)
  auto use_strips = [&](int J0, int Wp) -> bool {
    const int below = R - (J0 + Wp);             // row blocks under the panel's top block (the ride block is one of them)
    return std::is_same<Real, double>::value && Wp > 1 && Wp <= MT_W && below > 0 &&
           (h->opt_panel_mode == 1 || (h->opt_panel_mode == 2 && (long)below * nb >= h->opt_strip_min));
  };
  auto panel_top = [&](int J0, int Wp) -> int {
    if (!use_strips(J0, Wp)) return panel_any(J0, Wp, R, false);
    // panel_mode 1: recursion on the top Wp x Wp block only, then every 128-row strip below it is solved by one
    // workgroup walking the panel's columns (panel_strip_kernel): the lower rows are read and written once
    return panel_any(J0, Wp, J0 + Wp, true);
  };
  auto panel_strips = [&](int J0, int Wp) -> int {
    if (!use_strips(J0, Wp)) return SIGP_OK;
    const int below = R - (J0 + Wp);
    int rc;
    Real* mt = (Real*)s.mt;
    const long mtStride = MT_LD * MT_LD;
    {
      ProfScope ps(h, sp, SIGP_KC_UPDATE_SMALL, nb * 2.0 * NB * NB * NB * (Wp * (Wp - 1) / 2), nb * 3.0 * NB * NB * 8 * (Wp * (Wp + 1) / 2));
      hipLaunchKernelGGL(mt_diag_kernel<Real>, dim3(Wp, nb), dim3(256), 0, sp, dinvp + (long)J0 * NB * NB, dinvStride, mt, MT_LD, mtStride);
      HIPCHK(h, hipGetLastError());
      for (int j = 1; j < Wp; ++j) {             // Mt[j, 0:j] = -inv(L_jj) L[j, 0:j]
        GemmArgsT<Real> g{};
        g.A = dinvp + (long)(J0 + j) * NB * NB; g.lda = NB; g.sA = dinvStride;
        g.B = M + (long)(J0 + j) * NB * ld + (long)J0 * NB; g.ldb = ld; g.sB = matStride;     // K x N row-major (BT)
        g.C = mt + (long)j * NB * MT_LD; g.ldc = MT_LD; g.sC = mtStride;
        g.batch = nb; g.K = NB; g.r0 = 0; g.r1 = 4; g.c0 = 0; g.c1 = j; g.lower = 0;
        if ((rc = launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SETNEG, true>(h, sp, g))) return rc;
      }
    }
    {
      ProfScope ps(h, sp, SIGP_KC_TRSM, nb * (double)below * 2.0 * NB * NB * NB * (Wp * (Wp + 1) / 2), nb * (double)below * 2.0 * Wp * NB * NB * 8);
      static AttrOnce strip_attr, strip_attr_full;
      StripArgsT<Real> a{M, ld, matStride, mt, MT_LD, mtStride, J0 + Wp, J0, Wp};
      if (h->opt_strip_tri) {
        HIPCHK(h, strip_attr.set(h->device, (const void*)panel_strip_kernel<Real>, SY_LDS_BYTES));
        hipLaunchKernelGGL(panel_strip_kernel<Real>, dim3(below, nb), dim3(256), SY_LDS_BYTES, sp, a);
      } else {
        HIPCHK(h, strip_attr_full.set(h->device, (const void*)panel_strip_kernel<Real, false>, SY_LDS_BYTES));
        hipLaunchKernelGGL((panel_strip_kernel<Real, false>), dim3(below, nb), dim3(256), SY_LDS_BYTES, sp, a);
      }
      HIPCHK(h, hipGetLastError());
    }
    return SIGP_OK;
  };
  auto panel = [&](int J0, int Wp) -> int {
    int rc = panel_top(J0, Wp);
    return rc ? rc : panel_strips(J0, Wp);
  };
  auto outer = [&](hipStream_t st, int J, int Wc, int c0, int c1) -> int {
    // persistent form (update_wgs): only for outer trailing updates, and with update_late only for the last panels, where the
    // updates are small and the panel chain they share the chip with is what the step waits for
    const int panels_left = (T - (J + Wc) + W - 1) / W;
    h->persist_now = h->opt_update_wgs > 0 && (h->opt_update_late == 0 || panels_left <= h->opt_update_late);
    const int rc_ = update(st, SIGP_KC_SYRK128, J, Wc, J + Wc, c0, c1, R);
    h->persist_now = false;
    return rc_;
  };

  int rc = panel(0, std::min(W, T));
  if (rc) return rc;
  if (h->opt_schedule == 1) {
    // Left-looking outer schedule: panel q (columns J..J+Wq) is brought up to date in two launches,
    //   A(q): C_q -= L[:, 0 : J-W] L[q rows, 0 : J-W]^T   (all panels but the last one: K = 128 (J-W), up to n - 2*128 W)
    //   B(q): C_q -= P_{q-1} P_{q-1}^T                    (the panel factored last: K = 128 W)
    // and then factored, F(q).  A(q+1) only needs panels 0..q-1, so it runs on the update stream while the panel
    // stream does B(q), F(q).  Each C tile is read and written twice per panel instead of once per EARLIER panel,
    // and almost all flops run at K >= 1024.  The k order of every tile's sum is the same as in the right-looking
    // schedule (panels in order, k ascending), so the factor is bit-identical.
    hipEvent_t evF[2] = {s.ev_pan, s.ev_done};
    if (la) HIPCHK(h, hipEventRecord(evF[0], sp));                 // F(0)
    int q = 1;
    for (int J = W; J < T; J += W, ++q) {
      const int Wq = std::min(W, T - J);
      if (la) {
        if (q >= 2) {
          HIPCHK(h, hipStreamWaitEvent(su, evF[q & 1], 0));          // F(q-2) done (same parity as q)
          if ((rc = update(su, SIGP_KC_SYRK128, 0, J - W, J, 0, Wq, R))) return rc;    // A(q)
          HIPCHK(h, hipEventRecord(s.ev_la, su));
          HIPCHK(h, hipStreamWaitEvent(sp, s.ev_la, 0));
        }
        if ((rc = update(sp, SIGP_KC_SYRK128, J - W, W, J, 0, Wq, R))) return rc;      // B(q), after F(q-1) in stream order
        if ((rc = panel(J, Wq))) return rc;                                             // F(q)
        HIPCHK(h, hipEventRecord(evF[q & 1], sp));
      } else {
        if ((rc = update(su, SIGP_KC_SYRK128, 0, J, J, 0, Wq, R))) return rc;          // A(q)+B(q) in one launch
        if ((rc = panel(J, Wq))) return rc;
      }
    }
    if (la) {
      HIPCHK(h, hipEventRecord(s.ev_pan, sp));
      HIPCHK(h, hipStreamWaitEvent(su, s.ev_pan, 0));
    }
    return SIGP_OK;
  }
  bool have_rest = false;                      // first_on_panel: an update of the rest of the trailing matrix is in flight on su
  bool tail_marked = false;                    // ev_tail recorded (pipeline_head = 3)
  for (int J = 0; J < T; J += W) {
    const int Wc = std::min(W, T - J);
    const int ncols = T - (J + Wc);            // trailing column blocks
    if (ncols <= 0) break;
    const int Wn = std::min(W, ncols);         // width of the next panel
    if (la && !tail_marked && ncols <= h->opt_head_gate) {   // panel J is the last one before the tail: the next group's head may start behind it
      HIPCHK(h, hipEventRecord(s.ev_tail, sp));
      tail_marked = true;
    }
    if (la && (h->opt_first_on_panel == 2 || (h->opt_first_on_panel == 1 && !use_strips(J + Wc, Wn)))) {
      // The update of the NEXT panel's columns stays on the panel stream (stream order, no inter-queue hand-off in the chain
      // panel -> first update -> next panel: each hand-off is a barrier packet pair, 11-13 us measured); the update stream gets
      // the rest of the trailing matrix, which the panel stream only has to see finished one panel later.
      if (have_rest) HIPCHK(h, hipStreamWaitEvent(sp, s.ev_done, 0));   // rest(J - W) wrote these columns too
      if ((rc = outer(sp, J, Wc, 0, Wn))) return rc;
      HIPCHK(h, hipEventRecord(s.ev_pan, sp));            // panel J and the first update done: the rest starts behind them, so the
      HIPCHK(h, hipStreamWaitEvent(su, s.ev_pan, 0));     // update the chain waits for has the chip to itself
      have_rest = ncols > Wn;
      if (have_rest) {
        if ((rc = outer(su, J, Wc, Wn, ncols))) return rc;
        HIPCHK(h, hipEventRecord(s.ev_done, su));
      }
      if (h->opt_strips_after_update && use_strips(J + Wc, Wn)) {
        if ((rc = panel_top(J + Wc, Wn))) return rc;
        if (have_rest) HIPCHK(h, hipStreamWaitEvent(sp, s.ev_done, 0));
        if ((rc = panel_strips(J + Wc, Wn))) return rc;
      } else if ((rc = panel(J + Wc, Wn))) return rc;
    } else if (la) {
      HIPCHK(h, hipEventRecord(s.ev_pan, sp));            // panel J done
      HIPCHK(h, hipStreamWaitEvent(su, s.ev_pan, 0));
      rc = outer(su, J, Wc, 0, Wn);                       // next panel's columns first
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(s.ev_la, su));
      HIPCHK(h, hipStreamWaitEvent(sp, s.ev_la, 0));
      if (h->opt_strips_after_update && use_strips(J + Wc, Wn)) {
        // only the next panel's top block (a latency chain of small launches) overlaps the rest of the update; its strip solve --
        // MFMA work for the whole chip -- starts when that update is done instead of sharing the chip with it
        if ((rc = panel_top(J + Wc, Wn))) return rc;
        if ((rc = outer(su, J, Wc, Wn, ncols))) return rc;
        HIPCHK(h, hipEventRecord(s.ev_done, su));
        HIPCHK(h, hipStreamWaitEvent(sp, s.ev_done, 0));
        if ((rc = panel_strips(J + Wc, Wn))) return rc;
      } else {
        rc = panel(J + Wc, Wn);                           // next panel overlaps the rest of the update
        if (rc) return rc;
        rc = outer(su, J, Wc, Wn, ncols);
        if (rc) return rc;
        HIPCHK(h, hipEventRecord(s.ev_done, su));         // (a later panel may take the first_on_panel form and wait for this)
      }
      have_rest = ncols > Wn;
    } else {
      rc = outer(su, J, Wc, 0, ncols);
      if (rc) return rc;
      rc = panel(J + Wc, Wn);
      if (rc) return rc;
    }
  }
  if (la) {   // join: the update stream is the slot's completion stream
    if (!tail_marked) HIPCHK(h, hipEventRecord(s.ev_tail, sp));
    HIPCHK(h, hipEventRecord(s.ev_pan, sp));
    HIPCHK(h, hipStreamWaitEvent(su, s.ev_pan, 0));
  }
  return SIGP_OK;
}

int potrf_slot(sigp_handle* h, Slot& s, int nb, long n_pad, bool head_on_panel = false, int ride_rows = RIDE) {
  return potrf_core<double>(h, s, s.mat, s.matStride, s.dinv, s.dinvStride, nb, n_pad, head_on_panel, ride_rows);
}

// stand-alone pieces of potrf_core for the multi-GPU driver (one member; Mm / dinvp = the fp64 slot matrix or the fp32 engine's)
template <typename Real>
int dist_update(sigp_handle* h, Real* Mm, hipStream_t st, long n_pad, int kcol0, int kw, int ccol0, int c0, int c1) {
  const long ld = n_pad;
  const int T = (int)(n_pad / NB), R = T + 1;
  const long o = (long)ccol0 * NB;
  GemmArgsT<Real> g{};
  g.A = Mm + o * ld + (long)kcol0 * NB; g.lda = ld;
  g.B = g.A; g.ldb = ld;
  g.C = Mm + o * ld + o; g.ldc = ld;
  g.batch = 1; g.sA = g.sB = g.sC = 0;
  g.K = kw * NB; g.r0 = 0; g.r1 = R - ccol0; g.c0 = c0; g.c1 = c1; g.lower = 1; g.patch = 0;
  return gemm_sub_auto(h, st, g);
}

// Mm = (virtual) origin of the storage the panel's columns live in, row stride ld: the slot's square matrix (ld = n_pad) or one
// rank's block columns (sigp_dist_local_*: ld = its column count, origin shifted so that GLOBAL column indices land in it)
// on_col (may be null): called right after the column solve of block column c has been enqueued -- that column is then final
// (the sharded fit streams it to the other ranks while the chain goes on)
template <typename Real>
int dist_panel_rec(sigp_handle* h, Slot& s, Real* Mm, long ld, Real* dinvp, hipStream_t sp, long n_pad, int J0, int Wp, const std::function<int(int)>* on_col = nullptr);
// the panel in its latency-chain form (potrf_core's panel_chain, one member): right-looking column by column, the update of the
// columns beyond the next one riding in the next diagonal block's launch.  Same k order per tile as the recursion: bit-identical.
template <typename Real>
int dist_panel(sigp_handle* h, Slot& s, Real* Mm, long ld, Real* dinvp, hipStream_t sp, long n_pad, int J0, int Wp, const std::function<int(int)>* on_col = nullptr) {
  const int T = (int)(n_pad / NB), R = T + 1;
  if (!(h->opt_panel_chain & 1) || Wp <= 2 || (long)(R - J0) > h->opt_chain_rows) return dist_panel_rec<Real>(h, s, Mm, ld, dinvp, sp, n_pad, J0, Wp, on_col);
  return chain_panel<Real>(h, s, sp, Mm, ld, 0L, dinvp, 0L, 1, J0, Wp, R, on_col);
}

template <typename Real>
int dist_panel_rec(sigp_handle* h, Slot& s, Real* Mm, long ld, Real* dinvp, hipStream_t sp, long n_pad, int J0, int Wp, const std::function<int(int)>* on_col) {
  const int T = (int)(n_pad / NB), R = T + 1;
  if (Wp == 1) {
    const int c = J0;
    hipLaunchKernelGGL(potrf_diag_kernel<Real>, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, sp, Mm + (long)c * NB * ld + (long)c * NB, ld,
                       dinvp + (long)c * NB * NB, s.info, c * NB, 0, 0L, 0L);
    HIPCHK(h, hipGetLastError());
    const long o = (long)(c + 1) * NB;
    const int rows_below = R - (c + 1);
    GemmArgsT<Real> g{};
    g.A = Mm + o * ld + (long)c * NB; g.lda = ld;
    g.B = dinvp + (long)c * NB * NB; g.ldb = NB;
    g.C = Mm + o * ld + (long)c * NB; g.ldc = ld;
    g.batch = 1; g.K = NB; g.r0 = 0; g.c0 = 0; g.c1 = 1; g.lower = 0;
    int rc1;
    if (rows_below >= h->opt_trsm128) { g.r1 = rows_below; rc1 = launch_syrk128_t<Real, true>(h, sp, g); }
    else { g.r1 = rows_below * 4; rc1 = launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SET, false>(h, sp, g); }
    if (rc1) return rc1;
    return on_col ? (*on_col)(c) : SIGP_OK;
  }
  const int hw = Wp / 2;
  int rc = dist_panel_rec<Real>(h, s, Mm, ld, dinvp, sp, n_pad, J0, hw, on_col);
  if (rc) return rc;
  {   // columns of the right half -= (left half)(left half)^T, rows from the right half's diagonal block down
    const long o = (long)(J0 + hw) * NB;
    GemmArgsT<Real> g{};
    g.A = Mm + o * ld + (long)J0 * NB; g.lda = ld;
    g.B = g.A; g.ldb = ld;
    g.C = Mm + o * ld + o; g.ldc = ld;
    g.batch = 1; g.K = hw * NB; g.r0 = 0; g.r1 = R - (J0 + hw); g.c0 = 0; g.c1 = Wp - hw; g.lower = 1;
    if ((rc = gemm_sub_auto(h, sp, g))) return rc;
  }
  return dist_panel_rec<Real>(h, s, Mm, ld, dinvp, sp, n_pad, J0 + hw, Wp - hw, on_col);
}

// epilogue reductions on the ride blocks of slot s + async copy of results / info to pinned host memory
int epilogue_slot(sigp_handle* h, Slot& s, int nb, long n, long n_pad, long m) {
  const long ld = n_pad;
  double* Z = s.mat + n_pad * ld;
  {
    ProfScope ps(h, s.s_upd, SIGP_KC_EPILOGUE, nb * 4.0 * (m + 1) * n, nb * 8.0 * (m + 2) * n);
    hipLaunchKernelGGL(epilogue_kernel<double>, dim3((unsigned)(m + 2), (unsigned)nb), dim3(256), 0, s.s_upd, Z, ld, Z, s.mat, ld, (int)n,
                       (int)n_pad, (int)(m + 1), s.res, s.matStride, s.matStride, s.matStride);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipMemcpyAsync(s.res_host, s.res, (size_t)nb * 512 * sizeof(double), hipMemcpyDeviceToHost, s.s_upd));
  HIPCHK(h, hipMemcpyAsync(s.info_host, s.info, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, s.s_upd));
  return SIGP_OK;
}

// host-side scalar epilogue (north/June1st.py:267-268, 246, 276-277) from the reductions
void finish_results(const double* res, int info, long n, long m, double sn_tilde, const double* kss_unit, double* out,
                    double* mean, double* var) {
  const double inf = std::numeric_limits<double>::infinity();
  if (info != 0) {
    out[0] = inf; out[1] = inf; out[2] = (double)info; out[3] = inf;
    for (long j = 0; j < m; ++j) { if (mean) mean[j] = std::nan(""); if (var) var[j] = std::nan(""); }
    return;
  }
  const double sf = res[0] / (double)n;                            // sigma_f = y^T A~ / n
  const double nlml = 0.5 * n + res[256] + 0.5 * n * std::log(sf) + 0.5 * n * std::log(2.0 * M_PI);
  out[0] = sf; out[1] = nlml; out[2] = 0.0; out[3] = sf * sn_tilde;
  for (long j = 0; j < m; ++j) {
    if (mean) mean[j] = res[1 + j];                                // k*^T alpha = v~^T z
    if (var) var[j] = sf * (kss_unit[j] + sn_tilde - res[128 + 1 + j]);
  }
}

template <typename Real>
static int solve_rows_forward_t(sigp_handle* h, hipStream_t st, const Real* Mat, const Real* dinvp, Real* Z, long n_pad, int nchunks) {
  // Z[128*nchunks][n_pad] <- Z L~^-T : per block column kb: Z[:,kb] = Z[:,kb] inv(L_kk)^T ; Z[:,kb+1:] -= Z[:,kb] L[kb+1:,kb]^T
  // nchunks > 1: all 128-row chunks advance in lockstep (grid.y = chunk)
  const long ld = n_pad;
  const int T = (int)(n_pad / NB);
  for (int kb = 0; kb < T; ++kb) {
    GemmArgsT<Real> g{};
    g.A = Z + (long)kb * NB; g.lda = ld; g.B = dinvp + (long)kb * NB * NB; g.ldb = NB; g.C = Z + (long)kb * NB; g.ldc = ld;
    g.K = NB; g.r0 = 0; g.r1 = RIDE / 32; g.c0 = 0; g.c1 = 1; g.lower = 0;
    g.batch = nchunks; g.sA = g.sC = (long)RIDE * ld; g.sB = 0;
    int rc = launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SET, false>(h, st, g);
    if (rc) return rc;
    if (kb + 1 < T) {
      GemmArgsT<Real> u{};
      u.batch = nchunks; u.sA = u.sC = (long)RIDE * ld; u.sB = 0;
      u.A = Z + (long)kb * NB; u.lda = ld;
      u.B = Mat + (long)(kb + 1) * NB * ld + (long)kb * NB; u.ldb = ld;
      u.C = Z + (long)(kb + 1) * NB; u.ldc = ld;
      u.K = NB; u.r0 = 0; u.r1 = RIDE / 64; u.c0 = 0; u.c1 = (T - kb - 1) * 2; u.lower = 0;
      rc = launch_gemm_cfg<Real, 64, 64, 2, 2, GEMM_SUB, false>(h, st, u);
      if (rc) return rc;
    }
  }
  return SIGP_OK;
}

template <typename Real>
static int solve_rows_backward_t(sigp_handle* h, hipStream_t st, const Real* Mat, const Real* dinvp, Real* Z, long n_pad) {
  // Z[128][n_pad] <- Z L~^-1 : from the last block column: Z[:,kb] = Z[:,kb] inv(L_kk) ; Z[:, :kb] -= Z[:,kb] L[kb, :kb]
  const long ld = n_pad;
  const int T = (int)(n_pad / NB);
  for (int kb = T - 1; kb >= 0; --kb) {
    GemmArgsT<Real> g{};
    g.A = Z + (long)kb * NB; g.lda = ld; g.B = dinvp + (long)kb * NB * NB; g.ldb = NB; g.C = Z + (long)kb * NB; g.ldc = ld;
    g.K = NB; g.r0 = 0; g.r1 = RIDE / 32; g.c0 = 0; g.c1 = 1; g.lower = 0;
    int rc = launch_gemm_cfg<Real, 32, 128, 1, 4, GEMM_SET, true>(h, st, g);
    if (rc) return rc;
    if (kb > 0) {
      GemmArgsT<Real> u{};
      u.A = Z + (long)kb * NB; u.lda = ld;
      u.B = Mat + (long)kb * NB * ld; u.ldb = ld;       // K x N image: rows kb*128.., all columns < kb*128
      u.C = Z; u.ldc = ld;
      u.K = NB; u.r0 = 0; u.r1 = RIDE / 64; u.c0 = 0; u.c1 = kb * 2; u.lower = 0;
      rc = launch_gemm_cfg<Real, 64, 64, 2, 2, GEMM_SUB, true>(h, st, u);
      if (rc) return rc;
    }
  }
  return SIGP_OK;
}

static int solve_rows_forward(sigp_handle* h, Slot& s, double* Z, long n_pad, int nchunks = 1) {
  return solve_rows_forward_t<double>(h, s.s_upd, s.mat, s.dinv, Z, n_pad, nchunks);
}
static int solve_rows_backward(sigp_handle* h, Slot& s, double* Z, long n_pad) {
  return solve_rows_backward_t<double>(h, s.s_upd, s.mat, s.dinv, Z, n_pad);
}

// U = L^-T (upper triangular, row-major, leading dimension ld) by recursive doubling from the inverse diagonal blocks
// the factorisation left behind:  [L11 0; L21 L22]^-T = [U11  -U11 L21^T U22 ; 0  U22].  Level s joins every pair of
// finished s-blocks (the last pair may be ragged) with two products -- P = U11 L21^T (LDS-DMA kernel, k starts at the
// row's diagonal block) and U12 = -P U22 (k stops at the column's diagonal block) -- all pairs of a level in one launch
// (grid.y = pair).  n^3/3 flops, every product MFMA work with K >= 128.  Levels stop below `span` (a power of two, in
// 128-blocks): span >= T inverts the whole factor, span = S leaves the inverses of the aligned S-block diagonal blocks
// (what the block triangular solves of the fp32 refinement use).  P is scratch of the same shape as U.
// Blocks of U below its block diagonal are never written NOR read (the products skip them through GemmArgsT::ktri).
// nz > 1: nz lockstep members in every launch (grid.z; member strides zL, zD, zU of Lm, dinvp and of U / P).
template <typename Real>
int trtri_levels(sigp_handle* h, hipStream_t st, const Real* Lm, long ldl, const Real* dinvp, Real* U, Real* P, long ld, int T, int span,
                 int nz = 1, long zL = 0, long zD = 0, long zU = 0) {
  // Lm has its own leading dimension ldl (a rank's block columns of a sharded factor: the diagonal block of one of its panels
  // sits in storage whose row stride is its column count); U and P share ld
  int rc;
  hipLaunchKernelGGL(transpose_blocks_kernel<Real>, dim3(T, nz), dim3(256), 0, st, dinvp, U, ld, zD, zU);
  HIPCHK(h, hipGetLastError());
  for (int sblk = 1; sblk < T && sblk < span; sblk *= 2) {
    const int npairs_full = T / (2 * sblk);                           // pairs whose right block is a whole s-block
    const int tail_left = npairs_full * 2 * sblk;                     // a ragged pair: left [tail_left, +s), right the rest
    const int tail_r = T - tail_left - sblk;                          // > 0 when it exists
    for (int pass = 0; pass < 2; ++pass) {
      const int nb2 = pass == 0 ? npairs_full : (tail_r > 0 ? 1 : 0);
      if (nb2 <= 0) continue;
      const int b0 = pass == 0 ? 0 : tail_left;
      const int rs = pass == 0 ? sblk : tail_r;                       // blocks in the right part
      const long o = (long)b0 * NB * (ld + 1), pairStride = (long)2 * sblk * NB * (ld + 1);
      const long oL = (long)b0 * NB * (ldl + 1), pairStrideL = (long)2 * sblk * NB * (ldl + 1);
      GemmArgsT<Real> g1{};                                           // P = U11 L21^T
      g1.A = U + o; g1.lda = ld; g1.sA = pairStride;
      g1.B = Lm + oL + (long)sblk * NB * ldl; g1.ldb = ldl; g1.sB = pairStrideL;
      g1.C = P + o + (long)sblk * NB; g1.ldc = ld; g1.sC = pairStride;
      g1.batch = nb2; g1.K = sblk * NB; g1.r0 = 0; g1.r1 = sblk; g1.c0 = 0; g1.c1 = rs; g1.lower = 0; g1.ktri = 1;
      g1.zcount = nz; g1.zA = zU; g1.zB = zL; g1.zC = zU;
      if ((rc = launch_syrk128_t<Real, true>(h, st, g1))) return rc;
      GemmArgsT<Real> g2{};                                           // U12 = -P U22
      g2.A = P + o + (long)sblk * NB; g2.lda = ld; g2.sA = pairStride;
      g2.B = U + o + (long)sblk * NB * (ld + 1); g2.ldb = ld; g2.sB = pairStride;
      g2.C = U + o + (long)sblk * NB; g2.ldc = ld; g2.sC = pairStride;
      g2.batch = nb2; g2.K = rs * NB; g2.r0 = 0; g2.r1 = sblk; g2.c0 = 0; g2.c1 = rs; g2.lower = 0; g2.ktri = 2;
      g2.zcount = nz; g2.zA = zU; g2.zB = zU; g2.zC = zU;
      if ((rc = launch_gemm_cfg<Real, 128, 128, 2, 2, GEMM_SETNEG, true>(h, st, g2))) return rc;
    }
  }
  return SIGP_OK;
}

#include "sigp_f32.inc"   // fp32 engine: build, factor, block solves, fp64 refinement

}  // namespace

// =====================================================================================================
extern "C" {

int sigp_version(void) { return 500; }   // 4.0: sigp_transport grew scatter / allgather (3.x callers: sigp_dist_init_transport2 with their struct's size); 5.0: sigp_small_run_grad, sigp_small_set_dweights, sigp_dist_init_transport2

// which HIP runtime serves this process (a process that also loads PyTorch-ROCm has two on disk; the first one mapped wins)
int sigp_runtime_info(char* buf, int64_t len) {
  if (!buf || len < 1) return SIGP_BAD_ARG;
  int ver = 0;
  const hipError_t e = hipRuntimeGetVersion(&ver);
  Dl_info di{};
  const char* path = (dladdr((void*)&hipRuntimeGetVersion, &di) && di.dli_fname) ? di.dli_fname : "?";
  snprintf(buf, (size_t)len, "HIP runtime %d (%s) from %s", ver, e == hipSuccess ? "ok" : hipGetErrorString(e), path);
  return SIGP_OK;
}

int sigp_create(sigp_handle** out, int device_id, int dtype) {
  if (!out) return SIGP_BAD_ARG;
  *out = nullptr;
  if (dtype != SIGP_F64 && dtype != SIGP_F32) return SIGP_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SIGP_HIP_ERROR;
  if (hipSetDevice(device_id) != hipSuccess) return SIGP_HIP_ERROR;
  sigp_handle* h = new sigp_handle();
  h->device = device_id;
  h->dtype = dtype;
  if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || h->ncu < 1) h->ncu = 256;
  int rc = slot_init(h, h->slots[0]);
  if (rc) { delete h; return rc; }
  h->nslots = 1;
  *out = h;
  return SIGP_OK;
}

int sigp_destroy(sigp_handle* h) {
  if (!h) return SIGP_BAD_ARG;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  prof_drain(h);
  for (auto& s : h->slots) slot_free(s);
  double* bufs[] = {h->X, h->y, h->Xs, h->scratchZ, h->T, h->Sig, h->XsA, h->stage, h->bX, h->by, h->bXs, h->gU, h->gK, h->gD, h->gPart, h->gSig, h->gT, h->xq, h->rq, h->rpart, h->fpart, h->sm_A, h->sm_y, h->sm_lam, h->sm_dlam, h->sm_out};
  dist_release(h);
  if (h->sm_sets_dev) (void)hipFree(h->sm_sets_dev);
  if (h->sm_probs) (void)hipFree(h->sm_probs);
  if (h->pred_kps) (void)hipFree(h->pred_kps);
  if (h->gKps) (void)hipFree(h->gKps);
  if (h->kq) (void)hipFree(h->kq);
  if (h->fmat) (void)hipFree(h->fmat);
  if (h->fdinv) (void)hipFree(h->fdinv);
  if (h->fZ) (void)hipFree(h->fZ);
  for (float* p : {h->fU, h->fV, h->fXw}) if (p) (void)hipFree(p);
  for (double* p : bufs) if (p) (void)hipFree(p);
  delete h;
  return SIGP_OK;
}

const char* sigp_last_error(const sigp_handle* h) { return h ? h->err.c_str() : "null handle"; }

int sigp_set_option(sigp_handle* h, const char* name, int64_t value) {
  if (!h || !name) return SIGP_BAD_ARG;
  {   // measurement switches and rejected experiments: their code is compiled into libsigp_debug.so only
    static const char* const dbg_only[] = {"xcd_chunks", "update_wgs", "update_late", "pipeline_head", "head_gate", "wide_tiles", "n64_tiles", "patch",
                                           "small_nt64", "reserve_cus", "panel_ll", "c_dma", "syrk_v2", "update_dbg"};
    if (!DBG_MASK)
      for (const char* nm : dbg_only)
        if (!strcmp(name, nm)) return fail(h, SIGP_BAD_ARG, "%s is a measurement switch of libsigp_debug.so (make debug), not of the product library", nm);
  }
  if (!strcmp(name, "outer_blocks")) { if (value < 1 || value > 64) return SIGP_BAD_ARG; h->opt_outer = (int)value; h->outer_set = true; return SIGP_OK; }
  if (!strcmp(name, "lookahead")) { h->opt_lookahead = value ? 1 : 0; return SIGP_OK; }
  if (!strcmp(name, "pan_priority")) {
    if ((int)(value != 0) == h->opt_pan_priority) return SIGP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    h->opt_pan_priority = value ? 1 : 0;
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (auto& s : h->slots) {
      if (!s.s_pan) continue;
      HIPCHK(h, hipStreamDestroy(s.s_pan));
      s.s_pan = nullptr;
      HIPCHK(h, hipStreamCreateWithPriority(&s.s_pan, hipStreamNonBlocking, h->opt_pan_priority ? hi : lo));
    }
    return SIGP_OK;
  }
  if (!strcmp(name, "refine_tol_e")) { if (value < 0 || value > 16) return SIGP_BAD_ARG; h->opt_refine_tol_e = (int)value; return SIGP_OK; }
  if (!strcmp(name, "refine_iters")) { if (value < 0 || value > 20) return SIGP_BAD_ARG; h->opt_refine_iters = (int)value; return SIGP_OK; }
  if (!strcmp(name, "panel_mode")) { if (value < 0 || value > 2) return SIGP_BAD_ARG; h->opt_panel_mode = (int)value; return SIGP_OK; }
  if (!strcmp(name, "ride_reps")) { if (value < -1 || value > 64) return SIGP_BAD_ARG; h->opt_ride_reps = (int)value; return SIGP_OK; }
  if (!strcmp(name, "kbuild_mfma")) { if (value < 0 || value > 2) return SIGP_BAD_ARG; h->opt_kbuild_mfma = (int)value; return SIGP_OK; }
  if (!strcmp(name, "diag_prio")) { h->opt_diag_prio = value != 0; return SIGP_OK; }
  if (!strcmp(name, "update_dbg")) { h->opt_update_dbg = (int)value; return SIGP_OK; }
  if (!strcmp(name, "c_dma")) {
    if (!DBG_MASK) return fail(h, SIGP_BAD_ARG, "c_dma is a measurement switch of libsigp_debug.so");
    h->opt_c_dma = value != 0; return SIGP_OK;
  }
  if (!strcmp(name, "diag_tiles")) { h->opt_diag_tiles = value != 0; return SIGP_OK; }
  if (!strcmp(name, "ride_tiles")) { h->opt_ride_tiles = value != 0; return SIGP_OK; }
  if (!strcmp(name, "strip_tri")) { h->opt_strip_tri = value != 0; return SIGP_OK; }
  if (!strcmp(name, "strip_min")) { if (value < 1) return SIGP_BAD_ARG; h->opt_strip_min = (int)value; return SIGP_OK; }
  if (!strcmp(name, "schedule")) { if (value < 0 || value > 1) return SIGP_BAD_ARG; h->opt_schedule = (int)value; return SIGP_OK; }
  if (!strcmp(name, "small_nt64")) { h->opt_small_nt64 = value != 0; return SIGP_OK; }
  if (!strcmp(name, "owner_only")) { h->opt_owner_only = value != 0; return SIGP_OK; }
  if (!strcmp(name, "dist_timeout_ms")) { if (value < 0) return SIGP_BAD_ARG; h->opt_dist_timeout_ms = (long)value; return SIGP_OK; }
  if (!strcmp(name, "dist_stats")) { h->opt_dist_stats = value != 0; return SIGP_OK; }
  if (!strcmp(name, "dist_panel_split")) { if (value < -1 || value > 1) return SIGP_BAD_ARG; h->opt_dist_split = (int)value; return SIGP_OK; }
  if (!strcmp(name, "dist_lookahead2d")) { h->opt_dist_la2d = value != 0; return SIGP_OK; }
  if (!strcmp(name, "dist_segment")) { if (value < 1 || value > 64) return SIGP_BAD_ARG; h->opt_dist_seg = (int)value; return SIGP_OK; }
  if (!strcmp(name, "panel_ll")) { if (value < 0 || value > 64) return SIGP_BAD_ARG; h->opt_panel_ll = (int)value; return SIGP_OK; }
  if (!strcmp(name, "trsm128_threshold")) { if (value < 0) return SIGP_BAD_ARG; h->opt_trsm128 = (int)value; return SIGP_OK; }
  if (!strcmp(name, "syrk_v2")) {   // the generic 128-tile kernel spills 12 B/lane to scratch: never beside a second stream
    if (!DBG_MASK) return fail(h, SIGP_BAD_ARG, "syrk_v2 is a measurement switch of libsigp_debug.so");
    h->opt_syrk_v2 = value ? 1 : 0; return SIGP_OK;
  }
  if (!strcmp(name, "patch")) { if (value < 0 || value > 16) return SIGP_BAD_ARG; h->opt_patch = (int)value; return SIGP_OK; }
  if (!strcmp(name, "xcd_chunks")) { if (value < 0 || value > 16) return SIGP_BAD_ARG; h->opt_xcd_chunks = (int)value; return SIGP_OK; }
  if (!strcmp(name, "n64_tiles")) { if (value < 0 || value > 3) return SIGP_BAD_ARG; h->opt_n64_tiles = (int)value; return SIGP_OK; }
  if (!strcmp(name, "wide_tiles")) { if (value < 0 || value > 3) return SIGP_BAD_ARG; h->opt_wide_tiles = (int)value; return SIGP_OK; }
  if (!strcmp(name, "link_rows")) { if (value < 0) return SIGP_BAD_ARG; h->opt_link_rows = (int)value; return SIGP_OK; }
  if (!strcmp(name, "chain_rows")) { if (value < 0) return SIGP_BAD_ARG; h->opt_chain_rows = (int)value; return SIGP_OK; }
  if (!strcmp(name, "refine_sym")) { if (value < 0 || value > 1) return SIGP_BAD_ARG; h->opt_refine_sym = (int)value; return SIGP_OK; }
  if (!strcmp(name, "refine_stored")) { if (value < 0 || value > 1) return SIGP_BAD_ARG; h->opt_refine_stored = (int)value; return SIGP_OK; }
  if (!strcmp(name, "first_on_panel")) { if (value < 0 || value > 2) return SIGP_BAD_ARG; h->opt_first_on_panel = (int)value; return SIGP_OK; }
  if (!strcmp(name, "panel_chain")) { if (value < 0 || value > 15) return SIGP_BAD_ARG; h->opt_panel_chain = (int)value; return SIGP_OK; }
  if (!strcmp(name, "strips_after_update")) { h->opt_strips_after_update = value != 0; return SIGP_OK; }
  if (!strcmp(name, "pipeline_head")) { if (value < 0 || value > 3) return SIGP_BAD_ARG; h->opt_pipeline_head = (int)value; return SIGP_OK; }
  if (!strcmp(name, "head_gate")) { if (value < 0) return SIGP_BAD_ARG; h->opt_head_gate = (int)value; return SIGP_OK; }
  if (!strcmp(name, "update_late")) { if (value < 0 || value > 4096) return SIGP_BAD_ARG; h->opt_update_late = (int)value; return SIGP_OK; }
  if (!strcmp(name, "update_wgs")) { if (value < 0 || value > 4096) return SIGP_BAD_ARG; h->opt_update_wgs = (int)value; return SIGP_OK; }
  if (!strcmp(name, "group")) { if (value < 1 || value > 256) return SIGP_BAD_ARG; h->opt_group = (int)value; return SIGP_OK; }
  if (!strcmp(name, "host_timing")) { h->opt_host_timing = (int)value; return SIGP_OK; }
  if (!strcmp(name, "reserve_cus")) {
    if (value < 0 || value > 64) return SIGP_BAD_ARG;
    if ((int)value == h->opt_reserve_cus) return SIGP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    h->opt_reserve_cus = (int)value;
    for (auto& s : h->slots) {
      if (!s.s_upd) continue;
      HIPCHK(h, hipStreamDestroy(s.s_upd));
      s.s_upd = nullptr;
      int rc = make_update_stream(h, &s.s_upd);
      if (rc) return rc;
    }
    return SIGP_OK;
  }
  if (!strcmp(name, "tiny_tile_threshold")) { if (value < 0) return SIGP_BAD_ARG; h->opt_tiny_tiles = (int)value; return SIGP_OK; }
  if (!strcmp(name, "small_tile_threshold")) { if (value < 0) return SIGP_BAD_ARG; h->opt_small_tiles = (int)value; return SIGP_OK; }
  return fail(h, SIGP_BAD_ARG, "unknown option %s", name);
}

int sigp_set_train(sigp_handle* h, const double* X, int64_t n, int64_t d, int64_t ldx, const double* y) {
  if (!h || !X || !y || n < 1 || d < 1 || ldx < d) return fail(h, SIGP_BAD_ARG, "set_train: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const long n_pad = round_up(n, NB), dp = round_up(d, 64);
  int rc;
  if ((rc = ensure(h, &h->X, &h->cap_X, n_pad * dp))) return rc;
  if ((rc = ensure(h, &h->y, &h->cap_y, n_pad))) return rc;
  if ((rc = ensure(h, &h->Xs, &h->cap_Xs, (long)RIDE * dp))) return rc;
  if ((rc = ensure(h, &h->stage, &h->cap_stage, std::max<long>(n * ldx, n_pad)))) return rc;
  if ((rc = slot_reserve(h, h->slots[0], (h->dtype == SIGP_F64 && !h->opt_owner_only) ? n_pad : (long)NB, 1))) return rc;
  h->dl.on = false;
  hipStream_t st = h->slots[0].s_upd;
  HIPCHK(h, hipMemcpyAsync(h->stage, X, (size_t)((n - 1) * ldx + d) * sizeof(double), hipMemcpyHostToDevice, st));
  {
    const long tot = n_pad * dp;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldx, (int)n, (int)d,
                       h->X, (int)n_pad, (int)dp);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(st));   // stage is reused below
  HIPCHK(h, hipMemsetAsync(h->y, 0, (size_t)n_pad * sizeof(double), st));
  HIPCHK(h, hipMemcpyAsync(h->y, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(h->Xs, 0, (size_t)RIDE * dp * sizeof(double), st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->n = n; h->d = d; h->dp = dp; h->n_pad = n_pad; h->m = 0;
  h->kss_unit.clear();
  h->built = h->factored = h->fitted = false;
  return SIGP_OK;
}

int sigp_set_test(sigp_handle* h, const double* Xs, int64_t m, int64_t ldxs) {
  if (!h || h->n == 0 || m < 0 || m > SIGP_MAX_RIDE || (m > 0 && (!Xs || ldxs < h->d))) return fail(h, SIGP_BAD_ARG, "set_test: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->slots[0].s_upd;
  HIPCHK(h, hipMemsetAsync(h->Xs, 0, (size_t)RIDE * h->dp * sizeof(double), st));
  if (m > 0) {
    int rc;
    if ((rc = ensure(h, &h->stage, &h->cap_stage, m * ldxs))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage, Xs, (size_t)((m - 1) * ldxs + h->d) * sizeof(double), hipMemcpyHostToDevice, st));
    const long tot = m * h->dp;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldxs, (int)m, (int)h->d,
                       h->Xs, (int)m, (int)h->dp);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(st));
  h->m = m;
  h->kss_unit.assign((size_t)m, 1.0);   // rbf / matern: k~(x,x) = 1; the reference kernel overwrites this at build
  h->built = h->factored = h->fitted = false;
  return SIGP_OK;
}

int sigp_kernel_build(sigp_handle* h, int kernel_id, double ell, double sn_tilde) {
  if (!h || h->n == 0) return fail(h, SIGP_BAD_ARG, "kernel_build: call set_train first");
  if (kernel_id != SIGP_KERNEL_RBF && kernel_id != SIGP_KERNEL_MATERN52) return fail(h, SIGP_BAD_ARG, "kernel_build: kernel_id must be RBF or MATERN52 (use kernel_build_from_sigma for the reference kernel)");
  if (!(ell > 0) || !(sn_tilde >= 0)) return fail(h, SIGP_BAD_ARG, "kernel_build: ell > 0 and sn_tilde >= 0 required");
  HIPCHK(h, hipSetDevice(h->device));
  h->kp = make_kparams(kernel_id, ell, sn_tilde, 0);
  h->kernel_id = kernel_id; h->ell = ell; h->sn_tilde = sn_tilde;
  h->kss_unit.assign((size_t)h->m, 1.0);
  if (h->dtype == SIGP_F32) {   // fp32 engine: the matrix goes to its own buffers; only the sharded panel loop (sigp_dist_*) continues from here
    int rc32 = f32_build(h, kernel_id, ell, sn_tilde, h->X, h->y, h->Xs, h->n, h->d, h->dp, h->n_pad, h->m);
    if (rc32) return rc32;
    if ((rc32 = sync_slot(h, h->slots[0]))) return rc32;
    h->built = true; h->factored = h->fitted = false;
    return SIGP_OK;
  }
  int rc = slot_reserve(h, h->slots[0], h->n_pad, 1);
  if (rc) return rc;
  h->slots[0].kps_host[0] = h->kp;
  if ((rc = upload_kparams(h, h->slots[0], 1))) return rc;
  rc = build_cov(h, h->slots[0], 1, h->X, 0, h->y, 0, h->Xs, 0, h->n, h->d, h->dp, h->n_pad, h->m);
  if (rc) return rc;
  rc = sync_slot(h, h->slots[0]);
  if (rc) return rc;
  h->built = true; h->factored = h->fitted = false;
  return SIGP_OK;
}

// Reference kernel, GEMM form (north/June1st.py:264-265): stage Sigma~ and T = X Sigma~ (sigma_prepare), then emit the block
// columns [cb0, cb1) of K~ = T X^T + sn I (lower 64-tiles) and of the ride rows [y ; Xs T^T] into storage given by its (virtual)
// origin Cm with row stride ldc (sigma_emit): the slot's square matrix, or one rank's block columns (sigp_dist_local_build).
static int sigma_prepare(sigp_handle* h, const double* Sigma, int64_t ldsigma) {
  Slot& s = h->slots[0];
  hipStream_t st = s.s_upd;
  const long n = h->n, N = h->d, dp = h->dp, n_pad = h->n_pad;
  int rc;
  if ((rc = ensure(h, &h->Sig, &h->cap_Sig, dp * dp))) return rc;
  if ((rc = ensure(h, &h->T, &h->cap_T, n_pad * dp))) return rc;
  if ((rc = ensure(h, &h->stage, &h->cap_stage, N * ldsigma))) return rc;
  if (!h->XsA) HIPCHK(h, hipMalloc((void**)&h->XsA, (size_t)RIDE * 4096 * sizeof(double)));
  if (dp > 4096) return fail(h, SIGP_BAD_ARG, "reference kernel: more than 4096 features not supported");
  HIPCHK(h, hipMemcpyAsync(h->stage, Sigma, (size_t)((N - 1) * ldsigma + N) * sizeof(double), hipMemcpyHostToDevice, st));
  {
    const long tot = dp * dp;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldsigma, (int)N, (int)N,
                       h->Sig, (int)dp, (int)dp);
    HIPCHK(h, hipGetLastError());
  }
  // T = X Sigma~   (Sigma~ symmetric => X Sigma~^T), 64x64 tiles: n_pad/64 x dp/64
  {
    GemmArgs g{};
    g.A = h->X; g.lda = dp; g.B = h->Sig; g.ldb = dp; g.C = h->T; g.ldc = dp; g.K = (int)dp;
    g.r0 = 0; g.r1 = (int)(n_pad / 64); g.c0 = 0; g.c1 = (int)(dp / 64); g.lower = 0;
    ProfScope ps(h, st, SIGP_KC_KBUILD, 2.0 * n * N * N, 8.0 * (2 * n * N + N * N));
    if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g))) return rc;
  }
  // XsA = Xs shifted down by one row (ride row 0 is y)
  HIPCHK(h, hipMemsetAsync(h->XsA, 0, (size_t)RIDE * dp * sizeof(double), st));
  if (h->m > 0)
    HIPCHK(h, hipMemcpyAsync(h->XsA + dp, h->Xs, (size_t)h->m * dp * sizeof(double), hipMemcpyDeviceToDevice, st));
  s.kps_host[0] = h->kp;
  return upload_kparams(h, s, 1);
}

static int sigma_emit(sigp_handle* h, double* Cm, long ldc, int cb0, int cb1, double sn_tilde) {
  Slot& s = h->slots[0];
  hipStream_t st = s.s_upd;
  const long n = h->n, N = h->d, dp = h->dp, n_pad = h->n_pad;
  int rc;
  {   // K~ lower tiles = T X^T
    GemmArgs g{};
    g.A = h->T; g.lda = dp; g.B = h->X; g.ldb = dp; g.C = Cm; g.ldc = ldc; g.K = (int)dp;
    g.r0 = 0; g.r1 = (int)(n_pad / 64); g.c0 = cb0 * 2; g.c1 = cb1 * 2; g.lower = 1;
    ProfScope ps(h, st, SIGP_KC_KBUILD, (double)n * n * N, 4.0 * n * (n + 1) + 16.0 * n * N);
    if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g))) return rc;
    const int i0 = cb0 * NB, ic = (cb1 - cb0) * NB;
    hipLaunchKernelGGL(diag_fix_kernel, dim3((unsigned)((ic + 255) / 256)), dim3(256), 0, st, Cm, ldc, (int)n, (int)n_pad, sn_tilde, i0, ic);
    HIPCHK(h, hipGetLastError());
  }
  {   // ride rows: row 0 <- y, rows 1..m <- Xs T^T
    GemmArgs g{};
    g.A = h->XsA; g.lda = dp; g.B = h->T; g.ldb = dp; g.C = Cm + n_pad * ldc; g.ldc = ldc; g.K = (int)dp;
    g.r0 = 0; g.r1 = RIDE / 64; g.c0 = cb0 * 2; g.c1 = cb1 * 2; g.lower = 0;
    if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g))) return rc;
    const int i0 = cb0 * NB, ic = (cb1 - cb0) * NB;
    hipLaunchKernelGGL(ride_build_kernel<double>, dim3((unsigned)((ic + 255) / 256), 1, 1), dim3(256), 0, st, h->X, 0L, h->Xs, 0L, h->y, 0L, (int)dp,
                       (int)h->d, (int)n, (int)n_pad, (int)h->m, 1, Cm + n_pad * ldc, 0L, ldc, s.kps, 0, i0, ic);
    HIPCHK(h, hipGetLastError());
  }
  return SIGP_OK;
}

static int build_from_sigma_async(sigp_handle* h, const double* Sigma, int64_t ldsigma, double sn_tilde) {
  int rc = sigma_prepare(h, Sigma, ldsigma);
  if (rc) return rc;
  return sigma_emit(h, h->slots[0].mat, h->n_pad, 0, (int)(h->n_pad / NB), sn_tilde);
}

// k~** = xs Sigma~ xs^T for the ride-along test points, on the host (m x N x N flops)
static int sigma_kss(sigp_handle* h, const double* Sigma, int64_t ldsigma) {
  if (h->m <= 0) return SIGP_OK;
  std::vector<double> xs((size_t)h->m * h->dp);
  HIPCHK(h, hipMemcpyAsync(xs.data(), h->Xs, xs.size() * sizeof(double), hipMemcpyDeviceToHost, h->slots[0].s_upd));
  HIPCHK(h, hipStreamSynchronize(h->slots[0].s_upd));
  h->kss_unit.assign((size_t)h->m, 0.0);
  for (long j = 0; j < h->m; ++j) {
    double acc = 0.0;
    for (long a = 0; a < h->d; ++a) {
      double t = 0.0;
      for (long b = 0; b < h->d; ++b) t += Sigma[a * ldsigma + b] * xs[j * h->dp + b];
      acc += xs[j * h->dp + a] * t;
    }
    h->kss_unit[j] = acc;
  }
  return SIGP_OK;
}

int sigp_kernel_build_from_sigma(sigp_handle* h, const double* Sigma, int64_t ldsigma, double sn_tilde) {
  if (h && h->dtype != SIGP_F64) return fail(h, SIGP_BAD_ARG, "kernel_build_from_sigma: the fp32 engine exposes the fused path only (sigp_fit_predict / batch / predict)");

  if (!h || h->n == 0 || !Sigma || ldsigma < h->d) return fail(h, SIGP_BAD_ARG, "kernel_build_from_sigma: bad argument");
  if (!(sn_tilde >= 0)) return fail(h, SIGP_BAD_ARG, "sn_tilde >= 0 required");
  HIPCHK(h, hipSetDevice(h->device));
  h->kp = make_kparams(SIGP_KERNEL_NETDIFFUSION, 1.0, sn_tilde, 0);
  h->kernel_id = SIGP_KERNEL_NETDIFFUSION; h->ell = 0; h->sn_tilde = sn_tilde;
  int rc = build_from_sigma_async(h, Sigma, ldsigma, sn_tilde);
  if (rc) return rc;
  if ((rc = sigma_kss(h, Sigma, ldsigma))) return rc;
  rc = sync_slot(h, h->slots[0]);
  if (rc) return rc;
  h->built = true; h->factored = h->fitted = false;
  return SIGP_OK;
}

int sigp_potrf(sigp_handle* h, int64_t* info) {
  if (h && h->dtype != SIGP_F64) return fail(h, SIGP_BAD_ARG, "potrf: the fp32 engine exposes the fused path only (sigp_fit_predict / batch / predict)");

  if (!h || !h->built) return fail(h, SIGP_BAD_ARG, "potrf: build the kernel matrix first");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[0];
  int rc = potrf_slot(h, s, 1, h->n_pad, false, 1 + (int)h->m);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(s.info_host, s.info, sizeof(int), hipMemcpyDeviceToHost, s.s_upd));
  rc = sync_slot(h, s);
  if (rc) return rc;
  h->built = false;
  if (info) *info = *s.info_host;
  if (*s.info_host != 0) { h->factored = false; return fail(h, SIGP_NOT_SPD, "potrf: matrix is not positive definite (pivot %d)", *s.info_host); }
  h->factored = true; h->fitted = false;
  return SIGP_OK;
}

int sigp_fit(sigp_handle* h, double* sigma_f, double* nlml) {
  if (h && h->dtype != SIGP_F64) return fail(h, SIGP_BAD_ARG, "fit: the fp32 engine exposes the fused path only (sigp_fit_predict / batch / predict)");

  if (!h || !h->factored) return fail(h, SIGP_BAD_ARG, "fit: call potrf first");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[0];
  int rc = epilogue_slot(h, s, 1, h->n, h->n_pad, h->m);
  if (rc) return rc;
  rc = sync_slot(h, s);
  if (rc) return rc;
  double out[4];
  h->fit_res.assign(s.res_host, s.res_host + 512);
  finish_results(s.res_host, 0, h->n, 0, h->sn_tilde, nullptr, out, nullptr, nullptr);
  h->sigma_f = out[0]; h->nlml = out[1]; h->fitted = true;
  if (sigma_f) *sigma_f = out[0];
  if (nlml) *nlml = out[1];
  return SIGP_OK;
}

int sigp_predict_ride(sigp_handle* h, double* mean, double* var) {
  if (!h || !h->fitted) return fail(h, SIGP_BAD_ARG, "predict_ride: call fit first");
  double out[4];
  finish_results(h->fit_res.data(), 0, h->n, h->m, h->sn_tilde, h->kss_unit.data(), out, mean, var);
  return SIGP_OK;
}

int sigp_fit_predict(sigp_handle* h, int kernel_id, double ell, double sn_tilde, const double* Sigma, int64_t ldsigma,
                     double* out, double* mean, double* var) {
  if (!h || h->n == 0 || !out) return fail(h, SIGP_BAD_ARG, "fit_predict: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[0];
  int rc;
  if (h->dtype == SIGP_F32) {
    if (kernel_id != SIGP_KERNEL_RBF && kernel_id != SIGP_KERNEL_MATERN52) return fail(h, SIGP_BAD_ARG, "fp32 engine: RBF / MATERN52 only");
    if (!(ell > 0) || !(sn_tilde >= 0)) return fail(h, SIGP_BAD_ARG, "ell > 0 and sn_tilde >= 0 required");
    h->kp = make_kparams(kernel_id, ell, sn_tilde, 0);
    h->kernel_id = kernel_id; h->ell = ell; h->sn_tilde = sn_tilde;
    h->kss_unit.assign((size_t)h->m, 1.0);
    if ((rc = f32_fit(h, kernel_id, ell, sn_tilde, h->X, h->y, h->Xs, h->n, h->d, h->dp, h->n_pad, h->m, out, mean, var))) return rc;
    const int info32 = (int)out[2];
    h->built = false;
    h->factored = h->fitted = (info32 == 0);
    h->sigma_f = out[0]; h->nlml = out[1];
    if (info32 != 0) return fail(h, SIGP_NOT_SPD, "fit_predict: matrix is not positive definite (pivot %d)", info32);
    return SIGP_OK;
  }
  if (kernel_id == SIGP_KERNEL_NETDIFFUSION) {
    if (!Sigma) return fail(h, SIGP_BAD_ARG, "fit_predict: Sigma required for the reference kernel");
    rc = sigp_kernel_build_from_sigma(h, Sigma, ldsigma, sn_tilde);
    if (rc) return rc;
  } else {
    if (kernel_id != SIGP_KERNEL_RBF && kernel_id != SIGP_KERNEL_MATERN52) return fail(h, SIGP_BAD_ARG, "bad kernel_id");
    if (!(ell > 0) || !(sn_tilde >= 0)) return fail(h, SIGP_BAD_ARG, "ell > 0 and sn_tilde >= 0 required");
    h->kp = make_kparams(kernel_id, ell, sn_tilde, 0);
    h->kernel_id = kernel_id; h->ell = ell; h->sn_tilde = sn_tilde;
    h->kss_unit.assign((size_t)h->m, 1.0);
    if ((rc = slot_reserve(h, s, h->n_pad, 1))) return rc;
    s.kps_host[0] = h->kp;
    if ((rc = upload_kparams(h, s, 1))) return rc;
    rc = build_cov(h, s, 1, h->X, 0, h->y, 0, h->Xs, 0, h->n, h->d, h->dp, h->n_pad, h->m);
    if (rc) return rc;
  }
  if ((rc = potrf_slot(h, s, 1, h->n_pad, false, 1 + (int)h->m))) return rc;
  if ((rc = epilogue_slot(h, s, 1, h->n, h->n_pad, h->m))) return rc;
  if ((rc = sync_slot(h, s))) return rc;
  const int info = *s.info_host;
  h->fit_res.assign(s.res_host, s.res_host + 512);
  finish_results(s.res_host, info, h->n, h->m, sn_tilde, h->kss_unit.data(), out, mean, var);
  h->built = false;
  h->factored = h->fitted = (info == 0);
  h->sigma_f = out[0]; h->nlml = out[1];
  if (info != 0) return fail(h, SIGP_NOT_SPD, "fit_predict: matrix is not positive definite (pivot %d)", info);
  return SIGP_OK;
}

// ---- forward / backward block solves on a scratch ride block (predict for new points, alpha) --------
int sigp_get_alpha(sigp_handle* h, double* alpha_tilde) {
  if (!h || !h->factored || !alpha_tilde) return fail(h, SIGP_BAD_ARG, "get_alpha: call potrf first");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->dtype == SIGP_F32) {   // the refined solution (row 0 of xq)
    HIPCHK(h, hipMemcpy(alpha_tilde, h->xq, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost));
    return SIGP_OK;
  }
  Slot& s = h->slots[0];
  const long n_pad = h->n_pad, ld = n_pad;
  int rc;
  if ((rc = ensure(h, &h->scratchZ, &h->cap_Z, (long)RIDE * n_pad))) return rc;
  HIPCHK(h, hipMemsetAsync(h->scratchZ, 0, (size_t)RIDE * n_pad * sizeof(double), s.s_upd));
  HIPCHK(h, hipMemcpyAsync(h->scratchZ, s.mat + n_pad * ld, (size_t)n_pad * sizeof(double), hipMemcpyDeviceToDevice, s.s_upd));   // row 0 = z
  if ((rc = solve_rows_backward(h, s, h->scratchZ, n_pad))) return rc;
  HIPCHK(h, hipMemcpyAsync(alpha_tilde, h->scratchZ, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost, s.s_upd));
  return sync_slot(h, s);
}

int sigp_predict(sigp_handle* h, const double* Xs, int64_t m, int64_t ldxs, double* mean, double* var) {
  if (!h || !h->fitted || !Xs || m < 1 || ldxs < h->d || !mean || !var) return fail(h, SIGP_BAD_ARG, "predict: bad argument or fit() not called");
  if (h->kernel_id == SIGP_KERNEL_NETDIFFUSION && !h->T) return fail(h, SIGP_BAD_ARG, "predict: no Sigma state");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->dtype == SIGP_F32) {
    // mean = k*^T alpha~ in fp64 against the refined alpha~; variance from the fp32 factor (forward block solve)
    Slot& s = h->slots[0];
    hipStream_t st = s.s_upd;
    const long n = h->n, n_pad = h->n_pad, ld = n_pad, dp = h->dp;
    const long xs_off = RIDE * std::max<long>(ldxs, dp);
    int rc;
    if ((rc = ensure(h, &h->stage, &h->cap_stage, xs_off + RIDE * dp))) return rc;
    double* xs_dev = h->stage + xs_off;
    for (long c0 = 0; c0 < m; c0 += RIDE) {
      const long mc = std::min<long>(RIDE, m - c0);
      HIPCHK(h, hipMemcpyAsync(h->stage, Xs + c0 * ldxs, (size_t)((mc - 1) * ldxs + h->d) * sizeof(double), hipMemcpyHostToDevice, st));
      const long tot = RIDE * dp;
      hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldxs, (int)mc, (int)h->d, xs_dev, RIDE, (int)dp);
      HIPCHK(h, hipGetLastError());
      s.kps_host[0] = h->kp;
      if ((rc = upload_kparams(h, s, 1))) return rc;
      dim3 g2((unsigned)((n_pad + 255) / 256), RIDE, 1);
      hipLaunchKernelGGL(ride_build_kernel<float>, g2, dim3(256), 0, st, h->X, 0L, xs_dev, 0L, (const double*)nullptr, 0L, (int)dp, (int)h->d, (int)n,
                         (int)n_pad, (int)mc, 0, h->fZ, 0L, ld, s.kps, 1);
      HIPCHK(h, hipGetLastError());
      if ((rc = solve_rows_forward_t<float>(h, st, h->fmat, h->fdinv, h->fZ, n_pad, 1))) return rc;
      hipLaunchKernelGGL(epilogue_kernel<float>, dim3((unsigned)(mc + 1), 1), dim3(256), 0, st, (const float*)h->fZ, ld, (const float*)h->fZ, (const float*)nullptr, ld,
                         (int)n, (int)n_pad, (int)mc, s.res, 0L, 0L, 0L);
      HIPCHK(h, hipGetLastError());
      hipLaunchKernelGGL(cross_mean_kernel, dim3((unsigned)mc), dim3(256), 0, st, h->X, xs_dev, (int)dp, (int)h->d, (int)n, (const double*)h->xq, s.res + 256 + 8, h->kp);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(s.res_host, s.res, 512 * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      for (long j = 0; j < mc; ++j) {
        mean[c0 + j] = s.res_host[256 + 8 + j];
        var[c0 + j] = h->sigma_f * (1.0 + h->sn_tilde - s.res_host[128 + j]);
      }
    }
    return SIGP_OK;
  }
  Slot& s = h->slots[0];
  const long n = h->n, n_pad = h->n_pad, ld = n_pad, dp = h->dp;
  int rc;
  if ((rc = ensure(h, &h->scratchZ, &h->cap_Z, (long)RIDE * n_pad))) return rc;
  const long xs_off = RIDE * std::max<long>(ldxs, dp);   // stage = [raw chunk | padded chunk [128][dp]]
  if ((rc = ensure(h, &h->stage, &h->cap_stage, xs_off + RIDE * dp))) return rc;
  double* xs_dev = h->stage + xs_off;
  hipStream_t st = s.s_upd;
  const double* z = s.mat + n_pad * ld;   // solved row 0 of the ride block: z = L~^-1 y
  if (h->kernel_id != SIGP_KERNEL_NETDIFFUSION && m > RIDE) {
    // Many test points: groups of up to PRED_CHUNKS 128-row chunks advance through the forward solve in lockstep (one launch
    // per block column for the whole group instead of one per chunk: the solve becomes MFMA work instead of a latency chain of
    // 2 T launches per 128 points), one host synchronisation per group.
    constexpr int PRED_CHUNKS = 16;
    const long gmax = std::min<long>(PRED_CHUNKS, (m + RIDE - 1) / RIDE);
    if ((rc = ensure(h, &h->scratchZ, &h->cap_Z, gmax * RIDE * n_pad))) return rc;
    const long raw = gmax * RIDE * std::max<long>(ldxs, dp);          // stage = [raw group | padded group [gmax*128][dp] | results [gmax][512]]
    if ((rc = ensure(h, &h->stage, &h->cap_stage, raw + gmax * RIDE * dp + gmax * 512))) return rc;
    double* xs_grp = h->stage + raw;
    double* res_grp = xs_grp + gmax * RIDE * dp;
    if (!h->pred_kps) HIPCHK(h, hipMalloc((void**)&h->pred_kps, PRED_CHUNKS * sizeof(KParams)));
    std::vector<KParams> kpc((size_t)gmax, h->kp);
    for (long c = 0; c < gmax; ++c) kpc[(size_t)c].ds = (int)c;       // "data set" c = chunk c of the padded group (strideXs below)
    HIPCHK(h, hipMemcpyAsync(h->pred_kps, kpc.data(), (size_t)gmax * sizeof(KParams), hipMemcpyHostToDevice, st));
    std::vector<double> res_host((size_t)gmax * 512);
    for (long c0 = 0; c0 < m; c0 += gmax * RIDE) {
      const long mg = std::min<long>(gmax * RIDE, m - c0);
      const int nch = (int)((mg + RIDE - 1) / RIDE);
      HIPCHK(h, hipMemcpyAsync(h->stage, Xs + c0 * ldxs, (size_t)((mg - 1) * ldxs + h->d) * sizeof(double), hipMemcpyHostToDevice, st));
      const long tot = (long)nch * RIDE * dp;
      hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldxs, (int)mg, (int)h->d, xs_grp, nch * RIDE, (int)dp);
      HIPCHK(h, hipGetLastError());
      hipLaunchKernelGGL(ride_build_kernel<double>, dim3((unsigned)((n_pad + 255) / 256), RIDE, (unsigned)nch), dim3(256), 0, st, h->X, 0L, xs_grp, (long)RIDE * dp,
                         (const double*)nullptr, 0L, (int)dp, (int)h->d, (int)n, (int)n_pad, (int)RIDE, 0, h->scratchZ, (long)RIDE * ld, ld, h->pred_kps, 1);
      HIPCHK(h, hipGetLastError());
      if ((rc = solve_rows_forward(h, s, h->scratchZ, n_pad, nch))) return rc;
      hipLaunchKernelGGL(epilogue_kernel<double>, dim3((unsigned)(RIDE + 1), (unsigned)nch), dim3(256), 0, st, h->scratchZ, ld, z, (const double*)nullptr, ld, (int)n,
                         (int)n_pad, (int)RIDE, res_grp, (long)RIDE * ld, 0L, 0L);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(res_host.data(), res_grp, (size_t)nch * 512 * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      for (long j = 0; j < mg; ++j) {
        const double* rj = res_host.data() + (j / RIDE) * 512;
        mean[c0 + j] = rj[j % RIDE];
        var[c0 + j] = h->sigma_f * (1.0 + h->sn_tilde - rj[128 + j % RIDE]);
      }
    }
    return SIGP_OK;
  }
  std::vector<double> xs_host((size_t)RIDE * dp);
  for (long c0 = 0; c0 < m; c0 += RIDE) {
    const long mc = std::min<long>(RIDE, m - c0);
    HIPCHK(h, hipMemcpyAsync(h->stage, Xs + c0 * ldxs, (size_t)((mc - 1) * ldxs + h->d) * sizeof(double), hipMemcpyHostToDevice, st));
    {
      const long tot = RIDE * dp;
      hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)ldxs, (int)mc, (int)h->d,
                         xs_dev, RIDE, (int)dp);
      HIPCHK(h, hipGetLastError());
    }
    std::vector<double> kss((size_t)mc, 1.0);
    if (h->kernel_id == SIGP_KERNEL_NETDIFFUSION) {
      GemmArgs g{};
      g.A = xs_dev; g.lda = dp; g.B = h->T; g.ldb = dp; g.C = h->scratchZ; g.ldc = ld; g.K = (int)dp;
      g.r0 = 0; g.r1 = RIDE / 64; g.c0 = 0; g.c1 = (int)(n_pad / 64); g.lower = 0;
      if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g))) return rc;
      // k~** = xs Sigma~ xs^T : T_s = xs Sigma~ via the same GEMM on a 128 x dp block, then row dots on the host
      GemmArgs g2{};
      double* ts = h->XsA;   // reuse [128][dp] workspace
      g2.A = xs_dev; g2.lda = dp; g2.B = h->Sig; g2.ldb = dp; g2.C = ts; g2.ldc = dp; g2.K = (int)dp;
      g2.r0 = 0; g2.r1 = RIDE / 64; g2.c0 = 0; g2.c1 = (int)(dp / 64); g2.lower = 0;
      if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g2))) return rc;
      std::vector<double> ts_host((size_t)RIDE * dp);
      HIPCHK(h, hipMemcpyAsync(ts_host.data(), ts, ts_host.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(xs_host.data(), xs_dev, xs_host.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      for (long j = 0; j < mc; ++j) {
        double acc = 0.0;
        for (long a = 0; a < h->d; ++a) acc += ts_host[j * dp + a] * xs_host[j * dp + a];
        kss[j] = acc;
      }
    } else {
      dim3 g2((unsigned)((n_pad + 255) / 256), RIDE);
      s.kps_host[0] = h->kp;
      if ((rc = upload_kparams(h, s, 1))) return rc;
      hipLaunchKernelGGL(ride_build_kernel<double>, g2, dim3(256), 0, st, h->X, 0L, xs_dev, 0L, (const double*)nullptr, 0L, (int)dp, (int)h->d, (int)n,
                         (int)n_pad, (int)mc, 0, h->scratchZ, 0L, ld, s.kps, 1);
      HIPCHK(h, hipGetLastError());
    }
    if ((rc = solve_rows_forward(h, s, h->scratchZ, n_pad))) return rc;
    hipLaunchKernelGGL(epilogue_kernel<double>, dim3((unsigned)(mc + 1), 1), dim3(256), 0, st, h->scratchZ, ld, z, (const double*)nullptr, ld, (int)n,
                       (int)n_pad, (int)mc, s.res, 0L, 0L, 0L);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(s.res_host, s.res, 512 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    for (long j = 0; j < mc; ++j) {
      mean[c0 + j] = s.res_host[j];
      var[c0 + j] = h->sigma_f * (kss[j] + h->sn_tilde - s.res_host[128 + j]);
    }
  }
  return SIGP_OK;
}

int sigp_get_matrix(sigp_handle* h, int which, double* out, int64_t ldo) {
  if (!h || !out || ldo < h->n || h->n == 0) return fail(h, SIGP_BAD_ARG, "get_matrix: bad argument");
  if (which == SIGP_MAT_K && !h->built) return fail(h, SIGP_BAD_ARG, "get_matrix: K~ is not resident (build it, and fetch before potrf)");
  if (which == SIGP_MAT_L && !h->factored) return fail(h, SIGP_BAD_ARG, "get_matrix: L~ is not resident");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[0];
  if (h->dtype == SIGP_F32) {
    std::vector<float> tmp((size_t)h->n * h->n);
    HIPCHK(h, hipMemcpy2D(tmp.data(), (size_t)h->n * sizeof(float), h->fmat, (size_t)h->n_pad * sizeof(float), (size_t)h->n * sizeof(float),
                          (size_t)h->n, hipMemcpyDeviceToHost));
    for (long i = 0; i < h->n; ++i)
      for (long j = 0; j < h->n; ++j) out[i * ldo + j] = (j <= i) ? (double)tmp[(size_t)i * h->n + j] : 0.0;
    return SIGP_OK;
  }
  HIPCHK(h, hipMemcpy2D(out, (size_t)ldo * sizeof(double), s.mat, (size_t)h->n_pad * sizeof(double), (size_t)h->n * sizeof(double),
                        (size_t)h->n, hipMemcpyDeviceToHost));
  for (long i = 0; i < h->n; ++i)
    for (long j = i + 1; j < h->n; ++j) out[i * ldo + j] = 0.0;
  return SIGP_OK;
}

// ---- batch ---------------------------------------------------------------------------------------------
int sigp_batch_upload(sigp_handle* h, int64_t batch, const double* X, int64_t strideX, const double* y, int64_t stridey,
                      const double* Xs, int64_t strideXs, int64_t n, int64_t d, int64_t m) {
  if (!h || batch < 1 || !X || !y || n < 1 || d < 1 || m < 0 || m > SIGP_MAX_RIDE || (m > 0 && !Xs)) return fail(h, SIGP_BAD_ARG, "batch_upload: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  const long n_pad = round_up(n, NB), dp = round_up(d, 64);
  for (double** p : {&h->bX, &h->by, &h->bXs}) if (*p) { HIPCHK(h, hipFree(*p)); *p = nullptr; }
  HIPCHK(h, hipMalloc((void**)&h->bX, (size_t)batch * n_pad * dp * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->by, (size_t)batch * n_pad * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->bXs, (size_t)batch * RIDE * dp * sizeof(double)));
  HIPCHK(h, hipMemset(h->by, 0, (size_t)batch * n_pad * sizeof(double)));
  HIPCHK(h, hipMemset(h->bXs, 0, (size_t)batch * RIDE * dp * sizeof(double)));
  HIPCHK(h, hipDeviceSynchronize());   // the slot streams are non-blocking: order them after the null-stream memsets
  int rc;
  if ((rc = ensure(h, &h->stage, &h->cap_stage, std::max<long>(n * d, RIDE * d)))) return rc;
  hipStream_t st = h->slots[0].s_upd;
  for (long b = 0; b < batch; ++b) {
    const double* Xb = X + b * strideX;
    HIPCHK(h, hipMemcpyAsync(h->stage, Xb, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, st));
    long tot = n_pad * dp;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)d, (int)n, (int)d,
                       h->bX + b * n_pad * dp, (int)n_pad, (int)dp);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipMemcpyAsync(h->by + b * n_pad, y + b * stridey, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    if (m > 0) {
      HIPCHK(h, hipMemcpyAsync(h->stage, Xs + b * strideXs, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, st));
      tot = m * dp;
      hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)d, (int)m, (int)d,
                         h->bXs + b * RIDE * dp, (int)m, (int)dp);
      HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipStreamSynchronize(st));
  }
  h->b_count = batch; h->b_n = n; h->b_d = d; h->b_dp = dp; h->b_m = m; h->b_npad = n_pad;
  return SIGP_OK;
}

int sigp_batch_reserve(sigp_handle* h, int64_t group, int concurrency) {
  if (!h || h->b_count == 0 || group < 1 || group > 256 || concurrency < 1 || concurrency > MAX_SLOTS) return fail(h, SIGP_BAD_ARG, "batch_reserve: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->dtype == SIGP_F32) return f32_reserve(h, h->b_npad, (int)group);
  for (int k = 0; k < concurrency; ++k) {
    int rc = slot_reserve(h, h->slots[k], h->b_npad, (int)group);
    if (rc) return rc;
    if (h->opt_panel_mode != 0 && (rc = slot_ensure_mt(h, h->slots[k], (int)group))) return rc;
  }
  h->nslots = std::max(h->nslots, concurrency);
  return SIGP_OK;
}

int sigp_batch_run(sigp_handle* h, int64_t first, int64_t count, int kernel_id, const double* ell, const double* sn_tilde,
                   int concurrency, double* out, double* mean, double* var) {
  if (!h || h->b_count == 0 || first < 0 || count < 1 || !ell || !sn_tilde || !out) return fail(h, SIGP_BAD_ARG, "batch_run: bad argument");
  if (kernel_id != SIGP_KERNEL_RBF && kernel_id != SIGP_KERNEL_MATERN52) return fail(h, SIGP_BAD_ARG, "batch_run: RBF / MATERN52 only");
  if (concurrency < 1 || concurrency > MAX_SLOTS) return fail(h, SIGP_BAD_ARG, "batch_run: concurrency must be 1..16");
  HIPCHK(h, hipSetDevice(h->device));
  const long n = h->b_n, d = h->b_d, dp = h->b_dp, m = h->b_m, n_pad = h->b_npad;
  if (h->dtype == SIGP_F32) {   // fp32 engine: lockstep groups for the factorisation, refinement member by member; data sets resident in HBM
    for (long i = 0; i < count; ++i)
      if (!(ell[i] > 0) || !(sn_tilde[i] >= 0)) return fail(h, SIGP_BAD_ARG, "batch_run: ell > 0 and sn_tilde >= 0 required");
    const int G32 = (int)std::max<long>(1, std::min<long>(h->opt_group, count));
    std::vector<long> dsv((size_t)G32);
    for (long g0 = 0; g0 < count; g0 += G32) {
      const int nb = (int)std::min<long>(G32, count - g0);
      for (int b = 0; b < nb; ++b) dsv[(size_t)b] = (first + g0 + b) % h->b_count;
      int rc32 = f32_fit_lockstep(h, nb, kernel_id, ell + g0, sn_tilde + g0, dsv.data(), h->bX, h->by, h->bXs, n, d, dp, n_pad, m, out + 4 * g0,
                                  mean ? mean + g0 * m : nullptr, var ? var + g0 * m : nullptr);
      if (rc32) return rc32;
    }
    h->built = h->factored = h->fitted = false;
    return SIGP_OK;
  }
  // fits are factorised in lockstep groups of G (one launch covers the same step of G fits); `concurrency`
  // groups are in flight on separate stream pairs so one group's panel chain overlaps another's updates
  const int G = (int)std::max<long>(1, std::min<long>(h->opt_group, count));
  const long ngroups = (count + G - 1) / G;
  const int nslots = (int)std::min<long>(concurrency, ngroups);
  int rc;
  for (int k = 0; k < nslots; ++k)
    if ((rc = slot_reserve(h, h->slots[k], n_pad, G))) return rc;
  h->nslots = std::max(h->nslots, nslots);
  for (long i = 0; i < count; ++i)
    if (!(ell[i] > 0) || !(sn_tilde[i] >= 0)) return fail(h, SIGP_BAD_ARG, "batch_run: ell > 0 and sn_tilde >= 0 required");
  std::vector<double> kss((size_t)std::max<long>(m, 1), 1.0);
  std::vector<long> inflight((size_t)nslots, -1);   // group index running on each slot
  auto retire = [&](int k) -> int {
    Slot& s = h->slots[k];
    HIPCHK(h, hipStreamSynchronize(s.s_upd));
    const long g0 = inflight[k] * G;
    const int nb = (int)std::min<long>(G, count - g0);
    for (int b = 0; b < nb; ++b) {
      const long i = g0 + b;
      finish_results(s.res_host + 512 * b, s.info_host[b], n, m, sn_tilde[i], kss.data(), out + 4 * i, mean ? mean + i * m : nullptr,
                     var ? var + i * m : nullptr);
    }
    inflight[k] = -1;
    return SIGP_OK;
  };
  double enq_ms = 0, wait_ms = 0;
  for (long g = 0; g < ngroups; ++g) {
    const int k = (int)(g % nslots);
    auto tw0 = std::chrono::steady_clock::now();
    if (inflight[k] >= 0 && (rc = retire(k))) return rc;
    auto tw1 = std::chrono::steady_clock::now();
    wait_ms += std::chrono::duration<double, std::milli>(tw1 - tw0).count();
    Slot& s = h->slots[k];
    const long g0 = g * G;
    const int nb = (int)std::min<long>(G, count - g0);
    for (int b = 0; b < nb; ++b) s.kps_host[b] = make_kparams(kernel_id, ell[g0 + b], sn_tilde[g0 + b], (int)((first + g0 + b) % h->b_count));
    // head pipelining: with >= 2 slots, group g's covariance build and first panel go on its panel stream and run while
    // group g-1 is still in its trailing updates (where the first panel of a group otherwise leaves the update engine idle:
    // ~8 % of a step); its update stream waits for group g-1 to finish, so the bulk of two groups never competes
    const bool head = h->opt_pipeline_head && nslots >= 2 && h->opt_lookahead && h->opt_schedule == 0;
    if (head && g > 0) HIPCHK(h, hipStreamWaitEvent(s.s_upd, h->slots[(g - 1) % nslots].ev_group, 0));
    // pipeline_head = 3: the head does not start as soon as it is enqueued (that is the bulk of group g-1, where the chip is
    // saturated anyway) but when group g-1 enters its tail -- its last panels, where the update stream runs dry
    if (head && h->opt_pipeline_head == 3 && g > 0) HIPCHK(h, hipStreamWaitEvent(s.s_pan, h->slots[(g - 1) % nslots].ev_tail, 0));
    hipStream_t sb = head ? s.s_pan : s.s_upd;
    if ((rc = upload_kparams(h, s, nb, sb))) return rc;
    if ((rc = build_cov(h, s, nb, h->bX, n_pad * dp, h->by, n_pad, h->bXs, (long)RIDE * dp, n, d, dp, n_pad, m, sb))) return rc;
    // pipeline_head = 2: only the covariance build (HBM / VALU work) runs under the previous group; the first panel waits too
    if (head && h->opt_pipeline_head == 2 && g > 0) HIPCHK(h, hipStreamWaitEvent(s.s_pan, h->slots[(g - 1) % nslots].ev_group, 0));
    if ((rc = potrf_slot(h, s, nb, n_pad, head, 1 + (int)m))) return rc;
    if ((rc = epilogue_slot(h, s, nb, n, n_pad, m))) return rc;
    if (head) HIPCHK(h, hipEventRecord(s.ev_group, s.s_upd));
    inflight[k] = g;
    enq_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count();
  }
  for (int k = 0; k < nslots; ++k)
    if (inflight[k] >= 0 && (rc = retire(k))) return rc;
  if (h->opt_host_timing)
    fprintf(stderr, "[sigp] batch_run: %ld fits in %ld groups of %d, host enqueue %.3f ms/fit, host wait-on-retire %.3f ms/fit\n", (long)count,
            ngroups, G, enq_ms / count, wait_ms / count);
  // slot 0 no longer holds the single-fit state
  h->built = h->factored = h->fitted = false;
  return SIGP_OK;
}

int sigp_fit_batch(sigp_handle* h, int64_t batch, int kernel_id, const double* X, int64_t strideX, const double* y, int64_t stridey,
                   const double* Xs, int64_t strideXs, int64_t n, int64_t d, int64_t m, const double* ell, const double* sn_tilde,
                   int concurrency, double* out, double* mean, double* var) {
  // distinct data sets = distinct (X, y, Xs) triples; with all strides 0 the batch is one data set x many grid points
  const int64_t nsets = (strideX == 0 && stridey == 0 && strideXs == 0) ? 1 : batch;
  int rc = sigp_batch_upload(h, nsets, X, strideX, y, stridey, Xs, strideXs, n, d, m);
  if (rc) return rc;
  return sigp_batch_run(h, 0, batch, kernel_id, ell, sn_tilde, concurrency, out, mean, var);
}

// D <- X S X^T for a symmetric S [N][N] on the host (full symmetric result, zero on the padding).  Uses its own
// workspaces (gSig, gT): the fitted state (Sig, T) that sigp_predict reads stays intact.
static int build_xsxt(sigp_handle* h, const double* S, int64_t lds, double* D) {
  hipStream_t st = h->slots[0].s_upd;
  const long N = h->d, dp = h->dp, n_pad = h->n_pad, ld = n_pad;
  int rc;
  if ((rc = ensure(h, &h->gSig, &h->cap_gSig, dp * dp))) return rc;
  if ((rc = ensure(h, &h->gT, &h->cap_gT, n_pad * dp))) return rc;
  if ((rc = ensure(h, &h->stage, &h->cap_stage, N * lds))) return rc;
  HIPCHK(h, hipMemcpyAsync(h->stage, S, (size_t)((N - 1) * lds + N) * sizeof(double), hipMemcpyHostToDevice, st));
  const long tot = dp * dp;
  hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->stage, (long)lds, (int)N, (int)N, h->gSig, (int)dp, (int)dp);
  HIPCHK(h, hipGetLastError());
  GemmArgs g{};
  g.A = h->X; g.lda = dp; g.B = h->gSig; g.ldb = dp; g.C = h->gT; g.ldc = dp; g.K = (int)dp;
  g.r0 = 0; g.r1 = (int)(n_pad / 64); g.c0 = 0; g.c1 = (int)(dp / 64); g.lower = 0;
  if ((rc = launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g))) return rc;
  GemmArgs g2{};
  g2.A = h->gT; g2.lda = dp; g2.B = h->X; g2.ldb = dp; g2.C = D; g2.ldc = ld; g2.K = (int)dp;
  g2.r0 = 0; g2.r1 = (int)(n_pad / 64); g2.c0 = 0; g2.c1 = (int)(n_pad / 64); g2.lower = 0;
  return launch_gemm_cfg<64, 64, 2, 2, GEMM_SET, false>(h, st, g2);
}

int sigp_nlml_grad(sigp_handle* h, int kernel_id, const double theta[2], const double* Sigma, const double* MSigma, int64_t ldsigma,
                   int grad_mode, double* nlml, double grad[2]) {
  if (!h || !theta || !nlml || h->n == 0) return fail(h, SIGP_BAD_ARG, "nlml_grad: bad argument");
  if (grad_mode < 0 || grad_mode > 2) return fail(h, SIGP_BAD_ARG, "nlml_grad: grad_mode must be 0, 1 (reference formulae) or 2 (exact)");
  if (grad_mode != 0 && !grad) return fail(h, SIGP_BAD_ARG, "nlml_grad: grad buffer required");
  if (grad_mode != 0 && h->dtype != SIGP_F64) return fail(h, SIGP_BAD_ARG, "nlml_grad: gradients need the fp64 engine");
  if (grad_mode == 1 && kernel_id != SIGP_KERNEL_NETDIFFUSION) return fail(h, SIGP_BAD_ARG, "nlml_grad: the reference gradient formulae exist for the reference kernel only");
  if (grad_mode != 0 && kernel_id == SIGP_KERNEL_NETDIFFUSION && !MSigma) return fail(h, SIGP_BAD_ARG, "nlml_grad: M @ Sigma~ required");
  const double inf = std::numeric_limits<double>::infinity();
  const double ell = std::exp(theta[0]), snt = std::exp(theta[1]);
  if (!std::isfinite(ell) || !std::isfinite(snt)) { *nlml = inf; if (grad) grad[0] = grad[1] = inf; return SIGP_NOT_SPD; }
  double out[4];
  int rc = sigp_fit_predict(h, kernel_id, ell, snt, Sigma, ldsigma, out, nullptr, nullptr);
  if (rc == SIGP_NOT_SPD) { *nlml = inf; if (grad) grad[0] = grad[1] = inf; return rc; }
  if (rc) return rc;
  *nlml = out[1];
  if (grad_mode == 0) return SIGP_OK;

  // ---- K13/K14: tr(K~^-1 dK~) and A~^T dK~ A~ -------------------------------------------------------
  Slot& s = h->slots[0];
  hipStream_t st = s.s_upd;
  const long n = h->n, n_pad = h->n_pad, ld = n_pad;
  const int T = (int)(n_pad / NB);
  const double sf = out[0];
  if ((rc = ensure(h, &h->gU, &h->cap_gU, n_pad * n_pad))) return rc;
  if ((rc = ensure(h, &h->gK, &h->cap_gK, n_pad * n_pad))) return rc;
  if ((rc = ensure(h, &h->gD, &h->cap_gD, n_pad * n_pad))) return rc;
  if ((rc = ensure(h, &h->gPart, &h->cap_gPart, 4 * n_pad))) return rc;
  if ((rc = ensure(h, &h->scratchZ, &h->cap_Z, (long)RIDE * n_pad))) return rc;
  // U = L~^-T (upper triangular, row-major in gU); P parks in gK
  {
    ProfScope ps(h, st, SIGP_KC_MLII, (double)n_pad * n_pad * n_pad / 3, 0.0);
    if ((rc = trtri_levels<double>(h, st, s.mat, ld, s.dinv, h->gU, h->gK, ld, T, T))) return rc;
  }
  // K~^-1 = U U^T on the lower 128-tiles (LDS-DMA kernel; rows of U are zero left of their diagonal block, so tile (i, j)
  // sums k from 128 i: n^3/3 flops)
  {
    ProfScope ps(h, st, SIGP_KC_MLII, (double)n_pad * n_pad * n_pad / 3, 0.0);
    GemmArgs g{};
    g.A = h->gU; g.lda = ld; g.B = h->gU; g.ldb = ld; g.C = h->gK; g.ldc = ld; g.K = (int)n_pad;
    g.r0 = 0; g.r1 = T; g.c0 = 0; g.c1 = T; g.lower = 1; g.ktri = 1;
    if ((rc = launch_syrk128_t<double, true>(h, st, g))) return rc;
  }
  // A~ = L~^-T z = U z: one skinny product with the upper-triangular U (one wave per row, k from the diagonal) instead of a
  // backward block solve of 2 launches per 128 columns
  hipLaunchKernelGGL(rowdot_kernel<double>, dim3((unsigned)((n_pad + 3) / 4)), dim3(256), 0, st, (const double*)h->gU, ld, (int)n_pad, (int)n_pad, 2,
                     (const double*)(s.mat + n_pad * ld), ld, h->scratchZ, ld, 1, 0);
  HIPCHK(h, hipGetLastError());
  // dK~_1 (full symmetric): the reference's X (M Sigma~) X^T, or d k~/d log l for RBF / Matern
  if (kernel_id == SIGP_KERNEL_NETDIFFUSION) {
    if ((rc = build_xsxt(h, MSigma, ldsigma, h->gD))) return rc;
  } else {
    s.kps_host[0] = make_kparams(kernel_id == SIGP_KERNEL_RBF ? KID_RBF_DLOGL : KID_MATERN52_DLOGL, ell, 0.0, 0);
    if ((rc = upload_kparams(h, s, 1))) return rc;
    dim3 grid((unsigned)(n_pad / KB_TN), (unsigned)(n_pad / KB_TM), 1);
    launch_kbuild<double>(h, grid, st, h->X, 0L, (int)h->dp, (int)h->d, (int)n, h->gD, 0L, ld, s.kps, 1);
    HIPCHK(h, hipGetLastError());
  }
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)n), dim3(256), 0, st, h->gK, h->gD, h->scratchZ, ld, (int)n, h->gPart);
  HIPCHK(h, hipGetLastError());
  std::vector<double> part((size_t)4 * n);
  HIPCHK(h, hipMemcpyAsync(part.data(), h->gPart, part.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  double Td = 0, Qd = 0, trKinv = 0, aa = 0;   // T(D) = tr(K~^-1 D), Q(D) = A~^T D A~, tr(K~^-1), A~^T A~
  for (long i = 0; i < n; ++i) { Td += part[4 * i]; Qd += part[4 * i + 1]; trKinv += part[4 * i + 2]; aa += part[4 * i + 3]; }
  if (grad_mode == 1) {
    // reference formulae (north/June1st.py:248-252) with K = sf K~, alpha = A~/sf:
    //   dK_l = sf [X (M S~) X^T + sn~ I],  dK_s = sf [X S~ X^T + I],  X S~ X^T = K~ - sn~ I
    grad[0] = 0.5 * (Td + snt * trKinv) - (Qd + snt * aa) / (2.0 * sf);
    const double T2 = (double)n - snt * trKinv + trKinv;            // tr(K~^-1 (K~ - sn~ I + I))
    const double Q2 = (double)n * sf - snt * aa + aa;               // A~^T (K~ - sn~ I + I) A~, A~^T K~ A~ = y^T A~ = n sf
    grad[1] = 0.5 * T2 - Q2 / (2.0 * sf);
  } else {
    // exact derivative of the profiled nlML: d/dtheta = 1/2 tr(K~^-1 dK~) - A~^T dK~ A~ / (2 sf)
    const double scale1 = (kernel_id == SIGP_KERNEL_NETDIFFUSION) ? ell : 1.0;   // dK~/dlog l = l X (M S~) X^T
    grad[0] = scale1 * (0.5 * Td - Qd / (2.0 * sf));
    grad[1] = snt * (0.5 * trKinv - aa / (2.0 * sf));
  }
  return SIGP_OK;
}

// MLII for a lockstep group of RBF / Matern fits on the resident batch data (sigp_batch_upload): value and exact gradient of
// the profiled nlML (north/June1st.py:235-257 with the true derivative, as sigp_nlml_grad's grad_mode 2) for `count`
// (data set, theta) pairs -- fit i uses data set (first + i) % batch -- in groups of `group` members that advance through
// every launch together: covariance build, blocked Cholesky, L~^-T by recursive triangular inversion (grid.z = member),
// K~^-1 = U U^T (grid.y = member), A~ = U z, and the trace / quadratic-form reductions with dK~/dlog l recomputed on the
// fly.  This is what a multi-start optimiser over the retrospective years calls once per iteration.
// nlml [count], grad [count][2] (may be NULL with grad_mode 0); a non-SPD member gets +inf (the reference's except branch).
int sigp_nlml_grad_batch(sigp_handle* h, int64_t first, int64_t count, int kernel_id, const double* theta, int grad_mode, double* nlml, double* grad) {
  if (!h || h->b_count == 0 || first < 0 || count < 1 || !theta || !nlml) return fail(h, SIGP_BAD_ARG, "nlml_grad_batch: bad argument (sigp_batch_upload first)");
  if (h->dtype != SIGP_F64) return fail(h, SIGP_BAD_ARG, "nlml_grad_batch: fp64 engine only");
  if (kernel_id != SIGP_KERNEL_RBF && kernel_id != SIGP_KERNEL_MATERN52) return fail(h, SIGP_BAD_ARG, "nlml_grad_batch: RBF / MATERN52 only");
  if (grad_mode != 0 && grad_mode != 2) return fail(h, SIGP_BAD_ARG, "nlml_grad_batch: grad_mode 0 (value) or 2 (exact gradient)");
  if (grad_mode != 0 && !grad) return fail(h, SIGP_BAD_ARG, "nlml_grad_batch: grad buffer required");
  HIPCHK(h, hipSetDevice(h->device));
  const long n = h->b_n, d = h->b_d, dp = h->b_dp, n_pad = h->b_npad, ld = n_pad;
  const int T = (int)(n_pad / NB);
  const int G = (int)std::max<long>(1, std::min<long>(h->opt_group, count));
  const double inf = std::numeric_limits<double>::infinity();
  Slot& s = h->slots[0];
  hipStream_t st = s.s_upd;
  int rc;
  if ((rc = slot_reserve(h, s, n_pad, G))) return rc;
  if (grad_mode != 0) {
    if ((rc = ensure(h, &h->gU, &h->cap_gU, (long)G * n_pad * n_pad))) return rc;
    if ((rc = ensure(h, &h->gK, &h->cap_gK, (long)G * n_pad * n_pad))) return rc;
    if ((rc = ensure(h, &h->gPart, &h->cap_gPart, (long)G * (4 * n_pad + 4)))) return rc;
    if ((rc = ensure(h, &h->scratchZ, &h->cap_Z, (long)G * n_pad))) return rc;
  }
  std::vector<double> sums((size_t)G * 4);
  for (long g0 = 0; g0 < count; g0 += G) {
    const int nb = (int)std::min<long>(G, count - g0);
    std::vector<double> ell((size_t)nb), snt((size_t)nb);
    std::vector<char> ok((size_t)nb, 1);
    for (int b = 0; b < nb; ++b) {
      ell[(size_t)b] = std::exp(theta[2 * (g0 + b)]); snt[(size_t)b] = std::exp(theta[2 * (g0 + b) + 1]);
      if (!std::isfinite(ell[(size_t)b]) || !std::isfinite(snt[(size_t)b]) || !(ell[(size_t)b] > 0)) { ok[(size_t)b] = 0; ell[(size_t)b] = 1.0; snt[(size_t)b] = 1.0; }
      s.kps_host[b] = make_kparams(kernel_id, ell[(size_t)b], snt[(size_t)b], (int)((first + g0 + b) % h->b_count));
    }
    if ((rc = upload_kparams(h, s, nb))) return rc;
    if ((rc = build_cov(h, s, nb, h->bX, n_pad * dp, h->by, n_pad, h->bXs, (long)RIDE * dp, n, d, dp, n_pad, 0))) return rc;
    if ((rc = potrf_slot(h, s, nb, n_pad, false, 1))) return rc;
    if ((rc = epilogue_slot(h, s, nb, n, n_pad, 0))) return rc;
    if (grad_mode != 0) {
      // U = L~^-T for every member at once, P parks in gK
      {
        ProfScope ps(h, st, SIGP_KC_MLII, nb * (double)n_pad * n_pad * n_pad / 3, 0.0);
        if ((rc = trtri_levels<double>(h, st, s.mat, ld, s.dinv, h->gU, h->gK, ld, T, T, nb, s.matStride, s.dinvStride, n_pad * n_pad))) return rc;
      }
      {   // K~^-1 = U U^T, lower tiles
        ProfScope ps(h, st, SIGP_KC_MLII, nb * (double)n_pad * n_pad * n_pad / 3, 0.0);
        GemmArgs g{};
        g.A = h->gU; g.lda = ld; g.B = h->gU; g.ldb = ld; g.C = h->gK; g.ldc = ld; g.K = (int)n_pad;
        g.batch = nb; g.sA = g.sB = g.sC = n_pad * n_pad;
        g.r0 = 0; g.r1 = T; g.c0 = 0; g.c1 = T; g.lower = 1; g.ktri = 1;
        if ((rc = launch_syrk128_t<double, true>(h, st, g))) return rc;
      }
      // A~ = U z  (z = solved ride row 0 of each member)
      hipLaunchKernelGGL(rowdot_kernel<double>, dim3((unsigned)((n_pad + 3) / 4), (unsigned)nb), dim3(256), 0, st, (const double*)h->gU, ld, (int)n_pad, (int)n_pad, 2,
                         (const double*)(s.mat + n_pad * ld), ld, h->scratchZ, ld, 1, 0, (const double*)nullptr, 0, 0, n_pad * n_pad, s.matStride, n_pad);
      HIPCHK(h, hipGetLastError());
      // derivative covariance ids for the on-the-fly dK~/dlog l (same data sets, same length scales); from pageable memory: the
      // copy call returns once the host buffer has been staged, so the vector may go out of scope
      std::vector<KParams> dkp((size_t)nb);
      for (int b = 0; b < nb; ++b) { dkp[(size_t)b] = make_kparams(kernel_id, ell[(size_t)b], snt[(size_t)b], (int)((first + g0 + b) % h->b_count)); dkp[(size_t)b].kernel_id = kernel_id == SIGP_KERNEL_RBF ? KID_RBF_DLOGL : KID_MATERN52_DLOGL; }
      if (h->cap_gKps < nb) {
        if (h->gKps) HIPCHK(h, hipFree(h->gKps));
        h->gKps = nullptr; h->cap_gKps = 0;
        HIPCHK(h, hipMalloc((void**)&h->gKps, (size_t)G * sizeof(KParams)));
        h->cap_gKps = G;
      }
      HIPCHK(h, hipMemcpyAsync(h->gKps, dkp.data(), (size_t)nb * sizeof(KParams), hipMemcpyHostToDevice, st));
      double* part = h->gPart;                   // [nb][n_pad][4], then [nb][4]
      hipLaunchKernelGGL(grad_reduce_cov_kernel, dim3((unsigned)((n + GR_ROWS - 1) / GR_ROWS), (unsigned)nb), dim3(256), 0, st, (const double*)h->gK, n_pad * n_pad, ld,
                         (const double*)h->scratchZ, n_pad, (const double*)h->bX, n_pad * dp, (int)dp, (int)d, (int)n, (const KParams*)h->gKps, part, 4 * n_pad);
      hipLaunchKernelGGL(grad_sum_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const double*)part, 4 * n_pad, (int)n, part + (long)G * 4 * n_pad);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(sums.data(), part + (long)G * 4 * n_pad, (size_t)nb * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if ((rc = sync_slot(h, s))) return rc;
    for (int b = 0; b < nb; ++b) {
      const long i = g0 + b;
      double out[4];
      finish_results(s.res_host + 512 * b, ok[(size_t)b] ? s.info_host[b] : 1, n, 0, snt[(size_t)b], nullptr, out, nullptr, nullptr);
      nlml[i] = out[1];
      if (grad_mode == 0) continue;
      if (!ok[(size_t)b] || s.info_host[b] != 0) { grad[2 * i] = grad[2 * i + 1] = inf; continue; }
      const double sf = out[0], Td = sums[(size_t)b * 4], Qd = sums[(size_t)b * 4 + 1], trKinv = sums[(size_t)b * 4 + 2], aa = sums[(size_t)b * 4 + 3];
      grad[2 * i] = 0.5 * Td - Qd / (2.0 * sf);                      // d/dlog l   of the profiled nlML
      grad[2 * i + 1] = snt[(size_t)b] * (0.5 * trKinv - aa / (2.0 * sf));   // d/dlog sn~
    }
  }
  h->built = h->factored = h->fitted = false;
  return SIGP_OK;
}

#include "sigp_callers.inc"   // sigp_small_*, sigp_corr_tau, sigp_area_sums, sigp_detrend

}  // extern "C"

namespace {
#include "sigp_shard.inc"     // the sharded fit: transports, panel loop, sharded solves (templates: C++ linkage)
}  // namespace

extern "C" {

#include "sigp_dist.inc"      // sigp_dist_*: the C ABI of the sharded fit

int sigp_get_stat(sigp_handle* h, const char* name, double* value) {
  if (!h || !name || !value) return SIGP_BAD_ARG;
  if (!strcmp(name, "refine_residual")) { *value = h->refine_resid; return SIGP_OK; }
  if (!strcmp(name, "matrix_bytes")) {
    double b = 0;
    for (const auto& s : h->slots) if (s.mat) b += (double)s.capB * (double)(s.cap_npad + RIDE) * (double)s.cap_npad * sizeof(double);
    if (h->fmat) b += (double)std::max(1, h->cap_f_G) * (double)(h->cap_f_npad + RIDE) * (double)h->cap_f_npad * sizeof(float);
    if (h->dl.mat) b += (double)h->dl.cap;
    *value = b;
    return SIGP_OK;
  }
  const struct { const char* nm; const double* v; } ds[] = {
      {"dist_fit_ms", &h->dc.st_fit_ms}, {"dist_factor_ms", &h->dc.st_factor_ms}, {"dist_bcast_bytes", &h->dc.st_bcast_bytes}, {"dist_comm_ms", &h->dc.st_comm_ms},
      {"dist_stall_ms", &h->dc.st_stall_ms}, {"dist_solve_ms", &h->dc.st_solve_ms}, {"dist_collectives", &h->dc.st_collectives},
      {"dist_host_comm_ms", &h->dc.st_host_comm_ms}, {"dist_enqueue_ms", &h->dc.st_enqueue_ms}, {"dist_link_bytes", &h->dc.st_link_bytes},
      {"dist_owner_ms", &h->dc.st_owner_ms}, {"dist_split_panels", &h->dc.st_split_panels}, {"dist_link_panel_max", &h->dc.st_link_panel_max}};
  for (const auto& e : ds)
    if (!strcmp(name, e.nm)) { *value = *e.v; return SIGP_OK; }
  if (!strcmp(name, "dist_comm_ranks")) {        // what the communicator itself reports (ncclCommCount); 0 = no RCCL communicator on this handle
    *value = 0;
    if (h->dc.kind == 1 && h->dc.comm) {
      RcclApi* api = rccl_api(nullptr);
      int cnt = 0;
      if (api && api->CommCount && api->CommCount((ncclComm_t)h->dc.comm, &cnt) == ncclSuccess) *value = cnt;
    }
    return SIGP_OK;
  }
  return fail(h, SIGP_BAD_ARG, "get_stat: unknown name %s", name);
}

int sigp_profile(sigp_handle* h, int enable) {
  if (!h) return SIGP_BAD_ARG;
  if (!enable) prof_drain(h);
  // enable = 1: every class; otherwise a bit mask (bit k+8 = class k), e.g. (1 << (8 + SIGP_KC_SYRK128))
  h->prof = enable == 0 ? 0u : enable == 1 ? 0xffu : ((unsigned)enable >> 8) & 0xffu;
  return SIGP_OK;
}

int sigp_synchronize(sigp_handle* h) {
  if (!h) return SIGP_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  return SIGP_OK;
}

int sigp_profile_reset(sigp_handle* h) {
  if (!h) return SIGP_BAD_ARG;
  prof_drain(h);
  for (int k = 0; k < SIGP_KC_COUNT; ++k) { h->p_ms[k] = 0; h->p_n[k] = 0; h->p_flops[k] = 0; h->p_bytes[k] = 0; }
  return SIGP_OK;
}

int sigp_profile_get(sigp_handle* h, int kclass, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (!h || kclass < 0 || kclass >= SIGP_KC_COUNT) return SIGP_BAD_ARG;
  prof_drain(h);
  if (total_ms) *total_ms = h->p_ms[kclass];
  if (launches) *launches = h->p_n[kclass];
  if (flops) *flops = h->p_flops[kclass];
  if (bytes) *bytes = h->p_bytes[kclass];
  return SIGP_OK;
}

#ifdef SIGP_DEBUG_TOOLS
#include "sigp_debug.inc"
#endif

}  // extern "C"
