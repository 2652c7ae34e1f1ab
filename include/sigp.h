/*
 * sigp.h -- C ABI of the MI355X-native Gaussian-process engine (libsigp.so).
 *
 * Drop-in boundary for the GPR hot path of William-gregory/SeaIceExtentForecasting.  The reference has
 * no function boundary there: the path is the inline statement block
 *     north/June1st.py:231-277   (M ... fvar)          and the closure
 *     north/June1st.py:235-257   MLII(hyperparameters) -> (nlML, grad[2])
 * repeated byte-identically in all 14 forecast scripts.  Each entry point below cites
 * the statements it replaces.  Binding on the reference side is ctypes.
 *
 * Conventions
 *   - plain C, no C++ exceptions cross the ABI; every call returns SIGP_OK / SIGP_NOT_SPD /
 *     SIGP_BAD_ARG / SIGP_HIP_ERROR; message via sigp_last_error().
 *   - host arrays are row-major float64 (NumPy C order) owned by the caller; the library owns all
 *     device memory and frees it in sigp_destroy().
 *   - one handle = one GPU + its streams and workspaces.  Calls on a handle are serialised on its
 *     streams and are synchronous on return; a handle is NOT thread-safe, distinct handles are
 *     independent.
 *   - there is no CPU fallback: without a usable HIP device sigp_create() fails with SIGP_HIP_ERROR.
 */
#ifndef SIGP_H
#define SIGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sigp_handle sigp_handle;

enum { SIGP_OK = 0, SIGP_NOT_SPD = 1, SIGP_BAD_ARG = 2, SIGP_HIP_ERROR = 3 };
enum { SIGP_F64 = 0, SIGP_F32 = 1 };
/* covariance functions: 0 is the reference's own (north/June1st.py:264-265); 1,2 are the kernels
 * BASELINE.json's configs add on the same fit/solve/predict skeleton. */
enum { SIGP_KERNEL_NETDIFFUSION = 0, SIGP_KERNEL_RBF = 1, SIGP_KERNEL_MATERN52 = 2 };
/* sigp_get_matrix selectors */
enum { SIGP_MAT_K = 0, SIGP_MAT_L = 1 };
/* kernel classes for sigp_profile_get (one per device kernel, so totals line up with rocprofv3 --stats):
 * kbuild_kernel(+ride_build), potrf_diag_kernel, gemm_mfma_kernel<32,128> (panel solve),
 * gemm_mfma_kernel<64,64> (updates with few tiles), syrk128_kernel (inner + trailing updates), epilogue_kernel */
enum { SIGP_KC_KBUILD = 0, SIGP_KC_DIAG = 1, SIGP_KC_TRSM = 2, SIGP_KC_UPDATE_SMALL = 3,
       SIGP_KC_SYRK128 = 4, SIGP_KC_EPILOGUE = 5, SIGP_KC_SMALL = 6 /* smallgp_kernel */,
       SIGP_KC_MLII = 7 /* triangular inversion + U U^T of sigp_nlml_grad */, SIGP_KC_COUNT = 8 };

#define SIGP_MAX_RIDE 127 /* test points that can ride along one factorisation */

int sigp_version(void);
/* "HIP runtime <version> (<status>) from <path of the libamdhip64 that serves this process>" -- for error reports: a process that
 * also loads PyTorch-ROCm has two HIP runtimes on disk, and whichever is mapped first serves everyone. */
int sigp_runtime_info(char* buf, int64_t len);

/* lifetime ------------------------------------------------------------------------------------ */
int sigp_create(sigp_handle** h, int device_id, int dtype);
int sigp_destroy(sigp_handle* h);
const char* sigp_last_error(const sigp_handle* h);

/* data staging: the enclosing-scope variables the reference block reads ------------------------- */
/* X [n,d] (row stride ldx), y [n]:  north/June1st.py:214, 226-229 (y, X).  Copies host -> HBM. */
int sigp_set_train(sigp_handle* h, const double* X, int64_t n, int64_t d, int64_t ldx, const double* y);
/* Xs [m,d], m <= SIGP_MAX_RIDE: the test row(s) north/June1st.py:228 (Xs).  These "ride along" the
 * factorisation as extra right-hand-side rows, so fit+predict needs no separate triangular solve.
 * m = 0 clears them. */
int sigp_set_test(sigp_handle* h, const double* Xs, int64_t m, int64_t ldxs);

/* K5: kernel-matrix build  K~ = k(X,X) + sn_tilde*I  into HBM (lower triangle), plus the ride-along
 * rows [y ; k~(Xs,X)].  Replaces north/June1st.py:265 (the multi_dot + eye term).
 * RBF / Matern-5/2: ell is the length scale.  */
int sigp_kernel_build(sigp_handle* h, int kernel_id, double ell, double sn_tilde);
/* Reference kernel: host passes Sigma~ = expm(ell*M) [N,N] (north/June1st.py:264; N = d of set_train);
 * device forms X Sigma~ X^T + sn_tilde*I (north/June1st.py:265). */
int sigp_kernel_build_from_sigma(sigp_handle* h, const double* Sigma, int64_t ldsigma, double sn_tilde);

/* K6: blocked Cholesky  K~ = L~ L~^T  in place (north/June1st.py:265 np.linalg.cholesky).
 * info: 0, or LAPACK-style 1-based index of the first non-positive pivot (-> SIGP_NOT_SPD). */
int sigp_potrf(sigp_handle* h, int64_t* info);

/* K7,K8,K12: A~ = K~^-1 y, sigma_f = y^T A~/n (north/June1st.py:266-268), nlML (north/June1st.py:246).
 * Requires sigp_potrf. */
int sigp_fit(sigp_handle* h, double* sigma_f, double* nlml);

/* K9-K11 for the ride-along test rows: mean[m], var[m] (var includes sigma_n: north/June1st.py:272-277). */
int sigp_predict_ride(sigp_handle* h, double* mean, double* var);
/* K9-K11 for arbitrary new test points after a fit (any m): cross-kernel build + forward solve. */
int sigp_predict(sigp_handle* h, const double* Xs, int64_t m, int64_t ldxs, double* mean, double* var);

/* Fused hot path: build -> potrf -> fit -> predict_ride with one host synchronisation.
 * Sigma may be NULL unless kernel_id == SIGP_KERNEL_NETDIFFUSION.
 * out[4] = { sigma_f, nlml, (double)info, sigma_n };  mean/var [m_ride] may be NULL when m_ride == 0. */
int sigp_fit_predict(sigp_handle* h, int kernel_id, double ell, double sn_tilde, const double* Sigma,
                     int64_t ldsigma, double* out, double* mean, double* var);

/* Batch of independent fits that share (n, d, m): the retrospective loop over years
 * (north/retrospective_forecasts/September1st_retro.py:176-248) and the hyper-parameter grid
 * (north/June1st.py:210-211).  Problem b uses X + b*strideX, y + b*stridey, Xs + b*strideXs
 * (a stride of 0 shares the array between problems, e.g. one data set x many grid points).
 * RBF / Matern only.  out [batch,4], mean/var [batch,m].  Fits are pipelined over `concurrency`
 * stream sets (1..16) so one fit's panel factorisations overlap another's trailing updates. */
int sigp_fit_batch(sigp_handle* h, int64_t batch, int kernel_id, const double* X, int64_t strideX,
                   const double* y, int64_t stridey, const double* Xs, int64_t strideXs, int64_t n,
                   int64_t d, int64_t m, const double* ell, const double* sn_tilde, int concurrency,
                   double* out, double* mean, double* var);
/* The same batch in two steps: sigp_batch_upload copies the data sets (host pointers, same layout/strides as
 * sigp_fit_batch) into HBM once, where they stay resident; sigp_batch_run then runs fits [first, first+count) on the
 * resident data (fit i uses data set i % batch) -- the timed region of bench.py is one sigp_batch_run call. */
int sigp_batch_upload(sigp_handle* h, int64_t batch, const double* X, int64_t strideX, const double* y,
                      int64_t stridey, const double* Xs, int64_t strideXs, int64_t n, int64_t d, int64_t m);
/* allocate the lockstep slots for (group, concurrency) ahead of time, so no allocation falls inside a timed batch_run */
int sigp_batch_reserve(sigp_handle* h, int64_t group, int concurrency);
int sigp_batch_run(sigp_handle* h, int64_t first, int64_t count, int kernel_id, const double* ell,
                   const double* sn_tilde, int concurrency, double* out, double* mean, double* var);

/* The reference's OWN kernel at the reference's OWN size, batched: the retro loop
 * (north/retrospective_forecasts/September1st_retro.py:176-248: 3 regions x ~40 years of independent fits, n = year - 1979 <= 45)
 * times the 20 x 20 (l, sn~) grid of north/June1st.py:210-211 -- ONE WORKGROUP PER FIT, everything in LDS, one launch for the lot.
 * Each fit is north/June1st.py:264-277 + :246: K~ = X expm(l M) X^T + sn~ I, Cholesky, A~, sigma_f, k*, k**, fmean, fvar, nlML.
 * Data sets are ragged (n, N, m per set, 1 <= n <= 128, 0 <= m <= 8) and staged once in factored form (one
 * eigendecomposition per data set serves every l of the grid):
 *   A   [(n+m)][N] = [X ; Xs] Q      lam [N] = eigenvalues of M = Q diag(lam) Q^T        (lam_mode 0: weights exp(l lam_k))
 *   A   [(n+m)][N] = [X ; Xs] U      lam [N] = eigenvalues of a host-side Sigma~ = U diag(lam) U^T   (lam_mode 1: weights lam_k;
 *                                    this is how scipy's Pade expm(l M) -- the reference's own numbers at l = 3.1e10 -- goes through)
 * so that K~ = A_train diag(w) A_train^T + sn~ I.  Pools are flat double arrays; *_off give each set's start (in doubles).
 * sigp_small_run: fit i uses data set set_index[i] with (ell[i], sn_tilde[i]); out [nprob][4] = sigma_f, nlML, info, sigma_n
 * (info > 0: LAPACK pivot index, other entries +inf, mean/var NaN -- north/June1st.py:254-256); mean / var [nprob][mstride]. */
int sigp_small_upload(sigp_handle* h, int64_t nsets, const int64_t* n, const int64_t* N, const int64_t* m, const int32_t* lam_mode,
                      const double* A_pool, const int64_t* A_off, const double* y_pool, const int64_t* y_off,
                      const double* lam_pool, const int64_t* lam_off);
int sigp_small_run(sigp_handle* h, int64_t nprob, const int64_t* set_index, const double* ell, const double* sn_tilde,
                   double* out, double* mean, double* var, int64_t mstride);
/* The same launch with the MLII closure's second return value (north/June1st.py:235-257, the call the reference left commented out at
 * :259-262 `minimize(MLII, x0, method='CG', jac=True)`), for every (data set, theta) of the list at once: out8 [nprob][8] =
 * sigma_f, nlML, info, sigma_n, then the reference's own 2-vector "gradient" (:248-252: tr(K^-1 dK)/2 - alpha^T dK alpha/2 with
 * dKdl = X (M Sigma) X^T + sigma_n I, dKds = X Sigma X^T + sigma_f I -- NOT the derivative of nlML), then the exact
 * derivative of the profiled nlML w.r.t. (log l, log sn~).  A non-SPD K~ gives +inf in all of them (:254-256).  The inverse factor
 * L~^-1 is formed in LDS over L~; M Sigma~ = Q diag(lam exp(l lam)) Q^T needs no second matrix (lam_mode 0).  Sets staged with
 * lam_mode 1 (weights of a host-side Pade expm) need the derivative weights dlam_k = u_k^T M u_k * lam_k, given once per upload by
 * sigp_small_set_dweights (dlam_pool mirrors lam_pool: same offsets, `count` = its total length); without them their gradient is NaN. */
int sigp_small_set_dweights(sigp_handle* h, const double* dlam_pool, int64_t count);
int sigp_small_run_grad(sigp_handle* h, int64_t nprob, const int64_t* set_index, const double* ell, const double* sn_tilde,
                        double* out8, double* mean, double* var, int64_t mstride);

/* The caller that produces the GP's features: ComplexNetworks.Network.tau (ComplexNetworks.py:31-47) on the device.
 * series [N][T] (row stride lds): the anomaly series of the N active grid cells.  Forms the N x N cell-to-cell correlation matrix
 * (np.corrcoef: one fp64 MFMA product of the standardised rows, clipped to [-1, 1], NaN on the diagonal) into R [N][N] (row stride
 * ldr; may be NULL) and returns the sum and the count of its entries with r >= 0 and r > r_crit -- the entries whose one-sided
 * t-test p-value (dof = T - 2) is below the significance level, r_crit = t_c / sqrt(dof + t_c^2), t_c = t.isf(significance, dof).
 * tau = sum / count. */
int sigp_corr_tau(sigp_handle* h, const double* series, int64_t N, int64_t T, int64_t lds, double r_crit, double* R, int64_t ldr,
                  double* sum_out, double* count_out);

/* ComplexNetworks.Network.intra_links (ComplexNetworks.py:298-309): the per-area anomaly series that become the GP's features.
 * data [P][T] pixel-major, weight [P] (sqrt(cell area) or sqrt(cos lat)), label [P] = area index 0..A-1 of the pixel or -1;
 * out [A][T] = sum over the area's pixels of data * weight (NaN products as 0), pixels added in ascending order. */
int sigp_area_sums(sigp_handle* h, const double* data, int64_t P, int64_t T, const double* weight, const int32_t* label, int64_t A, double* out);

/* ComplexNetworks.Network.area_level (ComplexNetworks.py:49-281) on the HOST (sequential greedy region growing + merging; no
 * handle, no GPU): the same decisions as the reference, in the same order, on means formed in np.nanmean's summation order, so the
 * areas are identical (csrc/area_level.cpp).  R [N][N]: the cell-to-cell correlations (NaN on the diagonal) of the N active cells;
 * node_of_cell [dimX*dimY]: row of R of a grid cell (row-major), -1 for an inactive one; cell_nan: flat index of the first NaN
 * cell of the data (the reference's out-of-bounds sentinel); tau: the threshold of sigp_corr_tau; latlon != 0: left / right
 * neighbours wrap.  Outputs (caller-allocated, N entries each, area_offsets N + 1): the areas in the reference's dict order --
 * area_ids[a] = its key, cells_out[area_offsets[a] .. area_offsets[a+1]) = its cells (flat indices) in list order -- and the
 * cells set aside by the merging step, in order (Network.unavail): at most unavail_cap of them are written, *n_unavail_out is the
 * full length (it can exceed N on a lat-lon grid; call again with a larger buffer then). */
int sigp_area_level(const double* R, int64_t N, const int32_t* node_of_cell, int64_t dimX, int64_t dimY, int32_t cell_nan, double tau,
                    int latlon, int32_t* cells_out, int64_t* area_offsets, int32_t* area_ids, int64_t* n_areas_out, int32_t* unavail_out,
                    int64_t unavail_cap, int64_t* n_unavail_out);
/* np.nanmean of a contiguous array in NumPy's own summation order (what sigp_area_level decides on; exported for its test) */
double sigp_host_nanmean(const double* a, int64_t n);

/* The step before the feature pipeline: detrend() of north/June1st.py:179-194 (one cut: cut_len = {T}) and of the retro
 * scripts (north/retrospective_forecasts/June1st_retro.py:178-195: one detrended cube per cut-off year) in ONE launch.
 * data [P][T] pixel-major anomaly series; cut c removes the least-squares line (scipy.stats.linregress semantics, NaN propagating)
 * fitted to the first cut_len[c] steps.  dt_out: the cuts' [P][cut_len[c]] blocks back to back; trend_out [ncuts][P][2] = slope, intercept. */
int sigp_detrend(sigp_handle* h, const double* data, int64_t P, int64_t T, int64_t ncuts, const int64_t* cut_len, double* dt_out, double* trend_out);

/* named scalars of the last operation: "refine_residual" (fp32 engine: max|y - K~ alpha~| / max|y| after the last refinement
 * step), "matrix_bytes" (device bytes held by this handle's matrix / factor buffers), "dist_*" (see sigp_dist_fit). */
int sigp_get_stat(sigp_handle* h, const char* name, double* value);

/* K7 (explicit): alpha~ = K~^-1 y  [n]  (north/June1st.py:266; alpha of :271 is alpha~/sigma_f). */
int sigp_get_alpha(sigp_handle* h, double* alpha_tilde);
/* copy the lower triangle of K~ (before potrf) or L~ (after) to host, [n,n] row-major, upper = 0 */
int sigp_get_matrix(sigp_handle* h, int which, double* out, int64_t ldo);

/* K13/K14: MLII(theta) (north/June1st.py:235-257): theta = (log ell, log sn_tilde).
 * grad_mode 0 = none, 1 = reference formulae (:248-252), 2 = exact derivative of the profiled nlML.
 * Non-SPD -> returns SIGP_NOT_SPD and nlml = grad = +inf (the reference's except branch :254-256).
 * For the reference kernel the caller passes Sigma~ and dSigma = M @ Sigma~ (host, [N,N]). */
int sigp_nlml_grad(sigp_handle* h, int kernel_id, const double theta[2], const double* Sigma,
                   const double* MSigma, int64_t ldsigma, int grad_mode, double* nlml, double grad[2]);

/* MLII for a lockstep group of fits on the data sets resident after sigp_batch_upload (RBF / Matern, fp64): value and EXACT gradient
 * of the profiled nlML (north/June1st.py:235-257 with the true derivative = sigp_nlml_grad's grad_mode 2) for `count` (data set,
 * theta) pairs; fit i uses data set (first + i) % batch and theta[2i .. 2i+1] = (log ell, log sn~).  Groups of `group` members
 * (sigp_set_option) advance through every launch together: build, blocked Cholesky, L~^-T by recursive triangular inversion,
 * K~^-1 = U U^T, A~ = U z, reductions with dK~/dlog ell recomputed on the fly.  One call per optimiser iteration serves every
 * retrospective year (the call the reference left commented out, north/June1st.py:259-262, batched over the retro loop of
 * north/retrospective_forecasts/September1st_retro.py:176-248).  grad_mode 0 (value only; grad may be NULL) or 2.
 * nlml [count], grad [count][2]; a non-SPD member gets +inf in both (the reference's except branch :254-256). */
int sigp_nlml_grad_batch(sigp_handle* h, int64_t first, int64_t count, int kernel_id, const double* theta, int grad_mode, double* nlml, double* grad);

/* One large fit sharded over the GPUs of a node (BASELINE configs[3] fp64, configs[4] fp32 + fp64 refinement): 1-D block-cyclic
 * ownership of outer panels (W column blocks of 128; panel q belongs to rank q % nranks), OWNER-ONLY storage -- a rank allocates,
 * builds and updates only the block columns of its own panels (per-rank matrix bytes ~ 1/nranks) -- and the block-row panel
 * broadcast inside the library: per panel the owner factors it on its panel stream, packs [rows below its top block, ride rows
 * included] x [its columns] (what the later panels read; the last panel does not travel) in SEGMENTS of `dist_segment` column blocks:
 * a segment is broadcast on a communication stream (RCCL over xGMI) as soon as its last column is solved, while the owner's chain
 * goes on with the next columns and every rank is still applying the previous panel (look-ahead); the next panel's owner applies
 * each segment to its columns as it arrives, every other rank assembles the segments and updates its later panels in one launch.
 * Streams are ordered with events only: no host synchronisation and no device-to-host read per panel; the pivot info is MIN-reduced once at the end together with the 512 partial ride-row reductions (the only other fp64 exchange).
 * Replaces north/June1st.py:265 (np.linalg.cholesky) and :266-277, :246 for one large K~, as sigp_fit_predict does on one GPU.
 * fp32 handles (SIGP_F32): the fp32 factor is sharded the same way; x = K~^-1 [y k*] then comes from triangular solves on the
 * DISTRIBUTED factor (per panel: one skinny product with the explicit inverse of the owner's diagonal block + one skinny update
 * of the owner's column panel, and one 16 KB collective) and the fp64 residual of the iterative refinement is sharded by rows
 * (each rank recomputes the covariance of its rows only; one all-reduce assembles it).
 *
 *   sigp_dist_unique_id(id)            rank 0: a fresh 128-byte ncclUniqueId, to be handed to the other ranks by any channel
 *   sigp_dist_init(h, nranks, rank, id)  open THIS handle's communicator on its device (collective: every rank calls it).
 *                                      librccl is bound at run time (dlopen): a process that already has an RCCL mapped -- PyTorch-ROCm
 *                                      ships its own -- gets that one; nranks == 1 needs no RCCL at all (id NULL; with an id a one-rank
 *                                      communicator is opened and every collective of the fit still goes through it)
 *   sigp_dist_init_transport(...)      instead of the library's communicator, the caller's: callbacks that move a buffer.  device_buffers
 *                                      = 1: they receive DEVICE pointers and the hipStream_t to enqueue on (a communicator the caller
 *                                      already owns); 0: HOST pointers (the library stages through pinned memory and synchronises per
 *                                      collective -- rehearsal on a box with fewer GPUs than ranks, fabrics without device collectives).
 *                                      Same panel loop, same arithmetic, same order.
 *   sigp_dist_fit(...)                 after sigp_set_train / sigp_set_test with the SAME data on every rank (n d 8 bytes): the whole
 *                                      sharded fit; out / mean / var as sigp_fit_predict, identical on every rank.  Sigma as there
 *                                      (reference kernel: fp64 only).  SIGP_NOT_SPD + LAPACK pivot in out[2] on every rank otherwise.
 *                                      The factor stays spread over the ranks; other test points: sigp_dist_predict.
 *   set_option("owner_only", 1) before sigp_set_train keeps an fp64 handle from allocating the n x n single-GPU matrix;
 *   set_option("dist_stats", 1) times the broadcasts with HIP events: sigp_get_stat "dist_fit_ms", "dist_factor_ms" (device time of the
 *   panel loop), "dist_bcast_bytes", "dist_comm_ms" (per panel: first segment ready -> last segment arrived, summed), "dist_stall_ms" (time
 *   the update stream sat idle waiting for a panel: what look-ahead did NOT hide), "dist_solve_ms", "dist_collectives", "dist_comm_ranks"
 *   (what the communicator itself reports: ncclCommCount), "dist_host_comm_ms", "dist_enqueue_ms" (host time inside the collectives'
 *   enqueue calls / for the whole panel loop). */
typedef struct sigp_transport {
  void* ctx;
  int device_buffers;
  /* broadcast `bytes` from rank `root` in place; stream: hipStream_t to order on (device_buffers) or NULL (host pointers: blocking) */
  int (*bcast)(void* ctx, void* buf, uint64_t bytes, int root, void* stream);
  /* all-reduce `count` elements in place; is_f32: float, else double; op 0 = sum, 1 = min.  Non-zero return = failure. */
  int (*allreduce)(void* ctx, void* buf, uint64_t count, int is_f32, int op, void* stream);
  /* Only for set_option("dist_panel_split", 1) (may be NULL otherwise), both in place on a buffer of nranks chunks of `chunk_bytes`:
   * scatter: the root's chunk r goes to rank r at buf + r chunk_bytes (the root keeps its own);
   * allgather: rank r contributes the chunk at buf + r chunk_bytes, every rank ends with all of them. */
  int (*scatter)(void* ctx, void* buf, uint64_t chunk_bytes, int root, void* stream);
  int (*allgather)(void* ctx, void* buf, uint64_t chunk_bytes, void* stream);
} sigp_transport;
int sigp_dist_unique_id(void* id128);
int sigp_dist_init(sigp_handle* h, int nranks, int rank, const void* nccl_id);
int sigp_dist_init_transport(sigp_handle* h, int nranks, int rank, const sigp_transport* transport);
/* The same for a caller compiled against ANOTHER version of this header: struct_bytes = sizeof(sigp_transport) as the caller knows it; members
 * beyond it read as NULL (the struct grew from four to six members in 4.0: a 3.x caller passes its shorter struct here, or sets scatter /
 * allgather to NULL).  sigp_version() >= 500 has this entry point. */
int sigp_dist_init_transport2(sigp_handle* h, int nranks, int rank, const sigp_transport* tr, int64_t struct_bytes);
int sigp_dist_fit(sigp_handle* h, int kernel_id, double ell, double sn_tilde, const double* Sigma, int64_t ldsigma, int64_t W, int lookahead,
                  double* out, double* mean, double* var);
/* Predictions at NEW test points after a sharded fit (north/June1st.py:272-277): collective -- every rank calls it with the same Xs [m,d]
 * and gets the same mean / var (var includes sigma_n).  mean = k~*^T alpha~ with alpha~ = K~^-1 y replicated once per fit (fp64: one backward
 * solve on the distributed factor; fp32: the refined solution); var from forward solves on the distributed factor, 4 points per pass, the
 * squares summed per rank and all-reduced once.  RBF / Matern fits; fp32 handles: the variance carries the fp32 factor's accuracy
 * (as sigp_predict on an fp32 handle). */
int sigp_dist_predict(sigp_handle* h, const double* Xs, int64_t m, int64_t ldxs, double* mean, double* var);
int sigp_dist_shutdown(sigp_handle* h);                                /* destroy the communicator, free the sharded storage */
int64_t sigp_num_blocks(sigp_handle* h);                               /* T = n_pad / 128                        */

/* measurement ---------------------------------------------------------------------------------- */
/* enable=1: bracket every kernel launch with HIP events on the stream it is launched on and
 * accumulate per kernel class; enable = bit mask with bit (8+k) set: only kernel class k (less perturbation);
 * enable=0: off (default).  sigp_profile_get drains finished events. */
int sigp_profile(sigp_handle* h, int enable);
int sigp_profile_get(sigp_handle* h, int kclass, double* total_ms, int64_t* launches, double* flops,
                     double* bytes);
int sigp_profile_reset(sigp_handle* h);
/* Wait for everything the handle's device has been given (hipDeviceSynchronize on the handle's device).  Every entry point above is
 * synchronous on return already; a caller that brackets a timed region (bench.py) uses this for the bracket itself, so that the
 * measurement runs on the HIP runtime the library links and needs no second one (PyTorch's) in the process. */
int sigp_synchronize(sigp_handle* h);
/* tuning knobs; returns SIGP_BAD_ARG for an unknown name or an invalid value.  Defaults in brackets.
 *   outer_blocks [8]      outer panel width in 128-column blocks (K of the trailing update = 128 x this); left unset, single fits of at most
 *                         24 block columns (n <= 3072) are one panel
 *   lookahead [1]         factor the next panel on the panel stream while the trailing update runs
 *   schedule [0]          0 right-looking outer panels, 1 left-looking (same factor bit for bit)
 *   panel_mode [2]        rows below a panel's top block: 0 recursion, 1 strip solve, 2 strips when strips x members >= strip_min [512]
 *   diag_tiles [1]        symmetric trailing updates: a diagonal 128 x 128 tile loads, multiplies and stores only the 36 of its 64 16 x 16 pairs on or below
 *                         the diagonal (nothing reads the rest; same lower halves bit for bit; 0 = whole tiles, for A/B timing)
 *   ride_tiles [1]        trailing updates: while at most 16 rows of the ride-along block are in use (y and up to 15 test points), its tiles stage and
 *                         multiply those 16 rows only (the other rows are zero rows; same results bit for bit; 0 = whole tiles, for A/B timing)
 *   strip_tri [1]         strip solves skip the zero 16-column x 16-k tile-slices of the inverse diagonal blocks (same bits; 0 = the full products, for A/B timing)
 *   group [8]             fits factorised in lockstep per launch (batch path, fp64 and fp32; 1..256)
 *   small_tile_threshold [320], tiny_tile_threshold [256], trsm128_threshold [256]   tile-shape switches by tile count
 *   refine_iters [3]      fp32 engine: fp64 refinement steps at most; refine_tol_e [12]: stop once every residual is <= 1e-12 (0 = never early)
 *   refine_stored [1]     fp32 engine: the covariance build also writes K~ in fp64 (8 n^2 bytes per lockstep member, skipped above 40 GB) and the
 *                         refinement's residuals read it (HBM-bound) instead of recomputing n^2 covariances each; 0 = recompute (no extra memory)
 *   refine_sym [1]        ... reading the stored lower triangle ONCE per residual (4 n^2 bytes; row and column sums of every tile from LDS), 0 = in two passes
 *   owner_only [0]        sigp_set_train does not allocate the n x n single-GPU matrix (sigp_dist_fit); dist_stats [0] see sigp_dist_fit;
 *   dist_segment [2]      column blocks per streamed broadcast segment of the sharded fit (>= the panel width: panels travel whole)
 *   dist_panel_split [-1] (-1 = by the number of ranks: on from four; 0 / 1)  sharded fit, panel exchange by ROW PIECES ("all-gather of block-row pieces"): the owner factors only the panel's W x W top
 *                         block and broadcasts it (8 MB at W = 8); the rows below it are scattered in `world` pieces, every rank solves its piece, and
 *                         an in-place all-gather assembles the panel on every rank -- the owner's throughput work leaves the chain and each link carries
 *                         1/world of the panel instead of all of it.  Rows are independent: bit-identical to dist_panel_split = 0 wherever both run the same
 *                         tile kernels (every order up to ~ 10 000 at W = 8); beyond, a whole panel's in-panel updates go to the 128-tile kernel and a
 *                         row piece's do not (another k order inside a 16-slice): last-bit differences.
 *   dist_lookahead2d [1]  with the row-split exchange: the NEXT panel's first update is divided by rows as well -- every rank applies it to its own piece
 *                         before solving it; the next owner keeps the update of its top block (from the rows right below the panel's top block, which the
 *                         owner solves itself behind its chain and broadcasts ahead of the pieces) and starts its chain beside the piece solves and the
 *                         all-gather.  A schedule: bit-identical to 0 at every tested shape.
 *   ride_reps [0]         tile pairs per riding workgroup of the diagonal-block launch (0 = one; -1 = by the launch, about four riders per CU; n = fixed):
 *                         shortens that launch in lockstep batches and lengthens the trailing update beside it by as much (docs/EXPERIMENTS.md)
 *   dist_timeout_ms [120000] deadline (time WITHOUT progress of the update stream's panel counter) of every host-side wait of the sharded path; RCCL's asynchronous error state is polled meanwhile.  On an
 *                         error / when it passes: ncclCommAbort, SIGP_HIP_ERROR (the panel reached is in sigp_last_error), the handle's sharded
 *                         state is dead until sigp_dist_shutdown + a fresh sigp_dist_init* -- a dead peer is an error, not a hang; 0 = wait for ever
 *   strips_after_update [1] (look-ahead: the next panel's strip solve -- MFMA work for the whole chip -- waits for the rest of the trailing update; 0 = beside
 *   it: same throughput, a batch is bound by the sum of its MFMA work, but the update's launches then last as long as both kernels' work: bench.py's
 *   roofline.strips_beside_update)
 *   schedules of the latency chain -- bit-identical results, DESIGN.md section 7:  panel_chain [15] (bit 0: panels that are not strip-solved,
 *   bit 1: top blocks of strip-solved panels, are factored column by column with update work riding in the diagonal-block launches;
 *   bit 2: ONE launch between two diagonal blocks (chain_link_kernel: what the next block needs, the column solve riding, every other update
 *   riding in the next diagonal-block launch), bit 3: the binary recursion's two-column leaves take that form too; 0 = binary recursion), chain_rows [80] (bit 0 applies while rows-below x members <= this),
 *   first_on_panel [1] (the update of the next panel's columns on the panel stream: 0 never, 1 for chain-form panels, 2 always)
 *   pan_priority, diag_prio, host_timing   measurement switches
 * The rejected experiments of DESIGN.md section 7 (xcd_chunks, update_wgs / update_late, pipeline_head / head_gate, wide_tiles, n64_tiles,
 * patch, small_nt64, reserve_cus, panel_ll, c_dma, syrk_v2) are compiled into libsigp_debug.so only (make debug; include/sigp_debug.h):
 * the product library answers SIGP_BAD_ARG to their names and carries none of their kernels. */
int sigp_set_option(sigp_handle* h, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SIGP_H */
