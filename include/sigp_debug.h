/*
 * sigp_debug.h -- micro-benchmark / diagnostic entry points of libsigp.so (NOT part of the drop-in ABI of sigp.h).
 * They exist so that single kernels can be timed and ablated on the GPU box from tools/*.py; nothing in
 * seaiceextentforecasting_amd/ binds them.
 */
#ifndef SIGP_DEBUG_H
#define SIGP_DEBUG_H
#include "sigp.h"
#ifdef __cplusplus
extern "C" {
#endif
/* wall-clock phase stamps of one potrf_diag_kernel launch (tools/diag_phases.py) */
int sigp_debug_diag_stamps(sigp_handle* h, const double* A128, double* out_us, int nout);
/* time potrf_diag_kernel on one 128x128 SPD block; `skip` bit mask switches phases off (tools/diag_bench.py) */
int sigp_debug_time_diag(sigp_handle* h, const double* A128, int skip, int reps, double* ms_avg, double* L_out, double* Linv_out);
/* time the covariance build for nb lockstep members of order n; flags: 2 no covariance function, 4 no store (tools/kbuild_bench.py) */
int sigp_debug_time_kbuild(sigp_handle* h, int n, int d, int nb, int kernel_id, int flags, int reps, double* ms_avg);
/* sustained v_mfma_f64_16x16x4_f64 rate; seed < 0: pseudo-random operands (tools/mfma_peak.py) */
int sigp_debug_mfma_peak(sigp_handle* h, int blocks, int iters, double* tflops, double seed);
/* time the lower-tile update on a synthetic panel: rt row tiles, depth K; small = 0 generic 128-tile kernel,
 * 1 generic 64-tile kernel, 2 syrk128_kernel; dbg = ablation bits; clock_ghz = in-kernel clock (tools/syrk_bench.py) */
int sigp_debug_time_syrk(sigp_handle* h, int rt, int K, int patch, int small, int reps, double* ms_avg, double* tflops, int dbg,
                         double* clock_ghz);
/* do small kernels on different streams overlap? (tools/stream_conc.py) */
int sigp_debug_stream_concurrency(sigp_handle* h, int nstreams, int reps, int blocks, int iters, int use_slot_streams, double* ms_out);
/* which CUs does a hipExtStreamCreateWithCUMask mask enable? out2[2*b] = XCC_ID, out2[2*b+1] = HW_ID of block b (tools/cumask_probe.py) */
int sigp_debug_cumask_probe(sigp_handle* h, const unsigned* mask8, int blocks, unsigned* out2);
#ifdef __cplusplus
}
#endif
#endif
