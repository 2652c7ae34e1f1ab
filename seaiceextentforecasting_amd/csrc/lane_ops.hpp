// Cross-lane helpers shared by the diagonal-block kernels (potrf_diag.hpp) and the refinement's residual kernel (kernels_misc.hpp).
#pragma once
#include <hip/hip_runtime.h>

namespace sigp {

// value of lane l (wave-uniform l) of a register
__device__ inline double readlane_t(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ inline float readlane_t(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }

// ---- DPP row broadcasts (gfx90a+: row_newbcast:C = lane C of every 16-lane row to the whole row; the only DPP form the fp64 ALU takes).
// A wave is four rows of 16 lanes; these move nothing through SGPRs (v_readlane + a dependent VALU read of the scalar pair is three
// issue slots per fp64 value, the fused forms below are one).  The assembler does not see into inline asm, so the hazard
// "VALU writes a VGPR, a DPP op reads it: 2 wait states" is the caller's: a freshly written source goes through dpp_ready() (or
// bcast16, which waits itself) before its first DPP read.
template <int C> __device__ __forceinline__ double bcast16(double x) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
  return r;
}
template <int C> __device__ __forceinline__ float bcast16(float x) {
  float r;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
  return r;
}
// acc += (lane C of src's row) * t
template <int C> __device__ __forceinline__ void fmac_bcast16(double& acc, double src, double t) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(t), "n"(C));
}
template <int C> __device__ __forceinline__ void fmac_bcast16(float& acc, float src, float t) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(t), "n"(C));
}
// lane C of x's row, x already past its DPP wait states (dpp_ready).  (v_mul_f64 / v_add_f64 are VOP3-only: they have no DPP form; the
// fp64 DPP instructions are v_mov_b64, v_fmac_f64 and the other VOP1 / VOP2 encodings)
template <int C> __device__ __forceinline__ double bcast16_ready(double x) {
  double r;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
  return r;
}
template <int C> __device__ __forceinline__ float bcast16_ready(float x) {
  float r;
  asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
  return r;
}
// ---- the same operations as ORDERED statements (asm volatile: the compiler keeps their relative order and does not schedule across
// them) for hand-scheduled latency chains: potrf_diag.hpp places the independent fused multiply-adds of one pivot into the latency
// shadows of the next pivot's dependent reciprocal chain itself -- a wave issues in order, and the machine scheduler has no latency
// model for inline asm.
template <int C> __device__ __forceinline__ void o_fmac_bcast16(double& acc, double src, double t) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(t), "n"(C));
}
template <int C> __device__ __forceinline__ void o_fmac_bcast16(float& acc, float src, float t) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(t), "n"(C));
}
template <int C> __device__ __forceinline__ void o_bcast16(double& r, double x) {       // (waits out the DPP hazard itself)
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
}
template <int C> __device__ __forceinline__ void o_bcast16(float& r, float x) {
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(C));
}
__device__ __forceinline__ void o_rcp(double& r, double x) { asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(x)); }
__device__ __forceinline__ void o_rcp(float& r, float x) { asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x)); }
// e = 1 - d x;   x += x e  (one Newton step of the reciprocal in two statements)
__device__ __forceinline__ void o_nr_err(double& e, double d, double x) { asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(d), "v"(x)); }
__device__ __forceinline__ void o_nr_err(float& e, float d, float x) { asm volatile("v_fma_f32 %0, -%1, %2, 1.0" : "=v"(e) : "v"(d), "v"(x)); }
__device__ __forceinline__ void o_nr_upd(double& x, double e) { asm volatile("v_fmac_f64 %0, %0, %1" : "+v"(x) : "v"(e)); }
__device__ __forceinline__ void o_nr_upd(float& x, float e) { asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(x) : "v"(e)); }
// r = -(a x)
__device__ __forceinline__ void o_mul_neg(double& r, double a, double x) { asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(x)); }
__device__ __forceinline__ void o_mul_neg(float& r, float a, float x) { asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(x)); }

__device__ __forceinline__ double dpp_ready(double x) { asm("s_nop 1" : "+v"(x)); return x; }
__device__ __forceinline__ float dpp_ready(float x) { asm("s_nop 1" : "+v"(x)); return x; }

}  // namespace sigp
