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

}  // namespace sigp
