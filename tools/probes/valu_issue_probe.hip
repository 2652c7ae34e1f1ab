// What one wavefront gets out of the vector ALU on gfx950: issue interval and dependent latency of the fp64 operations the diagonal-block
// kernel's pivot loop is made of, alone on its SIMD and with a second wave of the workgroup beside it (idle at a barrier, spinning in
// s_sleep, or issuing MFMAs).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_issue_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define REP16(X) X X X X X X X X X X X X X X X X

// mode: 0 independent v_fma_f64, 1 independent v_fmac_f64_dpp, 2 dependent v_fma_f64, 3 dependent v_rcp_f64, 4 v_readlane x2 + fma (old loop),
//       5 independent v_fmac_f64 with distinct sources, 6 dependent v_fmac_f64_dpp chain, 7 independent v_mov_b64_dpp
// other: what waves 1.. do meanwhile: 0 nothing (exit), 1 wait at the barrier, 2 s_sleep spin on an LDS flag, 3 MFMA loop, 4 same VALU stream
template <int MODE>
__device__ __forceinline__ unsigned long long body(double* out) {
  double a0 = 1.0 + threadIdx.x, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8, s = 1.0000001, t = 0.999999;
  double b0 = 1.5, b1 = 2.5, b2 = 3.5, b3 = 4.5, b4 = 5.5, b5 = 6.5, b6 = 7.5, b7 = 8.5;
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(s), "+v"(t));
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define I8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define J8(OP) OP(a0, b0) OP(a1, b1) OP(a2, b2) OP(a3, b3) OP(a4, b4) OP(a5, b5) OP(a6, b6) OP(a7, b7)
#define FMA(x) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "v"(s), "v"(t));
#define FMACD(x) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(s), "v"(t));
#define FMAC2(x, y) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(y), "v"(t));
#define MOVD(x, y) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
  if (MODE == 0) { REP16(I8(FMA)) REP16(I8(FMA)) }
  if (MODE == 1) { REP16(I8(FMACD)) REP16(I8(FMACD)) }
  if (MODE == 5) { REP16(J8(FMAC2)) REP16(J8(FMAC2)) }
  if (MODE == 7) { REP16(J8(MOVD)) REP16(J8(MOVD)) }
#define DEP(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(t));
  if (MODE == 2) { REP16(REP16(DEP(a0))) }
#define RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
  if (MODE == 3) { REP16(REP16(RCP(a0))) }
#define DEPD(x) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(t));
  if (MODE == 6) { REP16(REP16(DEPD(a0))) }
  if (MODE == 4) {
#pragma unroll
    for (int i = 0; i < 256; ++i) {
      int lo = __builtin_amdgcn_readlane(__double2loint(a7), i & 15), hi = __builtin_amdgcn_readlane(__double2hiint(a7), i & 15);
      const double u = __hiloint2double(hi, lo);
      double& acc = (i & 7) == 0 ? a0 : (i & 7) == 1 ? a1 : (i & 7) == 2 ? a2 : (i & 7) == 3 ? a3 : (i & 7) == 4 ? a4 : (i & 7) == 5 ? a5 : a6;
      acc = __builtin_fma(-t, u, acc);
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  return t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(double* out, unsigned long long* cyc, int other, int prio) {
  __shared__ volatile int flag;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  if (wave == 0) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long c = body<MODE>(out);
    if (threadIdx.x == 0) { cyc[0] = c; flag = 1; }
    if (other == 1) __syncthreads();
    return;
  }
  if (other == 0) return;
  if (other == 1) { __syncthreads(); return; }
  if (other == 2) { for (int spin = 0; flag == 0 && spin < (1 << 22); ++spin) __builtin_amdgcn_s_sleep(1); return; }
  if (other == 3) {
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x, b = 1.0;
    for (int it = 0; flag == 0 && it < (1 << 20); ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    out[512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    return;
  }
  if (other == 5 || other == 6) {          // MFMAs only from waves on the OTHER SIMDs (5: wave 4 exits, 6: wave 4 spins in s_sleep)
    if ((wave & 3) == 0) { if (other == 6) for (int spin = 0; flag == 0 && spin < (1 << 22); ++spin) __builtin_amdgcn_s_sleep(1); return; }
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x, b = 1.0;
    for (int it = 0; flag == 0 && it < (1 << 20); ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    out[512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    return;
  }
  if (other == 7) {                        // MFMAs ONLY from wave 4 (the pivot wave's presumed SIMD mate)
    if (wave != 4) return;
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x, b = 1.0;
    for (int it = 0; flag == 0 && it < (1 << 20); ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    out[512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    return;
  }
  if (other == 4) { const unsigned long long c = body<MODE>(out + 1024 * wave); if ((threadIdx.x & 63) == 0) cyc[wave] = c; return; }
}

template <int MODE> void run(const char* name, int ninstr, double* out, unsigned long long* cyc) {
  const char* others[] = {"alone", "7 waves at the barrier", "7 waves in s_sleep spins", "7 waves issuing MFMAs", "8 waves, same stream",
                          "MFMAs on waves 1,2,3,5,6,7", "same, wave 4 in s_sleep", "MFMAs on wave 4 only"};
  for (int threads : {64, 512})
    for (int other = 0; other < 8; ++other) {
      if (threads == 64 && other != 0) continue;
      if (threads == 512 && other == 0) continue;
      for (int prio = 0; prio < 2; ++prio) {
        if (prio && other < 2) continue;
        unsigned long long best = ~0ull, h[8];
        for (int rep = 0; rep < 5; ++rep) {
          hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, other, prio);
          hipDeviceSynchronize();
          hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
          if (h[0] < best) best = h[0];
        }
        printf("%-44s %-26s prio %d: %6llu cycles / %d = %5.2f per instruction\n", name, others[other], prio, best, ninstr, (double)best / ninstr);
      }
    }
}

__global__ void where(unsigned* o) { if ((threadIdx.x & 63) == 0) o[threadIdx.x >> 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }

int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 16384 * sizeof(double)); hipMalloc(&cyc, 64);
  {
    unsigned* o; unsigned h[8];
    hipMalloc(&o, 32);
    hipLaunchKernelGGL(where, dim3(1), dim3(512), 0, 0, o);
    hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) printf("wave %d: HW_ID %08x  wave_id %u simd %u cu %u\n", w, h[w], h[w] & 15, (h[w] >> 4) & 3, (h[w] >> 8) & 15);
  }
  run<0>("v_fma_f64 independent (8 accumulators)", 256, out, cyc);

  run<1>("v_fmac_f64_dpp row_newbcast independent", 256, out, cyc);

  run<2>("v_fma_f64 dependent chain", 256, out, cyc);
  run<6>("v_fmac_f64_dpp dependent chain (+s_nop 1)", 256, out, cyc);
  run<3>("v_rcp_f64 dependent chain", 256, out, cyc);

  return 0;
}
