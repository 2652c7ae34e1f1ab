// The reference's OWN kernel at the reference's OWN size (n <= 128 training years, any number of features), batched:
// one workgroup per fit, everything in LDS.  Replaces, per (region, year, grid point),
//     north/June1st.py:264-277   (expm -> X Sigma~ X^T + sn~ I -> cholesky -> A~ -> sigma_f -> k*, k** -> v -> fmean, fvar)
//     north/June1st.py:246       (nlML)
// for the retro loop (north/retrospective_forecasts/September1st_retro.py:176-248) times the 20 x 20 hyper-parameter
// grid of north/June1st.py:210-211 -- ~5e4 fits of order <= 45 that are launch-latency bound one at a time.
//
// Covariance in factored form.  M is symmetric negative semi-definite, M = Q diag(lam) Q^T, so
//     Sigma~(l) = expm(l M) = Q diag(exp(l lam)) Q^T,        K~ = (X Q) diag(exp(l lam)) (X Q)^T + sn~ I :
// a data set is staged ONCE as A = [X ; Xs] Q (rows 0..n-1 training, rows n.. test points) with lam, and every grid point
// of that data set rebuilds K~ from A with its own weights (one eigendecomposition per data set).
// lam_mode 1: the weights are given directly (w_k = lam_k >= 0): A = [X ; Xs] U, lam = eigenvalues of a host-side
// Sigma~ = U diag(lam) U^T -- this is how a Pade expm(l M) from the host (the reference's own numbers at extreme l,
// SURVEY App. C-11) goes through the same kernel.
//
// Layout in LDS: augmented lower-trapezoid  Kp[(n + 1 + m)][ldk]: rows 0..n-1 = K~ (lower), row n = y, rows n+1.. = k~*_j.
// A right-looking column Cholesky of the bordered matrix leaves  z = L~^-1 y  in row n and  v_j = L~^-1 k~*_j  below it
// (the same ride-along trick as the blocked path), so sigma_f = z.z/n, fmean_j = v_j.z, fvar_j = sigma_f (k~** + sn~ - v_j.v_j).
#pragma once
#include <hip/hip_runtime.h>

namespace sigp {

constexpr int SM_NMAX = 128;   // largest order one workgroup handles
constexpr int SM_MMAX = 8;     // test points per fit

struct SmallSet {
  long a_off;      // doubles into the A pool: A [(n + m)][N] row-major
  long y_off;      // doubles into the y pool: y [n]
  long lam_off;    // doubles into the lam pool: lam [N]
  int n, N, m;
  int lam_mode;    // 0: w_k = exp(ell * lam_k);  1: w_k = max(lam_k, 0)
};
struct SmallProb {
  int set;         // data set of this fit
  int pad;
  double ell;      // length scale (lam_mode 0)
  double sn;       // sigma_n tilde
};

inline long smallgp_lds_bytes(int n, int m, int ch) {
  const long ldk = n | 1;
  return ((long)(n + 1 + m) * ldk + (long)(n + m) * (ch + 1) + ch + 4 * SM_MMAX + 16 + n) * (long)sizeof(double);
}

// sum over the NT threads of the workgroup; result valid on every thread
template <int NT>
__device__ inline double smallgp_allsum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sh[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) t += sh[w];
  return t;
}

// out [nprob][OW]: OW = 4 (GRAD = false): sigma_f, nlML, info (LAPACK pivot index, 0 = ok), sigma_n;  mean / var [nprob][mstride]
// OW = 8 (GRAD = true): the same four, then the reference's MLII "gradient" (north/June1st.py:248-252) and the exact derivative of
// the profiled nlML w.r.t. (log l, log sn~).  In factored form every matrix of :248-252 is a reweighting of the same A:
//     dKdl = X (M Sigma) X^T + sigma_n I = sigma_f (D1 + sn~ I),  D1 = A diag(dw) A^T,  dw_k = lam_k exp(l lam_k)   (M Sigma~ = Q diag(lam w) Q^T)
//     dKds = X Sigma X^T + sigma_f I     = sigma_f (K~ - sn~ I + I),      K^-1 = K~^-1 / sigma_f,  alpha = a~ / sigma_f,
// so with T0 = tr(K~^-1), T1 = tr(K~^-1 D1), q0 = a~.a~, q1 = a~^T D1 a~ :
//     ref   g1 = (T1 + sn~ T0)/2 - (q1 + sn~ q0)/(2 sigma_f)        g2 = (1 - sn~)(T0 - q0/sigma_f)/2      (a~^T y = n sigma_f)
//     exact g1 = l (T1 - q1/sigma_f)/2                              g2 = sn~ (T0 - q0/sigma_f)/2
// T0, T1 come from the explicit inverse factor X = L~^-1 (formed in place over L~, row by row): T0 = |X|_F^2,
// T1 = sum_c dw_c |X a_c|^2 (a_c = column c of A; X a_c is never stored), a~ = X^T z, q1 = sum_c dw_c (a_c.a~)^2.
// lam_mode 1 sets (weights from a host-side Pade expm) take dw from the `dlam` pool (sigp_small_set_dweights); without it their
// gradient entries are NaN.
// NT = threads per workgroup: 256 (four wavefronts per fit) is the default at every order; NT = 64 (one wavefront per fit, every
// barrier of the column loop a wave-local no-op) is kept as a measurement switch ("small_nt64") -- on the reference-size grid it
// is SLOWER (1.71 vs 1.02 ms for 48 000 fits of n = 6 .. 45): the rank-1 updates have ~n^2/2 elements, enough for four waves,
// and one resident wave per fit leaves the LDS pipeline idle between dependent steps.
template <int NT, bool GRAD = false>
__global__ __launch_bounds__(NT) void smallgp_kernel(const SmallSet* __restrict__ sets, const SmallProb* __restrict__ probs,
                                                      const double* __restrict__ Apool, const double* __restrict__ ypool,
                                                      const double* __restrict__ lampool, int ch, double* __restrict__ out,
                                                      double* __restrict__ mean, double* __restrict__ var, int mstride,
                                                      const double* __restrict__ dlampool) {
  constexpr int OW = GRAD ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const SmallProb pb = probs[blockIdx.x];
  const SmallSet st = sets[pb.set];
  const int n = st.n, N = st.N, m = st.m;
  const int R = n + 1 + m;          // rows of the bordered matrix
  const int ldk = n | 1;            // odd pitch: column walks spread over the banks
  const int lda = ch + 1;
  double* Kp = (double*)smem_raw;                 // [R][ldk]
  double* Ac = Kp + (long)R * ldk;                // [n + m][ch + 1]: feature chunk, pre-scaled by sqrt(w_k)
  double* wk = Ac + (long)(n + m) * lda;          // [ch]
  double* kss = wk + ch;                          // [SM_MMAX] k~**_j
  double* red = kss + SM_MMAX;                    // reduction scratch
  __shared__ int s_info;
  constexpr int TY = NT / 16;       // thread rows of the 16-wide element mapping
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const double* A = Apool + st.a_off;
  const double* lam = lampool + st.lam_off;

  for (int e = tid; e < R * ldk; e += NT) Kp[e] = 0.0;
  if (tid < SM_MMAX) kss[tid] = 0.0;
  if (tid == 0) s_info = 0;

  // ---- K~ (lower), k~*, k~** from the factored covariance, ch features at a time --------------------------------
  for (int k0 = 0; k0 < N; k0 += ch) {
    const int kc = min(ch, N - k0);
    __syncthreads();
    if (tid < kc) {
      const double l = lam[k0 + tid];
      wk[tid] = sqrt(st.lam_mode ? fmax(l, 0.0) : exp(pb.ell * l));
    }
    __syncthreads();
    for (int e = tid; e < (n + m) * kc; e += NT) {
      const int r = e / kc, k = e - r * kc;
      Ac[r * lda + k] = A[(long)r * N + k0 + k] * wk[k];
    }
    __syncthreads();
    for (int i = ty; i < n + m; i += TY) {
      const int krow = i < n ? i : i + 1;               // test rows sit below the y row
      const double* ai = Ac + i * lda;
      const int jmax = i < n ? i : n - 1;
      for (int j = tx; j <= jmax; j += 16) {
        const double* aj = Ac + j * lda;
        double acc = 0.0;
        for (int k = 0; k < kc; ++k) acc = fma(ai[k], aj[k], acc);
        Kp[krow * ldk + j] += acc;
      }
    }
    if (tid < m) {
      const double* as = Ac + (n + tid) * lda;
      double acc = 0.0;
      for (int k = 0; k < kc; ++k) acc = fma(as[k], as[k], acc);
      kss[tid] += acc;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    Kp[i * ldk + i] += pb.sn;
    Kp[n * ldk + i] = ypool[st.y_off + i];
  }
  __syncthreads();

  // ---- right-looking Cholesky of the bordered matrix, one column per step --------------------------------------
  for (int j = 0; j < n; ++j) {
    double d = Kp[j * ldk + j];
    if (!(d > 0.0)) {               // non-positive or NaN pivot: LAPACK info = 1-based index of the first one
      if (tid == 0 && s_info == 0) s_info = j + 1;
      d = 1.0;
    }
    const double s = sqrt(d), inv = 1.0 / s;
    __syncthreads();                // everyone has read the pivot
    for (int i = j + tid; i < R; i += NT) Kp[i * ldk + j] = (i == j) ? s : Kp[i * ldk + j] * inv;
    __syncthreads();
    for (int i = j + 1 + ty; i < R; i += TY) {
      const double lij = Kp[i * ldk + j];
      const int cmax = i < n ? i : n - 1;
      for (int c = j + 1 + tx; c <= cmax; c += 16) Kp[i * ldk + c] = fma(-lij, Kp[c * ldk + j], Kp[i * ldk + c]);
    }
    __syncthreads();
  }

  // ---- reductions: sigma_f, nlML, mean, variance (north/June1st.py:267-268, 246, 276-277) ------------------------
  const double* z = Kp + n * ldk;
  double zq = 0.0, ld_ = 0.0;
  for (int i = tid; i < n; i += NT) { zq = fma(z[i], z[i], zq); ld_ += log(Kp[i * ldk + i]); }
  const double zz = smallgp_allsum<NT>(zq, red);
  const double logdet = smallgp_allsum<NT>(ld_, red);
  const int info = s_info;
  const double inf = __builtin_huge_val(), qnan = __builtin_nan("");
  const double sf = zz / (double)n;
  if (tid == 0) {
    double* o = out + (long)blockIdx.x * OW;
    if (info == 0) {
      o[0] = sf;
      o[1] = 0.5 * n + logdet + 0.5 * n * log(sf) + 0.5 * n * log(2.0 * M_PI);
      o[2] = 0.0;
      o[3] = sf * pb.sn;
    } else {
      o[0] = inf; o[1] = inf; o[2] = (double)info; o[3] = inf;
      if (GRAD) { o[4] = inf; o[5] = inf; o[6] = inf; o[7] = inf; }     // north/June1st.py:254-256
    }
  }
  for (int j = 0; j < m; ++j) {
    const double* v = Kp + (n + 1 + j) * ldk;
    double a = 0.0, b = 0.0;
    for (int i = tid; i < n; i += NT) { a = fma(v[i], z[i], a); b = fma(v[i], v[i], b); }
    const double vz = smallgp_allsum<NT>(a, red);
    const double vv = smallgp_allsum<NT>(b, red);
    if (tid == 0) {
      mean[(long)blockIdx.x * mstride + j] = info == 0 ? vz : qnan;
      var[(long)blockIdx.x * mstride + j] = info == 0 ? sf * (kss[j] + pb.sn - vv) : qnan;
    }
  }

  if constexpr (GRAD) {
    if (info != 0) return;           // uniform
    double* at = red + 16;           // [n] a~ = L~^-T z
    // ---- X = L~^-1 in place, row by row: X(i,j) = -X(i,i) sum_{k=j}^{i-1} L(i,k) X(k,j); rows above i are complete, row i of L~ is
    // read by everyone before anyone overwrites it.  Element j of the row belongs to a 16-lane group (k split over its lanes).
    for (int i = 0; i < n; ++i) {
      const double xii = 1.0 / Kp[i * ldk + i];
      constexpr int MC = (SM_NMAX + TY - 1) / TY;
      double mine[MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const int j = ty + c * TY;
        if (j < i) {
          double acc = 0.0;
          for (int k = j + tx; k < i; k += 16) acc = fma(Kp[i * ldk + k], Kp[k * ldk + j], acc);
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
          mine[c] = -xii * acc;
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const int j = ty + c * TY;
        if (j < i && tx == 0) Kp[i * ldk + j] = mine[c];
      }
      if (tid == 0) Kp[i * ldk + i] = xii;
      __syncthreads();
    }
    for (int i = tid; i < n; i += NT) {
      double a = 0.0;
      for (int k = i; k < n; ++k) a = fma(Kp[k * ldk + i], z[k], a);
      at[i] = a;
    }
    double t0 = 0.0, t1 = 0.0, q0 = 0.0, q1 = 0.0;
    for (int e = tid; e < n * n; e += NT) {
      const int i = e / n, j = e - i * n;
      if (j <= i) { const double x = Kp[i * ldk + j]; t0 = fma(x, x, t0); }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) q0 = fma(at[i], at[i], q0);
    const double* dlam = dlampool ? dlampool + st.lam_off : nullptr;
    for (int k0 = 0; k0 < N; k0 += ch) {
      const int kc = min(ch, N - k0);
      __syncthreads();
      if (tid < kc) {
        const double l = lam[k0 + tid];
        wk[tid] = st.lam_mode ? (dlam ? dlam[k0 + tid] : qnan) : l * exp(pb.ell * l);
      }
      for (int e = tid; e < n * kc; e += NT) {
        const int r = e / kc, k = e - r * kc;
        Ac[r * lda + k] = A[(long)r * N + k0 + k];
      }
      __syncthreads();
      for (int e = tid; e < n * kc; e += NT) {
        const int i = e / kc, c = e - i * kc;
        const double* xi = Kp + i * ldk;
        double v = 0.0;
        for (int k = 0; k <= i; ++k) v = fma(xi[k], Ac[k * lda + c], v);
        t1 = fma(wk[c] * v, v, t1);
      }
      if (tid < kc) {
        double b = 0.0;
        for (int i = 0; i < n; ++i) b = fma(Ac[i * lda + tid], at[i], b);
        q1 = fma(wk[tid] * b, b, q1);
      }
    }
    const double T0 = smallgp_allsum<NT>(t0, red);
    const double T1 = smallgp_allsum<NT>(t1, red);
    const double Q0 = smallgp_allsum<NT>(q0, red);
    const double Q1 = smallgp_allsum<NT>(q1, red);
    if (tid == 0) {
      double* o = out + (long)blockIdx.x * OW;
      const double sn = pb.sn;
      o[4] = 0.5 * (T1 + sn * T0) - 0.5 * (Q1 + sn * Q0) / sf;
      o[5] = 0.5 * (1.0 - sn) * (T0 - Q0 / sf);
      o[6] = 0.5 * pb.ell * (T1 - Q1 / sf);
      o[7] = 0.5 * sn * (T0 - Q0 / sf);
    }
  }
}

}  // namespace sigp
