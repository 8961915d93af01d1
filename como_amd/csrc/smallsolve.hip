// Conditioning of the small SPD systems of the DepthCov path, batched, one workgroup per matrix, the matrix resident in LDS.
//
// north_star: "covariance-kernel matrix assembly AND CONDITIONING" hand-written.  The reference conditions with
// torch.linalg.cholesky(_ex) + torch.cholesky_solve / solve_triangular on m x m (m <= 64) systems:
//   K_mm + 1e-6 I -> L_mm, K_mm^-1                 como/odom/Mapping.py:450-458 (prep_predictor)
//   normal equations of the depth distillation      como/depth_cov/core/distill_depth.py:31-36, utils/lin_alg.py:82-87
//   get_predictor                                   como/depth_cov/core/distill_depth.py:30-48
//   the sampler's initial factor and obs_info       como/depth_cov/core/samplers.py:110-165 (precalc_entropy_vars)
//   the (6 + m) two-frame system                    como/odom/frontend/two_frame_sfm.py:288-293
//   L_mm^-1 of the sparse log-depth prior           como/odom/frontend/two_frame_sfm.py:115-125
// These are latency problems (a 64 x 64 factorisation is 90 kFLOP), once per keyframe: the design goal is ONE launch per batch
// with no library round trips (hipSOLVER / MAGMA pick a batched potrf + two trsm launches + workspace queries, ~0.3 ms for
// B = 1), graph-capturable, the same arithmetic in float32 and float64 (the reference's DepthCov sampler runs in float32).
//   como_chol_small_*  : A -> L (optional), A^-1 (optional), A^-1 B (optional), info (cholesky_ex semantics)
//   como_trsm_lower_*  : X = L^-1 B (or L^-T B) for MANY right-hand sides (obs_info: d ~ 49 k columns; K~ = (K_mm^-1 K_mn)^T of
//                        the two-frame initialisation: d = every pixel), one thread per column
#include "common.cuh"
#include "../../include/como_hip.h"
#include <cstdlib>

namespace como {

constexpr int SM_MAXN = 80;       // 6 + 64 two-frame unknowns fit; 80 x 81 doubles = 51.8 KB of LDS
constexpr int SM_RHS = 8;         // right-hand-side columns solved per sweep

// A (n,n) row-major, lower triangle read.  L, Ainv, X optional (nullptr).  rhs (n,k) row-major.
template <typename T>
__global__ __launch_bounds__(256) void chol_small_kernel(const T* __restrict__ A, int n, T* __restrict__ Lout,
                                                         T* __restrict__ Ainv, const T* __restrict__ rhs, int k,
                                                         T* __restrict__ X, int* __restrict__ info) {
  __shared__ T a[SM_MAXN * (SM_MAXN + 1)];
  __shared__ T v[SM_MAXN];
  __shared__ T rb[SM_MAXN * SM_RHS];
  __shared__ int bad;
  const int b = blockIdx.x, tid = threadIdx.x, ld = n + 1;
  const T* Ab = A + (long)b * n * n;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    a[i * ld + j] = (j <= i) ? Ab[e] : T(0);
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  // ---- right-looking Cholesky, one column per step (LAPACK potf2 order: the pivot is a_jj minus the already-applied updates)
  for (int j = 0; j < n; ++j) {
    const T piv = a[j * ld + j];
    if (tid == 0 && !(piv > T(0)) && bad == 0) bad = j + 1;          // cholesky_ex: first non-positive leading minor
    const T d = sqrt(piv);
    __syncthreads();
    for (int i = j + tid; i < n; i += 256) a[i * ld + j] = (i == j) ? d : a[i * ld + j] / d;
    __syncthreads();
    // trailing update of the lower triangle: (i, c) with j < c <= i < n -- 16 x 16 thread tiles (the flat form spent its time in
    // two integer divisions per element: 150 us for the 70 x 70 system of the two-frame initialiser); one update per element
    // either way
    const int r = n - 1 - j;
    for (int ii = tid >> 4; ii < r; ii += 16) {
      const int i = j + 1 + ii;
      const T aij = a[i * ld + j];
      for (int cc = tid & 15; cc <= ii; cc += 16) a[i * ld + j + 1 + cc] -= aij * a[(j + 1 + cc) * ld + j];
    }
    __syncthreads();
  }
  if (tid == 0 && info) info[b] = bad;
  if (Lout) {
    T* Lb = Lout + (long)b * n * n;
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e - i * n;
      Lb[e] = (j <= i) ? a[i * ld + j] : T(0);
    }
  }
  // ---- X = A^-1 B: forward then backward substitution, SM_RHS columns per sweep, row-parallel updates (exactly the
  //      operation order of a column-oriented trsv: y_j fixed, then every later row subtracts l_ij y_j)
  if (rhs && X) {
    for (int c0 = 0; c0 < k; c0 += SM_RHS) {
      const int kc = min(SM_RHS, k - c0);
      for (int e = tid; e < n * kc; e += 256) {
        const int i = e / kc, c = e - i * kc;
        rb[i * SM_RHS + c] = rhs[((long)b * n + i) * k + c0 + c];
      }
      __syncthreads();
      for (int j = 0; j < n; ++j) {                     // L y = b
        if (tid < kc) rb[j * SM_RHS + tid] /= a[j * ld + j];
        __syncthreads();
        for (int i = j + 1 + (tid >> 3); i < n; i += 32) {
          const int c = tid & 7;
          if (c < kc) rb[i * SM_RHS + c] -= a[i * ld + j] * rb[j * SM_RHS + c];
        }
        __syncthreads();
      }
      for (int j = n - 1; j >= 0; --j) {                // L^T x = y
        if (tid < kc) rb[j * SM_RHS + tid] /= a[j * ld + j];
        __syncthreads();
        for (int i = tid >> 3; i < j; i += 32) {
          const int c = tid & 7;
          if (c < kc) rb[i * SM_RHS + c] -= a[j * ld + i] * rb[j * SM_RHS + c];
        }
        __syncthreads();
      }
      for (int e = tid; e < n * kc; e += 256) {
        const int i = e / kc, c = e - i * kc;
        X[((long)b * n + i) * k + c0 + c] = rb[i * SM_RHS + c];
      }
      __syncthreads();
    }
  }
  // ---- A^-1 = L^-T L^-1: L^-1 in place (LAPACK trti2, lower: column j from the already inverted trailing block), then the
  //      symmetric product straight to global memory
  if (Ainv) {
    for (int j = n - 1; j >= 0; --j) {
      const T ajj = T(1) / a[j * ld + j];
      if (tid < n && tid > j) v[tid] = a[tid * ld + j];
      __syncthreads();
      for (int i = j + 1 + tid; i < n; i += 256) {
        T s = T(0);
        for (int c = j + 1; c <= i; ++c) s += a[i * ld + c] * v[c];      // row i of the inverted trailing block . column j of L
        a[i * ld + j] = -s * ajj;
      }
      if (tid == 0) a[j * ld + j] = ajj;
      __syncthreads();
    }
    T* Ib = Ainv + (long)b * n * n;
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e - i * n;
      if (j > i) continue;
      T s = T(0);
      for (int c = i; c < n; ++c) s += a[c * ld + i] * a[c * ld + j];    // (L^-T L^-1)_ij = sum_{c >= max(i,j)} Linv_ci Linv_cj
      Ib[i * n + j] = s;
      Ib[j * n + i] = s;
    }
  }
}

// X (B,n,d) = L^-1 Bm for a lower-triangular L (B,n,n), n <= 64: one thread per right-hand-side column, L in LDS (broadcast
// reads), the column's n values in registers; loads / stores of consecutive columns are coalesced (row-major (n,d)).
template <typename T, bool TRANS>      // TRANS: X = L^-T Bm (backward substitution with the same lower factor)
__global__ __launch_bounds__(256) void trsm_lower_kernel(const T* __restrict__ L, const T* __restrict__ Bm, T* __restrict__ X,
                                                         int n, long d) {
  __shared__ T l[64 * 65];
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < 64 * 65; e += 256) l[e] = T(0);
  __syncthreads();
  for (int e = threadIdx.x; e < n * n; e += 256) l[(e / n) * 65 + (e % n)] = L[(long)b * n * n + e];
  __syncthreads();
  const long col = (long)blockIdx.x * 256 + threadIdx.x;
  if (col >= d) return;
  const T* Bc = Bm + (long)b * n * d + col;
  T* Xc = X + (long)b * n * d + col;
  T y[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) y[i] = (i < n) ? Bc[(long)i * d] : T(0);
  if constexpr (!TRANS) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i < n) {
        T s = y[i];
#pragma unroll
        for (int c = 0; c < i; ++c) s -= l[i * 65 + c] * y[c];
        y[i] = s / l[i * 65 + i];
        Xc[(long)i * d] = y[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 63; i >= 0; --i) {
      if (i < n) {
        T s = y[i];
#pragma unroll
        for (int c = i + 1; c < 64; ++c) s -= l[c * 65 + i] * y[c];      // rows >= n of l are zero, y[c >= n] = 0
        y[i] = s / l[i * 65 + i];
        Xc[(long)i * d] = y[i];
      }
    }
  }
}

// csrc/chol.hip: n <= 64, float64 on the dense solver's tile machinery (matrix-core tile factorisation, ~15 us per system)
int chol_small64_f64(const double* A, int B, int n, double* L, double* Ainv, const double* rhs, int k, double* X, int* info,
                     hipStream_t s);

template <typename T>
int chol_small(const T* A, int B, int n, T* L, T* Ainv, const T* rhs, int k, T* X, int* info, hipStream_t s) {
  if (!A || B < 0 || n <= 0 || n > SM_MAXN || k < 0 || ((rhs != nullptr) != (X != nullptr)) || (rhs && k <= 0)) return COMO_ERR_ARG;
  if (B == 0) return COMO_OK;
  if constexpr (sizeof(T) == 8) {
    static const bool fast = [] { const char* e = getenv("COMO_CHOL_SMALL_FAST"); return !e || e[0] != '0'; }();
    if (fast && n <= 64 && n >= 8) return chol_small64_f64(A, B, n, L, Ainv, rhs, k, X, info, s);
  }
  hipLaunchKernelGGL(chol_small_kernel<T>, dim3(B), dim3(256), 0, s, A, n, L, Ainv, rhs, k, X, info);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

template <typename T>
int trsm_lower(const T* L, const T* Bm, T* X, int B, int n, long d, int trans, hipStream_t s) {
  if (!L || !Bm || !X || B < 0 || n <= 0 || n > 64 || d < 0) return COMO_ERR_ARG;
  if (B == 0 || d == 0) return COMO_OK;
  if (trans)
    hipLaunchKernelGGL((trsm_lower_kernel<T, true>), dim3((unsigned)((d + 255) / 256), B), dim3(256), 0, s, L, Bm, X, n, d);
  else
    hipLaunchKernelGGL((trsm_lower_kernel<T, false>), dim3((unsigned)((d + 255) / 256), B), dim3(256), 0, s, L, Bm, X, n, d);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_chol_small_f32(const float* A, int B, int n, float* L, float* Ainv, const float* rhs, int k, float* X, int* info,
                        como_stream_t stream) {
  return como::chol_small<float>(A, B, n, L, Ainv, rhs, k, X, info, (hipStream_t)stream);
}
int como_chol_small_f64(const double* A, int B, int n, double* L, double* Ainv, const double* rhs, int k, double* X, int* info,
                        como_stream_t stream) {
  return como::chol_small<double>(A, B, n, L, Ainv, rhs, k, X, info, (hipStream_t)stream);
}
int como_trsm_lower_f32(const float* L, const float* Bm, float* X, int B, int n, long d, int trans, como_stream_t stream) {
  return como::trsm_lower<float>(L, Bm, X, B, n, d, trans, (hipStream_t)stream);
}
int como_trsm_lower_f64(const double* L, const double* Bm, double* X, int B, int n, long d, int trans, como_stream_t stream) {
  return como::trsm_lower<double>(L, Bm, X, B, n, d, trans, (hipStream_t)stream);
}

}  // extern "C"
