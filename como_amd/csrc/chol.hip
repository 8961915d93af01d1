// Dense SPD solve of the window's normal equations: delta = H^-1 g, float64, D ~ 760 .. 2400.
//
// Reference: como/odom/backend/linear_system.py:101-112 (`solve_system`: cholesky_ex(check_errors=False) +
// cholesky_solve).  hipSOLVER needs ~2.5 ms for D = 760 (dozens of tiny launches, not graph-capturable); here the
// factorisation is a right-looking blocked Cholesky with ONE launch per 32-column panel, no host synchronisation:
//
//   chol_pack      : W (Dp x Dp workspace) <- lower(H), with g appended as row D (so the forward substitution
//                    L y = g falls out of the factorisation: row D of L is y^T) and an identity pad up to Dp = 64 * nb.
//   chol_first / chol_panel(k): look-ahead panels (see the kernels): one launch per 32-column panel, the panel solve is
//                    a GEMM against the pre-inverted diagonal block, the next diagonal block is factored and inverted
//                    by the workgroup that just updated it.  A non-positive pivot is reported in `info` (1-based, first failure)
//                    instead of being swallowed; the factorisation then continues with pivot 1 as a defined value.
//   chol_backsub   : one workgroup, L^T delta = y, right-to-left over the 64-blocks.
#include "common.cuh"
#include "../../include/como_hip.h"
#include <type_traits>
#include <utility>

namespace como {

constexpr int CB = 32;        // panel / tile width (in-tile factor/solve latency grows as CB^2 per panel: 32 beats 64)
constexpr int TPR = 256 / CB; // threads per row in the tile triangular solve
constexpr int CLD = CB + 1;   // padded LDS leading dimension

__global__ __launch_bounds__(256) void chol_pack_kernel(const double* __restrict__ H, const double* __restrict__ g,
                                                        double* __restrict__ W, int D, int Dp, int* __restrict__ info) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) *info = 0;
  if (idx >= (long)Dp * Dp) return;
  const int i = (int)(idx / Dp), j = (int)(idx % Dp);
  double v = 0.0;
  if (i < D && j < D) v = (j <= i) ? H[(long)i * D + j] : 0.0;
  else if (i == D && j < D) v = g[j];
  else if (i == D && j == D) v = 1e300;          // pivot of the appended row: irrelevant, just positive
  else if (i == j) v = 1.0;                      // identity pad
  W[idx] = v;
}

// ---- register-resident 32x32 tile kernels: lane r of wave 0 owns row r, every loop is unrolled at compile time.
// reciprocal: hardware estimate + two Newton steps (double accuracy)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = r * (2.0 - d * r);
  r = r * (2.0 - d * r);
  return r;
}

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// Factor the CBxCB tile held by 256 threads (thread (ty, tx) owns elements (ty+16a, tx+16b), lower part valid) and invert
// the factor, fused into ONE rolled loop with a single barrier per pivot.  (A fully unrolled single-wave register
// version was instruction-fetch bound: ~40 KB of straight-line code executed once per launch.)
//   step c, publish : owners of column c store the raw column; owners of inverse row c-1 store X[c-1][:]
//   barrier
//   step c, apply   : A[r][cc] -= A[r][c] A[cc][c] / d  (cc > c);   L[:,c] = A[:,c] / sqrt(d) -> Lw
//                     S[i][:] -= L[i][c-1] X[c-1][:]     (i > c-1)   (forward substitution, one pivot behind)
// sC: 3 x CB doubles (column ring: read during two steps), sX: 2 x CB doubles; oL, oX: CB x CLD staging tiles (global
// stores inside the loop would stall every barrier on vmcnt(0)).
__device__ __forceinline__ void factor_invert_tile(double (&v)[2][2], double* sC, double* sX, double* oL, double* oX,
                                                   double* __restrict__ Lw,
                                                   double* __restrict__ Iw, int Dp, int k, int D, int* __restrict__ info) {
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const long kk = (long)k * CB;
  double S[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) S[a][b] = (ty + 16 * a == tx + 16 * b) ? 1.0 : 0.0;
  double isq_prev = 0.0, pend0 = 0.0, pend1 = 0.0;
  int ring = 0;                                        // c % 3
  // every step: all LDS stores, ONE barrier, all LDS loads in one batch, then arithmetic only (selects, no branches):
  // each extra store->load ordering inside a step costs a full LDS round trip on the serial chain
  for (int c = 0; c <= CB; ++c) {
    double* col = sC + ring * CB;
    const double* colp = sC + (ring == 0 ? 2 : ring - 1) * CB;
    double* xr = sX + (c & 1) * CB;
    const int t = c - 1;
    if (c < CB && tx == (c & 15)) {
      const bool hi = c >= 16;
      col[ty] = hi ? v[0][1] : v[0][0];
      col[ty + 16] = hi ? v[1][1] : v[1][0];
    }
    if (c > 0) {
      if (tx == (t & 15)) {                            // column t of L, finished in the previous step
        oL[ty * CLD + t] = pend0;
        oL[(ty + 16) * CLD + t] = pend1;
      }
      if (ty == (t & 15)) {                            // row t of the inverse
        const bool hi = t >= 16;
        const double x0 = (hi ? S[1][0] : S[0][0]) * isq_prev, x1 = (hi ? S[1][1] : S[0][1]) * isq_prev;
        xr[tx] = x0;
        xr[tx + 16] = x1;
        oX[t * CLD + tx] = (tx <= t) ? x0 : 0.0;
        oX[t * CLD + tx + 16] = (tx + 16 <= t) ? x1 : 0.0;
      }
    }
    __syncthreads();
    const int cs = (c < CB) ? c : CB - 1;
    double d = col[cs];
    const double c0 = col[tx], c1 = col[tx + 16], r0 = col[ty], r1 = col[ty + 16];
    double x0 = xr[tx], x1 = xr[tx + 16];
    const double p0 = colp[ty], p1 = colp[ty + 16];
    if (c == 0) { x0 = 0.0; x1 = 0.0; }                 // nothing published yet (0 * garbage would poison S)
    if (c < CB && !(d > 0.0)) {
      if (tid == 0 && kk + c < D) atomicCAS(info, 0, (int)kk + c + 1);
      d = 1.0;
    }
    const double rinv = fast_rcp(d);
    double isq = __builtin_amdgcn_rsq(d);
    isq = isq * (1.5 - 0.5 * d * isq * isq);
    isq = isq * (1.5 - 0.5 * d * isq * isq);
    const bool live = c < CB;
    const double w0 = r0 * rinv, w1 = r1 * rinv;
    const double m0 = (live && tx > c) ? c0 : 0.0, m1 = (live && tx + 16 > c) ? c1 : 0.0;
    v[0][0] -= w0 * m0;
    v[0][1] -= w0 * m1;
    v[1][0] -= w1 * m0;
    v[1][1] -= w1 * m1;
    pend0 = (ty == c) ? d * isq : r0 * isq;
    pend1 = (ty + 16 == c) ? d * isq : r1 * isq;
    const double l0 = (c > 0 && ty > t) ? p0 * isq_prev : 0.0, l1 = (c > 0 && ty + 16 > t) ? p1 * isq_prev : 0.0;
    S[0][0] -= l0 * x0;
    S[0][1] -= l0 * x1;
    S[1][0] -= l1 * x0;
    S[1][1] -= l1 * x1;
    isq_prev = isq;
    ring = (ring == 2) ? 0 : ring + 1;
  }
  __syncthreads();
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e / CB, c = e % CB;
    if (c <= r) Lw[(kk + r) * Dp + kk + c] = oL[r * CLD + c];
    Iw[(long)k * CB * CB + e] = oX[r * CLD + c];
  }
}

// Look-ahead blocked Cholesky.  chol_first: L_00, L_00^-1.  chol_panel(k), k = 0..nb-2, one launch each, one workgroup
// per trailing tile (i, j), k < j <= i:  L_ik = A_ik L_kk^-T as a 32^3 GEMM against the PRE-INVERTED diagonal block
// (no serial triangular solve), A_ij -= L_ik L_jk^T, and the workgroup that owns tile (k+1, k+1) immediately factors and
// inverts it for the next launch -- the only serial work left on the critical path of a panel.
// W: working copy (trailing tiles updated in place); Lw: the factor (a SEPARATE matrix: other workgroups of the launch
// still read the un-factored panel blocks from W); Iw: inverses of the diagonal blocks of L (nb x CB x CB).
__global__ __launch_bounds__(256) void chol_first_kernel(const double* __restrict__ W, double* __restrict__ Lw,
                                                         double* __restrict__ Iw, int Dp, int D, int* __restrict__ info) {
  __shared__ double sC[3 * CB];
  __shared__ double sX[2 * CB];
  __shared__ double oL[CB * CLD];
  __shared__ double oX[CB * CLD];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double v[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = ty + 16 * a, c = tx + 16 * b;
      v[a][b] = (c <= r) ? W[(long)r * Dp + c] : 0.0;
    }
  factor_invert_tile(v, sC, sX, oL, oX, Lw, Iw, Dp, 0, D, info);
}

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ W, double* __restrict__ Lw,
                                                         double* __restrict__ Iw, int Dp, int D, int k, int nb,
                                                         int* __restrict__ info) {
  __shared__ double sV[CB * CLD];      // L_kk^-1
  __shared__ double sA[CB * CLD];      // A_ik
  __shared__ double sB[CB * CLD];      // A_jk
  __shared__ double sI[CB * CLD];      // L_ik (later: the rings of factor_invert_tile)
  __shared__ double sJ[CB * CLD];      // L_jk
  const int tid = threadIdx.x;
  int ti = -1, tj = -1;
  {
    int t = blockIdx.x;
    for (int i = k + 1; i < nb; ++i) {
      const int cnt = i - k;                 // j = k+1 .. i
      if (t < cnt) { ti = i; tj = k + 1 + t; break; }
      t -= cnt;
    }
  }
  const long kk = (long)k * CB;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e / CB, c = e % CB;
    sV[r * CLD + c] = Iw[((long)k * CB + r) * CB + c];
    sA[r * CLD + c] = W[((long)ti * CB + r) * Dp + kk + c];
    sB[r * CLD + c] = W[((long)tj * CB + r) * Dp + kk + c];
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  constexpr int RT = CB / 16;
  // L_ik[r][c] = sum_t A_ik[r][t] Linv[c][t]   (and the same for j)
  {
    double ai[RT][RT], aj[RT][RT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) { ai[a][b] = 0.0; aj[a][b] = 0.0; }
#pragma unroll 8
    for (int t = 0; t < CB; ++t) {
      double xi[RT], xj[RT], vv[RT];
#pragma unroll
      for (int a = 0; a < RT; ++a) { xi[a] = sA[(ty + 16 * a) * CLD + t]; xj[a] = sB[(ty + 16 * a) * CLD + t]; vv[a] = sV[(tx + 16 * a) * CLD + t]; }
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) { ai[a][b] += xi[a] * vv[b]; aj[a][b] += xj[a] * vv[b]; }
    }
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) {
        sI[(ty + 16 * a) * CLD + tx + 16 * b] = ai[a][b];
        sJ[(ty + 16 * a) * CLD + tx + 16 * b] = aj[a][b];
      }
  }
  __syncthreads();
  if (tj == ti) {   // the diagonal-tile workgroup of block-row i publishes L_ik
    for (int e = tid; e < CB * CB; e += 256) {
      const int r = e / CB, c = e % CB;
      Lw[((long)ti * CB + r) * Dp + kk + c] = sI[r * CLD + c];
    }
  }
  // A_ij -= L_ik L_jk^T : 16x16 threads, (CB/16)^2 outputs each
  double acc[RT][RT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b) acc[a][b] = 0.0;
#pragma unroll 8
  for (int t = 0; t < CB; ++t) {
    double xi[RT], xj[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) { xi[a] = sI[(ty + 16 * a) * CLD + t]; xj[a] = sJ[(tx + 16 * a) * CLD + t]; }
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) acc[a][b] += xi[a] * xj[b];
  }
  const bool next_diag = (ti == k + 1) && (tj == k + 1);
  if (!next_diag) {
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        if (ti != tj || c <= r) W[((long)ti * CB + r) * Dp + (long)tj * CB + c] -= acc[a][b];
      }
  } else {
    // the updated diagonal tile stays in registers and is factored + inverted right here for the next launch
    double v[2][2];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        v[a][b] = (c <= r) ? W[((long)ti * CB + r) * Dp + (long)tj * CB + c] - acc[a][b] : 0.0;
      }
    __syncthreads();                                   // sA / sB / sI / sJ are dead: rings in sI, staging tiles in sA / sB
    factor_invert_tile(v, sI, sI + 3 * CB, sA, sB, Lw, Iw, Dp, k + 1, D, info);
  }
}

// L^T delta = y with y = row D of L (columns 0..D-1).  One workgroup of 1024 threads.
// Per 32-block (right to left): x_k = L_kk^-T y_k is a 32x32 mat-vec with the pre-inverted diagonal block, then
// y_j -= L[k rows, j]^T x_k for the columns to the left (coalesced along j).
__global__ __launch_bounds__(1024) void chol_backsub_kernel(const double* __restrict__ Lw, const double* __restrict__ Iw,
                                                            int Dp, int D, int nb, double* __restrict__ delta) {
  __shared__ double y[4096];
  __shared__ double xb[CB];
  const int tid = threadIdx.x;
  for (int j = tid; j < Dp; j += 1024) y[j] = (j < D) ? Lw[(long)D * Dp + j] : 0.0;
  __syncthreads();
  for (int k = nb - 1; k >= 0; --k) {
    const long kk = (long)k * CB;
    if (tid < CB) {                                       // x_c = sum_r inv[r][c] y_r
      double s = 0.0;
      const double* inv = Iw + (long)k * CB * CB;
#pragma unroll 8
      for (int r = 0; r < CB; ++r) s += inv[r * CB + tid] * y[kk + r];
      if (kk + tid >= D) s = 0.0;                         // appended row / pad rows carry no unknowns
      xb[tid] = s;
      if (kk + tid < D) delta[kk + tid] = s;
    }
    __syncthreads();
    for (int j = tid; j < kk; j += 1024) {
      double v[CB];
#pragma unroll
      for (int r = 0; r < CB; ++r) v[r] = Lw[(kk + r) * Dp + j];
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < CB; ++r) s += v[r] * xb[r];
      y[j] -= s;
    }
    __syncthreads();
  }
}

}  // namespace como

extern "C" {

long como_chol_workspace_bytes(int D) {
  const long nb = (D + 1 + como::CB - 1) / como::CB;
  const long Dp = nb * como::CB;
  return (2 * Dp * Dp + nb * como::CB * como::CB) * (long)sizeof(double);
}

int como_chol_solve_f64(const double* H, const double* g, double* delta, void* workspace, int D, int* info,
                        como_stream_t stream) {
  using namespace como;
  if (!H || !g || !delta || !workspace || !info || D <= 0 || D > 4000) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nb = (D + 1 + CB - 1) / CB;
  const int Dp = nb * CB;
  double* W = (double*)workspace;
  const long tot = (long)Dp * Dp;
  double* Lw = W + tot;
  double* Iw = Lw + tot;
  hipLaunchKernelGGL(chol_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, H, g, W, D, Dp, info);
  COMO_CHECK_LAUNCH();
  if (Dp > 4096) return COMO_ERR_ARG;
  hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(256), 0, s, W, Lw, Iw, Dp, D, info);
  COMO_CHECK_LAUNCH();
  for (int k = 0; k + 1 < nb; ++k) {
    const int r = nb - 1 - k;
    const int tiles = r * (r + 1) / 2;
    hipLaunchKernelGGL(chol_panel_kernel, dim3(tiles), dim3(256), 0, s, W, Lw, Iw, Dp, D, k, nb, info);
    COMO_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(1024), 0, s, Lw, Iw, Dp, D, nb, delta);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
