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

__device__ __forceinline__ double rcp_refined(double d) {      // hardware estimate (24 bits) + one cubic correction
  const double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ double rsq_refined(double d) {      // hardware estimate + two Newton steps
  double r = __builtin_amdgcn_rsq(d);
  r = r * (1.5 - 0.5 * d * r * r);
  r = r * (1.5 - 0.5 * d * r * r);
  return r;
}

// Factor a CBxCB tile and invert the factor with 512 threads in ONE rolled loop, one barrier per pivot.
// Measured cost model on MI355X (scripts/micro): one wave issues ~1 VALU instruction per 6 cycles, an LDS store ->
// barrier -> load hop is ~150 cycles, f64 FMAs are not the problem.  The pivot loop is therefore bound by the
// INSTRUCTIONS each wave executes per pivot, so the work is split by role:
//   threads   0..255 (F): thread (ty, tx) owns A[ty+16a][tx+16b]; step c: load raw column c, 1/d, rank-1 update,
//                         owners of column c+1 store it raw (oC[c+1][:]) for the next step
//   threads 256..511 (I): thread (ty, tx) owns S[ty+16a][tx+16b] (S = I initially); step c handles pivot t = c-1:
//                         load raw column t and raw row t of S, 1/sqrt(d) -> sq[t], S[i][:] -= L[i][t] X[t][:],
//                         owners of row t+1 store it raw (oS[t+1][:])
// Nothing is scaled inside the loop: L[r][c] = oC[c][r] * sq[c] and L^-1[t][j] = oS[t][j] * sq[t] are formed by
// the copy-out.  (Earlier versions: a fully unrolled single-wave register kernel was instruction-fetch bound -- 40 KB
// of straight-line code executed once; 256 threads doing both roles took 1100 cycles per pivot.)
// oC, oS: CB x CB doubles each; sq: CB doubles.
__device__ __forceinline__ void factor_invert_tile(double (&v)[2][2], double* oC, double* oS, double* sq,
                                                   double* __restrict__ Lw, double* __restrict__ Iw, int Dp, int k, int D,
                                                   int* __restrict__ info) {
  const int tid = threadIdx.x, role = tid >> 8, lt = tid & 255, tx = lt & 15, ty = lt >> 4;
  const long kk = (long)k * CB;
  if (role == 0) {
    if (tx == 0) { oC[ty] = v[0][0]; oC[ty + 16] = v[1][0]; }
    for (int c = 0; c <= CB; ++c) {
      __syncthreads();
      if (c == CB) break;
      const double* col = oC + c * CB;
      double d = col[c];
      const double c0 = col[tx], c1 = col[tx + 16], r0 = col[ty], r1 = col[ty + 16];
      if (!(d > 0.0)) {
        if (tid == 0 && kk + c < D) atomicCAS(info, 0, (int)kk + c + 1);
        d = 1.0;
      }
      const double rinv = rcp_refined(d);
      const double w0 = r0 * rinv, w1 = r1 * rinv;
      const double m0 = (tx > c) ? c0 : 0.0, m1 = (tx + 16 > c) ? c1 : 0.0;
      v[0][0] -= w0 * m0;
      v[0][1] -= w0 * m1;
      v[1][0] -= w1 * m0;
      v[1][1] -= w1 * m1;
      if (c + 1 < CB && tx == ((c + 1) & 15)) {
        const bool hi = c + 1 >= 16;
        double* nxt = oC + (c + 1) * CB;
        nxt[ty] = hi ? v[0][1] : v[0][0];
        nxt[ty + 16] = hi ? v[1][1] : v[1][0];
      }
    }
  } else {
    double S[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) S[a][b] = (ty + 16 * a == tx + 16 * b) ? 1.0 : 0.0;
    if (ty == 0) { oS[tx] = S[0][0]; oS[tx + 16] = S[0][1]; }
    for (int c = 0; c <= CB; ++c) {
      __syncthreads();
      if (c == 0) continue;
      const int t = c - 1;
      const double* col = oC + t * CB;
      const double* row = oS + t * CB;
      double d = col[t];
      const double p0 = col[ty], p1 = col[ty + 16], x0 = row[tx], x1 = row[tx + 16];
      if (!(d > 0.0)) d = 1.0;
      const double isq = rsq_refined(d);
      if (lt == 0) sq[t] = isq;
      const double xs0 = x0 * isq, xs1 = x1 * isq;                 // X[t][:]
      const double l0 = (ty > t) ? p0 * isq : 0.0, l1 = (ty + 16 > t) ? p1 * isq : 0.0;
      S[0][0] -= l0 * xs0;
      S[0][1] -= l0 * xs1;
      S[1][0] -= l1 * xs0;
      S[1][1] -= l1 * xs1;
      if (t + 1 < CB && ty == ((t + 1) & 15)) {
        const bool hi = t + 1 >= 16;
        double* nxt = oS + (t + 1) * CB;
        nxt[tx] = hi ? S[1][0] : S[0][0];
        nxt[tx + 16] = hi ? S[1][1] : S[0][1];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < CB * CB; e += 512) {
    const int r = e / CB, c = e % CB;
    if (c <= r) Lw[(kk + r) * Dp + kk + c] = oC[c * CB + r] * sq[c];
    Iw[(long)k * CB * CB + e] = (c <= r) ? oS[e] * sq[r] : 0.0;
  }
}

// Look-ahead blocked Cholesky.  chol_first: L_00, L_00^-1.  chol_panel(k), k = 0..nb-2, one launch each, one workgroup
// per trailing tile (i, j), k < j <= i:  L_ik = A_ik L_kk^-T as a 32^3 GEMM against the PRE-INVERTED diagonal block
// (no serial triangular solve), A_ij -= L_ik L_jk^T, and the workgroup that owns tile (k+1, k+1) immediately factors and
// inverts it for the next launch -- the only serial work left on the critical path of a panel.
// W: working copy (trailing tiles updated in place); Lw: the factor (a SEPARATE matrix: other workgroups of the launch
// still read the un-factored panel blocks from W); Iw: inverses of the diagonal blocks of L (nb x CB x CB).
__global__ __launch_bounds__(512) void chol_first_kernel(const double* __restrict__ W, double* __restrict__ Lw,
                                                         double* __restrict__ Iw, int Dp, int D, int* __restrict__ info) {
  __shared__ double oC[CB * CB];
  __shared__ double oS[CB * CB];
  __shared__ double sq[CB];
  const int tx = threadIdx.x & 15, ty = (threadIdx.x & 255) >> 4;
  double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  if (threadIdx.x < 256) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        v[a][b] = (c <= r) ? W[(long)r * Dp + c] : 0.0;
      }
  }
  factor_invert_tile(v, oC, oS, sq, Lw, Iw, Dp, 0, D, info);
}

__global__ __launch_bounds__(512) void chol_panel_kernel(double* __restrict__ W, double* __restrict__ Lw,
                                                         double* __restrict__ Iw, int Dp, int D, int k, int nb,
                                                         int* __restrict__ info) {
  __shared__ double sV[CB * CLD];      // L_kk^-1
  __shared__ double sA[CB * CLD];      // A_ik   (later: raw columns of the next diagonal tile)
  __shared__ double sB[CB * CLD];      // A_jk   (later: raw rows of its inverse)
  __shared__ double sI[CB * CLD];      // L_ik   (later: 1/sqrt(pivots))
  __shared__ double sJ[CB * CLD];      // L_jk
  const int tid = threadIdx.x, half = tid >> 8, lt = tid & 255;
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
  for (int e = tid; e < CB * CB; e += 512) {
    const int r = e / CB, c = e % CB;
    sV[r * CLD + c] = Iw[((long)k * CB + r) * CB + c];
    sA[r * CLD + c] = W[((long)ti * CB + r) * Dp + kk + c];
    sB[r * CLD + c] = W[((long)tj * CB + r) * Dp + kk + c];
  }
  __syncthreads();
  const int tx = lt & 15, ty = lt >> 4;
  constexpr int RT = CB / 16;
  // L_ik[r][c] = sum_t A_ik[r][t] Linv[c][t]  (threads 0..255), the same for L_jk (threads 256..511)
  {
    const double* src = half ? sB : sA;
    double* dst = half ? sJ : sI;
    double ai[RT][RT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) ai[a][b] = 0.0;
#pragma unroll 8
    for (int t = 0; t < CB; ++t) {
      double xi[RT], vv[RT];
#pragma unroll
      for (int a = 0; a < RT; ++a) { xi[a] = src[(ty + 16 * a) * CLD + t]; vv[a] = sV[(tx + 16 * a) * CLD + t]; }
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) ai[a][b] += xi[a] * vv[b];
    }
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) dst[(ty + 16 * a) * CLD + tx + 16 * b] = ai[a][b];
  }
  __syncthreads();
  if (tj == ti) {   // the diagonal-tile workgroup of block-row i publishes L_ik
    for (int e = tid; e < CB * CB; e += 512) {
      const int r = e / CB, c = e % CB;
      Lw[((long)ti * CB + r) * Dp + kk + c] = sI[r * CLD + c];
    }
  }
  const bool next_diag = (ti == k + 1) && (tj == k + 1);
  // A_ij -= L_ik L_jk^T : 16x16 threads, (CB/16)^2 outputs each (threads 0..255)
  double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  if (half == 0) {
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
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < RT; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        double* w = W + ((long)ti * CB + r) * Dp + (long)tj * CB + c;
        if (next_diag) v[a][b] = (c <= r) ? *w - acc[a][b] : 0.0;   // stays on chip: factored + inverted right below
        else if (ti != tj || c <= r) *w -= acc[a][b];
      }
  }
  if (next_diag) {
    __syncthreads();                                   // sA / sB / sI are dead now
    factor_invert_tile(v, sA, sB, sI, Lw, Iw, Dp, k + 1, D, info);
  }
}

// L^T delta = y with y = row D of L (columns 0..D-1).  One workgroup of 1024 threads.
// Bound by what ONE compute unit can pull through the fabric (~10 B/clk: 2.3 MB of L at D = 760 -> ~65 us; a version
// that prefetched every panel a step ahead into registers measured the same 69 us).  Spreading the panel products over
// many workgroups needs a device-wide hand-off per panel (~4 us each, 24 of them): no better.
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
  hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, info);
  COMO_CHECK_LAUNCH();
  for (int k = 0; k + 1 < nb; ++k) {
    const int r = nb - 1 - k;
    const int tiles = r * (r + 1) / 2;
    hipLaunchKernelGGL(chol_panel_kernel, dim3(tiles), dim3(512), 0, s, W, Lw, Iw, Dp, D, k, nb, info);
    COMO_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(1024), 0, s, Lw, Iw, Dp, D, nb, delta);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
