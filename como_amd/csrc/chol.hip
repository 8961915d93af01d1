// Dense SPD solve of the window's normal equations: delta = H^-1 g, float64, D ~ 760 .. 2400.
//
// Reference: como/odom/backend/linear_system.py:101-112 (`solve_system`: cholesky_ex(check_errors=False) +
// cholesky_solve).  hipSOLVER needs ~2.5 ms for D = 760 (dozens of tiny launches, not graph-capturable); here the
// factorisation is a right-looking blocked Cholesky with ONE launch per 32-column panel, no host synchronisation:
//
//   chol_pack      : W (Dp x Dp workspace) <- lower(H), with g appended as row D (so the forward substitution
//                    L y = g falls out of the factorisation: row D of L is y^T) and an identity pad up to Dp = 64 * nb.
//   chol_panel(k)  : grid = 1 + #tiles (i,j), k < j <= i.  Every workgroup factors the CBxCB diagonal block A_kk in LDS
//                    (redundantly -- the CUs would idle otherwise), solves its two panel blocks L_ik, L_jk against it
//                    and updates ITS trailing tile A_ij -= L_ik L_jk^T.  Workgroup 0 stores L_kk, diagonal-tile
//                    workgroups store L_ik.  A non-positive pivot is reported in `info` (1-based, first failure)
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

// ---- register-resident 32x32 tile kernels: lane r of wave 0 owns row r; cross-lane operands come from
// v_readlane (compile-time lane index), every loop is unrolled at compile time -> no LDS / memory latency on the
// serial pivot chain.
__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

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

// Right-looking Cholesky of the tile held as rows in a[CB] (lanes 0..CB-1).  On exit lane r holds L[r][0..r].
__device__ __forceinline__ void factor_rows(double (&a)[CB], int lane, int row0, int D, int* __restrict__ info, bool report) {
  sfor<CB>([&](auto ic) {
    constexpr int c = decltype(ic)::value;
    double d = readlane_d(a[c], c);
    if (!(d > 0.0)) {
      if (report && lane == 0 && row0 + c < D) atomicCAS(info, 0, row0 + c + 1);
      d = 1.0;
    }
    // 1/sqrt(d) from the hardware estimate + two Newton steps (full double accuracy, ~12 instructions instead of
    // the ~70 of an IEEE sqrt followed by an IEEE divide -- this chain is the serial critical path of the solve)
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    const double ld = d * inv;
    a[c] = (lane == c) ? ld : a[c] * inv;
    sfor<CB - 1 - c>([&](auto jc) {
      constexpr int cc = c + 1 + decltype(jc)::value;
      a[cc] -= a[c] * readlane_d(a[c], cc);          // A[r][cc] -= L[r][c] L[cc][c]
    });
  });
}

// x <- x L^-T (row per lane, all 64 lanes: two 32-row blocks at once); L rows live in a[] of lanes 0..CB-1.
__device__ __forceinline__ void trsm_rows(double (&x)[CB], const double (&a)[CB]) {
  sfor<CB>([&](auto ic) {
    constexpr int c = decltype(ic)::value;
    sfor<c>([&](auto jt) {
      constexpr int t = decltype(jt)::value;
      x[c] -= x[t] * readlane_d(a[t], c);            // L[c][t] sits in lane c
    });
    x[c] *= fast_rcp(readlane_d(a[c], c));
  });
}

// inverse of the lower-triangular tile: lane j computes column j of L^-1 (stored as row j of inv^T): forward substitution
__device__ __forceinline__ void invert_rows(double (&v)[CB], const double (&a)[CB], int lane) {
  // v[i] = (L^-1)[i][lane]
  sfor<CB>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    double s = (lane == i) ? 1.0 : 0.0;
    sfor<i>([&](auto jt) {
      constexpr int t = decltype(jt)::value;
      s -= readlane_d(a[t], i) * v[t];               // L[i][t] (lane i) * inv[t][lane]
    });
    v[i] = s * fast_rcp(readlane_d(a[i], i));
  });
}

// W: working copy (trailing tiles updated in place); Lw: the factor (blocks go to a SEPARATE matrix because other
// workgroups of the same launch still read the un-factored panel blocks A_ik / A_kk from W); Iw: inverses of the
// diagonal blocks of L (nb x CB x CB, [k][i][j] = (L_kk^-1)[i][j]) for the back-substitution.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ W, double* __restrict__ Lw,
                                                         double* __restrict__ Iw, int Dp, int D, int k, int nb,
                                                         int* __restrict__ info) {
  __shared__ double sI[CB * CLD];      // L_ik
  __shared__ double sJ[CB * CLD];      // L_jk
  const int tid = threadIdx.x;
  int ti = -1, tj = -1;
  if (blockIdx.x > 0) {
    int t = blockIdx.x - 1;
    for (int i = k + 1; i < nb; ++i) {
      const int cnt = i - k;                 // j = k+1 .. i
      if (t < cnt) { ti = i; tj = k + 1 + t; break; }
      t -= cnt;
    }
  }
  const long kk = (long)k * CB;
  if (tid < 64) {                            // wave 0: factor + triangular solves in registers
    const int lane = tid;
    double a[CB], x[CB];
    {
      const int r = lane & (CB - 1);
      const double* src = W + (kk + r) * Dp + kk;
#pragma unroll
      for (int c = 0; c < CB; ++c) a[c] = src[c];
    }
    if (ti >= 0) {
      const int r = lane & (CB - 1);
      const double* src = W + ((long)((lane < CB) ? ti : tj) * CB + r) * Dp + kk;
#pragma unroll
      for (int c = 0; c < CB; ++c) x[c] = src[c];
    }
    factor_rows(a, lane, (int)kk, D, info, blockIdx.x == 0);
    if (blockIdx.x == 0) {
      if (lane < CB) {
        double* dst = Lw + (kk + lane) * Dp + kk;
#pragma unroll
        for (int c = 0; c < CB; ++c)
          if (c <= lane) dst[c] = a[c];
      }
      double v[CB];
      invert_rows(v, a, lane);
      if (lane < CB) {
#pragma unroll
        for (int i = 0; i < CB; ++i) Iw[((long)k * CB + i) * CB + lane] = (i >= lane) ? v[i] : 0.0;
      }
    } else {
      trsm_rows(x, a);
      double* dstS = (lane < CB) ? sI : sJ;
      const int r = lane & (CB - 1);
#pragma unroll
      for (int c = 0; c < CB; ++c) dstS[r * CLD + c] = x[c];
      if (tj == ti && lane < CB) {           // the diagonal-tile workgroup of block-row i publishes L_ik
        double* dst = Lw + ((long)ti * CB + r) * Dp + kk;
#pragma unroll
        for (int c = 0; c < CB; ++c) dst[c] = x[c];
      }
    }
  }
  if (blockIdx.x == 0) return;
  __syncthreads();
  // A_ij -= L_ik L_jk^T : 16x16 threads, (CB/16)^2 outputs each
  const int tx = tid & 15, ty = tid >> 4;
  constexpr int RT = CB / 16;
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
      if (ti != tj || c <= r) W[((long)ti * CB + r) * Dp + (long)tj * CB + c] -= acc[a][b];
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
  for (int k = 0; k < nb; ++k) {
    const int r = nb - 1 - k;
    const int tiles = r * (r + 1) / 2;
    hipLaunchKernelGGL(chol_panel_kernel, dim3(1 + tiles), dim3(256), 0, s, W, Lw, Iw, Dp, D, k, nb, info);
    COMO_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(1024), 0, s, Lw, Iw, Dp, D, nb, delta);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
