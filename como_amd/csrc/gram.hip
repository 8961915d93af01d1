// Weighted normal equations of a tall least-squares problem: A^T W A and A^T W r for A (n x m), n >> m <= 64, float64.
//
// Reference: como/utils/lin_alg.py:82-87 (`lstsq_chol`: A.mT @ A, A.mT @ b) as used by the depth distillation of a new keyframe
// (como/depth_cov/core/distill_depth.py:52-84, 122-148): A = [prior rows ; stdev_inv_obs * K~ rows of the observed pixels],
// b = stdev_inv_obs * (log z_obs [- K~[:, :m1] log z_1]).  The reference first gathers the valid rows (boolean indexing),
// scales and concatenates them, then forms the m x m products with one library GEMM whose single output tile is walked by ONE
// workgroup over n = 49 k ... 300 k rows (a batched slab GEMM + sum was this repository's round-1 work-around).  Here the rows
// are consumed IN PLACE, one pass at HBM rate:
//   r_i = y_i - sum_k A[i][k] c[k]            (c optional: the part of the right-hand side explained by known columns)
//   AtA = sum_i w_i A_i A_i^T,  Atb = sum_i w_i A_i r_i,  stats = {sum w, sum w r, sum w r^2, #(w != 0)}
// with w_i = (validity mask) x (1 / stdev^2): masked rows contribute exact zeros, no gather, no concatenation; the few prior rows
// are added by the caller (an m x m product).  Lane (c, q) of a wave holds the quad A[row 4 s + q][4 c .. 4 c + 3] of step s
// -- one fully coalesced KiB per wave load -- and feeds element t of it to column block t, exactly the operand layout of the
// window kernels (csrc/ba.hip): ten 16x16 tiles of v_mfma_f64_16x16x4_f64 cover the upper block triangle, K = rows.
// Two kernels: per-workgroup partial sums (fixed row ranges), then the partials are added in a fixed order (four waves per 64
// outputs, gram_finish_kernel) -> the result does not depend on scheduling.
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

typedef double g4_t __attribute__((ext_vector_type(4)));
constexpr int GRAM_REC = 10 * 256 + 64 + 4;       // ten tiles | A^T W r | stats

__global__ __launch_bounds__(256) void gram_partial_kernel(const double* __restrict__ A, long row_stride, int n, int m,
                                                           const double* __restrict__ w, const double* __restrict__ y,
                                                           const double* __restrict__ cvec, double* __restrict__ part) {
  __shared__ double red[3][GRAM_REC];
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, c = l & 15, q = l >> 4;
  const bool colok = 4 * c < m;
  double cq[4] = {0.0, 0.0, 0.0, 0.0};
  if (cvec && colok) { cq[0] = cvec[4 * c]; cq[1] = cvec[4 * c + 1]; cq[2] = cvec[4 * c + 2]; cq[3] = cvec[4 * c + 3]; }
  g4_t acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = g4_t{0.0, 0.0, 0.0, 0.0};
  double gv[4] = {0.0, 0.0, 0.0, 0.0}, st[4] = {0.0, 0.0, 0.0, 0.0};
  // rows are dealt to the waves in steps of 4 (one step = 4 rows x 64 columns), a fixed contiguous range per wave
  const long nsteps = ((long)n + 3) / 4;
  const long nwaves = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + wv;
  const long per = (nsteps + nwaves - 1) / nwaves;
  const long s0 = gw * per, s1 = min(nsteps, s0 + per);
  for (long s = s0; s < s1; ++s) {
    const long row = 4 * s + q;
    const bool in = row < n;
    const long rc = in ? row : (long)n - 1;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (colok) {
      const double2 lo = *reinterpret_cast<const double2*>(A + rc * row_stride + 4 * c);
      const double2 hi = *reinterpret_cast<const double2*>(A + rc * row_stride + 4 * c + 2);
      a[0] = lo.x; a[1] = lo.y; a[2] = hi.x; a[3] = hi.y;
    }
    const double wi = in ? (w ? w[rc] : 1.0) : 0.0;
    const double yi = y[rc];
    // r = y - A_row . c: the 16 lanes of a row hold four columns each
    double d = a[0] * cq[0] + a[1] * cq[1] + a[2] * cq[2] + a[3] * cq[3];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) d += __shfl_xor(d, off, 64);
    const bool live = wi != 0.0;                       // masked rows: exact zeros whatever A, y hold (NaN-safe)
    const double r = live ? yi - d : 0.0;
    double wa[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = live ? a[e] : 0.0; wa[e] = wi * a[e]; gv[e] = __builtin_fma(wa[e], r, gv[e]); }
    if (c == 0) { st[0] += wi; st[1] = __builtin_fma(wi, r, st[1]); st[2] = __builtin_fma(wi * r, r, st[2]); st[3] += live ? 1.0 : 0.0; }
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = ti; tj < 4; ++tj) { acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[ti], a[tj], acc[t], 0, 0, 0); ++t; }
  }
  // lanes of the same column quad (q = 0..3) and, for the statistics, the four row lanes
#pragma unroll
  for (int e = 0; e < 4; ++e) { gv[e] += __shfl_xor(gv[e], 16, 64); gv[e] += __shfl_xor(gv[e], 32, 64); }
#pragma unroll
  for (int e = 0; e < 4; ++e) { st[e] += __shfl_xor(st[e], 16, 64); st[e] += __shfl_xor(st[e], 32, 64); }
  // cross-wave reduction in wave order (deterministic), wave 0 writes the workgroup's record
  auto put = [&](double* rec) {
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) rec[t * 256 + rg * 64 + l] = acc[t][rg];
    if (q == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rec[2560 + 4 * c + e] = gv[e];
    }
    if (l == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rec[2624 + e] = st[e];
    }
  };
  if (wv > 0) put(red[wv - 1]);
  __syncthreads();
  if (wv == 0) {
    double* rec = part + (long)blockIdx.x * GRAM_REC;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int o = t * 256 + rg * 64 + l;
        rec[o] = ((acc[t][rg] + red[0][o]) + red[1][o]) + red[2][o];
      }
    if (q == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int o = 2560 + 4 * c + e; rec[o] = ((gv[e] + red[0][o]) + red[1][o]) + red[2][o]; }
    }
    if (l == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int o = 2624 + e; rec[o] = ((st[e] + red[0][o]) + red[1][o]) + red[2][o]; }
    }
  }
}

// partial records summed in a fixed order: a workgroup owns 64 output elements, its four waves each sum every fourth record (eight
// loads in flight), the four partial sums are added in wave order (a single workgroup walking 256 records serially measured 650 us
// -- seven times the partial kernel).  Tile (ti, tj) element (i, j) is A^T W A[4 i + ti][4 j + tj]
__global__ __launch_bounds__(256) void gram_finish_kernel(const double* __restrict__ part, int nrec, int m, double* __restrict__ AtA,
                                                          double* __restrict__ Atb, double* __restrict__ stats) {
  __shared__ double acc[4][64];
  const int q = threadIdx.x >> 6, ol = threadIdx.x & 63;
  const int o = blockIdx.x * 64 + ol;
  double s = 0.0;
  if (o < GRAM_REC) {
    int r = q;
    for (; r + 28 < nrec; r += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(r + 4 * u) * GRAM_REC + o];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrec; r += 4) s += part[(long)r * GRAM_REC + o];
  }
  acc[q][ol] = s;
  __syncthreads();
  if (q != 0 || o >= GRAM_REC) return;
  s = ((acc[0][ol] + acc[1][ol]) + acc[2][ol]) + acc[3][ol];
  if (o < 2560) {
    const int t = o >> 8, rg = (o >> 6) & 3, l = o & 63;
    int ti = 0, tj = 0, k = 0;
    for (int a = 0; a < 4; ++a)
      for (int b = a; b < 4; ++b) { if (k == t) { ti = a; tj = b; } ++k; }
    const int i = (l >> 4) + 4 * rg, j = l & 15;
    const int row = 4 * i + ti, col = 4 * j + tj;
    if (row < m && col < m) {
      if (ti != tj) { AtA[(long)row * m + col] = s; AtA[(long)col * m + row] = s; }
      else if (col >= row) { AtA[(long)row * m + col] = s; AtA[(long)col * m + row] = s; }   // diagonal tiles: upper half is the owner
    }
  } else if (o < 2624) {
    if (o - 2560 < m) Atb[o - 2560] = s;
  } else if (stats) {
    stats[o - 2624] = s;
  }
}

// ---- the GP predictor of the distillation (distill_depth.py:30-48 get_predictor) ---------------------------------------------------
//   Kt = K_nm K_mm^-1 (n x m),   var_i = K_nn,ii - sum_c K_nm[i][c] Kt[i][c]
// The reference forms Kt with a library GEMM (157 MB in, 157 MB out at 640x480, m = 64), then an elementwise product and a row
// reduction (two more passes over both matrices).  One pass here: a wave takes 16 rows; lane (j = l & 15, g = l >> 4) holds the
// contiguous columns 16 g .. 16 g + 15 of row j as the A operands of sixteen v_mfma_f64_16x16x4_f64 steps (step s multiplies the
// columns {s, 16 + s, 32 + s, 48 + s}), K_mm^-1 sits in registers in the matching order (64 values per lane, loaded once per wave),
// the four 16-column output tiles are stored with row stride `ldo` (columns m .. ldo - 1 come out as exact zeros: the padded form
// `gram_weighted` reads) and folded with the L1-hot K_nm entries into the row's variance.
__global__ __launch_bounds__(256) void predictor_kernel(const double* __restrict__ Knm, const double* __restrict__ inv,
                                                        const double* __restrict__ diag, int n, int m, int ldo,
                                                        double* __restrict__ Kt, double* __restrict__ var) {
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, j = l & 15, g = l >> 4;
  double Bq[16][4];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int kr = 16 * g + s, kc = 16 * c + j;
      Bq[s][c] = (kr < m && kc < m) ? inv[(long)kr * m + kc] : 0.0;
    }
  const long ntiles = ((long)n + 15) / 16;
  for (long t = (long)blockIdx.x * 4 + wv; t < ntiles; t += (long)gridDim.x * 4) {
    const long r0 = 16 * t;
    const long ra = min(r0 + j, (long)n - 1);                      // (rows past the end: clamped loads, nothing stored)
    double a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = (16 * g + s < m) ? Knm[ra * m + 16 * g + s] : 0.0;
    g4_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = g4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], Bq[s][c], acc[c], 0, 0, 0);
    // tile c, register rg: row r0 + 4 rg + g, column 16 c + j
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const long row = r0 + 4 * rg + g;
      const bool rin = row < n;
      const long rr = rin ? row : (long)n - 1;
      double p = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 16 * c + j;
        if (rin && col < ldo) Kt[rr * ldo + col] = acc[c][rg];
        const double k = (col < m) ? Knm[rr * m + col] : 0.0;
        p = __builtin_fma(k, acc[c][rg], p);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) p += __shfl_xor(p, off, 64);
      if (rin && j == 0) var[row] = diag[row] - p;
    }
  }
}

}  // namespace como

extern "C" {

int como_predictor_f64(const double* Knm, const double* inv, const double* diag, int n, int m, int ldo, double* Kt, double* var,
                       como_stream_t stream) {
  if (!Knm || !inv || !diag || !Kt || !var || n <= 0 || m <= 0 || m > 64 || ldo < m || ldo > 64) return COMO_ERR_ARG;
  long tiles = ((long)n + 15) / 16;
  int blocks = (int)((tiles + 3) / 4);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(como::predictor_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Knm, inv, diag, n, m, ldo, Kt, var);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

long como_gram_workspace_bytes(void) { return 256L * como::GRAM_REC * (long)sizeof(double); }

int como_gram_f64(const double* A, long row_stride, int n, int m, const double* w, const double* y, const double* c, double* AtA,
                  double* Atb, double* stats, void* workspace, como_stream_t stream) {
  if (!A || !y || !AtA || !Atb || !workspace || n <= 0 || m <= 0 || m > 64 || (m & 3) || row_stride < m || (row_stride & 1) ||
      (reinterpret_cast<uintptr_t>(A) & 15))
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long steps = ((long)n + 3) / 4;
  int blocks = (int)((steps + 15) / 16);                 // >= 4 steps per wave
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(como::gram_partial_kernel, dim3(blocks), dim3(256), 0, s, A, row_stride, n, m, w, y, c, (double*)workspace);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(como::gram_finish_kernel, dim3((como::GRAM_REC + 63) / 64), dim3(256), 0, s, (const double*)workspace, blocks, m, AtA,
                     Atb, stats);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
