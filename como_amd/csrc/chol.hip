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
                                                   int* __restrict__ info, double* ldsInv = nullptr) {
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
    const double iv = (c <= r) ? oS[e] * sq[r] : 0.0;
    Iw[(long)k * CB * CB + e] = iv;
    if (ldsInv) ldsInv[r * CLD + c] = iv;                 // kept on chip for the second column of a column pair
  }
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// 32x32x32 product X Y^T of two LDS tiles (leading dimension CLD) on the f64 matrix cores: 4 waves, wave w owns the
// 16x16 quadrant (w >> 1, w & 1) and issues 8 v_mfma_f64_16x16x4_f64 on 16 LDS reads (the VALU version -- 128 reads and
// 128 FMAs per thread -- cost ~1.4 us per product, all of it on the serial chain of a panel step).
// Lane l feeds A[row = l & 15][k] and B[k][col = l & 15] with k-group q = l >> 4; the k values of group q are
// 16 (q & 1) + 8 (q >> 1) + s, s = 0..7, which makes the 64-bit LDS reads of each half-wave bank-conflict free.
__device__ __forceinline__ void tile_nt_mfma(const double* X, const double* Y, int w, int l, d4_t& acc) {
  const int r = l & 15, q = l >> 4, ko = 16 * (q & 1) + 8 * (q >> 1);
  const double* x = X + (16 * (w >> 1) + r) * CLD + ko;
  const double* y = Y + (16 * (w & 1) + r) * CLD + ko;
  double xa[8], ya[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) { xa[s] = x[s]; ya[s] = y[s]; }
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s], ya[s], acc, 0, 0, 0);
}
// element (row, col) of accumulator register i of lane l of wave w
__device__ __forceinline__ int mrow(int w, int l, int i) { return 16 * (w >> 1) + (l >> 4) + 4 * i; }
__device__ __forceinline__ int mcol(int w, int l) { return 16 * (w & 1) + (l & 15); }

__device__ __forceinline__ void tile_store_mfma(double* dst, int w, int l, const d4_t& acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[mrow(w, l, i) * CLD + mcol(w, l)] = acc[i];
}

constexpr int TSZ = CB * CLD;      // one LDS tile
constexpr int NT2 = 7;             // tiles of LDS used by the column-pair kernels (59 KB)

// Factor the 2x2 block of tiles [T00 . ; T10 T11] (all already updated by every earlier column): T00 and T11 arrive in
// the registers of threads 0..255 (thread (ty, tx) owns (ty+16a, tx+16b)), T10 in LDS tile 0.  Publishes L_d0d0, L_d1d0,
// L_d1d1 and the two inverted diagonal blocks.  Tiles 1..6 of `sm` are scratch.
__device__ __forceinline__ void factor_pair_tail(double (&v)[2][2], double (&v2)[2][2], double* sm, bool has1, int d0,
                                                 double* __restrict__ Lw, double* __restrict__ Iw, int Dp, int D,
                                                 int* __restrict__ info) {
  const int tid = threadIdx.x, half = tid >> 8, lt = tid & 255, tx = lt & 15, ty = lt >> 4;
  const int w = (tid >> 6) & 3, l = tid & 63;
  double* T10 = sm;
  double* Vn = sm + 1 * TSZ;
  double* Ln = sm + 2 * TSZ;
  double* Up = sm + 6 * TSZ;
  factor_invert_tile(v, sm + 3 * TSZ, sm + 4 * TSZ, sm + 5 * TSZ, Lw, Iw, Dp, d0, D, info, Vn);
  if (!has1) return;
  __syncthreads();
  if (half == 0) {                                       // L_d1d0 = T10 L_d0d0^-T
    d4_t r = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(T10, Vn, w, l, r);
    tile_store_mfma(Ln, w, l, r);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      Lw[((long)(d0 + 1) * CB + mrow(w, l, i)) * Dp + (long)d0 * CB + mcol(w, l)] = r[i];
  }
  __syncthreads();
  if (half == 0) {                                       // T11 -= L_d1d0 L_d1d0^T
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(Ln, Ln, w, l, acc);
    tile_store_mfma(Up, w, l, acc);
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        v2[a][b] = (c <= r) ? v2[a][b] - Up[r * CLD + c] : 0.0;
      }
  }
  factor_invert_tile(v2, sm + 3 * TSZ, sm + 4 * TSZ, sm + 5 * TSZ, Lw, Iw, Dp, d0 + 1, D, info);
}

// Column-pair start: factors block columns 0 and 1 (their 2x2 block of diagonal tiles).
__global__ __launch_bounds__(512) void chol_first2_kernel(const double* __restrict__ W, double* __restrict__ Lw,
                                                          double* __restrict__ Iw, int Dp, int D, int nb,
                                                          int* __restrict__ info) {
  __shared__ double sm[NT2 * TSZ];
  const int tid = threadIdx.x, lt = tid & 255, tx = lt & 15, ty = lt >> 4;
  const bool has1 = nb > 1;
  double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, v2[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  if (tid < 256) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        v[a][b] = (c <= r) ? W[(long)r * Dp + c] : 0.0;
        if (has1) v2[a][b] = (c <= r) ? W[(long)(CB + r) * Dp + CB + c] : 0.0;
      }
  }
  if (has1)
    for (int e = tid; e < CB * CB; e += 512) {
      const int r = e / CB, c = e % CB;
      sm[r * CLD + c] = W[(long)(CB + r) * Dp + c];
    }
  __syncthreads();
  factor_pair_tail(v, v2, sm, has1, 0, Lw, Iw, Dp, D, info);
}

// Look-ahead blocked Cholesky, two block columns per launch.  The serial chain of a panel step -- launch gap, tile
// loads, the in-tile factorisation of the next diagonal block -- bounds this solver (D = 760: 24 block columns, at most
// 276 trailing tiles), so columns c0 and c1 = c0 + 1 are eliminated by ONE launch and the extra products run on the
// matrix cores.  Needs L_c0c0^-1, L_c1c0 and L_c1c1^-1, published by the previous launch's chain workgroup.
// One workgroup per trailing tile (i, j), c1 < j <= i:
//   L_i0 = A_i0 V0^T, A_i1 -= L_i0 L_10^T, L_i1 = A_i1 V1^T  (waves 0..3; the same for block-row j on waves 4..7),
//   A_ij -= L_i0 L_j0^T + L_i1 L_j1^T.
// L_ik = A_ik L_kk^-T is a product against the PRE-INVERTED diagonal block: no serial triangular solve anywhere.
// The chain workgroup is tile (d1, d0) with d0 = c0 + 2, d1 = d0 + 1: from the same four L blocks it also forms the
// updated diagonal tiles (d0, d0) and (d1, d1), keeps all three on chip, and factors + inverts the pair for the next
// launch (factor_pair_tail).
// W: working copy (trailing tiles updated in place); Lw: the factor (a SEPARATE matrix: other workgroups of the launch
// still read the un-factored panel blocks from W); Iw: inverses of the diagonal blocks of L (nb x CB x CB).
__global__ __launch_bounds__(512) void chol_panel2_kernel(double* __restrict__ W, double* __restrict__ Lw,
                                                          double* __restrict__ Iw, int Dp, int D, int c0, int nb,
                                                          int* __restrict__ info) {
  __shared__ double sm[NT2 * TSZ];
  double* sV0 = sm;
  double* sV1 = sm + 1 * TSZ;
  double* sL10 = sm + 2 * TSZ;
  double* sI0 = sm + 3 * TSZ;      // A_i,c0 -> L_i,c0
  double* sJ0 = sm + 4 * TSZ;
  double* sI1 = sm + 5 * TSZ;      // A_i,c1 -> L_i,c1
  double* sJ1 = sm + 6 * TSZ;
  const int tid = threadIdx.x, half = tid >> 8, lt = tid & 255, tx = lt & 15, ty = lt >> 4;
  const int w = (tid >> 6) & 3, l = tid & 63;
  const int c1 = c0 + 1, d0 = c0 + 2, d1 = c0 + 3;
  const bool has1 = d1 < nb;
  int ti = -1, tj = -1;
  {
    int t = blockIdx.x;
    for (int i = d0; i < nb; ++i) {
      const int cnt = i - d0 + 1;            // j = d0 .. i
      if (t < cnt) { ti = i; tj = d0 + t; break; }
      t -= cnt;
    }
  }
  if (has1 && ti == tj && ti <= d1) return;  // tiles (d0,d0), (d1,d1) belong to the chain workgroup
  const bool chain = has1 ? (ti == d1 && tj == d0) : (ti == d0);
  const long k0 = (long)c0 * CB, k1 = (long)c1 * CB;
  {                                                       // all 14 tile loads of a thread in flight at once
    double t[2][7];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, r = e / CB, c = e % CB;
      t[u][0] = Iw[((long)c0 * CB + r) * CB + c];
      t[u][1] = Iw[((long)c1 * CB + r) * CB + c];
      t[u][2] = Lw[(k1 + r) * Dp + k0 + c];
      t[u][3] = W[((long)ti * CB + r) * Dp + k0 + c];
      t[u][4] = W[((long)tj * CB + r) * Dp + k0 + c];
      t[u][5] = W[((long)ti * CB + r) * Dp + k1 + c];
      t[u][6] = W[((long)tj * CB + r) * Dp + k1 + c];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, o = (e / CB) * CLD + e % CB;
#pragma unroll
      for (int q = 0; q < 7; ++q) sm[q * TSZ + o] = t[u][q];
    }
  }
  // the chain workgroup's own trailing tiles, fetched now so that their latency hides behind the products
  double wpre[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  if (chain) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = mrow(w, l, i), c = mcol(w, l);
      if (half == 0) {
        wpre[0][i] = W[((long)d0 * CB + r) * Dp + (long)d0 * CB + c];
        if (has1) wpre[1][i] = W[((long)d1 * CB + r) * Dp + (long)d1 * CB + c];
      } else if (has1) {
        wpre[0][i] = W[((long)d1 * CB + r) * Dp + (long)d0 * CB + c];
      }
    }
  }
  __syncthreads();
  double* s0 = half ? sJ0 : sI0;
  double* s1 = half ? sJ1 : sI1;
  {
    d4_t r = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(s0, sV0, w, l, r);                       // L_x0 = A_x0 V0^T
    __syncthreads();
    tile_store_mfma(s0, w, l, r);
  }
  __syncthreads();
  {
    d4_t r = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(s0, sL10, w, l, r);                      // A_x1 -= L_x0 L_10^T  (own elements only: no barrier before)
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[mrow(w, l, i) * CLD + mcol(w, l)] -= r[i];
  }
  __syncthreads();
  {
    d4_t r = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(s1, sV1, w, l, r);                       // L_x1 = A_x1 V1^T
    __syncthreads();
    tile_store_mfma(s1, w, l, r);
  }
  __syncthreads();
  if (ti == tj || chain) {                                // publish the L blocks of block-row i (and of row j: chain)
    for (int e = tid; e < CB * CB; e += 512) {
      const int r = e / CB, c = e % CB, o = r * CLD + c;
      Lw[((long)ti * CB + r) * Dp + k0 + c] = sI0[o];
      Lw[((long)ti * CB + r) * Dp + k1 + c] = sI1[o];
      if (chain && has1) {
        Lw[((long)tj * CB + r) * Dp + k0 + c] = sJ0[o];
        Lw[((long)tj * CB + r) * Dp + k1 + c] = sJ1[o];
      }
    }
  }
  if (!chain) {
    if (half == 0) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      tile_nt_mfma(sI0, sJ0, w, l, acc);
      tile_nt_mfma(sI1, sJ1, w, l, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = mrow(w, l, i), c = mcol(w, l);
        if (ti != tj || c <= r) W[((long)ti * CB + r) * Dp + (long)tj * CB + c] -= acc[i];
      }
    }
    return;
  }
  // chain workgroup: the updated tiles stay on chip (tiles 0..2 -- V0, V1, L10 -- are dead by now)
  if (half == 0) {                                        // T00 from the row-j blocks (row i when there is no d1)
    const double* a0 = has1 ? sJ0 : sI0;
    const double* a1 = has1 ? sJ1 : sI1;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(a0, a0, w, l, acc);
    tile_nt_mfma(a1, a1, w, l, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = mrow(w, l, i), c = mcol(w, l);
      sm[1 * TSZ + r * CLD + c] = wpre[0][i] - acc[i];
    }
    if (has1) {                                           // T11 from the row-i blocks
      d4_t a11 = {0.0, 0.0, 0.0, 0.0};
      tile_nt_mfma(sI0, sI0, w, l, a11);
      tile_nt_mfma(sI1, sI1, w, l, a11);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = mrow(w, l, i), c = mcol(w, l);
        sm[2 * TSZ + r * CLD + c] = wpre[1][i] - a11[i];
      }
    }
  } else if (has1) {                                      // T10
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(sI0, sJ0, w, l, acc);
    tile_nt_mfma(sI1, sJ1, w, l, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = mrow(w, l, i), c = mcol(w, l);
      sm[r * CLD + c] = wpre[0][i] - acc[i];
    }
  }
  __syncthreads();
  double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, v2[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  if (half == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = ty + 16 * a, c = tx + 16 * b;
        v[a][b] = (c <= r) ? sm[1 * TSZ + r * CLD + c] : 0.0;
        if (has1) v2[a][b] = (c <= r) ? sm[2 * TSZ + r * CLD + c] : 0.0;
      }
  }
  __syncthreads();
  factor_pair_tail(v, v2, sm, has1, d0, Lw, Iw, Dp, D, info);
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
  hipLaunchKernelGGL(chol_first2_kernel, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, nb, info);
  COMO_CHECK_LAUNCH();
  for (int c0 = 0; c0 + 2 < nb; c0 += 2) {               // two block columns per launch
    const int r = nb - (c0 + 2);
    const int tiles = r * (r + 1) / 2;
    hipLaunchKernelGGL(chol_panel2_kernel, dim3(tiles), dim3(512), 0, s, W, Lw, Iw, Dp, D, c0, nb, info);
    COMO_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(1024), 0, s, Lw, Iw, Dp, D, nb, delta);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
