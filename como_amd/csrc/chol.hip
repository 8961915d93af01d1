// Dense SPD solve of the window's normal equations: delta = H^-1 g, float64, D ~ 760 .. 2400.
//
// Reference: como/odom/backend/linear_system.py:101-112 (`solve_system`: cholesky_ex(check_errors=False) +
// cholesky_solve).  hipSOLVER needs ~2.5 ms for D = 760 (dozens of tiny launches, not graph-capturable); here the
// factorisation is a right-looking blocked Cholesky with ONE launch per PAIR of 32-column panels, no host synchronisation:
//
//   chol_pack      : W (Dp x Dp workspace) <- lower(H), with g appended as row D (so the forward substitution
//                    L y = g falls out of the factorisation: row D of L is y^T) and an identity pad up to Dp = 32 * nb.
//   chol_first2 / chol_panel2(c0): look-ahead column pairs (see the kernels): the panel solve is a product against the
//                    pre-inverted diagonal blocks on the f64 matrix cores, the next pair of diagonal blocks is factored and
//                    inverted by the workgroup that just updated it (factor_invert_tile: four pivots per barrier, rank-4
//                    MFMA updates).  A non-positive pivot is reported in `info` (1-based, first failure) instead of being
//                    swallowed; the factorisation then continues with pivot 1 as a defined value.
//   back-substitution L^T delta = y: up to 40 block columns (D < 1280) it RIDES ALONG with the factorisation (an identity block
//                    appended under g turns into L^-T tile by tile, delta accumulates in the panel launches; chol_xfinish);
//                    above that chol_backsub + chol_backsub_rect, 4 block ranges: triangle / rectangle / triangle ... launches.
#include "chol_tile.cuh"
#include "../../include/como_hip.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace como {

int cholp_init();                                                                      // csrc/cholp.hip
int cholp_solve(double* delta, void* workspace, int D, int* info, hipStream_t s);      // COMO_OK, or COMO_ERR_ARG: not applicable
int cholp_set_enabled(int e);
int cholp_enabled();
void cholp_set_stall(int on);


__global__ __launch_bounds__(256) void chol_pack_kernel(const double* __restrict__ H, const double* __restrict__ g,
                                                        double* __restrict__ W, int D, int Dp, int* __restrict__ info) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) *info = 0;
  cholp_reset_sync(W, D, idx);
  if (idx >= (long)Dp * Dp) return;
  const int i = (int)(idx / Dp), j = (int)(idx % Dp);
  double v = 0.0;
  if (i < D && j < D) v = (j <= i) ? H[(long)i * D + j] : 0.0;
  else if (i == D && j < D) v = g[j];
  else if (i == D && j == D) v = 1e300;          // pivot of the appended row: irrelevant, just positive
  else if (i == j) v = 1.0;                      // identity pad
  W[idx] = v;
}



// Column-pair start: factors block columns 0 and 1 (their 2x2 block of diagonal tiles).
template <bool LEAN>
__global__ __launch_bounds__(512) void chol_first2_kernel(const double* __restrict__ W, double* __restrict__ Lw,
                                                          double* __restrict__ Iw, int Dp, int D, int nb,
                                                          int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double sm[NT2 * TSZ];
  const int tid = threadIdx.x;
  const bool has1 = nb > 1;
  for (int e = tid; e < CB * CB; e += 512) {
    const int r = e / CB, c = e % CB, o = r * CLD + c;
    sm[1 * TSZ + o] = W[(long)r * Dp + c];
    if (has1) {
      sm[o] = W[(long)(CB + r) * Dp + c];
      sm[2 * TSZ + o] = W[(long)(CB + r) * Dp + CB + c];
    }
  }
  __syncthreads();
#ifdef COMO_AB_VARIANTS
  if (!LEAN) { factor_pair_tail(sm, has1, 0, Lw, Iw, Dp, D, info); return; }
#endif
  factor_pair_lean(sm, has1, 0, Lw, Iw, Dp, D, info);
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
template <bool LEAN>
__global__ __launch_bounds__(512) void chol_panel2_kernel(double* __restrict__ W, double* __restrict__ Lw,
                                                          double* __restrict__ Iw, int Dp, int D, int c0, int nb,
                                                          int* __restrict__ info, double* __restrict__ xacc) {
  __shared__ __attribute__((aligned(16))) double sm[NT2 * TSZ];
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
  const int ntrail = nb - d0, tiles_reg = ntrail * (ntrail + 1) / 2;
  const bool app = (int)blockIdx.x >= tiles_reg;   // a tile of the appended identity rows (the ride-along back-substitution)
  int ar = 0;                                      // its block row among the appended rows
  if (app) {
    const int t = blockIdx.x - tiles_reg;
    ar = t / ntrail;
    ti = nb + ar;                                  // block row nb + ar of the 2 Dp x Dp working copy
    tj = d0 + t % ntrail;
  } else {
    int t = blockIdx.x;
    for (int i = d0; i < nb; ++i) {
      const int cnt = i - d0 + 1;            // j = d0 .. i
      if (t < cnt) { ti = i; tj = d0 + t; break; }
      t -= cnt;
    }
  }
  // an appended row that enters with this column pair has never been written: its tiles are still [.. 0 I 0 ..]
  const bool virgin = app && ar >= c0;
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
      t[u][3] = virgin ? ((ar == c0 && r == c) ? 1.0 : 0.0) : W[((long)ti * CB + r) * Dp + k0 + c];
      t[u][4] = W[((long)tj * CB + r) * Dp + k0 + c];
      t[u][5] = virgin ? ((ar == c1 && r == c) ? 1.0 : 0.0) : W[((long)ti * CB + r) * Dp + k1 + c];
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
        double* dst = &W[((long)ti * CB + r) * Dp + (long)tj * CB + c];
        if (ti != tj || c <= r) *dst = (virgin ? 0.0 : *dst) - acc[i];
      }
    } else if (app && tj == D / CB && tid < 256 + CB) {
      // x_r += (L^-T)_{r,c0} y_c0 + (L^-T)_{r,c1} y_c1: sI0 / sI1 hold the two finished tiles of appended row r, and row
      // D mod 32 of sJ0 / sJ1 -- block row D / 32 holds the appended right-hand side: the last block row, or the one before it
      // when the system is padded to whole column PAIRS (common.cuh chol_dp) -- is y for these two block columns.
      // One owner per x_r and launch, launches in stream order: a fixed summation order, no atomics.
      const int t = tid - 256, gl = D - (D / CB) * CB;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int k = 0; k < CB; ++k) {
        s0 = __builtin_fma(sI0[t * CLD + k], sJ0[gl * CLD + k], s0);
        s1 = __builtin_fma(sI1[t * CLD + k], sJ1[gl * CLD + k], s1);
      }
      xacc[ar * CB + t] = (virgin ? 0.0 : xacc[ar * CB + t]) + (s0 + s1);
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
  // (LDS only: the L blocks published above are fire-and-forget stores -- __syncthreads() would wait for their acknowledgement)
#ifdef COMO_AB_VARIANTS
  if (!LEAN) { __syncthreads(); factor_pair_tail(sm, has1, d0, Lw, Iw, Dp, D, info); return; }
#endif
  lds_only_barrier();
  factor_pair_lean(sm, has1, d0, Lw, Iw, Dp, D, info);
}

// Ride-along back-substitution (systems of up to RIDE_MAX_NB block columns).  The working copy carries Dp more rows: an
// identity block B appended under H and g.  The panel operations turn appended row block r into row block r of L^-T (tile (r, c)
// is final once column c is eliminated, like row D turns g into y = L^-1 g), and delta = L^-T y accumulates tile by tile in the
// launches that already exist: no pass over the finished factor at all (the two-triangle back-substitution it replaces was
// 55 us of a 300 us solve at D = 760, on ONE compute unit).  The appended tiles double the trailing updates -- work that is off
// the serial chain of a launch (it ends long before the chain workgroup does) as long as every workgroup of a launch is
// co-resident; from RIDE_MAX_NB on the trailing updates are what a launch waits for, and the substitution kernels below are
// used instead.  Nothing initialises the appended rows: a row block that has not met a column pair yet is known to be
// [0 .. I .. 0] (`virgin`).
// chol_xfinish: the last one or two block columns (cf, cf + 1 -- factored by the last chain workgroup, no trailing tiles left)
// applied to every appended row block r, and delta_r = x_r + (L^-T)_{r,cf} y_cf + (L^-T)_{r,cf+1} y_cf+1.
constexpr int RIDE_MAX_NB = 40;
__global__ __launch_bounds__(256) void chol_xfinish_kernel(const double* __restrict__ W, const double* __restrict__ Lw,
                                                           const double* __restrict__ Iw, int Dp, int D, int nb, int cf,
                                                           const double* __restrict__ xacc, double* __restrict__ delta) {
  __shared__ double sm[5 * TSZ];
  __shared__ double yv[2 * CB];
  double* sV0 = sm;
  double* sV1 = sm + 1 * TSZ;
  double* sL10 = sm + 2 * TSZ;
  double* sB0 = sm + 3 * TSZ;
  double* sB1 = sm + 4 * TSZ;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int r = blockIdx.x, c1 = cf + 1;
  const bool two = c1 < nb, virgin = r >= cf;
  for (int e = tid; e < CB * CB; e += 256) {
    const int rr = e / CB, c = e % CB, o = rr * CLD + c;
    sV0[o] = Iw[((long)cf * CB + rr) * CB + c];
    sB0[o] = virgin ? ((r == cf && rr == c) ? 1.0 : 0.0) : W[((long)(nb + r) * CB + rr) * Dp + (long)cf * CB + c];
    if (two) {
      sV1[o] = Iw[((long)c1 * CB + rr) * CB + c];
      sL10[o] = Lw[((long)c1 * CB + rr) * Dp + (long)cf * CB + c];
      sB1[o] = virgin ? ((r == c1 && rr == c) ? 1.0 : 0.0) : W[((long)(nb + r) * CB + rr) * Dp + (long)c1 * CB + c];
    }
  }
  if (tid < 2 * CB) {                                      // y of the last columns; the appended row's own pivot and the pad are no unknowns
    const int col = cf * CB + tid;
    yv[tid] = (col < D && (tid < CB || two)) ? Lw[(long)D * Dp + col] : 0.0;
  }
  __syncthreads();
  {
    d4_t a = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(sB0, sV0, w, l, a);                       // (L^-T)_{r,cf} = B_r,cf V0^T
    __syncthreads();
    tile_store_mfma(sB0, w, l, a);
  }
  __syncthreads();
  if (two) {
    d4_t b = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(sB0, sL10, w, l, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) sB1[mrow(w, l, i) * CLD + mcol(w, l)] -= b[i];
    __syncthreads();
    d4_t c = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(sB1, sV1, w, l, c);
    __syncthreads();
    tile_store_mfma(sB1, w, l, c);
    __syncthreads();
  }
  if (tid < CB) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int k = 0; k < CB; ++k) s0 = __builtin_fma(sB0[tid * CLD + k], yv[k], s0);
    if (two) {
#pragma unroll 8
      for (int k = 0; k < CB; ++k) s1 = __builtin_fma(sB1[tid * CLD + k], yv[CB + k], s1);
    }
    const double x = (virgin ? 0.0 : xacc[r * CB + tid]) + (s0 + s1);
    if (r * CB + tid < D) delta[r * CB + tid] = x;
  }
}

// L^T delta = y with y = row D of L (columns 0..D-1).
// One workgroup walking the whole factor is bound by what ONE compute unit can pull through the fabric (~12 B/clk: 2.3 MB
// of L at D = 760 -> 68 us; prefetching panels / inverse blocks one or two steps ahead changes nothing at that size), and
// spreading the panel products of every step over many workgroups needs a device-wide hand-off per panel (~4 us each, 24
// of them): no better.  So the solve is split into P block ranges (2, or 4 from 48 blocks on), right to left, with launch
// boundaries as the only synchronisation:
//   chol_backsub(range p)       one workgroup, the triangle of the range (1 / P^2 of the bytes)
//   chol_backsub_rect(range p)  many workgroups: y_j -= L[rows of range p, j]^T x for ALL columns left of the range
//   chol_backsub(range p - 1)   ... (its right-hand side = row D minus the partials of every rectangle to its right)
// Per 32-block (right to left): x_k = L_kk^-T y_k is a 32x32 mat-vec with the pre-inverted diagonal block, then
// y_j -= L[k rows, j]^T x_k for the columns to the left within the range (coalesced along j).
// ysrc: right-hand side of the range (NULL = row D of L); yinit: when not NULL, receives row D of L for the columns left
// of the range (the rect kernel subtracts from it).
constexpr int BS_THREADS = 512;
__global__ __launch_bounds__(BS_THREADS) void chol_backsub_kernel(const double* __restrict__ Lw, const double* __restrict__ Iw,
                                                                  int Dp, int D, int k_lo, int k_hi,
                                                                  const double* __restrict__ ysrc, double* __restrict__ yinit,
                                                                  double* __restrict__ delta,
                                                                  const double* __restrict__ ypart, int nslab, int pstride) {
  __shared__ double y[4096];
  __shared__ double xb[CB];
  const int tid = threadIdx.x;
  const int j_lo = k_lo * CB, j_hi = k_hi * CB;
  for (int j = j_lo + tid; j < j_hi; j += BS_THREADS) {
    double v = (j < D) ? (ysrc ? ysrc[j] : Lw[(long)D * Dp + j]) : 0.0;
    // the rectangle's row slabs, subtracted in slab order (no floating-point atomics: the solve is bit-reproducible)
    if (ypart && j < D) {                             // (nslab is a multiple of 8; eight loads in flight, fixed order of the sum)
      for (int sl = 0; sl < nslab; sl += 8) {
        double pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = ypart[(long)(sl + u) * pstride + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) v -= pv[u];
      }
    }
    y[j] = v;
  }
  if (yinit)
    for (int j = tid; j < j_lo; j += BS_THREADS) yinit[j] = Lw[(long)D * Dp + j];
  // a step is two dependent global-load latencies (inverse block, then the panel rows) unless they are taken off the chain:
  // the inverse block of the NEXT step is staged in LDS during the current one, the first panel slab is fetched before x
  // is known
  __shared__ double sInv[2][CB * CB];
  for (int e = tid; e < CB * CB; e += BS_THREADS) sInv[(k_hi - 1) & 1][e] = Iw[(long)(k_hi - 1) * CB * CB + e];
  // workgroup barrier that waits for LDS traffic only: the panel loads issued for the NEXT step stay in flight across it
  // (__syncthreads() drains the vector-memory counter too)
  auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int j0 = j_lo + tid;
  double vc[CB];                                          // panel slab of the current step, fetched one step ahead
  {
    const long kk = (long)(k_hi - 1) * CB;
    if (j0 < kk) {
#pragma unroll
      for (int r = 0; r < CB; ++r) vc[r] = Lw[(kk + r) * Dp + j0];
    }
  }
  __syncthreads();
  // one step; `cur` was fetched during the previous step, `nxt` is fetched now and consumed by the next one (two register
  // sets used alternately: no copies, the wait for a slab falls a whole step after its issue)
  constexpr int NU = CB * CB / BS_THREADS;
  auto step = [&](int k, double (&cur)[CB], double (&nxt)[CB]) {
    const long kk = (long)k * CB;
    double nx[NU];
    if (k > k_lo) {
      if (j0 < kk - CB) {
#pragma unroll
        for (int r = 0; r < CB; ++r) nxt[r] = Lw[(kk - CB + r) * Dp + j0];
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) nx[u] = Iw[(long)(k - 1) * CB * CB + tid + u * BS_THREADS];
    }
    if (tid < CB) {                                       // x_c = sum_r inv[r][c] y_r
      const double* inv = sInv[k & 1];
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < CB; ++r) s += inv[r * CB + tid] * y[kk + r];
      if (kk + tid >= D) s = 0.0;                         // appended row / pad rows carry no unknowns
      xb[tid] = s;
      if (kk + tid < D) delta[kk + tid] = s;
    }
    lds_barrier();
    if (j0 < kk) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < CB; ++r) s += cur[r] * xb[r];
      y[j0] -= s;
    }
    for (int j = j0 + BS_THREADS; j < kk; j += BS_THREADS) {
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < CB; ++r) s += Lw[(kk + r) * Dp + j] * xb[r];
      y[j] -= s;
    }
    if (k > k_lo) {
#pragma unroll
      for (int u = 0; u < NU; ++u) sInv[(k - 1) & 1][tid + u * BS_THREADS] = nx[u];
    }
    lds_barrier();
  };
  double vd[CB];
  for (int k = k_hi - 1; k >= k_lo; k -= 2) {
    step(k, vc, vd);
    if (k - 1 >= k_lo) step(k - 1, vd, vc);
  }
}

// ypart[slab][j] = sum_{r in slab of [r_lo, r_hi)} L[r][j] x[r] for j < ncols: blockIdx.x = 128-column chunk, blockIdx.y = row slab.
// One partial per (slab, column), summed in slab order by the consumer (atomics would make the solve order-dependent).
__global__ __launch_bounds__(128) void chol_backsub_rect_kernel(const double* __restrict__ Lw, int Dp, int r_lo, int r_hi, int ncols,
                                                                const double* __restrict__ x, double* __restrict__ ypart,
                                                                int pstride) {
  const int j = blockIdx.x * 128 + threadIdx.x;
  const int rows = r_hi - r_lo, per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = r_lo + blockIdx.y * per, r1 = min(r0 + per, r_hi);
  if (j >= ncols) return;
  double s = 0.0;
#pragma unroll 4
  for (int r = r0; r < r1; ++r) s += Lw[(long)r * Dp + j] * x[r];
  ypart[(long)blockIdx.y * pstride + j] = s;
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the small SPD systems of the DepthCov path (n <= 64, float64: K_mm + 1e-6 I -> L_mm, K_mm^-1 once per keyframe, the
// normal equations of the depth distillation; csrc/smallsolve.hip has the interface and the general kernel) on the machinery
// above: ONE workgroup per matrix -- the matrix padded with an identity to 32 / 64 and factored by factor_pair_lean with L and the
// inverted diagonal tiles V0, V1 kept in LDS (~9 us), then
//   L^-1 = [V0 0; X V1], X = -V1 (L10 V0);  A^-1 = L^-T L^-1 = [V0^T V0 + X^T X, .; V1^T X, V1^T V1]   (six 32^3 matrix-core products)
//   A^-1 B by block substitution with the inverted diagonal tiles: y0 = V0 b0, y1 = V1 (b1 - L10 y0), x1 = V1^T y1,
//   x0 = V0^T (y0 - L10^T x1)  -- what the dense solver's ride-along back-substitution does.
// The pivot-by-pivot LDS kernel it replaces ran 145 ... 224 us per 64 x 64 system (192 barriers for the factor alone).

__global__ __launch_bounds__(512) void chol_small64_kernel(const double* __restrict__ A, int n, double* __restrict__ Lout,
                                                           double* __restrict__ Ainv, const double* __restrict__ rhs, int k,
                                                           double* __restrict__ X, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double sm[NT2 * TSZ];
  const int b = blockIdx.x, tid = threadIdx.x, w = (tid >> 6) & 3, l = tid & 63, half = tid >> 8;
  const double* Ab = A + (long)b * n * n;
  const bool has1 = n > CB;
  const int np = has1 ? 2 * CB : CB;
  int* infob = info ? info + b : nullptr;
  __shared__ int info_s;
  if (tid == 0) info_s = 0;
  for (int e = tid; e < np * np; e += 512) {
    const int i = e / np, j = e % np;
    const double v = (i < n && j < n) ? (j <= i ? Ab[(long)i * n + j] : 0.0) : (i == j ? 1.0 : 0.0);     // identity pad
    const int t = i < CB ? 1 : (j < CB ? 0 : 2);           // tile 1 = T00, 0 = T10, 2 = T11
    if (i >= CB || j < CB) sm[t * TSZ + (i & 31) * CLD + (j & 31)] = v;
  }
  __syncthreads();
  factor_pair_lean(sm, has1, 0, (double*)nullptr, (double*)nullptr, 0, n, &info_s, sm, sm + 3 * TSZ);
  __syncthreads();
  if (tid == 0 && infob) *infob = info_s;
  const double* L10 = sm;
  const double* L00 = sm + 1 * TSZ;
  const double* L11 = sm + 2 * TSZ;
  const double* V0 = sm + 3 * TSZ;
  const double* V1 = sm + 4 * TSZ;
  double* S5 = sm + 5 * TSZ;
  double* S6 = sm + 6 * TSZ;
  if (Lout) {
    double* Lb = Lout + (long)b * n * n;
    for (int e = tid; e < n * n; e += 512) {
      const int i = e / n, j = e - i * n;
      const double* t = i < CB ? L00 : (j < CB ? L10 : L11);
      Lb[e] = (j <= i) ? t[(i & 31) * CLD + (j & 31)] : 0.0;
    }
  }
  if (rhs && X) {
    // block substitution, 8 right-hand-side columns per sweep; thread (i, c) = (tid >> 3, tid & 7) for tid < 256
    double* rb = S5;                                        // [64][8] right-hand side -> y -> x
    double* tb = S5 + 512;                                  // [32][8] temporaries
    const int i = (tid >> 3) & 31, c = tid & 7;
    for (int c0 = 0; c0 < k; c0 += 8) {
      const int kc = min(8, k - c0);
      for (int e = tid; e < 64 * 8; e += 512) {
        const int r = e >> 3, cc = e & 7;
        rb[e] = (r < n && cc < kc) ? rhs[((long)b * n + r) * k + c0 + cc] : 0.0;
      }
      __syncthreads();
      double v = 0.0;
      if (tid < 256) {                                      // y0 = V0 b0
#pragma unroll 8
        for (int q = 0; q < CB; ++q) v = __builtin_fma(V0[i * CLD + q], rb[q * 8 + c], v);
      }
      __syncthreads();
      if (tid < 256) rb[i * 8 + c] = v;
      __syncthreads();
      if (has1) {
        v = 0.0;
        if (tid < 256) {                                    // t = b1 - L10 y0
          v = rb[(CB + i) * 8 + c];
#pragma unroll 8
          for (int q = 0; q < CB; ++q) v = __builtin_fma(-L10[i * CLD + q], rb[q * 8 + c], v);
          tb[i * 8 + c] = v;
        }
        __syncthreads();
        v = 0.0;
        if (tid < 256) {                                    // y1 = V1 t
#pragma unroll 8
          for (int q = 0; q < CB; ++q) v = __builtin_fma(V1[i * CLD + q], tb[q * 8 + c], v);
        }
        __syncthreads();
        if (tid < 256) tb[i * 8 + c] = v;
        __syncthreads();
        v = 0.0;
        if (tid < 256) {                                    // x1 = V1^T y1
#pragma unroll 8
          for (int q = 0; q < CB; ++q) v = __builtin_fma(V1[q * CLD + i], tb[q * 8 + c], v);
          rb[(CB + i) * 8 + c] = v;
        }
        __syncthreads();
        v = 0.0;
        if (tid < 256) {                                    // u = y0 - L10^T x1
          v = rb[i * 8 + c];
#pragma unroll 8
          for (int q = 0; q < CB; ++q) v = __builtin_fma(-L10[q * CLD + i], rb[(CB + q) * 8 + c], v);
          tb[i * 8 + c] = v;
        }
        __syncthreads();
      } else {
        if (tid < 256) tb[i * 8 + c] = rb[i * 8 + c];
        __syncthreads();
      }
      v = 0.0;
      if (tid < 256) {                                      // x0 = V0^T u
#pragma unroll 8
        for (int q = 0; q < CB; ++q) v = __builtin_fma(V0[q * CLD + i], tb[q * 8 + c], v);
        rb[i * 8 + c] = v;
      }
      __syncthreads();
      for (int e = tid; e < 64 * 8; e += 512) {
        const int r = e >> 3, cc = e & 7;
        if (r < n && cc < kc) X[((long)b * n + r) * k + c0 + cc] = rb[e];
      }
      __syncthreads();
    }
  }
  if (Ainv) {
    double* Ib = Ainv + (long)b * n * n;
    auto put = [&](int r0, int c0, const d4_t& a, bool mirror) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + mrow(w, l, q), cc = c0 + mcol(w, l);
        if (r < n && cc < n) {
          Ib[(long)r * n + cc] = a[q];
          if (mirror) Ib[(long)cc * n + r] = a[q];
        }
      }
    };
    if (has1) {
      if (half == 0) {                                      // Y = L10 V0
        d4_t y = {0.0, 0.0, 0.0, 0.0};
        tile_mm_mfma<false, false>(L10, V0, w, l, y);
        tile_store_mfma(S5, w, l, y);
      }
      __syncthreads();
      if (half == 0) {                                      // X = -V1 Y
        d4_t x = {0.0, 0.0, 0.0, 0.0};
        tile_mm_mfma<false, false>(V1, S5, w, l, x);
#pragma unroll
        for (int q = 0; q < 4; ++q) S6[mrow(w, l, q) * CLD + mcol(w, l)] = -x[q];
      }
      __syncthreads();
      if (half == 0) {                                      // A^-1_00 = V0^T V0 + X^T X ; A^-1_11 = V1^T V1
        d4_t a = {0.0, 0.0, 0.0, 0.0};
        tile_mm_mfma<true, false>(V0, V0, w, l, a);
        tile_mm_mfma<true, false>(S6, S6, w, l, a);
        put(0, 0, a, false);
      } else {                                              // A^-1_10 = V1^T X (and its mirror), then A^-1_11
        d4_t a = {0.0, 0.0, 0.0, 0.0};
        tile_mm_mfma<true, false>(V1, S6, w, l, a);
        put(CB, 0, a, true);
        d4_t c = {0.0, 0.0, 0.0, 0.0};
        tile_mm_mfma<true, false>(V1, V1, w, l, c);
        put(CB, CB, c, false);
      }
    } else if (half == 0) {
      d4_t a = {0.0, 0.0, 0.0, 0.0};
      tile_mm_mfma<true, false>(V0, V0, w, l, a);
      put(0, 0, a, false);
    }
  }
}

int chol_small64_f64(const double* A, int B, int n, double* L, double* Ainv, const double* rhs, int k, double* X, int* info,
                     hipStream_t s) {
  hipLaunchKernelGGL(chol_small64_kernel, dim3(B), dim3(512), 0, s, A, n, L, Ainv, rhs, k, X, info);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

long como_chol_workspace_bytes(int D) {
  (void)como::cholp_init();                        // (the persistent kernel's LDS attribute: set outside any stream capture)
  const long Dp = como::chol_dp(D);
  const long nb = Dp / como::CB;
  // working copy incl. the appended identity rows (2 Dp x Dp) | factor (Dp x Dp) | inverted diagonal blocks | x accumulator
  return (3 * Dp * Dp + nb * como::CB * como::CB + Dp) * (long)sizeof(double);
}

static int chol_solve_impl(const double* H, const double* g, double* delta, void* workspace, int D, int* info,
                           como_stream_t stream, bool packed) {
  using namespace como;
  if ((!packed && (!H || !g)) || !delta || !workspace || !info || D <= 0 || D > 4000) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Dp = (int)chol_dp(D);
  const int nb = Dp / CB;
  double* W = (double*)workspace;
  const long tot = (long)Dp * Dp;
  double* Lw = W + 2 * tot;
  double* Iw = Lw + tot;
  double* xacc = Iw + (long)nb * CB * CB;
  static const int ride_max = [] {                       // COMO_CHOL_RIDE_MAX: measurement / test override of the cross-over
    const char* e = getenv("COMO_CHOL_RIDE_MAX");
    return e ? atoi(e) : RIDE_MAX_NB;
  }();
  const bool ride = nb <= ride_max;
#ifdef COMO_AB_VARIANTS
  static const bool lean = [] {                           // COMO_CHOL_LEAN=0: round 3's eight-wave tile factorisation (A/B)
    const char* e = getenv("COMO_CHOL_LEAN");
    return e ? atoi(e) != 0 : true;
  }();
#endif
  if (!packed) {                                      // (packed: como_sys_finalize_pack wrote W and reset info)
    hipLaunchKernelGGL(chol_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, H, g, W, D, Dp, info);
    COMO_CHECK_LAUNCH();
  }
  if (Dp > 4096) return COMO_ERR_ARG;
  // 2 .. 16 column pairs on a device with enough compute units: the whole solve in ONE persistent launch (csrc/cholp.hip)
  if (cholp_solve(delta, workspace, D, info, s) == COMO_OK) return COMO_OK;
#ifdef COMO_AB_VARIANTS
  if (!lean) hipLaunchKernelGGL(chol_first2_kernel<false>, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, nb, info);
  else
#endif
  hipLaunchKernelGGL(chol_first2_kernel<true>, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, nb, info);
  COMO_CHECK_LAUNCH();
  for (int c0 = 0; c0 + 2 < nb; c0 += 2) {               // two block columns per launch
    const int r = nb - (c0 + 2);
    const int tiles = r * (r + 1) / 2;
    const int app = ride ? (c0 + 2) * r : 0;             // appended rows 0 .. c0 + 1 x the r trailing block columns
#ifdef COMO_AB_VARIANTS
    if (!lean) hipLaunchKernelGGL(chol_panel2_kernel<false>, dim3(tiles + app), dim3(512), 0, s, W, Lw, Iw, Dp, D, c0, nb, info, xacc);
    else
#endif
    hipLaunchKernelGGL(chol_panel2_kernel<true>, dim3(tiles + app), dim3(512), 0, s, W, Lw, Iw, Dp, D, c0, nb, info, xacc);
    COMO_CHECK_LAUNCH();
  }
  if (ride) {
    hipLaunchKernelGGL(chol_xfinish_kernel, dim3(nb), dim3(256), 0, s, (const double*)W, (const double*)Lw, (const double*)Iw, Dp,
                       D, nb, ((nb - 1) / 2) * 2, (const double*)xacc, delta);
    COMO_CHECK_LAUNCH();
  } else if (nb < 6) {
    hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(BS_THREADS), 0, s, Lw, Iw, Dp, D, 0, nb, (const double*)nullptr,
                       (double*)nullptr, delta, (const double*)nullptr, 0, 0);
    COMO_CHECK_LAUNCH();
  } else {
    // P block ranges, right to left: triangle(range p) -> rectangle(rows of range p x ALL columns left of it) -> triangle(p-1) ...
    // Each triangle is walked by one workgroup (1 / P^2 of the factor's bytes each), the rectangles by many; a column receives
    // the partials of every rectangle to its right, summed in a fixed order.  A triangle launch has ~10 us of fixed cost
    // (right-hand side, first inverse block and panel slab): D = 760 (24 blocks) is fastest with 2 ranges (55 us; 4 ranges:
    // 75 us), D = 2680 (84 blocks) with 4 (two 157 us triangles -> the iteration 4.08 -> 3.87 ms).
    const int P = nb >= 48 ? 4 : 2;
    const int NSLAB = nb >= 48 ? 32 : 8;                  // row slabs per rectangle (a multiple of 8: the consumer's unroll)
    double* ybuf = W;                                     // the working copy is dead once the factor is complete
    double* ypart = W + Dp;                               // (P - 1) x NSLAB row-slab partials, Dp apart (Dp >= 192 rows)
    for (int p = P - 1; p >= 0; --p) {
      const int k_lo = (int)((long)nb * p / P), k_hi = (int)((long)nb * (p + 1) / P);
      const bool first = p == P - 1;
      hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(BS_THREADS), 0, s, Lw, Iw, Dp, D, k_lo, k_hi,
                         first ? (const double*)nullptr : (const double*)ybuf, first ? ybuf : (double*)nullptr, delta,
                         first ? (const double*)nullptr : (const double*)ypart, NSLAB * (P - 1 - p), Dp);
      COMO_CHECK_LAUNCH();
      if (p > 0) {
        const int ncols = k_lo * CB, r_lo = k_lo * CB, r_hi = min(k_hi * CB, D);
        hipLaunchKernelGGL(chol_backsub_rect_kernel, dim3((ncols + 127) / 128, NSLAB), dim3(128), 0, s, Lw, Dp, r_lo, r_hi, ncols,
                           delta, ypart + (long)(P - 1 - p) * NSLAB * Dp, Dp);
        COMO_CHECK_LAUNCH();
      }
    }
  }
  return COMO_OK;
}

int como_chol_solve_f64(const double* H, const double* g, double* delta, void* workspace, int D, int* info,
                        como_stream_t stream) {
  return chol_solve_impl(H, g, delta, workspace, D, info, stream, false);
}

int como_chol_solve_packed_f64(double* delta, void* workspace, int D, int* info, como_stream_t stream) {
  return chol_solve_impl(nullptr, nullptr, delta, workspace, D, info, stream, true);
}

int como_chol_set_persistent(int enable) { return como::cholp_set_enabled(enable); }
int como_chol_persistent_state(void) { return como::cholp_enabled(); }
void como_chol_debug_stall(int on) { como::cholp_set_stall(on); }

}  // extern "C"
