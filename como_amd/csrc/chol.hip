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
#include "common.cuh"
#include "../../include/como_hip.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace como {

constexpr int CB = 32;        // panel / tile width (in-tile factor/solve latency grows as CB^2 per panel: 32 beats 64)
constexpr int CLD = CB + 1;   // padded LDS leading dimension
constexpr int TSZ = CB * CLD;      // one LDS tile
constexpr int NT2 = 7;             // tiles of LDS used by the column-pair kernels (59 KB)

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

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double rsq_cubic(double d) {        // hardware estimate (24 bits) + one third-order correction
  const double r = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * r), r, 1.0);
  return __builtin_fma(r, e * __builtin_fma(0.375, e, 0.5), r);
}

// 4x4 Cholesky M = L L^T and W = L^-1 (row-major lower), ~45 dependent operations.  A non-positive pivot is replaced by 1 and
// reported in `bad` (1-based position inside the block, first failure).
struct Micro4 {
  double i0, i1, i2, i3, w10, w20, w21, w30, w31, w32;
  int bad;
};
__device__ __forceinline__ Micro4 micro_chol4(double m00, double m10, double m20, double m30, double m11, double m21, double m31,
                                              double m22, double m32, double m33) {
  // Two 2x2 blocks, each in closed form: the second pivot of a block is det / first pivot, so rsq(first pivot) and rsq(det) are
  // INDEPENDENT -- two dependent v_rsq_f64 (+ correction) on the chain instead of four.  (This function is the serial chain of the
  // tile factorisation: it runs on the look-ahead wave, once per four pivots.)  det = a c - b^2 cancels exactly as c - (b / sqrt a)^2
  // does; 1 / l11 = sqrt(a) / sqrt(det) = (a rsq(a)) rsq(det).
  Micro4 o;
  int bad = 0;
  double d0 = m00;
  if (!(d0 > 0.0)) { bad = bad ? bad : 1; d0 = 1.0; }
  double detA = __builtin_fma(d0, m11, -(m10 * m10));
  if (!(detA > 0.0)) { bad = bad ? bad : 2; detA = d0; }            // (second pivot 1, like the pivot-by-pivot form)
  const double i0 = rsq_cubic(d0), ra = rsq_cubic(detA);
  const double i1 = ra * (d0 * i0);
  const double l10 = m10 * i0, l20 = m20 * i0, l30 = m30 * i0;
  const double l21 = __builtin_fma(-l20, l10, m21) * i1, l31 = __builtin_fma(-l30, l10, m31) * i1;
  double s22 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, m22));
  const double s32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, m32));
  const double s33 = __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, m33));
  if (!(s22 > 0.0)) { bad = bad ? bad : 3; s22 = 1.0; }
  double detS = __builtin_fma(s22, s33, -(s32 * s32));
  if (!(detS > 0.0)) { bad = bad ? bad : 4; detS = s22; }
  const double i2 = rsq_cubic(s22), rs = rsq_cubic(detS);
  const double i3 = rs * (s22 * i2);
  const double l32 = s32 * i2;
  o.i0 = i0; o.i1 = i1; o.i2 = i2; o.i3 = i3;
  o.w10 = -i1 * (l10 * i0);
  o.w21 = -i2 * (l21 * i1);
  o.w32 = -i3 * (l32 * i2);
  o.w20 = -i2 * __builtin_fma(l21, o.w10, l20 * i0);
  o.w31 = -i3 * __builtin_fma(l32, o.w21, l31 * i1);
  o.w30 = -i3 * __builtin_fma(l32, o.w20, __builtin_fma(l31, o.w10, l30 * i0));
  o.bad = bad;
  return o;
}

// Factor a 32x32 tile (lower part of the LDS tile At, leading dimension CLD) and invert the factor, 512 threads, FOUR
// pivots per barrier.  Measured cost model on MI355X (scripts/micro): a wave issues ~1 VALU instruction per 4-6 cycles, an
// LDS store -> barrier -> load hop is ~150 cycles, dependent f64 ops ~6 cycles: a pivot-by-pivot loop (one barrier per
// pivot, 32 of them) cost ~500 cycles per pivot = 6.7 us per tile, all of it on the serial chain of a panel step.
// Blocked by 4, with the work of a block step split over three ROLES that run concurrently between two barriers:
//   * MICRO (wave 5), one block AHEAD: the 4x4 diagonal micro-block of block s + 1 is brought up to date with a private rank-4
//     look-ahead update (lane (i, j): D'[i][j] = D[i][j] - sum_k Pb[i][k] Pb[j][k], Pb = the four rows of the current panel,
//     24 FMAs), exchanged inside the wave, then every lane runs the 4x4 Cholesky + inverse W_{s+1} (~45 dependent operations)
//     and lane 0 publishes W.  This is the serial chain of the tile: ~60 % of a step of round 2's version, where every
//     update wave ran the micro-block itself BEFORE it could start on its panel (5.4 us per tile).
//   * UPDATE (waves 0..3): the panel P[r][0..3] = A[r][p0..p0+3] W_s^T (W_s read from LDS, one row per lane) and the rank-4
//     trailing update A -= P P^T as ONE v_mfma_f64_16x16x4_f64 per 16x16 quadrant: the waves keep the quadrants of A in the MFMA
//     accumulator layout, lane l supplies P[row l&15][k l>>4].  They publish the next four columns (raw) and the 4x4 diagonal
//     block after next for the micro wave.
//   * INVERSE (waves 4, 6, 7: the three non-zero quadrants of the triangular inverse), one block BEHIND: X_B = W S_B,
//     S -= P X_B, reading W and P from LDS.
// One barrier per block step, 9 in all.  L and L^-1 go to global memory (and L^-1 to `ldsInv` when the caller needs it on chip)
// as they are produced.
// scratch: 880 doubles.  At may alias ldsInv (At is consumed before the first barrier, L^-1 is written after the second).
__device__ __forceinline__ void factor_invert_tile(const double* At, double* scratch, double* __restrict__ Lw,
                                                   double* __restrict__ Iw, int Dp, int k, int D, int* __restrict__ info,
                                                   double* ldsInv = nullptr) {
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const int role = wv == 5 ? 2 : (wv >> 2);          // 0 update, 1 inverse, 2 micro
  const int w = wv & 3;
  const int qa = w >> 1, qb = w & 1, lr = l & 15, kq = l >> 4;
  const long kk = (long)k * CB;
  double* colbuf = scratch;             // [2][4][32] raw columns of the current block: colbuf[t][r] = A[r][p0 + t]
  double* rowbuf = scratch + 256;       // [2][4][32] raw rows of S
  double* Pbuf = scratch + 512;         // [2][4][32] panel of the block the inverse works on: P[t][r] (0 for r < p0 + 4)
  double* Wbuf = scratch + 768;         // [4][16]    micro-inverses W_s (row-major), block s at slot s & 3
  double* Dbuf = scratch + 832;         // [2][16]    4x4 diagonal block two blocks ahead (lower part valid)
  double* Xbuf = scratch + 864;         // [16]       exchange inside the micro wave
  const int rowA = 16 * qa + lr, rowB = 16 * qb + lr;
  const int mi = (l >> 2) & 3, mj = l & 3;            // micro wave: lane <-> entry (mi, mj) of the 4x4 block
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  Micro4 mc = {};
  double draw = 0.0;
  if (role == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * qa + kq + 4 * i, c = 16 * qb + lr;
      acc[i] = (c <= r) ? At[r * CLD + c] : 0.0;
    }
    if (qb == 0 && lr < 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) colbuf[lr * 32 + 16 * qa + kq + 4 * i] = acc[i];
    }
  } else if (role == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (16 * qa + kq + 4 * i == 16 * qb + lr) ? 1.0 : 0.0;
    if (wv != 7) { const int e = (wv == 4 ? 0 : 64) + l, t = e >> 5, j = e & 31; rowbuf[t * 32 + j] = (t == j) ? 1.0 : 0.0; }
  } else {
    // block 0 straight from the tile; the raw block 1 entry of this lane for the first look-ahead
    mc = micro_chol4(At[0], At[CLD], At[2 * CLD], At[3 * CLD], At[CLD + 1], At[2 * CLD + 1], At[3 * CLD + 1], At[2 * CLD + 2],
                     At[3 * CLD + 2], At[3 * CLD + 3]);
    draw = At[(4 + mi) * CLD + 4 + mj];
    if (l == 0) {
      if (mc.bad && kk + mc.bad - 1 < D) atomicCAS(info, 0, (int)kk + mc.bad);
      double* wb = Wbuf;
      wb[0] = mc.i0; wb[1] = 0.0; wb[2] = 0.0; wb[3] = 0.0;
      wb[4] = mc.w10; wb[5] = mc.i1; wb[6] = 0.0; wb[7] = 0.0;
      wb[8] = mc.w20; wb[9] = mc.w21; wb[10] = mc.i2; wb[11] = 0.0;
      wb[12] = mc.w30; wb[13] = mc.w31; wb[14] = mc.w32; wb[15] = mc.i3;
    }
  }
  // ROLLED on purpose: unrolled, the nine steps are ~2000 instructions executed once each -- instruction-fetch bound
#pragma unroll 1
  for (int s = 0; s <= CB / 4; ++s) {
    __syncthreads();
    if (role == 0) {
      if (s == CB / 4) continue;
      const int p0 = 4 * s;
      const double* cb = colbuf + (s & 1) * 128;
      const double* wr = Wbuf + (s & 3) * 16 + 4 * kq;      // row kq of W_s
      const double wk0 = wr[0], wk1 = wr[1], wk2 = wr[2], wk3 = wr[3];
      // ---- panel entries this lane feeds to the matrix core: P[rowA][kq], P[rowB][kq]
      const double pA = __builtin_fma(cb[96 + rowA], wk3, __builtin_fma(cb[64 + rowA], wk2, __builtin_fma(cb[32 + rowA], wk1, cb[rowA] * wk0)));
      const double pB = __builtin_fma(cb[96 + rowB], wk3, __builtin_fma(cb[64 + rowB], wk2, __builtin_fma(cb[32 + rowB], wk1, cb[rowB] * wk0)));
      const double pAu = rowA >= p0 + 4 ? pA : 0.0;     // rows of the block itself (and above) take no part in the update
      const double pBu = rowB >= p0 + 4 ? pB : 0.0;
      if (w != 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pAu, pBu, acc, 0, 0, 0);   // quadrant (0,1) is above the diagonal
      if (qb == 0) {                                    // waves 0 and 2 hold P for rows 0..31 once each
        if (p0 + kq <= rowA) Lw[(kk + rowA) * Dp + kk + p0 + kq] = pA;
        Pbuf[(s & 1) * 128 + kq * 32 + rowA] = pAu;
      }
      const int c = 16 * qb + lr;
      if (s + 1 < CB / 4 && w != 1 && c >= p0 + 4 && c < p0 + 8) {
        double* nb = colbuf + ((s + 1) & 1) * 128 + (c - (p0 + 4)) * 32 + 16 * qa + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) nb[4 * i] = acc[i];
      }
      // the diagonal 4x4 block after next, updated through this step: rows p0 + 8 + kq of the diagonal quadrant that holds it
      if (s + 2 < CB / 4 && qa == qb && qa == ((p0 + 8) >> 4) && c >= p0 + 8 && c < p0 + 12) {
        const int isel = ((p0 + 8) & 15) >> 2;
        const double v = isel == 0 ? acc[0] : (isel == 1 ? acc[1] : (isel == 2 ? acc[2] : acc[3]));
        Dbuf[((s + 1) & 1) * 16 + kq * 4 + (c - (p0 + 8))] = v;
      }
    } else if (role == 2) {
      if (s >= CB / 4 - 1) continue;                     // W_1 .. W_7 at steps 0 .. 6
      const int p0 = 4 * s;
      const double* cb = colbuf + (s & 1) * 128 + p0 + 4;  // rows p0 + 4 .. p0 + 7 of the current raw columns
      const double a0 = cb[mi], a1 = cb[32 + mi], a2 = cb[64 + mi], a3 = cb[96 + mi];
      const double b0 = cb[mj], b1 = cb[32 + mj], b2 = cb[64 + mj], b3 = cb[96 + mj];
      const double dr = s == 0 ? draw : Dbuf[(s & 1) * 16 + mi * 4 + mj];
      // Pb = C W_s^T for rows mi and mj (W lower triangular)
      const double pa0 = a0 * mc.i0, pb0 = b0 * mc.i0;
      const double pa1 = __builtin_fma(a1, mc.i1, a0 * mc.w10), pb1 = __builtin_fma(b1, mc.i1, b0 * mc.w10);
      const double pa2 = __builtin_fma(a2, mc.i2, __builtin_fma(a1, mc.w21, a0 * mc.w20));
      const double pb2 = __builtin_fma(b2, mc.i2, __builtin_fma(b1, mc.w21, b0 * mc.w20));
      const double pa3 = __builtin_fma(a3, mc.i3, __builtin_fma(a2, mc.w32, __builtin_fma(a1, mc.w31, a0 * mc.w30)));
      const double pb3 = __builtin_fma(b3, mc.i3, __builtin_fma(b2, mc.w32, __builtin_fma(b1, mc.w31, b0 * mc.w30)));
      const double dn = __builtin_fma(-pa3, pb3, __builtin_fma(-pa2, pb2, __builtin_fma(-pa1, pb1, __builtin_fma(-pa0, pb0, dr))));
      if (l < 16) Xbuf[l] = dn;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its LDS accesses complete in order
      mc = micro_chol4(Xbuf[0], Xbuf[4], Xbuf[8], Xbuf[12], Xbuf[5], Xbuf[9], Xbuf[13], Xbuf[10], Xbuf[14], Xbuf[15]);
      if (l == 0) {
        if (mc.bad && kk + p0 + 4 + mc.bad - 1 < D) atomicCAS(info, 0, (int)kk + p0 + 4 + mc.bad);
        double* wb = Wbuf + ((s + 1) & 3) * 16;
        wb[0] = mc.i0; wb[1] = 0.0; wb[2] = 0.0; wb[3] = 0.0;
        wb[4] = mc.w10; wb[5] = mc.i1; wb[6] = 0.0; wb[7] = 0.0;
        wb[8] = mc.w20; wb[9] = mc.w21; wb[10] = mc.i2; wb[11] = 0.0;
        wb[12] = mc.w30; wb[13] = mc.w31; wb[14] = mc.w32; wb[15] = mc.i3;
      }
    } else {
      if (s == 0) continue;
      const int sb = s - 1, p0 = 4 * sb, prv = sb & 1;
      const double* wb = Wbuf + (sb & 3) * 16 + 4 * kq;
      const double* rb = rowbuf + prv * 128;
      const double pA = Pbuf[prv * 128 + kq * 32 + rowA];
      const double x = __builtin_fma(wb[3], rb[96 + rowB], __builtin_fma(wb[2], rb[64 + rowB], __builtin_fma(wb[1], rb[32 + rowB], wb[0] * rb[rowB])));
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pA, x, acc, 0, 0, 0);
      if (wv != 6) {                                    // waves 4 and 7 hold X[p0 + kq][0..15] / [16..31]
        Iw[(long)k * CB * CB + (p0 + kq) * CB + rowB] = x;
        if (ldsInv) ldsInv[(p0 + kq) * CLD + rowB] = x;
      }
      if (sb + 1 < CB / 4) {                            // rows p0+4 .. p0+7 of S are final: publish them raw
        if (qa == ((p0 + 4) >> 4)) {
          const int isel = ((p0 + 4) & 15) >> 2;
          const double v = isel == 0 ? acc[0] : (isel == 1 ? acc[1] : (isel == 2 ? acc[2] : acc[3]));
          rowbuf[(s & 1) * 128 + kq * 32 + rowB] = v;
        } else if (wv == 7) {                           // rows above 16 are zero right of column 15 (no wave holds that quadrant)
          rowbuf[(s & 1) * 128 + kq * 32 + rowB] = 0.0;
        }
      }
    }
  }
}



// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the 2x2 block of diagonal tiles [T00 . ; T10 T11] factored as ONE continuous 64x64 factorisation (16 four-pivot steps,
// 17 LDS-only barriers) by FIVE lean waves instead of: tile factor (8 waves, 9 barriers) -> L10 = T10 V0^T -> T11 -= L10 L10^T
// -> tile factor.  What the in-kernel timestamps of round 3 said: a four-pivot step cost ~0.55 us with eight waves queueing
// ~150 LDS instructions per step behind one another and draining their global stores at every __syncthreads; the arithmetic of
// a step is a 4x4 micro-factor (the serial chain) and a handful of rank-4 matrix-core updates.  So:
//   * the lower triangle of the 64x64 block lives in the MFMA accumulator layout of THREE update waves by quadrant
//     (16x16 quadrants (R, C) of the 4x4 quadrant grid):  U0: (0,0) (1,0) (1,1) = T00;  U1: (2,0) (2,1) (3,0) (3,1) = T10;
//     U2: (2,2) (3,2) (3,3) = T11.  Step s (pivots p0 = 4 s .. p0 + 3): every wave forms the panel entries it feeds to the
//     matrix core itself, P[row][k] = C_s[row][:] . W_s[k][:] (C_s = the raw columns p0..p0+3, W_s = the inverse of the 4x4
//     micro-factor), one v_mfma_f64_16x16x4_f64 per live quadrant, and the owner of the quadrant column that holds columns
//     p0+4..p0+7 publishes them (raw, through this step) for step s + 1.  T10 is eliminated by the SAME steps that factor
//     T00 (its panel rows are L10: no product against V0), T11 receives its rank-4 updates as they are produced, so the
//     factorisation of T11 simply continues at step 8 -- no hand-over, no ramp;
//   * MICRO wave, one block ahead (as in round 3): D_{s+1} = raw block - Pb Pb^T with a private rank-4 look-ahead, 4x4 factor +
//     inverse, publishes W_{s+1};
//   * INVERSE waves, one block behind: X_B = W S_B, S -= P X_B on the three non-zero quadrants of the 32x32 inverse; one wave
//     per diagonal tile (a wave that shares its SIMD with the update wave that is idle during its eight steps);
//   * nothing but LDS between two barriers of the chain: L and L^-1 go to memory as fire-and-forget stores, the barrier waits
//     for the LDS counter only.
// has1 = false: a single 32x32 tile (odd block-column count): U0, the micro wave and one inverse wave, 8 steps.
// scratch (1392 doubles): Cb [2][64][4] raw columns | Pb [2][64][4] masked panel (for the inverse) | Rb [2][4][32] raw rows of S
// | Wb [4][16] | Db [2][16] diagonal block two ahead | Xb [16].
#ifdef COMO_FP_PROFILE                                    // scripts/micro/chol_pair.hip: when does each role wave reach / leave a barrier
__device__ long* fp_prof = nullptr;                      // [wave 0..7][step 0..17][2]: cycle counter after the barrier / when the step's work is done
#define FP_STAMP(wave, step, which) do { if ((threadIdx.x & 63) == 0 && fp_prof) fp_prof[((wave) * 18 + (step)) * 2 + (which)] = __builtin_readcyclecounter(); } while (0)
#else
#define FP_STAMP(wave, step, which) do { } while (0)
#endif
#ifndef COMO_FP_ABLATE                                   // scripts/micro/chol_pair.hip only: bit 0 / 1 / 2 = the micro / update / inverse
#define COMO_FP_ABLATE 0                                 // waves skip their work (wrong numbers, the other roles' step time)
#endif
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int FP_PB = 512, FP_RB = 1024, FP_WB = 1280, FP_DB = 1344, FP_XB = 1376;

template <int ROLE> struct FpQuad {
  static constexpr int NQ = ROLE == 1 ? 4 : 3;
  static constexpr int R0 = ROLE == 0 ? 0 : 2;
  static constexpr int C0 = ROLE == 2 ? 2 : 0;
  __host__ __device__ static constexpr int qr(int q) { return ROLE == 1 ? R0 + (q >> 1) : R0 + (q + 1) / 2; }
  __host__ __device__ static constexpr int qc(int q) { return ROLE == 1 ? (q & 1) : C0 + (q == 2 ? 1 : 0); }
  __host__ __device__ static constexpr bool needs(int X) { return ROLE == 1 ? true : (X >= R0 && X < R0 + 2); }
};

// Ls (optional): the factor ALSO goes to the LDS tiles of the pair (tile 1 = L00, tile 0 = L10, tile 2 = L11 -- over the inputs,
// which are only read before the first barrier); Vs (optional): the inverted diagonal tiles to LDS (Vs = V0, Vs + TSZ = V1).
template <int ROLE>
__device__ __forceinline__ void fp_update_wave(const double* T, double* sc, int nsteps, long kk, double* __restrict__ Lw, int Dp,
                                               int l, double* Ls = nullptr) {
  using Q = FpQuad<ROLE>;
  const int lr = l & 15, kq = l >> 4;
  double* Cb = sc;
  double* Pb = sc + FP_PB;
  double* Wb = sc + FP_WB;
  double* Db = sc + FP_DB;
  d4_t acc[Q::NQ];
#pragma unroll
  for (int q = 0; q < Q::NQ; ++q) {
    const int R = Q::qr(q), C = Q::qc(q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * (R - Q::R0) + kq + 4 * i, c = 16 * (C - Q::C0) + lr;
      double v = T[r * CLD + c];
      if (ROLE != 1 && R == C && c > r) v = 0.0;
      acc[q][i] = v;
    }
    if (C == 0 && lr < 4) {                               // raw columns 0..3 for step 0
#pragma unroll
      for (int i = 0; i < 4; ++i) Cb[(16 * R + kq + 4 * i) * 4 + lr] = acc[q][i];
    }
  }
  const int s_end = ROLE == 2 ? nsteps : 8;               // T00 / T10 are final after step 7
  const int s_beg = 0;
#pragma unroll 1
  for (int s = 0; s <= nsteps; ++s) {
    FP_STAMP(ROLE, s, 1);
    lds_only_barrier();
    FP_STAMP(ROLE, s, 0);
    if ((COMO_FP_ABLATE & 2) || s < s_beg || s >= s_end) continue;
    const int p0 = 4 * s;
    const double* cb = Cb + (s & 1) * 256;
    const d2_t* wr = (const d2_t*)(Wb + (s & 3) * 16 + 4 * kq);
    const d2_t w01 = wr[0], w23 = wr[1];
    // No data-dependent or step-dependent branch around the matrix instructions: a panel entry of a row above the current
    // block is SELECTED to zero (stale / never-written rows of Cb may hold anything), and a rank-4 update with a zero operand
    // leaves its quadrant alone -- branches around v_mfma made the compiler shuttle whole accumulators between registers.
    double p[4] = {0.0, 0.0, 0.0, 0.0}, pm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int X = 0; X < 4; ++X) {
      if (!Q::needs(X)) continue;
      const int row = 16 * X + lr;
      const d2_t* cr = (const d2_t*)(cb + row * 4);
      const d2_t c01 = cr[0], c23 = cr[1];
      p[X] = __builtin_fma(c23[1], w23[1], __builtin_fma(c23[0], w23[0], __builtin_fma(c01[1], w01[1], c01[0] * w01[0])));
      pm[X] = row >= p0 + 4 ? p[X] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < Q::NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pm[Q::qr(q)], pm[Q::qc(q)], acc[q], 0, 0, 0);
    // the factor: rows 0..31 from U0, rows 32..63 from U1 while T00 is eliminated (that is L10), from U2 afterwards (L11)
    if (ROLE != 2 || p0 >= 32) {
#pragma unroll
      for (int X = 0; X < 4; ++X) {
        if (ROLE == 0 ? X >= 2 : X < 2) continue;
        const int row = 16 * X + lr;
        if (row >= p0 + kq) {
          if (Lw) Lw[(kk + row) * Dp + kk + p0 + kq] = p[X];
          if (Ls) Ls[(ROLE == 0 ? 1 : (ROLE == 1 ? 0 : 2)) * TSZ + (row & 31) * CLD + ((p0 + kq) & 31)] = p[X];
        }
        if (ROLE != 1) Pb[(s & 1) * 256 + row * 4 + kq] = pm[X];
      }
    }
    if (p0 + 4 < 4 * nsteps) {                             // raw columns p0+4 .. p0+7, updated through this step
      const int qcn = (p0 + 4) >> 4, cb0 = (p0 + 4) & 15;
      double* nb = Cb + ((s + 1) & 1) * 256;
#pragma unroll
      for (int q = 0; q < Q::NQ; ++q) {
        if (Q::qc(q) != qcn) continue;
        if (lr >= cb0 && lr < cb0 + 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) nb[(16 * Q::qr(q) + kq + 4 * i) * 4 + lr - cb0] = acc[q][i];
        }
      }
    }
    if (p0 + 8 < 4 * nsteps) {                             // the diagonal 4x4 block after next, for the micro wave's look-ahead
      const int qd = (p0 + 8) >> 4, cbase = (p0 + 8) & 15, isel = cbase >> 2;
#pragma unroll
      for (int q = 0; q < Q::NQ; ++q) {
        if (Q::qr(q) != Q::qc(q) || Q::qr(q) != qd) continue;
        if (lr >= cbase && lr < cbase + 4) {
          const double v = isel == 0 ? acc[q][0] : (isel == 1 ? acc[q][1] : (isel == 2 ? acc[q][2] : acc[q][3]));
          Db[((s + 1) & 1) * 16 + kq * 4 + lr - cbase] = v;
        }
      }
    }
  }
}

__device__ __forceinline__ void fp_publish_w(double* wb, const Micro4& mc) {
  d2_t* w = (d2_t*)wb;
  w[0] = d2_t{mc.i0, 0.0};   w[1] = d2_t{0.0, 0.0};
  w[2] = d2_t{mc.w10, mc.i1}; w[3] = d2_t{0.0, 0.0};
  w[4] = d2_t{mc.w20, mc.w21}; w[5] = d2_t{mc.i2, 0.0};
  w[6] = d2_t{mc.w30, mc.w31}; w[7] = d2_t{mc.w32, mc.i3};
}

__device__ __forceinline__ void fp_micro_wave(const double* T00, double* sc, int nsteps, long kk, int D, int* __restrict__ info,
                                              int l) {
  double* Cb = sc;
  double* Wb = sc + FP_WB;
  double* Db = sc + FP_DB;
  double* Xb = sc + FP_XB;
  const int mi = (l >> 2) & 3, mj = l & 3;
  Micro4 mc = micro_chol4(T00[0], T00[CLD], T00[2 * CLD], T00[3 * CLD], T00[CLD + 1], T00[2 * CLD + 1], T00[3 * CLD + 1],
                          T00[2 * CLD + 2], T00[3 * CLD + 2], T00[3 * CLD + 3]);
  const double draw = T00[(4 + mi) * CLD + 4 + mj];
  if (l == 0) {
    if (mc.bad && kk + mc.bad - 1 < D) atomicCAS(info, 0, (int)kk + mc.bad);
    fp_publish_w(Wb, mc);
  }
#pragma unroll 1
  for (int s = 0; s <= nsteps; ++s) {
    FP_STAMP(3, s, 1);
    lds_only_barrier();
    FP_STAMP(3, s, 0);
    if ((COMO_FP_ABLATE & 1) || s >= nsteps - 1) continue;                         // W_1 .. W_{nsteps-1} at steps 0 .. nsteps-2
    const int p0 = 4 * s;
    const double* cb = Cb + (s & 1) * 256 + (p0 + 4) * 4;  // rows p0+4 .. p0+7 of the current raw columns
    const d2_t* ca = (const d2_t*)(cb + mi * 4);
    const d2_t* cq = (const d2_t*)(cb + mj * 4);
    const d2_t a01 = ca[0], a23 = ca[1], b01 = cq[0], b23 = cq[1];
    const double a0 = a01[0], a1 = a01[1], a2 = a23[0], a3 = a23[1];
    const double b0 = b01[0], b1 = b01[1], b2 = b23[0], b3 = b23[1];
    const double dr = s == 0 ? draw : Db[(s & 1) * 16 + mi * 4 + mj];
    const double pa0 = a0 * mc.i0, pb0 = b0 * mc.i0;
    const double pa1 = __builtin_fma(a1, mc.i1, a0 * mc.w10), pb1 = __builtin_fma(b1, mc.i1, b0 * mc.w10);
    const double pa2 = __builtin_fma(a2, mc.i2, __builtin_fma(a1, mc.w21, a0 * mc.w20));
    const double pb2 = __builtin_fma(b2, mc.i2, __builtin_fma(b1, mc.w21, b0 * mc.w20));
    const double pa3 = __builtin_fma(a3, mc.i3, __builtin_fma(a2, mc.w32, __builtin_fma(a1, mc.w31, a0 * mc.w30)));
    const double pb3 = __builtin_fma(b3, mc.i3, __builtin_fma(b2, mc.w32, __builtin_fma(b1, mc.w31, b0 * mc.w30)));
    const double dn = __builtin_fma(-pa3, pb3, __builtin_fma(-pa2, pb2, __builtin_fma(-pa1, pb1, __builtin_fma(-pa0, pb0, dr))));
    if (l < 16) Xb[l] = dn;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // one wave: its LDS accesses complete in order
    mc = micro_chol4(Xb[0], Xb[4], Xb[8], Xb[12], Xb[5], Xb[9], Xb[13], Xb[10], Xb[14], Xb[15]);
    if (l == 0) {
      if (mc.bad && kk + p0 + 4 + mc.bad - 1 < D) atomicCAS(info, 0, (int)kk + p0 + 4 + mc.bad);
      fp_publish_w(Wb + ((s + 1) & 3) * 16, mc);
    }
  }
}

// inverse of diagonal tile `itile` (0 / 1): active at steps 8 itile + 1 .. 8 itile + 8, one block behind the update waves
__device__ __forceinline__ void fp_inverse_wave(double* sc, int nsteps, int itile, double* __restrict__ Iw_tile, int l,
                                                double* Vs = nullptr) {
  const int lr = l & 15, kq = l >> 4;
  const double* Pb = sc + FP_PB;
  double* Rb = sc + FP_RB;
  const double* Wb = sc + FP_WB;
  d4_t S00, S10 = {0.0, 0.0, 0.0, 0.0}, S11;
#pragma unroll
  for (int i = 0; i < 4; ++i) S00[i] = (kq + 4 * i == lr) ? 1.0 : 0.0;
  S11 = S00;
  if (itile == 0) {                                        // rows 0..3 of the identity (tile 1: written during step 8, below)
    Rb[kq * 32 + lr] = kq == lr ? 1.0 : 0.0;
    Rb[kq * 32 + 16 + lr] = 0.0;
  }
  const int s_first = 8 * itile + 1;
#pragma unroll 1
  for (int s = 0; s <= nsteps; ++s) {
    FP_STAMP(4 + itile, s, 1);
    lds_only_barrier();
    FP_STAMP(4 + itile, s, 0);
    if (itile == 1 && s == 8) {                            // Rb[0] is free during step 8 (tile 0's last inverse step reads Rb[1])
      Rb[kq * 32 + lr] = kq == lr ? 1.0 : 0.0;
      Rb[kq * 32 + 16 + lr] = 0.0;
    }
    if ((COMO_FP_ABLATE & 4) || s < s_first || s >= s_first + 8) continue;
    const int sb = s - 1, p0 = 4 * sb, p0l = p0 - 32 * itile, prv = sb & 1;
    const d2_t* wr = (const d2_t*)(Wb + (sb & 3) * 16 + 4 * kq);
    const d2_t w01 = wr[0], w23 = wr[1];
    const double* rb = Rb + prv * 128;
    const double xLo = __builtin_fma(w23[1], rb[96 + lr], __builtin_fma(w23[0], rb[64 + lr], __builtin_fma(w01[1], rb[32 + lr], w01[0] * rb[lr])));
    const double xHi = __builtin_fma(w23[1], rb[112 + lr], __builtin_fma(w23[0], rb[80 + lr], __builtin_fma(w01[1], rb[48 + lr], w01[0] * rb[16 + lr])));
    const double pLo = Pb[prv * 256 + (32 * itile + lr) * 4 + kq], pHi = Pb[prv * 256 + (32 * itile + 16 + lr) * 4 + kq];
    if (p0l < 12) {                                        // (rows below 16 take part in the update only while p0 + 4 < 16)
      S00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pLo, xLo, S00, 0, 0, 0);
    }
    S10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pHi, xLo, S10, 0, 0, 0);
    S11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pHi, xHi, S11, 0, 0, 0);
    if (Iw_tile) {
      Iw_tile[(p0l + kq) * CB + lr] = xLo;
      Iw_tile[(p0l + kq) * CB + 16 + lr] = xHi;
    }
    if (Vs) {
      Vs[itile * TSZ + (p0l + kq) * CLD + lr] = xLo;
      Vs[itile * TSZ + (p0l + kq) * CLD + 16 + lr] = xHi;
    }
    if (p0l + 4 < CB) {                                    // rows p0l+4 .. p0l+7 of S are final: publish them raw
      const int qa = (p0l + 4) >> 4, isel = ((p0l + 4) & 15) >> 2;
      double vlo, vhi;
      if (qa == 0) {
        vlo = isel == 0 ? S00[0] : (isel == 1 ? S00[1] : (isel == 2 ? S00[2] : S00[3]));
        vhi = 0.0;
      } else {
        vlo = isel == 0 ? S10[0] : (isel == 1 ? S10[1] : (isel == 2 ? S10[2] : S10[3]));
        vhi = isel == 0 ? S11[0] : (isel == 1 ? S11[1] : (isel == 2 ? S11[2] : S11[3]));
      }
      Rb[(s & 1) * 128 + kq * 32 + lr] = vlo;
      Rb[(s & 1) * 128 + kq * 32 + 16 + lr] = vhi;
    }
  }
}

// T10 in tile 0, T00 in tile 1, T11 in tile 2 of `sm` (as factor_pair_tail); scratch = tiles 5, 6.
__device__ __forceinline__ void factor_pair_lean(double* sm, bool has1, int d0, double* __restrict__ Lw, double* __restrict__ Iw,
                                                 int Dp, int D, int* __restrict__ info, double* Ls = nullptr, double* Vs = nullptr) {
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const double* T10 = sm;
  const double* T00 = sm + 1 * TSZ;
  const double* T11 = sm + 2 * TSZ;
  double* sc = sm + 5 * TSZ;
  const int nsteps = has1 ? 16 : 8;
  const long kk = (long)d0 * CB;
  if (wv == 0) fp_update_wave<0>(T00, sc, nsteps, kk, Lw, Dp, l, Ls);
  else if (wv == 1 && has1) fp_update_wave<1>(T10, sc, nsteps, kk, Lw, Dp, l, Ls);
  else if (wv == 2 && has1) fp_update_wave<2>(T11, sc, nsteps, kk, Lw, Dp, l, Ls);
  else if (wv == 3) fp_micro_wave(T00, sc, nsteps, kk, D, info, l);
  else if (wv == 6) fp_inverse_wave(sc, nsteps, 0, Iw ? Iw + (long)d0 * CB * CB : nullptr, l, Vs);
  else if (wv == 4 && has1) fp_inverse_wave(sc, nsteps, 1, Iw ? Iw + (long)(d0 + 1) * CB * CB : nullptr, l, Vs);
  else {
#pragma unroll 1
    for (int s = 0; s <= nsteps; ++s) lds_only_barrier();
  }
}

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


// Factor the 2x2 block of tiles [T00 . ; T10 T11] (all already updated by every earlier column), given in LDS: T10 in tile
// 0, T00 in tile 1, T11 in tile 2 (lower parts valid).  Publishes L_d0d0, L_d1d0, L_d1d1 and the two inverted diagonal
// blocks.  Tiles 3 and 5 are scratch; tile 1 receives L_d0d0^-1.
__device__ __forceinline__ void factor_pair_tail(double* sm, bool has1, int d0, double* __restrict__ Lw,
                                                 double* __restrict__ Iw, int Dp, int D, int* __restrict__ info) {
  const int tid = threadIdx.x, half = tid >> 8;
  const int w = (tid >> 6) & 3, l = tid & 63;
  double* T10 = sm;
  double* Vn = sm + 1 * TSZ;
  double* T11 = sm + 2 * TSZ;
  double* Ln = sm + 3 * TSZ;
  double* scratch = sm + 5 * TSZ;
  factor_invert_tile(Vn, scratch, Lw, Iw, Dp, d0, D, info, Vn);
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
  if (half == 0) {                                       // T11 -= L_d1d0 L_d1d0^T, in place (each lane its own elements)
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    tile_nt_mfma(Ln, Ln, w, l, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) T11[mrow(w, l, i) * CLD + mcol(w, l)] -= acc[i];
  }
  __syncthreads();
  factor_invert_tile(T11, scratch, Lw, Iw, Dp, d0 + 1, D, info);
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
  if (LEAN) factor_pair_lean(sm, has1, 0, Lw, Iw, Dp, D, info);
  else factor_pair_tail(sm, has1, 0, Lw, Iw, Dp, D, info);
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
    } else if (app && tj == nb - 1 && tid < 256 + CB) {
      // x_r += (L^-T)_{r,c0} y_c0 + (L^-T)_{r,c1} y_c1: sI0 / sI1 hold the two finished tiles of appended row r, and row
      // D - 32 (nb - 1) of sJ0 / sJ1 (block row nb - 1 holds the appended right-hand side) is y for these two block columns.
      // One owner per x_r and launch, launches in stream order: a fixed summation order, no atomics.
      const int t = tid - 256, gl = D - (nb - 1) * CB;
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
  if (LEAN) lds_only_barrier(); else __syncthreads();
  if (LEAN) factor_pair_lean(sm, has1, d0, Lw, Iw, Dp, D, info);
  else factor_pair_tail(sm, has1, d0, Lw, Iw, Dp, D, info);
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
template <bool TA, bool TB>      // out = op(X) op(Y): element (i, k) of op(X) is X[i][k] (TA: X[k][i]); (k, j) of op(Y) is Y[k][j] (TB: Y[j][k])
__device__ __forceinline__ void tile_mm_mfma(const double* X, const double* Y, int w, int l, d4_t& acc) {
  const int r = l & 15, q = l >> 4;
  const int i = 16 * (w >> 1) + r, j = 16 * (w & 1) + r;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 4 * s + q;
    const double a = TA ? X[k * CLD + i] : X[i * CLD + k];
    const double b = TB ? Y[j * CLD + k] : Y[k * CLD + j];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
}

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
  const long nb = (D + 1 + como::CB - 1) / como::CB;
  const long Dp = nb * como::CB;
  // working copy incl. the appended identity rows (2 Dp x Dp) | factor (Dp x Dp) | inverted diagonal blocks | x accumulator
  return (3 * Dp * Dp + nb * como::CB * como::CB + Dp) * (long)sizeof(double);
}

static int chol_solve_impl(const double* H, const double* g, double* delta, void* workspace, int D, int* info,
                           como_stream_t stream, bool packed) {
  using namespace como;
  if ((!packed && (!H || !g)) || !delta || !workspace || !info || D <= 0 || D > 4000) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nb = (D + 1 + CB - 1) / CB;
  const int Dp = nb * CB;
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
  static const bool lean = [] {                           // COMO_CHOL_LEAN=0: round 3's eight-wave tile factorisation (A/B)
    const char* e = getenv("COMO_CHOL_LEAN");
    return e ? atoi(e) != 0 : true;
  }();
  if (!packed) {                                      // (packed: como_sys_finalize_pack wrote W and reset info)
    hipLaunchKernelGGL(chol_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, H, g, W, D, Dp, info);
    COMO_CHECK_LAUNCH();
  }
  if (Dp > 4096) return COMO_ERR_ARG;
  if (lean) hipLaunchKernelGGL(chol_first2_kernel<true>, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, nb, info);
  else hipLaunchKernelGGL(chol_first2_kernel<false>, dim3(1), dim3(512), 0, s, W, Lw, Iw, Dp, D, nb, info);
  COMO_CHECK_LAUNCH();
  for (int c0 = 0; c0 + 2 < nb; c0 += 2) {               // two block columns per launch
    const int r = nb - (c0 + 2);
    const int tiles = r * (r + 1) / 2;
    const int app = ride ? (c0 + 2) * r : 0;             // appended rows 0 .. c0 + 1 x the r trailing block columns
    if (lean) hipLaunchKernelGGL(chol_panel2_kernel<true>, dim3(tiles + app), dim3(512), 0, s, W, Lw, Iw, Dp, D, c0, nb, info, xacc);
    else hipLaunchKernelGGL(chol_panel2_kernel<false>, dim3(tiles + app), dim3(512), 0, s, W, Lw, Iw, Dp, D, c0, nb, info, xacc);
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

}  // extern "C"
