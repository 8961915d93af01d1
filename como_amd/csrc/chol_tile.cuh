// Tile-level building blocks of the dense SPD solvers (csrc/chol.hip: the multi-launch solver and the small batched systems;
// csrc/cholp.hip: the persistent one-launch solver): 4x4 micro-factor, the lean five-wave factorisation of a 2x2 block of diagonal
// tiles (factor_pair_lean), 32^3 products on the float64 matrix cores.  Reference: como/odom/backend/linear_system.py:101-112.
#pragma once
#include "common.cuh"
#include <type_traits>
#include <utility>

namespace como {

constexpr int CB = 32;        // panel / tile width (in-tile factor/solve latency grows as CB^2 per panel: 32 beats 64)
constexpr int CLD = CB + 1;   // padded LDS leading dimension
constexpr int TSZ = CB * CLD;      // one LDS tile
constexpr int NT2 = 7;             // tiles of LDS used by the column-pair kernels (59 KB)

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

#ifdef COMO_AB_VARIANTS   // round 3's eight-wave tile factorisation (lost to factor_pair_lean, 12.2 -> 9.4 us per pair): measurement builds only
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
#endif

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
// Idle: what waves 5 and 7 -- which have no role in the factorisation of a full pair and only keep the barrier count -- do
// between the barriers: idle(i, s), i = 0 (wave 5) / 1 (wave 7), after barrier s = 0 .. 16, fully unrolled.  It must never block
// for long (everybody waits for it at the next barrier).  The persistent solver (csrc/cholp.hip) prefetches the next pair's
// inputs there.
struct FpNoIdle { __device__ __forceinline__ void operator()(int, int) const {} };
template <class Idle = FpNoIdle>
__device__ __forceinline__ void factor_pair_lean(double* sm, bool has1, int d0, double* __restrict__ Lw, double* __restrict__ Iw,
                                                 int Dp, int D, int* __restrict__ info, double* Ls = nullptr, double* Vs = nullptr,
                                                 Idle&& idle = Idle()) {
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
  else if (!std::is_same<typename std::decay<Idle>::type, FpNoIdle>::value && has1 && (wv == 5 || wv == 7)) {
    const int i = wv == 7 ? 1 : 0;
#pragma unroll
    for (int s = 0; s <= 16; ++s) {
      lds_only_barrier();
      idle(i, s);
    }
  } else {
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

#ifdef COMO_AB_VARIANTS
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
#endif

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

}  // namespace como
