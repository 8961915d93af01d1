// Shared device helpers for the como_amd HIP kernels (gfx950 / CDNA4, wave64).
//
// Mask-feeding arithmetic (pose inverse, rigid transform, projection) must reproduce
// the reference's torch-CPU operation order bit for bit: every multiply/add is
// individually rounded, no FMA contraction ("#pragma clang fp contract(off)"), IEEE
// division (hipcc default).  Everything else may contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define COMO_OK 0
#define COMO_ERR_ARG 1
#define COMO_ERR_LAUNCH 2

#define COMO_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return COMO_ERR_LAUNCH;            \
  } while (0)

namespace como {

constexpr int WAVE = 64;

// ---- layout of the dense solver's workspace, shared by the packing kernels (csrc/ba.hip, csrc/chol.hip) and the solvers ------
// Systems of 2 .. CHOLP_MAX_NP column PAIRS (64 columns: D + 1 <= 2816) are padded to whole pairs -- the persistent one-launch
// solver (csrc/cholp.hip) works on 64 x 64 super-tiles --, larger (and tiny) ones to whole 32-wide block columns.
constexpr int CHOLP_MAX_NP = 44;
// the persistent solver runs up to this many pairs (measured, us persistent / multi-launch: D = 1240 305 / 386, 1500 409 / 556,
// 2000 726 / 854, 2680 1636 / 1502: beyond ~2300 columns the owners' three products per super-tile step lose to the multi-launch panels)
constexpr int CHOLP_RUN_MAX_NP = 34;
constexpr int CHOLP_SYNC_STRIDE = 32;                 // 32-bit words: every counter on its own 128-byte line
__host__ __device__ inline int chol_np(long D) { return (int)((D + 1 + 63) / 64); }
__host__ __device__ inline bool cholp_size_ok(long D) { return chol_np(D) >= 2 && chol_np(D) <= CHOLP_MAX_NP; }
__host__ __device__ inline long chol_dp(long D) { return cholp_size_ok(D) ? 64L * chol_np(D) : ((D + 1 + 31) / 32) * 32; }
// persistent solver: W (2 Dp x Dp) | published pair inverses (np x 64 x 64) | y of the last pair (64) | counters
__host__ __device__ inline long cholp_sync_offset(long Dp, long np) { return 2 * Dp * Dp + np * 4096 + 64; }   // in doubles
__host__ __device__ inline long cholp_sync_words(long np) { return (2 + 2 * np) * CHOLP_SYNC_STRIDE; }
// (called by the packing kernels with their linear thread index: the counters start every solve at zero)
__device__ __forceinline__ void cholp_reset_sync(double* __restrict__ W, long D, long idx) {
  if (!cholp_size_ok(D)) return;
  const long np = chol_np(D), Dp = 64 * np;
  if (idx < cholp_sync_words(np)) ((unsigned*)(W + cholp_sync_offset(Dp, np)))[idx] = 0u;
}

// ---- exact-order primitives (reference geometry/*.py as executed by torch CPU) ----------
template <typename T>
__device__ __forceinline__ T dot3_seq(T a0, T a1, T a2, T x, T y, T z) {
#pragma clang fp contract(off)
  return (a0 * x + a1 * y) + a2 * z;
}

// P' = R P + t with the 3x4 row-major matrix M (rows r0 r1 r2 | t): transforms.py:17-23
template <typename T>
__device__ __forceinline__ void rigid_apply(const T* __restrict__ M, T X, T Y, T Z, T& x, T& y, T& z) {
#pragma clang fp contract(off)
  x = dot3_seq(M[0], M[1], M[2], X, Y, Z) + M[3];
  y = dot3_seq(M[4], M[5], M[6], X, Y, Z) + M[7];
  z = dot3_seq(M[8], M[9], M[10], X, Y, Z) + M[11];
}

// u = fx*X/Z + cx: camera.py:20-26
template <typename T>
__device__ __forceinline__ T project1(T f, T X, T Z, T c) {
#pragma clang fp contract(off)
  return (f * X) / Z + c;
}

// inverse of a 4x4 row-major pose into a 3x4 row-major [R^T | -(R^T t)]: lie_algebra.py:83-93
template <typename T>
__device__ __forceinline__ void invert_pose34(const T* __restrict__ Tm, T* __restrict__ out) {
#pragma clang fp contract(off)
  for (int i = 0; i < 3; ++i) {
    out[i * 4 + 0] = Tm[0 * 4 + i];
    out[i * 4 + 1] = Tm[1 * 4 + i];
    out[i * 4 + 2] = Tm[2 * 4 + i];
    out[i * 4 + 3] = -dot3_seq(Tm[0 * 4 + i], Tm[1 * 4 + i], Tm[2 * 4 + i], Tm[3], Tm[7], Tm[11]);
  }
}

// 1 <= u < W-1 and 1 <= v < H-1 : photo.py:15-21, photo_utils.py:12-18
template <typename T>
__device__ __forceinline__ bool in_image(T u, T v, int H, int W) {
  return (u >= T(1)) && (u < T(W - 1)) && (v >= T(1)) && (v < T(H - 1));
}

// Sample position grid_sample(align_corners=False) really uses after the reference's
// normalize_coordinates: x_norm = (2a) u + a - 1 (a = 1/size in the CALLER's dtype, coords.py:12-20),
// then ATen's CPU unnormalize (x_norm + 1) * (size / 2) - 0.5.  Only sampled VALUES depend on this.
template <typename T>
__device__ __forceinline__ T grid_position(T u, int size, T a) {
#pragma clang fp contract(off)
  T xn = (T(2) * a) * u + a - T(1);
  return (xn + T(1)) * (T(size) / T(2)) - T(0.5);
}

// bilinear taps with zero padding
template <typename T>
struct Taps {
  int i00, i01, i10, i11;   // linear indices (clamped)
  T w00, w01, w10, w11;     // weights (0 where the tap is outside)
};

template <typename T>
__device__ __forceinline__ Taps<T> make_taps(T x, T y, int H, int W) {
  Taps<T> t;
  T xf = floor(x), yf = floor(y);
  T wx = x - xf, wy = y - yf;
  // guard against non-finite coordinates (pixels that are invalid anyway)
  bool fin = (fabs(x) < T(1e9)) && (fabs(y) < T(1e9));
  int x0 = fin ? (int)xf : -4, y0 = fin ? (int)yf : -4;
  int x1 = x0 + 1, y1 = y0 + 1;
  bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x1 >= 0) && (x1 < W);
  bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y1 >= 0) && (y1 < H);
  int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  t.i00 = cy0 * W + cx0; t.i01 = cy0 * W + cx1; t.i10 = cy1 * W + cx0; t.i11 = cy1 * W + cx1;
  t.w00 = (vx0 && vy0) ? (T(1) - wy) * (T(1) - wx) : T(0);
  t.w01 = (vx1 && vy0) ? (T(1) - wy) * wx : T(0);
  t.w10 = (vx0 && vy1) ? wy * (T(1) - wx) : T(0);
  t.w11 = (vx1 && vy1) ? wy * wx : T(0);
  return t;
}

template <typename T>
__device__ __forceinline__ T tap_sum(const T* __restrict__ plane, const Taps<T>& t) {
  return t.w00 * plane[t.i00] + t.w01 * plane[t.i01] + t.w10 * plane[t.i10] + t.w11 * plane[t.i11];
}

// Huber weight: robust_loss.py:9-16
template <typename T>
__device__ __forceinline__ T huber(T x) {
  T a = fabs(x);
  return (a < T(1.345)) ? T(1) : T(1.345) / a;
}

// ---- order-preserving integer keys of |r| for the exact median ---------------------------
__device__ __forceinline__ uint32_t abs_key(float r) { return __float_as_uint(fabsf(r)); }
__device__ __forceinline__ uint64_t abs_key(double r) { return (uint64_t)__double_as_longlong(fabs(r)); }
__device__ __forceinline__ float key_value(uint32_t k) { return __uint_as_float(k); }
__device__ __forceinline__ double key_value(uint64_t k) { return __longlong_as_double((long long)k); }

template <typename T> struct KeyOf;
template <> struct KeyOf<float> { using type = uint32_t; static constexpr int BITS = 32; };
template <> struct KeyOf<double> { using type = uint64_t; static constexpr int BITS = 64; };

// ---- wave / block reductions ----------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- exact, order-independent accumulation ------------------------------------------------------
// A double v is split into floor(v) (int64) and (v - floor(v)) * 2^56 (uint64) and the two parts are added with integer
// atomics: integer addition is associative, so the sum does not depend on the order in which workgroups, streams or ranks
// deliver their contributions (floating-point atomics do).  Range |v| < 2^62, resolution 2^-56 = 1.4e-17 absolute (the
// entries of the normal equations are >> 1e-10); up to 256 contributions per entry before the fraction word could wrap.
constexpr double FIX_SCALE = 72057594037927936.0;            // 2^56
constexpr int FIX_ERR_SLOTS = 8;
constexpr int FIX_POISON = 7;                                // err slot counting non-finite contributions

__device__ __forceinline__ void fix_split(double v, long long& hi, unsigned long long& lo) {
  const double f = floor(v);
  hi = (long long)f;
  lo = (unsigned long long)((v - f) * FIX_SCALE);            // v - f in [0, 1] exactly; scaling by 2^56 is exact
}
__device__ __forceinline__ double fix_value(long long hi, unsigned long long lo) {
  return (double)hi + (double)lo * (1.0 / FIX_SCALE);
}
// planes: fix[idx] (integer parts), fix[plane + idx] (fractions); poison = &fix[D*D + D + FIX_POISON]
__device__ __forceinline__ void fix_add(long long* __restrict__ fix, long plane, long idx, double v, long long* __restrict__ poison) {
  if (!(fabs(v) < 4.0e18)) {                                 // NaN / inf / overflow: flag it, the finalize pass poisons H
    atomicAdd((unsigned long long*)poison, 1ull);
    return;
  }
  long long hi;
  unsigned long long lo;
  fix_split(v, hi, lo);
  if (hi) atomicAdd((unsigned long long*)&fix[idx], (unsigned long long)hi);
  if (lo) atomicAdd((unsigned long long*)&fix[plane + idx], lo);
}

// ---- SE(3) exponential ------------------------------------------------------------------------
// T = Exp(xi) for COMO's tangent ordering xi = [omega (0:3), v (3:6)] (rotation first):
//   R = I + a [w]x + b [w]x^2,  t = (I + b [w]x + c [w]x^2) v,  a = sin(th)/th, b = (1-cos th)/th^2, c = (th-sin th)/th^3.
// The reference delegates to lietorch's SE3.exp with the vector REORDERED to [tau = v, phi = omega]
// (lie_algebra.py:45-56) -- lietorch is an unpinned third-party dependency, so this closed form is pinned against
// scipy.linalg.expm of the 4x4 twist matrix instead (tests/golden/se3_expm.npz, make_golden_r2.py::se3_case).
// ONE definition: win_update (T <- T Exp(delta)) and track_finish (T <- T Exp(-delta)) both call it.
__device__ inline void se3_exp_f64(const double* xi, double* Tm) {
  const double wx = xi[0], wy = xi[1], wz = xi[2];
  const double th2 = wx * wx + wy * wy + wz * wz;
  double a, b, c;
  if (th2 < 0.25) {
    // |omega| < 0.5 (every Gauss-Newton update): the power series in th^2 to th^18 -- truncation < 1e-20, no cancellation
    // (the closed forms (1 - cos th) / th^2 and (th - sin th) / th^3 lose digits for small th) and no f64 sin / cos
    const double x = th2;
    a = 1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 + x * (1.0 / 6227020800.0
        + x * (-1.0 / 1307674368000.0 + x * (1.0 / 355687428096000.0))))))));
    b = 0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 + x * (1.0 / 87178291200.0
        + x * (-1.0 / 20922789888000.0 + x * (1.0 / 6402373705728000.0))))))));
    c = 1.0 / 6 + x * (-1.0 / 120 + x * (1.0 / 5040 + x * (-1.0 / 362880 + x * (1.0 / 39916800 + x * (-1.0 / 6227020800.0
        + x * (1.0 / 1307674368000.0 + x * (-1.0 / 355687428096000.0 + x * (1.0 / 121645100408832000.0))))))));
  } else {
    const double th = sqrt(th2);
    a = sin(th) / th; b = (1.0 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th);
  }
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W2[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int j = 0; j < 3; ++j) {
      const double e = (i == j) ? 1.0 : 0.0;
      Tm[i * 4 + j] = e + a * W[i * 3 + j] + b * W2[i * 3 + j];
      t += (e + b * W[i * 3 + j] + c * W2[i * 3 + j]) * xi[3 + j];
    }
    Tm[i * 4 + 3] = t;
  }
  Tm[12] = Tm[13] = Tm[14] = 0.0;
  Tm[15] = 1.0;
}

}  // namespace como
