// Element-wise glue of a keyframe insertion, fused (round 6, third part).  Each of these was a chain of 4 .. 12 torch launches of
// ~4.5 us on tensors of 64 .. 300 k elements -- a keyframe frame of the sequential loop is bound by the NUMBER of its dispatches
// (439, of which ~330 under 6 us: DESIGN.md section 4.10), not by bytes or flops.  Every kernel applies the reference's operations
// in the reference's order, one rounding per torch op (no contraction), so the results are the torch chains' bit for bit; nothing
// here reduces in a different order (the one reduction, a minimum, is exact in any order).
//
//   predictor_sinv : depth_cov/core/distill_depth.py:42-46 of the reference -- the smallest conditional variance (over the rows
//                    that count) shifts every variance positive: var += min(var) + 1e-8, stdev_inv = 1 / sqrt(var).
//   distill_prep   : distill_depth.py:96-111, 152-166 -- the validity test of the observations as zero weights:
//                    ok = z > min_depth [& mask], y = log(ok ? z : 1), w = ok ? stdev_inv^2 : 0 (or ok as 0 / 1).
//   corr_good      : odom/frontend/corr.py:47-59, 113-118 -- max(|log z_a - log z_b|, |log z_c - log z_d|) < t1 & grad < t2.
//   normalize_coords : utils/coords.py:12-15 -- x_norm = 2 A x + A - 1 per coordinate (optionally with x / y swapped: :5-6).
//   grad_mag       : corr.py:95-96 -- sqrt(gx^2 + gy^2).
//   aff            : geometry/affine_brightness.py:5-16 -- composition / relative affine brightness parameters.
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

constexpr int KG_PARTS = 64;

// min with torch.min's NaN propagation
__device__ __forceinline__ double kg_min(double m, double v) { return (v < m || v != v) ? v : m; }

__global__ __launch_bounds__(256) void kg_min_partial_kernel(const double* __restrict__ var, const uint8_t* __restrict__ mask, long n,
                                                             double* __restrict__ part) {
  __shared__ double red[4];
  double m = __builtin_inf();
  bool nan = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (mask && !mask[i]) continue;
    const double v = var[i];
    nan = nan || (v != v);
    m = v < m ? v : m;
  }
  if (nan) m = __builtin_nan("");
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = kg_min(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = kg_min(kg_min(red[0], red[1]), kg_min(red[2], red[3]));
}

__global__ __launch_bounds__(256) void kg_sinv_kernel(const double* __restrict__ var, long n, const double* __restrict__ part, int nparts,
                                                      double* __restrict__ sinv) {
#pragma clang fp contract(off)
  double m = part[0];
  for (int k = 1; k < nparts; ++k) m = kg_min(m, part[k]);
  const double shift = m + 1e-8;                                 // (vmin + 1e-8): one rounding, as the 0-dim tensor op
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sinv[i] = 1.0 / sqrt(var[i] + shift);               // reciprocal(sqrt(var + shift)) (x 1.0: exact)
}

__global__ __launch_bounds__(256) void kg_distill_prep_kernel(const double* __restrict__ z, long zstride, const uint8_t* __restrict__ mask, long n,
                                                              double min_depth, const double* __restrict__ sinv, double sinv_scalar,
                                                              const double* __restrict__ stdev_dev, int weight_mode,
                                                              uint8_t* __restrict__ okm, double* __restrict__ zs,
                                                              double* __restrict__ y, double* __restrict__ w) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double zi = z[i * zstride];
  const bool ok = (zi > min_depth) && (!mask || mask[i]);
  const double zz = ok ? zi : 1.0;
  okm[i] = ok ? 1 : 0;
  if (zs) zs[i] = zz;
  y[i] = log(zz);
  if (weight_mode == 0) {
    w[i] = ok ? 1.0 : 0.0;
  } else {
    // (a fixed observation stdev that lives on the device -- the residual spread of an earlier distillation -- is inverted here:
    // 1.0 / stdev, the reference's (1.0 / stdev_obs) * ones)
    const double s = sinv ? sinv[i] : (stdev_dev ? 1.0 / stdev_dev[0] : sinv_scalar);
    w[i] = ok ? s * s : 0.0;                                     // (a select: a masked row may hold anything)
  }
}

__global__ __launch_bounds__(256) void kg_corr_good_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                           const double* __restrict__ c, const double* __restrict__ d, int stride,
                                                           const double* __restrict__ grad, long m, double corr_thresh,
                                                           double grad_thresh, uint8_t* __restrict__ good) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const double e1 = fabs(log(a[i * stride]) - log(b[i * stride]));
  const double e2 = fabs(log(c[i * stride]) - log(d[i * stride]));
  // torch.maximum propagates NaN; (NaN < t) is false either way
  const double e = (e1 != e1 || e2 != e2) ? __builtin_nan("") : (e1 > e2 ? e1 : e2);
  good[i] = ((e < corr_thresh) && (grad[i] < grad_thresh)) ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void kg_normalize_coords_kernel(const T* __restrict__ x, long n2, const T* __restrict__ A,
                                                                  const T* __restrict__ A2, T* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  const int k = (int)(i & 1);
  T t = A2[k] * x[i];
  t = t + A[k];
  out[i] = t - T(1);
}

// out[i ^ 1] instead of out[i]: the normalised coordinates with x and y swapped (swap_coords_xy of the result: a flip launch)
template <typename T>
__global__ __launch_bounds__(256) void kg_normalize_coords_swap_kernel(const T* __restrict__ x, long n2, const T* __restrict__ A,
                                                                       const T* __restrict__ A2, T* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  const int k = (int)(i & 1);
  T t = A2[k] * x[i];
  t = t + A[k];
  out[i ^ 1] = t - T(1);
}

// sqrt(gx * gx + gy * gy), one rounding per operation (corr.py:95-96 of the reference: the depth-discontinuity measure)
template <typename T>
__global__ __launch_bounds__(256) void kg_grad_mag_kernel(const T* __restrict__ gx, const T* __restrict__ gy, long n, T* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T a = gx[i] * gx[i];
  const T b = gy[i] * gy[i];
  out[i] = sqrt(a + b);
}

// affine brightness parameters (a, b) of B frames (geometry/affine_brightness.py:5-16 of the reference):
//   mode 0, get_aff_w_curr: (a_w + a_c, b_w + b_c exp(a_c));   mode 1, get_rel_aff: (a_1 - a_2, exp(-(a_1 - a_2)) (b_1 - b_2))
template <typename T>
__global__ void kg_aff_kernel(const T* __restrict__ p, const T* __restrict__ q, int B, int mode, T* __restrict__ out) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T p0 = p[2 * b], p1 = p[2 * b + 1], q0 = q[2 * b], q1 = q[2 * b + 1];
  if (mode == 0) {
    out[2 * b] = p0 + q0;
    const T t = q1 * exp(q0);
    out[2 * b + 1] = p1 + t;
  } else {
    const T r0 = p0 - q0;
    out[2 * b] = r0;
    const T e = exp(-r0);
    out[2 * b + 1] = e * (p1 - q1);
  }
}

// The small system of the conditional distillation (distill_depth.py:122-148 of the reference, normal-equation form): the known
// columns' coefficient vector c = [log z_1 ; 0] and  A22 = AtA[m1:, m1:] + sp2 I,  b2 = Atb[m1:] + sp2 s  (s: a device scalar).
__global__ __launch_bounds__(256) void kg_cond_c_kernel(const double* __restrict__ z1, int m1, int mp, double* __restrict__ c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < mp) c[i] = i < m1 ? log(z1[i]) : 0.0;
}
__global__ __launch_bounds__(256) void kg_cond_system_kernel(const double* __restrict__ AtA, const double* __restrict__ Atb, int ld,
                                                             int m1, int m2, double sp2, const double* __restrict__ s_med,
                                                             double* __restrict__ A22, double* __restrict__ b2) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < m2 * m2) {
    const int i = e / m2, j = e - i * m2;
    const double d = (i == j) ? sp2 * 1.0 : sp2 * 0.0;           // sp2 * eye
    A22[e] = AtA[(long)(m1 + i) * ld + (m1 + j)] + d;
  }
  if (e < m2) {
    const double t = sp2 * s_med[0];
    b2[e] = Atb[m1 + e] + t;
  }
}

// Unbiased standard deviation of the entries of `res` whose mask byte is set (torch.std of the gathered valid residuals,
// distill_depth.py:113-116 of the reference): count, mean, then the centred squares -- ONE workgroup, two passes over n <= a few
// 100 k values, every sum in a fixed order (per-thread strided partials, wave tree, 16 waves in sequence): deterministic; equal to
// torch's masked-sum form up to the rounding of another summation order (as that form was to the gathered one).
__global__ __launch_bounds__(1024) void kg_masked_std_kernel(const double* __restrict__ res, const uint8_t* __restrict__ okm, long n,
                                                             double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double red[2][16];
  __shared__ double bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double cnt = 0.0, s = 0.0;
  for (long i = tid; i < n; i += 1024)
    if (okm[i]) { cnt += 1.0; s += res[i]; }
  cnt = wave_sum(cnt);
  s = wave_sum(s);
  if (lane == 0) { red[0][wv] = cnt; red[1][wv] = s; }
  __syncthreads();
  if (tid == 0) {
    double c = 0.0, t = 0.0;
    for (int w = 0; w < 16; ++w) { c += red[0][w]; t += red[1][w]; }
    bc[0] = c;
    bc[1] = t / c;
  }
  __syncthreads();
  const double nn = bc[0], mean = bc[1];
  double ss = 0.0;
  for (long i = tid; i < n; i += 1024)
    if (okm[i]) { const double d = res[i] - mean; ss += d * d; }
  ss = wave_sum(ss);
  __syncthreads();
  if (lane == 0) red[0][wv] = ss;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[0][w];
    out[0] = sqrt(t / (nn - 1.0));
  }
}

template <typename T>
static int grad_mag(const T* gx, const T* gy, long n, T* out, hipStream_t s) {
  if (!gx || !gy || !out || n < 0) return COMO_ERR_ARG;
  if (n == 0) return COMO_OK;
  hipLaunchKernelGGL(kg_grad_mag_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gx, gy, n, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

template <typename T>
static int aff_op(const T* p, const T* q, int B, int mode, T* out, hipStream_t s) {
  if (!p || !q || !out || B <= 0 || mode < 0 || mode > 1) return COMO_ERR_ARG;
  hipLaunchKernelGGL(kg_aff_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, p, q, B, mode, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

template <typename T>
static int normalize_coords_swap(const T* x, long n2, const T* A, const T* A2, T* out, hipStream_t s) {
  if (!x || !A || !A2 || !out || n2 < 0 || (n2 & 1) || x == out) return COMO_ERR_ARG;
  if (n2 == 0) return COMO_OK;
  hipLaunchKernelGGL(kg_normalize_coords_swap_kernel<T>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, x, n2, A, A2, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

template <typename T>
static int normalize_coords(const T* x, long n2, const T* A, const T* A2, T* out, hipStream_t s) {
  if (!x || !A || !A2 || !out || n2 < 0 || (n2 & 1)) return COMO_ERR_ARG;
  if (n2 == 0) return COMO_OK;
  hipLaunchKernelGGL(kg_normalize_coords_kernel<T>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, x, n2, A, A2, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_kf_predictor_sinv_f64(const double* var_n, const uint8_t* row_mask, long n, double* partial, double* sinv,
                               como_stream_t stream) {
  using namespace como;
  if (!var_n || !partial || !sinv || n <= 0) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long g = (n + 255) / 256;
  if (g > KG_PARTS) g = KG_PARTS;
  hipLaunchKernelGGL(kg_min_partial_kernel, dim3((unsigned)g), dim3(256), 0, s, var_n, row_mask, n, partial);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(kg_sinv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, var_n, n, partial, (int)g, sinv);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_kf_distill_prep_f64(const double* z_obs, long z_stride, const uint8_t* obs_mask, long n, double min_depth, const double* sinv,
                             double sinv_scalar, const double* stdev_dev, int weight_mode, uint8_t* okm, double* zs, double* y,
                             double* w, como_stream_t stream) {
  using namespace como;
  if (!z_obs || z_stride < 1 || !okm || !y || !w || n <= 0 || weight_mode < 0 || weight_mode > 1) return COMO_ERR_ARG;
  hipLaunchKernelGGL(kg_distill_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z_obs, z_stride, obs_mask, n,
                     min_depth, sinv, sinv_scalar, stdev_dev, weight_mode, okm, zs, y, w);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_kf_corr_good_f64(const double* a, const double* b, const double* c, const double* d, int stride, const double* grad, long m,
                          double corr_thresh, double grad_thresh, uint8_t* good, como_stream_t stream) {
  using namespace como;
  if (!a || !b || !c || !d || !grad || !good || m < 0 || stride < 1) return COMO_ERR_ARG;
  if (m == 0) return COMO_OK;
  hipLaunchKernelGGL(kg_corr_good_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, d, stride, grad,
                     m, corr_thresh, grad_thresh, good);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_kf_normalize_coords_f32(const float* x, long n2, const float* A, const float* A2, float* out, como_stream_t stream) {
  return como::normalize_coords<float>(x, n2, A, A2, out, (hipStream_t)stream);
}
int como_kf_normalize_coords_f64(const double* x, long n2, const double* A, const double* A2, double* out, como_stream_t stream) {
  return como::normalize_coords<double>(x, n2, A, A2, out, (hipStream_t)stream);
}

int como_kf_masked_std_f64(const double* res, const uint8_t* okm, long n, double* out, como_stream_t stream) {
  using namespace como;
  if (!res || !okm || !out || n <= 0) return COMO_ERR_ARG;
  hipLaunchKernelGGL(kg_masked_std_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, res, okm, n, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_kf_cond_c_f64(const double* z1, int m1, int mp, double* c, como_stream_t stream) {
  using namespace como;
  if (!c || m1 < 0 || mp < m1 || mp <= 0 || (m1 > 0 && !z1)) return COMO_ERR_ARG;
  hipLaunchKernelGGL(kg_cond_c_kernel, dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z1, m1, mp, c);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_kf_cond_system_f64(const double* AtA, const double* Atb, int ld, int m1, int m2, double sp2, const double* s_med,
                            double* A22, double* b2, como_stream_t stream) {
  using namespace como;
  if (!AtA || !Atb || !s_med || !A22 || !b2 || m1 < 0 || m2 <= 0 || ld < m1 + m2) return COMO_ERR_ARG;
  hipLaunchKernelGGL(kg_cond_system_kernel, dim3((unsigned)((m2 * m2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, AtA, Atb, ld, m1,
                     m2, sp2, s_med, A22, b2);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_kf_normalize_coords_swap_f32(const float* x, long n2, const float* A, const float* A2, float* out, como_stream_t stream) {
  return como::normalize_coords_swap<float>(x, n2, A, A2, out, (hipStream_t)stream);
}
int como_kf_normalize_coords_swap_f64(const double* x, long n2, const double* A, const double* A2, double* out, como_stream_t stream) {
  return como::normalize_coords_swap<double>(x, n2, A, A2, out, (hipStream_t)stream);
}
int como_kf_grad_mag_f32(const float* gx, const float* gy, long n, float* out, como_stream_t stream) {
  return como::grad_mag<float>(gx, gy, n, out, (hipStream_t)stream);
}
int como_kf_grad_mag_f64(const double* gx, const double* gy, long n, double* out, como_stream_t stream) {
  return como::grad_mag<double>(gx, gy, n, out, (hipStream_t)stream);
}
int como_kf_aff_f32(const float* p, const float* q, int B, int mode, float* out, como_stream_t stream) {
  return como::aff_op<float>(p, q, B, mode, out, (hipStream_t)stream);
}
int como_kf_aff_f64(const double* p, const double* q, int B, int mode, double* out, como_stream_t stream) {
  return como::aff_op<double>(p, q, B, mode, out, (hipStream_t)stream);
}

}  // extern "C"
