// Inverse-compositional photometric tracking: one Gauss-Newton iteration on the device.
//
// Reference path (gray, c = 1): como/odom/frontend/photo_tracking.py:117-143 (`tracking_iter`)
//   transform_project (geometry/camera.py:57-68) -> img_interp (frontend/photo_utils.py:9-31)
//   -> affine residual, sigma = 1.4826 median|r| -> robustify_photo (:77-93) -> solve_delta (:96-99)
//   -> update_pose_ic (:103-114).
//
// Kernel chain (no host synchronisation, everything on `stream`):
//   track_residual  : warp + bilinear sample + residual + validity mask, pass-0 histogram of |r|
//   select_hist x(P-1): remaining radix-select digit passes (exact median)
//   track_reduce    : Huber weights, 8x8 J^T W J / J^T W r partials (wave shuffle + LDS reduction)
//   track_finish    : ordered sum of partials, 8x8 Cholesky solve, T <- T Exp(-delta), affine update
// HBM-bound: 53 B per pixel-iteration (unit A).
#include "select.cuh"
#include "../../include/como_hip.h"

namespace como {

template <typename T> int select_hist(const T*, const uint8_t*, long, int, uint32_t*, int, hipStream_t);

constexpr int TRK_ACC = 46;   // 36 (H upper) + 8 (g) + err + spare
constexpr int TRK_MAX_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void track_residual_kernel(
    const T* __restrict__ Tji, const T* __restrict__ Kmat, const T* __restrict__ aff, const T* __restrict__ P,
    const T* __restrict__ vals_i, const T* __restrict__ img, int H, int W, long N, T* __restrict__ J8,
    T* __restrict__ r_out, uint8_t* __restrict__ valid_out, T* __restrict__ pj_out, T* __restrict__ depth_out,
    uint32_t* __restrict__ hists, const uint8_t* __restrict__ in_mask) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) lh[b] = 0;
  // Pmat = K @ T[0:3, :] with the sequential-k order of torch's small-matmul kernel (camera.py:58)
  T Pm[12];
  {
#pragma clang fp contract(off)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j)
        Pm[i * 4 + j] = dot3_seq(Kmat[i * 3 + 0], Kmat[i * 3 + 1], Kmat[i * 3 + 2], Tji[0 * 4 + j], Tji[1 * 4 + j], Tji[2 * 4 + j]);
  }
  const T ax = T(1) / T(W), ay = T(1) / T(H);     // A_norm, photo_tracking.py:154-156
  const T ea = exp(-aff[0]), bb = aff[1];
  __syncthreads();
  const long stride = (long)gridDim.x * 256;
  const long iters = (N + stride - 1) / stride;               // uniform trip count (the aggregated histogram needs whole waves)
  for (long it = 0; it < iters; ++it) {
    const long i0 = it * stride + (long)blockIdx.x * 256 + threadIdx.x;
    const bool inr = i0 < N;
    const long i = inr ? i0 : N - 1;
    const T X = P[3 * i + 0], Y = P[3 * i + 1], Z = P[3 * i + 2];
    T hx, hy, hz;
    rigid_apply(Pm, X, Y, Z, hx, hy, hz);          // p_h = A P + b
    const T u = hx / hz, v = hy / hz;              // coords = p_h[:2] / depth
    // in_mask: the caller's reference-side selection (photo_tracking.py:20-26 gathers vals/P/dI_dT with it); a point that
    // is masked out behaves exactly like one that projects outside the image
    const bool ok = in_image(u, v, H, W) && (hz > T(0)) && (in_mask == nullptr || in_mask[i] != 0);
    Taps<T> t = make_taps(grid_position(u, W, ax), grid_position(v, H, ay), H, W);
    const T It = tap_sum(img, t);
    const T tmp = ea * It;                          // photo_tracking.py:124
    const T r = (tmp + bb) - vals_i[i];
    if (inr) {
      J8[8 * i + 6] = -tmp;                         // dI_dT[..., 6] = -tmp (in-place, photo_tracking.py:125)
      r_out[i] = r;
      valid_out[i] = ok ? 1 : 0;
      if (pj_out) { pj_out[2 * i] = u; pj_out[2 * i + 1] = v; }
      if (depth_out) depth_out[i] = hz;
    }
    sel_lds_add(lh, sel_digit<KeyT>(abs_key(r), 0), inr && ok);
  }
  __syncthreads();
  sel_flush(lh, hists);
}

template <typename T>
__global__ __launch_bounds__(256) void track_reduce_kernel(const T* __restrict__ J8, const T* __restrict__ r_in,
                                                           const uint8_t* __restrict__ valid, long N,
                                                           const uint32_t* __restrict__ hists,
                                                           double* __restrict__ partials) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ SelScratch sc;
  __shared__ double red[4][TRK_ACC];
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);
  const T info_sqrt = T(1) / sigma;
  T acc[TRK_ACC];
#pragma unroll
  for (int k = 0; k < TRK_ACC; ++k) acc[k] = T(0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
    if (!valid[i]) continue;                        // weight[invalid] = 0 (photo_tracking.py:80-81): contributes exactly
                                                    // zero, also when a masked-out point carries non-finite J / r
    const T r = r_in[i];
    const T wr = r * info_sqrt;
    const T w = huber(wr);
    T J[8];
    if (sizeof(T) == 4) {
      const float4 a = *reinterpret_cast<const float4*>(&J8[8 * i]);
      const float4 b = *reinterpret_cast<const float4*>(&J8[8 * i + 4]);
      J[0] = a.x; J[1] = a.y; J[2] = a.z; J[3] = a.w; J[4] = b.x; J[5] = b.y; J[6] = b.z; J[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) J[k] = J8[8 * i + k];
    }
    int q = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const T wa = w * J[a];
#pragma unroll
      for (int b = a; b < 8; ++b) acc[q++] += wa * J[b];   // H = sum (w J)^T J
      acc[36 + a] += wa * r;                               // g = sum w J r (unwhitened r)
    }
    acc[44] += w * wr * wr;                                // total_err
  }
  // block reduction through LDS in a fixed order: column k of the 256 x 46 tile is summed by 4 threads (64 rows each, fp64),
  // then combined.  (46 wave_sum() calls on doubles cost ~11 us per block: cross-lane f64 moves are two ds_bpermute each.)
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  T* tile = reinterpret_cast<T*>(dyn_lds);                 // [256][TRK_ACC + 1]
#pragma unroll
  for (int k = 0; k < TRK_ACC; ++k) tile[threadIdx.x * (TRK_ACC + 1) + k] = acc[k];
  __syncthreads();
  {
    const int k = threadIdx.x & 63, pp = threadIdx.x >> 6;
    if (k < TRK_ACC) {
      double s = 0;
      for (int rr = 0; rr < 64; ++rr) s += (double)tile[(pp * 64 + rr) * (TRK_ACC + 1) + k];
      red[pp][k] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < TRK_ACC)
    partials[(long)blockIdx.x * TRK_ACC + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// out layout (T): [0:64) H | [64:72) g | [72:80) delta | [80:96) T_new | [96:98) aff_new |
//                 98 mse | 99 grad_norm | 100 total_err | 101 sigma | 102 nvalid | 103 delta_norm | 104 chol_info
template <typename T>
__global__ __launch_bounds__(256) void track_finish_kernel(const double* __restrict__ partials, int nblocks,
                                                           const uint32_t* __restrict__ hists,
                                                           const T* __restrict__ Tji, const T* __restrict__ aff,
                                                           T* __restrict__ out) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ SelScratch sc;
  __shared__ double tot[TRK_ACC];
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  {
    // fixed-order sum of the block partials, 4 interleaved parts x 8 loads in flight (one dependent load per partial made
    // this single-workgroup kernel 250 us at 1200 partials)
    __shared__ double part[4][64];
    const int a = threadIdx.x & 63, pp = threadIdx.x >> 6;
    double s = 0;
    if (a < TRK_ACC) {
      int b = pp;
      for (; b + 28 < nblocks; b += 32) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = partials[(long)(b + 4 * q) * TRK_ACC + a];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
      }
      for (; b < nblocks; b += 4) s += partials[(long)b * TRK_ACC + a];
    }
    part[pp][a] = s;
    __syncthreads();
    if (threadIdx.x < TRK_ACC) tot[threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double Hm[64], g[8], L[64], y[8], d[8];
    int q = 0;
    for (int a = 0; a < 8; ++a)
      for (int b = a; b < 8; ++b) { Hm[a * 8 + b] = tot[q]; Hm[b * 8 + a] = tot[q]; ++q; }
    double gn = 0;
    for (int a = 0; a < 8; ++a) { g[a] = tot[36 + a]; gn += g[a] * g[a]; }
    // Cholesky (lower), errors reported not raised (cholesky_ex(check_errors=False), photo_tracking.py:97)
    int info = 0;
    for (int i = 0; i < 64; ++i) L[i] = 0;
    for (int j = 0; j < 8; ++j) {
      double s = Hm[j * 8 + j];
      for (int k = 0; k < j; ++k) s -= L[j * 8 + k] * L[j * 8 + k];
      if (!(s > 0) && info == 0) info = j + 1;
      const double dj = sqrt(s);
      L[j * 8 + j] = dj;
      for (int i = j + 1; i < 8; ++i) {
        double t = Hm[i * 8 + j];
        for (int k = 0; k < j; ++k) t -= L[i * 8 + k] * L[j * 8 + k];
        L[i * 8 + j] = t / dj;
      }
    }
    for (int i = 0; i < 8; ++i) { double t = g[i]; for (int k = 0; k < i; ++k) t -= L[i * 8 + k] * y[k]; y[i] = t / L[i * 8 + i]; }
    for (int i = 7; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < 8; ++k) t -= L[k * 8 + i] * d[k]; d[i] = t / L[i * 8 + i]; }
    double xi[6], E[16], Tn[16], dn = 0;
    for (int i = 0; i < 6; ++i) xi[i] = -d[i];
    for (int i = 0; i < 8; ++i) dn += d[i] * d[i];
    se3_exp_f64(xi, E);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double t = 0;
        for (int k = 0; k < 4; ++k) t += (double)Tji[i * 4 + k] * E[k * 4 + j];
        Tn[i * 4 + j] = t;
      }
    for (int i = 0; i < 64; ++i) out[i] = (T)Hm[i];
    for (int i = 0; i < 8; ++i) { out[64 + i] = (T)g[i]; out[72 + i] = (T)d[i]; }
    for (int i = 0; i < 16; ++i) out[80 + i] = (T)Tn[i];
    out[96] = (T)((double)aff[0] - d[6]);
    out[97] = (T)((double)aff[1] - d[7]);
    out[98] = (T)(tot[44] / (double)nv);
    out[99] = (T)sqrt(gn);
    out[100] = (T)tot[44];
    out[101] = T(1.4826) * key_value(prefix);
    out[102] = (T)nv;
    out[103] = (T)sqrt(dn);
    out[104] = (T)info;
  }
}

template <typename T>
int track_iter(const T* Tji, const T* Kmat, const T* aff, const T* P, const T* vals_i, const T* img, int H, int W,
               long N, T* J8, T* r_ws, uint8_t* valid_out, T* pj_out, T* depth_out, void* hists_v, double* partials,
               T* out, const uint8_t* in_mask, hipStream_t s) {
  using KeyT = typename KeyOf<T>::type;
  if (!Tji || !Kmat || !aff || !P || !vals_i || !img || !J8 || !r_ws || !valid_out || !hists_v || !partials || !out ||
      N <= 0 || H < 3 || W < 3)
    return COMO_ERR_ARG;
  uint32_t* hists = (uint32_t*)hists_v;
  if (!zero_words(hists, 6 * SEL_BINS, s)) return COMO_ERR_LAUNCH;
  long blocks = (N + 255) / 256;
  if (blocks > 1024) blocks = 1024;   // bounded: the digit-0 flush serialises per hot bin at the memory-side atomics
  hipLaunchKernelGGL(track_residual_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, Tji, Kmat, aff, P, vals_i, img,
                     H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists, in_mask);
  COMO_CHECK_LAUNCH();
  for (int p = 1; p < SelCfg<KeyT>::NPASS; ++p) {
    int rc = select_hist<T>(r_ws, valid_out, N, 1, hists, p, s);
    if (rc) return rc;
  }
  // ~4 pixels per thread: the 46-value block reduction (shuffles + LDS) is amortised, and track_finish sums fewer partials
  int rblocks = (int)((N + 1023) / 1024);
  if (rblocks < 1) rblocks = 1;
  if (rblocks > TRK_MAX_BLOCKS) rblocks = TRK_MAX_BLOCKS;
  hipLaunchKernelGGL(track_reduce_kernel<T>, dim3(rblocks), dim3(256), 256 * (TRK_ACC + 1) * sizeof(T), s, J8, r_ws, valid_out, N, hists,
                     partials);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(track_finish_kernel<T>, dim3(1), dim3(256), 0, s, partials, rblocks, hists, Tji, aff, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

long como_track_partials_bytes(void) { return (long)como::TRK_MAX_BLOCKS * como::TRK_ACC * (long)sizeof(double); }

int como_track_iter_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                        const float* img, int H, int W, long N, float* J8, float* r_ws, uint8_t* valid_out,
                        float* pj_out, float* depth_out, void* hists, void* partials, float* out, como_stream_t stream) {
  return como::track_iter<float>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                 (double*)partials, out, nullptr, (hipStream_t)stream);
}

int como_track_iter_masked_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                               const float* img, int H, int W, long N, float* J8, float* r_ws, uint8_t* valid_out,
                               float* pj_out, float* depth_out, void* hists, void* partials, float* out,
                               const uint8_t* in_mask, como_stream_t stream) {
  return como::track_iter<float>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                 (double*)partials, out, in_mask, (hipStream_t)stream);
}

int como_track_iter_masked_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                               const double* img, int H, int W, long N, double* J8, double* r_ws, uint8_t* valid_out,
                               double* pj_out, double* depth_out, void* hists, void* partials, double* out,
                               const uint8_t* in_mask, como_stream_t stream) {
  return como::track_iter<double>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                  (double*)partials, out, in_mask, (hipStream_t)stream);
}

int como_track_iter_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                        const double* img, int H, int W, long N, double* J8, double* r_ws, uint8_t* valid_out,
                        double* pj_out, double* depth_out, void* hists, void* partials, double* out,
                        como_stream_t stream) {
  return como::track_iter<double>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                  (double*)partials, out, nullptr, (hipStream_t)stream);
}

}  // extern "C"
