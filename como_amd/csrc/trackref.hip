// Per-frame glue of the tracker around the GN kernels, fused (each was 10-25 tiny torch launches, i.e. pure host time in the
// sequential loop: 1.3 ms + 0.5 ms per frame at 640x480):
//
//   track_reference : a keyframe's reference arrays for one pyramid level from its depth map -- back-projection, transform into
//                     the newest keyframe's frame, projection there, validity mask, inverse-compositional Jacobians.
//                     reference: como/odom/Tracking.py:255-300 (update_kf_reference: backprojection camera.py:43-54,
//                     transform_points transforms.py:17-23, projection camera.py:20-26, the in-image test :265-281,
//                     precalc_jacobians photo_tracking.py:46-74).
//   reproject_depth : the newest keyframe's finest-level points seen from the current frame as a depth image (NaN where nothing
//                     lands; where several points land on one pixel the LAST one wins, the order torch's CPU index_put applies),
//                     plus the number of pixels hit.  reference: como/odom/Tracking.py:163-185 (get_reproj_last_kf) and
//                     como/utils/coords.py:50-56 (fill_image).
// Element-wise, HBM-bound, a few MB per call.  Mask-feeding arithmetic keeps the reference's operation order (no contraction).
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

// depth (b, h*w) of level (h, w); rel (b,4,4) = T_lastkf^-1 T_kf; K (3,3) of the level; dI_dw (b,n,1,2); vals (b,n,1)
template <typename T>
__global__ __launch_bounds__(256) void track_reference_kernel(const T* __restrict__ depth, const T* __restrict__ rel,
                                                              const T* __restrict__ Kmat, const T* __restrict__ dI_dw,
                                                              const T* __restrict__ vals, int h, int w, T border, T depth_thresh,
                                                              T* __restrict__ P_out, uint8_t* __restrict__ mask_out,
                                                              T* __restrict__ J_out) {
  const long n = (long)h * w;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T* M = rel + 16 * (long)b;
  const long bi = (long)b * n + i;
  const T z = depth[bi];
  T X, Y, Z, u, v;
  {
#pragma clang fp contract(off)
    const T rx = (T((int)(i % w)) - cx) / fx;              // camera.py:43-47 with p = (col, row)
    const T ry = (T((int)(i / w)) - cy) / fy;
    const T px = z * rx, py = z * ry, pz = z * T(1);
    // P @ R^T + t (Tracking.py _rigid: one small GEMM per pose; accumulated in k order like the library's K = 3 product)
    X = ((px * M[0] + py * M[1]) + pz * M[2]) + M[3];
    Y = ((px * M[4] + py * M[5]) + pz * M[6]) + M[7];
    Z = ((px * M[8] + py * M[9]) + pz * M[10]) + M[11];
    u = fx * X / Z + cx;                                   // camera.py:20-26
    v = fy * Y / Z + cy;
  }
  P_out[3 * bi] = X; P_out[3 * bi + 1] = Y; P_out[3 * bi + 2] = Z;
  // closed interval grown by `border` and a minimum depth (Tracking.py:265-281)
  const bool ok = (u >= -border) && (u <= T(w - 1) + border) && (v >= -border) && (v <= T(h - 1) + border) && (Z > depth_thresh);
  mask_out[bi] = ok ? 1 : 0;
  // inverse-compositional Jacobian at theta = 0 (photo_tracking.py:46-74), same expressions as precalc_jac_kernel (image.hip)
  const T gx = dI_dw[2 * bi], gy = dI_dw[2 * bi + 1];
  const T a = gx * (fx / Z), bb = gy * (fy / Z);
  const T c = gx * (-(fx * X / Z) / Z) + gy * (-(fy * Y / Z) / Z);
  T* o = J_out + 8 * bi;
  o[0] = bb * (-Z) + c * Y;
  o[1] = a * Z + c * (-X);
  o[2] = a * (-Y) + bb * X;
  o[3] = a;
  o[4] = bb;
  o[5] = c;
  o[6] = vals[bi];
  o[7] = T(1);
}

// The whole reference pyramid of the tracker in ONE launch (Tracking.update_kf_reference, Tracking.py:187-313, with
// depth_interp_mode nearest_neighbor): rel_b = T_lastkf^-1 T_kf_b (se3_compose_kernel mode 1's arithmetic), the depth of level l at
// (y, x) = the finest depth at (y << l, x << l) -- what l applications of pyr_depth's nearest-neighbour pooling (depth_pool2_kernel
// mode 1: out(y, x) = in(2 y, 2 x)) leave -- and per pixel exactly what track_reference_kernel computes.  It replaces a pose
// composition, levels - 1 pooling launches and `levels` reference launches: five dependent launches of >= 4.5 us behind every
// mapping iteration of the sequential loop.
struct TrackRefPyr {
  const float* K[4];
  const float* dI_dw[4];
  const float* vals[4];
  float* P[4];
  uint8_t* mask[4];
  float* J[4];
  int h[4], w[4], shift[4];
  unsigned block0[5];      // first block of every level (prefix sums)
  int levels;
};

__global__ __launch_bounds__(256) void track_reference_pyr_kernel(const float* __restrict__ depth0, int W0, long hw0,
                                                                  const float* __restrict__ poses, int nk, TrackRefPyr A, float border,
                                                                  float depth_thresh) {
  int l = 0;
  while (l + 1 < A.levels && blockIdx.x >= A.block0[l + 1]) ++l;
  const int h = A.h[l], w = A.w[l];
  const long n = (long)h * w;
  const long i = (long)(blockIdx.x - A.block0[l]) * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  float M[12];
  {
#pragma clang fp contract(off)
    const float* pa = poses + 16 * (long)(nk - 1);
    const float* pb = poses + 16 * (long)b;
    float a[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float s = pa[r] * pa[3];
      s = s + pa[4 + r] * pa[7];
      s = s + pa[8 + r] * pa[11];
      a[4 * r + 0] = pa[r]; a[4 * r + 1] = pa[4 + r]; a[4 * r + 2] = pa[8 + r]; a[4 * r + 3] = -s;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = a[4 * r] * pb[c];
        s = s + a[4 * r + 1] * pb[4 + c];
        s = s + a[4 * r + 2] * pb[8 + c];
        s = s + a[4 * r + 3] * pb[12 + c];
        M[4 * r + c] = s;
      }
    }
  }
  const float* Kmat = A.K[l];
  const float fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const long bi = (long)b * n + i;
  const int x = (int)(i % w), y = (int)(i / w);
  const float z = depth0[(long)b * hw0 + ((long)y << A.shift[l]) * W0 + ((long)x << A.shift[l])];
  float X, Y, Z, u, v;
  {
#pragma clang fp contract(off)
    const float rx = (float(x) - cx) / fx;
    const float ry = (float(y) - cy) / fy;
    const float px = z * rx, py = z * ry, pz = z * 1.0f;
    X = ((px * M[0] + py * M[1]) + pz * M[2]) + M[3];
    Y = ((px * M[4] + py * M[5]) + pz * M[6]) + M[7];
    Z = ((px * M[8] + py * M[9]) + pz * M[10]) + M[11];
    u = fx * X / Z + cx;
    v = fy * Y / Z + cy;
  }
  float* P_out = A.P[l];
  P_out[3 * bi] = X; P_out[3 * bi + 1] = Y; P_out[3 * bi + 2] = Z;
  const bool ok = (u >= -border) && (u <= float(w - 1) + border) && (v >= -border) && (v <= float(h - 1) + border) && (Z > depth_thresh);
  A.mask[l][bi] = ok ? 1 : 0;
  const float gx = A.dI_dw[l][2 * bi], gy = A.dI_dw[l][2 * bi + 1];
  const float a_ = gx * (fx / Z), bb = gy * (fy / Z);
  const float c_ = gx * (-(fx * X / Z) / Z) + gy * (-(fy * Y / Z) / Z);
  float* o = A.J[l] + 8 * bi;
  o[0] = bb * (-Z) + c_ * Y;
  o[1] = a_ * Z + c_ * (-X);
  o[2] = a_ * (-Y) + bb * X;
  o[3] = a_;
  o[4] = bb;
  o[5] = c_;
  o[6] = A.vals[l][bi];
  o[7] = 1.0f;
}

// pass 1: every point that projects strictly inside the image with positive depth claims its (truncated) pixel with its index
template <typename T>
__global__ __launch_bounds__(256) void reproject_claim_kernel(const T* __restrict__ Tck, const T* __restrict__ Kmat,
                                                              const T* __restrict__ P, long n, int h, int w,
                                                              long long* __restrict__ order, T* __restrict__ zbuf,
                                                              int* __restrict__ nseen) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *nseen = 0;
  if (i >= n) return;
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  T X, Y, Z, u, v;
  {
#pragma clang fp contract(off)
    const T px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
    X = ((px * Tck[0] + py * Tck[1]) + pz * Tck[2]) + Tck[3];
    Y = ((px * Tck[4] + py * Tck[5]) + pz * Tck[6]) + Tck[7];
    Z = ((px * Tck[8] + py * Tck[9]) + pz * Tck[10]) + Tck[11];
    u = fx * X / Z + cx;
    v = fy * Y / Z + cy;
  }
  zbuf[i] = Z;
  // open interval, depth > 0 (Tracking.py:173-178)
  const bool ok = (u > T(0)) && (u < T(w - 1)) && (v > T(0)) && (v < T(h - 1)) && (Z > T(0));
  if (ok) {
    const long tgt = (long)((long long)v) * w + (long)((long long)u);      // .long(): truncation
    atomicMax((unsigned long long*)&order[tgt], (unsigned long long)(i + 1));   // 0 = empty; the LAST point wins
  }
}

// pass 2: the winners' depths, NaN elsewhere; the claim table is left cleared for the next call
template <typename T>
__global__ __launch_bounds__(256) void reproject_gather_kernel(long long* __restrict__ order, const T* __restrict__ zbuf, long hw,
                                                               T* __restrict__ img, uint8_t* __restrict__ seen,
                                                               int* __restrict__ nseen) {
  // grid-stride over the image, ONE atomic per workgroup: with one per wave the 4800 waves of a 640x480 image queued on the same
  // address (55 us for a 4 MB pass)
  __shared__ int wsum[4];
  int cnt = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
    const long long o = order[i];
    const bool hit = o > 0;
    img[i] = hit ? zbuf[o - 1] : (T)__builtin_nanf("");
    seen[i] = hit ? 1 : 0;
    if (hit) order[i] = 0;
    cnt += hit ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (tot) atomicAdd(nseen, tot);
  }
}

// Points of frame i (row/col coordinates + depths) seen from frame j: back-projection, rigid transform, projection and the
// in-image / minimum-depth test of the new keyframe's correspondence search (reference como/odom/frontend/corr.py:17-43:
// filter_reproj_coords + reproject_points = camera.py:43-54 backprojection, transforms.py:17-23 transform_points, camera.py:20-26
// projection) -- one launch instead of ~26 elementwise torch launches per call (three calls per keyframe, one of them over every
// pixel of the previous keyframe's depth image).  coords == nullptr: the points are the pixel grid of width `wgrid` (row = i / wgrid,
// col = i % wgrid).  keep == nullptr: no test.  Same operation order as the torch expressions (no contraction); the rigid transform
// accumulates in k order.
template <typename T>
__global__ __launch_bounds__(256) void reproject_points_kernel(const T* __restrict__ coords, const T* __restrict__ z,
                                                               const T* __restrict__ Tji, const T* __restrict__ Kmat, long n, int wgrid,
                                                               int h, int w, T min_depth, T* __restrict__ rc_out, T* __restrict__ P_out,
                                                               uint8_t* __restrict__ keep) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  T X, Y, Z, u, v;
  {
#pragma clang fp contract(off)
    const T row = coords ? coords[2 * i] : T((int)(i / wgrid));
    const T col = coords ? coords[2 * i + 1] : T((int)(i % wgrid));
    const T zi = z[i];
    const T rx = (col - cx) / fx, ry = (row - cy) / fy;
    const T px = zi * rx, py = zi * ry, pz = zi * T(1);
    X = ((px * Tji[0] + py * Tji[1]) + pz * Tji[2]) + Tji[3];
    Y = ((px * Tji[4] + py * Tji[5]) + pz * Tji[6]) + Tji[7];
    Z = ((px * Tji[8] + py * Tji[9]) + pz * Tji[10]) + Tji[11];
    u = fx * X / Z + cx;
    v = fy * Y / Z + cy;
  }
  rc_out[2 * i] = v;
  rc_out[2 * i + 1] = u;
  P_out[3 * i] = X; P_out[3 * i + 1] = Y; P_out[3 * i + 2] = Z;
  // at least one pixel inside the image, deeper than min_depth (corr.py:17-29)
  if (keep) keep[i] = ((u >= T(1)) && (u < T(w - 1)) && (v >= T(1)) && (v < T(h - 1)) && (Z > min_depth)) ? 1 : 0;
}

}  // namespace como

extern "C" {

#define COMO_DEF_TRACKREF(SFX, T)                                                                                              \
  int como_track_reference_##SFX(const T* depth, const T* rel, const T* K, const T* dI_dw, const T* vals, int b, int h, int w,    \
                                 T border, T depth_thresh, T* P_out, uint8_t* mask_out, T* J_out, como_stream_t stream) {         \
    if (!depth || !rel || !K || !dI_dw || !vals || !P_out || !mask_out || !J_out || b <= 0 || h <= 0 || w <= 0)                 \
      return COMO_ERR_ARG;                                                                                                      \
    const long n = (long)h * w;                                                                                                 \
    hipLaunchKernelGGL(como::track_reference_kernel<T>, dim3((unsigned)((n + 255) / 256), b), dim3(256), 0, (hipStream_t)stream, \
                       depth, rel, K, dI_dw, vals, h, w, border, depth_thresh, P_out, mask_out, J_out);                          \
    COMO_CHECK_LAUNCH();                                                                                                        \
    return COMO_OK;                                                                                                             \
  }                                                                                                                             \
  int como_reproject_depth_##SFX(const T* Tck, const T* K, const T* P, long n, int h, int w, void* order_ws, T* zbuf, T* img,     \
                                 uint8_t* seen, int* nseen, como_stream_t stream) {                                             \
    if (!Tck || !K || !P || !order_ws || !zbuf || !img || !seen || !nseen || n <= 0 || h <= 0 || w <= 0) return COMO_ERR_ARG;   \
    const long hw = (long)h * w;                                                                                                \
    hipLaunchKernelGGL(como::reproject_claim_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,    \
                       Tck, K, P, n, h, w, (long long*)order_ws, zbuf, nseen);                                                  \
    COMO_CHECK_LAUNCH();                                                                                                        \
    hipLaunchKernelGGL(como::reproject_gather_kernel<T>, dim3((unsigned)((hw + 1023) / 1024 < 512 ? (hw + 1023) / 1024 : 512)),  \
                       dim3(256), 0, (hipStream_t)stream,                                                                        \
                       (long long*)order_ws, (const T*)zbuf, hw, img, seen, nseen);                                              \
    COMO_CHECK_LAUNCH();                                                                                                        \
    return COMO_OK;                                                                                                             \
  }
COMO_DEF_TRACKREF(f32, float)
COMO_DEF_TRACKREF(f64, double)

int como_track_reference_pyr_f32(const float* depth0, int H0, int W0, const float* kf_poses, int nk, int levels, const int* hw,
                                 const float* const* K, const float* const* dI_dw, const float* const* vals, float* const* P_out,
                                 uint8_t* const* mask_out, float* const* J_out, float border, float depth_thresh, como_stream_t stream) {
  if (!depth0 || !kf_poses || !hw || !K || !dI_dw || !vals || !P_out || !mask_out || !J_out || nk <= 0 || levels < 1 || levels > 4 ||
      H0 <= 0 || W0 <= 0)
    return COMO_ERR_ARG;
  como::TrackRefPyr A;
  A.levels = levels;
  unsigned blocks = 0;
  for (int l = 0; l < levels; ++l) {
    // levels come coarse -> fine (DepthPyramidModule's order): level l is the finest depth pooled levels - 1 - l times
    const int sh = levels - 1 - l;
    int h = H0, w = W0;
    for (int k = 0; k < sh; ++k) { h = (h + 1) / 2; w = (w + 1) / 2; }
    if (hw[2 * l] != h || hw[2 * l + 1] != w || !K[l] || !dI_dw[l] || !vals[l] || !P_out[l] || !mask_out[l] || !J_out[l]) return COMO_ERR_ARG;
    A.K[l] = K[l]; A.dI_dw[l] = dI_dw[l]; A.vals[l] = vals[l]; A.P[l] = P_out[l]; A.mask[l] = mask_out[l]; A.J[l] = J_out[l];
    A.h[l] = h; A.w[l] = w; A.shift[l] = sh;
    A.block0[l] = blocks;
    blocks += (unsigned)(((long)h * w + 255) / 256);
  }
  A.block0[levels] = blocks;
  hipLaunchKernelGGL(como::track_reference_pyr_kernel, dim3(blocks, nk), dim3(256), 0, (hipStream_t)stream, depth0, W0, (long)H0 * W0,
                     kf_poses, nk, A, border, depth_thresh);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

#define COMO_DEF_REPROJ_POINTS(SFX, T)                                                                                          \
  int como_reproject_points_##SFX(const T* coords, const T* z, const T* Tji, const T* K, long n, int wgrid, int h, int w,         \
                                  T min_depth, T* rc_out, T* P_out, uint8_t* keep, como_stream_t stream) {                        \
    if (!z || !Tji || !K || !rc_out || !P_out || n <= 0 || (!coords && wgrid <= 0) || h <= 0 || w <= 0) return COMO_ERR_ARG;     \
    hipLaunchKernelGGL(como::reproject_points_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,    \
                       coords, z, Tji, K, n, wgrid, h, w, min_depth, rc_out, P_out, keep);                                        \
    COMO_CHECK_LAUNCH();                                                                                                        \
    return COMO_OK;                                                                                                             \
  }
COMO_DEF_REPROJ_POINTS(f32, float)
COMO_DEF_REPROJ_POINTS(f64, double)

}  // extern "C"
