// Per-frame image operators (gfx950): Scharr gradients, [1 2 1]^2 blur + decimation, max-gradient pixel sub-selection,
// inverse-compositional tracking Jacobians.
//
// Reference: como/utils/image_processing.py:8-44 (ImageGradientModule: Scharr/32, reflect padding), :47-87
// (GaussianBlurModule / ImagePyramidModule: blur then [0::2, 0::2]), como/odom/backend/sparse_map.py:116-142
// (subselect_pixels: max_pool2d(return_indices) of sqrt(gx^2+gy^2)), como/odom/frontend/photo_tracking.py:46-74
// (precalc_jacobians).  All HBM-bound stencils: one thread per output element, rows coalesced.
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// out (N, 3C, H, W) = cat(img, gx, gy) along channels (Mapping.get_img_and_grads layout)
template <typename T>
__global__ __launch_bounds__(256) void img_grads_kernel(const T* __restrict__ img, T* __restrict__ out, int C, int H, int W,
                                                        long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const long nc = i / ((long)W * H);
  const int c = (int)(nc % C);
  const long n = nc / C;
  const T* p = img + nc * H * W;
  const int xm = reflect1(x - 1, W), xp = reflect1(x + 1, W), ym = reflect1(y - 1, H), yp = reflect1(y + 1, H);
  const T tl = p[(long)ym * W + xm], tc = p[(long)ym * W + x], tr = p[(long)ym * W + xp];
  const T ml = p[(long)y * W + xm], mc = p[(long)y * W + x], mr = p[(long)y * W + xp];
  const T bl = p[(long)yp * W + xm], bc = p[(long)yp * W + x], br = p[(long)yp * W + xp];
  const T k3 = T(3) / T(32), k10 = T(10) / T(32);           // the reference's kernel entries (1/32 * {3, 10})
  const T gx = k3 * (tr - tl) + k10 * (mr - ml) + k3 * (br - bl);
  const T gy = k3 * (bl - tl) + k10 * (bc - tc) + k3 * (br - tr);
  const long HW = (long)H * W, pix = (long)y * W + x;
  T* o = out + n * 3 * C * HW;
  o[(long)c * HW + pix] = mc;
  o[(long)(C + c) * HW + pix] = gx;
  o[(long)(2 * C + c) * HW + pix] = gy;
}

// ITU-R 601-2 luma of (N,3,H,W) colour images -> (N,1,H,W): (0.2989 r + 0.587 g) + 0.114 b with every product and sum rounded on
// its own, i.e. the five elementwise launches of the torch expression (utils/image_processing.py rgb_to_grayscale; the reference
// calls torchvision's in Mapping.get_img_and_grads / Tracking.prep_tracking_img, on every frame) in one.
template <typename T>
__global__ __launch_bounds__(256) void rgb_to_gray_kernel(const T* __restrict__ rgb, T* __restrict__ out, long HW, long total) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long n = i / HW, p = i - n * HW;
  const T* c = rgb + n * 3 * HW + p;
  const T r = c[0], g = c[HW], b = c[2 * HW];
  const T s = T(0.2989) * r + T(0.587) * g;
  out[i] = s + T(0.114) * b;
}

// Mapping.get_img_and_grads (Mapping.py:369-379) of ONE gray-mode frame in one launch: luma of the (3,H,W) colour frame (float32 as
// the tracker holds it, widened -- exactly what `.to(float64)` does -- or float64), its Scharr gradients, written as the
// (3,H,W) float64 stack [I | dI/dx | dI/dy] straight into the window's slot, and -- when the per-pixel kernels run in float32 -- the
// rounded float32 copy into the mirror's slot.  The luma of a pixel's eight neighbours is recomputed from the colour planes (the
// same three products and two sums per value as rgb_to_gray_kernel<double>: identical bits wherever it is evaluated), the gradient
// expressions are img_grads_kernel<double>'s: the chain rgb.to(float64) -> rgb_to_grayscale -> ImageGradientModule -> cat ->
// copy into the sliding buffers (+ mirror), ~9 launches on every one-way frame and keyframe, as one.
template <typename TIN>
__global__ __launch_bounds__(256) void frame_stack_kernel(const TIN* __restrict__ rgb, double* __restrict__ out,
                                                          float* __restrict__ out_pix, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long HW = (long)H * W;
  if (i >= HW) return;
  const int x = (int)(i % W), y = (int)(i / W);
  auto luma = [&](int yy, int xx) -> double {
#pragma clang fp contract(off)
    const long q = (long)yy * W + xx;
    const double r = (double)rgb[q], g = (double)rgb[HW + q], b = (double)rgb[2 * HW + q];
    const double s = 0.2989 * r + 0.587 * g;
    return s + 0.114 * b;
  };
  using T = double;
  const int xm = reflect1(x - 1, W), xp = reflect1(x + 1, W), ym = reflect1(y - 1, H), yp = reflect1(y + 1, H);
  const T tl = luma(ym, xm), tc = luma(ym, x), tr = luma(ym, xp);
  const T ml = luma(y, xm), mc = luma(y, x), mr = luma(y, xp);
  const T bl = luma(yp, xm), bc = luma(yp, x), br = luma(yp, xp);
  const T k3 = T(3) / T(32), k10 = T(10) / T(32);           // the reference's kernel entries (1/32 * {3, 10})
  const T gx = k3 * (tr - tl) + k10 * (mr - ml) + k3 * (br - bl);
  const T gy = k3 * (bl - tl) + k10 * (bc - tc) + k3 * (br - tr);
  out[i] = mc;
  out[HW + i] = gx;
  out[2 * HW + i] = gy;
  if (out_pix) {
    out_pix[i] = (float)mc;
    out_pix[HW + i] = (float)gx;
    out_pix[2 * HW + i] = (float)gy;
  }
}

// out (NC, ceil(H/2), ceil(W/2)) = blur(img)[0::2, 0::2]
// the [1 2 1]^2 / 16 stencil value (ONE definition: blur_down_kernel and the fused frame pyramid must round alike)
template <typename T>
__device__ __forceinline__ T blur9(T tl, T tc, T tr, T ml, T mc, T mr, T bl, T bc, T br) {
  const T k1 = T(1) / T(16), k2 = T(2) / T(16), k4 = T(4) / T(16);
  return k1 * (tl + tr + bl + br) + k2 * (tc + ml + mr + bc) + k4 * mc;
}
template <typename T>
__global__ __launch_bounds__(256) void blur_down_kernel(const T* __restrict__ img, T* __restrict__ out, int H, int W, int Ho,
                                                        int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
  const long nc = i / ((long)Wo * Ho);
  const T* p = img + nc * H * W;
  const int x = 2 * xo, y = 2 * yo;
  const int xm = reflect1(x - 1, W), xp = reflect1(x + 1, W), ym = reflect1(y - 1, H), yp = reflect1(y + 1, H);
  out[i] = blur9<T>(p[(long)ym * W + xm], p[(long)ym * W + x], p[(long)ym * W + xp], p[(long)y * W + xm], p[(long)y * W + x],
                    p[(long)y * W + xp], p[(long)yp * W + xm], p[(long)yp * W + x], p[(long)yp * W + xp]);
}

// The tracker's image pyramid of ONE colour frame in one launch (Tracking.prep_tracking_img, como/odom/Tracking.py:103-107:
// rgb_to_grayscale + ImagePyramidModule with three levels): gray (H,W), level 1 = blur_down(gray), level 2 = blur_down(level 1).
// One workgroup per 16 x 16 tile of level 1: every thread evaluates one (or two) of the 17 x 17 level-1 values the tile's level-2
// stencils touch -- the luma values under them re-evaluated from the colour planes instead of waiting for a neighbour -- writes the
// ones the tile owns together with their 2 x 2 luma blocks, and 64 threads form the level-2 pixels from LDS: the same expressions on
// the same inputs (rgb_luma / blur9) give the same bits wherever they are evaluated, so the three images equal the three-launch
// chain's (tested).  (A first form -- a thread per level-1 pixel re-evaluating 81 luma values for its level-2 pixel -- took 14.5 us.)  The frame graph's head was three dependent launches of ~4.7 us for
// ~1 us of work; `z` clears up to eight small buffers in the same launch (the level kernels' barrier workspaces and the select
// histograms of the frame: four more launches).
struct ZeroList { uint4* p[8]; long n16[8]; int n; };
__device__ __forceinline__ float rgb_luma(const float* __restrict__ rgb, long HW, long q) {
#pragma clang fp contract(off)
  const float r = rgb[q], g = rgb[HW + q], b = rgb[2 * HW + q];
  const float s = 0.2989f * r + 0.587f * g;
  return s + 0.114f * b;
}
__global__ __launch_bounds__(256) void frame_pyramid3_kernel(const float* __restrict__ rgb, float* __restrict__ gray,
                                                             float* __restrict__ l1, float* __restrict__ l2, int H, int W, ZeroList z) {
  // one workgroup = a 16 x 16 tile of level 1 (its 32 x 32 luma block, its 8 x 8 level-2 pixels): the 17 x 17 level-1 values the
  // level-2 stencils touch (reflected at the level-1 border, as blur_down does) go through LDS
  __shared__ float t1[17][18];
  const int tid = threadIdx.x;
  const long nwg = (long)gridDim.x * gridDim.y, gtid = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid, gsz = nwg * 256;
  for (int k = 0; k < z.n; ++k)
    for (long e = gtid; e < z.n16[k]; e += gsz) z.p[k][e] = make_uint4(0u, 0u, 0u, 0u);
  const int H1 = (H + 1) / 2, W1 = (W + 1) / 2, H2 = (H1 + 1) / 2, W2 = (W1 + 1) / 2;
  const long HW = (long)H * W;
  const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
  for (int e = tid; e < 17 * 17; e += 256) {
    const int j = e / 17, i = e - j * 17;
    const int x1 = tx0 - 1 + i, y1 = ty0 - 1 + j;                      // level-1 position of this halo entry
    const int xr = reflect1(min(x1, W1), W1), yr = reflect1(min(y1, H1), H1);
    const int x = 2 * xr, y = 2 * yr;
    const int xm = reflect1(x - 1, W), xp = reflect1(x + 1, W), ym = reflect1(y - 1, H), yp = reflect1(y + 1, H);
    const float mc = rgb_luma(rgb, HW, (long)y * W + x), mr = rgb_luma(rgb, HW, (long)y * W + xp);
    const float bc = rgb_luma(rgb, HW, (long)yp * W + x), br = rgb_luma(rgb, HW, (long)yp * W + xp);
    const float v = blur9<float>(rgb_luma(rgb, HW, (long)ym * W + xm), rgb_luma(rgb, HW, (long)ym * W + x), rgb_luma(rgb, HW, (long)ym * W + xp),
                                 rgb_luma(rgb, HW, (long)y * W + xm), mc, mr, rgb_luma(rgb, HW, (long)yp * W + xm), bc, br);
    t1[j][i] = v;
    if (i >= 1 && j >= 1 && x1 < W1 && y1 < H1) {                      // this workgroup owns the pixel: level 1 and its 2 x 2 luma block
      l1[(long)y1 * W1 + x1] = v;
      gray[(long)y * W + x] = mc;
      if (x + 1 < W) gray[(long)y * W + x + 1] = mr;
      if (y + 1 < H) {
        gray[(long)(y + 1) * W + x] = bc;
        if (x + 1 < W) gray[(long)(y + 1) * W + x + 1] = br;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int lx = tid & 7, ly = tid >> 3;
    const int x2 = (tx0 >> 1) + lx, y2 = (ty0 >> 1) + ly;
    if (x2 < W2 && y2 < H2) {
      const int i = 2 * lx + 1, j = 2 * ly + 1;
      l2[(long)y2 * W2 + x2] = blur9<float>(t1[j - 1][i - 1], t1[j - 1][i], t1[j - 1][i + 1], t1[j][i - 1], t1[j][i], t1[j][i + 1],
                                            t1[j + 1][i - 1], t1[j + 1][i], t1[j + 1][i + 1]);
    }
  }
}


// [1 2 1]^2/16 blur, reflect padding, no decimation (GaussianBlurModule.forward, image_processing.py:60-65)
template <typename T>
__global__ __launch_bounds__(256) void blur_kernel(const T* __restrict__ img, T* __restrict__ out, int H, int W, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const long nc = i / ((long)W * H);
  const T* p = img + nc * H * W;
  const int xm = reflect1(x - 1, W), xp = reflect1(x + 1, W), ym = reflect1(y - 1, H), yp = reflect1(y + 1, H);
  const T k1 = T(1) / T(16), k2 = T(2) / T(16), k4 = T(4) / T(16);
  out[i] = k1 * (p[(long)ym * W + xm] + p[(long)ym * W + xp] + p[(long)yp * W + xm] + p[(long)yp * W + xp]) +
           k2 * (p[(long)ym * W + x] + p[(long)y * W + xm] + p[(long)y * W + xp] + p[(long)yp * W + x]) + k4 * p[(long)y * W + x];
}

// pyr_depth (como/data/depth_resize.py:6-36), kernel_size = stride = 2: mode 0 bilinear (2x2 mean), 1 nearest_neighbor,
// 2 max, 3 min, 4 masked_bilinear (mean over the non-NaN entries, 0 if there are none)
template <typename T>
__global__ __launch_bounds__(256) void depth_pool2_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int mode,
                                                          long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Ho = (mode == 1) ? (H + 1) / 2 : H / 2, Wo = (mode == 1) ? (W + 1) / 2 : W / 2;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
  const long nc = i / ((long)Wo * Ho);
  const T* p = in + nc * H * W + (long)(2 * y) * W + 2 * x;
  if (mode == 1) { out[i] = p[0]; return; }
  const T a = p[0], b = p[1], c = p[W], d = p[W + 1];
  T v;
  if (mode == 0) v = (((a + b) + c) + d) / T(4);
  else if (mode == 2 || mode == 3) {                       // max_pool2d semantics: a NaN in the window wins
    const T sgn = (mode == 2) ? T(1) : T(-1);
    const T e[4] = {sgn * a, sgn * b, sgn * c, sgn * d};
    T mx = e[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (e[k] > mx || e[k] != e[k]) mx = e[k];
    v = sgn * mx;
  }
  else {
    const T e[4] = {a, b, c, d};
    T sum = T(0), cnt = T(0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e[k] == e[k]) { sum += e[k]; cnt += T(1); }
    v = cnt > T(0) ? sum / cnt : T(0);
  }
  out[i] = v;
}

// One thread per window cell: first maximum of sqrt(gx^2 + gy^2) in row-major scan (max_pool2d semantics).
// img_and_grads (B, 3, H, W) gray; coords (B, n, 2) int64 (row, col); pixidx (B, n) int32 = row*W + col (may be NULL).
template <typename T>
__global__ __launch_bounds__(256) void subselect_kernel(const T* __restrict__ iag, int H, int W, int win, long* __restrict__ coords,
                                                        int* __restrict__ pixidx, long total) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Wo = W / win, Ho = H / win;
  const int cx = (int)(i % Wo), cy = (int)((i / Wo) % Ho);
  const long b = i / ((long)Wo * Ho);
  const T* gx = iag + (b * 3 + 1) * H * W;
  const T* gy = iag + (b * 3 + 2) * H * W;
  T best = T(0);
  int bi = -1;
  for (int dy = 0; dy < win; ++dy)
    for (int dx = 0; dx < win; ++dx) {
      const int idx = (cy * win + dy) * W + cx * win + dx;
      const T a = gx[idx], c = gy[idx];
      const T v = sqrt(a * a + c * c);
      if (bi < 0 || v > best || (v != v && best == best)) { best = v; bi = idx; }   // NaN propagates like max_pool2d
    }
  coords[2 * i] = bi / W;
  coords[2 * i + 1] = bi % W;
  if (pixidx) pixidx[i] = bi;
}

// J (N, 8) = [dI/dw . dpi/dP . [-[P]x , I3] (6), vals, 1]   (photo_tracking.py:46-74, c = 1)
template <typename T>
__global__ __launch_bounds__(256) void precalc_jac_kernel(const T* __restrict__ dI_dw, const T* __restrict__ P,
                                                          const T* __restrict__ vals, const T* __restrict__ K,
                                                          T* __restrict__ J, long N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const T fx = K[0], fy = K[4];
  const T X = P[3 * i], Y = P[3 * i + 1], Z = P[3 * i + 2];
  const T gx = dI_dw[2 * i], gy = dI_dw[2 * i + 1];
  // dI/dP = dI/dw . [[fx/Z, 0, -fx X/Z^2], [0, fy/Z, -fy Y/Z^2]]
  const T a = gx * (fx / Z), b = gy * (fy / Z);
  const T c = gx * (-(fx * X / Z) / Z) + gy * (-(fy * Y / Z) / Z);
  // dP/dT = [-[P]x , I]:  -[P]x = [[0, Z, -Y], [-Z, 0, X], [Y, -X, 0]]
  T* o = J + 8 * i;
  o[0] = b * (-Z) + c * Y;
  o[1] = a * Z + c * (-X);
  o[2] = a * (-Y) + b * X;
  o[3] = a;
  o[4] = b;
  o[5] = c;
  o[6] = vals[i];
  o[7] = T(1);
}

}  // namespace como

extern "C" {

#define COMO_DEF_IMAGE(SFX, T)                                                                                          \
  int como_img_grads_##SFX(const T* img, T* out, int N, int C, int H, int W, como_stream_t stream) {                   \
    if (!img || !out || N <= 0 || C <= 0 || H < 2 || W < 2) return COMO_ERR_ARG;                                       \
    const long total = (long)N * C * H * W;                                                                            \
    hipLaunchKernelGGL(como::img_grads_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,                 \
                       (hipStream_t)stream, img, out, C, H, W, total);                                                 \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_rgb_to_gray_##SFX(const T* rgb, T* out, int N, int H, int W, como_stream_t stream) {                        \
    if (!rgb || !out || N <= 0 || H <= 0 || W <= 0) return COMO_ERR_ARG;                                               \
    const long HW = (long)H * W, total = (long)N * HW;                                                                 \
    hipLaunchKernelGGL(como::rgb_to_gray_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,               \
                       (hipStream_t)stream, rgb, out, HW, total);                                                      \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_img_blur_down_##SFX(const T* img, T* out, int NC, int H, int W, como_stream_t stream) {                     \
    if (!img || !out || NC <= 0 || H < 2 || W < 2) return COMO_ERR_ARG;                                                \
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;                                                                      \
    const long total = (long)NC * Ho * Wo;                                                                             \
    hipLaunchKernelGGL(como::blur_down_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,                 \
                       (hipStream_t)stream, img, out, H, W, Ho, Wo, total);                                            \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_img_blur_##SFX(const T* img, T* out, int NC, int H, int W, como_stream_t stream) {                          \
    if (!img || !out || NC <= 0 || H < 2 || W < 2) return COMO_ERR_ARG;                                                \
    const long total = (long)NC * H * W;                                                                               \
    hipLaunchKernelGGL(como::blur_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,  \
                       img, out, H, W, total);                                                                         \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_depth_pool2_##SFX(const T* in, T* out, int NC, int H, int W, int mode, como_stream_t stream) {              \
    if (!in || !out || NC <= 0 || H < 2 || W < 2 || mode < 0 || mode > 4) return COMO_ERR_ARG;                         \
    const int Ho = (mode == 1) ? (H + 1) / 2 : H / 2, Wo = (mode == 1) ? (W + 1) / 2 : W / 2;                          \
    const long total = (long)NC * Ho * Wo;                                                                             \
    hipLaunchKernelGGL(como::depth_pool2_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,               \
                       (hipStream_t)stream, in, out, H, W, mode, total);                                               \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_subselect_pixels_##SFX(const T* img_and_grads, int B, int H, int W, int window, long* coords, int* pixidx,   \
                                  como_stream_t stream) {                                                              \
    if (!img_and_grads || !coords || B <= 0 || window <= 0 || H < window || W < window) return COMO_ERR_ARG;            \
    const long total = (long)B * (H / window) * (W / window);                                                          \
    hipLaunchKernelGGL(como::subselect_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,                 \
                       (hipStream_t)stream, img_and_grads, H, W, window, coords, pixidx, total);                       \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_track_precalc_jac_##SFX(const T* dI_dw, const T* P, const T* vals, const T* K, T* J, long N,                 \
                                   como_stream_t stream) {                                                             \
    if (!dI_dw || !P || !vals || !K || !J || N < 0) return COMO_ERR_ARG;                                               \
    if (!N) return COMO_OK;                                                                                            \
    hipLaunchKernelGGL(como::precalc_jac_kernel<T>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,                   \
                       (hipStream_t)stream, dI_dw, P, vals, K, J, N);                                                  \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }
COMO_DEF_IMAGE(f32, float)
COMO_DEF_IMAGE(f64, double)

int como_frame_stack_f64(const void* rgb, int rgb_is_f32, int H, int W, double* stack, float* stack_pix, como_stream_t stream) {
  if (!rgb || !stack || H < 2 || W < 2) return COMO_ERR_ARG;
  const long HW = (long)H * W;
  const dim3 grid((unsigned)((HW + 255) / 256)), blk(256);
  if (rgb_is_f32)
    hipLaunchKernelGGL(como::frame_stack_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)rgb, stack, stack_pix, H, W);
  else
    hipLaunchKernelGGL(como::frame_stack_kernel<double>, grid, blk, 0, (hipStream_t)stream, (const double*)rgb, stack, stack_pix, H, W);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}


/* The tracker's three-level image pyramid of one colour frame (3,H,W) in one launch -- gray (H,W), l1 (ceil(H/2),ceil(W/2)), l2 (half of
 * that again): bit-identical to como_rgb_to_gray_f32 + 2 x como_img_blur_down_f32.  zero_ptrs / zero_bytes (n_zero <= 8, 16-byte
 * aligned, multiples of 16 bytes): buffers cleared in the same launch. */
int como_track_frame_pyramid3_f32(const float* rgb, float* gray, float* l1, float* l2, int H, int W, void* const* zero_ptrs,
                                  const long* zero_bytes, int n_zero, como_stream_t stream) {
  if (!rgb || !gray || !l1 || !l2 || H < 4 || W < 4 || n_zero < 0 || n_zero > 8 || (n_zero && (!zero_ptrs || !zero_bytes)))
    return COMO_ERR_ARG;
  como::ZeroList z;
  z.n = n_zero;
  long most = 0;
  for (int k = 0; k < 8; ++k) {
    z.p[k] = nullptr; z.n16[k] = 0;
    if (k < n_zero) {
      if (!zero_ptrs[k] || ((uintptr_t)zero_ptrs[k] & 15) || zero_bytes[k] < 0 || (zero_bytes[k] & 15)) return COMO_ERR_ARG;
      z.p[k] = (uint4*)zero_ptrs[k];
      z.n16[k] = zero_bytes[k] / 16;
      if (z.n16[k] > most) most = z.n16[k];
    }
  }
  (void)most;                                                // (the clears are grid-stride loops over the tiles' threads)
  const int H1 = (H + 1) / 2, W1 = (W + 1) / 2;
  hipLaunchKernelGGL(como::frame_pyramid3_kernel, dim3((unsigned)((W1 + 15) / 16), (unsigned)((H1 + 15) / 16)), dim3(256), 0,
                     (hipStream_t)stream, rgb, gray, l1, l2, H, W, z);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
