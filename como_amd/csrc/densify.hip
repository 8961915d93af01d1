// Dense reference points and the GP predictor K~ = K_nm K_mm^-1.
//
// dense_ref  (per GN iteration)  reference: como/odom/backend/sparse_map.py:184-230
//   (backproject_cloud + setup_test_points) and Mapping.prep_dense_ref (Mapping.py:661-699).
//   Per selected pixel n of keyframe b: logz_n = K~[n,:] logz_m, z_n = e^{logz_n}, P_c = z_n ray,
//   P_w = T_wc P_c, and the FACTORS of the Jacobians instead of the (B,n,3,m) tensor the reference builds:
//     dPwn_dzm[n,:,k] = uvec[n,:] * K~[n,k] / z_mk          with uvec = R_wc ray z_n
//     dPwn_dTwc       = [-R [P_c]x , R] + uvec (x) (K~[n,:] dlogz_m/dT_wc)
//   plus the pass-0 histogram of z_n for the exact per-keyframe median depth (sparse_map.py:220).
//   HBM-bound: one m-wide K~ row (4m B) read per pixel, 25 scalars written (structure-of-arrays planes).
//
// ktilde     (once per keyframe)  reference: como/odom/Mapping.py:430-468 (prep_predictor), kernel formulas of
//   como/depth_cov/core/kernels.py:22-88 (the Python twin: C = 2 d1^.25 d2^.25 / sqrt(det + 1e-8), coordinate
//   differences cast to float32) and the bilinear border lookup of gaussian_kernel.py:52-79.
#include "select.cuh"
#include "../../include/como_hip.h"

namespace como {

template <typename T> int select_hist(const T*, const uint8_t*, long, int, uint32_t*, int, hipStream_t);

template <typename T> struct Q4 { T x, y, z, w; };
template <typename T>
__device__ __forceinline__ Q4<T> ld4(const T* __restrict__ p) {
  Q4<T> v;
  if constexpr (sizeof(T) == 4) {
    const float4 f = *reinterpret_cast<const float4*>(p);
    v.x = f.x; v.y = f.y; v.z = f.z; v.w = f.w;
  } else {
    const double2 a = *reinterpret_cast<const double2*>(p);
    const double2 b = *reinterpret_cast<const double2*>(p + 2);
    v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void dense_ref_kernel(
    const T* __restrict__ Kt, long kt_slot_stride, const int* __restrict__ pixidx, const T* __restrict__ logzm,
    const T* __restrict__ Twc, const T* __restrict__ Kmat, const T* __restrict__ dlogzm_dTwc, int n, int m, int Wimg,
    T* __restrict__ Pwn, T* __restrict__ dPwn_dTwc, T* __restrict__ uvec, T* __restrict__ zbuf, T* __restrict__ logzn_out,
    uint32_t* __restrict__ hists, const int* __restrict__ pixcoord, int compact) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ T coef[64][8];          // per inducing point: {logz_m, dlogz_m/dT (6), 0}
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < SEL_BINS; k += 256) lh[k] = 0;
  for (int k = threadIdx.x; k < 64 * 8; k += 256) {
    const int j = k >> 3, e = k & 7;
    T v = T(0);
    if (j < m) {
      if (e == 0) v = logzm[(long)b * m + j];
      else if (e < 7) v = dlogzm_dTwc[((long)b * m + j) * 6 + (e - 1)];
    }
    coef[j][e] = v;
  }
  T Tm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tm[k] = Twc[16 * (long)b + k];
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  __syncthreads();
  const int stride = gridDim.x * 256;
  const int iters = (n + stride - 1) / stride;
  for (int it = 0; it < iters; ++it) {
    const int i0 = it * stride + blockIdx.x * 256 + threadIdx.x;
    const bool inr = i0 < n;
    const int i = inr ? i0 : n - 1;
    const int row = pixidx ? pixidx[(long)b * n + i] : i;
    const T* Kr = Kt + (long)b * kt_slot_stride + (long)row * m;
    T acc[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    for (int j = 0; j < m; j += 4) {
      const Q4<T> kv = ld4(Kr + j);
      const T kk[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int a = 0; a < 7; ++a) acc[a] += kk[e] * coef[j + e][a];
      }
    }
    const T logz = acc[0];
    const T z = exp(logz);                                       // depth.py:6-9
    const int crow = pixcoord ? pixcoord[(long)b * n + i] : row;
    T rx, ry;
    {
#pragma clang fp contract(off)
      rx = (T(crow % Wimg) - cx) / fx;                           // camera.py:43-47 with p = (col, row)
      ry = (T(crow / Wimg) - cy) / fy;
    }
    const T Xc = z * rx, Yc = z * ry, Zc = z;                    // P = z * ray
    T Xw, Yw, Zw;
    rigid_apply(Tm, Xc, Yc, Zc, Xw, Yw, Zw);                     // transforms.py:17-23 (exact order: feeds masks)
    const T u0 = Tm[0] * Xc + Tm[1] * Yc + Tm[2] * Zc;           // uvec = R (ray z)
    const T u1 = Tm[4] * Xc + Tm[5] * Yc + Tm[6] * Zc;
    const T u2 = Tm[8] * Xc + Tm[9] * Yc + Tm[10] * Zc;
    if (inr && Pwn && compact) {
      // compact form: the block kernels rebuild dPwn_dTwc = [-[u]x R, R] + u (x) dlogz_n/dT_wc from P_w, the pose and these
      // six dot products (u = P_w - t_wc): 9 planes written instead of 24
      const long base3 = (long)b * 3 * n + i, base6 = (long)b * 6 * n + i;
      Pwn[base3] = Xw; Pwn[base3 + n] = Yw; Pwn[base3 + 2 * (long)n] = Zw;
#pragma unroll
      for (int k = 0; k < 6; ++k) dPwn_dTwc[base6 + (long)k * n] = acc[1 + k];
    } else if (inr && Pwn) {                                       // Pwn == nullptr: depth-only pass (full-image median)
      const long base3 = (long)b * 3 * n + i, base18 = (long)b * 18 * n + i;
      Pwn[base3] = Xw; Pwn[base3 + n] = Yw; Pwn[base3 + 2 * (long)n] = Zw;
      uvec[base3] = u0; uvec[base3 + n] = u1; uvec[base3 + 2 * (long)n] = u2;
      const T uu[3] = {u0, u1, u2};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const T r0 = Tm[r * 4 + 0], r1 = Tm[r * 4 + 1], r2 = Tm[r * 4 + 2];
        // -(R [P_c]x) row r, then R row r; plus uvec_r * dlogz_n/dT
        const T s0 = -(r1 * Zc - r2 * Yc), s1 = -(r2 * Xc - r0 * Zc), s2 = -(r0 * Yc - r1 * Xc);
        dPwn_dTwc[base18 + (long)(r * 6 + 0) * n] = s0 + uu[r] * acc[1];
        dPwn_dTwc[base18 + (long)(r * 6 + 1) * n] = s1 + uu[r] * acc[2];
        dPwn_dTwc[base18 + (long)(r * 6 + 2) * n] = s2 + uu[r] * acc[3];
        dPwn_dTwc[base18 + (long)(r * 6 + 3) * n] = r0 + uu[r] * acc[4];
        dPwn_dTwc[base18 + (long)(r * 6 + 4) * n] = r1 + uu[r] * acc[5];
        dPwn_dTwc[base18 + (long)(r * 6 + 5) * n] = r2 + uu[r] * acc[6];
      }
    }
    if (inr) {
      zbuf[(long)b * n + i] = Zc;
      if (logzn_out) logzn_out[(long)b * n + i] = logz;
    }
    sel_lds_add(lh, sel_digit<KeyT>(abs_key(Zc), 0), inr);
  }
  __syncthreads();
  if (hists) sel_flush(lh, hists + (long)b * 6 * SEL_BINS);
}

// float32 fast path: the (pixels x m) . (m x 7) product runs on the matrix cores straight out of the load registers.
// One wave owns 64 consecutive pixels; load (it, s) of lane (c = l & 15, q = l >> 4) fetches the 16 bytes
// K~[pixel 16 it + c][16 s + 4 q ..+3], i.e. every wave-load touches 16 rows x 64 contiguous bytes (the thread-per-row
// version touched 64 rows x 16 bytes: four times the cache-line lookups for the same data).  Those registers ARE the A
// operands of v_mfma_f32_16x16x4_f32 (row = l & 15, k = l >> 4; the k permutation k = 16 s + 4 q + e is applied to the
// coefficient operand too); D (rows 4 q + r, column l & 15) goes through a 2 KB LDS transpose so that the epilogue
// (exp, ray, pose Jacobian, 27 coalesced plane stores) runs one pixel per lane.
template <typename T> struct DAcc { typedef T type __attribute__((ext_vector_type(4))); };
__device__ __forceinline__ typename DAcc<float>::type d_mfma(float a, float b, typename DAcc<float>::type c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ typename DAcc<double>::type d_mfma(double a, double b, typename DAcc<double>::type c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// C/D row of (lane, register) of the 16x16x4 MFMA: float32 4 (lane >> 4) + reg, float64 (lane >> 4) + 4 reg
template <typename T> __device__ __forceinline__ int d_mfma_row(int lane, int reg);
template <> __device__ __forceinline__ int d_mfma_row<float>(int lane, int reg) { return (lane >> 4) * 4 + reg; }
template <> __device__ __forceinline__ int d_mfma_row<double>(int lane, int reg) { return (lane >> 4) + 4 * reg; }

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Pass 1 of the window's photometric system (csrc/ba.hip ba_residual_kernel: warp into the target frame, bilinear sample, affine
// residual, validity, first digit of the robust scale's histogram) FUSED into the dense reference (round 6): the reference point
// P_w of a pixel is in a register here, and the pairs that use keyframe b as their reference (two keyframe pairs, plus one-way
// frames) only need it, the pixel's reference intensity and their own 12 + 2 pair constants -- the separate kernel re-read P_w
// (3 planes) and the intensities for every PAIR of the pixel: 138 MB of the dense float64 window's 177, 53 us per iteration.
// Same operations on the same values (warp_point / make_taps / tap_sum of common.cuh, the residual expression of
// ba_residual_kernel): r, the validity bytes and the histogram are those of the separate kernel, bit for bit (tested).
template <typename T>
struct DRFuse {
  const int* ref_pairs;     // (B, np_max): the pairs whose reference slot is keyframe b, -1 padded
  int np_max;
  const T* pair_T;          // (b, 12) inverse target poses      (ba_pair_setup_kernel)
  const T* pair_aff;        // (b, 2)  exp(a_j - a_i), b_j - b_i
  const T* vals;            // (B, n)  reference intensities (one channel)
  const T* img_base;        // target stacks: img_base + tgt_img[p]
  const long* tgt_img;
  T* r_out;                 // (b, n)
  uint8_t* valid_out;       // (b, n)
  uint32_t* rhists;         // the robust scale's select workspace (digit 0 accumulated here)
  int H, W, anorm_f32;
};

template <typename T, bool FUSE>
__global__ __launch_bounds__(256) void dense_ref_mfma_kernel(
    const T* __restrict__ Kt, long kt_slot_stride, const int* __restrict__ pixidx, const T* __restrict__ logzm,
    const T* __restrict__ Twc, const T* __restrict__ Kmat, const T* __restrict__ dlogzm_dTwc, int n, int m,
    int Wimg, T* __restrict__ Pwn, T* __restrict__ dPwn_dTwc, T* __restrict__ uvec, T* __restrict__ zbuf,
    T* __restrict__ logzn_out, uint32_t* __restrict__ hists, const int* __restrict__ pixcoord, int compact, DRFuse<T> fz) {
  using KeyT = typename KeyOf<T>::type;
  using acc_t = typename DAcc<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ uint32_t lh2[FUSE ? SEL_BINS : 1];
  __shared__ T sD[4][64 * 9];
  if constexpr (FUSE) {
    for (int k = threadIdx.x; k < SEL_BINS; k += 256) lh2[k] = 0;
  }
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, q = lane >> 4;
  for (int k = tid; k < SEL_BINS; k += 256) lh[k] = 0;
  T bop[16];                          // coefficient operand: column c of {logz_m, dlogz_m/dT (6), 0...}, k = 16 s + 4 q + e
#pragma unroll
  for (int se = 0; se < 16; ++se) {
    const int k = 16 * (se >> 2) + 4 * q + (se & 3);
    T v = T(0);
    if (k < m) {
      if (c == 0) v = logzm[(long)b * m + k];
      else if (c < 7) v = dlogzm_dTwc[((long)b * m + k) * 6 + (c - 1)];
    }
    bop[se] = v;
  }
  T Tm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tm[k] = Twc[16 * (long)b + k];
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  __syncthreads();
  T* myD = sD[wv];
  const int tiles = (n + 63) >> 6;
  // Two half-tile register buffers (2 x 16 pixels each): the second half's K~ rows and the NEXT tile's first half are in flight
  // while the matrix cores work on the other buffer and while the per-pixel epilogue runs -- one buffer of the whole tile made
  // every wave wait for its 16 loads, multiply, then write, with nothing in flight in between (float64: 327 -> 288 us with the
  // matrix-core kernel alone).
  Q4<T> kA[2][4], kB[2][4];           // (float64: 32 B per lane and load -- 16 rows x 128 contiguous bytes per wave load)
  auto load_half = [&](Q4<T> (&dst)[2][4], int tile_, int half) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int ii = min((tile_ << 6) + 16 * (2 * half + h2) + c, n - 1);
      const int rw = pixidx ? pixidx[(long)b * n + ii] : ii;
      const T* Kr = Kt + (long)b * kt_slot_stride + (long)rw * m;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = min(16 * s + 4 * q, m - 4);           // clamped: the matching coefficients are zero
        dst[h2][s] = ld4(Kr + kk);
      }
    }
  };
  auto mma_half = [&](const Q4<T> (&src)[2][4], int half) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      acc_t acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc = d_mfma(src[h2][s].x, bop[4 * s + 0], acc);
        acc = d_mfma(src[h2][s].y, bop[4 * s + 1], acc);
        acc = d_mfma(src[h2][s].z, bop[4 * s + 2], acc);
        acc = d_mfma(src[h2][s].w, bop[4 * s + 3], acc);
      }
      if (c < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) myD[(16 * (2 * half + h2) + d_mfma_row<T>(lane, r)) * 9 + c] = acc[r];
      }
    }
  };
  const int tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wv;
  if (tile < tiles) load_half(kA, tile, 0);
  for (; tile < tiles; tile += tstride) {
    const int px0 = tile << 6;
    load_half(kB, tile, 1);
    mma_half(kA, 0);
    if (tile + tstride < tiles) load_half(kA, tile + tstride, 0);     // (wave-uniform)
    mma_half(kB, 1);
    wave_lds_fence();
    T a7[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) a7[a] = myD[lane * 9 + a];
    wave_lds_fence();
    const int i0 = px0 + lane;
    const bool inr = i0 < n;
    const int i = inr ? i0 : n - 1;
    const int row = pixidx ? pixidx[(long)b * n + i] : i;
    const T logz = a7[0];
    const T z = exp(logz);                                       // depth.py:6-9
    const int crow = pixcoord ? pixcoord[(long)b * n + i] : row;
    T rx, ry;
    {
#pragma clang fp contract(off)
      rx = (T(crow % Wimg) - cx) / fx;                           // camera.py:43-47 with p = (col, row)
      ry = (T(crow / Wimg) - cy) / fy;
    }
    const T Xc = z * rx, Yc = z * ry, Zc = z;
    T Xw, Yw, Zw;
    rigid_apply(Tm, Xc, Yc, Zc, Xw, Yw, Zw);
    const T u0 = Tm[0] * Xc + Tm[1] * Yc + Tm[2] * Zc;
    const T u1 = Tm[4] * Xc + Tm[5] * Yc + Tm[6] * Zc;
    const T u2 = Tm[8] * Xc + Tm[9] * Yc + Tm[10] * Zc;
    if (inr && Pwn && compact) {
      const long base3 = (long)b * 3 * n + i, base6 = (long)b * 6 * n + i;
      Pwn[base3] = Xw; Pwn[base3 + n] = Yw; Pwn[base3 + 2 * (long)n] = Zw;
#pragma unroll
      for (int k = 0; k < 6; ++k) dPwn_dTwc[base6 + (long)k * n] = a7[1 + k];
    } else if (inr && Pwn) {
      const long base3 = (long)b * 3 * n + i, base18 = (long)b * 18 * n + i;
      Pwn[base3] = Xw; Pwn[base3 + n] = Yw; Pwn[base3 + 2 * (long)n] = Zw;
      uvec[base3] = u0; uvec[base3 + n] = u1; uvec[base3 + 2 * (long)n] = u2;
      const T uu[3] = {u0, u1, u2};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const T r0 = Tm[r * 4 + 0], r1 = Tm[r * 4 + 1], r2 = Tm[r * 4 + 2];
        const T s0 = -(r1 * Zc - r2 * Yc), s1 = -(r2 * Xc - r0 * Zc), s2 = -(r0 * Yc - r1 * Xc);
        dPwn_dTwc[base18 + (long)(r * 6 + 0) * n] = s0 + uu[r] * a7[1];
        dPwn_dTwc[base18 + (long)(r * 6 + 1) * n] = s1 + uu[r] * a7[2];
        dPwn_dTwc[base18 + (long)(r * 6 + 2) * n] = s2 + uu[r] * a7[3];
        dPwn_dTwc[base18 + (long)(r * 6 + 3) * n] = r0 + uu[r] * a7[4];
        dPwn_dTwc[base18 + (long)(r * 6 + 4) * n] = r1 + uu[r] * a7[5];
        dPwn_dTwc[base18 + (long)(r * 6 + 5) * n] = r2 + uu[r] * a7[6];
      }
    }
    if (inr) {
      zbuf[(long)b * n + i] = Zc;
      if (logzn_out) logzn_out[(long)b * n + i] = logz;
    }
    sel_lds_add(lh, sel_digit<KeyT>(abs_key(Zc), 0), inr);
    if constexpr (FUSE) {
      const T vr = fz.vals[(long)b * n + i];
      const T ax = fz.anorm_f32 ? (T)(1.0f / (float)fz.W) : T(1) / T(fz.W), ay = fz.anorm_f32 ? (T)(1.0f / (float)fz.H) : T(1) / T(fz.H);
      for (int q = 0; q < fz.np_max; ++q) {
        const int p = fz.ref_pairs[(long)b * fz.np_max + q];                 // (uniform)
        if (p < 0) break;
        const T* M = fz.pair_T + 12 * (long)p;
        const T scale = fz.pair_aff[2 * p], bias = fz.pair_aff[2 * p + 1];
        const T* img = fz.img_base + fz.tgt_img[p];
        // warp_point of csrc/ba.hip (photo.py:106, camera.py:20-26, photo.py:15-21)
        T X, Y, Z;
        rigid_apply(M, Xw, Yw, Zw, X, Y, Z);
        const T pu = project1(fx, X, Z, cx);
        const T pv = project1(fy, Y, Z, cy);
        const bool ok = in_image(pu, pv, fz.H, fz.W) && (Z > T(0));
        Taps<T> t = make_taps(grid_position(pu, fz.W, ax), grid_position(pv, fz.H, ay), fz.H, fz.W);
        const T It = tap_sum(img, t);
        const T r = It - scale * vr + bias;         // photo.py:114-118
        if (inr) {
          const long oi = (long)p * n + i;
          fz.r_out[oi] = r;
          fz.valid_out[oi] = ok ? 1 : 0;
        }
        sel_lds_add(lh2, sel_digit<KeyT>(abs_key(r), 0), inr && ok);
      }
    }
  }
  __syncthreads();
  if (hists) sel_flush(lh, hists + (long)b * 6 * SEL_BINS);
  if constexpr (FUSE) sel_flush(lh2, fz.rhists);
}

template <typename T>
__global__ void copy_elems_kernel(const T* __restrict__ src, T* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// ---- full-image median depth without re-reading K~ (Mapping.store_vars, Mapping.py:749-758) ---------------------------
// With sub-selected reference pixels (nonmax window > 1) the priors and the landmark re-initialisation still need the exact
// per-keyframe median of the FULL depth image z_n = exp(K~[n,:] logz_m): a depth-only pass over all H W rows of K~ per GN
// iteration (630 MB at 8 x 640x480 in float32), only to find ONE order statistic.  Between two iterations logz_m moves by a
// small step, and |delta logz_n| <= ||K~[n,:]||_1 max_k |delta logz_m,k|: a pixel whose cached log-depth interval
// [lref - err, lref + err] lies wholly on one side of where the new median can be is counted without being evaluated.
//   state per pixel : lref (log-depth at its last exact evaluation), err (bound on its drift since then), l1 (||K~ row||_1)
//   per keyframe    : logzm_prev (log-depths at the previous call), l1max, the previous exact median (med_prev3[b][0])
//   new median in   : [log med_prev - l1max delta, log med_prev + l1max delta]      (every pixel moves by at most l1max delta)
// A pixel below that interval writes depth 0, one above it +inf, a CANDIDATE is evaluated exactly (and its state refreshed):
// the median of the resulting plane IS the exact median of the full image as long as the true median is a candidate's value,
// which the interval guarantees.  The usual select passes run on the plane unchanged.  init != 0: every pixel is a candidate
// (builds the state).  Cost: 3 state planes read (+ rewritten where they change) + the candidates' K~ rows.
template <typename T>
__global__ __launch_bounds__(256) void depth_band_kernel(
    const T* __restrict__ Kt, long kt_slot_stride, const T* __restrict__ logzm, const T* __restrict__ logzm_prev, int rows, int m,
    T* __restrict__ lref, T* __restrict__ err, T* __restrict__ l1, float* __restrict__ l1max, const T* __restrict__ med_prev3,
    T* __restrict__ zbuf, uint32_t* __restrict__ hists, unsigned* __restrict__ ncand, int init) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ T coef[64];
  __shared__ T sdelta, sabs;
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < SEL_BINS; k += 256) lh[k] = 0;
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    const T cur = (k < m) ? logzm[(long)b * m + k] : T(0);
    const T prv = (k < m && !init) ? logzm_prev[(long)b * m + k] : cur;
    coef[k] = cur;
    T d = fabs(cur - prv), a = fabs(cur);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d = fmax(d, __shfl_xor(d, off, 64)); a = fmax(a, __shfl_xor(a, off, 64)); }
    if (k == 0) { sdelta = d; sabs = a; }
  }
  __syncthreads();
  // rounding head-room of one evaluated dot product (m terms) relative to ||row||_1 max|logz_m|
  const T eps = (sizeof(T) == 4) ? T(4e-6) : T(4e-15);
  const T delta = sdelta * (T(1) + T(64) * eps) + (sdelta > T(0) ? eps : T(0));
  const T mp = init ? T(0) : log(med_prev3[3 * b]);
  const T band = init ? T(0) : (T)l1max[b] * delta + eps * (fabs(mp) + T(1));
  const T inf = (T)__builtin_inff();
  const int stride = gridDim.x * 256;
  unsigned mycand = 0;
  float myl1max = 0.f;
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 - (int)threadIdx.x < rows; i0 += stride) {
    const bool inr = i0 < rows;
    const int i = inr ? i0 : rows - 1;
    const long si = (long)b * rows + i;
    bool cand = init != 0;
    T e = T(0), lr = T(0), rl1 = T(0);
    if (!init) {
      rl1 = l1[si];
      lr = lref[si];
      e = err[si] + rl1 * delta;
      cand = !((lr + e < mp - band) || (lr - e > mp + band));            // (NaN anywhere -> candidate)
    }
    T z;
    if (cand) {
      const T* Kr = Kt + (long)b * kt_slot_stride + (long)i * m;
      T acc = T(0), s1 = T(0);
      for (int j = 0; j < m; j += 4) {
        const Q4<T> kv = ld4(Kr + j);
        const T kk[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc += kk[q] * coef[j + q]; s1 += fabs(kk[q]); }
      }
      z = exp(acc);                                                       // depth.py:6-9
      if (inr) {
        lref[si] = acc;
        err[si] = eps * (s1 * sabs + T(1));
        if (init) { l1[si] = s1; myl1max = fmaxf(myl1max, (float)s1 * 1.000001f); }
        ++mycand;
      }
    } else {
      z = (lr < mp) ? T(0) : inf;
      if (inr) err[si] = e;
    }
    if (inr) zbuf[si] = z;
    sel_lds_add(lh, sel_digit<KeyT>(abs_key(z), 0), inr);
  }
  if (init) {                                                             // positive floats order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) myl1max = fmaxf(myl1max, __shfl_xor(myl1max, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)&l1max[b], __float_as_uint(myl1max));
  }
  if (ncand) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mycand += __shfl_xor(mycand, off, 64);
    if ((threadIdx.x & 63) == 0 && mycand) atomicAdd(&ncand[b], mycand);
  }
  __syncthreads();
  if (hists) sel_flush(lh, hists + (long)b * 6 * SEL_BINS);
}

// ---- K~ --------------------------------------------------------------------------------------------------------
// bilinear lookup with border padding at normalised (row, col) coordinates (grid_sample align_corners=False)
template <typename T>
__device__ __forceinline__ void cov_lookup(const T* __restrict__ cov, int H, int W, T rn, T cn, T* E) {
  T y = ((rn + T(1)) * T(H) - T(1)) / T(2);
  T x = ((cn + T(1)) * T(W) - T(1)) / T(2);
  x = fmin(fmax(x, T(0)), T(W - 1));
  y = fmin(fmax(y, T(0)), T(H - 1));
  const T xf = floor(x), yf = floor(y);
  const T wx = x - xf, wy = y - yf;
  const int x0 = (int)xf, y0 = (int)yf;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const long HW = (long)H * W;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const T* pl = cov + ch * HW;
    E[ch] = pl[(long)y0 * W + x0] * ((T(1) - wy) * (T(1) - wx)) + pl[(long)y0 * W + x1] * ((T(1) - wy) * wx) +
            pl[(long)y1 * W + x0] * (wy * (T(1) - wx)) + pl[(long)y1 * W + x1] * (wy * wx);
  }
}

// Python-twin kernel value (kernels.py:22-88): note diff is cast to float32 there.
template <typename T>
__device__ __forceinline__ T kernel_py(T r1, T c1, const T* E1, T r2, T c2, const T* E2) {
  const T d0 = (T)(float)(r1 - r2), d1 = (T)(float)(c1 - c2);
  const T e00 = E1[0] + E2[0], e01 = E1[1] + E2[1], e11 = E1[3] + E2[3];
  T q = e11 * (d0 * d0);
  q += T(-2) * e01 * d0 * d1;
  q += e00 * (d1 * d1);
  const T det = e00 * e11 - e01 * e01;
  q = (q / det) * T(0.5);
  const T da = sqrt(sqrt(E1[0] * E1[3] - E1[1] * E1[2]));
  const T db = sqrt(sqrt(E2[0] * E2[3] - E2[1] * E2[2]));
  const T C = T(2) * da * db / sqrt(det + T(1e-8));
  const T tmp = T(1.7320508075688772) * sqrt(q + T(1e-8));
  return ((T(1) + tmp) * exp(-tmp)) * C;
}

// out[b, i, j] = scale * k(x1_i, x2_j)   (CovarianceModule / CrossCovarianceModule forward, covariance.py:10-39)
template <typename T>
__global__ __launch_bounds__(256) void kernel_matrix_py_kernel(const T* __restrict__ x1, const T* __restrict__ E1,
                                                               const T* __restrict__ x2, const T* __restrict__ E2, T scale,
                                                               T* __restrict__ out, int N, int M) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)N * M) return;
  const int i = (int)(p / M), j = (int)(p % M);
  const T* a = x1 + ((long)b * N + i) * 2;
  const T* c = x2 + ((long)b * M + j) * 2;
  out[((long)b * N + i) * M + j] =
      kernel_py(a[0], a[1], E1 + ((long)b * N + i) * 4, c[0], c[1], E2 + ((long)b * M + j) * 4) * scale;
}

// ---- keyframe-insertion glue: the covariance parameters AT a list of points, the diagonal prior variance, back-projection --------
// One launch each for what the keyframe path (corr.py track_and_init, distill_depth.py calc_kernel_matrices, samplers.py,
// Mapping.prep_predictor) did as chains of tiny torch launches -- ~8 per call of normalize_coordinates + interpolate_kernel_params,
// ~20 per DiagonalCovarianceModule call, 7 per backprojection --, with EXACTLY the arithmetic of those chains: every torch
// elementwise kernel rounds its own result (no fusion across them: contraction off here), torch's grid_sample kernel is one
// expression per value (contracted by the compiler: the same expressions under the same default here).
//
// normalize_coordinates (coords.py:12-15 of the reference): x_norm = (2A x + A) - 1, A = 1 / dims, in the coordinates' type
template <typename TC>
__device__ __forceinline__ TC norm_coord(TC x, int dim) {
#pragma clang fp contract(off)
  const TC A = TC(1) / TC(dim);
  const TC A2 = TC(2) * A;
  TC t = A2 * x;
  t = t + A;
  return t - TC(1);
}

// torch.nn.functional.grid_sample(bilinear, padding_mode="border", align_corners=False) at ONE normalised (x, y): the four
// channels of the covariance image (ATen GridSampler.cu grid_sampler_2d_kernel: unnormalise, clip, floor, corner weights as
// products of differences, out = sum of value * weight over the in-bounds corners in the order nw, ne, sw, se)
template <typename T>
__device__ __forceinline__ void grid_sample_border4(const T* __restrict__ img, int H, int W, T x, T y, T* __restrict__ out) {
  T ix = ((x + 1.f) * W - 1) / 2;
  T iy = ((y + 1.f) * H - 1) / 2;
  ix = fmin((T)(W - 1), fmax(ix, (T)0));
  iy = fmin((T)(H - 1), fmax(iy, (T)0));
  const long ix_nw = (long)floor(ix), iy_nw = (long)floor(iy);
  const long ix_ne = ix_nw + 1, iy_ne = iy_nw, ix_sw = ix_nw, iy_sw = iy_nw + 1, ix_se = ix_nw + 1, iy_se = iy_nw + 1;
  const T nw = (ix_se - ix) * (iy_se - iy);
  const T ne = (ix - ix_sw) * (iy_sw - iy);
  const T sw = (ix_ne - ix) * (iy - iy_ne);
  const T se = (ix - ix_nw) * (iy - iy_nw);
  const long HW = (long)H * W;
  auto inb = [&](long yy, long xx) { return yy >= 0 && yy < H && xx >= 0 && xx < W; };
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const T* pl = img + c * HW;
    T acc = 0;
    if (inb(iy_nw, ix_nw)) acc += pl[iy_nw * W + ix_nw] * nw;
    if (inb(iy_ne, ix_ne)) acc += pl[iy_ne * W + ix_ne] * ne;
    if (inb(iy_sw, ix_sw)) acc += pl[iy_sw * W + ix_sw] * sw;
    if (inb(iy_se, ix_se)) acc += pl[iy_se * W + ix_se] * se;
    out[c] = acc;
  }
}

// coords (B,N,2) row/col pixel coordinates (type TC) -> cn (B,N,2) normalised (type T: computed in TC, then cast -- the sampler
// normalises float64 coordinates and casts to float32) and E (B,N,4) = interpolate_kernel_params(cov, cn) (gaussian_kernel.py:52-79)
template <typename TC, typename T>
__global__ __launch_bounds__(256) void cov_params_at_kernel(const T* __restrict__ cov, int H, int W, const TC* __restrict__ coords,
                                                            int N, T* __restrict__ cn, T* __restrict__ E) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const long o = (long)b * N + i;
  const T r = (T)norm_coord<TC>(coords[2 * o], H), c = (T)norm_coord<TC>(coords[2 * o + 1], W);
  cn[2 * o] = r;
  cn[2 * o + 1] = c;
  T e[4];
  grid_sample_border4<T>(cov + (long)b * 4 * H * W, H, W, c, r, e);
#pragma unroll
  for (int k = 0; k < 4; ++k) E[4 * o + k] = e[k];
}

// DiagonalCovarianceModule (covariance.py:42-50 + kernels.py:69-88 with Q = 0), operation by operation
template <typename T>
__global__ __launch_bounds__(256) void diag_cov_kernel(const T* __restrict__ E, long total, T scale, T* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const T e00 = E[4 * i], e01 = E[4 * i + 1], e10 = E[4 * i + 2], e11 = E[4 * i + 3];
  const T p0 = e00 * e11, p1 = e01 * e10;
  const T det = p0 - p1;
  const T f00 = T(2) * e00, f01 = T(2) * e01, f10 = T(2) * e10, f11 = T(2) * e11;
  const T q0 = f00 * f11, q1 = f01 * f10;
  const T det2 = q0 - q1;
  const T num = T(2.0) * sqrt(det);
  const T C = num / sqrt(det2 + T(1e-8));
  const T q = sqrt(T(0) + T(1e-8));
  const T tmp = T(1.7320508075688772) * q;
  const T ex = exp(-tmp);
  const T mat = (T(1) + tmp) * ex;
  const T v = C * mat;
  out[i] = v * scale;
}

// backprojection values (camera.py:43-54): P = z * ((p_x - cx) / fx, (p_y - cy) / fy, 1); p (n,2) x/y, z (n), K (3,3) row-major
template <typename T>
__global__ __launch_bounds__(256) void backproject_kernel(const T* __restrict__ K, const T* __restrict__ p, const T* __restrict__ z,
                                                          long n, T* __restrict__ P) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  const T r0 = (p[2 * i] - cx) / fx, r1 = (p[2 * i + 1] - cy) / fy;
  const T zz = z[i];
  P[3 * i] = zz * r0;
  P[3 * i + 1] = zz * r1;
  P[3 * i + 2] = zz * T(1);
}

// K~ rows for every photo pixel on the matrix cores.  One wave owns 16 pixels per trip: lane (px = l & 15, kq = l >> 4)
// evaluates the 16 kernel values k(pixel px, inducing point 4 s + kq), s = 0..15 -- which is exactly the A-operand layout of
// the 16x16x4 MFMA (row = l & 15, k = l >> 4) -- and multiplies by K_mm^-1 (B operand from LDS: one ds_read per MFMA
// instead of one per FMA: the thread-per-pixel version issued 4096 broadcast LDS reads per pixel and was LDS-bound).
// m <= 64; float32 and float64 (v_mfma_f64_16x16x4_f64).
template <typename T> struct KAcc { typedef T type __attribute__((ext_vector_type(4))); };
__device__ __forceinline__ typename KAcc<float>::type k_mfma(float a, float b, typename KAcc<float>::type c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ typename KAcc<double>::type k_mfma(double a, double b, typename KAcc<double>::type c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ int k_mfma_row(int lane, int reg);          // C/D row of (lane, register)
template <> __device__ __forceinline__ int k_mfma_row<float>(int lane, int reg) { return (lane >> 4) * 4 + reg; }
template <> __device__ __forceinline__ int k_mfma_row<double>(int lane, int reg) { return (lane >> 4) + 4 * reg; }

template <typename T>
__global__ __launch_bounds__(256) void ktilde_kernel(const T* __restrict__ cov, int Hc, int Wc, const T* __restrict__ xm,
                                                     const T* __restrict__ Em, const T* __restrict__ Kinv, T scale,
                                                     int Hp, int Wp, int m, T* __restrict__ out, float* __restrict__ out32) {
  // out32 (float64 instantiation, may be NULL): the same values rounded to float32 -- the per-pixel kernels' mirror of K~, written
  // here instead of by a conversion pass over the 157 MB the kernel has just stored
  using acc_t = typename KAcc<T>::type;
  __shared__ T sK[64 * 65];                                // K_mm^-1, row stride 65 (zero-padded to 64 x 64)
  __shared__ T sx[64 * 2];
  __shared__ T sE[64 * 4];
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < 64 * 64; k += 256) {
    const int r = k >> 6, c = k & 63;
    sK[r * 65 + c] = (r < m && c < m) ? Kinv[(long)b * m * m + (long)r * m + c] : T(0);
  }
  for (int k = threadIdx.x; k < 64 * 2; k += 256) sx[k] = (k < m * 2) ? xm[(long)b * m * 2 + k] : T(0);
  for (int k = threadIdx.x; k < 64 * 4; k += 256) sE[k] = (k < m * 4) ? Em[(long)b * m * 4 + k] : ((k & 3) == 0 || (k & 3) == 3 ? T(1) : T(0));
  __syncthreads();
  const long np = (long)Hp * Wp;
  const T ar = T(1) / T(Hc), ac = T(1) / T(Wc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, px = lane & 15, kq = lane >> 4;
  const long groups = (np + 15) / 16;
  for (long grp = (long)blockIdx.x * 4 + wv; grp < groups; grp += (long)gridDim.x * 4) {
    const long p = min(grp * 16 + px, np - 1);
    const int row = (int)(p / Wp), col = (int)(p % Wp);
    T rn, cn;
    {
#pragma clang fp contract(off)
      rn = (T(2) * ar) * T(row) + ar - T(1);           // normalize_coordinates, coords.py:12-15
      cn = (T(2) * ac) * T(col) + ac - T(1);
    }
    T En[4];
    cov_lookup(cov + (long)b * 4 * Hc * Wc, Hc, Wc, rn, cn, En);
    T a[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int k = 4 * s2 + kq;
      const T v = kernel_py(rn, cn, En, sx[2 * k], sx[2 * k + 1], sE + 4 * k) * scale;
      a[s2] = (k < m) ? v : T(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (16 * t >= m) break;
      acc_t acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc = k_mfma(a[s2], sK[(4 * s2 + kq) * 65 + 16 * t + px], acc);
      const int j = 16 * t + px;                        // here l & 15 is the output column
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long pr = grp * 16 + k_mfma_row<T>(lane, r);
        if (pr < np && j < m) {
          out[((long)b * np + pr) * m + j] = acc[r];
          if (out32) out32[((long)b * np + pr) * m + j] = (float)acc[r];
        }
      }
    }
  }
}

template <typename T>
int dense_ref(const T* Kt, long kt_slot_stride, const int* pixidx, const T* logzm, const T* Twc, const T* Kmat,
              const T* dlogzm_dTwc, int B, int n, int m, int Wimg, T* Pwn, T* dPwn_dTwc, T* uvec, T* zbuf, T* logzn_out,
              void* hists_v, T* med_out3, const int* pixcoord, int flags, hipStream_t s, const DRFuse<T>* fuse = nullptr) {
  using KeyT = typename KeyOf<T>::type;
  const int compact = (flags & 16) ? 1 : 0;   // dPwn_dTwc receives the 6 planes dlogz_n/dT_wc only, uvec is not written
  const bool depth_only = (flags & 8) != 0;   // z_n = exp(K~ logz_m) of every row + its exact median: Mapping.store_vars (Mapping.py:749-758)
  if (!Kt || !logzm || !Twc || !Kmat || !dlogzm_dTwc || !zbuf || !hists_v || !med_out3 ||
      B <= 0 || n <= 0 || m <= 0 || m > 64 || (m & 3) || Wimg <= 0)
    return COMO_ERR_ARG;
  if (depth_only) { Pwn = nullptr; dPwn_dTwc = nullptr; uvec = nullptr; }
  else if (!Pwn || !dPwn_dTwc || (!uvec && !compact)) return COMO_ERR_ARG;
  uint32_t* hists = (uint32_t*)hists_v;
  if (!(flags & 4)) {
  // flags & 64 (with flags & 2, points / depths only): nobody will ask for the median -- no histogram zero-fill, no flush
  if ((flags & 64) && (flags & 2)) hists = nullptr;
  if (hists && !(flags & 1) && !zero_words(hists, (size_t)B * 6 * SEL_BINS, s))
    return COMO_ERR_LAUNCH;
  int gx = (n + 255) / 256;
  if (gx > 512) gx = 512;
  // float32: always the matrix-core kernel.  float64: the matrix-core kernel for the dense reference POINTS (row-major tile
  // loads: 16 rows x 128 contiguous bytes per wave load instead of 64 rows x 32 bytes); the depth-only pass keeps the
  // thread-per-pixel kernel, whose summation order the band median (depth_band_kernel) reproduces bit for bit.
  if (sizeof(T) == 4 || (!depth_only && !(flags & 32))) {
    int gm = ((n + 63) / 64 + 3) / 4;
    if (gm > 256) gm = 256;
    if (fuse && !depth_only)
      hipLaunchKernelGGL((dense_ref_mfma_kernel<T, true>), dim3(gm, B), dim3(256), 0, s, Kt, kt_slot_stride, pixidx, logzm, Twc, Kmat,
                         dlogzm_dTwc, n, m, Wimg, Pwn, dPwn_dTwc, uvec, zbuf, logzn_out, hists, pixcoord, compact, *fuse);
    else
      hipLaunchKernelGGL((dense_ref_mfma_kernel<T, false>), dim3(gm, B), dim3(256), 0, s, Kt, kt_slot_stride, pixidx, logzm, Twc, Kmat,
                         dlogzm_dTwc, n, m, Wimg, Pwn, dPwn_dTwc, uvec, zbuf, logzn_out, hists, pixcoord, compact, DRFuse<T>{});
  } else {
    if (fuse) return COMO_ERR_ARG;                          // (the fused pass exists in the matrix-core kernel only)
    hipLaunchKernelGGL(dense_ref_kernel<T>, dim3(gx, B), dim3(256), 0, s, Kt, kt_slot_stride, pixidx, logzm, Twc, Kmat,
                       dlogzm_dTwc, n, m, Wimg, Pwn, dPwn_dTwc, uvec, zbuf, logzn_out, hists, pixcoord, compact);
  }
  COMO_CHECK_LAUNCH();
  }
  if (flags & 2) return COMO_OK;                          // points only: the median passes run elsewhere (flags & 4)
  for (int p = 1; p < SelCfg<KeyT>::NPASS; ++p) {
    int rc = select_hist<T>(zbuf, nullptr, n, B, hists, p | 0x100, s);     // (all remaining passes here: candidates allowed)
    if (rc) return rc;
  }
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_select_finish_f32(const void*, int, float*, como_stream_t);
int como_select_finish_f64(const void*, int, double*, como_stream_t);

int como_dense_ref_f32(const float* Kt, long kt_slot_stride, const int* pixidx, const float* logzm, const float* Twc,
                       const float* K, const float* dlogzm_dTwc, int B, int n, int m, int Wimg, float* Pwn,
                       float* dPwn_dTwc, float* uvec, float* zbuf, float* logzn_out, void* hists, float* med_out3,
                       const int* pixcoord, int flags, como_stream_t stream) {
  int rc = como::dense_ref<float>(Kt, kt_slot_stride, pixidx, logzm, Twc, K, dlogzm_dTwc, B, n, m, Wimg, Pwn, dPwn_dTwc, uvec,
                                  zbuf, logzn_out, hists, med_out3, pixcoord, flags, (hipStream_t)stream);
  if (rc || (flags & 2)) return rc;
  return como_select_finish_f32(hists, B, med_out3, stream);
}
int como_dense_ref_f64(const double* Kt, long kt_slot_stride, const int* pixidx, const double* logzm, const double* Twc,
                       const double* K, const double* dlogzm_dTwc, int B, int n, int m, int Wimg, double* Pwn,
                       double* dPwn_dTwc, double* uvec, double* zbuf, double* logzn_out, void* hists, double* med_out3,
                       const int* pixcoord, int flags, como_stream_t stream) {
  int rc = como::dense_ref<double>(Kt, kt_slot_stride, pixidx, logzm, Twc, K, dlogzm_dTwc, B, n, m, Wimg, Pwn, dPwn_dTwc, uvec,
                                   zbuf, logzn_out, hists, med_out3, pixcoord, flags, (hipStream_t)stream);
  if (rc || (flags & 2)) return rc;
  return como_select_finish_f64(hists, B, med_out3, stream);
}

/* como_dense_ref_* with pass 1 of the photometric system fused in (see DRFuse above / include/como_hip.h como_dr_fuse). */
#define COMO_DEF_DR_FUSED(SFX, T)                                                                                               \
  int como_dense_ref_fused_##SFX(const T* Kt, long kt_slot_stride, const int* pixidx, const T* logzm, const T* Twc, const T* K,   \
                                 const T* dlogzm_dTwc, int B, int n, int m, int Wimg, T* Pwn, T* dPwn_dTwc, T* uvec, T* zbuf,     \
                                 T* logzn_out, void* hists, T* med_out3, const int* pixcoord, int flags, const como_dr_fuse* f,   \
                                 como_stream_t stream) {                                                                        \
    if (!f || !f->ref_pairs || f->np_max <= 0 || !f->pair_T || !f->pair_aff || !f->vals || !f->img_base || !f->tgt_img ||        \
        !f->r_out || !f->valid_out || !f->rhists || f->H < 3 || f->W < 3 || (flags & (4 | 8)))                                  \
      return COMO_ERR_ARG;                                                                                                      \
    como::DRFuse<T> fz{f->ref_pairs, f->np_max, (const T*)f->pair_T, (const T*)f->pair_aff, (const T*)f->vals,                   \
                       (const T*)f->img_base, f->tgt_img, (T*)f->r_out, (uint8_t*)f->valid_out, (uint32_t*)f->rhists, f->H, f->W,  \
                       f->anorm_f32};                                                                                           \
    int rc = como::dense_ref<T>(Kt, kt_slot_stride, pixidx, logzm, Twc, K, dlogzm_dTwc, B, n, m, Wimg, Pwn, dPwn_dTwc, uvec, zbuf, \
                                logzn_out, hists, med_out3, pixcoord, flags, (hipStream_t)stream, &fz);                         \
    if (rc || (flags & 2)) return rc;                                                                                           \
    return como_select_finish_##SFX(hists, B, med_out3, stream);                                                                \
  }
COMO_DEF_DR_FUSED(f32, float)
COMO_DEF_DR_FUSED(f64, double)

#define COMO_DEF_BAND(SFX, T)                                                                                          \
  int como_depth_band_##SFX(const T* Kt, long kt_slot_stride, const T* logzm, T* logzm_prev, int B, int rows, int m, T* lref,  \
                            T* err, T* l1, float* l1max, const T* med_prev3, T* zbuf, void* hists, unsigned* ncand, int init,  \
                            como_stream_t stream) {                                                                            \
    if (!Kt || !logzm || !logzm_prev || !lref || !err || !l1 || !l1max || !med_prev3 || !zbuf || !hists || B <= 0 ||          \
        rows <= 0 || m <= 0 || m > 64 || (m & 3))                                                                             \
      return COMO_ERR_ARG;                                                                                                    \
    int gx = (rows + 255) / 256;                                                                                              \
    if (gx > 512) gx = 512;                                                                                                   \
    hipLaunchKernelGGL(como::depth_band_kernel<T>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, Kt, kt_slot_stride, logzm,  \
                       (const T*)logzm_prev, rows, m, lref, err, l1, l1max, med_prev3, zbuf, (uint32_t*)hists, ncand, init);   \
    COMO_CHECK_LAUNCH();                                                                                                      \
    /* the log-depths this call was evaluated at become the next call's reference: a KERNEL node (select.cuh: captured  */    \
    /* memset / memcpy nodes are avoided in replayed graphs), after the band kernel on the same stream                   */    \
    hipLaunchKernelGGL(como::copy_elems_kernel<T>, dim3((B * m + 255) / 256), dim3(256), 0, (hipStream_t)stream, logzm,        \
                       logzm_prev, B * m);                                                                                    \
    COMO_CHECK_LAUNCH();                                                                                                      \
    return COMO_OK;                                                                                                           \
  }
COMO_DEF_BAND(f32, float)
COMO_DEF_BAND(f64, double)

#define COMO_DEF_KMAT(SFX, T)                                                                                          \
  int como_kernel_matrix_##SFX(const T* x1, const T* E1, const T* x2, const T* E2, T scale, T* out, int B, int N, int M, \
                               como_stream_t stream) {                                                                 \
    if (!x1 || !E1 || !x2 || !E2 || !out || B < 0 || N < 0 || M < 0) return COMO_ERR_ARG;                              \
    if (!B || !N || !M) return COMO_OK;                                                                                \
    hipLaunchKernelGGL(como::kernel_matrix_py_kernel<T>, dim3((unsigned)(((long)N * M + 255) / 256), B), dim3(256), 0,  \
                       (hipStream_t)stream, x1, E1, x2, E2, scale, out, N, M);                                         \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }                                                                                                                    \
  int como_ktilde_##SFX(const T* cov, int Hc, int Wc, const T* xm, const T* Em, const T* Kinv, T scale, int B, int Hp,   \
                        int Wp, int m, T* out, como_stream_t stream) {                                                 \
    if (!cov || !xm || !Em || !Kinv || !out || B <= 0 || m <= 0 || m > 64 || Hp <= 0 || Wp <= 0) return COMO_ERR_ARG;  \
    long blocks = ((long)Hp * Wp + 63) / 64;                                                                           \
    if (blocks > 2048) blocks = 2048;                                                                                  \
    hipLaunchKernelGGL(como::ktilde_kernel<T>, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, cov, Hc,   \
                       Wc, xm, Em, Kinv, scale, Hp, Wp, m, out, (float*)nullptr);                                      \
    COMO_CHECK_LAUNCH();                                                                                               \
    return COMO_OK;                                                                                                    \
  }
COMO_DEF_KMAT(f32, float)
COMO_DEF_KMAT(f64, double)

int como_ktilde_mirror_f64(const double* cov, int Hc, int Wc, const double* xm, const double* Em, const double* Kinv, double scale, int B,
                           int Hp, int Wp, int m, double* out, float* out_f32, como_stream_t stream) {
  if (!cov || !xm || !Em || !Kinv || !out || B <= 0 || m <= 0 || m > 64 || Hp <= 0 || Wp <= 0) return COMO_ERR_ARG;
  long blocks = ((long)Hp * Wp + 63) / 64;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(como::ktilde_kernel<double>, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, cov, Hc, Wc, xm, Em,
                     Kinv, scale, Hp, Wp, m, out, out_f32);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

#define COMO_DEF_KFGLUE(SFX, T)                                                                                                  \
  int como_diag_cov_##SFX(const T* E, long total, T scale, T* out, como_stream_t stream) {                                       \
    if (!E || !out || total < 0) return COMO_ERR_ARG;                                                                            \
    if (!total) return COMO_OK;                                                                                                  \
    hipLaunchKernelGGL(como::diag_cov_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E,     \
                       total, scale, out);                                                                                       \
    COMO_CHECK_LAUNCH();                                                                                                         \
    return COMO_OK;                                                                                                              \
  }                                                                                                                              \
  int como_backproject_##SFX(const T* K, const T* p, const T* z, long n, T* P, como_stream_t stream) {                           \
    if (!K || !p || !z || !P || n < 0) return COMO_ERR_ARG;                                                                      \
    if (!n) return COMO_OK;                                                                                                      \
    hipLaunchKernelGGL(como::backproject_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, p,   \
                       z, n, P);                                                                                                 \
    COMO_CHECK_LAUNCH();                                                                                                         \
    return COMO_OK;                                                                                                              \
  }                                                                                                                              \
  /* calc_kernel_matrices (distill_depth.py:8-27): normalised coordinates + interpolated parameters of both point sets, */       \
  /* K_mm, K_nm and the diagonal of K_nn in one call (5 launches)                                                        */      \
  int como_kernel_matrices_##SFX(const T* cov, int Hc, int Wc, const T* coords_m, int m, const T* coords_n, int n, T scale,      \
                                 T* cm, T* Em, T* cn, T* En, T* Kmm, T* Knm, T* Kdiag, int B, como_stream_t stream) {            \
    if (!cov || B <= 0 || m < 0 || n < 0 || Hc <= 0 || Wc <= 0 || (m && (!coords_m || !cm || !Em || !Kmm)) ||                    \
        (n && (!coords_n || !cn || !En || !Kdiag)) || (m && n && !Knm))       /* (an empty set's buffers may be null) */          \
      return COMO_ERR_ARG;                                                                                                       \
    hipStream_t s = (hipStream_t)stream;                                                                                         \
    if (m) {                                                                                                                     \
      hipLaunchKernelGGL((como::cov_params_at_kernel<T, T>), dim3((m + 255) / 256, B), dim3(256), 0, s, cov, Hc, Wc, coords_m,   \
                         m, cm, Em);                                                                                             \
      COMO_CHECK_LAUNCH();                                                                                                       \
      hipLaunchKernelGGL(como::kernel_matrix_py_kernel<T>, dim3((unsigned)(((long)m * m + 255) / 256), B), dim3(256), 0, s,       \
                         (const T*)cm, (const T*)Em, (const T*)cm, (const T*)Em, scale, Kmm, m, m);                              \
      COMO_CHECK_LAUNCH();                                                                                                       \
    }                                                                                                                            \
    if (n) {                                                                                                                     \
      hipLaunchKernelGGL((como::cov_params_at_kernel<T, T>), dim3((n + 255) / 256, B), dim3(256), 0, s, cov, Hc, Wc, coords_n,   \
                         n, cn, En);                                                                                             \
      COMO_CHECK_LAUNCH();                                                                                                       \
      hipLaunchKernelGGL(como::diag_cov_kernel<T>, dim3((unsigned)(((long)B * n + 255) / 256)), dim3(256), 0, s, (const T*)En,    \
                         (long)B * n, scale, Kdiag);                                                                             \
      COMO_CHECK_LAUNCH();                                                                                                       \
    }                                                                                                                            \
    if (m && n) {                                                                                                                \
      hipLaunchKernelGGL(como::kernel_matrix_py_kernel<T>, dim3((unsigned)(((long)n * m + 255) / 256), B), dim3(256), 0, s,       \
                         (const T*)cn, (const T*)En, (const T*)cm, (const T*)Em, scale, Knm, n, m);                              \
      COMO_CHECK_LAUNCH();                                                                                                       \
    }                                                                                                                            \
    return COMO_OK;                                                                                                              \
  }
COMO_DEF_KFGLUE(f32, float)
COMO_DEF_KFGLUE(f64, double)

/* normalize_coordinates + interpolate_kernel_params at a list of points: coords of type f64 or f32 (coords_is_f64), covariance
 * image / outputs f32 or f64 (out_is_f64; f32 coordinates with f64 outputs are not a combination the path has) */
int como_cov_params_at(const void* cov, int Hc, int Wc, const void* coords, int coords_is_f64, int out_is_f64, int N, void* cn,
                       void* E, int B, como_stream_t stream) {
  if (!cov || !coords || !cn || !E || B <= 0 || N < 0 || Hc <= 0 || Wc <= 0 || (out_is_f64 && !coords_is_f64)) return COMO_ERR_ARG;
  if (!N) return COMO_OK;
  const dim3 grid((N + 255) / 256, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_is_f64)
    hipLaunchKernelGGL((como::cov_params_at_kernel<double, double>), grid, blk, 0, s, (const double*)cov, Hc, Wc,
                       (const double*)coords, N, (double*)cn, (double*)E);
  else if (coords_is_f64)
    hipLaunchKernelGGL((como::cov_params_at_kernel<double, float>), grid, blk, 0, s, (const float*)cov, Hc, Wc,
                       (const double*)coords, N, (float*)cn, (float*)E);
  else
    hipLaunchKernelGGL((como::cov_params_at_kernel<float, float>), grid, blk, 0, s, (const float*)cov, Hc, Wc,
                       (const float*)coords, N, (float*)cn, (float*)E);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
