// Window bundle adjustment: photometric linearisation of all keyframe pairs on the device.
//
// Reference path: como/odom/backend/photo.py:83-233 (`batch_photo_cost`); written out for gray images (c = 1) -- a colour
// pair is c entries of the pair arrays, one per channel (BAPairs::chan, include/como_hip.h como_ba_args.channels):
//   per pair (i -> j) and reference pixel n:  P_cj = T_wcj^-1 P_wn ; sample [I, gx, gy]_j ;
//   r = I_j - e^{a_j - a_i} I_i + (b_j - b_i) ; GLOBAL sigma = 1.4826 median|r| over all valid
//   pixels of all pairs ; Huber ; row = [J_i (8) | J_j (8) | J_z (m)] ; blocks J^T J, J^T r ;
//   z -> 3-D landmark expansion with the per-frame constant dz/dP_w ; scatter into dense H, g.
//
// Kernel chain (all on `stream`, no host synchronisation):
//   ba_pair_setup   : per-pair target inverse pose (exact op order), affine scale / bias
//   ba_residual     : warp + bilinear sample + residual + validity mask + pass-0 |r| histogram
//   select_hist x(P-1)
//   ba_blocks       : per (pair, pixel-chunk) workgroup; per wave: phase A = per-pixel rows (VALU,
//                     staged in LDS), phase B = the symmetric (16+m)x(16+m) Gram matrix of the rows
//                     on the MATRIX cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64, K = pixels),
//                     J^T r by VALU FMAs + wave shuffles.  J is never written to memory.
//   ba_reduce_assemble : ordered (deterministic) fp64 sum of the per-wave partials, landmark expansion,
//                     accumulation into H (both triangles) and g.
//
// Algorithmic traffic (SURVEY.md section 8d unit B): (34 + m) scalars per pixel-pair.
#include "select.cuh"
#include "../../include/como_hip.h"

#include <type_traits>
#include <utility>

#ifndef COMO_F64_PF
#define COMO_F64_PF 4      // depth of the K~ register ring of the float64 two-pair kernel
#endif

namespace como {

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0..N-1 (guarantees static register indexing)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <typename T> int select_hist(const T*, const uint8_t*, long, int, uint32_t*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
template <typename T> struct Acc4 { typedef T type __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ typename Acc4<float>::type mfma16(float a, float b, typename Acc4<float>::type c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ typename Acc4<double>::type mfma16(double a, double b, typename Acc4<double>::type c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// LDS written by this wave is visible to this wave's later reads (no other wave involved)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ float fast_rcpf(float x) {              // hardware reciprocal + one Newton step (~0.5 ulp)
  const float y = __builtin_amdgcn_rcpf(x);
  return fmaf(y, fmaf(-x, y, 1.0f), y);
}
// 1/x and 1/sqrt(x) in double from the hardware seed + one Newton step (second / third order): ~1 ulp, 4-6 instructions
// instead of the ~25 of an IEEE division or sqrt.  Only used for Jacobian WEIGHTS (never for what feeds a validity mask).
__device__ __forceinline__ double fast_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double fast_rsq(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}
// C/D register layout of the 16x16x4 MFMA: row of (lane, reg); column is lane & 15 for both.
template <typename T> __host__ __device__ constexpr int mfma_row(int lane, int reg);
template <> __host__ __device__ constexpr int mfma_row<float>(int lane, int reg) { return (lane >> 4) * 4 + reg; }
template <> __host__ __device__ constexpr int mfma_row<double>(int lane, int reg) { return (lane >> 4) + 4 * reg; }

struct BAPairs {
  const int* ref_slot;     // [b] index of the reference keyframe's slot in the per-reference arrays
  const int* ref_aff;      // [b] index into aff_all of the reference affine params
  const int* tgt_aff;      // [b] index into aff_all of the target affine params
  const int* tgt_pose;     // [b] index into poses_all of the target pose
  const long* tgt_img;     // [b] element offset of the target [I,gx,gy] stack relative to img_base
  int anorm_f32;           // sampling normalisation rounded to float32 first (two_frame_sfm.py:187-190)
  const int* chan;         // [b] image channel of each pair (NULL: 0).  A c-channel image (`color: rgb`) is linearised as c
  int C;                   //     pairs per keyframe pair, one per channel: photo.py:112-128 treats (pixel, channel) residuals alike
  const int* ref_pose;     // [b] index into poses_all of the REFERENCE keyframe's pose T_wc (zmode 2 only, else NULL)
};
// channel ch of the target stack [I_0..I_C-1 | gx_0.. | gy_0..] (photo.py:24-27, 44-52) and of the reference values (slots,n,C)
__device__ __forceinline__ int pair_chan(const BAPairs& pr, int p) { return pr.chan ? pr.chan[p] : 0; }

template <typename T>
__global__ void ba_pair_setup_kernel(const T* __restrict__ poses_all, const T* __restrict__ aff_all, BAPairs pr, int b,
                                     T* __restrict__ pair_T, T* __restrict__ pair_aff, T* __restrict__ pair_ref) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= b) return;
  invert_pose34(poses_all + 16 * (long)pr.tgt_pose[p], pair_T + 12 * (long)p);   // photo.py:105
  if (pr.ref_pose) {                                                             // zmode 2: [R_wc | t_wc] of the reference keyframe
#pragma unroll
    for (int k = 0; k < 12; ++k) pair_ref[12 * (long)p + k] = poses_all[16 * (long)pr.ref_pose[p] + k];
  }
  const T* ai = aff_all + 2 * (long)pr.ref_aff[p];
  const T* aj = aff_all + 2 * (long)pr.tgt_aff[p];
  pair_aff[2 * p + 0] = exp(aj[0] - ai[0]);                                      // photo.py:115
  pair_aff[2 * p + 1] = aj[1] - ai[1];                                           // photo.py:117
}

// Everything one (pair, pixel) needs from the warp; shared by pass 1 and pass 2.
template <typename T>
struct Warp {
  T X, Y, Z, u, v;
  bool ok;
};

template <typename T>
__device__ __forceinline__ Warp<T> warp_point(const T* __restrict__ M, T fx, T fy, T cx, T cy, T Px, T Py, T Pz, int H, int W) {
  Warp<T> w;
  rigid_apply(M, Px, Py, Pz, w.X, w.Y, w.Z);            // photo.py:106, transforms.py:17-23
  w.u = project1(fx, w.X, w.Z, cx);                     // camera.py:20-26
  w.v = project1(fy, w.Y, w.Z, cy);
  w.ok = in_image(w.u, w.v, H, W) && (w.Z > T(0));      // photo.py:15-21
  return w;
}

// zmode 2 (compact dense reference): the block kernels rebuild the reference-pose block from P_w, the reference pose
// [R | t] = T_wc and the six dot products dl = K~[n,:] dlogz_m/dT_wc instead of loading 18 + 3 planes:
//   dP_w/dT_wc = [-[u]x R , R] + u (x) dl   with u = R ray z_n = P_w - t   (densify.hip dense_ref: -(R [P_c]x) = -[u]x R)
//   b^T dP_w/dT_wc = [ R^T (u x b) , R^T b ] + (b . u) dl
// Returns u . b (the depth scale before the row weight) and the six geometric entries jr[0..5]; the caller adds
// (s b.u) dl[k].  Only Jacobian VALUES depend on this (never a validity mask).
// Precision of u = P_w - t (the materialised form stored u = R (ray z_n) directly): the subtraction is exact to one rounding of
// P_w, so u carries a relative error of eps |P_w| / |u|.  In float64 (the reference's mapping dtype, 1.1e-16) that is invisible at
// any trajectory length; in the float32 mixed-precision path (6e-8) a window |t_wc| = 100 m from the origin looking at 2 m of
// depth loses a factor 50: 3e-6 relative in the reference-pose block and the depth scale b.u -- still below the float32 bar of
// the full-size pins (2e-4 on H); beyond ~1 km from the origin use the float64 path (or hand the window over re-centred on its
// anchor keyframe: the photometric system is invariant under a common rigid shift of poses and landmarks).
template <typename T>
__device__ __forceinline__ T ref_pose_geom(const T* __restrict__ Rf, T Px, T Py, T Pz, T b0, T b1, T b2, T* __restrict__ jr) {
  const T u0 = Px - Rf[3], u1 = Py - Rf[7], u2 = Pz - Rf[11];
  const T w0 = u1 * b2 - u2 * b1, w1 = u2 * b0 - u0 * b2, w2 = u0 * b1 - u1 * b0;      // u x b
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    jr[j] = Rf[j] * w0 + Rf[4 + j] * w1 + Rf[8 + j] * w2;
    jr[3 + j] = Rf[j] * b0 + Rf[4 + j] * b1 + Rf[8 + j] * b2;
  }
  return b0 * u0 + b1 * u1 + b2 * u2;
}

// ---------------------------------------- pass 1 -------------------------------------------------
template <typename T, bool SOA>
__global__ __launch_bounds__(256) void ba_residual_kernel(
    const T* __restrict__ Pwn, const T* __restrict__ vals, BAPairs pr, const T* __restrict__ pair_T,
    const T* __restrict__ pair_aff, const T* __restrict__ img_base, const T* __restrict__ Kmat, int H, int W, int n,
    int pix_begin, int pix_end, T* __restrict__ r_out, uint8_t* __restrict__ valid_out, T* __restrict__ pj_out,
    uint32_t* __restrict__ hists) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) lh[b] = 0;
  const int p = blockIdx.y;
  const int slot = pr.ref_slot[p];
  const T* M = pair_T + 12 * (long)p;
  T Mr[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mr[k] = M[k];
  const T scale = pair_aff[2 * p], bias = pair_aff[2 * p + 1];
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
  const int ch = pair_chan(pr, p);
  const T* img = img_base + pr.tgt_img[p] + (long)ch * H * W;
  const long HW = (long)pr.C * H * W;
  __syncthreads();
  const int stride = gridDim.x * 256;
  const int nl = pix_end - pix_begin;                   // this rank's pixel range [pix_begin, pix_end) of every pair
  const int iters = (nl + stride - 1) / stride;         // uniform trip count (whole waves for the aggregated histogram)
  // UN pixels per trip: their P_w / I_ref loads, then their tap loads, are in flight together (one pixel per trip left the
  // kernel latency-bound: 124 MB in 45 us)
  constexpr int UN = 4;
  for (int it0 = 0; it0 < iters; it0 += UN) {
    int idx[UN];
    bool inr[UN];
    T Px[UN], Py[UN], Pz[UN], vr[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int i0 = pix_begin + (it0 + k) * stride + blockIdx.x * 256 + threadIdx.x;
      inr[k] = (it0 + k < iters) && (i0 < pix_end);
      idx[k] = inr[k] ? i0 : pix_end - 1;
      const long ri = (long)slot * n + idx[k];
      if constexpr (SOA) {
        Px[k] = Pwn[((long)slot * 3 + 0) * n + idx[k]]; Py[k] = Pwn[((long)slot * 3 + 1) * n + idx[k]]; Pz[k] = Pwn[((long)slot * 3 + 2) * n + idx[k]];
      } else { Px[k] = Pwn[3 * ri]; Py[k] = Pwn[3 * ri + 1]; Pz[k] = Pwn[3 * ri + 2]; }
      vr[k] = vals[ri * pr.C + ch];
    }
    Warp<T> w[UN];
    T It[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      w[k] = warp_point(Mr, fx, fy, cx, cy, Px[k], Py[k], Pz[k], H, W);
      Taps<T> t = make_taps(grid_position(w[k].u, W, ax), grid_position(w[k].v, H, ay), H, W);
      It[k] = tap_sum(img, t);
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const T r = It[k] - scale * vr[k] + bias;         // photo.py:114-118
      if (inr[k]) {
        const long oi = (long)p * nl + (idx[k] - pix_begin);
        r_out[oi] = r;
        valid_out[oi] = w[k].ok ? 1 : 0;
        if (pj_out) { pj_out[2 * oi] = w[k].u; pj_out[2 * oi + 1] = w[k].v; }
      }
      sel_lds_add(lh, sel_digit<KeyT>(abs_key(r), 0), inr[k] && w[k].ok);
    }
  }
  __syncthreads();
  sel_flush(lh, hists);
}

// ---------------------------------------- pass 2 -------------------------------------------------
constexpr int JP_STRIDE = 66;   // LDS row stride of the staged pose rows: conflict-free for both phases

// Row layout of one residual: 80 columns = 5 blocks of 16: block 0 = [J_i (8) | J_j (8)], blocks 1..4 = depth
// columns.  Lane-column ci of depth block t owns depth column kcol(t, ci) = 4 ci + (t - 1): every lane loads ONE
// contiguous quad (4ci..4ci+3) of the m-wide K~ / dPwn_dzm row (a fully coalesced 1 KiB wave load per 4 pixels)
// and feeds element t-1 of it to block t.  m <= 64, m % 4 == 0; columns >= m are zero.
struct BACfg {
  static constexpr int NB = 5;
  static constexpr int NT = NB * (NB + 1) / 2;         // 15 upper-triangular 16x16 tiles
  static constexpr int REC = NT * 256 + NB * 16 + 16;  // per-wave partial record (elements) = 3936
};
__host__ __device__ constexpr int kcol(int t, int ci) { return 4 * ci + (t - 1); }
// upper-triangular tile enumeration: tile tt = (ti, tj >= ti), rows in order
__host__ __device__ constexpr int tile_first(int ti) { return ti * BACfg::NB - ti * (ti - 1) / 2; }
__host__ __device__ constexpr int tile_row(int tt) {
  int ti = 0;
  while (ti + 1 < BACfg::NB && tile_first(ti + 1) <= tt) ++ti;
  return ti;
}

template <typename T> struct V4 { T x, y, z, w; };
template <typename T>
__device__ __forceinline__ V4<T> load4(const T* __restrict__ p) {
  V4<T> v;
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

// ZMODE 0: materialised dPwn_dzm (slots, n, 3, m) exactly as the reference passes it (photo.py:92)
// ZMODE 1: factored rank-1 form  dPwn_dzm[n,:,k] = uvec[n,:] * Kt[pix(n), k] * invz[k]
//          (sparse_map.py:184-230: R_wc ray z_n * K~[n,k] / z_mk), K~ read straight from the dense predictor.
template <typename T, int ZMODE>
__global__ __launch_bounds__(256) void ba_blocks_kernel(
    const T* __restrict__ Pwn, const T* __restrict__ vals, const T* __restrict__ dPwn_dTwc,
    const T* __restrict__ zjac,      // ZMODE 0: dPwn_dzm ; ZMODE 1: K~ dense (slots_kf, HW_kt, m)
    const T* __restrict__ uvec,      // ZMODE 1: (slots, n, 3)
    const int* __restrict__ pixidx,  // ZMODE 1: (slots, n) row index into K~ of that slot (nullptr -> identity)
    const T* __restrict__ invz,      // ZMODE 1: (slots, m)
    long kt_slot_stride,             // ZMODE 1: elements between consecutive slots of K~
    BAPairs pr, const T* __restrict__ pair_T, const T* __restrict__ pair_aff, const T* __restrict__ pair_ref,
    const T* __restrict__ img_base, const T* __restrict__ Kmat, int H, int W, int n, int m, int pix_begin, int pix_end,
    int chunk_len, const uint32_t* __restrict__ hists, T* __restrict__ partials, T* __restrict__ sigma_out) {
  using KeyT = typename KeyOf<T>::type;
  using Cfg = BACfg;
  using acc_t = typename Acc4<T>::type;
  __shared__ SelScratch sc;
  constexpr int STG = 16 * JP_STRIDE + 64 * 5;                  // per-wave staging elements
  constexpr int LDS_ELEMS = (4 * STG > 2 * Cfg::REC) ? 4 * STG : 2 * Cfg::REC;
  __shared__ T lds[LDS_ELEMS];                                  // staging, later reused for the cross-wave reduction

  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);          // photo.py:128
  const T info_sqrt = T(1) / sigma;                       // photo.py:68
  if (sigma_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sigma_out[0] = sigma; sigma_out[1] = (T)nv; }

  const int p = blockIdx.y;
  const int slot = pr.ref_slot[p];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  T* Jp = lds + wv * STG;                   // [16][JP_STRIDE]
  T* Sv = Jp + 16 * JP_STRIDE;         // [5][64]: 0 = r~, 1..3 = s*dI/dPw (ZMODE 0) or 1 = s*(dI/dPw . u) (ZMODE 1)

  T Mr[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mr[k] = pair_T[12 * (long)p + k];
  const T scale = pair_aff[2 * p], bias = pair_aff[2 * p + 1];
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
  const int ch = pair_chan(pr, p);
  const T* img = img_base + pr.tgt_img[p] + (long)ch * H * W;
  const long HW = (long)pr.C * H * W;

  T Rf[12];                                 // ZMODE 2: [R | t] of the reference keyframe's pose
#pragma unroll
  for (int k = 0; k < 12; ++k) Rf[k] = (ZMODE == 2) ? pair_ref[12 * (long)p + k] : T(0);
  T invz4[4] = {T(0), T(0), T(0), T(0)};
  if constexpr (ZMODE >= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) invz4[j] = (4 * c + j < m) ? invz[(long)slot * m + 4 * c + j] : T(0);
  }

  acc_t acc[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc[t] = acc_t{T(0), T(0), T(0), T(0)};
  T gacc[Cfg::NB];
#pragma unroll
  for (int t = 0; t < Cfg::NB; ++t) gacc[t] = T(0);
  T err = T(0);

  const int begin = pix_begin + blockIdx.x * chunk_len;
  const int end = min(pix_end, begin + chunk_len);
  // K~ quads of the tile are fetched BEFORE phase A (they depend only on the pixel index), PF steps deep: the
  // matrix-core phase then runs from registers/LDS only.  f32: whole tile (16 steps, 64 VGPRs); f64: ring of 8.
  constexpr int PF = (sizeof(T) == 4) ? 16 : 8;
  V4<T> kq[PF];
  for (int tile = begin + wv * 64; tile < end; tile += 256) {
    int myrow = 0;
    if constexpr (ZMODE >= 1) {
      const int i = min(tile + lane, end - 1);
      myrow = pixidx ? pixidx[(long)slot * n + i] : i;
      static_for<PF>([&](auto ic) {
        constexpr int st = decltype(ic)::value;
        const int row = __shfl(myrow, 4 * st + q, 64);
        kq[st] = V4<T>{T(0), T(0), T(0), T(0)};
        if (4 * c < m) kq[st] = load4(zjac + (long)slot * kt_slot_stride + (long)row * m + 4 * c);
      });
    }
    // ------------------------- phase A: lane = pixel ------------------------------------------
    {
      const int i = tile + lane;
      const bool inr = i < end;
      const int ic = inr ? i : (end - 1);
      const long ri = (long)slot * n + ic;
      T Px, Py, Pz;
      if constexpr (ZMODE >= 1) {       // fast path: structure-of-arrays planes (slot, component, n)
        Px = Pwn[((long)slot * 3 + 0) * n + ic]; Py = Pwn[((long)slot * 3 + 1) * n + ic]; Pz = Pwn[((long)slot * 3 + 2) * n + ic];
      } else {
        Px = Pwn[3 * ri]; Py = Pwn[3 * ri + 1]; Pz = Pwn[3 * ri + 2];
      }
      Warp<T> w = warp_point(Mr, fx, fy, cx, cy, Px, Py, Pz, H, W);
      Taps<T> tp = make_taps(grid_position(w.u, W, ax), grid_position(w.v, H, ay), H, W);
      const T It = tap_sum(img, tp), gx = tap_sum(img + HW, tp), gy = tap_sum(img + 2 * HW, tp);
      const T Iref_s = scale * vals[ri * pr.C + ch];
      const T r = It - Iref_s + bias;
      const bool ok = inr && w.ok;
      const T wr = r * info_sqrt;
      const T wgt = ok ? huber(wr) : T(0);                 // photo.py:70-72
      const T ws = sqrt(wgt);
      const T s = ok ? info_sqrt * ws : T(0);              // invalid pixels contribute exactly zero
      err += ok ? (ws * wr) * (ws * wr) : T(0);            // photo.py:79
      // dI/dP_cj = [gx gy] dp/dP_c  (camera.py:28-35), zeroed when invalid so no NaN/inf leaks through s = 0
      const T iz = ok ? T(1) / w.Z : T(0);
      const T a0 = gx * fx * iz, a1 = gy * fy * iz;
      const T a2 = -(a0 * w.X + a1 * w.Y) * iz;
      // dI/dP_w = dI/dP_c R_cw  (photo.py:135)
      const T b0 = a0 * Mr[0] + a1 * Mr[4] + a2 * Mr[8];
      const T b1 = a0 * Mr[1] + a1 * Mr[5] + a2 * Mr[9];
      const T b2 = a0 * Mr[2] + a1 * Mr[6] + a2 * Mr[10];
      // reference-pose block: dI/dP_w dP_w/dT_wci (photo.py:145)
      T bu = T(0);                        // ZMODE 2: b . u
      if constexpr (ZMODE == 2) {
        T jr[6];
        bu = ref_pose_geom(Rf, Px, Py, Pz, b0, b1, b2, jr);
        const T* D = dPwn_dTwc + (long)slot * 6 * n + ic;            // planes dlogz_n/dT_wc
#pragma unroll
        for (int k = 0; k < 6; ++k) Jp[k * JP_STRIDE + lane] = s * (jr[k] + bu * D[(long)k * n]);
      } else if constexpr (ZMODE == 1) {
        const T* D = dPwn_dTwc + (long)slot * 18 * n + ic;
#pragma unroll
        for (int k = 0; k < 6; ++k)
          Jp[k * JP_STRIDE + lane] = s * (b0 * D[(long)k * n] + b1 * D[(long)(6 + k) * n] + b2 * D[(long)(12 + k) * n]);
      } else {
        const T* D = dPwn_dTwc + 18 * ri;
#pragma unroll
        for (int k = 0; k < 6; ++k) Jp[k * JP_STRIDE + lane] = s * (b0 * D[k] + b1 * D[6 + k] + b2 * D[12 + k]);
      }
      Jp[6 * JP_STRIDE + lane] = s * Iref_s;               // photo.py:121
      Jp[7 * JP_STRIDE + lane] = -s;
      // target-pose block: dP_c/dT_wcj = [[P_c]x, -I]  (= dPc/dTcw (-Ad(T_wc)), photo.py:107,146)
      const T Xc = ok ? w.X : T(0), Yc = ok ? w.Y : T(0), Zc = ok ? w.Z : T(0);
      Jp[8 * JP_STRIDE + lane] = s * (a1 * Zc - a2 * Yc);
      Jp[9 * JP_STRIDE + lane] = s * (a2 * Xc - a0 * Zc);
      Jp[10 * JP_STRIDE + lane] = s * (a0 * Yc - a1 * Xc);
      Jp[11 * JP_STRIDE + lane] = -s * a0;
      Jp[12 * JP_STRIDE + lane] = -s * a1;
      Jp[13 * JP_STRIDE + lane] = -s * a2;
      Jp[14 * JP_STRIDE + lane] = -s * Iref_s;
      Jp[15 * JP_STRIDE + lane] = s;
      Sv[0 * 64 + lane] = s * r;                           // whitened residual r~
      if constexpr (ZMODE == 0) {
        Sv[1 * 64 + lane] = s * b0; Sv[2 * 64 + lane] = s * b1; Sv[3 * 64 + lane] = s * b2;
      } else if constexpr (ZMODE == 2) {
        Sv[1 * 64 + lane] = s * bu;
      } else {
        const T* U = uvec + (long)slot * 3 * n + ic;
        Sv[1 * 64 + lane] = s * (b0 * U[0] + b1 * U[n] + b2 * U[2 * (long)n]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ------------------------- phase B: 4 pixels per MFMA step --------------------------------
    const long zbase0 = (long)slot * n;
    static_for<16>([&](auto ic) {
      constexpr int st = decltype(ic)::value;
      const int px = 4 * st + q;                           // this lane's pixel inside the tile
      T a[Cfg::NB];
      a[0] = Jp[c * JP_STRIDE + px];
      const T rt = Sv[px];
      if constexpr (ZMODE == 0) {
        const int i = min(tile + px, end - 1);
        const T* Zr = zjac + (zbase0 + i) * 3 * (long)m;
        const T d0 = Sv[64 + px], d1 = Sv[128 + px], d2 = Sv[192 + px];
        V4<T> z0{T(0), T(0), T(0), T(0)}, z1 = z0, z2 = z0;
        if (4 * c < m) { z0 = load4(Zr + 4 * c); z1 = load4(Zr + m + 4 * c); z2 = load4(Zr + 2 * (long)m + 4 * c); }
        const T j0 = d0 * z0.x + d1 * z1.x + d2 * z2.x;
        const T j1 = d0 * z0.y + d1 * z1.y + d2 * z2.y;
        const T j2 = d0 * z0.z + d1 * z1.z + d2 * z2.z;
        const T j3 = d0 * z0.w + d1 * z1.w + d2 * z2.w;
        a[1] = j0; a[2] = j1; a[3] = j2; a[4] = j3;
      } else {
        const T sz = Sv[64 + px];
        const V4<T> k4 = kq[st % PF];
        if constexpr (PF < 16) {          // ring refill: the quad of step st + PF lands while PF steps of MFMAs run
          if (st + PF < 16) {
            const int row = __shfl(myrow, 4 * (st + PF) + q, 64);
            kq[st % PF] = V4<T>{T(0), T(0), T(0), T(0)};
            if (4 * c < m) kq[st % PF] = load4(zjac + (long)slot * kt_slot_stride + (long)row * m + 4 * c);
          }
        }
        a[1] = sz * k4.x * invz4[0]; a[2] = sz * k4.y * invz4[1]; a[3] = sz * k4.z * invz4[2]; a[4] = sz * k4.w * invz4[3];
      }
      static_for<Cfg::NB>([&](auto it) { gacc[decltype(it)::value] += a[decltype(it)::value] * rt; });
      static_for<Cfg::NT>([&](auto it) {
        constexpr int tt = decltype(it)::value;
        constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
        acc[tt] = mfma16(a[ti], a[tj], acc[tt]);
      });
    });
    __builtin_amdgcn_wave_barrier();
  }

  // ------------------------- epilogue: ordered cross-wave reduction, one record per workgroup ---
#pragma unroll
  for (int t = 0; t < Cfg::NB; ++t) {
    gacc[t] += __shfl_xor(gacc[t], 16, 64);
    gacc[t] += __shfl_xor(gacc[t], 32, 64);
  }
  err = wave_sum(err);
  auto put = [&](T* dst) {
#pragma unroll
    for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[t * 256 + rg * 64 + lane] = acc[t][rg];
    if (lane < 16) {
#pragma unroll
      for (int t = 0; t < Cfg::NB; ++t) dst[Cfg::NT * 256 + t * 16 + lane] = gacc[t];
    }
    if (lane == 0) dst[Cfg::NT * 256 + Cfg::NB * 16] = err;
  };
  auto add = [&](const T* src) {
#pragma unroll
    for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) acc[t][rg] += src[t * 256 + rg * 64 + lane];
#pragma unroll
    for (int t = 0; t < Cfg::NB; ++t) gacc[t] += src[Cfg::NT * 256 + t * 16 + (lane & 15)];
    err += src[Cfg::NT * 256 + Cfg::NB * 16];
  };
  __syncthreads();                                   // every wave is done with its staging area
  if (wv >= 2) put(lds + (wv - 2) * Cfg::REC);
  __syncthreads();
  if (wv < 2) add(lds + wv * Cfg::REC);              // wave0 += wave2, wave1 += wave3
  __syncthreads();
  if (wv == 1) put(lds);
  __syncthreads();
  if (wv == 0) {
    add(lds);
    put(partials + (long)(p * gridDim.x + blockIdx.x) * Cfg::REC);
  }
}

// ---------------------------------------- pass 2, software-pipelined fast path ---------------------
// Same maths as ba_blocks_kernel<T, 1>, restructured so that NO memory latency sits between two matrix-core phases
// of a wave (PMC on the straightforward version: MFMA pipe 44 % busy, 20 k of 28 k wave-cycles per tile spent
// outside phase B, mostly waiting on the two dependent load rounds of phase A).  Per wave and tile t:
//     S2(t)   consume the loads issued one tile ago -> 16 pose/affine entries + depth scale + r~ -> LDS
//     S1(t+1) warp the NEXT tile's points (loaded two stages ago), issue its 12 tap loads + 7 plane loads
//     S0(t+2) issue the P_w loads of the tile after that
//     S3(t)   16 MFMA steps; after consuming K~ quad `st` the same register is re-loaded for tile t+1
// so every load has a full matrix phase (~7.7 k cycles) to land.  Compact dense reference (zmode 2): `dlz` holds the six
// planes dlogz_n/dT_wc; the reference-pose block is rebuilt from P_w and the reference pose (ref_pose_geom).
template <typename T, int WPS, int ABL = 0>   // ABL (ablation, timing only): 1 = no MFMA, 2 = no S2 arithmetic, 3 = no loads in S1
__global__ __launch_bounds__(256, WPS) void ba_blocks_pipe_kernel(
    const T* __restrict__ Pwn, const T* __restrict__ vals, const T* __restrict__ dlz, const T* __restrict__ Kt,
    const int* __restrict__ pixidx, const T* __restrict__ invz, long kt_slot_stride,
    BAPairs pr, const T* __restrict__ pair_T, const T* __restrict__ pair_aff, const T* __restrict__ pair_ref, const T* __restrict__ img_base,
    const T* __restrict__ Kmat, int H, int W, int n, int m, int pix_begin, int pix_end, int chunk_len,
    const uint32_t* __restrict__ hists, T* __restrict__ partials, T* __restrict__ sigma_out, int stagger,
    const int* __restrict__ pair_map) {
  using KeyT = typename KeyOf<T>::type;
  using Cfg = BACfg;
  using acc_t = typename Acc4<T>::type;
  __shared__ SelScratch sc;
  constexpr int STG = 16 * JP_STRIDE + 64 * 2;
  constexpr int LDS_ELEMS = (4 * STG > 2 * Cfg::REC) ? 4 * STG : 2 * Cfg::REC;
  __shared__ T lds[LDS_ELEMS];

  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);
  const T info_sqrt = T(1) / sigma;
  if (sigma_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sigma_out[0] = sigma; sigma_out[1] = (T)nv; }

  const int p = pair_map ? pair_map[blockIdx.y] : blockIdx.y;
  const int slot = pr.ref_slot[p];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  T* Jp = lds + wv * STG;
  T* Sv = Jp + 16 * JP_STRIDE;              // [0] r~, [1] s (dI/dPw . u)
  T Mr[12], Rf[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) { Mr[k] = pair_T[12 * (long)p + k]; Rf[k] = pair_ref[12 * (long)p + k]; }
  const T scale = pair_aff[2 * p], bias = pair_aff[2 * p + 1];
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
  const int ch = pair_chan(pr, p);
  const T* img = img_base + pr.tgt_img[p] + (long)ch * H * W;
  const long HW = (long)pr.C * H * W;
  T invz4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) invz4[j] = (4 * c + j < m) ? invz[(long)slot * m + 4 * c + j] : T(0);
  // lanes whose quad lies beyond m read quad 0 instead (finite values) and are nulled by invz4 = 0: keeps every load
  // unconditional, so the compiler can count outstanding loads exactly (no exec-mask branches -> no vmcnt(0) stalls)
  const T* KtS = Kt + (long)slot * kt_slot_stride + ((4 * c < m) ? 4 * c : 0);

  acc_t acc[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc[t] = acc_t{T(0), T(0), T(0), T(0)};
  T gacc[Cfg::NB];
#pragma unroll
  for (int t = 0; t < Cfg::NB; ++t) gacc[t] = T(0);
  T err = T(0);

  const int begin = pix_begin + blockIdx.x * chunk_len;
  const int end = min(pix_end, begin + chunk_len);
  constexpr int PF = 8;
  V4<T> kq[PF];

  // ---- pipeline registers ----
  T pw0, pw1, pw2;                     // S0: P_w of the tile that S1 will warp next
  T tv[12], Dv[6], pq0 = 0, pq1 = 0, pq2 = 0, valv;   // S1: loaded values (+ the warped tile's P_w) of the tile S2 will consume
  T wX = 0, wY = 0, wZ = 0, w00 = 0, w01 = 0, w10 = 0, w11 = 0;
  bool wok = false;
  int row_cur = 0, row_nxt = 0;

  auto s0_load = [&](int tile) {
    const int ic = min(tile + lane, end - 1);
    pw0 = Pwn[((long)slot * 3 + 0) * n + ic]; pw1 = Pwn[((long)slot * 3 + 1) * n + ic]; pw2 = Pwn[((long)slot * 3 + 2) * n + ic];
  };
  auto s1_issue = [&](int tile) {
    const int i = tile + lane;
    const bool inr = i < end;
    const int ic = inr ? i : (end - 1);
    Warp<T> w = warp_point(Mr, fx, fy, cx, cy, pw0, pw1, pw2, H, W);
    Taps<T> tp = make_taps(grid_position(w.u, W, ax), grid_position(w.v, H, ay), H, W);
    wX = w.X; wY = w.Y; wZ = w.Z; wok = inr && w.ok;
    pq0 = pw0; pq1 = pw1; pq2 = pw2;
    w00 = tp.w00; w01 = tp.w01; w10 = tp.w10; w11 = tp.w11;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const T* P = img + pl * HW;
      tv[4 * pl + 0] = P[tp.i00]; tv[4 * pl + 1] = P[tp.i01]; tv[4 * pl + 2] = P[tp.i10]; tv[4 * pl + 3] = P[tp.i11];
    }
    const T* D = dlz + (long)slot * 6 * n + ic;
#pragma unroll
    for (int k = 0; k < 6; ++k) Dv[k] = D[(long)k * n];
    valv = vals[((long)slot * n + ic) * pr.C + ch];
    row_nxt = pixidx ? pixidx[(long)slot * n + ic] : ic;
  };
  auto s2_rows = [&]() {
    if constexpr (ABL == 2) {
#pragma unroll
      for (int k2 = 0; k2 < 12; ++k2) asm volatile("" ::"v"(tv[k2]));
#pragma unroll
      for (int k2 = 0; k2 < 6; ++k2) asm volatile("" ::"v"(Dv[k2]));
      asm volatile("" ::"v"(valv));
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) Jp[k2 * JP_STRIDE + lane] = T(0.001) * T(k2 + lane);
      Sv[lane] = T(0.01); Sv[64 + lane] = T(0.02);
      return;
    }
    const T It = w00 * tv[0] + w01 * tv[1] + w10 * tv[2] + w11 * tv[3];
    const T gx = w00 * tv[4] + w01 * tv[5] + w10 * tv[6] + w11 * tv[7];
    const T gy = w00 * tv[8] + w01 * tv[9] + w10 * tv[10] + w11 * tv[11];
    const T Iref_s = scale * valv;
    const T r = It - Iref_s + bias;
    const bool ok = wok;
    const T wr = r * info_sqrt;
    const T wgt = ok ? huber(wr) : T(0);
    const T ws = sqrt(wgt);
    const T s = ok ? info_sqrt * ws : T(0);
    err += ok ? (ws * wr) * (ws * wr) : T(0);
    const T iz = ok ? T(1) / wZ : T(0);
    const T a0 = gx * fx * iz, a1 = gy * fy * iz;
    const T a2 = -(a0 * wX + a1 * wY) * iz;
    const T b0 = a0 * Mr[0] + a1 * Mr[4] + a2 * Mr[8];
    const T b1 = a0 * Mr[1] + a1 * Mr[5] + a2 * Mr[9];
    const T b2 = a0 * Mr[2] + a1 * Mr[6] + a2 * Mr[10];
    T jr[6];
    const T bu = ref_pose_geom(Rf, pq0, pq1, pq2, b0, b1, b2, jr);
#pragma unroll
    for (int k = 0; k < 6; ++k) Jp[k * JP_STRIDE + lane] = s * (jr[k] + bu * Dv[k]);
    Jp[6 * JP_STRIDE + lane] = s * Iref_s;
    Jp[7 * JP_STRIDE + lane] = -s;
    const T Xc = ok ? wX : T(0), Yc = ok ? wY : T(0), Zc = ok ? wZ : T(0);
    Jp[8 * JP_STRIDE + lane] = s * (a1 * Zc - a2 * Yc);
    Jp[9 * JP_STRIDE + lane] = s * (a2 * Xc - a0 * Zc);
    Jp[10 * JP_STRIDE + lane] = s * (a0 * Yc - a1 * Xc);
    Jp[11 * JP_STRIDE + lane] = -s * a0;
    Jp[12 * JP_STRIDE + lane] = -s * a1;
    Jp[13 * JP_STRIDE + lane] = -s * a2;
    Jp[14 * JP_STRIDE + lane] = -s * Iref_s;
    Jp[15 * JP_STRIDE + lane] = s;
    Sv[lane] = s * r;
    Sv[64 + lane] = s * bu;
  };

  // Phase stagger: the two waves that share a SIMD (one from each co-resident workgroup) would otherwise run in
  // lockstep -- both in the VALU stages, then both fighting for the matrix pipe.  Odd hardware wave slots start half a
  // tile period later, so one wave's S3 overlaps the other's S2/S1 from then on (all tiles take the same time).
  if (stagger > 0) {
    const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID.WAVE_ID
    if (wave_slot & 1u)
      for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(16);                   // 16 x 64 cycles each
  }
  const int tile0 = begin + wv * 64;
  if (tile0 < end) {
    // prologue: tile0 through S0+S1, its first PF K~ quads, S0 of the next tile
    s0_load(tile0);
    s1_issue(tile0);
    row_cur = row_nxt;
    static_for<PF>([&](auto ic_) {
      constexpr int st = decltype(ic_)::value;
      const int row = __shfl(row_cur, 4 * st + q, 64);
      kq[st] = load4(KtS + (long)row * m);
    });
    s0_load(tile0 + 256);
  }
  for (int tile = tile0; tile < end; tile += 256) {
    s2_rows();                                   // S2(t)
    s1_issue(tile + 256);                        // S1(t+1): its loads fly during S3(t) (clamped + masked past the end)
    s0_load(tile + 512);                         // S0(t+2)
    __builtin_amdgcn_wave_barrier();
    for (int half = 0; half < 16 / PF; ++half) {  // S3(t): PF steps unrolled (static ring index), 16/PF rounds
      static_for<PF>([&](auto ic_) {
        constexpr int sl = decltype(ic_)::value;
        const int st = half * PF + sl;
        const int px = 4 * st + q;
        T a[Cfg::NB];
        a[0] = Jp[c * JP_STRIDE + px];
        const T rt = Sv[px];
        const T sz = Sv[64 + px];
        const V4<T> k4 = kq[sl];
        {   // refill this ring slot: step st+PF of this tile, or step st+PF-16 of the next tile
          const int nst = st + PF;
          const bool same = nst < 16;          // (for the last tile row_nxt is a clamped in-range row: a harmless re-read)
          const int row = __shfl(same ? row_cur : row_nxt, 4 * (same ? nst : nst - 16) + q, 64);
          kq[sl] = load4(KtS + (long)row * m);
        }
        a[1] = sz * k4.x * invz4[0]; a[2] = sz * k4.y * invz4[1]; a[3] = sz * k4.z * invz4[2]; a[4] = sz * k4.w * invz4[3];
        static_for<Cfg::NB>([&](auto it) { gacc[decltype(it)::value] += a[decltype(it)::value] * rt; });
        if constexpr (ABL == 1) {
          asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]));
        } else {
          static_for<Cfg::NT>([&](auto it) {
            constexpr int tt = decltype(it)::value;
            constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
            acc[tt] = mfma16(a[ti], a[tj], acc[tt]);
          });
        }
      });
    }
    __builtin_amdgcn_wave_barrier();
    row_cur = row_nxt;
  }

  // ---- epilogue: ordered cross-wave reduction, one record per workgroup (identical to ba_blocks_kernel) ----
#pragma unroll
  for (int t = 0; t < Cfg::NB; ++t) {
    gacc[t] += __shfl_xor(gacc[t], 16, 64);
    gacc[t] += __shfl_xor(gacc[t], 32, 64);
  }
  err = wave_sum(err);
  auto put = [&](T* dst) {
#pragma unroll
    for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[t * 256 + rg * 64 + lane] = acc[t][rg];
    if (lane < 16) {
#pragma unroll
      for (int t = 0; t < Cfg::NB; ++t) dst[Cfg::NT * 256 + t * 16 + lane] = gacc[t];
    }
    if (lane == 0) dst[Cfg::NT * 256 + Cfg::NB * 16] = err;
  };
  auto add = [&](const T* src) {
#pragma unroll
    for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) acc[t][rg] += src[t * 256 + rg * 64 + lane];
#pragma unroll
    for (int t = 0; t < Cfg::NB; ++t) gacc[t] += src[Cfg::NT * 256 + t * 16 + (lane & 15)];
    err += src[Cfg::NT * 256 + Cfg::NB * 16];
  };
  __syncthreads();
  if (wv >= 2) put(lds + (wv - 2) * Cfg::REC);
  __syncthreads();
  if (wv < 2) add(lds + wv * Cfg::REC);
  __syncthreads();
  if (wv == 1) put(lds);
  __syncthreads();
  if (wv == 0) {
    add(lds);
    put(partials + (long)(p * gridDim.x + blockIdx.x) * Cfg::REC);
  }
}

// ---------------------------------------- pass 2, two pairs sharing their reference keyframe ---------------------------
// Consecutive-keyframe graphs give every inner keyframe TWO pairs with the same reference (i -> i+1, i -> i-1).  Both
// rows of a reference pixel carry the SAME K~ row, scaled by their own s_g: the depth x depth block of the normal
// equations only needs  sum_g s_g^2 K~ K~^T  -- one weighted Gram instead of two.  One workgroup walks the reference
// pixels ONCE for both pairs: P_w, dlogz_n/dT_wc (compact dense reference, zmode 2), vals and the K~ ring are loaded once, the
// matrix work per pixel is 10 (zz, weight sqrt(s_0^2 + s_1^2)) + 2 x 5 (pose x pose, pose x depth with the pose row
// rescaled by s_g / sqrt(s_0^2 + s_1^2)) = 20 tile-steps instead of 30.  Records: pair 0 carries the zz tiles and the
// depth gradient, pair 1's are zero -- the assembly kernel is unchanged.  float32 only.
template <int WPS, int PF>
__global__ __launch_bounds__(256, WPS) void ba_blocks_pair2_kernel(
    const float* __restrict__ Pwn, const float* __restrict__ vals, const float* __restrict__ dlz,
    const float* __restrict__ Kt, const int* __restrict__ pixidx,
    const float* __restrict__ invz, long kt_slot_stride, BAPairs pr, const float* __restrict__ pair_T,
    const float* __restrict__ pair_aff, const float* __restrict__ pair_ref, const float* __restrict__ img_base, const float* __restrict__ Kmat, int H, int W, int n,
    int m, int pix_begin, int pix_end, int chunk_len, const uint32_t* __restrict__ hists, float* __restrict__ partials,
    float* __restrict__ sigma_out, const int* __restrict__ grp_pairs) {
  using T = float;
  using KeyT = typename KeyOf<T>::type;
  using Cfg = BACfg;
  using acc_t = typename Acc4<T>::type;
  constexpr int G = 2;
  __shared__ SelScratch sc;
  constexpr int STG1 = 16 * JP_STRIDE + 64 * 2;     // per pair: 16 pose/affine rows, r~, depth scale / joint weight
  constexpr int STG = G * STG1 + 64 * 2;           // + per pixel of the tile: joint weight sqrt(s0^2 + s1^2), joint whitened residual
  constexpr int LDS_ELEMS = (4 * STG > 2 * Cfg::REC) ? 4 * STG : 2 * Cfg::REC;
  __shared__ T lds[LDS_ELEMS];

  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);
  const T info_sqrt = T(1) / sigma;
  if (sigma_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sigma_out[0] = sigma; sigma_out[1] = (T)nv; }

  // a row {p, -1} is a reference keyframe with a single pair: its second lane set is masked off (rows of zeros)
  const int pg0 = grp_pairs[2 * blockIdx.y], pg1 = grp_pairs[2 * blockIdx.y + 1];
  const bool has1 = pg1 >= 0;
  const int pg[G] = {pg0, has1 ? pg1 : pg0};
  const int slot = pr.ref_slot[pg[0]];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  T* Jp[G];
  T* Sv[G];
  T* Px = lds + wv * STG + G * STG1;
  T Mr[G][12], scale[G], bias[G];
  const T* img[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    Jp[g] = lds + wv * STG + g * STG1;
    Sv[g] = Jp[g] + 16 * JP_STRIDE;
#pragma unroll
    for (int k = 0; k < 12; ++k) Mr[g][k] = pair_T[12 * (long)pg[g] + k];
    scale[g] = pair_aff[2 * pg[g]];
    bias[g] = pair_aff[2 * pg[g] + 1];
    img[g] = img_base + pr.tgt_img[pg[g]] + (long)pair_chan(pr, pg[g]) * H * W;
  }
  T Rf[12];                                         // [R | t] of the shared reference keyframe
#pragma unroll
  for (int k = 0; k < 12; ++k) Rf[k] = pair_ref[12 * (long)pg[0] + k];
  const int ch = pair_chan(pr, pg[0]);              // both pairs of a group share reference slot AND channel (one I_ref load)
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
  const long HW = (long)pr.C * H * W;
  T invz4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) invz4[j] = (4 * c + j < m) ? invz[(long)slot * m + 4 * c + j] : T(0);
  const T* KtS = Kt + (long)slot * kt_slot_stride + ((4 * c < m) ? 4 * c : 0);

  acc_t azz[10], aTT[G], aTz[G][4];
#pragma unroll
  for (int t = 0; t < 10; ++t) azz[t] = acc_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G; ++g) {
    aTT[g] = acc_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) aTz[g][e] = acc_t{0.f, 0.f, 0.f, 0.f};
  }
  T gT[G] = {0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f}, err[G] = {0.f, 0.f};

  const int begin = pix_begin + blockIdx.x * chunk_len;
  const int end = min(pix_end, begin + chunk_len);
  V4<T> kq[PF];

  // ---- pipeline registers (shared: P_w, Dv, valv, rows; per pair: warp state + taps) ----
  T pw0, pw1, pw2, pq0 = 0.f, pq1 = 0.f, pq2 = 0.f;
  T Dv[6], valv;
  T tv[G][12];
  T wX[G], wY[G], wZ[G], w00[G], w01[G], w10[G], w11[G];
  bool wok[G];
  int row_cur = 0, row_nxt = 0;

  auto s0_load = [&](int tile) {
    const int ic = min(tile + lane, end - 1);
    pw0 = Pwn[((long)slot * 3 + 0) * n + ic]; pw1 = Pwn[((long)slot * 3 + 1) * n + ic]; pw2 = Pwn[((long)slot * 3 + 2) * n + ic];
  };
  auto s1_issue = [&](int tile) {
    const int i = tile + lane;
    const bool inr = i < end;
    const int ic = inr ? i : (end - 1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      Warp<T> w = warp_point(Mr[g], fx, fy, cx, cy, pw0, pw1, pw2, H, W);
      Taps<T> tp = make_taps(grid_position(w.u, W, ax), grid_position(w.v, H, ay), H, W);
      wX[g] = w.X; wY[g] = w.Y; wZ[g] = w.Z; wok[g] = inr && w.ok && (g == 0 || has1);
      w00[g] = tp.w00; w01[g] = tp.w01; w10[g] = tp.w10; w11[g] = tp.w11;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const T* P = img[g] + pl * HW;
        tv[g][4 * pl + 0] = P[tp.i00]; tv[g][4 * pl + 1] = P[tp.i01]; tv[g][4 * pl + 2] = P[tp.i10]; tv[g][4 * pl + 3] = P[tp.i11];
      }
    }
    pq0 = pw0; pq1 = pw1; pq2 = pw2;
    const T* D = dlz + (long)slot * 6 * n + ic;
#pragma unroll
    for (int k = 0; k < 6; ++k) Dv[k] = D[(long)k * n];
    valv = vals[((long)slot * n + ic) * pr.C + ch];
    row_nxt = pixidx ? pixidx[(long)slot * n + ic] : ic;
  };
  auto s2_rows = [&]() {
    T rsv[G], szv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const T It = w00[g] * tv[g][0] + w01[g] * tv[g][1] + w10[g] * tv[g][2] + w11[g] * tv[g][3];
      const T gx = w00[g] * tv[g][4] + w01[g] * tv[g][5] + w10[g] * tv[g][6] + w11[g] * tv[g][7];
      const T gy = w00[g] * tv[g][8] + w01[g] * tv[g][9] + w10[g] * tv[g][10] + w11[g] * tv[g][11];
      const T Iref_s = scale[g] * valv;
      const T r = It - Iref_s + bias[g];
      const bool ok = wok[g];
      const T wr = r * info_sqrt;
      // WEIGHTS of the Jacobian rows (never what feeds a validity mask): hardware reciprocal + one Newton step and the
      // hardware square root (1 ulp) instead of the IEEE division / sqrt expansions (~30 instructions per pair and pixel)
      const T awr = fabsf(wr);
      const T hub = (awr < T(1.345)) ? T(1) : T(1.345) * fast_rcpf(awr);          // robust_loss.py:9-16
      const T ws = ok ? __builtin_amdgcn_sqrtf(hub) : T(0);
      const T s = info_sqrt * ws;
      err[g] += (ws * wr) * (ws * wr);
      const T iz = ok ? fast_rcpf(wZ[g]) : T(0);
      const T a0 = gx * fx * iz, a1 = gy * fy * iz;
      const T a2 = -(a0 * wX[g] + a1 * wY[g]) * iz;
      const T b0 = a0 * Mr[g][0] + a1 * Mr[g][4] + a2 * Mr[g][8];
      const T b1 = a0 * Mr[g][1] + a1 * Mr[g][5] + a2 * Mr[g][9];
      const T b2 = a0 * Mr[g][2] + a1 * Mr[g][6] + a2 * Mr[g][10];
      T* J = Jp[g];
      T jr[6];
      const T bu = ref_pose_geom(Rf, pq0, pq1, pq2, b0, b1, b2, jr);
#pragma unroll
      for (int k = 0; k < 6; ++k) J[k * JP_STRIDE + lane] = s * (jr[k] + bu * Dv[k]);
      J[6 * JP_STRIDE + lane] = s * Iref_s;
      J[7 * JP_STRIDE + lane] = -s;
      const T Xc = ok ? wX[g] : T(0), Yc = ok ? wY[g] : T(0), Zc = ok ? wZ[g] : T(0);
      J[8 * JP_STRIDE + lane] = s * (a1 * Zc - a2 * Yc);
      J[9 * JP_STRIDE + lane] = s * (a2 * Xc - a0 * Zc);
      J[10 * JP_STRIDE + lane] = s * (a0 * Yc - a1 * Xc);
      J[11 * JP_STRIDE + lane] = -s * a0;
      J[12 * JP_STRIDE + lane] = -s * a1;
      J[13 * JP_STRIDE + lane] = -s * a2;
      J[14 * JP_STRIDE + lane] = -s * Iref_s;
      J[15 * JP_STRIDE + lane] = s;
      rsv[g] = s * r;
      szv[g] = s * bu;
      Sv[g][lane] = rsv[g];
    }
    // per-PIXEL factors of the joint depth row, once per tile (lane = pixel) instead of in every step of every column lane
    const T cs = szv[0] * szv[0] + szv[1] * szv[1];
    const T rs = cs > T(0) ? __builtin_amdgcn_rsqf(cs) : T(0);
    Px[lane] = cs * rs;                                          // sqrt(s_0^2 + s_1^2)
    Px[64 + lane] = (szv[0] * rsv[0] + szv[1] * rsv[1]) * rs;    // whitened residual of the joint depth row
    Sv[0][64 + lane] = szv[0] * rs;                              // pose rows rescaled for the pose x depth tiles
    Sv[1][64 + lane] = szv[1] * rs;
  };

  const int tile0 = begin + wv * 64;
  if (tile0 < end) {
    s0_load(tile0);
    s1_issue(tile0);
    row_cur = row_nxt;
    static_for<PF>([&](auto ic_) {
      constexpr int st = decltype(ic_)::value;
      const int row = __shfl(row_cur, 4 * st + q, 64);
      kq[st] = load4(KtS + (long)row * m);
    });
    s0_load(tile0 + 256);
  }
  for (int tile = tile0; tile < end; tile += 256) {
    s2_rows();
    s1_issue(tile + 256);
    s0_load(tile + 512);
    __builtin_amdgcn_wave_barrier();
    // (rolled on purpose, and the K~ rows come through the row table even when it is the identity: the fully unrolled
    //  matrix phase, or a computed row index, doubles the kernel's time -- the register allocation has no slack)
    for (int half = 0; half < 16 / PF; ++half) {
      static_for<PF>([&](auto ic_) {
        constexpr int sl = decltype(ic_)::value;
        const int st = half * PF + sl;
        const int px = 4 * st + q;
        const T a00 = Jp[0][c * JP_STRIDE + px], a01 = Jp[1][c * JP_STRIDE + px];
        const T rt0 = Sv[0][px], rt1 = Sv[1][px];
        const T sc2 = Px[px], gzs = Px[64 + px];
        const V4<T> k4 = kq[sl];
        {
          const int nst = st + PF;
          const bool same = nst < 16;
          const int row = __shfl(same ? row_cur : row_nxt, 4 * (same ? nst : nst - 16) + q, 64);
          kq[sl] = load4(KtS + (long)row * m);
        }
        // depth columns WITHOUT the 1 / z_m factor: constant over pixels, applied to the accumulators once in the epilogue
        const T zq[4] = {sc2 * k4.x, sc2 * k4.y, sc2 * k4.z, sc2 * k4.w};
        const T p0 = a00 * Sv[0][64 + px], p1 = a01 * Sv[1][64 + px];
        gT[0] += a00 * rt0;
        gT[1] += a01 * rt1;
#pragma unroll
        for (int e = 0; e < 4; ++e) gz[e] += zq[e] * gzs;
        aTT[0] = mfma16(a00, a00, aTT[0]);
        aTT[1] = mfma16(a01, a01, aTT[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          aTz[0][e] = mfma16(p0, zq[e], aTz[0][e]);
          aTz[1][e] = mfma16(p1, zq[e], aTz[1][e]);
        }
        static_for<10>([&](auto it) {
          constexpr int tt = decltype(it)::value + 5;          // tiles 5..14 of the 15-tile enumeration = depth x depth
          constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
          azz[tt - 5] = mfma16(zq[ti - 1], zq[tj - 1], azz[tt - 5]);
        });
      });
    }
    __builtin_amdgcn_wave_barrier();
    row_cur = row_nxt;
  }

  // the 1 / z_m factors of the depth columns (dPwn_dzm = u K~ / z_m), once per accumulator: column of depth block t at
  // lane-column ci is kcol(t, ci); the f32 MFMA row of (lane, reg) is 4 (lane >> 4) + reg
  static_for<10>([&](auto it) {
    constexpr int tt = decltype(it)::value + 5;
    constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int k1 = kcol(ti, mfma_row<T>(lane, rg));
      const T f1 = (k1 < m) ? invz[(long)slot * m + k1] : T(0);
      azz[tt - 5][rg] *= f1 * invz4[tj - 1];
    }
  });
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gz[e] *= invz4[e];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) aTz[g][e][rg] *= invz4[e];
  }
  // ---- epilogue: one record per pair, ordered cross-wave reduction (record layout of ba_blocks_kernel) ----
#pragma unroll
  for (int g = 0; g < G; ++g) {
    gT[g] += __shfl_xor(gT[g], 16, 64);
    gT[g] += __shfl_xor(gT[g], 32, 64);
    err[g] = wave_sum(err[g]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gz[e] += __shfl_xor(gz[e], 16, 64);
    gz[e] += __shfl_xor(gz[e], 32, 64);
  }
  static_for<G>([&](auto ig) {
    constexpr int g = decltype(ig)::value;
    acc_t acc[Cfg::NT];
    T gacc[Cfg::NB];
    acc[0] = aTT[g];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[1 + e] = aTz[g][e];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[5 + t] = (g == 0) ? azz[t] : acc_t{0.f, 0.f, 0.f, 0.f};
    gacc[0] = gT[g];
#pragma unroll
    for (int e = 0; e < 4; ++e) gacc[1 + e] = (g == 0) ? gz[e] : T(0);
    T er = err[g];
    auto put = [&](T* dst) {
#pragma unroll
      for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) dst[t * 256 + rg * 64 + lane] = acc[t][rg];
      if (lane < 16) {
#pragma unroll
        for (int t = 0; t < Cfg::NB; ++t) dst[Cfg::NT * 256 + t * 16 + lane] = gacc[t];
      }
      if (lane == 0) dst[Cfg::NT * 256 + Cfg::NB * 16] = er;
    };
    auto add = [&](const T* src) {
#pragma unroll
      for (int t = 0; t < Cfg::NT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[t][rg] += src[t * 256 + rg * 64 + lane];
#pragma unroll
      for (int t = 0; t < Cfg::NB; ++t) gacc[t] += src[Cfg::NT * 256 + t * 16 + (lane & 15)];
      er += src[Cfg::NT * 256 + Cfg::NB * 16];
    };
    __syncthreads();
    if (wv >= 2) put(lds + (wv - 2) * Cfg::REC);
    __syncthreads();
    if (wv < 2) add(lds + wv * Cfg::REC);
    __syncthreads();
    if (wv == 1) put(lds);
    __syncthreads();
    if (wv == 0 && (g == 0 || has1)) {
      add(lds);
      put(partials + (long)(pg[g] * gridDim.x + blockIdx.x) * Cfg::REC);
    }
  });
}

// ---------------------------------------- pass 2, float64: two pairs, role-specialised wave pair -------------------------
// The reference's mapping dtype is double (config/como.yml:28).  In double the 20 accumulator tiles of the two-pair kernel
// are 160 registers and its pipeline state another ~100: more than one wave can hold without spilling.  Here a workgroup
// is TWO waves that walk the same 64-pixel tiles of one reference keyframe and split the work by ROLE:
//   wave 0: warp / taps / Jacobian row of pair 0 (stages S1, S2)  +  the 10 depth x depth tiles (weight sqrt(s0^2 + s1^2))
//   wave 1: warp / taps / Jacobian row of pair 1                  +  the 2 pose x pose and 8 pose x depth tiles
// = 10 v_mfma_f64_16x16x4_f64 per 4-pixel step and 80 accumulator registers per wave.  The staged rows go through ONE LDS
// buffer (two workgroup barriers per tile: 23 KB per workgroup, so four workgroups fit a CU); P_w / dlogz_n/dT_wc and the
// K~ quads are loaded by both waves (the second read of a line is an L1 / L2 hit -- HBM sees each byte once).
// Every accumulator element is owned by exactly one wave, so there is no cross-wave reduction: each wave writes its part
// of the two per-pair records.
// Compact dense reference (zmode 2): the reference-pose block is rebuilt from P_w, the reference pose and the six planes
// dlogz_n/dT_wc (ref_pose_geom) -- 10 planes per pixel instead of 25, which is what lets the software pipeline (the NEXT
// tile's taps / planes in registers across the matrix phase) fit the 256-register budget of TWO waves per SIMD.
// PIPE = true : four-stage software pipeline, no load latency between two matrix phases of a wave.
// PIPE = false: load -> warp -> rows -> matrix phase per tile, the co-resident wave fills the gaps.
// K~ rows are addressed by 32-bit byte offsets from the (uniform) slot base: the host checks kt_slot_stride * 8 < 4 GiB.
// Round 4 (M4 = true, COMO_BA_VARIANT=13: built, parity-green, SLOWER -- 660 us against 585 -- kept as the measured answer to "can the
// float64 matrix ceiling be moved"): the Gram tiles on v_mfma_f64_4x4x4_4b_f64 instead of v_mfma_f64_16x16x4_f64.  With every CU busy
// (scripts/micro/mfma_f64_rate.hip, profiles/r4_mfma_f64_rate.txt; sclk stays at 2.39 GHz in both cases): the 16x16x4 shape
// sustains 36 / 47.5 / 49 TFLOP/s at 1 / 2 / 8 waves per SIMD (62 % of the 78.6 of the data sheet at best: 102 ... 140 cycles per
// instruction instead of 64), the 4x4x4 shape 74 ... 75 TFLOP/s at ANY occupancy (95 %: 17 cycles per instruction).  Its operand
// layout (scripts/micro/mfma_f64_4x4_layout.hip): A_blk[i][k] at lane 16 k + 4 blk + i, B_blk[k][j] at lane 16 k + 4 blk + j,
// D_blk[i][j] at lane 16 i + 4 blk + j (ONE register) -- i.e. with the operands of the 16x16x4 form (lane = 16 k + c) one
// instruction yields the four DIAGONAL 4x4 blocks of the 16x16 tile; the other twelve blocks come from the same instruction with
// one operand rotated by 4, 8, 12 lanes inside its row of 16 (two v_mov_b32 row_ror per double).  A tile = four instructions
// into the four accumulator registers it had before; only the (lane, register) -> (row, column) map of the epilogue changes:
//   B rotated by 4 d:  register d of lane 16 i + 4 blk + j = element (4 blk + i, 4 ((blk + d) & 3) + j)
//   A rotated by 4 d:  register d of lane 16 i + 4 blk + j = element (4 ((blk + d) & 3) + i, 4 blk + j)
// Why it loses inside the kernel (scripts/micro/mfma_f64_gram_step.hip: the bare step is 819 cycles against 1398): a wave-tile of the
// default kernel costs 18.6 k cycles for 160 16x16x4 instructions = 160 x 64 (the NOMINAL rate) + ~8.4 k cycles of everything else
// -- interleaved with the warp / Jacobian vector work the 16x16x4 stream does run at its data-sheet rate; it is a bare back-to-back
// stream that drops to 102 ... 140 cycles.  Both shapes have the same nominal throughput (512 flop / 16 cycles), so the 4x4x4 form
// only ADDS its 24 ... 36 rotation moves per step to the vector side: the kernel's bound is matrix + vector issue on one pipe.
// result[lane c of its row] = v[lane (c + 4 D) mod 16 of the row]   (row_ror:n: lane c receives lane (c - n) mod 16)
template <int D> __device__ __forceinline__ double row_rot4(double v) {
  if constexpr (D == 0) {
    return v;
  } else {
    constexpr int ctrl = 0x120 + (16 - 4 * D);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
}
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

__device__ __forceinline__ V4<double> load4_off(const char* __restrict__ base, uint32_t off) {
  const double2 a = *reinterpret_cast<const double2*>(base + off);
  const double2 b = *reinterpret_cast<const double2*>(base + off + 16);
  return V4<double>{a.x, a.y, b.x, b.y};
}

template <int PF, bool PIPE, int WPS, bool M4 = false>
__global__ __launch_bounds__(128, WPS) void ba_blocks_pair2_f64_kernel(
    const double* __restrict__ Pwn, const double* __restrict__ vals, const double* __restrict__ dlz,
    const double* __restrict__ Kt, const int* __restrict__ pixidx, const double* __restrict__ invz, long kt_slot_stride,
    BAPairs pr, const double* __restrict__ pair_T, const double* __restrict__ pair_aff, const double* __restrict__ pair_ref,
    const double* __restrict__ img_base, const double* __restrict__ Kmat, int H, int W, int n, int m, int pix_begin,
    int pix_end, int chunk_len, const uint32_t* __restrict__ hists, double* __restrict__ partials,
    double* __restrict__ sigma_out, const int* __restrict__ grp_pairs) {
  using T = double;
  using KeyT = typename KeyOf<T>::type;
  using Cfg = BACfg;
  using acc_t = typename Acc4<T>::type;
  __shared__ SelScratch sc;
  constexpr int STG1 = 16 * JP_STRIDE + 64 * 2;       // one pair's staged tile: 16 pose/affine rows + r~ + depth scale
  __shared__ T lds[2 * STG1];                          // [pair]
  __shared__ T pxs[2][2 * 64];                         // per wave, per pixel of the tile: {sqrt(s0^2 + s1^2), joint whitened residual}

  // robust scale from the finished histograms; sel_resolve is written for 256-thread blocks: feed it 128 threads x 2 rounds
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve_n<KeyT, 128>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);
  const T info_sqrt = T(1) / sigma;
  if (sigma_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sigma_out[0] = sigma; sigma_out[1] = (T)nv; }

  const int pg0 = grp_pairs[2 * blockIdx.y], pg1 = grp_pairs[2 * blockIdx.y + 1];
  const bool has1 = pg1 >= 0;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform
  const int q = lane >> 4, c = lane & 15;
  const int pgm = (role == 0 || !has1) ? pg0 : pg1;                       // the pair whose rows this wave produces
  const bool live = role == 0 || has1;
  const int slot = pr.ref_slot[pg0];
  T Mr[12], Rf[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) { Mr[k] = pair_T[12 * (long)pgm + k]; Rf[k] = pair_ref[12 * (long)pg0 + k]; }
  const T scale = pair_aff[2 * pgm], bias = pair_aff[2 * pgm + 1];
  const int ch = pair_chan(pr, pg0);                // both pairs of a group share reference slot AND channel
  const T* img = img_base + pr.tgt_img[pgm] + (long)pair_chan(pr, pgm) * H * W;
  const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
  const long HW = (long)pr.C * H * W;
  // lanes whose quad lies beyond m read quad 0 instead (finite values); their columns are nulled by invz = 0 in the epilogue
  const char* KtB = reinterpret_cast<const char*>(Kt + (long)slot * kt_slot_stride);
  const uint32_t cbyte = (uint32_t)(((4 * c < m) ? 4 * c : 0) * sizeof(T));
  const uint32_t row_bytes = (uint32_t)(m * sizeof(T));
  const T* PwS = Pwn + (long)slot * 3 * n;
  const T* DlS = dlz + (long)slot * 6 * n;
  const T* VaS = vals + (long)slot * n * pr.C + ch;
  const int* PiS = pixidx ? pixidx + (long)slot * n : nullptr;

  acc_t acc[10];      // role 0: the 10 depth x depth tiles; role 1: {TT_0, TT_1, Tz_0[4], Tz_1[4]}
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = acc_t{T(0), T(0), T(0), T(0)};
  T gv[4] = {T(0), T(0), T(0), T(0)};    // role 0: depth gradient of this lane's quad; role 1: gv[0], gv[1] = pose gradients
  T err = T(0);

  const int begin = pix_begin + blockIdx.x * chunk_len;
  const int end = min(pix_end, begin + chunk_len);
  V4<T> kq[PF];
  T pw0, pw1, pw2, pq0 = 0, pq1 = 0, pq2 = 0;
  T Dv[6], valv, tv[12];
  T wX = 0, wY = 0, wZ = 0, w00 = 0, w01 = 0, w10 = 0, w11 = 0;
  bool wok = false;
  uint32_t roff_cur = 0, roff_nxt = 0;   // byte offset of this lane's pixel's K~ row inside the slot

  uint32_t pix_s0 = 0;                   // K~ row index of the pixel whose P_w the S0 stage holds (fetched with it, two tiles ahead)
  auto s0_load = [&](int tile) {
    const int ic = min(tile + lane, end - 1);
    pw0 = PwS[ic]; pw1 = PwS[(long)n + ic]; pw2 = PwS[2 * (long)n + ic];
    pix_s0 = (uint32_t)(PiS ? PiS[ic] : ic);
  };
  auto s1_issue = [&](int tile) {
    const int i = tile + lane;
    const bool inr = i < end;
    const int ic = inr ? i : (end - 1);
    Warp<T> w = warp_point(Mr, fx, fy, cx, cy, pw0, pw1, pw2, H, W);
    Taps<T> tp = make_taps(grid_position(w.u, W, ax), grid_position(w.v, H, ay), H, W);
    wX = w.X; wY = w.Y; wZ = w.Z; wok = inr && w.ok && live;
    pq0 = pw0; pq1 = pw1; pq2 = pw2;
    w00 = tp.w00; w01 = tp.w01; w10 = tp.w10; w11 = tp.w11;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const T* P = img + pl * HW;
      tv[4 * pl + 0] = P[tp.i00]; tv[4 * pl + 1] = P[tp.i01]; tv[4 * pl + 2] = P[tp.i10]; tv[4 * pl + 3] = P[tp.i11];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Dv[k] = DlS[(long)k * n + ic];
    valv = VaS[(long)ic * pr.C];
    roff_nxt = pix_s0 * row_bytes;
  };
  auto kq_prefetch = [&]() {
    static_for<PF>([&](auto ic_) {
      constexpr int st = decltype(ic_)::value;
      kq[st] = load4_off(KtB, (uint32_t)__builtin_amdgcn_ds_bpermute(4 * q + 16 * st, (int)roff_cur) + cbyte);
    });
  };
  auto s2_rows = [&](T* J, T* S) {     // this wave's pair: 16 rows + r~ + depth scale of the tile loaded one stage ago
    const T It = w00 * tv[0] + w01 * tv[1] + w10 * tv[2] + w11 * tv[3];
    const T gx = w00 * tv[4] + w01 * tv[5] + w10 * tv[6] + w11 * tv[7];
    const T gy = w00 * tv[8] + w01 * tv[9] + w10 * tv[10] + w11 * tv[11];
    const T Iref_s = scale * valv;
    const T r = It - Iref_s + bias;
    const bool ok = wok;
    const T wr = r * info_sqrt;
    const T awr = fabs(wr);
    // sqrt of the Huber weight (robust_loss.py:9-16): 1 inside the band, sqrt(1.345 / |x|) outside
    const T ws = (awr < T(1.345)) ? T(1) : T(1.1597413504743201) * fast_rsq(awr);
    const T s = ok ? info_sqrt * ws : T(0);
    err += ok ? (ws * wr) * (ws * wr) : T(0);
    const T iz = ok ? fast_rcp(wZ) : T(0);
    const T a0 = gx * fx * iz, a1 = gy * fy * iz;
    const T a2 = -(a0 * wX + a1 * wY) * iz;
    const T b0 = a0 * Mr[0] + a1 * Mr[4] + a2 * Mr[8];
    const T b1 = a0 * Mr[1] + a1 * Mr[5] + a2 * Mr[9];
    const T b2 = a0 * Mr[2] + a1 * Mr[6] + a2 * Mr[10];
    T jr[6];
    const T bu = ref_pose_geom(Rf, pq0, pq1, pq2, b0, b1, b2, jr);
    const T sbu = s * bu;
#pragma unroll
    for (int k = 0; k < 6; ++k) J[k * JP_STRIDE + lane] = s * jr[k] + sbu * Dv[k];
    J[6 * JP_STRIDE + lane] = s * Iref_s;
    J[7 * JP_STRIDE + lane] = -s;
    const T Xc = ok ? wX : T(0), Yc = ok ? wY : T(0), Zc = ok ? wZ : T(0);
    J[8 * JP_STRIDE + lane] = s * (a1 * Zc - a2 * Yc);
    J[9 * JP_STRIDE + lane] = s * (a2 * Xc - a0 * Zc);
    J[10 * JP_STRIDE + lane] = s * (a0 * Yc - a1 * Xc);
    J[11 * JP_STRIDE + lane] = -s * a0;
    J[12 * JP_STRIDE + lane] = -s * a1;
    J[13 * JP_STRIDE + lane] = -s * a2;
    J[14 * JP_STRIDE + lane] = -s * Iref_s;
    J[15 * JP_STRIDE + lane] = s;
    S[lane] = s * r;
    S[64 + lane] = sbu;
  };

  if (PIPE && begin < end) {
    s0_load(begin);
    s1_issue(begin);
    roff_cur = roff_nxt;
    kq_prefetch();
    s0_load(begin + 64);
  }
  T* const Jmine = lds + role * STG1;
  const T* const S0 = lds + 16 * JP_STRIDE;
  const T* const S1 = lds + STG1 + 16 * JP_STRIDE;
  T* const my = pxs[role];
  // Every per-step address of the matrix phase is ONE base register + a compile-time offset (the instruction's offset field);
  // the 16 steps are fully unrolled with static ring slots, so the compiler counts outstanding loads exactly (a rolled loop
  // over ring rounds made it wait for vmcnt(0) -- i.e. for K~ quads issued a few hundred cycles earlier -- once per round).
  const char* const Jc = reinterpret_cast<const char*>(lds) + (c * JP_STRIDE + q) * 8;       // pose row c, pixel q of pair 0
  const char* const Sq = reinterpret_cast<const char*>(lds + 16 * JP_STRIDE) + q * 8;        // {r~, depth scale} of pixel q, pair 0
  const char* const Mq = reinterpret_cast<const char*>(my) + q * 16;                         // role 0: {joint weight, joint r~}
  const int qsel = q * 4;                                                                     // ds_bpermute address of lane q
  auto refill = [&](auto sl_, auto st_) {
    constexpr int sl = decltype(sl_)::value, nst = decltype(st_)::value + PF;
    constexpr bool same = nst < 16;
    if constexpr (PIPE || same) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(qsel + 16 * (same ? nst : nst - 16), (int)(same ? roff_cur : roff_nxt));
      kq[sl] = load4_off(KtB, off + cbyte);
    }
  };
  for (int tile = begin; tile < end; tile += 64) {
    if constexpr (!PIPE) {
      s0_load(tile);
      s1_issue(tile);
      roff_cur = roff_nxt;
      kq_prefetch();
    }
    __syncthreads();                               // the other wave is done reading the previous tile's rows
    s2_rows(Jmine, Jmine + 16 * JP_STRIDE);
    __syncthreads();                               // both pairs' rows of this tile are staged
    if constexpr (PIPE) {
      s1_issue(tile + 64);                         // (clamped past the end: harmless re-reads, masked by `inr`)
      s0_load(tile + 128);
    }
    if (role == 0) {
      // per-PIXEL factors of the joint depth row once per tile (lane = pixel): the two rows of a reference pixel share the K~ row
      // scaled by their depth scales s_g; weight of the joint row = sqrt(s_0^2 + s_1^2), and its whitened residual
      const T rt0 = S0[lane], rt1 = S1[lane], sz0 = S0[64 + lane], sz1 = S1[64 + lane];
      const T cs = sz0 * sz0 + sz1 * sz1;
      const T rs = cs > T(0) ? fast_rsq(cs) : T(0);
      my[2 * lane] = cs * rs;
      my[2 * lane + 1] = (sz0 * rt0 + sz1 * rt1) * rs;
      wave_lds_sync();
      static_for<16>([&](auto ic_) {
        constexpr int st = decltype(ic_)::value;
        constexpr int sl = st % PF;
        const double2 pa = *reinterpret_cast<const double2*>(Mq + st * 64);
        // depth columns WITHOUT the 1 / z_m factor: it is constant over pixels and is applied to the accumulators once
        const T zq[4] = {pa.x * kq[sl].x, pa.x * kq[sl].y, pa.x * kq[sl].z, pa.x * kq[sl].w};
        refill(std::integral_constant<int, sl>{}, ic_);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] += zq[e] * pa.y;
        if constexpr (M4) {
          T zr[3][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { zr[0][e] = row_rot4<1>(zq[e]); zr[1][e] = row_rot4<2>(zq[e]); zr[2][e] = row_rot4<3>(zq[e]); }
          static_for<10>([&](auto it) {
            constexpr int tt = decltype(it)::value + 5;
            constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
            acc[tt - 5][0] = mfma4(zq[ti - 1], zq[tj - 1], acc[tt - 5][0]);
            acc[tt - 5][1] = mfma4(zq[ti - 1], zr[0][tj - 1], acc[tt - 5][1]);
            acc[tt - 5][2] = mfma4(zq[ti - 1], zr[1][tj - 1], acc[tt - 5][2]);
            acc[tt - 5][3] = mfma4(zq[ti - 1], zr[2][tj - 1], acc[tt - 5][3]);
          });
        } else {
        static_for<10>([&](auto it) {
          constexpr int tt = decltype(it)::value + 5;          // tiles 5..14 of the 15-tile enumeration = depth x depth
          constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
          acc[tt - 5] = mfma16(zq[ti - 1], zq[tj - 1], acc[tt - 5]);
        });
        }
      });
    } else {
      static_for<16>([&](auto ic_) {
        constexpr int st = decltype(ic_)::value;
        constexpr int sl = st % PF;
        const T a00 = *reinterpret_cast<const T*>(Jc + st * 32);
        const T a01 = *reinterpret_cast<const T*>(Jc + STG1 * 8 + st * 32);
        const T r0 = *reinterpret_cast<const T*>(Sq + st * 32), sz0 = *reinterpret_cast<const T*>(Sq + 64 * 8 + st * 32);
        const T r1 = *reinterpret_cast<const T*>(Sq + STG1 * 8 + st * 32), sz1 = *reinterpret_cast<const T*>(Sq + STG1 * 8 + 64 * 8 + st * 32);
        // pose x depth tiles: (pose row x its depth scale) (x) the raw K~ quad (the joint weight cancels: a s/|s| . |s| k)
        const T p0 = a00 * sz0, p1 = a01 * sz1;
        gv[0] += a00 * r0;
        gv[1] += a01 * r1;
        const T kk[4] = {kq[sl].x, kq[sl].y, kq[sl].z, kq[sl].w};
        if constexpr (M4) {
          // pose x pose: B rotated; pose x depth: the POSE operand rotated (2 values instead of the 4 of the quad), and
          // rot(a s) = rot(a) s -- the depth scale is per pixel, i.e. constant along a row of 16 lanes
          const T a0r[3] = {row_rot4<1>(a00), row_rot4<2>(a00), row_rot4<3>(a00)};
          const T a1r[3] = {row_rot4<1>(a01), row_rot4<2>(a01), row_rot4<3>(a01)};
          acc[0][0] = mfma4(a00, a00, acc[0][0]);
          acc[1][0] = mfma4(a01, a01, acc[1][0]);
#pragma unroll
          for (int d = 1; d < 4; ++d) {
            acc[0][d] = mfma4(a00, a0r[d - 1], acc[0][d]);
            acc[1][d] = mfma4(a01, a1r[d - 1], acc[1][d]);
          }
          const T p0r[4] = {p0, a0r[0] * sz0, a0r[1] * sz0, a0r[2] * sz0};
          const T p1r[4] = {p1, a1r[0] * sz1, a1r[1] * sz1, a1r[2] * sz1};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              acc[2 + e][d] = mfma4(p0r[d], kk[e], acc[2 + e][d]);
              acc[6 + e][d] = mfma4(p1r[d], kk[e], acc[6 + e][d]);
            }
          }
        } else {
        acc[0] = mfma16(a00, a00, acc[0]);
        acc[1] = mfma16(a01, a01, acc[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 + e] = mfma16(p0, kk[e], acc[2 + e]);
          acc[6 + e] = mfma16(p1, kk[e], acc[6 + e]);
        }
        }
        refill(std::integral_constant<int, sl>{}, ic_);
      });
    }
    if constexpr (PIPE) roff_cur = roff_nxt;
  }

  // ---- epilogue: every element has one owner; record layout of ba_blocks_kernel (15 tiles | 5 x 16 gradient | err) ----
  err = wave_sum(err);
  T invz4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) invz4[j] = (4 * c + j < m) ? invz[(long)slot * m + 4 * c + j] : T(0);
  // the 1 / z_m factors of the depth columns (dPwn_dzm = u K~ / z_m, sparse_map.py:184-230), once per accumulator:
  // column k of depth block t at lane-column ci is kcol(t, ci) = 4 ci + t - 1; the f64 MFMA row of (lane, reg) is (lane >> 4) + 4 reg
  // M4: (lane, register d) -> (row, column) of a tile: see row_rot4
  const int m4_i = lane >> 4, m4_blk = (lane >> 2) & 3, m4_j = lane & 3;
  auto invz_at = [&](int k) { return (k < m) ? invz[(long)slot * m + k] : T(0); };
  if (role == 0) {
    static_for<10>([&](auto it) {
      constexpr int tt = decltype(it)::value + 5;
      constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        if constexpr (M4) {
          const int row = 4 * m4_blk + m4_i, col = 4 * ((m4_blk + rg) & 3) + m4_j;
          acc[tt - 5][rg] *= invz_at(kcol(ti, row)) * invz_at(kcol(tj, col));
        } else {
          const int k1 = kcol(ti, mfma_row<T>(lane, rg));
          const T f1 = (k1 < m) ? invz[(long)slot * m + k1] : T(0);
          acc[tt - 5][rg] *= f1 * invz4[tj - 1];
        }
      }
    });
#pragma unroll
    for (int e = 0; e < 4; ++e) gv[e] *= invz4[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const T fz = M4 ? invz_at(kcol(e + 1, 4 * m4_blk + m4_j)) : invz4[e];    // (pose operand rotated: column = 4 blk + j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { acc[2 + e][rg] *= fz; acc[6 + e][rg] *= fz; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gv[e] += __shfl_xor(gv[e], 16, 64);
    gv[e] += __shfl_xor(gv[e], 32, 64);
  }
  T* rec0 = partials + (long)(pg0 * gridDim.x + blockIdx.x) * Cfg::REC;
  T* rec1 = has1 ? partials + (long)(pg1 * gridDim.x + blockIdx.x) * Cfg::REC : nullptr;
  // rotA: the tile's A operand was the rotated one (pose x depth tiles of role 1)
  auto put_tile = [&](T* rec, int tt, const acc_t& a, bool rotA = false) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      if constexpr (M4) {
        const int o = (m4_blk + rg) & 3;
        const int row = rotA ? 4 * o + m4_i : 4 * m4_blk + m4_i, col = rotA ? 4 * m4_blk + m4_j : 4 * o + m4_j;
        rec[tt * 256 + (row >> 2) * 64 + (row & 3) * 16 + col] = a[rg];     // where the 16x16x4 layout keeps (row, col)
      } else {
        rec[tt * 256 + rg * 64 + lane] = a[rg];
      }
    }
  };
  const acc_t zero4 = acc_t{T(0), T(0), T(0), T(0)};
  if (role == 0) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      put_tile(rec0, 5 + t, acc[t]);
      if (rec1) put_tile(rec1, 5 + t, zero4);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rec0[Cfg::NT * 256 + (1 + e) * 16 + lane] = gv[e];
        if (rec1) rec1[Cfg::NT * 256 + (1 + e) * 16 + lane] = T(0);
      }
    }
    if (lane == 0) rec0[Cfg::NT * 256 + Cfg::NB * 16] = err;
  } else {
    put_tile(rec0, 0, acc[0]);
#pragma unroll
    for (int e = 0; e < 4; ++e) put_tile(rec0, 1 + e, acc[2 + e], true);
    if (lane < 16) rec0[Cfg::NT * 256 + lane] = gv[0];
    if (rec1) {
      put_tile(rec1, 0, acc[1]);
#pragma unroll
      for (int e = 0; e < 4; ++e) put_tile(rec1, 1 + e, acc[6 + e], true);
      if (lane < 16) rec1[Cfg::NT * 256 + lane] = gv[1];
      if (lane == 0) rec1[Cfg::NT * 256 + Cfg::NB * 16] = err;
    }
  }
}

// ---------------------------------------- pass 2, float64: WAVE-SPECIALISED two-pair kernel ----------------------------
// The role-split kernel above makes every wave both a producer (warp, taps, Jacobian rows: ~25 dependent global loads per
// pixel) and a consumer (160 v_mfma_f64 per tile): the matrix pipe waits whenever a tile's loads are late, and hiding them
// with a software pipeline costs more registers than two waves per SIMD have (147 spilled VGPRs at 256).  Here the two jobs
// are different WAVES of a 192-thread workgroup that meet once per 64-pixel tile at an LDS-only barrier:
//   producer  (no accumulators): both pairs' warp / taps / rows of tile t+1 -> LDS buffer (t+1) & 1 while the consumers
//             multiply tile t; it runs one tile ahead, its load latencies are nobody's critical path;
//   consumer Z: the 10 depth x depth tiles  (A = B = sqrt(s0^2 + s1^2) K~ quad), depth gradient;
//   consumer T: the 2 pose x pose + 8 pose x depth tiles (A = pose row x depth scale, B = the raw K~ quad), pose gradients.
// A consumer's only global loads are its K~ quads (a register ring PF steps deep, addressed by pixel index alone, so they
// run ahead across tile boundaries) -- between two barriers it issues 160 MFMAs and ~8 other instructions per step.
// Three waves per SIMD (<= 168 registers each): four workgroups per CU, 39.9 KB of LDS each (two stage buffers).  Roles
// rotate with the workgroup index so that every SIMD hosts a mix of producers and consumers whatever the dispatcher's
// wave -> SIMD assignment is.  Records as in the role-split kernel: every element has exactly one owner.
// 12 consecutive doubles at a wave-uniform address through the scalar cache, into SGPRs, NOT hoisted out of loops (volatile):
// the producer's pose constants are re-read where they are used instead of occupying 72 SGPRs across its whole loop.
typedef int sgpr8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sload12(const double* __restrict__ p, double* __restrict__ out) {
  sgpr8_t a, b, c;
  asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dwordx8 %2, %3, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c)
               : "s"(p)
               : "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    out[k] = __hiloint2double(a[2 * k + 1], a[2 * k]);
    out[4 + k] = __hiloint2double(b[2 * k + 1], b[2 * k]);
    out[8 + k] = __hiloint2double(c[2 * k + 1], c[2 * k]);
  }
}

__device__ __forceinline__ void lds_barrier() {      // this wave's LDS traffic is complete; global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef COMO_AB_VARIANTS   // the wave-specialised float64 kernel LOST (0.66-0.78 ms against 0.58): kept for measurement builds only
template <int PFZ, int PFT, int WPS>   // depth of the K~ register ring of consumer Z / T (steps of 4 pixels; divides 16), waves / SIMD
__global__ __launch_bounds__(192, WPS) void ba_blocks_ws_f64_kernel(
    const double* __restrict__ Pwn, const double* __restrict__ vals, const double* __restrict__ dlz,
    const double* __restrict__ Kt, const int* __restrict__ pixidx, const double* __restrict__ invz, long kt_slot_stride,
    BAPairs pr, const double* __restrict__ pair_T, const double* __restrict__ pair_aff, const double* __restrict__ pair_ref,
    const double* __restrict__ img_base, const double* __restrict__ Kmat, int H, int W, int n, int m, int pix_begin,
    int pix_end, int chunk_len, const uint32_t* __restrict__ hists, double* __restrict__ partials,
    double* __restrict__ sigma_out, const int* __restrict__ grp_pairs) {
  using T = double;
  using KeyT = typename KeyOf<T>::type;
  using Cfg = BACfg;
  using acc_t = typename Acc4<T>::type;
  constexpr int ROWS = 16 * JP_STRIDE;               // one pair's 16 pose / affine rows of a tile
  constexpr int PXA = 2 * ROWS;                      // per pixel {sqrt(s0^2 + s1^2), joint whitened residual}
  constexpr int PXB = PXA + 64 * 2;                  // per pixel {depth scale 0, depth scale 1, r~0, r~1}
  constexpr int BUF = PXB + 64 * 4;                  // 2496 doubles
  __shared__ T lds[2 * BUF];                         // two stage buffers: 39,936 B (four workgroups per CU)
  static_assert(sizeof(SelScratch) <= sizeof(T) * BUF, "resolve scratch aliases the first stage buffer");

  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve_part<KeyT, 128, 3>(hists, SelCfg<KeyT>::NPASS, reinterpret_cast<SelScratch*>(lds), prefix, k_rem, nv);
  const T sigma = T(1.4826) * key_value(prefix);
  const T info_sqrt = T(1) / sigma;
  if (sigma_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sigma_out[0] = sigma; sigma_out[1] = (T)nv; }

  const int pg0 = grp_pairs[2 * blockIdx.y], pg1r = grp_pairs[2 * blockIdx.y + 1];
  const bool has1 = pg1r >= 0;
  const int pg1 = has1 ? pg1r : pg0;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef COMO_WS_FORCE_ROLE                                                  // (register-budget analysis builds only)
  const int role = COMO_WS_FORCE_ROLE;
#else
  const int role = (wv + (int)blockIdx.x + (int)blockIdx.y) % 3;         // wave-uniform: 0 producer, 1 consumer Z, 2 consumer T
#endif
  const int q = lane >> 4, c = lane & 15;
  const int slot = pr.ref_slot[pg0];
  const int begin = pix_begin + blockIdx.x * chunk_len;
  const int end = min(pix_end, begin + chunk_len);
  T* rec0 = partials + (long)(pg0 * gridDim.x + blockIdx.x) * Cfg::REC;
  T* rec1 = has1 ? partials + (long)(pg1 * gridDim.x + blockIdx.x) * Cfg::REC : nullptr;
  const int* PiS = pixidx ? pixidx + (long)slot * n : nullptr;
  const uint32_t row_bytes = (uint32_t)(m * sizeof(T));

  if (role == 0) {
    // =========================================== producer ===========================================
    // The pose constants of BOTH pairs ([R | t] of two targets and of the reference: 36 doubles = 72 SGPRs) do not fit the
    // scalar register file next to everything else; kept live across the loop the compiler parks them in VGPRs and spills.
    // They are (re)read through the scalar cache where they are used instead (sload12).
    constexpr int G = 2;
    const int pg[G] = {pg0, pg1};
    T scale[G], bias[G];
    const T* img[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      scale[g] = pair_aff[2 * pg[g]];
      bias[g] = pair_aff[2 * pg[g] + 1];
      img[g] = img_base + pr.tgt_img[pg[g]] + (long)pair_chan(pr, pg[g]) * H * W;
    }
    const int ch = pair_chan(pr, pg0);              // both pairs of a group share reference slot AND channel
    const T fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
    const T ax = pr.anorm_f32 ? (T)(1.0f / (float)W) : T(1) / T(W), ay = pr.anorm_f32 ? (T)(1.0f / (float)H) : T(1) / T(H);
    const long HW = (long)pr.C * H * W;
    const T* PwS = Pwn + (long)slot * 3 * n;
    const T* DlS = dlz + (long)slot * 6 * n;
    const T* VaS = vals + (long)slot * n * pr.C + ch;
    T err[G] = {T(0), T(0)};
    T pw0, pw1, pw2;
    {
      const int ic = min(begin + lane, end - 1);
      pw0 = PwS[ic]; pw1 = PwS[(long)n + ic]; pw2 = PwS[2 * (long)n + ic];
    }
    int buf = 0;
    for (int tile = begin; tile < end; tile += 64, buf ^= 1) {
      const int i = tile + lane;
      const bool inr = i < end;
      const int ic = inr ? i : (end - 1);
      // ---- S1: warp the tile's points into both targets, issue every load of the tile
      const T Px = pw0, Py = pw1, Pz = pw2;
      T tv[G][12], wX[G], wY[G], wZ[G], w00[G], w01[G], w10[G], w11[G];
      bool wok[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        T M[12];
        sload12(pair_T + 12 * (long)pg[g], M);
        Warp<T> w = warp_point(M, fx, fy, cx, cy, Px, Py, Pz, H, W);
        Taps<T> tp = make_taps(grid_position(w.u, W, ax), grid_position(w.v, H, ay), H, W);
        wX[g] = w.X; wY[g] = w.Y; wZ[g] = w.Z; wok[g] = inr && w.ok && (g == 0 || has1);
        w00[g] = tp.w00; w01[g] = tp.w01; w10[g] = tp.w10; w11[g] = tp.w11;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const T* P = img[g] + pl * HW;
          tv[g][4 * pl + 0] = P[tp.i00]; tv[g][4 * pl + 1] = P[tp.i01]; tv[g][4 * pl + 2] = P[tp.i10]; tv[g][4 * pl + 3] = P[tp.i11];
        }
      }
      T Dv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Dv[k] = DlS[(long)k * n + ic];
      const T valv = VaS[(long)ic * pr.C];
      {   // next tile's points (clamped past the end)
        const int ic2 = min(tile + 64 + lane, end - 1);
        pw0 = PwS[ic2]; pw1 = PwS[(long)n + ic2]; pw2 = PwS[2 * (long)n + ic2];
      }
      // ---- S2: rows of both pairs -> stage buffer
      T* B = lds + buf * BUF;
      T rsv[G], szv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const T It = w00[g] * tv[g][0] + w01[g] * tv[g][1] + w10[g] * tv[g][2] + w11[g] * tv[g][3];
        const T gx = w00[g] * tv[g][4] + w01[g] * tv[g][5] + w10[g] * tv[g][6] + w11[g] * tv[g][7];
        const T gy = w00[g] * tv[g][8] + w01[g] * tv[g][9] + w10[g] * tv[g][10] + w11[g] * tv[g][11];
        const T Iref_s = scale[g] * valv;
        const T r = It - Iref_s + bias[g];
        const bool ok = wok[g];
        const T wr = r * info_sqrt;
        const T awr = fabs(wr);
        // sqrt of the Huber weight (robust_loss.py:9-16): 1 inside the band, sqrt(1.345 / |x|) outside
        const T ws = (awr < T(1.345)) ? T(1) : T(1.1597413504743201) * fast_rsq(awr);
        const T s = ok ? info_sqrt * ws : T(0);
        err[g] += ok ? (ws * wr) * (ws * wr) : T(0);
        const T iz = ok ? fast_rcp(wZ[g]) : T(0);
        const T a0 = gx * fx * iz, a1 = gy * fy * iz;
        const T a2 = -(a0 * wX[g] + a1 * wY[g]) * iz;
        T b0, b1, b2;
        {
          T Mg[12];
          sload12(pair_T + 12 * (long)pg[g], Mg);
          b0 = a0 * Mg[0] + a1 * Mg[4] + a2 * Mg[8];
          b1 = a0 * Mg[1] + a1 * Mg[5] + a2 * Mg[9];
          b2 = a0 * Mg[2] + a1 * Mg[6] + a2 * Mg[10];
        }
        T jr[6], bu;
        {
          T Rf[12];
          sload12(pair_ref + 12 * (long)pg0, Rf);
          bu = ref_pose_geom(Rf, Px, Py, Pz, b0, b1, b2, jr);
        }
        const T sbu = s * bu;
        T* J = B + g * ROWS;
#pragma unroll
        for (int k = 0; k < 6; ++k) J[k * JP_STRIDE + lane] = s * jr[k] + sbu * Dv[k];
        J[6 * JP_STRIDE + lane] = s * Iref_s;
        J[7 * JP_STRIDE + lane] = -s;
        const T Xc = ok ? wX[g] : T(0), Yc = ok ? wY[g] : T(0), Zc = ok ? wZ[g] : T(0);
        J[8 * JP_STRIDE + lane] = s * (a1 * Zc - a2 * Yc);
        J[9 * JP_STRIDE + lane] = s * (a2 * Xc - a0 * Zc);
        J[10 * JP_STRIDE + lane] = s * (a0 * Yc - a1 * Xc);
        J[11 * JP_STRIDE + lane] = -s * a0;
        J[12 * JP_STRIDE + lane] = -s * a1;
        J[13 * JP_STRIDE + lane] = -s * a2;
        J[14 * JP_STRIDE + lane] = -s * Iref_s;
        J[15 * JP_STRIDE + lane] = s;
        rsv[g] = s * r;
        szv[g] = sbu;
      }
      // per-PIXEL factors of the joint depth row (both rows of a reference pixel share the K~ row scaled by their own depth
      // scale): weight sqrt(s0^2 + s1^2) and its whitened residual
      const T cs = szv[0] * szv[0] + szv[1] * szv[1];
      const T rs = cs > T(0) ? fast_rsq(cs) : T(0);
      B[PXA + 2 * lane] = cs * rs;
      B[PXA + 2 * lane + 1] = (szv[0] * rsv[0] + szv[1] * rsv[1]) * rs;
      B[PXB + 4 * lane] = szv[0];
      B[PXB + 4 * lane + 1] = szv[1];
      B[PXB + 4 * lane + 2] = rsv[0];
      B[PXB + 4 * lane + 3] = rsv[1];
      lds_barrier();                               // tile staged; the consumers are done with the other buffer
    }
    err[0] = wave_sum(err[0]);
    err[1] = wave_sum(err[1]);
    if (lane == 0) {
      rec0[Cfg::NT * 256 + Cfg::NB * 16] = err[0];
      if (rec1) rec1[Cfg::NT * 256 + Cfg::NB * 16] = err[1];
    }
    return;
  }

  // ============================================== consumers ==============================================
  // lanes whose quad lies beyond m read quad 0 instead (finite values); their columns are nulled by invz = 0 in the epilogue
  const char* KtB = reinterpret_cast<const char*>(Kt + (long)slot * kt_slot_stride);
  const uint32_t cbyte = (uint32_t)(((4 * c < m) ? 4 * c : 0) * sizeof(T));
  acc_t acc[10];      // Z: the 10 depth x depth tiles; T: {TT_0, TT_1, Tz_0[4], Tz_1[4]}
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = acc_t{T(0), T(0), T(0), T(0)};
  T gv[4] = {T(0), T(0), T(0), T(0)};    // Z: depth gradient of this lane's quad; T: gv[0], gv[1] = pose gradients
  static_assert(16 % PFZ == 0 && 16 % PFT == 0, "the ring slot of a step must be the same in every tile");
  constexpr int PFM = PFZ > PFT ? PFZ : PFT;
  V4<T> kq[PFM];
  // K~ row index of this lane's pixel of a tile, fetched TWO tiles ahead: by the time it is multiplied into a byte offset
  // the load has long landed (its wait must not drain the K~ ring that was issued after it)
  auto row_idx = [&](int tile) -> uint32_t {
    const int ic = min(tile + lane, end - 1);
    return (uint32_t)(PiS ? PiS[ic] : ic);
  };
  uint32_t roff_cur = row_idx(begin) * row_bytes, roff_nxt = 0;
  uint32_t pix_n1 = row_idx(begin + 64), pix_n2 = 0;
  static_for<PFM>([&](auto ic_) {
    constexpr int st = decltype(ic_)::value;
    if (st < (role == 1 ? PFZ : PFT))
      kq[st] = load4_off(KtB, (uint32_t)__builtin_amdgcn_ds_bpermute(4 * q + 16 * st, (int)roff_cur) + cbyte);
  });
  // Every per-step address is ONE per-tile base register + a compile-time offset (the instruction's offset field): written
  // out with explicit byte arithmetic -- indexed by the step, the compiler hoisted 16 steps x 3 loop-invariant address
  // vectors out of the tile loop and spilled them (scratch reloads + vmcnt(0) drains of the K~ ring inside the loop).
  const int qsel = q * 4;                                      // ds_bpermute byte address of lane q (+ 16 per step)
  // refill ring slot sl after step st: the quad of step st + PF of this tile, or of step st + PF - 16 of the next one
  auto refill = [&](auto pf_, auto sl_, auto st_) {
    constexpr int sl = decltype(sl_)::value, nst = decltype(st_)::value + decltype(pf_)::value;
    constexpr bool same = nst < 16;
    const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(qsel + 16 * (same ? nst : nst - 16), (int)(same ? roff_cur : roff_nxt));
    kq[sl] = load4_off(KtB, off + cbyte);
  };
  int buf = 0;
  if (role == 1) {
    for (int tile = begin; tile < end; tile += 64, buf ^= 1) {
      pix_n2 = row_idx(tile + 128);
      roff_nxt = pix_n1 * row_bytes;
      lds_barrier();                               // the producer has staged this tile
      const char* Bq = reinterpret_cast<const char*>(lds + buf * BUF + PXA) + q * 16;      // {sqrt(s0^2 + s1^2), joint r~} of pixel q
      static_for<16>([&](auto ic_) {
        constexpr int st = decltype(ic_)::value;
        constexpr int sl = st % PFZ;
        const double2 pa = *reinterpret_cast<const double2*>(Bq + st * 64);
        // depth columns WITHOUT the 1 / z_m factor: it is constant over pixels and is applied to the accumulators once
        const T zq[4] = {pa.x * kq[sl].x, pa.x * kq[sl].y, pa.x * kq[sl].z, pa.x * kq[sl].w};
        refill(std::integral_constant<int, PFZ>{}, std::integral_constant<int, sl>{}, ic_);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] += zq[e] * pa.y;
        static_for<10>([&](auto it) {
          constexpr int tt = decltype(it)::value + 5;          // tiles 5..14 of the 15-tile enumeration = depth x depth
          constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
          acc[tt - 5] = mfma16(zq[ti - 1], zq[tj - 1], acc[tt - 5]);
        });
      });
      roff_cur = roff_nxt;
      pix_n1 = pix_n2;
    }
  } else {
    for (int tile = begin; tile < end; tile += 64, buf ^= 1) {
      pix_n2 = row_idx(tile + 128);
      roff_nxt = pix_n1 * row_bytes;
      lds_barrier();
      const char* Bc = reinterpret_cast<const char*>(lds + buf * BUF) + (c * JP_STRIDE + q) * 8;   // pose row c, pixel q
      const char* Bq = reinterpret_cast<const char*>(lds + buf * BUF + PXB) + q * 32;              // {sz0, sz1, r~0, r~1} of pixel q
      static_for<16>([&](auto ic_) {
        constexpr int st = decltype(ic_)::value;
        constexpr int sl = st % PFT;
        const T a00 = *reinterpret_cast<const T*>(Bc + st * 32);
        const T a01 = *reinterpret_cast<const T*>(Bc + ROWS * 8 + st * 32);
        const double2 sz = *reinterpret_cast<const double2*>(Bq + st * 128);          // depth scales of the two rows
        const double2 rr = *reinterpret_cast<const double2*>(Bq + st * 128 + 16);     // their whitened residuals
        const T p0 = a00 * sz.x, p1 = a01 * sz.y;              // pose row x depth scale; the B operand is the raw K~ quad
        gv[0] += a00 * rr.x;
        gv[1] += a01 * rr.y;
        acc[0] = mfma16(a00, a00, acc[0]);
        acc[1] = mfma16(a01, a01, acc[1]);
        const T kk[4] = {kq[sl].x, kq[sl].y, kq[sl].z, kq[sl].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 + e] = mfma16(p0, kk[e], acc[2 + e]);
          acc[6 + e] = mfma16(p1, kk[e], acc[6 + e]);
        }
        refill(std::integral_constant<int, PFT>{}, std::integral_constant<int, sl>{}, ic_);
      });
      roff_cur = roff_nxt;
      pix_n1 = pix_n2;
    }
  }

  // ---- epilogue: record layout of ba_blocks_kernel (15 tiles | 5 x 16 gradient | err) ----
  T invz4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) invz4[j] = (4 * c + j < m) ? invz[(long)slot * m + 4 * c + j] : T(0);
  auto put_tile = [&](T* rec, int tt, const acc_t& a) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) rec[tt * 256 + rg * 64 + lane] = a[rg];
  };
  const acc_t zero4 = acc_t{T(0), T(0), T(0), T(0)};
  if (role == 1) {
    // the 1 / z_m factors of the depth columns (dPwn_dzm = u K~ / z_m, sparse_map.py:184-230), once per accumulator: column k
    // of depth block t at lane-column ci is kcol(t, ci) = 4 ci + t - 1; the f64 MFMA row of (lane, reg) is (lane >> 4) + 4 reg
    static_for<10>([&](auto it) {
      constexpr int tt = decltype(it)::value + 5;
      constexpr int ti = tile_row(tt), tj = tt - tile_first(ti) + ti;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int k1 = kcol(ti, mfma_row<T>(lane, rg));
        const T f1 = (k1 < m) ? invz[(long)slot * m + k1] : T(0);
        acc[tt - 5][rg] *= f1 * invz4[tj - 1];
      }
    });
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gv[e] *= invz4[e];
      gv[e] += __shfl_xor(gv[e], 16, 64);
      gv[e] += __shfl_xor(gv[e], 32, 64);
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      put_tile(rec0, 5 + t, acc[t]);
      if (rec1) put_tile(rec1, 5 + t, zero4);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rec0[Cfg::NT * 256 + (1 + e) * 16 + lane] = gv[e];
        if (rec1) rec1[Cfg::NT * 256 + (1 + e) * 16 + lane] = T(0);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { acc[2 + e][rg] *= invz4[e]; acc[6 + e][rg] *= invz4[e]; }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      gv[e] += __shfl_xor(gv[e], 16, 64);
      gv[e] += __shfl_xor(gv[e], 32, 64);
    }
    put_tile(rec0, 0, acc[0]);
#pragma unroll
    for (int e = 0; e < 4; ++e) put_tile(rec0, 1 + e, acc[2 + e]);
    if (lane < 16) rec0[Cfg::NT * 256 + lane] = gv[0];
    if (rec1) {
      put_tile(rec1, 0, acc[1]);
#pragma unroll
      for (int e = 0; e < 4; ++e) put_tile(rec1, 1 + e, acc[6 + e]);
      if (lane < 16) rec1[Cfg::NT * 256 + lane] = gv[1];
    }
  }
}
#endif  // COMO_AB_VARIANTS

// ---------------------------------------- stage 2 ------------------------------------------------
// One thread per record element: fixed-order fp64 sum over the pair's wave partials, then the
// landmark expansion (photo.py:169-182) and accumulation into H / g (photo.py:184-231).
// TH = float / double: floating-point atomics into the caller's H, both triangles (the reference-signature contract:
// batch_photo_cost accumulates into whatever H already holds).  TH = long long: ORDER-INDEPENDENT mode -- exact integer
// atomics into the fixed-point system buffer, lower triangle only (como_sys_finalize mirrors it), see common.cuh fix_add.
// MODE 0: reduce + expand; 1: reduce only -> blocks_fix; 2: expand from blocks_fix (after the all-reduce of the shards).
template <typename T, typename TH, int MODE>
__global__ __launch_bounds__(256) void ba_reduce_assemble_kernel(
    const T* __restrict__ partials, int nrec_per_pair, BAPairs pr, const long* __restrict__ pose_ref_inds,
    const long* __restrict__ pose_tgt_inds, const long* __restrict__ landmark_inds, const T* __restrict__ dzdP,
    int m, TH* __restrict__ Hm, long D, TH* __restrict__ gv, double* __restrict__ err_out,
    double* __restrict__ pair_blocks, long fix_plane, long long* __restrict__ blocks_fix) {
  using Cfg = BACfg;
  constexpr bool FIX = std::is_same<TH, long long>::value;
  constexpr int NE = Cfg::NT * 256 + Cfg::NB * 16 + 1;
  // Non-finite sums of the sharded (MODE 1 -> all-reduce -> MODE 2) path travel as COUNTS, never as a magic value inside the
  // summed number (an additive sentinel wraps: 4 ranks x 2^62 = 0): workgroup k of a pair stores its "saw a non-finite sum"
  // flag into plane (k & 1) of padding slot NE + (k >> 1) of the pair's record -- plain stores, every padding word of the
  // record is written (the all-reduced buffers are bit-identical from run to run); MODE 2 poisons when any count is nonzero.
  constexpr int NFLAG_BLOCKS = (Cfg::REC + 255) / 256;
  static_assert((NFLAG_BLOCKS + 1) / 2 <= Cfg::REC - NE, "one flag word per workgroup fits the record padding");
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int p = blockIdx.y;
  if constexpr (MODE == 0) {
    if (e >= NE) return;
  }
  double s = 0;
  if constexpr (MODE != 2) {
    if (e < NE) {
      const T* base = partials + (long)p * nrec_per_pair * Cfg::REC + e;
      // fixed summation order, but 8 loads in flight (one dependent 16 KB-strided load per step was 45 us of this kernel)
      int w = 0;
      for (; w + 8 <= nrec_per_pair; w += 8) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = base[(long)(w + q) * Cfg::REC];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += (double)v[q];
      }
      for (; w < nrec_per_pair; ++w) s += (double)base[(long)w * Cfg::REC];
    }
  }
  if constexpr (MODE == 1) {                       // this rank's share of the pair sums, in fixed point (exact all-reduce)
    long long* rec = blocks_fix + 2 * (long)p * Cfg::REC;
    const bool bad = (e < NE) && !(fabs(s) < 4.0e18);
    if (e < NE) {
      long long hi = 0;
      unsigned long long lo = 0;
      if (!bad) fix_split(s, hi, lo);
      rec[2 * e] = hi;
      rec[2 * e + 1] = (long long)lo;
    } else if (e < Cfg::REC && e >= NE + (NFLAG_BLOCKS + 1) / 2) {
      rec[2 * e] = 0;                              // padding beyond the flag words
      rec[2 * e + 1] = 0;
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) rec[2 * (NE + ((int)blockIdx.x >> 1)) + ((int)blockIdx.x & 1)] = any_bad ? 1 : 0;
    if ((NFLAG_BLOCKS & 1) && blockIdx.x == 0 && threadIdx.x == 1) rec[2 * (NE + (NFLAG_BLOCKS >> 1)) + 1] = 0;   // the unpaired flag word
    return;
  }
  if constexpr (MODE == 2) {
    if (e >= NE) return;
    const long long* rec = blocks_fix + 2 * (long)p * Cfg::REC;
    long long nbad = 0;
#pragma unroll
    for (int k = 0; k < 2 * ((NFLAG_BLOCKS + 1) / 2); ++k) nbad |= rec[2 * NE + k];
    s = nbad ? __builtin_nan("") : fix_value(rec[2 * e], (unsigned long long)rec[2 * e + 1]);
  }
  if (pair_blocks) pair_blocks[(long)p * Cfg::REC + e] = s;
  const int slot = pr.ref_slot[p];
  const long* pri = pose_ref_inds + 8 * (long)p;
  const long* pti = pose_tgt_inds + 8 * (long)p;
  const long* lmi = landmark_inds + 3 * (long)m * p;
  const T* dz = dzdP + 3 * (long)slot;
  long long* poison = nullptr;
  if constexpr (FIX) poison = (long long*)Hm + D * D + D + FIX_POISON;
  // one symmetric entry pair (ia, ib) / (ib, ia) of H
  auto add_sym = [&](long ia, long ib, double v, bool both) {
    if constexpr (FIX) {
      const long r = ia > ib ? ia : ib, c = ia > ib ? ib : ia;
      fix_add((long long*)Hm, fix_plane, r * D + c, v, poison);
    } else {
      atomicAdd(&Hm[ia * D + ib], (TH)v);
      if (both) atomicAdd(&Hm[ib * D + ia], (TH)v);
    }
  };
  // column id -> (kind, index): block 0 entry i: pose index (i<8 ref, else target); block t>=1: depth kcol(t,i)
  if (e < Cfg::NT * 256) {
    const int tt = e >> 8, rg = (e >> 6) & 3, lane = e & 63;
    int ti = 0, tj = 0, cnt = 0;
    for (int a = 0; a < Cfg::NB; ++a)
      for (int b = a; b < Cfg::NB; ++b) { if (cnt == tt) { ti = a; tj = b; } ++cnt; }
    const int ri = mfma_row<T>(lane, rg), ci = lane & 15;
    if (ti == 0 && tj == 0) {
      // the full 16 x 16 tile is present: (ri, ci) and (ci, ri) carry the same bits -- fixed-point mode keeps one of them
      const long ia = ri < 8 ? pri[ri] : pti[ri - 8];
      const long ib = ci < 8 ? pri[ci] : pti[ci - 8];
      if (!FIX || ia >= ib) add_sym(ia, ib, s, false);
    } else if (ti == 0) {
      const int k = kcol(tj, ci);
      if (k < m) {
        const long ia = ri < 8 ? pri[ri] : pti[ri - 8];
        for (int d = 0; d < 3; ++d) add_sym(ia, lmi[3 * k + d], s * (double)dz[d], true);
      }
    } else {
      const int k1 = kcol(ti, ri), k2 = kcol(tj, ci);
      if (k1 < m && k2 < m) {
        for (int d1 = 0; d1 < 3; ++d1)
          for (int d2 = 0; d2 < 3; ++d2) {
            const double v = (double)dz[d1] * s * (double)dz[d2];
            const long i1 = lmi[3 * k1 + d1], i2 = lmi[3 * k2 + d2];
            // diagonal tiles hold both (k1, k2) and (k2, k1): each thread adds its own ordered entry; off-diagonal tiles
            // hold (k1, k2) once: both triangles (float modes) / the lower one (fixed-point mode)
            if (ti != tj) add_sym(i1, i2, v, true);
            else if (!FIX || i1 >= i2) add_sym(i1, i2, v, false);
          }
      }
    }
  } else if (e < Cfg::NT * 256 + Cfg::NB * 16) {
    const int t = (e - Cfg::NT * 256) >> 4, ci = e & 15;
    const double gval = -s;                                  // get_gradient: g = -sum J r (linear_system.py:24-26)
    auto add_g = [&](long ia, double v) {
      if constexpr (FIX) fix_add((long long*)Hm, fix_plane, D * D + ia, v, poison);
      else atomicAdd(&gv[ia], (TH)v);
    };
    if (t == 0) {
      add_g(ci < 8 ? pri[ci] : pti[ci - 8], gval);
    } else {
      const int k = kcol(t, ci);
      if (k < m)
        for (int d = 0; d < 3; ++d) add_g(lmi[3 * k + d], gval * (double)dz[d]);
    }
  } else {
    if constexpr (FIX) fix_add((long long*)Hm, fix_plane, D * D + D, s, poison);
    else atomicAdd(err_out, s);
  }
}

// The same assembly with the pairs GROUPED BY REFERENCE SLOT (fixed-point mode, reduce + expand in one launch): the pairs of a
// group write the same depth x depth blocks (192 x 192 after the expansion), the same reference-pose rows and gradient parts --
// everything except what involves their TARGET frames.  Thread e forms every pair's contributions exactly as the per-pair kernel
// does (the pair's wave partials summed in the same fixed order, the same dz products), splits each into its fixed-point parts and
// adds those INTEGERS over the group's pairs in registers; one pair of integer atomics per system entry and group follows instead
// of one per pair: a window of the sequential loop holds 3-6 pairs per reference keyframe (forward, backward, one-way frames),
// i.e. a quarter of the atomics -- they were 120-190 us of a full window's iteration, most of it contention of a group's pairs on
// the same entries.  Integer addition is associative: the system buffer ends with the SAME bits as after the per-pair kernel
// (and as on every rank of the sharded form, which keeps the per-pair kernel).  Target-frame entries are added per pair.
struct FixAcc {
  long long hi = 0;
  unsigned long long lo = 0;
  unsigned bad = 0;
  __device__ __forceinline__ void add(double v) {
    if (!(fabs(v) < 4.0e18)) { ++bad; return; }
    long long h;
    unsigned long long l;
    fix_split(v, h, l);
    hi += h;
    lo += l;
  }
  __device__ __forceinline__ void emit(long long* __restrict__ fix, long plane, long idx, long long* __restrict__ poison) const {
    if (bad) atomicAdd((unsigned long long*)poison, (unsigned long long)bad);
    if (hi) atomicAdd((unsigned long long*)&fix[idx], (unsigned long long)hi);
    if (lo) atomicAdd((unsigned long long*)&fix[plane + idx], lo);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void ba_reduce_assemble_grouped_kernel(
    const T* __restrict__ partials, int nrec_per_pair, const int* __restrict__ grp_start, const int* __restrict__ grp_list,
    BAPairs pr, const long* __restrict__ pose_ref_inds, const long* __restrict__ pose_tgt_inds,
    const long* __restrict__ landmark_inds, const T* __restrict__ dzdP, int m, long long* __restrict__ Hm, long D, long fix_plane) {
  using Cfg = BACfg;
  constexpr int NE = Cfg::NT * 256 + Cfg::NB * 16 + 1;
  // 64 elements per workgroup, the group's pairs dealt out over its four waves (the sum over a pair's records is a chain of
  // dependent round trips to the memory side: 4.4 pairs x 37 records one after the other were 91 us in the sequential loop's
  // windows); the waves' integer accumulators meet in LDS -- integer addition: the same bits as one thread walking all pairs
  const int ql = threadIdx.x >> 6;
  const int e_raw = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool act = e_raw < NE;
  const int e = act ? e_raw : NE - 1;
  const int g0 = grp_start[blockIdx.y], g1 = grp_start[blockIdx.y + 1];
  if (g1 <= g0) return;
  __shared__ long long sh_hi[3][9][64];
  __shared__ unsigned long long sh_lo[3][9][64];
  __shared__ unsigned sh_bad[3][9][64];
  long long* poison = Hm + D * D + D + FIX_POISON;
  auto pair_sum = [&](int p) {
    const T* base = partials + (long)p * nrec_per_pair * Cfg::REC + e;
    double s = 0;
    int w = 0;
    for (; w + 8 <= nrec_per_pair; w += 8) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = base[(long)(w + q) * Cfg::REC];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
    for (; w < nrec_per_pair; ++w) s += (double)base[(long)w * Cfg::REC];
    return s;
  };
  auto low = [&](long ia, long ib) { return (ia > ib ? ia : ib) * D + (ia > ib ? ib : ia); };     // entry of the lower triangle
  const int p0 = grp_list[g0];
  const int slot = pr.ref_slot[p0];
  const long* pri = pose_ref_inds + 8 * (long)p0;
  const long* lmi = landmark_inds + 3 * (long)m * p0;
  const T* dz = dzdP + 3 * (long)slot;
  // what this element is, and whether it involves the target frame (then it is added per pair)
  bool per_pair = false;
  int tt = 0, rg = 0, lane = 0, ti = 0, tj = 0, ri = 0, ci = 0, gt = 0;
  const bool is_tile = e < Cfg::NT * 256, is_grad = !is_tile && e < Cfg::NT * 256 + Cfg::NB * 16;
  if (is_tile) {
    tt = e >> 8; rg = (e >> 6) & 3; lane = e & 63;
    int cnt = 0;
    for (int a = 0; a < Cfg::NB; ++a)
      for (int b = a; b < Cfg::NB; ++b) { if (cnt == tt) { ti = a; tj = b; } ++cnt; }
    ri = mfma_row<T>(lane, rg); ci = lane & 15;
    per_pair = (ti == 0) && (ri >= 8 || (tj == 0 && ci >= 8));
  } else if (is_grad) {
    gt = (e - Cfg::NT * 256) >> 4; ci = e & 15;
    per_pair = (gt == 0) && ci >= 8;
  }
  FixAcc acc[9];
  for (int q = g0 + ql; q < g1 && act; q += 4) {
    const int p = grp_list[q];
    const double s = pair_sum(p);
    const long* pti = pose_tgt_inds + 8 * (long)p;
    if (is_tile) {
      if (ti == 0 && tj == 0) {
        const long ia = ri < 8 ? pri[ri] : pti[ri - 8];
        const long ib = ci < 8 ? pri[ci] : pti[ci - 8];
        if (ia >= ib) {
          if (per_pair) fix_add(Hm, fix_plane, low(ia, ib), s, poison);
          else acc[0].add(s);
        }
      } else if (ti == 0) {
        const int k = kcol(tj, ci);
        if (k < m) {
          const long ia = ri < 8 ? pri[ri] : pti[ri - 8];
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const double v = s * (double)dz[d];
            if (per_pair) fix_add(Hm, fix_plane, low(ia, lmi[3 * k + d]), v, poison);
            else acc[d].add(v);
          }
        }
      } else {
        const int k1 = kcol(ti, ri), k2 = kcol(tj, ci);
        if (k1 < m && k2 < m) {
#pragma unroll
          for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) acc[3 * d1 + d2].add((double)dz[d1] * s * (double)dz[d2]);
        }
      }
    } else if (is_grad) {
      const double gval = -s;
      if (gt == 0) {
        if (per_pair) fix_add(Hm, fix_plane, D * D + pti[ci - 8], gval, poison);
        else acc[0].add(gval);
      } else {
        const int k = kcol(gt, ci);
        if (k < m) {
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[d].add(gval * (double)dz[d]);
        }
      }
    } else {
      acc[0].add(s);
    }
  }
  if (g1 - g0 > 1) {                                   // (uniform over the workgroup)
    const int l = threadIdx.x & 63;
    if (ql > 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) { sh_hi[ql - 1][k][l] = acc[k].hi; sh_lo[ql - 1][k][l] = acc[k].lo; sh_bad[ql - 1][k][l] = acc[k].bad; }
    }
    __syncthreads();
    if (ql == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int k = 0; k < 9; ++k) { acc[k].hi += sh_hi[w][k][l]; acc[k].lo += sh_lo[w][k][l]; acc[k].bad += sh_bad[w][k][l]; }
    }
  }
  if (per_pair || ql != 0 || !act) return;
  // one pair of integer atomics per entry for the whole group
  if (is_tile) {
    if (ti == 0 && tj == 0) {
      const long ia = pri[ri], ib = pri[ci];
      if (ia >= ib) acc[0].emit(Hm, fix_plane, low(ia, ib), poison);
    } else if (ti == 0) {
      const int k = kcol(tj, ci);
      if (k < m)
        for (int d = 0; d < 3; ++d) acc[d].emit(Hm, fix_plane, low(pri[ri], lmi[3 * k + d]), poison);
    } else {
      const int k1 = kcol(ti, ri), k2 = kcol(tj, ci);
      if (k1 < m && k2 < m)
        for (int d1 = 0; d1 < 3; ++d1)
          for (int d2 = 0; d2 < 3; ++d2) {
            const long i1 = lmi[3 * k1 + d1], i2 = lmi[3 * k2 + d2];
            if (ti != tj || i1 >= i2) acc[3 * d1 + d2].emit(Hm, fix_plane, low(i1, i2), poison);
          }
    }
  } else if (is_grad) {
    if (gt == 0) acc[0].emit(Hm, fix_plane, D * D + pri[ci], poison);
    else {
      const int k = kcol(gt, ci);
      if (k < m)
        for (int d = 0; d < 3; ++d) acc[d].emit(Hm, fix_plane, D * D + lmi[3 * k + d], poison);
    }
  } else {
    acc[0].emit(Hm, fix_plane, D * D + D, poison);
  }
}

// fixed-point system buffer -> float64 H (both triangles, exactly symmetric), g, errors
__global__ __launch_bounds__(256) void sys_finalize_kernel(const long long* __restrict__ fix, long plane, long D,
                                                           double* __restrict__ H, double* __restrict__ g,
                                                           double* __restrict__ err8) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const bool poisoned = fix[D * D + D + FIX_POISON] != 0;
  if (t < D * D) {
    const long i = t / D, j = t - i * D;
    if (j <= i) {
      double v = fix_value(fix[t], (unsigned long long)fix[plane + t]);
      if (poisoned && t == 0) v = __builtin_nan("");
      H[t] = v;
      if (j < i) H[j * D + i] = v;
    }
  } else if (t < D * D + D) {
    g[t - D * D] = fix_value(fix[t], (unsigned long long)fix[plane + t]);
  } else if (t < D * D + D + FIX_ERR_SLOTS && err8) {
    err8[t - D * D - D] = fix_value(fix[t], (unsigned long long)fix[plane + t]);
  }
}

// sys_finalize + the packing step of the Cholesky solve in ONE launch: besides H / g / err8 it writes the solver's working copy
// W (Dp x Dp, Dp = the dimension padded to whole 32-wide block columns incl. the appended right-hand-side row -- the layout of
// chol_pack_kernel, csrc/chol.hip) and resets the factorisation status.  One thread per element of W.
__global__ __launch_bounds__(256) void sys_finalize_pack_kernel(const long long* __restrict__ fix, long plane, long D,
                                                                double* __restrict__ H, double* __restrict__ g,
                                                                double* __restrict__ err8, double* __restrict__ W, long Dp,
                                                                int* __restrict__ info) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) *info = 0;
  cholp_reset_sync(W, D, idx);
  if (idx < FIX_ERR_SLOTS && err8) err8[idx] = fix_value(fix[D * D + D + idx], (unsigned long long)fix[plane + D * D + D + idx]);
  if (idx >= Dp * Dp) return;
  const long i = idx / Dp, j = idx - i * Dp;
  const bool poisoned = fix[D * D + D + FIX_POISON] != 0;
  double v = 0.0;
  if (i < D && j < D) {
    if (j <= i) {
      const long t = i * D + j;
      v = fix_value(fix[t], (unsigned long long)fix[plane + t]);
      if (poisoned && t == 0) v = __builtin_nan("");
      H[t] = v;
      if (j < i) H[j * D + i] = v;
    }
  } else if (i == D && j < D) {
    v = fix_value(fix[D * D + j], (unsigned long long)fix[plane + D * D + j]);
    g[j] = v;
  } else if (i == D && j == D) {
    v = 1e300;                                     // pivot of the appended row: irrelevant, just positive
  } else if (i == j) {
    v = 1.0;                                       // identity pad
  }
  W[idx] = v;
}

// ---------------------------------------- host side ----------------------------------------------
template <typename T>
int ba_linearize(const como_ba_args* A, hipStream_t s) {
  using KeyT = typename KeyOf<T>::type;
  if (!A || A->b <= 0 || A->n <= 0 || A->m <= 0 || A->m > 64 || (A->m & 3) || A->H < 3 || A->W < 3) return COMO_ERR_ARG;
  if (!A->Pwn || !A->vals || !A->dPwn_dTwc || !A->zjac || !A->poses_all || !A->aff_all || !A->img_base || !A->K ||
      !A->ref_slot || !A->ref_aff || !A->tgt_aff || !A->tgt_pose || !A->tgt_img || !A->ws_r || !A->ws_valid ||
      !A->ws_hists || !A->ws_pair || !A->ws_partials)
    return COMO_ERR_ARG;
  if (A->zmode < 0 || A->zmode > 2) return COMO_ERR_ARG;
  if (A->zmode == 1 && (!A->uvec || !A->invz)) return COMO_ERR_ARG;
  if (A->zmode == 2 && (!A->invz || !A->ref_pose)) return COMO_ERR_ARG;
  if (A->channels < 0 || (A->channels > 1 && !A->pair_chan)) return COMO_ERR_ARG;
  BAPairs pr{A->ref_slot, A->ref_aff, A->tgt_aff, A->tgt_pose, A->tgt_img, A->anorm_f32, A->channels > 1 ? A->pair_chan : nullptr,
             A->channels > 1 ? A->channels : 1, A->zmode == 2 ? A->ref_pose : nullptr};
  const int b = A->b, n = A->n, m = A->m;
  const int pb = A->pix_begin, pe = (A->pix_end > 0) ? A->pix_end : n;
  if (pb < 0 || pe > n || pb >= pe) return COMO_ERR_ARG;
  const int nl = pe - pb;
  T* pair_T = (T*)A->ws_pair;                       // ws_pair: [12 b] target inverse poses | [2 b] affine | [12 b] reference poses
  T* pair_aff = pair_T + 12 * (long)b;
  T* pair_ref = pair_aff + 2 * (long)b;
  uint32_t* hists = (uint32_t*)A->ws_hists;

  if (A->phase & 1024) {
    // the pair constants alone (phase 1 = constants + residual pass): the residual pass then runs fused into the dense reference
    // (csrc/densify.hip como_dense_ref_fused_*), which needs ws_pair before the reference points exist
    if (!(A->phase & 256) && !zero_words(hists, 6 * SEL_BINS, s)) return COMO_ERR_LAUNCH;
    hipLaunchKernelGGL(ba_pair_setup_kernel<T>, dim3((b + 63) / 64), dim3(64), 0, s, (const T*)A->poses_all,
                       (const T*)A->aff_all, pr, b, pair_T, pair_aff, pair_ref);
    COMO_CHECK_LAUNCH();
  }
  if (A->phase & 1) {
    if (!(A->phase & 256) && !zero_words(hists, 6 * SEL_BINS, s)) return COMO_ERR_LAUNCH;
    hipLaunchKernelGGL(ba_pair_setup_kernel<T>, dim3((b + 63) / 64), dim3(64), 0, s, (const T*)A->poses_all,
                       (const T*)A->aff_all, pr, b, pair_T, pair_aff, pair_ref);
    COMO_CHECK_LAUNCH();
    // ~1024 workgroups in total: every workgroup ends with global atomics on the few hot digit-0 bins, which
    // serialise per address at the memory side (~15 ns each) -- 14k workgroups cost > 200 us there.
    int gx = (nl + 255) / 256;
    const int cap = (1024 + b - 1) / b;
    if (gx > cap) gx = cap;
    if (A->zmode >= 1)
      hipLaunchKernelGGL((ba_residual_kernel<T, true>), dim3(gx, b), dim3(256), 0, s, (const T*)A->Pwn, (const T*)A->vals, pr,
                         pair_T, pair_aff, (const T*)A->img_base, (const T*)A->K, A->H, A->W, n, pb, pe, (T*)A->ws_r,
                         (uint8_t*)A->ws_valid, (T*)A->pj_out, hists);
    else
      hipLaunchKernelGGL((ba_residual_kernel<T, false>), dim3(gx, b), dim3(256), 0, s, (const T*)A->Pwn, (const T*)A->vals, pr,
                         pair_T, pair_aff, (const T*)A->img_base, (const T*)A->K, A->H, A->W, n, pb, pe, (T*)A->ws_r,
                         (uint8_t*)A->ws_valid, (T*)A->pj_out, hists);
    COMO_CHECK_LAUNCH();
  }
  // digit passes 1..P-1 (multi-GPU: the caller all-reduces hists between phases 1, 2a.. and 4)
  for (int ps = 1; ps < SelCfg<KeyT>::NPASS; ++ps) {
    if (A->phase & (2 << (ps - 1))) {
      // all later passes in this one call (single GPU): a double select may finish digits 4, 5 from collected candidates
      // (select.hip); the multi-GPU protocol runs one pass per call with a histogram all-reduce in between -> plain passes
      // (phase bit 512, multi-GPU double select: pass 3 collects the candidates for ONE exchange, como_select_cand_*: no tail)
      const int collect = ((A->phase & 0x3E) == 0x3E) ? 0x100 : ((A->phase & 512) && ps == 3 ? 0x300 : 0);
      int rc = select_hist<T>((const T*)A->ws_r, (const uint8_t*)A->ws_valid, (long)b * nl, 1, hists, ps | collect, s);
      if (rc) return rc;
    }
  }
  if (A->phase & 64) {
    const int chunks = A->chunks;
    if (chunks <= 0) return COMO_ERR_ARG;
    int chunk_len = (nl + chunks - 1) / chunks;
    chunk_len = ((chunk_len + 255) / 256) * 256;
    if ((long)chunk_len * chunks < nl) return COMO_ERR_ARG;
    dim3 grid(chunks, b), blk(256);
#define LAUNCH_BLOCKS(ZM)                                                                                            \
  hipLaunchKernelGGL((ba_blocks_kernel<T, ZM>), grid, blk, 0, s, (const T*)A->Pwn, (const T*)A->vals,                 \
                     (const T*)A->dPwn_dTwc, (const T*)A->zjac, (const T*)A->uvec, A->pixidx, (const T*)A->invz,      \
                     A->kt_slot_stride, pr, pair_T, pair_aff, pair_ref, (const T*)A->img_base, (const T*)A->K, A->H,  \
                     A->W, n, m, pb, pe, chunk_len, hists, (T*)A->ws_partials, (T*)A->sigma_out)
    // zmode 0 (the reference's materialised dPwn_dzm) and zmode 1 (materialised dPwn_dTwc / uvec planes) run the plain
    // kernel; the tuned kernels below take the compact dense reference (zmode 2).  variant 1 = plain kernel for A/B runs.
    const bool grouped = A->grp_pairs && A->ngrp > 0;
    if (A->zmode == 0) {
      LAUNCH_BLOCKS(0);
    } else if (A->zmode == 1) {
      LAUNCH_BLOCKS(1);
    } else if (A->variant == 1) {
      LAUNCH_BLOCKS(2);
    } else {
#define PIPE_ARGS(NPAIR_ROWS, MAP)                                                                                     \
  dim3(chunks, NPAIR_ROWS), blk, 0, s, (const float*)A->Pwn, (const float*)A->vals, (const float*)A->dPwn_dTwc,        \
      (const float*)A->zjac, A->pixidx, (const float*)A->invz, A->kt_slot_stride, pr, (const float*)pair_T,            \
      (const float*)pair_aff, (const float*)pair_ref, (const float*)A->img_base, (const float*)A->K, A->H, A->W, n, m,  \
      pb, pe, chunk_len, hists, (float*)A->ws_partials, (float*)A->sigma_out, A->stagger, MAP
      if constexpr (sizeof(T) == 4) {
        // variant 0: two waves per SIMD (no spills).  Measurement builds only (-DCOMO_AB_VARIANTS, `python -m como_amd.build --ab`):
        // 3: one wave per SIMD; 11 / 12: timing ablations; 2: one pair at a time
#ifdef COMO_AB_VARIANTS
        if (A->variant == 3) { hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 1>), PIPE_ARGS(b, (const int*)nullptr)); }
        else if (A->variant == 11) { hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 2, 1>), PIPE_ARGS(b, (const int*)nullptr)); }
        else if (A->variant == 12) { hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 2, 2>), PIPE_ARGS(b, (const int*)nullptr)); }
        else if (A->variant == 2) { hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 2>), PIPE_ARGS(b, (const int*)nullptr)); }
        else
#else
        if (A->variant != 0) return COMO_ERR_ARG;          // (variant 1 = the plain kernel was taken above)
#endif
        if (grouped && (A->nsingle == 0 || A->single_pairs)) {
          // pairs that share their reference keyframe go through the two-pair kernel, the rest through the one-pair kernel
          // two waves per SIMD and a 4-deep K~ ring (256 VGPRs): 329 us; one wave per SIMD with an 8-deep ring: 371 us
          hipLaunchKernelGGL((ba_blocks_pair2_kernel<2, 4>), dim3(chunks, A->ngrp), blk, 0, s, (const float*)A->Pwn,
                             (const float*)A->vals, (const float*)A->dPwn_dTwc, (const float*)A->zjac,
                             A->pixidx, (const float*)A->invz, A->kt_slot_stride, pr, (const float*)pair_T,
                             (const float*)pair_aff, (const float*)pair_ref, (const float*)A->img_base, (const float*)A->K,
                             A->H, A->W, n, m, pb, pe, chunk_len, hists, (float*)A->ws_partials, (float*)A->sigma_out,
                             A->grp_pairs);
          if (A->nsingle > 0) {
            COMO_CHECK_LAUNCH();
            hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 2>), PIPE_ARGS(A->nsingle, A->single_pairs));
          }
        }
        else { hipLaunchKernelGGL((ba_blocks_pipe_kernel<float, 2>), PIPE_ARGS(b, (const int*)nullptr)); }
      } else {
        // float64 = the reference's mapping dtype.  Default (variant 0): the role-split two-pair kernel, software-pipelined,
        // ONE wave per SIMD with a 4-deep K~ ring: 580 us on the dense 8-keyframe window (8-deep ring, variant 3: 615 us; two
        // waves per SIMD without the pipeline, variant 4: 679 us; with it -- 147 spilled registers --, variant 5: 1169 us).
        // 9 = 2-deep ring; 6 / 7 / 8 / 10 = the wave-specialised kernel (one
        // producer + two consumer waves per workgroup: 754 ... 786 us whatever the ring depth -- the dispatcher spreads the
        // waves over the SIMDs without regard to their role, so some SIMDs host three consumers and others none).
        // (32-bit K~ row offsets: a slot of the predictor must stay below 4 GiB, else the plain kernel)
        const bool fits32 = (unsigned long)A->kt_slot_stride * sizeof(T) < (1ul << 32) && (unsigned long)n * m * sizeof(T) < (1ul << 32);
#define F64_ARGS                                                                                                      \
  dim3(chunks, A->ngrp), dim3(128), 0, s, (const double*)A->Pwn, (const double*)A->vals, (const double*)A->dPwn_dTwc,   \
      (const double*)A->zjac, A->pixidx, (const double*)A->invz, A->kt_slot_stride, pr, (const double*)pair_T,          \
      (const double*)pair_aff, (const double*)pair_ref, (const double*)A->img_base, (const double*)A->K, A->H, A->W, n,  \
      m, pb, pe, chunk_len, hists, (double*)A->ws_partials, (double*)A->sigma_out, A->grp_pairs
#ifndef COMO_AB_VARIANTS
        if (A->variant != 0) return COMO_ERR_ARG;          // (variant 1 = the plain kernel was taken above)
        if (grouped && A->nsingle == 0 && fits32) {
          hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<4, true, 1>), F64_ARGS);
        } else {
          LAUNCH_BLOCKS(2);
        }
#else
        if (A->variant != 2 && grouped && A->nsingle == 0 && fits32) {
          if (A->variant == 0) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<4, true, 1>), F64_ARGS); }
          else if (A->variant == 13) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<4, true, 1, true>), F64_ARGS); }
          else if (A->variant == 3) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<8, true, 1>), F64_ARGS); }
          else if (A->variant == 9) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<2, true, 1>), F64_ARGS); }
          else if (A->variant == 4) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<COMO_F64_PF, false, 2>), F64_ARGS); }
          else if (A->variant == 5) { hipLaunchKernelGGL((ba_blocks_pair2_f64_kernel<COMO_F64_PF, true, 2>), F64_ARGS); }
          else {
#define WS_ARGS                                                                                                       \
  dim3(chunks, A->ngrp), dim3(192), 0, s, (const double*)A->Pwn, (const double*)A->vals, (const double*)A->dPwn_dTwc,   \
      (const double*)A->zjac, A->pixidx, (const double*)A->invz, A->kt_slot_stride, pr, (const double*)pair_T,          \
      (const double*)pair_aff, (const double*)pair_ref, (const double*)A->img_base, (const double*)A->K, A->H, A->W, n,  \
      m, pb, pe, chunk_len, hists, (double*)A->ws_partials, (double*)A->sigma_out, A->grp_pairs
            // K~ ring depths (Z, T) and waves per SIMD; 6 / 7 / 8: tuning runs
            if (A->variant == 6) { hipLaunchKernelGGL((ba_blocks_ws_f64_kernel<4, 4, 3>), WS_ARGS); }
            else if (A->variant == 7) { hipLaunchKernelGGL((ba_blocks_ws_f64_kernel<2, 2, 3>), WS_ARGS); }
            else if (A->variant == 8) { hipLaunchKernelGGL((ba_blocks_ws_f64_kernel<8, 8, 2>), WS_ARGS); }
            else if (A->variant == 10) { hipLaunchKernelGGL((ba_blocks_ws_f64_kernel<4, 2, 3>), WS_ARGS); }
            else return COMO_ERR_ARG;
#undef WS_ARGS
          }
        } else {
          LAUNCH_BLOCKS(2);
        }
#endif
#undef F64_ARGS
      }
#undef PIPE_ARGS
    }
#undef LAUNCH_BLOCKS
    COMO_CHECK_LAUNCH();
  }
  if (A->phase & 128) {
    if (!A->pose_ref_inds || !A->pose_tgt_inds || !A->landmark_inds || !A->dzdP || !A->Hmat ||
        (A->h_is_f64 != 2 && (!A->gvec || !A->err_out)))
      return COMO_ERR_ARG;
    const int nrec = A->chunks;
#define LAUNCH_ASM(TH, MODE)                                                                                         \
  hipLaunchKernelGGL((ba_reduce_assemble_kernel<T, TH, MODE>), dim3((BACfg::REC + 255) / 256, b), dim3(256), 0, s,    \
                     (const T*)A->ws_partials, nrec, pr, A->pose_ref_inds, A->pose_tgt_inds, A->landmark_inds,        \
                     (const T*)A->dzdP, m, (TH*)A->Hmat, A->D, (TH*)A->gvec, (double*)A->err_out,                     \
                     (double*)A->pair_blocks_out, A->fix_plane, (long long*)A->blocks_fix)
    if (A->h_is_f64 == 2) {
      if (A->fix_plane < A->D * A->D + A->D + FIX_ERR_SLOTS) return COMO_ERR_ARG;
      // fix_add's fraction plane holds 2^-56 units in 64 bits: 256 contributions to ONE entry are safe, more could lose a carry
      // silently.  An entry receives at most one contribution per pair entry (the error slot, shared landmark diagonals) plus
      // the <= 6 prior terms: refuse a batch that could exceed it instead of returning wrong normal equations.
      if (b > 240) return COMO_ERR_ARG;
      if (A->reduce_mode != 0 && !A->blocks_fix) return COMO_ERR_ARG;
      if (A->reduce_mode == 1) { LAUNCH_ASM(long long, 1); }
      else if (A->reduce_mode == 2) { LAUNCH_ASM(long long, 2); }
      else if (A->asm_grp_start && A->asm_grp_list && A->n_asm_grp > 0 && !A->pair_blocks_out) {
        hipLaunchKernelGGL((ba_reduce_assemble_grouped_kernel<T>), dim3((BACfg::NT * 256 + BACfg::NB * 16 + 1 + 63) / 64, A->n_asm_grp), dim3(256), 0, s,
                           (const T*)A->ws_partials, nrec, A->asm_grp_start, A->asm_grp_list, pr, A->pose_ref_inds, A->pose_tgt_inds,
                           A->landmark_inds, (const T*)A->dzdP, m, (long long*)A->Hmat, A->D, A->fix_plane);
      }
      else { LAUNCH_ASM(long long, 0); }
    } else if (A->h_is_f64) { LAUNCH_ASM(double, 0); } else { LAUNCH_ASM(float, 0); }
#undef LAUNCH_ASM
    COMO_CHECK_LAUNCH();
  }
  return COMO_OK;
}

}  // namespace como

extern "C" {

long como_ba_partials_elems(int b, int chunks, int m) {
  (void)m;
  return (long)b * chunks * como::BACfg::REC;
}

long como_sys_fix_plane_elems(long D) { return D * D + D + como::FIX_ERR_SLOTS; }

int como_sys_finalize(const void* sysfix, long fix_plane, long D, double* H, double* g, double* err8, como_stream_t stream) {
  if (!sysfix || !H || !g || D <= 0 || fix_plane < D * D + D + como::FIX_ERR_SLOTS) return COMO_ERR_ARG;
  const long total = D * D + D + como::FIX_ERR_SLOTS;
  hipLaunchKernelGGL(como::sys_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)sysfix, fix_plane, D, H, g, err8);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_sys_finalize_pack(const void* sysfix, long fix_plane, long D, double* H, double* g, double* err8, void* chol_workspace,
                           int* info, como_stream_t stream) {
  if (!sysfix || !H || !g || !chol_workspace || !info || D <= 0 || D > 4000 || fix_plane < D * D + D + como::FIX_ERR_SLOTS)
    return COMO_ERR_ARG;
  const long Dp = como::chol_dp(D);                // como_chol_workspace_bytes' padding (common.cuh)
  const long total = Dp * Dp;
  hipLaunchKernelGGL(como::sys_finalize_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)sysfix, fix_plane, D, H, g, err8, (double*)chol_workspace, Dp, info);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_ba_linearize_f32(const como_ba_args* a, como_stream_t stream) { return como::ba_linearize<float>(a, (hipStream_t)stream); }
int como_ba_linearize_f64(const como_ba_args* a, como_stream_t stream) { return como::ba_linearize<double>(a, (hipStream_t)stream); }

}  // extern "C"
