// Inverse-compositional photometric tracking: one Gauss-Newton iteration on the device.
//
// Reference path: como/odom/frontend/photo_tracking.py:117-143 (`tracking_iter`); c image channels = c residual entries per
// pixel (element = pixel * c + channel of vals_i / J8), gray is c = 1
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
    uint32_t* __restrict__ hists, const uint8_t* __restrict__ in_mask, int C) {
  // C image channels (`color: rgb`): element e = pixel * C + channel of vals_i (N,C), J8 (N,C,8), r_out / valid_out (N,C);
  // every (pixel, channel) residual is one entry of the median and of the sums (photo_tracking.py:117-143, 77-93)
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
  const long NE = N * C;
  const long iters = (NE + stride - 1) / stride;              // uniform trip count (the aggregated histogram needs whole waves)
  for (long it = 0; it < iters; ++it) {
    const long i0 = it * stride + (long)blockIdx.x * 256 + threadIdx.x;
    const bool inr = i0 < NE;
    const long e = inr ? i0 : NE - 1;
    const long i = C == 1 ? e : (long)((unsigned long)e / (unsigned)C);     // pixel
    const int ch = (int)(e - i * C);
    const T X = P[3 * i + 0], Y = P[3 * i + 1], Z = P[3 * i + 2];
    T hx, hy, hz;
    rigid_apply(Pm, X, Y, Z, hx, hy, hz);          // p_h = A P + b
    const T u = hx / hz, v = hy / hz;              // coords = p_h[:2] / depth
    // in_mask: the caller's reference-side selection (photo_tracking.py:20-26 gathers vals/P/dI_dT with it); a point that
    // is masked out behaves exactly like one that projects outside the image
    const bool ok = in_image(u, v, H, W) && (hz > T(0)) && (in_mask == nullptr || in_mask[i] != 0);
    Taps<T> t = make_taps(grid_position(u, W, ax), grid_position(v, H, ay), H, W);
    const T It = tap_sum(img + (long)ch * H * W, t);
    const T tmp = ea * It;                          // photo_tracking.py:124
    const T r = (tmp + bb) - vals_i[e];
    if (inr) {
      J8[8 * e + 6] = -tmp;                         // dI_dT[..., 6] = -tmp (in-place, photo_tracking.py:125)
      r_out[e] = r;
      valid_out[e] = ok ? 1 : 0;
      if (ch == 0) {
        if (pj_out) { pj_out[2 * i] = u; pj_out[2 * i + 1] = v; }
        if (depth_out) depth_out[i] = hz;
      }
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

// The 8x8 solve + pose update of one tracking iteration (photo_tracking.py:96-114: cholesky_ex / cholesky_solve, T <- T Exp(-delta)),
// float64, ONE definition for the per-iteration chain (track_finish_kernel) and the persistent level kernel -- both run it on one
// lane, on the critical path of every iteration.  The factorisation keeps 1 / L_jj (one v_rsq_f64-based reciprocal square root per
// column) and multiplies by it: 8 of them instead of 8 square roots + 52 divisions (~30 dependent instructions each), 3.0 -> 1.3 us.
// info as torch.linalg.cholesky_ex: 0, or the 1-based column of the first non-positive pivot.
struct TrkSolve { double d[8], Tn[16], gn, dn; int info; };
// index of H[a][b] (a <= b) in the 36 upper-triangle sums
__device__ __forceinline__ constexpr int trk_q(int a, int b) { return a * 8 - a * (a - 1) / 2 + (b - a); }
template <typename T>
__device__ inline void trk_solve8(const double* __restrict__ tot, const T* __restrict__ Tcur, TrkSolve& S) {
  double L[36], rd[8], y[8];          // lower triangle, row-major packed: L[i][j] at i (i + 1) / 2 + j
  double gn = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) gn += tot[36 + a] * tot[36 + a];
  S.gn = gn;
  int info = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double s = tot[trk_q(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    if (!(s > 0) && info == 0) info = j + 1;
    const double r = rsqrt(s);
    rd[j] = r;
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      double t = tot[trk_q(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      L[i * (i + 1) / 2 + j] = t * r;
    }
  }
  S.info = info;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double t = tot[36 + i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = t * rd[i];
  }
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    double t = y[i];
#pragma unroll
    for (int k = i + 1; k < 8; ++k) t -= L[k * (k + 1) / 2 + i] * S.d[k];
    S.d[i] = t * rd[i];
  }
  double xi[6], E[16], dn = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) xi[i] = -S.d[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) dn += S.d[i] * S.d[i];
  S.dn = dn;
  se3_exp_f64(xi, E);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) t += (double)Tcur[i * 4 + k] * E[k * 4 + j];
      S.Tn[i * 4 + j] = t;
    }
  }
}
// H (8x8, symmetric) and g of an iteration's record from the 46 sums
template <typename T>
__device__ inline void trk_store_Hg(const double* __restrict__ tot, T* __restrict__ out) {
  int q = 0;
  for (int a = 0; a < 8; ++a)
    for (int b = a; b < 8; ++b) { const T v = (T)tot[q++]; out[a * 8 + b] = v; out[b * 8 + a] = v; }
  for (int a = 0; a < 8; ++a) out[64 + a] = (T)tot[36 + a];
}

// out layout (T): [0:64) H | [64:72) g | [72:80) delta | [80:96) T_new | [96:98) aff_new |
//                 98 mse | 99 grad_norm | 100 total_err | 101 sigma | 102 nvalid | 103 delta_norm | 104 chol_info
template <typename T>
__global__ __launch_bounds__(256) void track_finish_kernel(const double* __restrict__ partials, int nblocks,
                                                           const uint32_t* __restrict__ hists,
                                                           const T* __restrict__ Tji, const T* __restrict__ aff,
                                                           T* __restrict__ out, int C) {
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
    TrkSolve S;
    trk_solve8<T>(tot, Tji, S);
    trk_store_Hg<T>(tot, out);
    for (int i = 0; i < 8; ++i) out[72 + i] = (T)S.d[i];
    for (int i = 0; i < 16; ++i) out[80 + i] = (T)S.Tn[i];
    out[96] = (T)((double)aff[0] - S.d[6]);
    out[97] = (T)((double)aff[1] - S.d[7]);
    out[98] = (T)(tot[44] / (double)(nv / C));    // mean over the valid PIXELS (photo_tracking.py:83-85), nv counts channels
    out[99] = (T)sqrt(S.gn);
    out[100] = (T)tot[44];
    out[101] = T(1.4826) * key_value(prefix);
    out[102] = (T)(nv / C);
    out[103] = (T)sqrt(S.dn);
    out[104] = (T)S.info;
  }
}

template <typename T>
int track_iter(const T* Tji, const T* Kmat, const T* aff, const T* P, const T* vals_i, const T* img, int H, int W,
               long N, T* J8, T* r_ws, uint8_t* valid_out, T* pj_out, T* depth_out, void* hists_v, double* partials,
               T* out, const uint8_t* in_mask, hipStream_t s, int C = 1) {
  using KeyT = typename KeyOf<T>::type;
  if (!Tji || !Kmat || !aff || !P || !vals_i || !img || !J8 || !r_ws || !valid_out || !hists_v || !partials || !out ||
      N <= 0 || H < 3 || W < 3 || C < 1 || C > 4)
    return COMO_ERR_ARG;
  const long NE = N * C;                             // (pixel, channel) elements
  uint32_t* hists = (uint32_t*)hists_v;
  if (!zero_words(hists, 6 * SEL_BINS, s)) return COMO_ERR_LAUNCH;
  long blocks = (NE + 255) / 256;
  if (blocks > 1024) blocks = 1024;   // bounded: the digit-0 flush serialises per hot bin at the memory-side atomics
  hipLaunchKernelGGL(track_residual_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, Tji, Kmat, aff, P, vals_i, img,
                     H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists, in_mask, C);
  COMO_CHECK_LAUNCH();
  for (int p = 1; p < SelCfg<KeyT>::NPASS; ++p) {
    int rc = select_hist<T>(r_ws, valid_out, NE, 1, hists, p, s);
    if (rc) return rc;
  }
  // ~4 pixels per thread: the 46-value block reduction (shuffles + LDS) is amortised, and track_finish sums fewer partials
  int rblocks = (int)((NE + 1023) / 1024);
  if (rblocks < 1) rblocks = 1;
  if (rblocks > TRK_MAX_BLOCKS) rblocks = TRK_MAX_BLOCKS;
  hipLaunchKernelGGL(track_reduce_kernel<T>, dim3(rblocks), dim3(256), 256 * (TRK_ACC + 1) * sizeof(T), s, J8, r_ws, valid_out, NE, hists,
                     partials);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(track_finish_kernel<T>, dim3(1), dim3(256), 0, s, partials, rblocks, hists, Tji, aff, out, C);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}


// ---------------------------------------- one pyramid level in ONE launch ------------------------------------------
// photo_level_tracking (photo_tracking.py:147-185): the whole Gauss-Newton loop of a level -- every iteration's warp /
// residual / exact median / Huber-weighted 8x8 system / solve / pose update AND the stop test -- in one persistent kernel.
// The chain above is 6 launches + a host read-back of three scalars per iteration (62 us per iteration at 640x480, all of
// it launch / dependency latency: an iteration moves 16 MB).  Here:
//   * every thread keeps its (up to TL_MAXP) reference pixels -- P, I_ref, the 7 constant Jacobian entries, the mask bit --
//     in REGISTERS for the whole level: from the second iteration on only the target image (L2-resident) is read;
//   * the four dependency points of an iteration (three radix-select digit histograms, the 46 sums) are device-wide
//     barriers instead of kernel boundaries;
//   * everything that crosses workgroups travels through device-scope ATOMICS (histogram bins, exact fixed-point sums,
//     barrier counters) and is read back with device-scope atomic loads; the 46 sums are order-independent, so every
//     workgroup reads the same bits; one wave per workgroup brackets the barrier with a release / acquire fence (L2 drain /
//     invalidate: measured necessary -- the XCD-private L2 posts atomics and keeps copies of device-scope loads);
//   * the 8x8 solve, the pose update and the stop test are computed by EVERY workgroup redundantly from those sums, so all
//     workgroups take the same exit without another exchange.
// MI355X: the 8 XCDs have private L2s; device-scope traffic is served by the memory-side cache, ~1.5 us per round trip --
// the barrier is one non-returning atomic per workgroup (8 counters, one per XCD-aligned residue class) + one polling round.
// All workgroups must be co-resident: the host launches at most one workgroup per compute unit.
constexpr int TL_MAXP = 5;          // reference pixels per thread
constexpr int TL_NC = 8;            // arrival counters of the device-wide barrier (workgroup b -> counter b % 8)
constexpr int TL_BAR_WORDS = 32 * (TL_NC + 2);       // counters 128 B apart, then the error flag, then the XCD census (own line)
constexpr int TL_XCC_WORD = 32 * (TL_NC + 1);        // 64-bit word: byte x = number of participating workgroups that run on XCD x
// device-wide sums: per parity THREE planes of 64 (46 used) exact fixed-point sums -- integer parts [plane][64], then fractions:
// plane 0 = pixels that are inliers of the Huber weight for every robust scale the median can still take, plane 1 = the
// certain outliers (band-split form, see the kernel), plane 2 = the sums of the exact form (scale known)
constexpr int TL_PLANE = 64;
constexpr int TL_NPLANE = 3;
constexpr int TL_SUM_PAR = 2 * TL_NPLANE * TL_PLANE;
constexpr int TL_SUM_WORDS = 2 * TL_SUM_PAR;         // two parities
constexpr int TL_POISON = 63;       // word of plane 0's integer parts that counts non-finite shares (TRK_ACC <= 63)
constexpr int TL_AMB_CAP = 128;     // pixels whose weight class depends on the last 10 bits of the median: region 4 of the histograms
constexpr int TL_AMB_WORDS = 12;    // [J0 J1 | J2 J3 | J4 J5 | J6 J7 | r -] as five 64-bit words (+ one spare), from word 16 on

template <typename U>
__device__ __forceinline__ U ld_dev(const U* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename U>
__device__ __forceinline__ void st_dev(U* p, U v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct TLBarrier {
  unsigned* bar;       // [32 c] arrival counter c, [32 TL_NC] error flag
  unsigned epoch;
  int G;
  int cached;          // the workspace lives in ordinary (L2-cacheable) device memory: bracket the barrier with an L2 invalidate
  int local;           // XCD-local mode: every participating workgroup sits behind ONE L2 (see track_level_kernel)
  int bidx;            // index of this workgroup among the participants
};

// returns false if the barrier timed out (a workgroup never arrived: not co-resident, or the device is wedged)
__device__ __forceinline__ bool tl_barrier(TLBarrier& B) {
  __shared__ int ok_s;
  __syncthreads();                    // every wave's atomics / stores of this phase are issued and acknowledged (vmcnt(0))
  B.epoch += 1;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    // (No release fence: every cross-workgroup write of a phase is a RETURNING atomic whose value the issuing wave has
    // already received -- see `flush` -- i.e. it is performed at the memory side before this arrival is issued.  Measured:
    // a release fence = L2 write-back costs 3.7 us per barrier, waiting for the returns 2 us.)
    if (lane == 0) {
      // (XCD-local: the read-modify-write is performed in the one L2 all participants share -- no trip to the memory side)
      if (B.local) __hip_atomic_fetch_add(&B.bar[32 * (B.bidx % TL_NC)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&B.bar[32 * (B.bidx % TL_NC)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned target = (unsigned)B.G * B.epoch;
    unsigned* errf = B.bar + 32 * TL_NC;
    int ok = 1;
    for (long spin = 0;; ++spin) {
      unsigned v = (lane < TL_NC) ? ld_dev(&B.bar[32 * lane]) : 0u;
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      v = __shfl(v, 0, 64);
      if (v >= target) break;
      if ((spin & 255) == 255 && ld_dev(errf)) { ok = 0; break; }
      if (spin > 400000) { if (lane == 0) atomicExch(errf, 1u); ok = 0; break; }      // ~0.4 s
      __builtin_amdgcn_s_sleep(1);
    }
    // acquire (cacheable workspace only): drop this XCD's L2 copies of the words other XCDs have updated since (device-scope
    // loads allocate in L2; a recycled histogram would be read stale).  1.7 us per barrier; an UNCACHED workspace
    // (como_track_level_workspace_create) does not need it.
    if (B.cached && !B.local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0) ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

struct TLCriteria { int max_iter; float delta_norm, rel_tol, grad_norm; };

// -DCOMO_TL_PROFILE: workgroup 0 stamps the phases of iteration 3 (100 MHz wall clock) into the tail of the workspace
#ifdef COMO_TL_PROFILE
#define TL_STAMP(k) do { if (bidx == 0 && tid == 0 && it == 3) stamps[k] = (long long)wall_clock64(); } while (0)
#else
#define TL_STAMP(k) do { } while (0)
#endif

// one more digit of the exact k-th key: `hist` (2048 device-scope counters) is the histogram of this digit among the keys
// that match the digits resolved so far; returns the digit and lowers k_rem to the rank inside that bin
// (the loads and the scan are separate so that the counters of the NEXT digit's speculated histogram travel with this digit's)
__device__ __forceinline__ void tl_load_hist(const uint32_t* hist, uint32_t (&c)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = ld_dev(&hist[threadIdx.x * 8 + j]);
}
__device__ __forceinline__ uint32_t tl_resolve_loaded(const uint32_t (&c)[8], uint32_t& k_rem, uint32_t* total, SelScratch* sc) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t local = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) local += c[j];
  uint32_t incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) sc->wave_tot[wv] = incl;
  if (tid == 0) { sc->found_bin = 0; sc->found_below = 0; }
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wv; ++w) base += sc->wave_tot[w];
  const uint32_t tot = sc->wave_tot[0] + sc->wave_tot[1] + sc->wave_tot[2] + sc->wave_tot[3];
  if (total) { *total = tot; k_rem = tot ? (tot - 1) / 2 : 0; }        // first digit: lower median of all valid keys
  const uint32_t excl = base + incl - local;
  if (local > 0 && k_rem >= excl && k_rem < excl + local) {
    uint32_t run = excl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (k_rem >= run && k_rem < run + c[j]) { sc->found_bin = tid * 8 + j; sc->found_below = run; }
      run += c[j];
    }
  }
  __syncthreads();
  const uint32_t bin = sc->found_bin;
  k_rem -= sc->found_below;
  __syncthreads();
  return bin;
}
__device__ __forceinline__ uint32_t tl_resolve_digit(const uint32_t* hist, uint32_t& k_rem, uint32_t* total, SelScratch* sc) {
  uint32_t c[8];
  tl_load_hist(hist, c);
  return tl_resolve_loaded(c, k_rem, total, sc);
}

// Sum over the 64 lanes of a wave of N values per lane in ~N shuffles instead of 6 N: at every step a lane keeps one half of its
// values, hands the other half to its partner (lane ^ mask) and adds what it receives -- the values halve while the lanes summed
// double.  Fixed tree: the result does not depend on anything but the inputs.  Afterwards lane l holds the complete sums of the
// values with index wr_index(l) (+ 64 for the second one); indices >= N are padding.
__device__ __forceinline__ int wr_index(int lane) {
  return ((lane & 1) << 5) | ((lane & 2) << 3) | ((lane & 4) << 1) | ((lane & 8) >> 1) | ((lane & 16) >> 3) | ((lane & 32) >> 5);
}
template <int NIN>
__device__ __forceinline__ void wr_step(const float (&in)[NIN], float (&out)[NIN / 2], int mask, bool hi) {
#pragma unroll
  for (int i = 0; i < NIN / 2; ++i) {
    const float keep = hi ? in[2 * i + 1] : in[2 * i];
    const float send = hi ? in[2 * i] : in[2 * i + 1];
    out[i] = keep + __shfl_xor(send, mask, 64);
  }
}
__device__ __forceinline__ void wave_reduce96(const float (&a)[96], int lane, float& r0, float& r1) {
  float b[48], c[24], d[12], e[6], f[3], g[4], h[2];
  wr_step<96>(a, b, 32, (lane & 32) != 0);
  wr_step<48>(b, c, 16, (lane & 16) != 0);
  wr_step<24>(c, d, 8, (lane & 8) != 0);
  wr_step<12>(d, e, 4, (lane & 4) != 0);
  wr_step<6>(e, f, 2, (lane & 2) != 0);
  g[0] = f[0]; g[1] = f[1]; g[2] = f[2]; g[3] = 0.f;
  wr_step<4>(g, h, 1, (lane & 1) != 0);
  r0 = h[0]; r1 = h[1];
}
__device__ __forceinline__ float wave_reduce48(const float (&a)[48], int lane) {
  float c[24], d[12], e[6], f[3], g[4], h[2], k[1];
  wr_step<48>(a, c, 32, (lane & 32) != 0);
  wr_step<24>(c, d, 16, (lane & 16) != 0);
  wr_step<12>(d, e, 8, (lane & 8) != 0);
  wr_step<6>(e, f, 4, (lane & 4) != 0);
  g[0] = f[0]; g[1] = f[1]; g[2] = f[2]; g[3] = 0.f;
  wr_step<4>(g, h, 2, (lane & 2) != 0);
  wr_step<2>(h, k, 1, (lane & 1) != 0);
  return k[0];
}

// 24 values per lane: five halving steps, then one butterfly add -- lanes l and l ^ 1 both hold the complete sum of the value
// with index wr_index24(l) (>= 24: padding)
__device__ __forceinline__ int wr_index24(int lane) {
  return ((lane & 2) << 3) | ((lane & 4) << 1) | ((lane & 8) >> 1) | ((lane & 16) >> 3) | ((lane & 32) >> 5);
}
__device__ __forceinline__ float wave_reduce24(const float (&a)[24], int lane) {
  float d[12], e[6], f[3], g[4], h[2], k[1];
  wr_step<24>(a, d, 32, (lane & 32) != 0);
  wr_step<12>(d, e, 16, (lane & 16) != 0);
  wr_step<6>(e, f, 8, (lane & 8) != 0);
  g[0] = f[0]; g[1] = f[1]; g[2] = f[2]; g[3] = 0.f;
  wr_step<4>(g, h, 4, (lane & 4) != 0);
  wr_step<2>(h, k, 2, (lane & 2) != 0);
  return k[0] + __shfl_xor(k[0], 1, 64);
}

typedef float tl_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void track_level_kernel(
    const float* __restrict__ Tji_init, const float* __restrict__ Kmat, const float* __restrict__ aff_init,
    const float* __restrict__ P, const float* __restrict__ vals_i, const float* __restrict__ img, int H, int W, long N,
    const float* __restrict__ J8, const uint8_t* __restrict__ in_mask, TLCriteria crit, unsigned* __restrict__ bar,
    uint32_t* __restrict__ hists2, long long* __restrict__ sums2, long long* __restrict__ stamps, float* __restrict__ out,
    int ppt, int ws_cached, int C, int xl, int split, int amb_cap) {
  using T = float;
  using KeyT = uint32_t;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ uint32_t lh1[SEL_BINS];   // speculative digit-1 histogram (see `spec_bin`)
  __shared__ SelScratch sc;
  __shared__ float redw[4][96];        // per-wave sums (wave_reduce96 / wave_reduce48)
  __shared__ float ambs[TL_AMB_CAP][10];
  __shared__ double tot[TRK_ACC];
  __shared__ float state[24];          // 16 T | 2 aff | mse | gnorm | dnorm
  __shared__ float rec_s[112];         // the iteration's record, staged (one lane's 106 global stores were 1 us per iteration)
  // XCD-LOCAL mode (xl, the coarse levels: <= 32 workgroups of pixels): the grid is launched 8x oversized and only the workgroups
  // the dispatcher places on XCD 0 stay (workgroup b goes to XCD b % 8: verified once per process by como_track_level_probe) --
  // all participants then sit behind ONE L2, so the counters, histograms and sums are updated by plain L2 read-modify-writes
  // (workgroup-scope atomics, no wait for a value returned from the memory side) and read back with L1-bypassing loads: a barrier
  // is an L2 round trip (~0.5 us) instead of a trip to the memory side + an L2 invalidate (~4.5 us).
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xdbg = xl >> 1;            // (debug switch of the census test: workgroup 1 reports a neighbouring XCD)
  xl &= 1;
  if (xl && (blockIdx.x & 7)) return;
  const int bidx = xl ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int G = xl ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  TLBarrier B{bar, 0u, G, ws_cached, xl, bidx};
  // XCD census: that the kept workgroups share one XCD is an assumption about the dispatcher (round robin from XCD 0), checked
  // INSIDE the launch that depends on it: every participant adds one to the byte of the XCD it really runs on (HW_REG_XCC_ID),
  // memory-side (agent scope) on a line nothing else touches; workgroup 0 reads the census after the last iteration and
  // reports status -2 unless all G sit on one XCD (the host then discards the result and stops using this form).
  if (xl && tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 7u;
    if (xdbg && bidx == 1) id = (id + 1u) & 7u;
    __hip_atomic_fetch_add((unsigned long long*)(bar + TL_XCC_WORD), 1ull << (8 * id), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- this thread's reference pixels, register-resident for the whole level ----
  // (C channels: an element = (pixel, channel) of vals_i (N,C) / J8 (N,C,8); its target plane is img + channel * H * W)
  T pX[TL_MAXP], pY[TL_MAXP], pZ[TL_MAXP], vref[TL_MAXP], Jc[TL_MAXP][7];
  int choff[TL_MAXP];
  bool sel[TL_MAXP];
#pragma unroll
  for (int k = 0; k < TL_MAXP; ++k) {
    const long i = ((long)k * G + bidx) * 256 + tid;
    sel[k] = (k < ppt) && (i < N * C);
    const long ic = sel[k] ? i : 0;
    const long ip = C == 1 ? ic : (long)((unsigned)ic / (unsigned)C);
    choff[k] = (int)(ic - ip * C) * H * W;
    pX[k] = P[3 * ip]; pY[k] = P[3 * ip + 1]; pZ[k] = P[3 * ip + 2];
    vref[k] = vals_i[ic];
    const float4 a = *reinterpret_cast<const float4*>(&J8[8 * ic]);
    const float4 b = *reinterpret_cast<const float4*>(&J8[8 * ic + 4]);
    Jc[k][0] = a.x; Jc[k][1] = a.y; Jc[k][2] = a.z; Jc[k][3] = a.w; Jc[k][4] = b.x; Jc[k][5] = b.y; Jc[k][6] = b.w;   // [6] = J7
    if (sel[k] && in_mask) sel[k] = in_mask[ip] != 0;
  }
  T Tc[16], a0 = aff_init[0], a1 = aff_init[1];
#pragma unroll
  for (int k = 0; k < 16; ++k) Tc[k] = Tji_init[k];
  T Kr[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kr[k] = Kmat[k];
  const T ax = T(1) / T(W), ay = T(1) / T(H);
  T mse_prev = __builtin_inff();
  int it = 0;
  bool alive = true;
  // (row, column) of the upper-triangle entry this thread owns when it evaluates the few band pixels (tid < 36)
  int qa = 0, qb = 0;
  {
    int q = 0;
    for (int a = 0; a < 8; ++a)
      for (int b = a; b < 8; ++b) { if (q == tid) { qa = a; qb = b; } ++q; }
  }
  // the result record starts as "no iteration done": the initial pose / affine parameters -- a barrier time-out in the
  // first iteration (workgroups not co-resident) must not hand uninitialised memory to the caller as the tracked pose
  if (bidx == 0 && tid < 24) {
    if (tid < 16) out[80 + tid] = Tji_init[tid];
    else if (tid < 18) out[96 + (tid - 16)] = aff_init[tid - 16];
    else if (tid == 18) out[104] = T(0);
    else if (tid == 19) out[105] = T(0);
  }

  // RETURNING atomics: the returned value comes from where the read-modify-write was performed (the memory side), so once
  // the wave has it (s_waitcnt in the barrier's __syncthreads) the update is globally performed -- a non-returning atomic is
  // acknowledged by the XCD's L2 before that, and the barrier arrival (another address, another channel) can overtake it
  auto flush = [&](uint32_t* gh, const uint32_t* src = nullptr) {
    if (!src) src = lh;
    uint32_t sink = 0;
    for (int b = tid; b < SEL_BINS; b += 256) {
      const uint32_t v = src[b];
      if (v) {
        if (xl) __hip_atomic_fetch_add(&gh[b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else sink += __hip_atomic_fetch_add(&gh[b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("" ::"v"(sink));
  };
  // this workgroup's share of sum `k` of `plane` into the device-wide sums, exact fixed point (common.cuh fix_split):
  // order-independent.  Non-finite: counted in a dedicated word (cleared with the rest of the buffer), never encoded in the summed
  // value -- an additive sentinel wraps once enough workgroups add it (8 x 2^61 = 0)
  auto share = [&](long long* sm, int plane, int k, double v) {
    long long hi = 0;
    unsigned long long lo = 0, sink = 0;
    unsigned long long* ip = (unsigned long long*)&sm[plane * TL_PLANE + k];
    unsigned long long* fp = (unsigned long long*)&sm[TL_NPLANE * TL_PLANE + plane * TL_PLANE + k];
    if (xl) {
      if (!(fabs(v) < 4.0e18)) __hip_atomic_fetch_add((unsigned long long*)&sm[TL_POISON], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else fix_split(v, hi, lo);
      if (hi) __hip_atomic_fetch_add(ip, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lo) __hip_atomic_fetch_add(fp, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (!(fabs(v) < 4.0e18)) sink += __hip_atomic_fetch_add((unsigned long long*)&sm[TL_POISON], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else fix_split(v, hi, lo);
      if (hi) sink += __hip_atomic_fetch_add(ip, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lo) sink += __hip_atomic_fetch_add(fp, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::"v"(sink));
  };
  auto plane_total = [&](const long long* sm, int plane, int k) {
    const long long hi = ld_dev(&sm[plane * TL_PLANE + k]);
    const unsigned long long lo = (unsigned long long)ld_dev(&sm[TL_NPLANE * TL_PLANE + plane * TL_PLANE + k]);
    return fix_value(hi, lo);
  };

  // SPECULATION on the first digit of the median: the robust scale moves little between iterations, so phase A also counts
  // the SECOND digit of the keys whose first digit equals the previous iteration's (`spec_bin`), into region 3 of the
  // histograms.  If the first digit resolves to that bin again, the second digit's histogram is already complete and one of
  // the device-wide synchronisations of the iteration (phase B: histogram, flush, barrier) is skipped; if not, region 3
  // is ignored and phase B runs as always -- the result is the exact median either way.
  //
  // BAND-SPLIT SUMS (`split`): once 22 of the 32 key bits of the median are known, the robust scale sigma = 1.4826 median is
  // known to 2^-13, and with it the Huber class of all but a handful of pixels: |r| / sigma < 1.345 for EVERY scale in that band
  // (weight 1) or >= 1.345 for every one (weight 1.345 sigma / |r|: LINEAR in sigma).  So the 46 sums do not wait for the last
  // digit: its histogram pass also accumulates  S_in = sum_in J^T J  and  S_out = sum_out (1.345 / |r|) J^T J  (planes 0 / 1) and
  // lists the few pixels whose class the last 10 bits decide (region 4 of the histograms: ~N x 2^-13 x density of them); after the
  // ONE barrier that completes the last digit every workgroup forms  S_in + S_out / info_sqrt  and adds the listed pixels with the
  // exact per-pixel arithmetic.  An iteration is two device-wide synchronisations instead of three (three instead of four when
  // the first-digit speculation misses).  More than TL_AMB_CAP listed pixels (a degenerate residual distribution): the exact
  // form runs after all (one more barrier) -- also what split = 0 selects.  MEASURED: the exact form is the faster one on MI355X
  // (the second set of 45 contended shares, the doubled multiply-adds and the list cost more than the barrier saves): split is OFF
  // by default, kept as a tested switch (como_track_level_set_split, COMO_TRACK_SPLIT=1).
  int spec_bin = -1;
  while (true) {
    uint32_t* hs = hists2 + (it & 1) * 6 * SEL_BINS;          // this iteration's digit histograms / sums
    uint32_t* hother = hists2 + ((it + 1) & 1) * 6 * SEL_BINS;
    long long* sm = sums2 + (it & 1) * TL_SUM_PAR;
    long long* smother = sums2 + ((it + 1) & 1) * TL_SUM_PAR;
    uint32_t* amb = hs + 4 * SEL_BINS;                         // [0] count, entries from word 16
    TL_STAMP(0);
    // ---- phase A: warp, sample, residual, validity, digit-0 histogram ----
    for (int b = tid; b < SEL_BINS; b += 256) { lh[b] = 0; lh1[b] = 0; }
    T Pm[12];
    {
#pragma clang fp contract(off)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
          Pm[i * 4 + j] = dot3_seq(Kr[i * 3 + 0], Kr[i * 3 + 1], Kr[i * 3 + 2], Tc[0 * 4 + j], Tc[1 * 4 + j], Tc[2 * 4 + j]);
    }
    const T ea = exp(-a0), bb = a1;
    __syncthreads();
    T rk[TL_MAXP], j6[TL_MAXP];
    bool ok[TL_MAXP];
#pragma unroll
    for (int k = 0; k < TL_MAXP; ++k) {
      T hx, hy, hz;
      rigid_apply(Pm, pX[k], pY[k], pZ[k], hx, hy, hz);
      const T u = hx / hz, v = hy / hz;
      ok[k] = sel[k] && in_image(u, v, H, W) && (hz > T(0));
      Taps<T> t = make_taps(grid_position(u, W, ax), grid_position(v, H, ay), H, W);
      const T It = tap_sum(img + choff[k], t);
      const T tmp = ea * It;
      j6[k] = -tmp;
      rk[k] = (tmp + bb) - vref[k];
      if (ok[k]) {
        const KeyT key = abs_key(rk[k]);
        const uint32_t d0 = sel_digit<KeyT>(key, 0);
        atomicAdd(&lh[d0], 1u);
        if ((int)d0 == spec_bin) atomicAdd(&lh1[sel_digit<KeyT>(key, 1)], 1u);
      }
    }
    __syncthreads();
    TL_STAMP(1);
    flush(hs);
    if (spec_bin >= 0) flush(hs + 3 * SEL_BINS, lh1);
    TL_STAMP(2);
    if (!tl_barrier(B)) { alive = false; break; }
    TL_STAMP(3);
    {
      // the other parity's histograms / sums / list counter: free since the barrier above, used by the next iteration.
      // Cleared with read-modify-write atomics (performed at the memory side like the adds that follow): a device-scope
      // STORE may linger in this XCD's write-back L2 and land after other workgroups' atomic adds, wiping them
      uint32_t sink = 0;
      for (int e = bidx * 256 + tid; e < 4 * SEL_BINS + 1; e += G * 256) {
        if (xl) __hip_atomic_fetch_and(&hother[e], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else sink += __hip_atomic_fetch_and(&hother[e], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (bidx == G - 1) {
        for (int e = tid; e < TL_SUM_PAR; e += 256) {
          if (xl) __hip_atomic_fetch_and((unsigned long long*)&smother[e], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else sink += (uint32_t)__hip_atomic_fetch_and((unsigned long long*)&smother[e], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("" ::"v"(sink));
    }
    // ---- digits 0 and 1 of the exact median (keys come from registers: no memory pass) ----
    KeyT prefix = 0;
    uint32_t k_rem = 0, nv = 0;
    for (int b = tid; b < SEL_BINS; b += 256) lh[b] = 0;
    {
      // (the speculated second digit's counters are requested together with the first digit's: one round trip for both)
      uint32_t c0[8], c1[8];
      tl_load_hist(hs, c0);
      if (spec_bin >= 0) tl_load_hist(hs + 3 * SEL_BINS, c1);
      const uint32_t bin = tl_resolve_loaded(c0, k_rem, &nv, &sc);                         // (orders the lh clear)
      prefix |= ((KeyT)bin) << SelCfg<KeyT>::shift(0);
      const bool spec_hit = (int)bin == spec_bin;     // uniform over the grid: every workgroup resolves the same histogram
      spec_bin = (int)bin;
      if (!spec_hit) {                                // phase B: the second digit's histogram, one more synchronisation
#pragma unroll
        for (int k = 0; k < TL_MAXP; ++k) {
          const KeyT key = abs_key(rk[k]);
          if (ok[k] && sel_match<KeyT>(key, prefix, 1)) atomicAdd(&lh[sel_digit<KeyT>(key, 1)], 1u);
        }
        __syncthreads();
        TL_STAMP(4);
        flush(hs + SEL_BINS);
        TL_STAMP(5);
        if (!tl_barrier(B)) { alive = false; break; }
        TL_STAMP(6);
        for (int b = tid; b < SEL_BINS; b += 256) lh[b] = 0;
        tl_load_hist(hs + SEL_BINS, c1);
      }
      const uint32_t bin1 = tl_resolve_loaded(c1, k_rem, nullptr, &sc);
      prefix |= ((KeyT)bin1) << SelCfg<KeyT>::shift(1);
    }
    // ---- phase C: the last digit's histogram + (split) the band-split sums ----
    {
      // every value 1 / sigma can still take lies in [i_lo, i_hi] (monotone float arithmetic, widened by 2^-20 against a
      // division that is not correctly rounded): a pixel is classed only if both ends agree
      const T m_lo = key_value(prefix), m_hi = key_value(prefix | (KeyT)0x3ffu);
      const T i_lo = (T(1) / (T(1.4826) * m_hi)) * T(1.0 - 1.0 / 1048576.0), i_hi = (T(1) / (T(1.4826) * m_lo)) * T(1.0 + 1.0 / 1048576.0);
      tl_f2 acc2[48];
#pragma unroll
      for (int k = 0; k < 48; ++k) acc2[k] = tl_f2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < TL_MAXP; ++k) {
        if (!ok[k]) continue;
        const T r = rk[k];
        const KeyT key = abs_key(r);
        if (sel_match<KeyT>(key, prefix, 2)) atomicAdd(&lh[sel_digit<KeyT>(key, 2)], 1u);
        if (!split) continue;
        const T a = fabsf(r);
        const bool inl = a * i_hi < T(1.345), outl = a * i_lo >= T(1.345);     // (a non-finite product is neither)
        const T J[8] = {Jc[k][0], Jc[k][1], Jc[k][2], Jc[k][3], Jc[k][4], Jc[k][5], j6[k], Jc[k][6]};
        if (inl || outl) {
          const tl_f2 c = {inl ? T(1) : T(0), outl ? T(1.345) / a : T(0)};
          const tl_f2 r2 = {r, r};
          int q = 0;
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const tl_f2 wa = c * tl_f2{J[x], J[x]};
#pragma unroll
            for (int y = x; y < 8; ++y) { acc2[q] = wa * tl_f2{J[y], J[y]} + acc2[q]; ++q; }
            acc2[36 + x] = wa * r2 + acc2[36 + x];
          }
          acc2[44] = tl_f2{inl ? r * r : T(0), outl ? a : T(0)} + acc2[44];
        } else {
          const uint32_t slot = xl ? __hip_atomic_fetch_add(&amb[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                   : __hip_atomic_fetch_add(&amb[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (slot < (uint32_t)amb_cap) {
            unsigned long long* e = (unsigned long long*)(amb + 16 + slot * TL_AMB_WORDS);
            unsigned long long sink = 0;
#pragma unroll
            for (int w = 0; w < 5; ++w) {
              const T lo = w < 4 ? J[2 * w] : r, hi = w < 4 ? J[2 * w + 1] : T(0);
              const unsigned long long bits = (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32);
              if (xl) __hip_atomic_exchange(&e[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else sink += __hip_atomic_exchange(&e[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("" ::"v"(sink));
          }
        }
      }
      TL_STAMP(7);
      if (split) {
        float a96[96];
#pragma unroll
        for (int k = 0; k < 48; ++k) { a96[k] = acc2[k].x; a96[48 + k] = acc2[k].y; }
        float r0, r1;
        wave_reduce96(a96, lane, r0, r1);
        const int ix = wr_index(lane);
        redw[wv][ix] = r0;
        if (ix < 32) redw[wv][64 + ix] = r1;
      }
      __syncthreads();
      if (split && tid < 96 && (tid % 48) < TRK_ACC - 1) {
        const double v = (((double)redw[0][tid] + (double)redw[1][tid]) + (double)redw[2][tid]) + (double)redw[3][tid];
        share(sm, tid / 48, tid % 48, v);
      }
      TL_STAMP(8);
      flush(hs + 2 * SEL_BINS);
      TL_STAMP(9);
      if (!tl_barrier(B)) { alive = false; break; }
    }
    {
      const uint32_t bin = tl_resolve_digit(hs + 2 * SEL_BINS, k_rem, nullptr, &sc);
      prefix |= ((KeyT)bin) << SelCfg<KeyT>::shift(2);
    }
    const T sigma = T(1.4826) * key_value(prefix);
    const T info_sqrt = T(1) / sigma;
    const uint32_t namb = split ? ld_dev(&amb[0]) : 0u;       // (complete since the barrier; the same value in every workgroup)
    TL_STAMP(10);
    if (!split || namb > (uint32_t)amb_cap) {
      // ---- phase D, exact form: robust weights with the known scale, 8x8 system sums (plane 2), one more barrier ----
      // (accumulated in the same two-lane registers as the split pass -- lane y idle: a plain float[48] here was demoted to scratch)
      tl_f2 acc2[48];
#pragma unroll
      for (int k = 0; k < 48; ++k) acc2[k] = tl_f2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < TL_MAXP; ++k) {
        if (!ok[k]) continue;
        const T r = rk[k];
        const T wr = r * info_sqrt;
        const T w = huber(wr);
        const T J[8] = {Jc[k][0], Jc[k][1], Jc[k][2], Jc[k][3], Jc[k][4], Jc[k][5], j6[k], Jc[k][6]};
        const tl_f2 c = {w, T(0)};
        const tl_f2 r2 = {r, r};
        int q = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const tl_f2 wa = c * tl_f2{J[x], J[x]};
#pragma unroll
          for (int y = x; y < 8; ++y) { acc2[q] = wa * tl_f2{J[y], J[y]} + acc2[q]; ++q; }
          acc2[36 + x] = wa * r2 + acc2[36 + x];
        }
        acc2[44] = tl_f2{w * wr * wr, T(0)} + acc2[44];
      }
      float a48[48];
#pragma unroll
      for (int k = 0; k < 48; ++k) a48[k] = acc2[k].x;
      const float rr = wave_reduce48(a48, lane);
      redw[wv][wr_index(lane)] = rr;
      __syncthreads();
      if (tid < TRK_ACC - 1) {
        const double v = (((double)redw[0][tid] + (double)redw[1][tid]) + (double)redw[2][tid]) + (double)redw[3][tid];
        share(sm, 2, tid, v);
      }
      TL_STAMP(11);
      if (!tl_barrier(B)) { alive = false; break; }
      TL_STAMP(12);
      if (tid < TRK_ACC) tot[tid] = (ld_dev(&sm[TL_POISON]) != 0) ? __builtin_nan("") : (tid < TRK_ACC - 1 ? plane_total(sm, 2, tid) : 0.0);
    } else {
      // ---- the totals from the two planes + the listed pixels (every workgroup, identically) ----
      for (int e = tid; e < (int)namb * 5; e += 256) {
        const unsigned long long bits = ld_dev((const unsigned long long*)(amb + 16 + (e / 5) * TL_AMB_WORDS) + (e % 5));
        ambs[e / 5][2 * (e % 5)] = __uint_as_float((uint32_t)bits);
        ambs[e / 5][2 * (e % 5) + 1] = __uint_as_float((uint32_t)(bits >> 32));
      }
      __syncthreads();
      TL_STAMP(11);
      if (tid < TRK_ACC) {
        double v = 0.0;
        if (tid < TRK_ACC - 1) {
          const double s_in = plane_total(sm, 0, tid), s_out = plane_total(sm, 1, tid);
          const double info = (double)info_sqrt;
          v = tid < 44 ? s_in + s_out / info : s_in * info * info + s_out * 1.345 * info;
          // the pixels the last digit classes: the exact form's float arithmetic per pixel, summed as integers (order-free)
          long long hi = 0;
          unsigned long long lo = 0;
          bool bad = false;
          for (int e = 0; e < (int)namb; ++e) {
            const T r = ambs[e][8];
            const T wr = r * info_sqrt;
            const T w = huber(wr);
            T t;
            if (tid < 36) t = (w * ambs[e][qa]) * ambs[e][qb];
            else if (tid < 44) t = (w * ambs[e][tid - 36]) * r;
            else t = w * wr * wr;
            const double td = (double)t;
            if (!(fabs(td) < 4.0e18)) { bad = true; continue; }
            long long h1;
            unsigned long long l1;
            fix_split(td, h1, l1);
            hi += h1;
            lo += l1;
          }
          v += fix_value(hi, lo);
          if (bad) v = __builtin_nan("");
        }
        tot[tid] = (ld_dev(&sm[TL_POISON]) != 0) ? __builtin_nan("") : v;
      }
      TL_STAMP(12);
    }
    // ---- phase E (every workgroup, identically): 8x8 Cholesky, T <- T Exp(-delta), stop test ----
    __syncthreads();
    TL_STAMP(13);
    // (H and g of the record: one entry per lane of wave 1, beside the solve on lane 0 of wave 0 -- as a loop on that one lane,
    // behind the solve, the 80 dependent LDS round trips were 1.4 us of every iteration)
    if (bidx == 0 && wv == 1) {
      const int a = lane >> 3, b = lane & 7;
      rec_s[lane] = (T)tot[a <= b ? trk_q(a, b) : trk_q(b, a)];
      if (lane < 8) rec_s[64 + lane] = (T)tot[36 + lane];
    }
    if (tid == 0) {
      TrkSolve S;
      trk_solve8<T>(tot, Tc, S);
      TL_STAMP(14);
#pragma unroll
      for (int i = 0; i < 16; ++i) state[i] = (T)S.Tn[i];
      TL_STAMP(15);
      state[16] = (T)((double)a0 - S.d[6]);
      state[17] = (T)((double)a1 - S.d[7]);
      state[18] = (T)(tot[44] / (double)(nv / C));    // mean over valid pixels
      state[19] = (T)sqrt(S.gn);
      state[20] = (T)sqrt(S.dn);
      if (bidx == 0) {                                         // the record of this iteration, layout of como_track_iter_*
#pragma unroll
        for (int i = 0; i < 8; ++i) rec_s[72 + i] = (T)S.d[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) rec_s[80 + i] = state[i];
        rec_s[96] = state[16]; rec_s[97] = state[17];
        rec_s[98] = state[18]; rec_s[99] = state[19]; rec_s[100] = (T)tot[44];
        rec_s[101] = sigma; rec_s[102] = (T)(nv / C); rec_s[103] = state[20]; rec_s[104] = (T)S.info;
        rec_s[105] = (T)(it + 1);
      }
    }
    __syncthreads();
    if (bidx == 0 && tid < 106) out[tid] = rec_s[tid];
#pragma unroll
    for (int k = 0; k < 16; ++k) Tc[k] = state[k];
    a0 = state[16]; a1 = state[17];
    const T mse = state[18], gnorm = state[19], dnorm = state[20];
    TL_STAMP(16);
    ++it;
    // photo_tracking.py:166-180, float32 arithmetic as the reference's 0-dim tensors
    const T rel = fabsf((mse_prev - mse) / mse_prev);
    mse_prev = mse;
    __syncthreads();                                           // `state` / `tot` are rewritten next iteration
    if (it >= crit.max_iter || dnorm < crit.delta_norm || rel < crit.rel_tol || gnorm < crit.grad_norm) break;
  }
  if (!alive && bidx == 0 && tid == 0) out[104] = T(-1);        // a barrier timed out
  if (xl && bidx == 0 && tid == 0) {
    // the census is complete when its bytes add up to G (the other workgroups added theirs when they started, tens of
    // microseconds ago; returning memory-side atomics as the read: never a stale cached copy); bounded wait
    unsigned long long v = 0;
    unsigned total = 0, used = 0;
    for (int spin = 0; spin < 20000; ++spin) {
      v = __hip_atomic_fetch_add((unsigned long long*)(bar + TL_XCC_WORD), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      total = 0; used = 0;
      for (int x = 0; x < 8; ++x) { const unsigned c = (unsigned)((v >> (8 * x)) & 0xffull); total += c; used += c != 0; }
      if (total >= (unsigned)G) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (total != (unsigned)G || used != 1) out[104] = T(-2);    // not one XCD: every cross-workgroup value above is suspect
  }
}

// ---------------------------------- one pyramid level in ONE WORKGROUP (the coarsest level) ------------------------------------
// A level of at most TL1_CAP elements (80x60 gray = 4800) fits ONE workgroup: the pixel waves keep their elements in LDS planes +
// registers, the three digit histograms of the exact median live in LDS, the 46 sums go wave tree -> LDS -> one fixed sum over the
// waves -- every dependency point of an iteration is a __syncthreads.  Nothing crosses workgroups: no workspace, no atomics outside
// LDS, no placement assumption, no time-out.  The XCD-local form spends ~8 of its ~19 us per iteration at this size on exchanges
// (flush 1.2 + 0.6, barriers 0.7 + 2.9 + 1.7, counters fetched from L2: profiles/r6b_track_stamps.txt).
// WAVE SPECIALISATION: the last wave holds no pixels -- it resolves the median's digits from the LDS histograms (32 bins per lane, one
// wave scan), sums the waves' partials and runs the 8x8 solve + pose update (one lane, float64, > 128 live registers with the SE(3)
// exponential); the pixel waves run the same loop skeleton (tl1_loop<NT, true / false>: ONE source for both, so the barrier
// sequences agree by construction) without any of it, so their registers are not spilled around the solve.
// One compute unit has 64 float32 lanes: the level is bound by its INSTRUCTION count (4800 elements x ~200 per iteration), hence the
// packed (two-lane) multiply-adds of the sums and the one-wave digit resolution.
// Same arithmetic per pixel as track_level_kernel / the como_track_iter_* chain (photo_tracking.py:117-185); the per-thread partial
// sums follow this kernel's element -> thread assignment (element i -> pixel thread i % PT), a fixed order.
constexpr int TL1_CAP = 4800;       // elements (80x60 gray): 15 pixel waves x 64 lanes x 5
constexpr int TL1_PLANES = 7;       // X Y Z I_ref J0 J1 J2 in LDS (131 KB); J3 J4 J5 J7 in registers
struct Sel1Scratch { uint32_t bin[3]; uint32_t hit; };
struct TL1Shared {
  uint32_t *lh0, *lh1, *lh2;
  Sel1Scratch* sc;
  float* redw;         // [pixel waves][48], packed order (tl1_packed_pos)
  double* tot;         // [TRK_ACC]
  float* state;        // 16 T | 2 aff | mse | gnorm | dnorm
  float* rec_s;        // [112]
  float* planes;       // [TL1_PLANES][TL1_CAP]
};

// one digit of the k-th key from a COMPLETE LDS histogram of NB bins, by ONE wave: lane l owns bins [l NB / 64, (l + 1) NB / 64)
template <int NB>
__device__ __forceinline__ uint32_t tl1_resolve_wave(const uint32_t* lh, uint32_t& k_rem, uint32_t* total, int lane) {
  constexpr int PER = NB / 64;
  uint32_t c[PER], local = 0;
#pragma unroll
  for (int j = 0; j < PER; j += 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(&lh[lane * PER + j]);
    c[j] = v.x; c[j + 1] = v.y; c[j + 2] = v.z; c[j + 3] = v.w;
    local += (v.x + v.y) + (v.z + v.w);
  }
  uint32_t incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const uint32_t tot = __shfl(incl, 63, 64);
  if (total) { *total = tot; k_rem = tot ? (tot - 1) / 2 : 0; }          // first digit: lower median of all valid keys
  const uint32_t excl = incl - local;
  const bool hit = local > 0 && k_rem >= excl && k_rem < excl + local;
  uint32_t bin = 0, below = 0, run = excl;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (k_rem >= run && k_rem < run + c[j]) { bin = lane * PER + j; below = run; }
    run += c[j];
  }
  const unsigned long long m = __ballot(hit ? 1 : 0);
  const int src = m ? (int)__ffsll((long long)m) - 1 : 0;
  bin = m ? __shfl(bin, src, 64) : 0u;
  below = m ? __shfl(below, src, 64) : 0u;
  k_rem -= below;
  return bin;
}

// The 45 sums of one pixel in 23 two-lane accumulators (v_pk_fma_f32): rows of the upper triangle in pairs -- even rows start at an
// even column (pairs (J0,J1) (J2,J3) (J4,J5) (J6,J7)), odd rows at an odd one (pairs (J1,J2) (J3,J4) (J5,J6) (J7,r): the last entry
// is g[row]); the even rows' g entries and the error follow.  tl1_packed_pos(q) = position of sum q (trk_q order, 36 + a = g[a],
// 44 = error) in that layout.
__device__ __forceinline__ constexpr int tl1_row_base(int x) { return x == 0 ? 0 : x == 1 ? 4 : x == 2 ? 8 : x == 3 ? 11 : x == 4 ? 14 : x == 5 ? 16 : x == 6 ? 18 : 19; }
__device__ __forceinline__ int tl1_packed_pos(int q) {
  if (q == 44) return 44;
  if (q >= 36) { const int a = q - 36; return (a & 1) ? 2 * (tl1_row_base(a) + (8 - a) / 2) + 1 : 40 + a / 2; }
  int x = 0;
  while (x < 7 && q >= trk_q(x + 1, x + 1)) ++x;
  const int d = q - trk_q(x, x);                        // y - x
  return 2 * (tl1_row_base(x) + d / 2) + (d & 1);
}
__device__ __forceinline__ void tl1_accumulate_pk(tl_f2 (&acc)[23], float w, const float (&J)[8], float r, float wr) {
  const tl_f2 Je[4] = {{J[0], J[1]}, {J[2], J[3]}, {J[4], J[5]}, {J[6], J[7]}};
  const tl_f2 Jo[4] = {{J[1], J[2]}, {J[3], J[4]}, {J[5], J[6]}, {J[7], r}};
  float wa[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) wa[x] = w * J[x];
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    const tl_f2 w2 = {wa[x], wa[x]};
#pragma unroll
    for (int i = x / 2; i < 4; ++i) {
      const int slot = tl1_row_base(x) + (i - x / 2);
      acc[slot] = w2 * ((x & 1) ? Jo[i] : Je[i]) + acc[slot];
    }
  }
  const tl_f2 r2 = {r, r};
  acc[20] = tl_f2{wa[0], wa[2]} * r2 + acc[20];
  acc[21] = tl_f2{wa[4], wa[6]} * r2 + acc[21];
  acc[22] = tl_f2{w * wr, 0.f} * tl_f2{wr, 0.f} + acc[22];
}

#ifndef TL1_GROUP
#define TL1_GROUP 4
#endif
#ifdef COMO_TL_PROFILE
// tid 0 (a pixel wave) stamps slots 0..15, lane 0 of the solver wave slots 16..31 of iteration 3
#define TL1_STAMP(k) do { if (it == 3 && (PIX ? tid == 0 : lane == 0)) stamps[(PIX ? 0 : 16) + (k)] = (long long)wall_clock64(); } while (0)
#else
#define TL1_STAMP(k) do { } while (0)
#endif
// The level's loop, ONE definition for the pixel waves (PIX) and the solver wave (!PIX): every __syncthreads below is executed by
// both instantiations the same number of times (all branch conditions around them are uniform over the workgroup).
template <int NT, bool PIX>
__device__ __forceinline__ void tl1_loop(const TL1Shared sh, const float* __restrict__ Kmat, const float* __restrict__ P,
                                         const float* __restrict__ vals_i, const float* __restrict__ img, int H, int W, int NE,
                                         const float* __restrict__ J8, const uint8_t* __restrict__ in_mask, TLCriteria crit,
                                         float* __restrict__ out, int C, long long* __restrict__ stamps) {
  using T = float;
  using KeyT = uint32_t;
  constexpr int PW = NT / 64 - 1, PT = PW * 64;                  // pixel waves / threads
  constexpr int MAXP = PIX ? (TL1_CAP + PT - 1) / PT : 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int HW = H * W;
  float* const pXs = sh.planes;
  float* const pYs = pXs + TL1_CAP;
  float* const pZs = pYs + TL1_CAP;
  float* const vrs = pZs + TL1_CAP;
  float* const j0s = vrs + TL1_CAP;
  float* const j1s = j0s + TL1_CAP;
  float* const j2s = j1s + TL1_CAP;

  T Jc[MAXP][4];
  bool sel[MAXP];
  int mypos = 0;                                                 // solver lane k < 45: where sum k sits in the packed order
  if (PIX) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int i = k * PT + tid;
      sel[k] = i < NE;
      const int ic = sel[k] ? i : 0;
      const int ip = C == 1 ? ic : (int)((unsigned)ic / (unsigned)C);
      const float4 a = *reinterpret_cast<const float4*>(&J8[8 * ic]);
      const float4 b = *reinterpret_cast<const float4*>(&J8[8 * ic + 4]);
      if (sel[k]) {
        pXs[i] = P[3 * ip]; pYs[i] = P[3 * ip + 1]; pZs[i] = P[3 * ip + 2];
        vrs[i] = vals_i[ic];
        j0s[i] = a.x; j1s[i] = a.y; j2s[i] = a.z;
      }
      Jc[k][0] = a.w; Jc[k][1] = b.x; Jc[k][2] = b.y; Jc[k][3] = b.w;     // J3 J4 J5 J7 (J6 = -exp(-a) I is per iteration)
      if (sel[k] && in_mask) sel[k] = in_mask[ip] != 0;
    }
  } else {
    mypos = tl1_packed_pos(lane < TRK_ACC - 1 ? lane : 0);
  }
  const T ax = T(1) / T(W), ay = T(1) / T(H);
  T mse_prev = __builtin_inff();
  int it = 0, spec_bin = -1;
  uint32_t k_rem = 0, nv = 0;                                    // (carried by the solver wave)

  while (true) {
    // ---- warp, sample, residual, validity; first digit's histogram (+ the second digit's, speculated: see track_level_kernel) ----
    TL1_STAMP(0);
    for (int b = tid; b < SEL_BINS; b += NT) { sh.lh0[b] = 0; sh.lh1[b] = 0; sh.lh2[b] = 0; }
    T Pm[12];
    T ea = T(0), bb = T(0);
    if (PIX) {
      {
#pragma clang fp contract(off)
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 4; ++j)
            Pm[i * 4 + j] = dot3_seq(Kmat[i * 3 + 0], Kmat[i * 3 + 1], Kmat[i * 3 + 2], sh.state[0 * 4 + j], sh.state[1 * 4 + j],
                                     sh.state[2 * 4 + j]);
      }
      ea = exp(-sh.state[16]);
      bb = sh.state[17];
    }
    __syncthreads();
    TL1_STAMP(1);
    T rk[MAXP], j6[MAXP];
    bool ok[MAXP];
    if (PIX) {
      // (straight-line over the thread's elements: their image gathers overlap; the histogram adds -- divergent -- come after)
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int i = k * PT + tid;
        const int il = i < NE ? i : 0;           // (slots beyond the level are never written: read element 0, discarded by sel)
        T hx, hy, hz;
        rigid_apply(Pm, pXs[il], pYs[il], pZs[il], hx, hy, hz);
        const T u = hx / hz, v = hy / hz;
        ok[k] = sel[k] && in_image(u, v, H, W) && (hz > T(0));
        Taps<T> t = make_taps(grid_position(u, W, ax), grid_position(v, H, ay), H, W);
        const int choff = C == 1 ? 0 : (int)((unsigned)il % (unsigned)C) * HW;
        const T It = tap_sum(img + choff, t);
        const T tmp = ea * It;
        j6[k] = -tmp;
        rk[k] = (tmp + bb) - vrs[il];
        if ((k % TL1_GROUP) == TL1_GROUP - 1) __builtin_amdgcn_sched_barrier(0);    // (bounds the gathers in flight: registers)
      }
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        if (ok[k]) {
          const KeyT key = abs_key(rk[k]);
          const uint32_t d0 = sel_digit<KeyT>(key, 0);
          atomicAdd(&sh.lh0[d0], 1u);
          if ((int)d0 == spec_bin) atomicAdd(&sh.lh1[sel_digit<KeyT>(key, 1)], 1u);
        }
      }
    }
    TL1_STAMP(2);
    __syncthreads();
    TL1_STAMP(3);
    // ---- exact lower median of |r| over the valid elements: three digits, histograms in LDS, resolved by the solver wave ----
    KeyT prefix = 0;
    if (!PIX) {
      const uint32_t bin = tl1_resolve_wave<SEL_BINS>(sh.lh0, k_rem, &nv, lane);
      const bool hit = (int)bin == spec_bin;
      uint32_t bin1 = 0;
      if (hit) bin1 = tl1_resolve_wave<SEL_BINS>(sh.lh1, k_rem, nullptr, lane);       // (the speculated histogram is complete)
      if (lane == 0) { sh.sc->bin[0] = bin; sh.sc->bin[1] = bin1; sh.sc->hit = hit ? 1u : 0u; }
    }
    __syncthreads();
    TL1_STAMP(4);
    {
      const uint32_t bin = sh.sc->bin[0];
      prefix |= ((KeyT)bin) << SelCfg<KeyT>::shift(0);
      const bool spec_hit = sh.sc->hit != 0;
      const bool spec_dirty = spec_bin >= 0;
      spec_bin = (int)bin;
      if (!spec_hit) {
        if (spec_dirty) {                                  // counts of the wrong first digit
          for (int b = tid; b < SEL_BINS; b += NT) sh.lh1[b] = 0;
          __syncthreads();
        }
        if (PIX) {
#pragma unroll
          for (int k = 0; k < MAXP; ++k) {
            const KeyT key = abs_key(rk[k]);
            if (ok[k] && sel_match<KeyT>(key, prefix, 1)) atomicAdd(&sh.lh1[sel_digit<KeyT>(key, 1)], 1u);
          }
        }
        __syncthreads();
        if (!PIX) {
          const uint32_t bin1 = tl1_resolve_wave<SEL_BINS>(sh.lh1, k_rem, nullptr, lane);
          if (lane == 0) sh.sc->bin[1] = bin1;
        }
        __syncthreads();
      }
      prefix |= ((KeyT)sh.sc->bin[1]) << SelCfg<KeyT>::shift(1);
      TL1_STAMP(5);
      if (PIX) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
          const KeyT key = abs_key(rk[k]);
          if (ok[k] && sel_match<KeyT>(key, prefix, 2)) atomicAdd(&sh.lh2[sel_digit<KeyT>(key, 2)], 1u);
        }
      }
      __syncthreads();
      TL1_STAMP(6);
      if (!PIX) {
        const uint32_t bin2 = tl1_resolve_wave<SEL_BINS / 2>(sh.lh2, k_rem, nullptr, lane);      // (10 bits)
        if (lane == 0) sh.sc->bin[2] = bin2;
      }
      __syncthreads();
      prefix |= ((KeyT)sh.sc->bin[2]) << SelCfg<KeyT>::shift(2);
      TL1_STAMP(7);
    }
    const T sigma = T(1.4826) * key_value(prefix);
    const T info_sqrt = T(1) / sigma;
    // ---- robust weights, the 8x8 system's sums: 23 two-lane accumulators, one wave tree ----
    if (PIX) {
      tl_f2 acc[23];
#pragma unroll
      for (int q = 0; q < 23; ++q) acc[q] = tl_f2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int i = k * PT + tid;
        const int il = i < NE ? i : 0;
        // (an invalid element contributes exactly zero, also when it carries non-finite J / r: selected away, not multiplied)
        const bool o = ok[k];
        const T r = o ? rk[k] : T(0), wr = r * info_sqrt, w = huber(wr);
        const T J[8] = {o ? j0s[il] : T(0), o ? j1s[il] : T(0), o ? j2s[il] : T(0), o ? Jc[k][0] : T(0), o ? Jc[k][1] : T(0),
                        o ? Jc[k][2] : T(0), o ? j6[k] : T(0), o ? Jc[k][3] : T(0)};
        tl1_accumulate_pk(acc, w, J, r, wr);
      }
      TL1_STAMP(8);
      float a48[48];
#pragma unroll
      for (int q = 0; q < 23; ++q) { a48[2 * q] = acc[q].x; a48[2 * q + 1] = acc[q].y; }
      a48[46] = 0.f; a48[47] = 0.f;
      const float rr = wave_reduce48(a48, lane);
      if (wr_index(lane) < 48) sh.redw[wv * 48 + wr_index(lane)] = rr;       // (64 lanes, 48 values: the rest is padding)
      TL1_STAMP(9);
    }
    __syncthreads();
    TL1_STAMP(10);
    // ---- the solver wave: totals, 8x8 Cholesky, T <- T Exp(-delta), the iteration's record ----
    if (!PIX) {
      // fixed-order sum over the pixel waves; a non-finite sum poisons all of them (as the fixed-point shares of
      // track_level_kernel: NaN totals -> info != 0)
      double v = 0.0;
      if (lane < TRK_ACC - 1) {
#pragma unroll
        for (int w = 0; w < PW; ++w) v += (double)sh.redw[w * 48 + mypos];
      }
      const bool bad = __any((lane < TRK_ACC - 1 && !(fabs(v) < 4.0e18)) ? 1 : 0) != 0;
      if (bad) v = __builtin_nan("");
      if (lane < TRK_ACC) sh.tot[lane] = lane < TRK_ACC - 1 ? v : 0.0;
      // (LDS operations of one wave complete in order: lane 0 below reads what the other lanes have just written)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      TL1_STAMP(11);
      // H (symmetric) and g of the record, one entry per lane
      {
        const int a = lane >> 3, b = lane & 7;
        sh.rec_s[lane] = (T)sh.tot[a <= b ? trk_q(a, b) : trk_q(b, a)];
        if (lane < 8) sh.rec_s[64 + lane] = (T)sh.tot[36 + lane];
      }
      if (lane == 0) {
        const T a0 = sh.state[16], a1 = sh.state[17];
        TrkSolve S;
        trk_solve8<T>(sh.tot, sh.state, S);     // (reads the pose from LDS; `state` is rewritten from S below)
        TL1_STAMP(14);
#pragma unroll
        for (int i = 0; i < 16; ++i) { const T t = (T)S.Tn[i]; sh.state[i] = t; sh.rec_s[80 + i] = t; }
        const T na0 = (T)((double)a0 - S.d[6]), na1 = (T)((double)a1 - S.d[7]);
        const T mse = (T)(sh.tot[44] / (double)(nv / C));         // mean over valid pixels
        const T gno = (T)sqrt(S.gn), dno = (T)sqrt(S.dn);
        sh.state[16] = na0; sh.state[17] = na1; sh.state[18] = mse; sh.state[19] = gno; sh.state[20] = dno;
#pragma unroll
        for (int i = 0; i < 8; ++i) sh.rec_s[72 + i] = (T)S.d[i];
        sh.rec_s[96] = na0; sh.rec_s[97] = na1;
        sh.rec_s[98] = mse; sh.rec_s[99] = gno; sh.rec_s[100] = (T)sh.tot[44];
        sh.rec_s[101] = sigma; sh.rec_s[102] = (T)(nv / C); sh.rec_s[103] = dno; sh.rec_s[104] = (T)S.info;
        sh.rec_s[105] = (T)(it + 1);
        TL1_STAMP(15);
      }
    }
    __syncthreads();
    TL1_STAMP(12);
    if (tid < 106) out[tid] = sh.rec_s[tid];
    const T mse = sh.state[18], gnorm = sh.state[19], dnorm = sh.state[20];
    ++it;
    // stop test: photo_tracking.py:166-180, float32 arithmetic as the reference's 0-dim tensors (every thread, identically)
    const T rel = fabsf((mse_prev - mse) / mse_prev);
    mse_prev = mse;
#ifdef COMO_TL_PROFILE
    if (it == 4 && (PIX ? tid == 0 : lane == 0)) stamps[(PIX ? 0 : 16) + 13] = (long long)wall_clock64();
#endif
    if (it >= crit.max_iter || dnorm < crit.delta_norm || rel < crit.rel_tol || gnorm < crit.grad_norm) break;
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void track_level_one_kernel(
    const float* __restrict__ Tji_init, const float* __restrict__ Kmat, const float* __restrict__ aff_init,
    const float* __restrict__ P, const float* __restrict__ vals_i, const float* __restrict__ img, int H, int W, long N,
    const float* __restrict__ J8, const uint8_t* __restrict__ in_mask, TLCriteria crit, float* __restrict__ out, int C,
    long long* __restrict__ stamps) {
  constexpr int NWV = NT / 64;
  // (keys are |r| bit patterns: the sign bit is clear, so the first digit uses 1024 of its 2048 bins, the last digit has 10 bits)
  __shared__ __attribute__((aligned(16))) uint32_t lh0[SEL_BINS];
  __shared__ __attribute__((aligned(16))) uint32_t lh1[SEL_BINS];
  __shared__ __attribute__((aligned(16))) uint32_t lh2[SEL_BINS];
  __shared__ Sel1Scratch sc;
  __shared__ float redw[(NWV - 1) * 48];
  __shared__ double tot[TRK_ACC];
  __shared__ float state[24];
  __shared__ float rec_s[112];
  extern __shared__ __attribute__((aligned(16))) unsigned char tl1_dyn[];
  const int tid = threadIdx.x;
  if (tid < 16) state[tid] = Tji_init[tid];
  else if (tid < 18) state[tid] = aff_init[tid - 16];
  // the record starts as "no iteration done" (as track_level_kernel)
  if (tid < 16) out[80 + tid] = Tji_init[tid];
  else if (tid < 18) out[96 + (tid - 16)] = aff_init[tid - 16];
  else if (tid == 18) out[104] = 0.f;
  else if (tid == 19) out[105] = 0.f;
  __syncthreads();
  const TL1Shared sh{lh0, lh1, lh2, &sc, redw, tot, state, rec_s, reinterpret_cast<float*>(tl1_dyn)};
  const int NE = (int)N * C;
  if ((tid >> 6) == NWV - 1) tl1_loop<NT, false>(sh, Kmat, P, vals_i, img, H, W, NE, J8, in_mask, crit, out, C, stamps);
  else tl1_loop<NT, true>(sh, Kmat, P, vals_i, img, H, W, NE, J8, in_mask, crit, out, C, stamps);
}

__global__ void xcc_probe_kernel(int* __restrict__ xcc) {
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = (int)(id & 0xf);
  }
}

}  // namespace como

static int g_track_local_enabled = 1;   // como_track_level_set_local
static int g_track_local_debug = 0;     // como_track_level_debug_mismatch
static int g_track_split = -1;          // como_track_level_set_split (-1: COMO_TRACK_SPLIT, default off)
static int g_track_amb_cap = como::TL_AMB_CAP;   // como_track_level_debug_amb_cap
static int g_track_one = -1;            // como_track_level_set_one (-1: COMO_TRACK_ONE, default 0 = off; 512 / 768 / 1024 = threads of the workgroup)

extern "C" {

long como_track_level_workspace_bytes(void);
int como_track_level_probe(void);

/* Optional: a workspace in UNCACHED device memory (hipDeviceMallocUncached: not held in the XCD-private L2s), created once
 * outside any stream capture.  NULL if the runtime refuses. */
void* como_track_level_workspace_create(void) {
  void* p = nullptr;
  const size_t bytes = (size_t)como_track_level_workspace_bytes();
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  return p;
}
void como_track_level_workspace_destroy(void* p) { if (p) (void)hipFree(p); }

long como_track_level_workspace_bytes(void) {
  (void)como_track_level_probe();                            // (once per process, outside any stream capture)
  return (long)como::TL_BAR_WORDS * 4 + 2L * 6 * como::SEL_BINS * 4 + (long)como::TL_SUM_WORDS * 8 + 32 * 8;   // (+ profile stamps)
}

/* Does the dispatcher place workgroup b of a launch on XCD b % 8 (and are there 8 of them)?  Probed once per process with a tiny
 * kernel (outside any stream capture: como_track_level_workspace_bytes calls it); the XCD-local form of the level kernel relies on it. */
int como_track_level_probe(void) {
  static int ok = -1;
  if (ok >= 0) return ok;
  ok = 0;
  const char* e = getenv("COMO_TRACK_LOCAL");                // COMO_TRACK_LOCAL=0: the device-wide form for every level (A/B)
  if (e && atoi(e) == 0) return ok;
  int* d = nullptr;
  if (hipMalloc((void**)&d, 64 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return ok; }
  hipLaunchKernelGGL(como::xcc_probe_kernel, dim3(64), dim3(64), 0, 0, d);
  int h[64];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
    bool good = true;
    unsigned seen = 0;
    for (int b = 0; b < 64; ++b) good = good && h[b] == h[b & 7];
    for (int b = 0; b < 8; ++b) seen |= 1u << (h[b] & 15);
    ok = (good && __builtin_popcount(seen) == 8) ? 1 : 0;
  }
  (void)hipGetLastError();
  (void)hipFree(d);
  return ok;
}

/* local_workspace (optional): a second workspace of como_track_level_workspace_bytes() bytes in ORDINARY (L2-cacheable) device
 * memory: levels of at most 32 x 256 x TL_MAXP elements then run XCD-local (see track_level_kernel) when the probe allows it. */
static int track_level_launch(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                              const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                              const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                              void* workspace, int workspace_uncached, void* local_workspace, float* out, como_stream_t stream,
                              bool prezeroed) {
  using namespace como;
  if (!Tji_init || !K || !aff_init || !P || !vals_i || !img || !J8 || !workspace || !out || N <= 0 || H < 3 || W < 3 ||
      max_iter < 1 || channels < 1 || channels > 4)
    return COMO_ERR_ARG;
  const long NE = N * channels;
  hipStream_t s = (hipStream_t)stream;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return COMO_ERR_LAUNCH;
    ncu = prop.multiProcessorCount;
  }
  if (g_track_one < 0) {
    // COMO_TRACK_ONE=512 / 768 / 1024: levels of at most 4800 elements in ONE workgroup of that many threads.  Default OFF --
    // measured on MI355X at 80x60: 16.6 / 17.3 / 17.7 us per iteration against 17.4 for the XCD-local form on 19 compute units
    // (one compute unit's 64 float32 lanes are instruction-bound at ~250 instructions per element: profiles/r6c_track_one.txt)
    const char* e = getenv("COMO_TRACK_ONE");
    g_track_one = e ? atoi(e) : 0;
    if (g_track_one != 0 && g_track_one != 512 && g_track_one != 768) g_track_one = 1024;
  }
  if (g_track_one && NE <= TL1_CAP) {
    // the coarsest level: ONE workgroup, everything in registers + LDS (no workspace, nothing to clear)
    TLCriteria crit1{max_iter, delta_norm, rel_tol, grad_norm};
    long long* stamps1 = (long long*)((uint32_t*)workspace + TL_BAR_WORDS + 2 * 6 * SEL_BINS) + TL_SUM_WORDS;   // (-DCOMO_TL_PROFILE only)
    constexpr unsigned dyn = (unsigned)(TL1_PLANES * TL1_CAP * sizeof(float));   // 133 KB of planes (static LDS: 28 KB)
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)track_level_one_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess ||
          hipFuncSetAttribute((const void*)track_level_one_kernel<768>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess ||
          hipFuncSetAttribute((const void*)track_level_one_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess)
        return COMO_ERR_LAUNCH;
      attr = true;
    }
    if (g_track_one == 768)
      hipLaunchKernelGGL(track_level_one_kernel<768>, dim3(1), dim3(768), dyn, s, Tji_init, K, aff_init, P, vals_i, img, H, W, N, J8,
                         in_mask, crit1, out, channels, stamps1);
    else if (g_track_one == 512)
      hipLaunchKernelGGL(track_level_one_kernel<512>, dim3(1), dim3(512), dyn, s, Tji_init, K, aff_init, P, vals_i, img, H, W, N, J8,
                         in_mask, crit1, out, channels, stamps1);
    else
      hipLaunchKernelGGL(track_level_one_kernel<1024>, dim3(1), dim3(1024), dyn, s, Tji_init, K, aff_init, P, vals_i, img, H, W, N, J8,
                         in_mask, crit1, out, channels, stamps1);
    COMO_CHECK_LAUNCH();
    return COMO_OK;
  }
  long G = (NE + 255) / 256;
  const bool local = local_workspace && NE <= 32L * 256 * TL_MAXP && ncu >= 256 && como_track_level_probe() == 1 &&
                     g_track_local_enabled;
  if (local) {
    if (G > 32) G = 32;                       // the compute units of one XCD: all co-resident
    workspace = local_workspace;
    workspace_uncached = 0;
  } else {
    if (G > ncu) G = ncu;                     // one workgroup per compute unit: all co-resident (the barrier needs that)
    if (G > 512) G = 512;
  }
  const long ppt = (NE + G * 256 - 1) / (G * 256);
  if (ppt > TL_MAXP) return COMO_ERR_ARG;     // larger than TL_MAXP x 256 x #CU pixels: use the como_track_iter_* chain
  unsigned* bar = (unsigned*)workspace;
  uint32_t* hists2 = (uint32_t*)workspace + TL_BAR_WORDS;
  long long* sums2 = (long long*)(hists2 + 2 * 6 * SEL_BINS);
  long long* stamps = sums2 + TL_SUM_WORDS;
  if (!prezeroed && !zero_words(workspace, TL_BAR_WORDS + 2 * 6 * SEL_BINS + 2 * TL_SUM_WORDS, s)) return COMO_ERR_LAUNCH;
  TLCriteria crit{max_iter, delta_norm, rel_tol, grad_norm};
  if (g_track_split < 0) {
    // (default OFF: measured slower than the exact form at every level -- one barrier less, but 90 instead of 45 contended
    // fixed-point shares per workgroup, twice the multiply-adds and the list: 34.5 / 30.6 / 20.8 / 18.7 us per iteration against
    // 33.5 / 30.5 / 18.8 / 17.7 at 640x480 .. 80x60)
    const char* e = getenv("COMO_TRACK_SPLIT");
    g_track_split = (e && atoi(e) != 0) ? 1 : 0;
  }
  hipLaunchKernelGGL(track_level_kernel, dim3((unsigned)(local ? 8 * G : G)), dim3(256), 0, s, Tji_init, K, aff_init, P, vals_i, img,
                     H, W, N, J8, in_mask, crit, bar, hists2, sums2, stamps, out, (int)ppt, workspace_uncached ? 0 : 1, channels,
                     local ? (1 | (g_track_local_debug ? 2 : 0)) : 0, g_track_split, g_track_amb_cap);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_track_level_local_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                               const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                               const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                               void* workspace, int workspace_uncached, void* local_workspace, float* out, como_stream_t stream) {
  return track_level_launch(Tji_init, K, aff_init, P, vals_i, img, H, W, N, channels, J8, in_mask, max_iter, delta_norm, rel_tol,
                            grad_norm, workspace, workspace_uncached, local_workspace, out, stream, false);
}

/* The same launch WITHOUT the clear of the barrier workspace: the caller has zeroed the first como_track_level_zero_bytes() bytes of
 * `workspace` AND of `local_workspace` on this stream since their last use (the tracker's frame graph clears the workspaces of its
 * three levels inside its first launch, como_track_frame_pyramid3_f32: a clear of its own is a dependent ~4.7 us launch per level). */
int como_track_level_prezeroed_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                                   const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                                   const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                                   void* workspace, int workspace_uncached, void* local_workspace, float* out, como_stream_t stream) {
  return track_level_launch(Tji_init, K, aff_init, P, vals_i, img, H, W, N, channels, J8, in_mask, max_iter, delta_norm, rel_tol,
                            grad_norm, workspace, workspace_uncached, local_workspace, out, stream, true);
}
long como_track_level_zero_bytes(void) {
  return ((long)como::TL_BAR_WORDS + 2L * 6 * como::SEL_BINS + 2L * como::TL_SUM_WORDS) * 4;
}

/* The band-split form of the level kernel's sums (two device-wide synchronisations per iteration, see track_level_kernel) can be
 * switched off: every iteration then resolves the robust scale first and sums afterwards, the exact form (A/B, tests; also
 * COMO_TRACK_SPLIT=0).  Returns the previous setting. */
/* Test switch: the capacity of the band-pixel list (0 .. TL_AMB_CAP = 128; negative restores the default) -- a small capacity makes the
 * overflow path (exact form after the split sums, one more barrier) run on ordinary data. */
void como_track_level_debug_amb_cap(int cap) {
  g_track_amb_cap = (cap < 0 || cap > como::TL_AMB_CAP) ? como::TL_AMB_CAP : cap;
}
int como_track_level_set_split(int enable) {
  const int prev = g_track_split;
  g_track_split = enable ? 1 : 0;
  return prev < 0 ? 0 : prev;
}

/* Levels of at most 4800 elements (80x60 gray) can run in ONE workgroup (track_level_one_kernel: LDS histograms, __syncthreads
 * only): como_track_level_set_one(512 / 768 / 1024) picks that form and its workgroup size, 0 (the default: measured no faster)
 * the multi-workgroup forms (A/B, tests; also COMO_TRACK_ONE); returns the previous setting. */
int como_track_level_set_one(int threads) {
  const int prev = g_track_one < 0 ? 0 : g_track_one;
  g_track_one = (threads == 0 || threads == 512 || threads == 768) ? threads : 1024;
  return prev;
}

/* The XCD-local form can be switched off for the rest of the process (the host does so when a level reports status -2 or a
 * time-out); returns the previous setting.  como_track_level_local_state: 1 = the coarse levels run XCD-local. */
int como_track_level_set_local(int enable) {
  const int prev = g_track_local_enabled;
  g_track_local_enabled = enable ? 1 : 0;
  return prev;
}
int como_track_level_local_state(void) { return (g_track_local_enabled && como_track_level_probe() == 1) ? 1 : 0; }
/* Test switch: workgroup 1 of an XCD-local launch reports a neighbouring XCD in the census -> status -2. */
void como_track_level_debug_mismatch(int on) { g_track_local_debug = on ? 1 : 0; }

int como_track_level_channels_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                                  const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                                  const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                                  void* workspace, int workspace_uncached, float* out, como_stream_t stream) {
  return como_track_level_local_f32(Tji_init, K, aff_init, P, vals_i, img, H, W, N, channels, J8, in_mask, max_iter, delta_norm,
                                    rel_tol, grad_norm, workspace, workspace_uncached, nullptr, out, stream);
}

int como_track_level_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P, const float* vals_i,
                         const float* img, int H, int W, long N, const float* J8, const uint8_t* in_mask, int max_iter,
                         float delta_norm, float rel_tol, float grad_norm, void* workspace, int workspace_uncached, float* out,
                         como_stream_t stream) {
  return como_track_level_channels_f32(Tji_init, K, aff_init, P, vals_i, img, H, W, N, 1, J8, in_mask, max_iter, delta_norm,
                                       rel_tol, grad_norm, workspace, workspace_uncached, out, stream);
}

int como_track_iter_channels_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                                 const float* img, int H, int W, long N, int channels, float* J8, float* r_ws,
                                 uint8_t* valid_out, float* pj_out, float* depth_out, void* hists, void* partials, float* out,
                                 const uint8_t* in_mask, como_stream_t stream) {
  return como::track_iter<float>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                 (double*)partials, out, in_mask, (hipStream_t)stream, channels);
}

int como_track_iter_channels_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                                 const double* img, int H, int W, long N, int channels, double* J8, double* r_ws,
                                 uint8_t* valid_out, double* pj_out, double* depth_out, void* hists, void* partials,
                                 double* out, const uint8_t* in_mask, como_stream_t stream) {
  return como::track_iter<double>(Tji, K, aff, P, vals_i, img, H, W, N, J8, r_ws, valid_out, pj_out, depth_out, hists,
                                  (double*)partials, out, in_mask, (hipStream_t)stream, channels);
}

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
