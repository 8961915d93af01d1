// DepthCov native ops: the two functions the reference exports from its `como_backends` extension.
//
//   cross_covariance       como/backend/src/cov_gpu.cu:17-84  (CPU twin src/cov_cpu.cpp:17-64)
//   get_new_chol_obs_info  como/backend/src/cov_gpu.cu:132-215 (CPU twin src/cov_cpu.cpp:66-85)
//
// Numerics follow the reference's device code: safe_sqrt / matern are `float` functions even when the
// kernel is dispatched for double (include/kernel_functions.h:5-14), and the 1e-8 sits inside
// sqrt(1/det + 1e-8) (cov_gpu.cu:51) -- unlike the Python twin (depth_cov/core/kernels.py:55-66).
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

__device__ __forceinline__ float ref_safe_sqrt(float x) { return (float)sqrt((double)x + 1e-8); }
__device__ __forceinline__ float ref_matern(float Q) {
  const float tmp = (float)(1.73205080757 * (double)ref_safe_sqrt(Q));
  return (1.0f + tmp) * expf(-tmp);
}


// k(x_a, E_a; x_b, E_b) * scale with exactly the float arithmetic of cross_cov_kernel<float> (E row-major 2x2)
__device__ __forceinline__ float cov_value_f32(float xa0, float xa1, const float* Ea, float xb0, float xb1, const float* Eb,
                                               float scale) {
  const float a00 = Ea[0], a01 = Ea[1], a10 = Ea[2], a11 = Ea[3];
  const float b00 = Eb[0], b01 = Eb[1], b10 = Eb[2], b11 = Eb[3];
  const float dx = xa0 - xb0;
  const float dy = xa1 - xb1;
  const float e00 = a00 + b00, e01 = a01 + b01, e11 = a11 + b11;
  const float det_inv = (float)(1.0 / (double)(e00 * e11 - e01 * e01));
  float Q = (e11 * dx * dx) - 2.f * (e01 * dx * dy) + (e00 * dy * dy);
  Q = (float)((double)Q * (0.5 * (double)det_inv));
  const float d1 = a00 * a11 - a01 * a10;
  const float d2 = b00 * b11 - b01 * b10;
  const float pw = powf(d1 * d2, 0.25f);
  const float C = (float)(2.0 * (double)pw * (double)ref_safe_sqrt(det_inv));
  return scale * C * ref_matern(Q);
}

struct CovStrides {
  long x1[3], E1[4], x2[3], E2[4];
};

template <typename T>
__global__ __launch_bounds__(256) void cross_cov_kernel(const T* __restrict__ x1, const T* __restrict__ E1,
                                                        const T* __restrict__ x2, const T* __restrict__ E2, T scale,
                                                        T* __restrict__ K12, int N, int M, CovStrides st) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)N * M) return;
  const int i = (int)(p / M), j = (int)(p % M);
  const T* xa = x1 + b * st.x1[0] + i * st.x1[1];
  const T* xb = x2 + b * st.x2[0] + j * st.x2[1];
  const T* Ea = E1 + b * st.E1[0] + i * st.E1[1];
  const T* Eb = E2 + b * st.E2[0] + j * st.E2[1];
  const T a00 = Ea[0], a01 = Ea[st.E1[3]], a10 = Ea[st.E1[2]], a11 = Ea[st.E1[2] + st.E1[3]];
  const T b00 = Eb[0], b01 = Eb[st.E2[3]], b10 = Eb[st.E2[2]], b11 = Eb[st.E2[2] + st.E2[3]];
  const T dx = xa[0] - xb[0];
  const T dy = xa[st.x1[2]] - xb[st.x2[2]];
  const T e00 = a00 + b00, e01 = a01 + b01, e11 = a11 + b11;
  const T det_inv = (T)(1.0 / (double)(e00 * e11 - e01 * e01));
  T Q = (e11 * dx * dx) - T(2) * (e01 * dx * dy) + (e00 * dy * dy);
  Q = (T)((double)Q * (0.5 * (double)det_inv));
  const T d1 = a00 * a11 - a01 * a10;
  const T d2 = b00 * b11 - b01 * b10;
  T pw;
  if constexpr (sizeof(T) == 8) pw = pow(d1 * d2, 0.25);
  else pw = (T)powf((float)(d1 * d2), 0.25f);              // float, and half (c10::Half pow = float powf rounded to half)
  const T C = (T)(2.0 * (double)pw * (double)ref_safe_sqrt((float)det_inv));
  K12[((long)b * N + i) * M + j] = scale * C * (T)ref_matern((float)Q);
}

// One wave per batch element: forward substitution for the new Cholesky row (cov_gpu.cu:132-160).
__global__ __launch_bounds__(64) void chol_row_kernel(float* __restrict__ L, const float* __restrict__ k_ni, float k_ii,
                                                      int n, int N) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* Lb = L + (long)b * n * n;
  float sum = (lane < N) ? k_ni[(long)b * N + lane] : 0.f;
  float sumsq = 0.f;
  for (int i = 0; i < N; ++i) {
    float li = 0.f;
    if (lane == i) li = sum / Lb[(long)i * n + i];
    li = __shfl(li, i, 64);
    sumsq += li * li;
    if (lane == i) Lb[(long)N * n + i] = li;
    if (lane > i && lane < N) sum -= Lb[(long)lane * n + i] * li;
  }
  if (lane == 0) Lb[(long)N * n + N] = sqrtf(k_ii - sumsq);
}

// obs_info row N and variance downdate over the whole domain (cov_gpu.cu:162-182); streams N rows of length d.
__global__ __launch_bounds__(256) void obs_info_kernel(const float* __restrict__ k_id, const float* __restrict__ L,
                                                       float* __restrict__ obs_info, float* __restrict__ var, int n,
                                                       int d, int N) {
  __shared__ float lrow[64];
  const int b = blockIdx.y;
  const float* Lb = L + (long)b * n * n + (long)N * n;
  if (threadIdx.x <= N && threadIdx.x < 64) lrow[threadIdx.x] = Lb[threadIdx.x];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= d) return;
  float* ob = obs_info + (long)b * n * d;
  float sum = k_id[(long)b * d + j];
  for (int i = 0; i < N; ++i) sum -= ob[(long)i * d + j] * lrow[i];
  const float v = sum / lrow[N];
  ob[(long)N * d + j] = v;
  var[(long)b * d + j] -= v * v;
}


// greedy_loop.get_next_inds (samplers.py:219-239): cost = (sqrt(var) with NaN -> 0, + 1e-10) * [every chosen point is
// farther than dist_thresh]; argmax (first maximum) per batch.  The "all chosen points" test is kept as a running mask
// (mask &= dist^2 to the k NEW points > thresh^2), which is the same predicate.  One workgroup per batch element.
__global__ __launch_bounds__(1024) void greedy_next_kernel(const float* __restrict__ var, const float* __restrict__ dom,
                                                           const float* __restrict__ chosen, int k, uint8_t* __restrict__ mask,
                                                           float thresh_sq, long* __restrict__ best_idx,
                                                           float* __restrict__ max_stdev, int d) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* vb = var + (long)b * d;
  const float* db = dom + (long)b * d * 2;
  uint8_t* mb = mask + (long)b * d;
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  for (int j = tid; j < d; j += 1024) {
    uint8_t ok = mb[j];
    const float y = db[2 * j], x = db[2 * j + 1];
    for (int c = 0; c < k; ++c) {
      const float dy = chosen[((long)b * k + c) * 2] - y, dx = chosen[((long)b * k + c) * 2 + 1] - x;
      const float d2 = dy * dy + dx * dx;
      ok = ok && (d2 > thresh_sq);
    }
    mb[j] = ok;
    float sd = sqrtf(vb[j]);
    if (sd != sd) sd = 0.f;
    sd += 1e-10f;
    const float cost = ok ? sd : 0.f;
    if (cost > best) { best = cost; bi = j; best_sd = sd; }       // ascending j per thread: first maximum kept
  }
  __shared__ float sc[1024], ss[1024];
  __shared__ int si[1024];
  sc[tid] = best; si[tid] = bi; ss[tid] = best_sd;
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if (tid < h) {
      const float c2 = sc[tid + h];
      const int i2 = si[tid + h];
      if (c2 > sc[tid] || (c2 == sc[tid] && i2 < si[tid])) { sc[tid] = c2; si[tid] = i2; ss[tid] = ss[tid + h]; }
    }
    __syncthreads();
  }
  if (tid == 0) { best_idx[b] = si[0]; max_stdev[b] = ss[0]; }
}


// ---- the whole greedy loop on the device (samplers.py:196-282 without early termination): per step two launches ----
// greedy_pick: get_next_inds (running distance mask, first maximum) and the gather of the chosen point into slot `slot`.
// (body shared with greedy_small_loop_kernel, where the arrays the steps write -- var, coords_n, E_n, mask -- are re-read inside
// ONE launch: no __restrict__ on them, so that no load of them is treated as invariant)
__device__ __forceinline__ void greedy_pick_body(const int b, const int tid, const float* var, const float* __restrict__ dom,
                                                 const float* __restrict__ Edom, float* coords_n, float* E_n, long* inds, int n,
                                                 int k0, int k, uint8_t* mask, float thresh_sq, int slot, long* best_idx,
                                                 float* max_stdev, int d) {
#pragma clang fp contract(off)
  const float* vb = var + (long)b * d;
  const float* db = dom + (long)b * d * 2;
  uint8_t* mb = mask + (long)b * d;
  const float* chosen = coords_n + ((long)b * n + k0) * 2;      // the k points added since the last call
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  for (int j = tid; j < d; j += 1024) {
    uint8_t ok = mb[j];
    const float y = db[2 * j], x = db[2 * j + 1];
    for (int c = 0; c < k; ++c) {
      const float dy = chosen[2 * c] - y, dx = chosen[2 * c + 1] - x;
      const float d2 = dy * dy + dx * dx;
      ok = ok && (d2 > thresh_sq);
    }
    mb[j] = ok;
    float sd = sqrtf(vb[j]);
    if (sd != sd) sd = 0.f;
    sd += 1e-10f;
    const float cost = ok ? sd : 0.f;
    if (cost > best) { best = cost; bi = j; best_sd = sd; }
  }
  __shared__ float sc[1024], ss[1024];
  __shared__ int si[1024];
  sc[tid] = best; si[tid] = bi; ss[tid] = best_sd;
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if (tid < h) {
      const float c2 = sc[tid + h];
      const int i2 = si[tid + h];
      if (c2 > sc[tid] || (c2 == sc[tid] && i2 < si[tid])) { sc[tid] = c2; si[tid] = i2; ss[tid] = ss[tid + h]; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int w = si[0];
    best_idx[b] = w;
    max_stdev[b] = ss[0];
    if (slot < n) {
      inds[(long)b * n + slot] = w;
      coords_n[((long)b * n + slot) * 2] = db[2 * w];
      coords_n[((long)b * n + slot) * 2 + 1] = db[2 * w + 1];
      for (int e = 0; e < 4; ++e) E_n[((long)b * n + slot) * 4 + e] = Edom[((long)b * d + w) * 4 + e];
    }
  }
}

__global__ __launch_bounds__(1024) void greedy_pick_kernel(const float* var, const float* __restrict__ dom,
                                                           const float* __restrict__ Edom, float* coords_n, float* E_n, long* inds,
                                                           int n, int k0, int k, uint8_t* mask, float thresh_sq, int slot,
                                                           long* best_idx, float* max_stdev, int d) {
  greedy_pick_body(blockIdx.x, threadIdx.x, var, dom, Edom, coords_n, E_n, inds, n, k0, k, mask, thresh_sq, slot, best_idx, max_stdev, d);
}

// Two-stage form of greedy_pick for large domains (one workgroup walking 300k candidates took 250 us per added point):
// greedy_scan updates the distance mask and finds the best candidate of its slice, greedy_pick2 reduces the slices and gathers
// the chosen point.  Same ordering rule at every level (largest cost, then smallest index), so the result is the one of the
// single-workgroup kernel.
__global__ __launch_bounds__(256) void greedy_scan_kernel(const float* __restrict__ var, const float* __restrict__ dom,
                                                          const float* __restrict__ coords_n, int n, int k0, int k,
                                                          uint8_t* __restrict__ mask, float thresh_sq, int d,
                                                          float4* __restrict__ part) {
#pragma clang fp contract(off)
  const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  const float* vb = var + (long)b * d;
  const float* db = dom + (long)b * d * 2;
  uint8_t* mb = mask + (long)b * d;
  const float* chosen = coords_n + ((long)b * n + k0) * 2;
  const int per = (d + G - 1) / G, j0 = g * per, j1 = min(d, j0 + per);
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  for (int j = j0 + tid; j < j1; j += 256) {
    uint8_t ok = mb[j];
    const float y = db[2 * j], x = db[2 * j + 1];
    for (int c = 0; c < k; ++c) {
      const float dy = chosen[2 * c] - y, dx = chosen[2 * c + 1] - x;
      const float d2 = dy * dy + dx * dx;
      ok = ok && (d2 > thresh_sq);
    }
    mb[j] = ok;
    float sd = sqrtf(vb[j]);
    if (sd != sd) sd = 0.f;
    sd += 1e-10f;
    const float cost = ok ? sd : 0.f;
    if (cost > best) { best = cost; bi = j; best_sd = sd; }
  }
  __shared__ float sc[256], ss[256];
  __shared__ int si[256];
  sc[tid] = best; si[tid] = bi; ss[tid] = best_sd;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) {
      const float c2 = sc[tid + h];
      const int i2 = si[tid + h];
      if (c2 > sc[tid] || (c2 == sc[tid] && i2 < si[tid])) { sc[tid] = c2; si[tid] = i2; ss[tid] = ss[tid + h]; }
    }
    __syncthreads();
  }
  if (tid == 0) part[(long)b * G + g] = make_float4(sc[0], __int_as_float(si[0]), ss[0], 0.f);
}

__global__ __launch_bounds__(256) void greedy_pick2_kernel(const float4* __restrict__ part, int G, const float* __restrict__ dom,
                                                           const float* __restrict__ Edom, float* __restrict__ coords_n,
                                                           float* __restrict__ E_n, long* __restrict__ inds, int n, int slot,
                                                           long* __restrict__ best_idx, float* __restrict__ max_stdev, int d) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* db = dom + (long)b * d * 2;
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  for (int g = tid; g < G; g += 256) {
    const float4 p = part[(long)b * G + g];
    const int i2 = __float_as_int(p.y);
    if (p.x > best || (p.x == best && i2 < bi)) { best = p.x; bi = i2; best_sd = p.z; }
  }
  __shared__ float sc[256], ss[256];
  __shared__ int si[256];
  sc[tid] = best; si[tid] = bi; ss[tid] = best_sd;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) {
      const float c2 = sc[tid + h];
      const int i2 = si[tid + h];
      if (c2 > sc[tid] || (c2 == sc[tid] && i2 < si[tid])) { sc[tid] = c2; si[tid] = i2; ss[tid] = ss[tid + h]; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int w = si[0];
    best_idx[b] = w;
    max_stdev[b] = ss[0];
    if (slot < n) {
      inds[(long)b * n + slot] = w;
      coords_n[((long)b * n + slot) * 2] = db[2 * w];
      coords_n[((long)b * n + slot) * 2 + 1] = db[2 * w + 1];
      for (int e = 0; e < 4; ++e) E_n[((long)b * n + slot) * 4 + e] = Edom[((long)b * d + w) * 4 + e];
    }
  }
}

// greedy_append: k_ni, the new Cholesky row (every workgroup redoes the tiny forward substitution from an LDS copy of L;
// workgroup 0 stores it), then k_id, the obs_info row and the variance downdate of this workgroup's 256 domain pixels.
// Arithmetic = cross_cov_kernel<float> + chol_row_kernel + obs_info_kernel.
template <int NT>
__device__ __forceinline__ void greedy_append_body(const int b, const int blk, const int tid, const float* coords_n, const float* E_n,
                                                   const float* __restrict__ dom, const float* __restrict__ Edom, float* L,
                                                   float* obs_info, float* var, float scale, float k_ii, int n, int d, int N) {
  __shared__ float sx[64 * 2], sE[64 * 4], sL[64 * 65], lrow[64];
  float* Lb = L + (long)b * n * n;
  for (int e = tid; e < (N + 1) * 2; e += NT) sx[e] = coords_n[(long)b * n * 2 + e];
  for (int e = tid; e < (N + 1) * 4; e += NT) sE[e] = E_n[(long)b * n * 4 + e];
  for (int e = tid; e < N * N; e += NT) { const int r = e / N, c = e % N; sL[r * 65 + c] = Lb[(long)r * n + c]; }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    float sum = 0.f;
    if (lane < N) sum = cov_value_f32(sx[2 * lane], sx[2 * lane + 1], sE + 4 * lane, sx[2 * N], sx[2 * N + 1], sE + 4 * N, scale);
    float sumsq = 0.f;
    // Row `lane` of L in registers, the step's value broadcast through a scalar register: the dependent chain of a step is
    // division -> readlane -> multiply-add (~0.05 us).  The first version read the row element and the diagonal from LDS inside the
    // step and broadcast through ds_bpermute: two LDS round trips, an LDS crossbar trip and three divergent branches per step,
    // 0.4 us each -- 25 us of a 30 us launch at N = 60, in every one of the 64 launches of a sampling pass.  Same operations on the
    // same operands (lane i's quotient is sum_i / L_ii; the other lanes' quotients are discarded).
    float Lr[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) Lr[i] = sL[lane * 65 + i];
    const float diag = sL[lane * 65 + lane];
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i < N) {
        const float q = sum / diag;
        const float li = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), i));
        sumsq += li * li;
        if (lane == i) mine = li;
        if (lane > i && lane < N) sum -= Lr[i] * li;
      }
    }
    if (lane < N) lrow[lane] = mine;
    if (lane == 0) lrow[N] = sqrtf(k_ii - sumsq);
  }
  __syncthreads();
  if (blk == 0 && tid <= N) Lb[(long)N * n + tid] = lrow[tid];
  const int j = blk * NT + tid;
  if (j < d) {
    const float* xd = dom + ((long)b * d + j) * 2;
    float sum = cov_value_f32(sx[2 * N], sx[2 * N + 1], sE + 4 * N, xd[0], xd[1], Edom + ((long)b * d + j) * 4, scale);
    float* ob = obs_info + (long)b * n * d;
    for (int i = 0; i < N; ++i) sum -= ob[(long)i * d + j] * lrow[i];
    const float v = sum / lrow[N];
    ob[(long)N * d + j] = v;
    var[(long)b * d + j] -= v * v;
  }
}

__global__ __launch_bounds__(256) void greedy_append_kernel(const float* coords_n, const float* E_n, const float* __restrict__ dom,
                                                            const float* __restrict__ Edom, float* L, float* obs_info, float* var,
                                                            float scale, float k_ii, int n, int d, int N) {
  greedy_append_body<256>(blockIdx.y, blockIdx.x, threadIdx.x, coords_n, E_n, dom, Edom, L, obs_info, var, scale, k_ii, n, d, N);
}

// greedy_append + the greedy_scan of the NEXT pick in one launch (large domains): the thread that just downdated the variance of
// its domain pixel applies the distance mask of the point being added (the one new point a scan after an append sees) and the
// workgroup leaves its best candidate for greedy_pick2 -- the separate scan launch re-read var, dom and mask (6 us x 33 per
// sampling pass at 640x480).  Same operations on the same values as the two kernels (the scan part keeps their contraction-off
// arithmetic), same ordering rule, so the same picks.
__global__ __launch_bounds__(256) void greedy_append_scan_kernel(const float* coords_n, const float* E_n, const float* __restrict__ dom,
                                                                 const float* __restrict__ Edom, float* L, float* obs_info, float* var,
                                                                 float scale, float k_ii, int n, int d, int N, uint8_t* mask,
                                                                 float thresh_sq, float4* __restrict__ part) {
  const int b = blockIdx.y, tid = threadIdx.x;
  greedy_append_body<256>(b, blockIdx.x, tid, coords_n, E_n, dom, Edom, L, obs_info, var, scale, k_ii, n, d, N);
  const int j = blockIdx.x * 256 + tid;
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  if (j < d) {
#pragma clang fp contract(off)
    uint8_t* mb = mask + (long)b * d;
    const float* db = dom + (long)b * d * 2;
    const float* chosen = coords_n + ((long)b * n + N) * 2;
    uint8_t ok = mb[j];
    const float y = db[2 * j], x = db[2 * j + 1];
    const float dy = chosen[0] - y, dx = chosen[1] - x;
    const float d2 = dy * dy + dx * dx;
    ok = ok && (d2 > thresh_sq);
    mb[j] = ok;
    float sd = sqrtf(var[(long)b * d + j]);                 // (this thread's own store in greedy_append_body)
    if (sd != sd) sd = 0.f;
    sd += 1e-10f;
    const float cost = ok ? sd : 0.f;
    if (cost > best) { best = cost; bi = j; best_sd = sd; }
  }
  __shared__ float sc[256], ss[256];
  __shared__ int si[256];
  sc[tid] = best; si[tid] = bi; ss[tid] = best_sd;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) {
      const float c2 = sc[tid + h];
      const int i2 = si[tid + h];
      if (c2 > sc[tid] || (c2 == sc[tid] && i2 < si[tid])) { sc[tid] = c2; si[tid] = i2; ss[tid] = ss[tid + h]; }
    }
    __syncthreads();
  }
  if (tid == 0) part[(long)b * gridDim.x + blockIdx.x] = make_float4(sc[0], __int_as_float(si[0]), ss[0], 0.f);
}

// Small domains (d <= 1024: the thinning of a keyframe's tracked points, <= 64 candidates) -- the whole greedy loop in ONE launch of
// one workgroup per batch item: the same two bodies step after step (pick, then append + pick per added point), workgroup barriers
// where the launches were.  Bit-identical to the launch-per-step form (same code on the same data in the same order); 2 (n - m) + 1
// launches of ~4 us become one.
__global__ __launch_bounds__(1024) void greedy_small_loop_kernel(float* coords_n, float* E_n, long* inds, const float* __restrict__ dom,
                                                                 const float* __restrict__ Edom, float* L, float* obs_info, float* var,
                                                                 uint8_t* mask, long* best_idx, float* max_stdev, float scale, float k_ii,
                                                                 float thresh_sq, int B, int n, int d, int m, float* sd_trace) {
  const int b = blockIdx.x, tid = threadIdx.x;
  greedy_pick_body(b, tid, var, dom, Edom, coords_n, E_n, inds, n, 0, m, mask, thresh_sq, m, best_idx,
                   sd_trace ? sd_trace + (long)m * B : max_stdev, d);
  for (int i = m; i < n; ++i) {
    __syncthreads();
    greedy_append_body<1024>(b, 0, tid, coords_n, E_n, dom, Edom, L, obs_info, var, scale, k_ii, n, d, i);
    __syncthreads();
    greedy_pick_body(b, tid, var, dom, Edom, coords_n, E_n, inds, n, i, 1, mask, thresh_sq, i + 1, best_idx,
                     sd_trace ? sd_trace + (long)(i + 1) * B : max_stdev, d);
  }
}

// The THINNING pass of a keyframe insertion (corr.py:166-176 of the reference: sample_sparse_coords with coords_domain = the <= 64
// tracked points, no current points, terminate_early) in ONE launch of one workgroup: the seed of precalc_entropy_vars' m = 0 branch
// (samplers.py:149-165: largest det E, K_nn = k(x0, x0) [+ fixed_var], L00 = sqrt, K_md, obs_info row 0 = K_md / L00,
// var = signal_var - obs^2 -- each with the rounding of the torch op / native kernel it replaces), then the greedy loop of
// greedy_small_loop_kernel, cut at the first step whose largest remaining standard deviation falls below the threshold
// (samplers.py:255-259).  count_out: number of valid entries of inds (0: K_nn not positive -- the caller raises as the reference's
// torch.linalg.cholesky does).  ~45 torch / native launches and one read-back of the trace become this launch and one read-back.
__global__ __launch_bounds__(1024) void greedy_thin_kernel(const float* __restrict__ dom, const float* __restrict__ Edom, float* coords_n,
                                                           float* E_n, long* inds, float* L, float* obs_info, float* var, uint8_t* mask,
                                                           long* best_idx, float* sd_trace, float scale, float signal_var,
                                                           float fixed_var, float thresh_sq, float stdev_thresh, int n, int d,
                                                           long* count_out) {
  const int tid = threadIdx.x;
  __shared__ float sa[1024];
  __shared__ int sj[1024];
  __shared__ float seedv[8];
  {
#pragma clang fp contract(off)
    float area = -__builtin_inff();
    if (tid < d) {
      const float* e = Edom + 4 * tid;
      const float p0 = e[0] * e[3], p1 = e[1] * e[2];
      area = p0 - p1;
    }
    sa[tid] = area; sj[tid] = tid < d ? tid : 0x7fffffff;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
      if (tid < h) {
        const float a2 = sa[tid + h];
        const int j2 = sj[tid + h];
        if (a2 > sa[tid] || (a2 == sa[tid] && j2 < sj[tid])) { sa[tid] = a2; sj[tid] = j2; }
      }
      __syncthreads();
    }
    const int w0 = sj[0];
    if (tid == 0) {
      inds[0] = w0;
      coords_n[0] = dom[2 * w0]; coords_n[1] = dom[2 * w0 + 1];
      for (int e = 0; e < 4; ++e) E_n[e] = Edom[4 * w0 + e];
      float k00 = cov_value_f32(dom[2 * w0], dom[2 * w0 + 1], Edom + 4 * w0, dom[2 * w0], dom[2 * w0 + 1], Edom + 4 * w0, scale);
      if (fixed_var != 0.f) k00 = k00 + fixed_var;
      seedv[0] = k00;
      seedv[1] = sqrtf(k00);
      L[0] = seedv[1];
    }
    __syncthreads();
    if (!(seedv[0] > 0.f)) {                              // not positive definite (or NaN): nothing is valid
      if (tid == 0) *count_out = 0;
      return;
    }
    if (tid < d) {
      const float kmd = cov_value_f32(dom[2 * w0], dom[2 * w0 + 1], Edom + 4 * w0, dom[2 * tid], dom[2 * tid + 1], Edom + 4 * tid, scale);
      const float o = kmd / seedv[1];
      obs_info[tid] = o;
      const float o2 = o * o;
      var[tid] = signal_var - o2;
      mask[tid] = 1;
    }
  }
  __syncthreads();
  const float k_ii = signal_var + fixed_var;
  int count = n;
  greedy_pick_body(0, tid, var, dom, Edom, coords_n, E_n, inds, n, 0, 1, mask, thresh_sq, 1, best_idx, sd_trace + 1, d);
  __syncthreads();
  for (int i = 1; i < n; ++i) {
    // (written by thread 0 before the barrier; read through an atomic load: a uniform address could otherwise be served from the
    // scalar cache, which does not see vector stores)
    if (__hip_atomic_load(&sd_trace[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < stdev_thresh) { count = i; break; }
    greedy_append_body<1024>(0, 0, tid, coords_n, E_n, dom, Edom, L, obs_info, var, scale, k_ii, n, d, i);
    __syncthreads();
    greedy_pick_body(0, tid, var, dom, Edom, coords_n, E_n, inds, n, i, 1, mask, thresh_sq, i + 1, best_idx, sd_trace + i + 1, d);
    __syncthreads();
  }
  if (tid == 0) *count_out = count;
}

// ---- the large-domain greedy loop as ONE persistent launch (round 6) -----------------------------------------------------------
// The launch-per-step form streams, for the i-th added point, the i previous obs_info rows of every domain pixel (1.2 MB each at a
// 640x480 domain: 34-63 rows = 40-76 MB per step, 19.7 us at 2.9 TB/s) -- byte-bound, ~1 ms per keyframe insertion -- although a
// pixel's column obs_info[0..i][j] is only ever read by the thread that owns pixel j.  Here every thread keeps the columns of its
// GP_PPT pixels in REGISTERS for the whole loop (64 rows x 3 pixels = 192 VGPRs: 448 pixel threads per workgroup, one workgroup per
// compute unit, 224 workgroups cover 300k pixels), so a step is arithmetic + ONE grid-wide exchange of the workgroups' best
// candidates (the "pick2" reduction every workgroup then redoes for itself):
//   * wave 7 of every workgroup is the CHAIN wave: it keeps row `lane` of L in registers and forms the new Cholesky row
//     (greedy_append_body's readlane chain) while the seven pixel waves evaluate k(x_new, x_j) for their pixels;
//   * the pixel waves then subtract their column . row, downdate the variance, apply the new point's distance mask and leave the
//     workgroup's best candidate in its own 128-byte line of `part[step]` (agent-scope stores, one arrival on the step's counter --
//     csrc/cholp.hip's hand-off protocol: every line is written once before anyone reads it);
//   * after the arrivals every workgroup reduces the G candidates with the same order rule (largest cost, then smallest index) and
//     workgroup 0 records the pick.
// Same operations on the same values as greedy_append_scan_kernel + greedy_pick2_kernel, hence the same picks (tested); obs_info
// rows and the downdated variance are NOT written back (the loop's caller discards them).  One image (B = 1).
constexpr int GP_THREADS = 512;
constexpr int GP_PIX_THREADS = 448;    // waves 0..6 own pixels, wave 7 is the chain wave
constexpr int GP_PPT = 3;
constexpr int GP_LINE = 32;            // floats per workgroup record of `part` (its own 128-byte line)
constexpr int GP_LROWS = 16;           // the first rows of every column live in LDS (86 KB), rows 16..63 in registers (144 VGPRs)
constexpr int GP_RROWS = 64 - GP_LROWS;
constexpr int GP_LDS_BYTES = GP_LROWS * GP_PPT * GP_PIX_THREADS * 4;

struct GPArgs {
  float* coords_n; float* E_n; long* inds;
  const float* dom; const float* Edom;
  float* L; const float* obs_info; const float* var; const uint8_t* mask;
  long* best_idx; float* max_stdev; float* sd_trace;
  float* part;            // (n + 1) x G x GP_LINE
  unsigned* cnt;          // (n + 2) x 32 words: arrival counter per step, then the error flag
  int* status;            // 0 ok, -1 a wait timed out (workgroups not co-resident)
  float scale, k_ii, thresh_sq;
  int n, d, m;
};

__device__ __forceinline__ bool gp_better(float c2, int i2, float c1, int i1) { return c2 > c1 || (c2 == c1 && i2 < i1); }

// workgroup argmax under the order rule (largest cost, then smallest index): wave shuffles, then the 8 wave results through LDS
__device__ __forceinline__ void gp_wg_best(float& c, int& i, float& sdv, float* red_c, int* red_i, float* red_s) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float c2 = __shfl_xor(c, off, 64);
    const int i2 = __shfl_xor(i, off, 64);
    const float s2 = __shfl_xor(sdv, off, 64);
    if (gp_better(c2, i2, c, i)) { c = c2; i = i2; sdv = s2; }
  }
  if (lane == 0) { red_c[wv] = c; red_i[wv] = i; red_s[wv] = sdv; }
  __syncthreads();
  c = red_c[0]; i = red_i[0]; sdv = red_s[0];
#pragma unroll
  for (int k = 1; k < GP_THREADS / 64; ++k)
    if (gp_better(red_c[k], red_i[k], c, i)) { c = red_c[k]; i = red_i[k]; sdv = red_s[k]; }
  __syncthreads();
}

// every workgroup: publish its best candidate (cost, idx, sd | y, x, E of that pixel), wait for all G, reduce -> the pick of this
// step (same on every workgroup) with its coordinates / kernel parameters in s_new.  false: a wait timed out.
__device__ __forceinline__ bool gp_exchange(const GPArgs& a, int step, int G, float cost, int idx, float sd, float* red_c, int* red_i,
                                            float* red_s, int* ok_s, float* s_new, int& w_out, float& sd_out) {
  const int tid = threadIdx.x;
  gp_wg_best(cost, idx, sd, red_c, red_i, red_s);
  float* line = a.part + ((long)step * G + blockIdx.x) * GP_LINE;
  unsigned* cnt = a.cnt + 32 * step;
  unsigned* errf = a.cnt + 32 * (a.n + 1);
  if (tid < 9) {
    // (idx is a pixel of THIS workgroup, or 0x7fffffff when it owns none: its record then never wins)
    const long jj = idx < a.d ? idx : 0;
    float v = tid == 0 ? cost : (tid == 1 ? __int_as_float(idx) : (tid == 2 ? sd : (tid < 5 ? a.dom[2 * jj + (tid - 3)] : a.Edom[4 * jj + (tid - 5)])));
    __hip_atomic_store(&line[tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid < 64) {                                            // (wave 0: its lanes' stores are acknowledged; one arrival, one poller)
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 1;
      for (long spin = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G; ++spin) {
        if ((spin & 63) == 63 && __hip_atomic_load(errf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
        if (spin > 1500000) { atomicExch(errf, 1u); ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      *ok_s = ok;
    }
  }
  __syncthreads();
  if (!*ok_s) return false;
  float c = -1.f, s2 = 0.f;
  int bi = 0x7fffffff;
  if (tid < G) {
    const float* ln = a.part + ((long)step * G + tid) * GP_LINE;
    c = __hip_atomic_load(&ln[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bi = __float_as_int(__hip_atomic_load(&ln[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    s2 = __hip_atomic_load(&ln[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  gp_wg_best(c, bi, s2, red_c, red_i, red_s);
  w_out = bi;
  sd_out = s2;
  // the winner's coordinates / kernel parameters ride in its workgroup's record: pixel j belongs to workgroup (j / 448) % G
  if (tid < 6) {
    const int gw = (int)(((long)bi / GP_PIX_THREADS) % G);
    s_new[tid] = __hip_atomic_load(&a.part[((long)step * G + gw) * GP_LINE + 3 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(GP_THREADS) void greedy_persist_kernel(GPArgs a) {
  __shared__ float red_c[GP_THREADS / 64], red_s[GP_THREADS / 64];
  __shared__ int red_i[GP_THREADS / 64];
  __shared__ float s_new[8];            // the point being added: y, x, E (4)
  __shared__ float s_lrow[64];          // its Cholesky row l_0 .. l_{N-1}, zero beyond
  __shared__ float s_lnn;               // ... and its diagonal entry
  __shared__ int ok_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = gridDim.x, n = a.n, d = a.d, m = a.m;
  const bool chain = wv == 7;
  // ONE register array for both roles (the compiler cannot know that the roles never meet in a wave): a pixel thread's columns
  // obs_info[16..63][pixel k] at R[48 k + i - 16] (rows 0..15 in LDS: s_col[i][k][tid], conflict-free); the chain wave's row `lane`
  // of L at R[0..63].  Every register index below is static.
  extern __shared__ float s_col[];
  float R[GP_PPT * GP_RROWS];
  float pvar[GP_PPT];
  int pj[GP_PPT];
  bool pin[GP_PPT], pok[GP_PPT];
  float cy = 0.f, cx = 0.f, cE[4] = {0.f, 0.f, 0.f, 0.f}, diag = 1.f;      // chain wave: point `lane`
#pragma unroll
  for (int k = 0; k < GP_PPT; ++k) {
    const long j = ((long)k * G + blockIdx.x) * GP_PIX_THREADS + tid;
    pin[k] = !chain && j < d;
    pj[k] = pin[k] ? (int)j : d - 1;
    pvar[k] = a.var[pj[k]];
    pok[k] = pin[k] && a.mask[pj[k]] != 0;
  }
  if (!chain) {
#pragma unroll
    for (int k = 0; k < GP_PPT; ++k) {
#pragma unroll
      for (int i = 0; i < GP_LROWS; ++i) s_col[(i * GP_PPT + k) * GP_PIX_THREADS + tid] = (i < m) ? a.obs_info[(long)i * d + pj[k]] : 0.f;
#pragma unroll
      for (int i = 0; i < GP_RROWS; ++i) R[GP_RROWS * k + i] = (i + GP_LROWS < m) ? a.obs_info[(long)(i + GP_LROWS) * d + pj[k]] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < GP_PPT * GP_RROWS; ++i) R[i] = (i < 64 && lane < m && i < m) ? a.L[(long)lane * n + i] : 0.f;
    if (lane < m) {
      cy = a.coords_n[2 * lane]; cx = a.coords_n[2 * lane + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) cE[e] = a.E_n[4 * lane + e];
      diag = a.L[(long)lane * n + lane];
    }
  }
  // ---- the first pick: distance mask against the m current points (greedy_scan_kernel with k = m) ----
  float best = -1.f, best_sd = 0.f;
  int bi = 0x7fffffff;
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < GP_PPT; ++k) {
      bool ok = pok[k];
      const float pyk = a.dom[2 * (long)pj[k]], pxk = a.dom[2 * (long)pj[k] + 1];
      for (int c = 0; c < m; ++c) {
        const float dy = a.coords_n[2 * c] - pyk, dx = a.coords_n[2 * c + 1] - pxk;
        const float d2 = dy * dy + dx * dx;
        ok = ok && (d2 > a.thresh_sq);
      }
      pok[k] = ok;
      float sd = sqrtf(pvar[k]);
      if (sd != sd) sd = 0.f;
      sd += 1e-10f;
      const float cost = ok ? sd : 0.f;
      if (pin[k] && cost > best) { best = cost; bi = pj[k]; best_sd = sd; }
    }
  }
  int w = 0;
  float wsd = 0.f;
  if (!gp_exchange(a, m, G, best, bi, best_sd, red_c, red_i, red_s, &ok_s, s_new, w, wsd)) {
    if (blockIdx.x == 0 && tid == 0) *a.status = -1;
    return;
  }
  auto record = [&](int slot) {          // greedy_pick2_kernel's tail (the new point already sits in s_new)
    if (tid == 0 && blockIdx.x == 0) {
      a.best_idx[0] = w;
      if (a.sd_trace) a.sd_trace[slot] = wsd; else a.max_stdev[0] = wsd;
      if (slot < n) {
        a.inds[slot] = w;
        a.coords_n[2 * slot] = s_new[0]; a.coords_n[2 * slot + 1] = s_new[1];
        for (int e = 0; e < 4; ++e) a.E_n[4 * slot + e] = s_new[2 + e];
      }
    }
  };
  record(m);
  // ---- the loop: point N = m .. n-1 ----
  for (int N = m; N < n; ++N) {
    float kid[GP_PPT], py[GP_PPT], px[GP_PPT];
    if (chain) {
      // new Cholesky row (greedy_append_body): lane i < N holds k(x_i, x_N), the chain resolves l_0 .. l_{N-1}
      float sum = 0.f;
      if (lane < N) sum = cov_value_f32(cy, cx, cE, s_new[0], s_new[1], s_new + 2, a.scale);
      float sumsq = 0.f, mine = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        if (i < N) {
          const float q = sum / diag;
          const float li = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), i));
          sumsq += li * li;
          if (lane == i) mine = li;
          if (lane > i && lane < N) sum -= R[i] * li;
        }
      }
      const float lNN = sqrtf(a.k_ii - sumsq);
      s_lrow[lane] = lane < N ? mine : 0.f;
      if (lane == 0) s_lnn = lNN;
      // lane N becomes point N: its row of L, its coordinates
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const float li = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));
        R[i] = (lane == N && i < N) ? li : ((lane == N && i == N) ? lNN : R[i]);
      }
      if (lane == N) {
        diag = lNN;
        cy = s_new[0]; cx = s_new[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) cE[e] = s_new[2 + e];
      }
      if (blockIdx.x == 0) {
        if (lane < N) a.L[(long)N * n + lane] = mine;
        if (lane == 0) a.L[(long)N * n + N] = lNN;
      }
    } else {
      // (coordinates / kernel parameters of the thread's pixels are re-read every step -- 24 bytes per pixel from L2 / the
      // memory-side cache -- rather than held in 18 more registers)
#pragma unroll
      for (int k = 0; k < GP_PPT; ++k) {
        py[k] = a.dom[2 * (long)pj[k]]; px[k] = a.dom[2 * (long)pj[k] + 1];
        const float4 e4 = *reinterpret_cast<const float4*>(a.Edom + 4 * (long)pj[k]);
        const float Ek[4] = {e4.x, e4.y, e4.z, e4.w};
        kid[k] = cov_value_f32(s_new[0], s_new[1], s_new + 2, py[k], px[k], Ek, a.scale);
      }
    }
    __syncthreads();
    best = -1.f; best_sd = 0.f; bi = 0x7fffffff;
    if (!chain) {
      const float lNN = s_lnn;
#pragma unroll
      for (int k = 0; k < GP_PPT; ++k) {
        float sum = kid[k];
#pragma unroll
        // (no `i < N` tests: rows N.. of the column and of l are exact zeros, and x - 0 * 0 = x bit for bit -- with the tests every
        // step of the dot product waited for its own LDS read: 8 us per added point)
        for (int i = 0; i < GP_LROWS; ++i) sum -= s_col[(i * GP_PPT + k) * GP_PIX_THREADS + tid] * s_lrow[i];
#pragma unroll
        for (int i = 0; i < GP_RROWS; ++i) sum -= R[GP_RROWS * k + i] * s_lrow[i + GP_LROWS];
        const float v = sum / lNN;
        if (N < GP_LROWS) s_col[(N * GP_PPT + k) * GP_PIX_THREADS + tid] = v;
#pragma unroll
        for (int i = 0; i < GP_RROWS; ++i) R[GP_RROWS * k + i] = (i + GP_LROWS == N) ? v : R[GP_RROWS * k + i];
        pvar[k] -= v * v;
      }
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < GP_PPT; ++k) {
          const float dy = s_new[0] - py[k], dx = s_new[1] - px[k];
          const float d2 = dy * dy + dx * dx;
          pok[k] = pok[k] && (d2 > a.thresh_sq);
          float sd = sqrtf(pvar[k]);
          if (sd != sd) sd = 0.f;
          sd += 1e-10f;
          const float cost = pok[k] ? sd : 0.f;
          if (pin[k] && cost > best) { best = cost; bi = pj[k]; best_sd = sd; }
        }
      }
    }
    if (!gp_exchange(a, N + 1, G, best, bi, best_sd, red_c, red_i, red_s, &ok_s, s_new, w, wsd)) {
      if (blockIdx.x == 0 && tid == 0) *a.status = -1;
      return;
    }
    record(N + 1);
  }
}

long greedy_persist_workspace_bytes(int n, int d) {
  const long G = (d + (long)GP_PIX_THREADS * GP_PPT - 1) / ((long)GP_PIX_THREADS * GP_PPT);
  return ((long)(n + 1) * G * GP_LINE + 32L * (n + 2) + 32) * 4;
}

// ---- the thinning pass for d <= 64 candidates in ONE WAVE (round 6) ----------------------------------------------------------------
// greedy_thin_kernel walks its <= 64 steps with 1024 threads and two workgroup barriers + an LDS tree reduction per step (4.4 us
// each: 280 us per keyframe insertion) although a keyframe never has more than 64 tracked points.  Here lane j IS candidate j (its
// obs_info column in registers) and lane i ALSO holds chosen point i (its row of L in registers): a step is register arithmetic,
// the Cholesky chain of greedy_append_body and two shuffle reductions -- no LDS, no barrier.  Same operations on the same values
// as greedy_thin_kernel, hence the same picks and the same cut (tested).  Writes inds, count, sd_trace, L, coords_n, E_n; the
// scratch arrays obs_info / var / mask of the wide kernel are not touched.
__device__ __forceinline__ void wave_best(float& c, int& i, float& sdv) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float c2 = __shfl_xor(c, off, 64);
    const int i2 = __shfl_xor(i, off, 64);
    const float s2 = __shfl_xor(sdv, off, 64);
    if (c2 > c || (c2 == c && i2 < i)) { c = c2; i = i2; sdv = s2; }
  }
}

__global__ __launch_bounds__(64) void greedy_thin_wave_kernel(const float* __restrict__ dom, const float* __restrict__ Edom,
                                                              float* __restrict__ coords_n, float* __restrict__ E_n, long* __restrict__ inds,
                                                              float* __restrict__ L, long* __restrict__ best_idx, float* __restrict__ sd_trace,
                                                              float scale, float signal_var, float fixed_var, float thresh_sq,
                                                              float stdev_thresh, int n, int d, long* __restrict__ count_out) {
  const int lane = threadIdx.x;
  const bool in = lane < d;
  const int jc = in ? lane : 0;
  // candidate role
  const float py = dom[2 * jc], px = dom[2 * jc + 1];
  float pE[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) pE[e] = Edom[4 * jc + e];
  float obs[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) obs[i] = 0.f;
  // chosen-point role: point `lane`
  float qy = 0.f, qx = 0.f, qE[4] = {1.f, 0.f, 0.f, 1.f}, Lr[64], diag = 1.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) Lr[i] = 0.f;
  // ---- seed: largest det E (first maximum) ----
  float area = -__builtin_inff();
  {
#pragma clang fp contract(off)
    if (in) {
      const float p0 = pE[0] * pE[3], p1 = pE[1] * pE[2];
      area = p0 - p1;
    }
  }
  int w = in ? lane : 0x7fffffff;
  {
    float dummy = 0.f;
    wave_best(area, w, dummy);
  }
  auto bcast = [&](float v, int src) { return __shfl(v, src, 64); };
  float ny = bcast(py, w), nx = bcast(px, w), nE[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) nE[e] = bcast(pE[e], w);
  float k00 = cov_value_f32(ny, nx, nE, ny, nx, nE, scale);
  {
#pragma clang fp contract(off)
    if (fixed_var != 0.f) k00 = k00 + fixed_var;
  }
  if (!(k00 > 0.f)) {                                      // not positive definite (or NaN): nothing is valid
    if (lane == 0) *count_out = 0;
    return;
  }
  const float l00 = sqrtf(k00);
  if (lane == 0) {
    inds[0] = w; coords_n[0] = ny; coords_n[1] = nx;
    for (int e = 0; e < 4; ++e) E_n[e] = nE[e];
    L[0] = l00;
    qy = ny; qx = nx;
    for (int e = 0; e < 4; ++e) qE[e] = nE[e];
    Lr[0] = l00; diag = l00;
  }
  float var;
  bool ok = in;
  {
#pragma clang fp contract(off)
    const float kmd = cov_value_f32(ny, nx, nE, py, px, pE, scale);
    const float o = kmd / l00;
    obs[0] = o;
    const float o2 = o * o;
    var = signal_var - o2;
  }
  const float k_ii = signal_var + fixed_var;
  // pick against the point just added: distance mask, largest remaining standard deviation (greedy_pick_body)
  auto pick = [&](float cy_, float cx_, int& w_out, float& sd_out) {
#pragma clang fp contract(off)
    const float dy = cy_ - py, dx = cx_ - px;
    const float d2 = dy * dy + dx * dx;
    ok = ok && (d2 > thresh_sq);
    float sd = sqrtf(var);
    if (sd != sd) sd = 0.f;
    sd += 1e-10f;
    float cost = in ? (ok ? sd : 0.f) : -1.f;
    int bi = in ? lane : 0x7fffffff;
    wave_best(cost, bi, sd);
    w_out = bi;
    sd_out = sd;
  };
  float wsd;
  pick(ny, nx, w, wsd);
  if (lane == 0) { best_idx[0] = w; sd_trace[1] = wsd; }
  int count = n;
  for (int N = 1; N < n; ++N) {
    if (wsd < stdev_thresh) { count = N; break; }          // sd_trace[N] of the wide kernel
    // point N = candidate w
    ny = bcast(py, w); nx = bcast(px, w);
#pragma unroll
    for (int e = 0; e < 4; ++e) nE[e] = bcast(pE[e], w);
    if (lane == 0) {
      inds[N] = w; coords_n[2 * N] = ny; coords_n[2 * N + 1] = nx;
      for (int e = 0; e < 4; ++e) E_n[4 * N + e] = nE[e];
    }
    // its Cholesky row (greedy_append_body): lane i < N holds k(x_i, x_N)
    float sum = 0.f;
    if (lane < N) sum = cov_value_f32(qy, qx, qE, ny, nx, nE, scale);
    float sumsq = 0.f, mine = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i < N) {
        const float q = sum / diag;
        const float li = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), i));
        sumsq += li * li;
        if (lane == i) mine = li;
        if (lane > i && lane < N) sum -= Lr[i] * li;
      }
    }
    const float lNN = sqrtf(k_ii - sumsq);
    if (lane < N) L[(long)N * n + lane] = mine;
    if (lane == 0) L[(long)N * n + N] = lNN;
    // the candidates' obs_info row N and the variance downdate; lane N becomes point N
    float s2 = cov_value_f32(ny, nx, nE, py, px, pE, scale);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const float li = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));      // 0 for i >= N
      s2 -= obs[i] * li;
      Lr[i] = (lane == N && i < N) ? li : ((lane == N && i == N) ? lNN : Lr[i]);
    }
    const float v = s2 / lNN;
#pragma unroll
    for (int i = 0; i < 64; ++i) obs[i] = (i == N) ? v : obs[i];
    var -= v * v;
    if (lane == N) {
      diag = lNN; qy = ny; qx = nx;
#pragma unroll
      for (int e = 0; e < 4; ++e) qE[e] = nE[e];
    }
    pick(ny, nx, w, wsd);
    if (lane == 0) { best_idx[0] = w; sd_trace[N + 1] = wsd; }
  }
  if (lane == 0) *count_out = count;
}

template <typename T>
int cross_cov(const T* x1, const T* E1, const T* x2, const T* E2, T scale, T* K12, int B, int N, int M,
              const long* strides_host, hipStream_t s) {
  if (!x1 || !E1 || !x2 || !E2 || !K12 || !strides_host || B < 0 || N < 0 || M < 0) return COMO_ERR_ARG;
  if (B == 0 || N == 0 || M == 0) return COMO_OK;
  CovStrides st;
  for (int k = 0; k < 3; ++k) { st.x1[k] = strides_host[k]; st.x2[k] = strides_host[7 + k]; }
  for (int k = 0; k < 4; ++k) { st.E1[k] = strides_host[3 + k]; st.E2[k] = strides_host[10 + k]; }
  const long total = (long)N * M;
  dim3 grid((unsigned)((total + 255) / 256), B);
  hipLaunchKernelGGL(cross_cov_kernel<T>, grid, dim3(256), 0, s, x1, E1, x2, E2, scale, K12, N, M, st);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_cross_covariance_f32(const float* x1, const float* E1, const float* x2, const float* E2, float scale,
                              float* K12, int B, int N, int M, const long* strides_host, como_stream_t stream) {
  return como::cross_cov<float>(x1, E1, x2, E2, scale, K12, B, N, M, strides_host, (hipStream_t)stream);
}
int como_cross_covariance_f64(const double* x1, const double* E1, const double* x2, const double* E2, double scale,
                              double* K12, int B, int N, int M, const long* strides_host, como_stream_t stream) {
  return como::cross_cov<double>(x1, E1, x2, E2, scale, K12, B, N, M, strides_host, (hipStream_t)stream);
}

/* half: the reference dispatches cross_covariance for at::Half too (AT_DISPATCH_FLOATING_TYPES_AND_HALF, cov_gpu.cu:73).
 * c10::Half arithmetic = every operation computed in float and rounded to half: exactly what _Float16 arithmetic gives. */
int como_cross_covariance_f16(const void* x1, const void* E1, const void* x2, const void* E2, float scale, void* K12, int B,
                              int N, int M, const long* strides_host, como_stream_t stream) {
  return como::cross_cov<_Float16>((const _Float16*)x1, (const _Float16*)E1, (const _Float16*)x2, (const _Float16*)E2,
                                   (_Float16)scale, (_Float16*)K12, B, N, M, strides_host, (hipStream_t)stream);
}

int como_chol_append_obs_info_f32(float* L, float* obs_info, float* var, const float* k_ni, const float* k_id,
                                  float k_ii, int B, int n, int d, int N, como_stream_t stream) {
  if (!L || !obs_info || !var || !k_ni || !k_id || B <= 0 || n <= 0 || n > 64 || d <= 0 || N < 0 || N >= n)
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(como::chol_row_kernel, dim3(B), dim3(64), 0, s, L, k_ni, k_ii, n, N);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(como::obs_info_kernel, dim3((d + 255) / 256, B), dim3(256), 0, s, k_id, L, obs_info, var, n, d, N);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_greedy_next_f32(const float* var, const float* coords_domain, const float* chosen, int k, uint8_t* mask,
                         float dist_thresh_sq, long* best_idx, float* max_stdev, int B, int d, como_stream_t stream) {
  if (!var || !coords_domain || !mask || !best_idx || !max_stdev || B <= 0 || d <= 0 || k < 0 || (k > 0 && !chosen))
    return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::greedy_next_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, var, coords_domain, chosen, k, mask,
                     dist_thresh_sq, best_idx, max_stdev, d);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

static int greedy_loop_impl(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                            float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* max_stdev, float scale,
                            float k_ii, float dist_thresh_sq, int B, int n, int d, int m, float* sd_trace, void* scratch,
                            long scratch_floats, como_stream_t stream) {
  using namespace como;
  if (!coords_n || !E_n || !coord_vec_inds || !coords_domain || !E_domain || !L || !obs_info || !var || !mask || !best_idx ||
      !max_stdev || B <= 0 || n <= 0 || n > 64 || d <= 0 || m < 1 || m > n)
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int G = (d + 1023) / 1024;
  if (G > 1024) G = 1024;
  const bool two_stage = scratch != nullptr && G > 1;
  auto pick = [&](int k0, int k, int slot, float* sd_out) {
    if (two_stage) {
      hipLaunchKernelGGL(greedy_scan_kernel, dim3(G, B), dim3(256), 0, s, var, coords_domain, coords_n, n, k0, k, mask,
                         dist_thresh_sq, d, (float4*)scratch);
      hipLaunchKernelGGL(greedy_pick2_kernel, dim3(B), dim3(256), 0, s, (const float4*)scratch, G, coords_domain, E_domain,
                         coords_n, E_n, coord_vec_inds, n, slot, best_idx, sd_out, d);
    } else {
      hipLaunchKernelGGL(greedy_pick_kernel, dim3(B), dim3(1024), 0, s, var, coords_domain, E_domain, coords_n, E_n,
                         coord_vec_inds, n, k0, k, mask, dist_thresh_sq, slot, best_idx, sd_out, d);
    }
  };
  static const bool fused_small = [] { const char* e = getenv("COMO_GREEDY_FUSED"); return !e || e[0] != '0'; }();
  if (fused_small && d <= 1024) {
    hipLaunchKernelGGL(greedy_small_loop_kernel, dim3(B), dim3(1024), 0, s, coords_n, E_n, coord_vec_inds, coords_domain, E_domain, L,
                       obs_info, var, mask, best_idx, max_stdev, scale, k_ii, dist_thresh_sq, B, n, d, m, sd_trace);
    COMO_CHECK_LAUNCH();
    return COMO_OK;
  }
  pick(0, m, m, sd_trace ? sd_trace + (long)m * B : max_stdev);
  COMO_CHECK_LAUNCH();
  // append + the next pick's scan in one launch when the scratch holds one candidate per append workgroup (COMO_GREEDY_FUSED_SCAN=0:
  // the separate scan, for A/B runs)
  static const bool fused_scan = [] { const char* e = getenv("COMO_GREEDY_FUSED_SCAN"); return !e || e[0] != '0'; }();
  const int G2 = (d + 255) / 256;
  if (fused_scan && two_stage && scratch_floats >= 4L * B * G2) {
    for (int i = m; i < n; ++i) {
      hipLaunchKernelGGL(greedy_append_scan_kernel, dim3(G2, B), dim3(256), 0, s, coords_n, E_n, coords_domain, E_domain, L, obs_info,
                         var, scale, k_ii, n, d, i, mask, dist_thresh_sq, (float4*)scratch);
      COMO_CHECK_LAUNCH();
      hipLaunchKernelGGL(greedy_pick2_kernel, dim3(B), dim3(256), 0, s, (const float4*)scratch, G2, coords_domain, E_domain, coords_n,
                         E_n, coord_vec_inds, n, i + 1, best_idx, sd_trace ? sd_trace + (long)(i + 1) * B : max_stdev, d);
      COMO_CHECK_LAUNCH();
    }
    return COMO_OK;
  }
  for (int i = m; i < n; ++i) {
    hipLaunchKernelGGL(greedy_append_kernel, dim3((d + 255) / 256, B), dim3(256), 0, s, coords_n, E_n, coords_domain, E_domain, L,
                       obs_info, var, scale, k_ii, n, d, i);
    COMO_CHECK_LAUNCH();
    pick(i, 1, i + 1, sd_trace ? sd_trace + (long)(i + 1) * B : max_stdev);
    COMO_CHECK_LAUNCH();
  }
  return COMO_OK;
}

/* The large-domain loop as one persistent launch (greedy_persist_kernel above).  workspace: como_greedy_persist_workspace_bytes(n, d)
 * bytes (cleared here); status (device int, zeroed here): -1 if a grid-wide wait timed out (the picks are then invalid).
 * COMO_ERR_ARG when the shape does not fit (B != 1 is not offered; more workgroups than compute units). */
long como_greedy_persist_workspace_bytes(int n, int d) { return como::greedy_persist_workspace_bytes(n, d); }

int como_greedy_persist_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                            float* L, const float* obs_info, const float* var, const uint8_t* mask, long* best_idx, float* max_stdev,
                            float scale, float k_ii, float dist_thresh_sq, int n, int d, int m, float* sd_trace, void* workspace,
                            int* status, como_stream_t stream) {
  using namespace como;
  if (!coords_n || !E_n || !coord_vec_inds || !coords_domain || !E_domain || !L || !obs_info || !var || !mask || !best_idx ||
      !max_stdev || !workspace || !status || n <= 0 || n > 64 || d <= 0 || m < 1 || m > n)
    return COMO_ERR_ARG;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = -1;
    int occ = 0;
    if (ncu > 0 && (hipFuncSetAttribute((const void*)greedy_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES) != hipSuccess ||
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)greedy_persist_kernel, GP_THREADS, GP_LDS_BYTES) != hipSuccess ||
                    occ < 1))
      ncu = -1;
    (void)hipGetLastError();
  }
  const long G = (d + (long)GP_PIX_THREADS * GP_PPT - 1) / ((long)GP_PIX_THREADS * GP_PPT);
  if (ncu < 1 || G > ncu) return COMO_ERR_ARG;             // every workgroup must be resident: at most one per compute unit
  hipStream_t s = (hipStream_t)stream;
  GPArgs a;
  a.coords_n = coords_n; a.E_n = E_n; a.inds = coord_vec_inds; a.dom = coords_domain; a.Edom = E_domain; a.L = L;
  a.obs_info = obs_info; a.var = var; a.mask = mask; a.best_idx = best_idx; a.max_stdev = max_stdev; a.sd_trace = sd_trace;
  a.part = (float*)workspace;
  a.cnt = (unsigned*)(a.part + (long)(n + 1) * G * GP_LINE);
  a.status = status;
  a.scale = scale; a.k_ii = k_ii; a.thresh_sq = dist_thresh_sq; a.n = n; a.d = d; a.m = m;
  if (hipMemsetAsync(a.cnt, 0, (32L * (n + 2)) * 4, s) != hipSuccess || hipMemsetAsync(status, 0, 4, s) != hipSuccess) return COMO_ERR_LAUNCH;
  hipLaunchKernelGGL(greedy_persist_kernel, dim3((unsigned)G), dim3(GP_THREADS), GP_LDS_BYTES, s, a);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_greedy_loop_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                         float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* max_stdev, float scale,
                         float k_ii, float dist_thresh_sq, int B, int n, int d, int m, float* sd_trace, void* scratch,
                         como_stream_t stream) {
  return greedy_loop_impl(coords_n, E_n, coord_vec_inds, coords_domain, E_domain, L, obs_info, var, mask, best_idx, max_stdev, scale,
                          k_ii, dist_thresh_sq, B, n, d, m, sd_trace, scratch, scratch ? 4096L * B : 0, stream);
}

int como_greedy_thin_f32(const float* coords_domain, const float* E_domain, float* coords_n, float* E_n, long* coord_vec_inds,
                         float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* sd_trace, float scale,
                         float signal_var, float fixed_var, float dist_thresh_sq, float stdev_thresh, int n, int d, long* count_out,
                         como_stream_t stream) {
  if (!coords_domain || !E_domain || !coords_n || !E_n || !coord_vec_inds || !L || !obs_info || !var || !mask || !best_idx ||
      !sd_trace || !count_out || n <= 0 || n > 64 || d <= 0 || d > 1024 || n > d)
    return COMO_ERR_ARG;
  static const bool wave_form = [] { const char* e = getenv("COMO_GREEDY_THIN_WAVE"); return !e || e[0] != '0'; }();
  if (wave_form && d <= 64) {
    // (<= 64 candidates -- every keyframe insertion: a keyframe holds at most 64 points -- run in ONE wave, registers only)
    hipLaunchKernelGGL(como::greedy_thin_wave_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, coords_domain, E_domain, coords_n, E_n,
                       coord_vec_inds, L, best_idx, sd_trace, scale, signal_var, fixed_var, dist_thresh_sq, stdev_thresh, n, d, count_out);
    COMO_CHECK_LAUNCH();
    return COMO_OK;
  }
  hipLaunchKernelGGL(como::greedy_thin_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, coords_domain, E_domain, coords_n, E_n,
                     coord_vec_inds, L, obs_info, var, mask, best_idx, sd_trace, scale, signal_var, fixed_var, dist_thresh_sq,
                     stdev_thresh, n, d, count_out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_greedy_loop_ws_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                            float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* max_stdev, float scale,
                            float k_ii, float dist_thresh_sq, int B, int n, int d, int m, float* sd_trace, void* scratch,
                            long scratch_floats, como_stream_t stream) {
  if (scratch && scratch_floats < 4096L * B) return COMO_ERR_ARG;
  return greedy_loop_impl(coords_n, E_n, coord_vec_inds, coords_domain, E_domain, L, obs_info, var, mask, best_idx, max_stdev, scale,
                          k_ii, dist_thresh_sq, B, n, d, m, sd_trace, scratch, scratch_floats, stream);
}

}  // extern "C"
