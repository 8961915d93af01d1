// Stand-alone entry points of the exact radix select (see select.cuh).
#include "select.cuh"
#include "../../include/como_hip.h"

namespace como {

template <typename T>
__global__ __launch_bounds__(256) void select_hist_kernel(const T* __restrict__ r, const uint8_t* __restrict__ valid,
                                                          long n, uint32_t* __restrict__ hists, int pass) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ SelScratch sc;
  // blockIdx.y = segment: independent selects over consecutive length-n slices (e.g. one median per keyframe)
  r += (long)blockIdx.y * n;
  if (valid) valid += (long)blockIdx.y * n;
  hists += (long)blockIdx.y * 6 * SEL_BINS;
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) lh[b] = 0;
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, pass, &sc, prefix, k_rem, nv);   // contains __syncthreads (also orders the lh init)
  __syncthreads();
  // 4 elements per thread and iteration (16 B + 4 B loads) when the slice allows it: the pass is pure streaming and
  // was latency-bound with one scalar load per iteration
  const bool vec = (sizeof(T) == 4) && ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(r) & 15) == 0) &&
                   (!valid || (reinterpret_cast<uintptr_t>(valid) & 3) == 0);
  if (vec) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const float4 rv = reinterpret_cast<const float4*>(r)[i];
      const uint32_t vv = valid ? reinterpret_cast<const uint32_t*>(valid)[i] : 0x01010101u;
      const float e[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((vv >> (8 * k)) & 0xffu) {
          KeyT key = abs_key((T)e[k]);
          if (sel_match<KeyT>(key, prefix, pass)) atomicAdd(&lh[sel_digit<KeyT>(key, pass)], 1u);
        }
      }
    }
  } else {
    // four independent elements per trip (their loads in flight together): with one element per trip the double-precision
    // pass streamed 34 MB in 32 us
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
      T e[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long i = i0 + k * stride;
        ok[k] = i < n;
        const long ic = ok[k] ? i : i0;
        e[k] = r[ic];
        if (valid) ok[k] = ok[k] && valid[ic];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ok[k]) {
          KeyT key = abs_key(e[k]);
          if (sel_match<KeyT>(key, prefix, pass)) atomicAdd(&lh[sel_digit<KeyT>(key, pass)], 1u);
        }
      }
    }
  }
  __syncthreads();
  sel_flush(lh, hists + pass * SEL_BINS);
}

template <typename T>
__global__ __launch_bounds__(256) void select_finish_kernel(const uint32_t* __restrict__ hists, T* __restrict__ out3) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ SelScratch sc;
  KeyT prefix; uint32_t k_rem, nv;
  hists += (long)blockIdx.x * 6 * SEL_BINS;
  out3 += 3 * (long)blockIdx.x;
  sel_resolve<KeyT>(hists, SelCfg<KeyT>::NPASS, &sc, prefix, k_rem, nv);
  if (threadIdx.x == 0) {
    T med = nv ? key_value(prefix) : T(NAN);
    out3[0] = med;
    out3[1] = T(1.4826) * med;      // sigma_r, photo.py:128
    out3[2] = (T)nv;
  }
}

template <typename T>
int select_hist(const T* r, const uint8_t* valid, long n, int nseg, uint32_t* hists, int pass, hipStream_t s) {
  using KeyT = typename KeyOf<T>::type;
  if (!r || !hists || n < 0 || nseg < 1 || pass < 0 || pass >= SelCfg<KeyT>::NPASS) return COMO_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks < 1) blocks = 1;
  // Few, fat workgroups: every workgroup ends with up to 2048 global atomics on the SAME histogram, which serialise per
  // address at the memory side (~15 ns each): measured 32 us with 2048 workgroups, 17 us with 512 (scalar loads).
  const long cap = (nseg >= 4) ? 64 : 256;
  blocks = (blocks + 3) / 4;                         // 4 elements per thread on the vector path
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(select_hist_kernel<T>, dim3((unsigned)blocks, nseg), dim3(256), 0, s, r, valid, n, hists, pass);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_abi_version(void) { return 1; }

int como_select_workspace_bytes(void) { return 6 * como::SEL_BINS * (int)sizeof(uint32_t); }   /* per segment */

int como_select_begin(void* hists, int nseg, como_stream_t stream_) {
  if (!hists || nseg < 1) return COMO_ERR_ARG;
  if (!como::zero_words(hists, (size_t)nseg * 6 * como::SEL_BINS, (hipStream_t)stream_)) return COMO_ERR_LAUNCH;
  return COMO_OK;
}

int como_select_hist_f32(const float* r, const uint8_t* valid, long n, int nseg, void* hists, int pass, como_stream_t stream) {
  return como::select_hist<float>(r, valid, n, nseg, (uint32_t*)hists, pass, (hipStream_t)stream);
}
int como_select_hist_f64(const double* r, const uint8_t* valid, long n, int nseg, void* hists, int pass, como_stream_t stream) {
  return como::select_hist<double>(r, valid, n, nseg, (uint32_t*)hists, pass, (hipStream_t)stream);
}
int como_select_finish_f32(const void* hists, int nseg, float* out3, como_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hists || !out3 || nseg < 1) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::select_finish_kernel<float>, dim3(nseg), dim3(256), 0, stream, (const uint32_t*)hists, out3);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_select_finish_f64(const void* hists, int nseg, double* out3, como_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hists || !out3 || nseg < 1) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::select_finish_kernel<double>, dim3(nseg), dim3(256), 0, stream, (const uint32_t*)hists, out3);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"

namespace como {
template int select_hist<float>(const float*, const uint8_t*, long, int, uint32_t*, int, hipStream_t);
template int select_hist<double>(const double*, const uint8_t*, long, int, uint32_t*, int, hipStream_t);
}  // namespace como
