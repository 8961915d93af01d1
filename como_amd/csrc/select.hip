// Stand-alone entry points of the exact radix select (see select.cuh).
#include <cstdlib>
#include "select.cuh"
#include "../../include/como_hip.h"

namespace como {

// Double keys need six digit passes, but after four of them (43 of 64 bits) the candidate set is a handful of keys: the
// pass-3 kernel also COLLECTS the keys that match the 33 bits resolved so far, a one-workgroup tail kernel finishes digits
// 4 and 5 from that list (writing the same histograms the full passes would) and passes 4, 5 return at once.  Scratch lives
// in the part of the workspace no pass uses: digits 4 and 5 are 10 bits wide, so the upper 1024 words of their slots are free
// -- slot 4: [1024] candidate count, [1025] "done"; slot 5: [1024, 2048) = up to 512 keys.  The tail clears count and keys
// again (consumers scan whole slots); "done" = 1 in an unused bin is harmless.  More candidates than fit (massive ties, e.g.
// constant images): the tail workgroup streams the data itself for digits 4 and 5 -- slow (one workgroup), but it keeps the
// two always-enqueued fallback launches (4.5 us each, every iteration) out of the common path.
constexpr int SEL_COLLECT = 0x100;          // flag or-ed into `pass` (pass 3 of a double select)
constexpr int SEL_NOTAIL = 0x200;           // with SEL_COLLECT on pass 3: collect only, no tail launch (multi-GPU: como_select_cand_*)
constexpr int SEL_CAND_CAP = 512;
constexpr int SEL_CAND_WORDS = 2 + 2 * SEL_CAND_CAP;   // packed candidate list of one segment: count | 0 | 512 keys (u64)
__device__ __forceinline__ uint32_t* sel_cand_count(uint32_t* h) { return h + 4 * SEL_BINS + 1024; }
__device__ __forceinline__ uint32_t* sel_cand_done(uint32_t* h) { return h + 4 * SEL_BINS + 1025; }
__device__ __forceinline__ uint64_t* sel_cand_keys(uint32_t* h) { return reinterpret_cast<uint64_t*>(h + 5 * SEL_BINS + 1024); }
// running maximum of the candidate keys and of their complements (slot 4, words 1026..1029: both start from the cleared scratch's
// zero): when the list overflows but max == min -- massive ties on ONE value, the only way thousands of residuals share 33 leading
// bits in practice (saturated / identical pixels) -- the candidates are `count` copies of that key and still representable
__device__ __forceinline__ unsigned long long* sel_cand_max(uint32_t* h) { return reinterpret_cast<unsigned long long*>(h + 4 * SEL_BINS + 1026); }
__device__ __forceinline__ unsigned long long* sel_cand_maxnot(uint32_t* h) { return reinterpret_cast<unsigned long long*>(h + 4 * SEL_BINS + 1028); }
__device__ __forceinline__ void sel_cand_add(uint32_t* hists, uint64_t key) {
  const uint32_t slot = atomicAdd(sel_cand_count(hists), 1u);
  if (slot < (uint32_t)SEL_CAND_CAP) sel_cand_keys(hists)[slot] = key;
  if (slot >= (uint32_t)SEL_CAND_CAP - 1) {             // (only lists that fill up pay for the two extra atomics; the entry that fills
    atomicMax(sel_cand_max(hists), (unsigned long long)key);          //  the last slot and every later one are folded in, the
    atomicMax(sel_cand_maxnot(hists), ~(unsigned long long)key);      //  stored ones are compared by the pack kernel)
  }
}

// NTHR = 256, or 1024 for long slices: the flush at the end of a workgroup (up to 2048 same-address global atomics) caps the
// number of workgroups at one per compute unit, and four waves per compute unit keep too few loads in flight to stream
// (34 MB of double keys took 22 us) -- sixteen waves share the same LDS histogram and the same single flush.
template <typename T, int NTHR>
__global__ __launch_bounds__(NTHR) void select_hist_kernel(const T* __restrict__ r, const uint8_t* __restrict__ valid,
                                                           long n, uint32_t* __restrict__ hists, int pass_flags) {
  using KeyT = typename KeyOf<T>::type;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ SelScratch sc;
  const int pass = pass_flags & 0xff;
  const bool collect = (sizeof(T) == 8) && (pass_flags & SEL_COLLECT) && pass == 3;
  // blockIdx.y = segment: independent selects over consecutive length-n slices (e.g. one median per keyframe)
  r += (long)blockIdx.y * n;
  if (valid) valid += (long)blockIdx.y * n;
  hists += (long)blockIdx.y * 6 * SEL_BINS;
  if (sizeof(T) == 8 && pass >= 4 && *sel_cand_done(hists)) return;      // the tail kernel already produced this digit
                                                                        // (plain passes 4, 5 after a collecting pass 3)
  for (int b = threadIdx.x; b < SEL_BINS; b += NTHR) lh[b] = 0;
  KeyT prefix; uint32_t k_rem, nv;
  if constexpr (NTHR == 256) sel_resolve<KeyT>(hists, pass, &sc, prefix, k_rem, nv);   // contains __syncthreads (also orders the lh init)
  else sel_resolve_wide<KeyT>(hists, pass, &sc, prefix, k_rem, nv);
  __syncthreads();
  // 4 elements per thread and iteration (16 B + 4 B loads) when the slice allows it: the pass is pure streaming and
  // was latency-bound with one scalar load per iteration
  const bool vec = (sizeof(T) == 4) && ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(r) & 15) == 0) &&
                   (!valid || (reinterpret_cast<uintptr_t>(valid) & 3) == 0);
  if (vec) {
    // four independent 16-byte loads per thread and trip: with one, 256 workgroups keep 1 MB in flight and the pass is
    // latency-bound (21 MB in 17 us)
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * NTHR;
    for (long i0 = (long)blockIdx.x * NTHR + threadIdx.x; i0 < n4; i0 += 4 * stride) {
      float4 rv[4];
      uint32_t vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * stride;
        const bool in = i < n4;
        rv[u] = reinterpret_cast<const float4*>(r)[in ? i : i0];
        vv[u] = !in ? 0u : (valid ? reinterpret_cast<const uint32_t*>(valid)[i] : 0x01010101u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float e[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((vv[u] >> (8 * k)) & 0xffu) {
            KeyT key = abs_key((T)e[k]);
            if (sel_match<KeyT>(key, prefix, pass)) atomicAdd(&lh[sel_digit<KeyT>(key, pass)], 1u);
          }
        }
      }
    }
  } else if (sizeof(T) == 8 && ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(r) & 31) == 0) &&
             (!valid || (reinterpret_cast<uintptr_t>(valid) & 3) == 0)) {
    // Round 4, double keys: 2 x (32 B of keys + 4 B of validity) per thread and trip instead of 4 x (8 B + 1 B): the scalar form
    // issued eight load instructions per four keys and streamed the 34 MB of the dense window's residuals at 2 TB/s (16.9 us per
    // pass, three passes on the critical path of every float64 iteration)
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * NTHR;
    for (long i0 = (long)blockIdx.x * NTHR + threadIdx.x; i0 < n4; i0 += 2 * stride) {
      double2 ra[2][2];
      uint32_t vv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long i = i0 + u * stride;
        const bool in = i < n4;
        const double2* rp = reinterpret_cast<const double2*>(r) + 2 * (in ? i : i0);
        ra[u][0] = rp[0]; ra[u][1] = rp[1];
        vv[u] = !in ? 0u : (valid ? reinterpret_cast<const uint32_t*>(valid)[i] : 0x01010101u);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const double e[4] = {ra[u][0].x, ra[u][0].y, ra[u][1].x, ra[u][1].y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((vv[u] >> (8 * k)) & 0xffu) {
            const KeyT key = abs_key((T)e[k]);
            if (sel_match<KeyT>(key, prefix, pass)) {
              atomicAdd(&lh[sel_digit<KeyT>(key, pass)], 1u);
              if (collect) sel_cand_add(hists, (uint64_t)key);
            }
          }
        }
      }
    }
  } else {
    // four independent elements per trip (their loads in flight together): with one element per trip the double-precision
    // pass streamed 34 MB in 32 us
    const long stride = (long)gridDim.x * NTHR;
    for (long i0 = (long)blockIdx.x * NTHR + threadIdx.x; i0 < n; i0 += 4 * stride) {
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
          if (sel_match<KeyT>(key, prefix, pass)) {
            atomicAdd(&lh[sel_digit<KeyT>(key, pass)], 1u);
            if constexpr (sizeof(T) == 8) {
              if (collect) sel_cand_add(hists, (uint64_t)key);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  sel_flush(lh, hists + pass * SEL_BINS);
}

// digits 4 and 5 of a double select from the collected candidates (one workgroup per segment)
__global__ __launch_bounds__(256) void select_tail_kernel(uint32_t* __restrict__ hists, const double* __restrict__ r,
                                                          const uint8_t* __restrict__ valid, long n) {
  using KeyT = uint64_t;
  __shared__ SelScratch sc;
  __shared__ uint64_t keys[SEL_CAND_CAP];
  __shared__ uint32_t lh[1024];
  __shared__ uint32_t part[64];
  __shared__ uint32_t found[2];
  hists += (long)blockIdx.x * 6 * SEL_BINS;
  r += (long)blockIdx.x * n;
  if (valid) valid += (long)blockIdx.x * n;
  const int tid = threadIdx.x;
  const uint32_t cnt = *sel_cand_count(hists);
  const bool fits = cnt <= (uint32_t)SEL_CAND_CAP;
  const uint32_t nk = fits ? cnt : (uint32_t)SEL_CAND_CAP;
  for (uint32_t i = tid; i < nk; i += 256) keys[i] = sel_cand_keys(hists)[i];
  __syncthreads();
  // clean the scratch: consumers scan whole slots
  for (int i = tid; i < 1024; i += 256) hists[5 * SEL_BINS + 1024 + i] = 0u;
  if (tid == 0) { *sel_cand_count(hists) = 0u; *sel_cand_max(hists) = 0ull; *sel_cand_maxnot(hists) = 0ull; }
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, 4, &sc, prefix, k_rem, nv);
  for (int p = 4; p < 6; ++p) {
    for (int b = tid; b < 1024; b += 256) lh[b] = 0;
    __syncthreads();
    if (fits) {
      for (uint32_t i = tid; i < cnt; i += 256)
        if (sel_match<KeyT>(keys[i], prefix, p)) atomicAdd(&lh[sel_digit<KeyT>(keys[i], p)], 1u);
    } else {                                           // massive ties: this workgroup streams the whole slice
      for (long i = tid; i < n; i += 256) {
        if (valid && !valid[i]) continue;
        const KeyT key = abs_key(r[i]);
        if (sel_match<KeyT>(key, prefix, p)) atomicAdd(&lh[sel_digit<KeyT>(key, p)], 1u);
      }
    }
    __syncthreads();
    for (int b = tid; b < 1024; b += 256) hists[p * SEL_BINS + b] = lh[b];     // what the full pass would have accumulated
    // the bin holding rank k_rem: 64 threads x 16 bins, prefix over the 64 partial sums by one thread
    if (tid < 64) {
      uint32_t s16 = 0;
      for (int j = 0; j < 16; ++j) s16 += lh[tid * 16 + j];
      part[tid] = s16;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      int t = 0;
      for (; t < 64; ++t) { if (k_rem < run + part[t]) break; run += part[t]; }
      uint32_t bin = 0, below = run;
      if (t < 64) {
        for (int j = 0; j < 16; ++j) {
          const uint32_t c = lh[t * 16 + j];
          if (k_rem < below + c) { bin = t * 16 + j; break; }
          below += c;
        }
      } else below = 0;
      found[0] = bin; found[1] = below;
    }
    __syncthreads();
    prefix |= ((KeyT)found[0]) << SelCfg<KeyT>::shift(p);
    k_rem -= found[1];
    __syncthreads();
  }
  if (tid == 0) *sel_cand_done(hists) = 1u;
}

// ---- multi-GPU double select: ONE exchange instead of three histogram all-reduces for digits 3, 4, 5 --------------------------
// After digits 0..2 (33 bits) are resolved GLOBALLY (three histogram all-reduces), every rank runs pass 3 with SEL_COLLECT |
// SEL_NOTAIL on its own slice: the keys that match the 33-bit prefix -- a handful -- land in the candidate scratch of the workspace.
//   select_cand_pack_kernel : scratch -> a contiguous record per segment (count | 0 | 512 keys), scratch cleared, the rank-LOCAL
//                             digit-3 histogram cleared (the merge writes the global one);
//   (the caller all-gathers the records: (world, nseg, SEL_CAND_WORDS) words)
//   select_cand_merge_kernel: every rank builds, from the union of all ranks' candidates, the histograms of digits 3, 4, 5 that
//                             three more all-reduced passes would have produced -- consumers resolve exactly as before.
// A rank with more than 512 candidates travels as (count, key) when they are all ONE value (massive ties: saturated / identical
// pixels -- the case that produces thousands of keys with 33 equal leading bits); more than 512 candidates of DIFFERENT values
// cannot be represented: the merge then clears the digit-0 histogram -- zero valid keys: the median reads NaN, the robust scale
// 0, the system is poisoned and the factorisation reports it.
__global__ __launch_bounds__(256) void select_cand_pack_kernel(uint32_t* __restrict__ hists, uint32_t* __restrict__ out) {
  hists += (long)blockIdx.x * 6 * SEL_BINS;
  out += (long)blockIdx.x * SEL_CAND_WORDS;
  const int tid = threadIdx.x;
  const uint32_t cnt = *sel_cand_count(hists);
  const uint32_t* kw = hists + 5 * SEL_BINS + 1024;
  // an overflowing list whose keys are all ONE value (stored ones included) travels as (count, that key): out[1] = 1
  __shared__ uint32_t differ;
  if (tid == 0) differ = 0u;
  __syncthreads();
  const unsigned long long kmax = *sel_cand_max(hists), kmin = ~*sel_cand_maxnot(hists);
  if (cnt > (uint32_t)SEL_CAND_CAP) {
    if (kmax != kmin) differ = 1u;
    for (int i = tid; i < SEL_CAND_CAP; i += 256)
      if (sel_cand_keys(hists)[i] != kmax) differ = 1u;
  }
  __syncthreads();
  const bool uniform = cnt > (uint32_t)SEL_CAND_CAP && !differ;
  for (int i = tid; i < 2 * SEL_CAND_CAP; i += 256) out[2 + i] = (i < 2 * (int)min(cnt, (uint32_t)SEL_CAND_CAP)) ? kw[i] : 0u;
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) hists[5 * SEL_BINS + 1024 + i] = 0u;
  for (int i = tid; i < SEL_BINS; i += 256) hists[3 * SEL_BINS + i] = 0u;
  if (tid == 0) {
    out[0] = cnt; out[1] = uniform ? 1u : 0u; *sel_cand_count(hists) = 0u;
    *sel_cand_max(hists) = 0ull; *sel_cand_maxnot(hists) = 0ull;
  }
}

__global__ __launch_bounds__(256) void select_cand_merge_kernel(uint32_t* __restrict__ hists, const uint32_t* __restrict__ gathered,
                                                                int world, int nseg_total, int seg0) {
  using KeyT = uint64_t;
  __shared__ SelScratch sc;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ uint32_t part[64];
  __shared__ uint32_t found[2];
  __shared__ uint32_t over;
  const int seg = blockIdx.x, tid = threadIdx.x;
  hists += (long)seg * 6 * SEL_BINS;
  if (tid == 0) over = 0u;
  __syncthreads();
  for (int r = tid; r < world; r += 256) {
    const uint32_t* rec = gathered + ((long)r * nseg_total + seg0 + seg) * SEL_CAND_WORDS;
    if (rec[0] > (uint32_t)SEL_CAND_CAP && rec[1] != 1u) over = 1u;      // (rec[1] == 1: rec[0] copies of ONE key)
  }
  __syncthreads();
  if (over) {                                            // not representable: zero valid keys (see above)
    for (int b = tid; b < SEL_BINS; b += 256) hists[b] = 0u;
    if (tid == 0) *sel_cand_done(hists) = 1u;
    return;
  }
  KeyT prefix; uint32_t k_rem, nv;
  sel_resolve<KeyT>(hists, 3, &sc, prefix, k_rem, nv);
  for (int p = 3; p < 6; ++p) {
    for (int b = tid; b < SEL_BINS; b += 256) lh[b] = 0;
    __syncthreads();
    for (int r = 0; r < world; ++r) {
      const uint32_t* rec = gathered + ((long)r * nseg_total + seg0 + seg) * SEL_CAND_WORDS;
      const uint32_t cnt = rec[0];
      const KeyT* keys = reinterpret_cast<const KeyT*>(rec + 2);
      if (rec[1] == 1u) {                                 // an overflowing list of one value
        if (tid == 0 && sel_match<KeyT>(keys[0], prefix, p)) atomicAdd(&lh[sel_digit<KeyT>(keys[0], p)], cnt);
        continue;
      }
      for (uint32_t i = tid; i < cnt; i += 256)
        if (sel_match<KeyT>(keys[i], prefix, p)) atomicAdd(&lh[sel_digit<KeyT>(keys[i], p)], 1u);
    }
    __syncthreads();
    const int nb = 1 << SelCfg<KeyT>::bits(p);
    for (int b = tid; b < nb; b += 256) hists[p * SEL_BINS + b] = lh[b];
    // the bin holding rank k_rem: 64 threads x 32 bins, prefix over the 64 partial sums by one thread
    if (tid < 64) {
      uint32_t s32 = 0;
      for (int j = 0; j < 32; ++j) s32 += lh[tid * 32 + j];
      part[tid] = s32;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      int t = 0;
      for (; t < 64; ++t) { if (k_rem < run + part[t]) break; run += part[t]; }
      uint32_t bin = 0, below = run;
      if (t < 64) {
        for (int j = 0; j < 32; ++j) {
          const uint32_t c = lh[t * 32 + j];
          if (k_rem < below + c) { bin = t * 32 + j; break; }
          below += c;
        }
      } else below = 0;
      found[0] = bin; found[1] = below;
    }
    __syncthreads();
    prefix |= ((KeyT)found[0]) << SelCfg<KeyT>::shift(p);
    k_rem -= found[1];
    __syncthreads();
  }
  if (tid == 0) *sel_cand_done(hists) = 1u;
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
  const int pass_flags = pass;
  pass &= 0xff;
  if (!r || !hists || n < 0 || nseg < 1 || pass < 0 || pass >= SelCfg<KeyT>::NPASS) return COMO_ERR_ARG;
  // double keys with the collecting protocol (flag on every pass): the tail launched with pass 3 produces digits 4 and 5
  if (sizeof(T) == 8 && (pass_flags & SEL_COLLECT) && pass >= 4) return COMO_OK;
  long blocks = (n + 255) / 256;
  if (blocks < 1) blocks = 1;
  // Few, fat workgroups: every workgroup ends with up to 2048 global atomics on the SAME histogram, which serialise per
  // address at the memory side (~15 ns each): measured 32 us with 2048 workgroups, 17 us with 512 (scalar loads).
  const long cap = (nseg >= 4) ? 64 : 256;
  blocks = (blocks + 3) / 4;                         // 4 elements per thread on the vector path
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  static const bool wide_ok = [] { const char* e = getenv("COMO_SEL_WIDE"); return !e || e[0] != '0'; }();
  // (long single slices only: the segmented per-keyframe medians run beside other kernels on a side stream, where fat workgroups
  // measured neutral to slightly negative)
  if (wide_ok && n >= (1L << 20) && blocks == cap && (n + 4095) / 4096 >= cap) {
    hipLaunchKernelGGL((select_hist_kernel<T, 1024>), dim3((unsigned)blocks, nseg), dim3(1024), 0, s, r, valid, n, hists, pass_flags);
  } else {
    hipLaunchKernelGGL((select_hist_kernel<T, 256>), dim3((unsigned)blocks, nseg), dim3(256), 0, s, r, valid, n, hists, pass_flags);
  }
  COMO_CHECK_LAUNCH();
  if (sizeof(T) == 8 && (pass_flags & SEL_COLLECT) && !(pass_flags & SEL_NOTAIL) && pass == 3) {
    hipLaunchKernelGGL(select_tail_kernel, dim3(nseg), dim3(256), 0, s, hists, (const double*)r, valid, n);
    COMO_CHECK_LAUNCH();
  }
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_abi_version(void) { return 2; }

// A failed stream capture (or any failed runtime call of the caller) leaves its code in the thread's "last error"; the
// launch check of the next como_* call would report it as that call's launch failure.  Returns the code it cleared.
int como_clear_last_error(void) { return (int)hipGetLastError(); }

// End a stream capture that went wrong (an operation the capture does not allow invalidates it, but the stream stays in
// capture mode -- and every later launch on it fails -- until hipStreamEndCapture is called).  Returns 1 if the stream was
// capturing, 0 if not; the partial graph, if any, is destroyed and the last error cleared.
int como_abort_capture(como_stream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) st = hipStreamCaptureStatusNone;
  int was = st != hipStreamCaptureStatusNone;
  if (was) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)stream, &g);
    if (g) (void)hipGraphDestroy(g);
  }
  (void)hipGetLastError();
  return was;
}

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
int como_select_cand_words(void) { return como::SEL_CAND_WORDS; }
int como_select_cand_pack(void* hists, int nseg, void* out, como_stream_t stream) {
  if (!hists || !out || nseg < 1) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::select_cand_pack_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, (uint32_t*)hists, (uint32_t*)out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_select_cand_merge(void* hists, int nseg, const void* gathered, int world, int nseg_total, int seg0, como_stream_t stream) {
  if (!hists || !gathered || nseg < 1 || world < 1 || seg0 < 0 || seg0 + nseg > nseg_total) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::select_cand_merge_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, (uint32_t*)hists,
                     (const uint32_t*)gathered, world, nseg_total, seg0);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
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
