// Exact device-wide k-th element (lower median, torch.median semantics) by MSB-first radix
// select over order-preserving integer keys of |r|.
//
// Reference semantics: sigma = 1.4826 * median(|r| over valid pixels of ALL pairs in the batch)
// (como/odom/backend/photo.py:124-128, frontend/photo_tracking.py:131-134,
//  frontend/two_frame_sfm.py:258-261).  A histogram *approximation* would change sigma and break
// pose parity, so the select is exact: 3 digit passes for float keys (11+11+10 bits), 6 for double.
//
// Layout: `hists` = NPASS x 2048 uint32 counters in device memory, zeroed before pass 0.
// Pass p histograms digit p of the keys whose higher digits equal the already-resolved prefix.
// No host round trip: every workgroup of a later kernel re-derives (prefix, k) from the
// finished histograms in its prologue (`resolve`), which costs one 8 KiB L2 read per pass.
// Multi-GPU: the per-rank histograms of a pass are summed with one all-reduce between passes.
#pragma once
#include "common.cuh"

namespace como {

constexpr int SEL_BINS = 2048;

template <typename KeyT> struct SelCfg;
template <> struct SelCfg<uint32_t> {
  static constexpr int NPASS = 3;
  __host__ __device__ static constexpr int shift(int p) { return p == 0 ? 21 : (p == 1 ? 10 : 0); }
  __host__ __device__ static constexpr int bits(int p) { return p == 2 ? 10 : 11; }
};
template <> struct SelCfg<uint64_t> {
  static constexpr int NPASS = 6;
  __host__ __device__ static constexpr int shift(int p) {
    return p == 0 ? 53 : p == 1 ? 42 : p == 2 ? 31 : p == 3 ? 20 : p == 4 ? 10 : 0;
  }
  __host__ __device__ static constexpr int bits(int p) { return p >= 4 ? 10 : 11; }
};

template <typename KeyT>
__device__ __forceinline__ uint32_t sel_digit(KeyT key, int p) {
  return (uint32_t)((key >> SelCfg<KeyT>::shift(p)) & (KeyT)((1u << SelCfg<KeyT>::bits(p)) - 1u));
}

// true if `key` agrees with `prefix` on all digits above digit p
template <typename KeyT>
__device__ __forceinline__ bool sel_match(KeyT key, KeyT prefix, int p) {
  if (p == 0) return true;
  const int sh = SelCfg<KeyT>::shift(p - 1);
  return (key >> sh) == (prefix >> sh);
}

struct SelScratch {           // LDS scratch for resolve(): 256-thread blocks
  uint32_t wave_tot[4];
  uint32_t found_bin;
  uint32_t found_below;
  uint32_t total;
};

// Cooperative (whole 256-thread block) resolution of digits 0..npass_done-1.
// Returns prefix (key bits resolved so far), k_rem (rank inside the remaining candidate set) and
// nvalid (sum of the pass-0 histogram).  All threads return identical values.
template <typename KeyT>
__device__ void sel_resolve(const uint32_t* __restrict__ hists, int npass_done, SelScratch* sc,
                            KeyT& prefix, uint32_t& k_rem, uint32_t& nvalid) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  prefix = 0;
  k_rem = 0;
  nvalid = 0;
  for (int p = 0; p < npass_done; ++p) {
    const uint32_t* h = hists + p * SEL_BINS;
    uint32_t c[8];
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // written by earlier kernels (or an all-reduce) -> plain loads are coherent across launches
      c[j] = h[tid * 8 + j];
      local += c[j];
    }
    // inclusive scan of `local` across the block
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) sc->wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wv; ++w) base += sc->wave_tot[w];
    const uint32_t tot = sc->wave_tot[0] + sc->wave_tot[1] + sc->wave_tot[2] + sc->wave_tot[3];
    if (p == 0) {
      nvalid = tot;
      k_rem = (tot > 0) ? (tot - 1) / 2 : 0;   // lower median index
    }
    const uint32_t excl = base + incl - local;
    if (tid == 0) { sc->found_bin = 0; sc->found_below = 0; }
    __syncthreads();
    if (local > 0 && k_rem >= excl && k_rem < excl + local) {
      uint32_t run = excl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (k_rem >= run && k_rem < run + c[j]) { sc->found_bin = tid * 8 + j; sc->found_below = run; }
        run += c[j];
      }
    }
    __syncthreads();
    prefix |= ((KeyT)sc->found_bin) << SelCfg<KeyT>::shift(p);
    k_rem -= sc->found_below;
    __syncthreads();
  }
}

// The same resolution for a block of NT threads (NT = 64, 128 or 256): thread t owns the 2048 / NT consecutive bins
// [t * PER, (t + 1) * PER).
template <typename KeyT, int NT>
__device__ void sel_resolve_n(const uint32_t* __restrict__ hists, int npass_done, SelScratch* sc,
                              KeyT& prefix, uint32_t& k_rem, uint32_t& nvalid) {
  constexpr int PER = SEL_BINS / NT, NWV = NT / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  prefix = 0;
  k_rem = 0;
  nvalid = 0;
  for (int p = 0; p < npass_done; ++p) {
    const uint32_t* h = hists + p * SEL_BINS;
    uint32_t local = 0;
    for (int j = 0; j < PER; ++j) local += h[tid * PER + j];
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) sc->wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < NWV; ++w) { if (w < wv) base += sc->wave_tot[w]; tot += sc->wave_tot[w]; }
    if (p == 0) {
      nvalid = tot;
      k_rem = (tot > 0) ? (tot - 1) / 2 : 0;
    }
    const uint32_t excl = base + incl - local;
    if (tid == 0) { sc->found_bin = 0; sc->found_below = 0; }
    __syncthreads();
    if (local > 0 && k_rem >= excl && k_rem < excl + local) {
      uint32_t run = excl;
      for (int j = 0; j < PER; ++j) {
        const uint32_t cj = h[tid * PER + j];
        if (k_rem >= run && k_rem < run + cj) { sc->found_bin = tid * PER + j; sc->found_below = run; }
        run += cj;
      }
    }
    __syncthreads();
    prefix |= ((KeyT)sc->found_bin) << SelCfg<KeyT>::shift(p);
    k_rem -= sc->found_below;
    __syncthreads();
  }
}

// The same resolution for a block of NW waves of which only the first NT threads own bins (NT = 64, 128 or 256; NW * 64 >= NT):
// blocks whose thread count does not divide 2048 (the 192-thread wave-specialised block kernel).
template <typename KeyT, int NT, int NW>
__device__ void sel_resolve_part(const uint32_t* __restrict__ hists, int npass_done, SelScratch* sc,
                                 KeyT& prefix, uint32_t& k_rem, uint32_t& nvalid) {
  static_assert(NW <= 4 && NT <= NW * 64, "SelScratch holds four wave totals");
  constexpr int PER = SEL_BINS / NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const bool own = tid < NT;
  prefix = 0;
  k_rem = 0;
  nvalid = 0;
  for (int p = 0; p < npass_done; ++p) {
    const uint32_t* h = hists + p * SEL_BINS;
    uint32_t local = 0;
    if (own)
      for (int j = 0; j < PER; ++j) local += h[tid * PER + j];
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) sc->wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < NW; ++w) { if (w < wv) base += sc->wave_tot[w]; tot += sc->wave_tot[w]; }
    if (p == 0) {
      nvalid = tot;
      k_rem = (tot > 0) ? (tot - 1) / 2 : 0;
    }
    const uint32_t excl = base + incl - local;
    if (tid == 0) { sc->found_bin = 0; sc->found_below = 0; }
    __syncthreads();
    if (own && local > 0 && k_rem >= excl && k_rem < excl + local) {
      uint32_t run = excl;
      for (int j = 0; j < PER; ++j) {
        const uint32_t cj = h[tid * PER + j];
        if (k_rem >= run && k_rem < run + cj) { sc->found_bin = tid * PER + j; sc->found_below = run; }
        run += cj;
      }
    }
    __syncthreads();
    prefix |= ((KeyT)sc->found_bin) << SelCfg<KeyT>::shift(p);
    k_rem -= sc->found_below;
    __syncthreads();
  }
}

// The same resolution inside a block of MORE than 256 threads (the 1024-thread streaming passes): the first 256 threads own the
// bins, every thread takes part in the barriers and returns the same values.
template <typename KeyT>
__device__ void sel_resolve_wide(const uint32_t* __restrict__ hists, int npass_done, SelScratch* sc,
                                 KeyT& prefix, uint32_t& k_rem, uint32_t& nvalid) {
  constexpr int PER = SEL_BINS / 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const bool own = tid < 256;
  prefix = 0;
  k_rem = 0;
  nvalid = 0;
  for (int p = 0; p < npass_done; ++p) {
    const uint32_t* h = hists + p * SEL_BINS;
    uint32_t local = 0;
    if (own)
      for (int j = 0; j < PER; ++j) local += h[tid * PER + j];
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63 && wv < 4) sc->wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < wv) base += sc->wave_tot[w]; tot += sc->wave_tot[w]; }
    if (p == 0) {
      nvalid = tot;
      k_rem = (tot > 0) ? (tot - 1) / 2 : 0;
    }
    const uint32_t excl = base + incl - local;
    if (tid == 0) { sc->found_bin = 0; sc->found_below = 0; }
    __syncthreads();
    if (own && local > 0 && k_rem >= excl && k_rem < excl + local) {
      uint32_t run = excl;
      for (int j = 0; j < PER; ++j) {
        const uint32_t cj = h[tid * PER + j];
        if (k_rem >= run && k_rem < run + cj) { sc->found_bin = tid * PER + j; sc->found_below = run; }
        run += cj;
      }
    }
    __syncthreads();
    prefix |= ((KeyT)sc->found_bin) << SelCfg<KeyT>::shift(p);
    k_rem -= sc->found_below;
    __syncthreads();
  }
}

// LDS histogram increment of the fused digit-0 pass.  Plain ds_add: the keys of a wave fall into ~10-30 exponent bins, the
// hardware resolves the few-way conflicts in a handful of cycles; a ballot loop that issued one add per DISTINCT bin
// cost ~150 instructions per wave and was half of the residual kernel (73 -> 43 us).
__device__ __forceinline__ void sel_lds_add(uint32_t* lh, uint32_t bin, bool active) {
  if (active) atomicAdd(&lh[bin], 1u);
}

// Accumulate a block-local LDS histogram into the global one (skipping empty bins).
__device__ __forceinline__ void sel_flush(const uint32_t* lds_hist, uint32_t* __restrict__ ghist) {
  for (int b = threadIdx.x; b < SEL_BINS; b += blockDim.x) {
    uint32_t v = lds_hist[b];
    if (v) atomicAdd(&ghist[b], v);
  }
}

// Workspace clears are KERNELS, never hipMemsetAsync: a memset node captured into a hipGraph was observed (ROCm 7.2, gfx950)
// to fill with a stale 64-bit pattern on later replays once unrelated eager work had run in between -- the node's fill
// parameters are not pinned with the graph.  A kernel node carries its arguments by value.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void sel_zero_kernel(uint32_t* __restrict__ p, size_t nwords) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < nwords) {
    *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (size_t k = i; k < nwords; ++k) p[k] = 0u;
  }
}

// p must be 16-byte aligned (every workspace here is a whole allocation or a multiple of 6 * SEL_BINS words into one)
inline bool zero_words(void* p, size_t nwords, hipStream_t s) {
  if (nwords == 0) return true;
  const unsigned blocks = (unsigned)((nwords + 1023) / 1024);
  hipLaunchKernelGGL(sel_zero_kernel<0>, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, nwords);
  return hipGetLastError() == hipSuccess;
}

}  // namespace como
