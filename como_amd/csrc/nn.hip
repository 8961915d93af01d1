// DepthCov covariance network (float32 inference), gfx950.
//
// Reference: como/depth_cov/nn/UNet.py:8-78 (UNet.forward), como/depth_cov/nn/layers.py:5-75 (ResidualConv, DownConv,
// UpConv), como/depth_cov/core/DepthCovModule.py:80-87, como/depth_cov/core/gaussian_kernel.py:6-49 (output
// activation), como/odom/Mapping.py:409-428 (run_model: antialiased bilinear resize in and out).
//
// conv2d (3x3 / 1x1, stride 1, "same" zero padding) is an implicit GEMM on v_mfma_f32_16x16x4_f32:
//   D[cout, px] += W[cout, k] * patch[k, px],   k = (ky, kx, cin)
//   A operand: lane (l & 15 = cout, l >> 4 = k)  <- weights pre-transposed to [ky][kx][cin][cout] (cout contiguous)
//   B operand: lane (l & 15 = px,   l >> 4 = k)  <- NCHW input: 16 consecutive pixels of 4 channel planes per step
//   D        : lane (l & 15 = px, rows 4 (l >> 4) + r = cout) -> NCHW stores of 16 consecutive pixels per cout
// One wave owns 64 consecutive (linearised) pixels x 16*MT output channels.  The whole network is ~3 GFLOP at
// 192x256: it is launch-count bound, not MFMA bound; the kernels are kept simple and exact-shape.
#include "common.cuh"
#include "../../include/como_hip.h"
#include <cstdlib>

namespace como {

typedef float nf4 __attribute__((ext_vector_type(4)));

// WV waves per workgroup share one output tile (64 pixels x 16*MT channels) and split the input channels between them
// (split-K): the deep levels have 48..768 pixels and up to 512 x 9 reduction steps, far too few tiles to fill 1024 SIMDs
// otherwise (one wave did 1152 dependent load->MFMA steps: 430 us for a 226 MFLOP layer).  WV = 1 is the plain mapping
// for the wide levels.  Loads are unconditional on clamped addresses and the channel loop is unrolled so that several
// steps' loads are in flight per wave.  Optional epilogue: per-channel sum / sum of squares of the outputs accumulated
// into gn_sums (N, G, 2) doubles -- the GroupNorm statistics of the next layer, without a separate pass over the tensor.
// Round 5 -- the network built around FUSED layers instead of one kernel per torch module:
//   PRO: the input is read through the GroupNorm + LeakyReLU that precedes the convolution (layers.py:21: act(norm(conv1(x))) feeds
//        conv2): x' = lrelu(x * sc[c] + sh[c]) with the per-channel scale / shift the PRODUCER of x left behind (gn_finalize /
//        deep_reduce below) -- the normalised tensor is never written or read;
//   res: the epilogue of ResidualConv's 1x1 skip convolution adds the normalised main branch and applies the activation
//        (layers.py:22-24: act(conv3(x) + norm(conv2(y)))): out = lrelu(conv + bias + res * rsc[c] + rsh[c]).
// A ResidualConv is three launches (conv1, conv2 with PRO, conv3 with res) instead of five, and two tensor round trips shorter.
// Round 6: the scale / shift of the GroupNorm that follows a wide-level convolution are formed INSIDE that convolution by its last
// wave (gn_finalize_kernel's arithmetic, value for value) -- the separate one-workgroup launch was 4.7 us of dependent latency per
// normalisation, twelve times per forward.  Protocol: a wave adds its statistics with RETURNING agent-scope atomics (the returned
// value comes from the memory side, where the read-modify-write was performed: once the wave holds it the add is globally done),
// then arrives on a counter that lives right behind the 32 slots (zeroed with them); the wave that finds total - 1 earlier
// arrivals reads every slot with agent-scope loads (first touch of those lines by this kernel: nothing stale in an XCD's L2).
struct GnFin {
  const float* gamma;
  const float* beta;
  float2* scsh;         // nullptr: no in-kernel finalisation (como_nn_gn_finalize_f32 does it)
  float eps;
};
__device__ __forceinline__ void gn_arrive_finalize(double* __restrict__ gn_sums, int N, int C, int G, int HW, unsigned total,
                                                   const GnFin& f, int lane) {
  unsigned* cnt = reinterpret_cast<unsigned*>(gn_sums + (long)32 * N * G * 2);
  unsigned prev = 0;
  if (lane == 0) prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  prev = __shfl(prev, 0, 64);
  if (prev + 1u != total) return;
  const double cntd = (double)(C / G) * (double)HW;
  for (int i = lane; i < N * C; i += 64) {
    const int n = i / C, ch = i - n * C, g = ch / (C / G);
    double s1 = 0.0, s2 = 0.0;
    for (int sl = 0; sl < 32; ++sl) {
      const double* src = gn_sums + (((long)sl * N + n) * G + g) * 2;
      s1 += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double m = s1 / cntd;
    double var = s2 / cntd - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = rs * f.gamma[ch];
    f.scsh[i] = float2{sc, f.beta[ch] - mf * sc};
  }
}
// a wave's share of a group's statistics; `returning`: wait for the memory side's answer (see above)
__device__ __forceinline__ void gn_add(double* dst, double s1, double s2, bool returning) {
  if (returning) {
    const double o1 = __hip_atomic_fetch_add(dst, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double o2 = __hip_atomic_fetch_add(dst + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(o1), "v"(o2));
  } else {
    atomicAdd(dst, s1);
    atomicAdd(dst + 1, s2);
  }
}

template <int KS, int MT, int WV, bool PRO = false>
__global__ __launch_bounds__(64 * WV) void conv_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                            int CinP, int Cout, int H, int W, int out_ctot, int out_coff,
                                                            double* __restrict__ gn_sums, int gn_groups,
                                                            const float2* __restrict__ pro_scsh = nullptr,
                                                            const float* __restrict__ res = nullptr,
                                                            const float2* __restrict__ res_scsh = nullptr, float slope = 0.f,
                                                            GnFin fin = GnFin{nullptr, nullptr, nullptr, 0.f}) {
  constexpr int GN_SLOTS = 32;
  constexpr int PAD = KS / 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
  const int HW = H * W;
  const int p0 = blockIdx.x * 64;
  const int co0 = blockIdx.y * 16 * MT;
  const int n = blockIdx.z;
  const float* inb = in + (long)n * Cin * HW;
  int py[4], px[4];
  bool pv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = p0 + 16 * t + c;
    pv[t] = p < HW;
    const int pc = pv[t] ? p : HW - 1;
    py[t] = pc / W;
    px[t] = pc - py[t] * W;
  }
  nf4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = nf4{0.f, 0.f, 0.f, 0.f};
  // this wave's channel slice [cbeg, cend) in steps of 4
  const int steps = CinP >> 2;
  const int per = (steps + WV - 1) / WV;
  const int cbeg = 4 * min(steps, wv * per), cend = 4 * min(steps, (wv + 1) * per);
  int wco[MT];
  float wokm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { const int co = co0 + 16 * mt + c; wokm[mt] = co < Cout ? 1.f : 0.f; wco[mt] = co < Cout ? co : 0; }
#pragma unroll 1
  for (int kk = 0; kk < KS * KS; ++kk) {
    const int ky = kk / KS, kx = kk - ky * KS;
    // padding / tail handling by MULTIPLYING with a 0/1 mask: with a select the compiler sinks every load into its own
    // exec-masked branch followed by s_waitcnt vmcnt(0) -- one exposed memory latency per load (1.6 us per step)
    int off[4];
    float okm[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int yy = py[t] + ky - PAD, xx = px[t] + kx - PAD;
      const bool ok = pv[t] && yy >= 0 && yy < H && xx >= 0 && xx < W;
      okm[t] = ok ? 1.f : 0.f;
      off[t] = ok ? yy * W + xx : 0;
    }
    const float* wk = wt + (long)kk * CinP * Cout;
    // Round 4, measured and NOT adopted (1.04 ms per forward stays): (i) a whole kernel ROW of taps per load batch (12 steps in
    // flight instead of 4): 0.95 ms, results equal to 4e-7 -- but a different summation order, enough to move the greedy sampler's
    // picks on the seeded random weights (the two-frame initialisation of one bench sequence went from frame 3 to frame 51:
    // nothing wrong, but no longer round 3's validated behaviour); (ii) double-buffered batches in THIS order (bit-identical
    // results): 1.12 ms -- the per-batch tap / offset bookkeeping costs more than the overlap gains.
    constexpr int U = 4;                                   // reduction steps whose loads are in flight together
    int ci0 = cbeg;
    for (; ci0 + 4 * U <= cend; ci0 += 4 * U) {
      float a[U][MT], b[U][4];
      float2 ss[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ci = ci0 + 4 * u + q;                    // ci < CinP: weight rows exist (zero rows beyond Cin)
        const int cic = min(ci, Cin - 1);
        const float* ip = inb + (long)cic * HW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[u][mt] = wk[(long)ci * Cout + wco[mt]];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[u][t] = ip[off[t]];
        if (PRO) ss[u] = pro_scsh[(long)n * Cin + cic];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float bv = b[u][t];
            if (PRO) { bv = __builtin_fmaf(bv, ss[u].x, ss[u].y); bv = bv > 0.f ? bv : bv * slope; }
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt] * wokm[mt], bv * okm[t], acc[mt][t], 0, 0, 0);
          }
    }
    for (; ci0 < cend; ci0 += 4) {
      const int ci = ci0 + q;
      const int cic = min(ci, Cin - 1);
      const float* ip = inb + (long)cic * HW;
      float2 s1 = {1.f, 0.f};
      if (PRO) s1 = pro_scsh[(long)n * Cin + cic];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float av = wk[(long)ci * Cout + wco[mt]] * wokm[mt];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float bv = ip[off[t]];
          if (PRO) { bv = __builtin_fmaf(bv, s1.x, s1.y); bv = bv > 0.f ? bv : bv * slope; }
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv * okm[t], acc[mt][t], 0, 0, 0);
        }
      }
    }
  }
  if constexpr (WV > 1) {
    // fixed-order cross-wave reduction through LDS (wave 0 adds the slices of waves 1..WV-1 in order)
    __shared__ float red[(WV - 1) * MT * 4 * 4 * 64];
    if (wv > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((((wv - 1) * MT + mt) * 4 + t) * 4 + r) * 64 + lane] = acc[mt][t][r];
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll 1
    for (int w2 = 0; w2 < WV - 1; ++w2)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][t][r] += red[(((w2 * MT + mt) * 4 + t) * 4 + r) * 64 + lane];
  }
  float* ob = out + ((long)n * out_ctot + out_coff) * HW;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * mt + 4 * q + r;
      const bool cok = co < Cout;
      const float bv = (bias && cok) ? bias[co] : 0.f;
      float2 rs = {0.f, 0.f};
      if (res && cok) rs = res_scsh[(long)n * Cout + co];
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int p = p0 + 16 * t + c;
        if (cok && p < HW) {
          float v = acc[mt][t][r] + bv;
          if (res) {
            v += __builtin_fmaf(res[((long)n * Cout + co) * HW + p], rs.x, rs.y);
            v = v > 0.f ? v : v * slope;
          }
          ob[(long)co * HW + p] = v;
          s1 += (double)v;
          s2 += (double)v * (double)v;
        }
      }
      if (gn_sums) {
        // 16 lanes (c) hold the same channel: reduce them, one atomic pair per (wave, channel)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 16); s2 += __shfl_xor(s2, o, 16); }
        if (c == 0 && cok) {
          // 32 slots per statistic: thousands of waves adding to the same 2 x G addresses serialise at the memory side
          const int g = co / (Cout / gn_groups);
          const int slot = (blockIdx.x + blockIdx.y) % GN_SLOTS;
          double* dst = gn_sums + (((long)slot * gridDim.z + n) * gn_groups + g) * 2;
          gn_add(dst, s1, s2, fin.scsh != nullptr);
        }
      }
    }
  // (one wave per workgroup reaches this point: the channel-slice waves have left above)
  if (gn_sums && fin.scsh)
    gn_arrive_finalize(gn_sums, (int)gridDim.z, Cout, gn_groups, HW, gridDim.x * gridDim.y * gridDim.z, fin, lane);
}

// (sum, sum of squares) the producing convolution accumulated (32 contention slots) -> per-channel scale / shift of the GroupNorm
// that follows: y = x * sc[c] + sh[c], sc = rstd_g gamma_c, sh = beta_c - mean_g sc  (biased variance, as nn.GroupNorm).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int N, int C, int G, int HW, float eps,
                                                          float2* __restrict__ scsh) {
  __shared__ float sm[2 * 256];
  for (int idx = threadIdx.x; idx < N * G; idx += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int sl = 0; sl < 32; ++sl) {
      s1 += sums[((long)sl * N * G + idx) * 2];
      s2 += sums[((long)sl * N * G + idx) * 2 + 1];
    }
    const double cnt = (double)(C / G) * (double)HW;
    const double m = s1 / cnt;
    double var = s2 / cnt - m * m;
    if (var < 0.0) var = 0.0;
    sm[2 * idx] = (float)m;
    sm[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * C; i += 256) {
    const int n = i / C, ch = i - n * C, g = ch / (C / G);
    const float sc = sm[2 * (n * G + g) + 1] * gamma[ch];
    scsh[i] = float2{sc, beta[ch] - sm[2 * (n * G + g)] * sc};
  }
}

// ---- the WIDE levels (48x64 .. 192x256): 3x3 layers with the input tile staged in LDS ----
// The generic kernel reads every input value nine times from L2 through per-lane gathers, behind which its matrix instructions
// wait: a 16 -> 16 layer at 192x256 is 768 one-wave workgroups, each a chain of nine load -> matrix batches (21 us for 226 MFLOP).
// Here a workgroup (4 waves) owns a TH x 32 pixel tile and 16 output channels: the (TH + 2) x 34 halo tile of EVERY input channel
// is fetched once, coalesced, all loads in flight together -- through the preceding GroupNorm + LeakyReLU when PRO, so the
// normalisation is applied once per value, not once per tap -- into LDS planes whose stride makes the matrix operand reads
// conflict-free (lane (x, q) reads plane 4 step + q at x: plane stride = 16 or 48 mod 64 words); the reduction loop then touches
// never touches global memory: the workgroup's 16 columns of the weights are staged too.  TH x KW = 8: the four waves are TH / 2
// pixel groups (two rows of 32 pixels each) x KW slices of the input channels (summed in LDS in a fixed order): 4 x 2 at 192x256,
// 2 x 4 at 96x128 and 48x64.  Measured: 21 -> 14-17 us per layer; what is left is the chain stage -> barrier -> reduction ->
// epilogue on one 4-wave workgroup per compute unit (each piece one round trip to the memory side: the tensors come from another XCD).
template <int TH, bool PRO>
__global__ __launch_bounds__(256) void conv3_tile_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, float* __restrict__ out, int Cin, int CinP,
                                                         int Cout, int H, int W, int out_ctot, int out_coff,
                                                         double* __restrict__ gn_sums, int gn_groups,
                                                         const float2* __restrict__ pro_scsh, float slope, GnFin fin) {
  constexpr int KW = 8 / TH, R = TH + 2, LW = TH == 2 ? 44 : 40, PS = R * LW, GN_SLOTS = 32;   // PS = 16 or 48 mod 64; interior at column 4
  extern __shared__ float tile[];                              // [CinP][R][LW]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, q = lane >> 4;
  const int pg = wv / KW, kw = wv - pg * KW;                   // pixel group (rows 2 pg, 2 pg + 1 of the tile), channel slice
  const int HW = H * W;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TH;
  const int ntc = (Cout + 15) / 16;
  const int n = blockIdx.z / ntc, co0 = (blockIdx.z - n * ntc) * 16;
  const float* inb = in + (long)n * Cin * HW;
  // ---- stage the halo tile of all channels (zero padding outside the image and beyond Cin) and this workgroup's 16 columns of
  //      the weights ([9][CinP][16]: lane (c, q) reads word 16 q + c of a 64-word row -- conflict-free) ----
  float* wl = tile + CinP * PS;
  // (every phase's FIRST round of loads is issued before anything is written to LDS: the tensors were just produced by another
  // kernel, usually on another XCD -- each dependent phase costs a ~2 us round trip to the memory side, and a small layer is three
  // phases of one round each)
  const int EV = CinP * R * 8, EH = CinP * R * 2, EW = 9 * CinP * 4;
  const bool vec = (Cout & 15) == 0;
  constexpr int SB = 8;
  auto load_in = [&](int e0, float4 (&v)[SB], int (&dst)[SB]) {
    // interior: rows of 32 floats = eight 16-byte vectors, 16-byte aligned in memory (W, x0 multiples of 32) and in LDS (column 4)
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int e = min(e0 + 256 * u, EV - 1);
      const int ci = e / (R * 8), rem = e - ci * (R * 8), r = rem >> 3, j = rem & 7;
      const int yy = y0 + r - 1;
      const bool ok = ci < Cin && yy >= 0 && yy < H;
      dst[u] = e0 + 256 * u < EV ? ci * PS + r * LW + 4 + 4 * j : -1;
      const int cic = min(ci, Cin - 1), yc = min(max(yy, 0), H - 1);
      float4 x = *reinterpret_cast<const float4*>(&inb[(long)cic * HW + (long)yc * W + x0 + 4 * j]);
      if (PRO) {
        const float2 ss = pro_scsh[(long)n * Cin + cic];
        x.x = __builtin_fmaf(x.x, ss.x, ss.y); x.x = x.x > 0.f ? x.x : x.x * slope;
        x.y = __builtin_fmaf(x.y, ss.x, ss.y); x.y = x.y > 0.f ? x.y : x.y * slope;
        x.z = __builtin_fmaf(x.z, ss.x, ss.y); x.z = x.z > 0.f ? x.z : x.z * slope;
        x.w = __builtin_fmaf(x.w, ss.x, ss.y); x.w = x.w > 0.f ? x.w : x.w * slope;
      }
      const float m = ok ? 1.f : 0.f;
      v[u] = float4{x.x * m, x.y * m, x.z * m, x.w * m};
    }
  };
  auto load_halo = [&](int e0, float (&v)[4], int (&dst)[4]) {      // the two halo columns (tile columns 3 and 36)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 256 * u, EH - 1);
      const int ci = e / (R * 2), rem = e - ci * (R * 2), r = rem >> 1, side = rem & 1;
      const int yy = y0 + r - 1, xx = side ? x0 + 32 : x0 - 1;
      const bool ok = ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W;
      dst[u] = e0 + 256 * u < EH ? ci * PS + r * LW + (side ? 36 : 3) : -1;
      const int cic = min(ci, Cin - 1), yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
      float x = inb[(long)cic * HW + (long)yc * W + xc];
      if (PRO) {
        const float2 ss = pro_scsh[(long)n * Cin + cic];
        x = __builtin_fmaf(x, ss.x, ss.y);
        x = x > 0.f ? x : x * slope;
      }
      v[u] = x * (ok ? 1.f : 0.f);
    }
  };
  auto load_w = [&](int e0, float4 (&v)[4]) {      // 16 columns of the weights: four 16-byte vectors per (tap, channel) row
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 256 * u, EW - 1), row = e >> 2, j = e & 3;
      if (vec) {
        v[u] = *reinterpret_cast<const float4*>(&wt[(long)row * Cout + co0 + 4 * j]);
      } else {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int cc = co0 + 4 * j + k; t[k] = wt[(long)row * Cout + min(cc, Cout - 1)] * (cc < Cout ? 1.f : 0.f); }
        v[u] = float4{t[0], t[1], t[2], t[3]};
      }
    }
  };
  {
    float4 vi[SB], vw[4];
    float vh[4];
    int di[SB], dh[4];
    load_in(tid, vi, di);
    load_halo(tid, vh, dh);
    load_w(tid, vw);
#pragma unroll
    for (int u = 0; u < SB; ++u)
      if (di[u] >= 0) *reinterpret_cast<float4*>(&tile[di[u]]) = vi[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dh[u] >= 0) tile[dh[u]] = vh[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (tid + 256 * u < EW) *reinterpret_cast<float4*>(&wl[4 * (tid + 256 * u)]) = vw[u];
    for (int e0 = tid + 256 * SB; e0 < EV; e0 += 256 * SB) {
      load_in(e0, vi, di);
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (di[u] >= 0) *reinterpret_cast<float4*>(&tile[di[u]]) = vi[u];
    }
    for (int e0 = tid + 256 * 4; e0 < EH; e0 += 256 * 4) {
      load_halo(e0, vh, dh);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dh[u] >= 0) tile[dh[u]] = vh[u];
    }
    for (int e0 = tid + 256 * 4; e0 < EW; e0 += 256 * 4) {
      load_w(e0, vw);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + 256 * u < EW) *reinterpret_cast<float4*>(&wl[4 * (e0 + 256 * u)]) = vw[u];
    }
  }
  __syncthreads();
  // ---- the reduction: this wave's pixel group x its slice of the channels, all nine taps -- LDS and the matrix pipe only ----
  const int steps = CinP >> 2, per = (steps + KW - 1) / KW;
  const int sbeg = min(steps, kw * per), send = min(steps, (kw + 1) * per);
  float bvr[4];                                                // (the epilogue's bias: fetched now, not after the reduction)
#pragma unroll
  for (int r = 0; r < 4; ++r) bvr[r] = bias ? bias[min(co0 + 4 * q + r, Cout - 1)] : 0.f;
  nf4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = nf4{0.f, 0.f, 0.f, 0.f};
  // operand t: row 2 pg + (t >> 1) of the tile, columns 16 (t & 1) + c
  const float* tb = tile + (2 * pg) * LW + c + q * PS;
  const float* wb = wl + 16 * q + c;
#pragma unroll 1
  for (int kk = 0; kk < 9; ++kk) {
    const int ky = kk / 3, kx = kk - 3 * ky;
    const float* wk = wb + kk * CinP * 16;
    const float* tk = tb + ky * LW + kx + 3;                   // (interior at column 4: tap kx - 1 of pixel x reads column 3 + x + kx)
    constexpr int U = 4;
    int st = sbeg;
    for (; st + U <= send; st += U) {
      float a[U], b[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = wk[64 * (st + u)];
        const float* tp = tk + 4 * (st + u) * PS;
        b[u][0] = tp[0]; b[u][1] = tp[16]; b[u][2] = tp[LW]; b[u][3] = tp[LW + 16];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][t], acc[t], 0, 0, 0);
    }
    for (; st < send; ++st) {
      const float av = wk[64 * st];
      const float* tp = tk + 4 * st * PS;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tp[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tp[16], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tp[LW], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tp[LW + 16], acc[3], 0, 0, 0);
    }
  }
  if (KW > 1) {                                                // channel slices: summed by slice 0 in slice order (over the dead tile)
    __syncthreads();
    float* red = tile;
    if (kw > 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((pg * (KW - 1) + kw - 1) * 4 + t) * 4 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (kw > 0) return;
#pragma unroll 1
    for (int k2 = 0; k2 < KW - 1; ++k2)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[(((pg * (KW - 1) + k2) * 4 + t) * 4 + r) * 64 + lane];
  }
  float* ob = out + ((long)n * out_ctot + out_coff) * HW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cor = co0 + 4 * q + r;
    const bool cok = cor < Cout;
    const float bv = cok ? bvr[r] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int yy = y0 + 2 * pg + (t >> 1), xx = x0 + 16 * (t & 1) + c;
      if (cok && yy < H && xx < W) {
        const float v = acc[t][r] + bv;
        ob[(long)cor * HW + (long)yy * W + xx] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
    if (gn_sums) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 16); s2 += __shfl_xor(s2, o, 16); }
      if (c == 0 && cok) {
        const int g = cor / (Cout / gn_groups);
        const int slot = (blockIdx.x + blockIdx.y + wv) % GN_SLOTS;
        double* dst = gn_sums + (((long)slot * (gridDim.z / ntc) + n) * gn_groups + g) * 2;
        gn_add(dst, s1, s2, fin.scsh != nullptr);
      }
    }
  }
  // (the TH / 2 waves of channel slice 0 reach this point, one arrival each)
  if (gn_sums && fin.scsh)
    gn_arrive_finalize(gn_sums, (int)(gridDim.z / ntc), Cout, gn_groups, HW, gridDim.x * gridDim.y * gridDim.z * (TH / 2), fin, lane);
}

// ---- the DEEP levels (24x32 and below: <= 768 pixels, 128 .. 512 channels, 0.3 .. 9.4 MB of weights per layer) ----
// The generic kernel maps a layer to (pixel tiles) x (Cout / 16) workgroups: 32 workgroups at 6x8, each a chain of 72 dependent
// load -> matrix batches behind one another -- 36 us for 226 MFLOP, on 32 of 256 compute units.  Here a 3x3 layer is cut along its
// REDUCTION dimension as well: workgroup (pixel tile, 16 output channels, channel slice) = 9 waves, one per kernel tap (its
// padding mask and offsets are computed once), each walking its slice of the input channels with 8 steps' loads in flight; the nine
// taps are summed in LDS in a fixed order, the channel slices leave partial sums, and deep_reduce_kernel adds them in a fixed order
// (+ bias), writes the layer's output, its GroupNorm statistics and the scale / shift of the normalisation that follows -- one
// workgroup per (sample, group): no atomics anywhere, the same bits every run.  512 / 384 / 384 workgroups at 6x8 / 12x16 / 24x32.
template <bool PRO>
__global__ __launch_bounds__(576) void conv3_deep_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                         float* __restrict__ part, int Cin, int CinP, int Cout, int H, int W,
                                                         int nslice, const float2* __restrict__ pro_scsh, float slope) {
  const int lane = threadIdx.x & 63, kk = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
  const int HW = H * W;
  const int p0 = blockIdx.x * 64, co0 = blockIdx.y * 16;
  const int n = blockIdx.z / nslice, slice = blockIdx.z - n * nslice;
  const float* inb = in + (long)n * Cin * HW;
  const int ky = kk / 3, kx = kk - 3 * ky;
  int off[4];
  float okm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = p0 + 16 * t + c;
    const int pc = p < HW ? p : HW - 1;
    const int y = pc / W, x = pc - y * W;
    const int yy = y + ky - 1, xx = x + kx - 1;
    const bool ok = p < HW && yy >= 0 && yy < H && xx >= 0 && xx < W;
    okm[t] = ok ? 1.f : 0.f;
    off[t] = ok ? yy * W + xx : 0;
  }
  const int co = co0 + c;
  const float wok = co < Cout ? 1.f : 0.f;
  const int wco = co < Cout ? co : 0;
  const int steps = CinP >> 2, per = (steps + nslice - 1) / nslice;
  const int cbeg = 4 * min(steps, slice * per), cend = 4 * min(steps, (slice + 1) * per);
  const float* wk = wt + (long)kk * CinP * Cout;
  nf4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = nf4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  int ci0 = cbeg;
  for (; ci0 + 4 * U <= cend; ci0 += 4 * U) {
    float a[U], b[U][4];
    float2 ss[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ci = ci0 + 4 * u + q, cic = min(ci, Cin - 1);
      const float* ip = inb + (long)cic * HW;
      a[u] = wk[(long)ci * Cout + wco];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[u][t] = ip[off[t]];
      if (PRO) ss[u] = pro_scsh[(long)n * Cin + cic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float bv = b[u][t];
        if (PRO) { bv = __builtin_fmaf(bv, ss[u].x, ss[u].y); bv = bv > 0.f ? bv : bv * slope; }
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u] * wok, bv * okm[t], acc[t], 0, 0, 0);
      }
  }
  for (; ci0 < cend; ci0 += 4) {
    const int ci = ci0 + q, cic = min(ci, Cin - 1);
    const float* ip = inb + (long)cic * HW;
    const float av = wk[(long)ci * Cout + wco] * wok;
    float2 s1 = {1.f, 0.f};
    if (PRO) s1 = pro_scsh[(long)n * Cin + cic];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float bv = ip[off[t]];
      if (PRO) { bv = __builtin_fmaf(bv, s1.x, s1.y); bv = bv > 0.f ? bv : bv * slope; }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv * okm[t], acc[t], 0, 0, 0);
    }
  }
  // the nine taps, summed by wave 0 in tap order
  __shared__ float red[8 * 4 * 4 * 64];
  if (kk > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((kk - 1) * 4 + t) * 4 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (kk > 0) return;
#pragma unroll 1
  for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] += red[((w2 * 4 + t) * 4 + r) * 64 + lane];
  float* pb = part + ((long)(n * nslice + slice) * Cout) * HW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cor = co0 + 4 * q + r;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = p0 + 16 * t + c;
      if (cor < Cout && p < HW) pb[(long)cor * HW + p] = acc[t][r];
    }
  }
}

// out[n][co][p] = bias[co] + sum over the channel slices (fixed order); one workgroup per (sample, group): the group's statistics
// in float64, then -- when gamma is given -- the scale / shift of the GroupNorm that follows (see gn_finalize_kernel).
__global__ __launch_bounds__(1024) void deep_reduce_kernel(const float* __restrict__ part, int nslice, const float* __restrict__ bias,
                                                           float* __restrict__ out, int N, int Cout, int HW, int out_ctot,
                                                           int out_coff, int G, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float2* __restrict__ scsh) {
  const int n = blockIdx.x / G, g = blockIdx.x - n * G;
  const int cg = Cout / G;
  const long cnt = (long)cg * HW;
  double s1 = 0.0, s2 = 0.0;
  if ((HW & 3) == 0) {                                       // four pixels per thread and step, the slices' loads in flight together
    const int HW4 = HW >> 2;
    for (int e = threadIdx.x; e < cg * HW4; e += 1024) {
      const int cl = e / HW4, p4 = e - cl * HW4, co = g * cg + cl;
      const float bv = bias ? bias[co] : 0.f;
      float4 v = {bv, bv, bv, bv};
      const float* base = part + ((long)n * nslice * Cout + co) * HW + 4 * p4;
#pragma unroll 4
      for (int sl = 0; sl < nslice; ++sl) {
        const float4 t = *reinterpret_cast<const float4*>(base + (long)sl * Cout * HW);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      *reinterpret_cast<float4*>(&out[((long)n * out_ctot + out_coff + co) * HW + 4 * p4]) = v;
      s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  } else {
    for (long e = threadIdx.x; e < cnt; e += 1024) {
      const int co = g * cg + (int)(e / HW), p = (int)(e % HW);
      float v = bias ? bias[co] : 0.f;
      for (int sl = 0; sl < nslice; ++sl) v += part[(((long)(n * nslice + sl)) * Cout + co) * HW + p];
      out[((long)n * out_ctot + out_coff + co) * HW + p] = v;
      s1 += (double)v;
      s2 += (double)v * (double)v;
    }
  }
  __shared__ double rs[16], rss[16];
  __shared__ float ms[2];
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s1; rss[threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0.0, SS = 0.0;
    for (int w = 0; w < 16; ++w) { S += rs[w]; SS += rss[w]; }
    const double mean = S / (double)cnt;
    double var = SS / (double)cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    ms[0] = (float)mean;
    ms[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (gamma && scsh)
    for (int i = threadIdx.x; i < cg; i += 1024) {
      const int ch = g * cg + i;
      const float sc = ms[1] * gamma[ch];
      scsh[(long)n * Cout + ch] = float2{sc, beta[ch] - ms[0] * sc};
    }
}

// GroupNorm statistics: one workgroup per (sample, group); mean and 1/sqrt(var + eps) (biased variance), fp64 sums.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int C, int G, int HW, float eps,
                                                       float* __restrict__ stats) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  const long cnt = (long)cg * HW;
  const float* p = x + ((long)n * C + (long)g * cg) * HW;
  double s = 0.0, ss = 0.0;
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const double v = p[i];
    s += v;
    ss += v * v;
  }
  __shared__ double rs[4], rss[4];
  s = wave_sum(s);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = rs[0] + rs[1] + rs[2] + rs[3], SS = rss[0] + rss[1] + rss[2] + rss[3];
    const double mean = S / (double)cnt;
    double var = SS / (double)cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * blockIdx.x] = (float)mean;
    stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// y = GroupNorm(x) (affine), then  act == 1: y = LeakyReLU(y);  act == 2: y = LeakyReLU(residual + y)   (layers.py:23-27)
// stats: from gn_stats_kernel (float mean, rstd) or, when sums != nullptr, the (sum, sum of squares) the producing
// convolution accumulated (biased variance, as nn.GroupNorm).
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                       const double* __restrict__ sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ residual,
                                                       float* __restrict__ out, int N, int C, int G, int HW, float eps,
                                                       float slope, int act, long total) {
  __shared__ float sm[2 * 256];
  if (sums) {                                            // N * G <= 256 (checked by the launcher)
    for (int idx = threadIdx.x; idx < N * G; idx += 256) {
      double s1 = 0.0, s2 = 0.0;
      for (int sl = 0; sl < 32; ++sl) {
        s1 += sums[((long)sl * N * G + idx) * 2];
        s2 += sums[((long)sl * N * G + idx) * 2 + 1];
      }
      const double cnt = (double)(C / G) * (double)HW;
      const double m = s1 / cnt;
      double var = s2 / cnt - m * m;
      if (var < 0.0) var = 0.0;
      sm[2 * idx] = (float)m;
      sm[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long nc = i / HW;
  const int ch = (int)(nc % C);
  const int n = (int)(nc / C);
  const int g = ch / (C / G);
  const float mean = sums ? sm[2 * (n * G + g)] : stats[2 * (n * G + g)];
  const float rstd = sums ? sm[2 * (n * G + g) + 1] : stats[2 * (n * G + g) + 1];
  float y = (x[i] - mean) * rstd * gamma[ch] + beta[ch];
  if (act == 2) y += residual[i];
  if (act) y = y > 0.f ? y : y * slope;
  out[i] = y;
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Ho = H / 2, Wo = W / 2;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
  const long nc = i / ((long)Wo * Ho);
  const float* p = in + nc * H * W + (long)(2 * y) * W + 2 * x;
  out[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
}

// bilinear x2, align_corners = False (nn.Upsample, layers.py:55): src = (dst + 0.5) / 2 - 0.5 clamped at 0
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                         long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Wo = 2 * W, Ho = 2 * H;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
  const long nc = i / ((long)Wo * Ho);
  const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float wy = sy - y0, wx = sx - x0;
  const float* p = in + nc * H * W;
  out[i] = (1.f - wy) * ((1.f - wx) * p[(long)y0 * W + x0] + wx * p[(long)y0 * W + x1]) +
           wy * ((1.f - wx) * p[(long)y1 * W + x0] + wx * p[(long)y1 * W + x1]);
}

struct Norm3 { float mean[3], sd[3]; };
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ in, float* __restrict__ out, int HW,
                                                        Norm3 nm, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)((i / HW) % 3);
  out[i] = (in[i] - nm.mean[ch]) / nm.sd[ch];                  // torchvision Normalize: sub_(mean).div_(std)
}

// normalize_params_cov + kernel_params_to_covariance (gaussian_kernel.py:6-49): 3 channels -> E = [x, s, s, z]
__global__ __launch_bounds__(256) void cov_act_kernel(const float* __restrict__ in, float* __restrict__ out, int HW,
                                                      long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long n = i / HW, p = i % HW;
  const float* k = in + n * 3 * HW + p;
  const float lo = (float)-6.907755278982137, hi = (float)9.210340371976184;   // log(1e-3), log(1e4)
  const float x = expf(fminf(fmaxf(k[0], lo), hi));
  const float z = expf(fminf(fmaxf(k[HW], lo), hi));
  const float cc = 0.99f * tanhf(k[2 * (long)HW]);
  const float s = sqrtf(x * z - 1e-8f) * cc;
  float* o = out + n * 4 * HW + p;
  o[0] = x; o[HW] = s; o[2 * (long)HW] = s; o[3 * (long)HW] = z;
}

// Antialiased bilinear resize (torchvision TF.resize(antialias=True) -> aten _upsample_bilinear2d_aa; weights as in
// aten/native/cpu/UpSampleKernel.cpp HelperInterpLinear::aa_filter / _compute_indices_min_size_weights_aa):
// triangle filter of support max(scale, 1) around centre scale * (i + 0.5), normalised.
template <typename T>
__device__ __forceinline__ void aa_taps(int i, T scale, int in_size, int& xmin, int& xsize, T& invscale, T& center) {
  const T support = (scale >= T(1)) ? scale : T(1);
  center = scale * (T(i) + T(0.5));
  invscale = (scale >= T(1)) ? T(1) / scale : T(1);
  xmin = max((int)(center - support + T(0.5)), 0);
  xsize = min((int)(center + support + T(0.5)), in_size) - xmin;
}
template <typename T>
__device__ __forceinline__ T aa_w(int j, int xmin, T center, T invscale) {
  T x = (T(j + xmin) - center + T(0.5)) * invscale;
  x = x < T(0) ? -x : x;
  return x < T(1) ? T(1) - x : T(0);
}
template <typename T>
__global__ __launch_bounds__(256) void resize_aa_kernel(const T* __restrict__ in, T* __restrict__ out, int Hi, int Wi, int Ho,
                                                        int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
  const long nc = i / ((long)Wo * Ho);
  const T sy = (T)Hi / (T)Ho, sx = (T)Wi / (T)Wo;
  int ymin, ysz, xmin, xsz;
  T yinv, ycen, xinv, xcen;
  aa_taps<T>(y, sy, Hi, ymin, ysz, yinv, ycen);
  aa_taps<T>(x, sx, Wi, xmin, xsz, xinv, xcen);
  T wys = T(0), wxs = T(0);
  for (int j = 0; j < ysz; ++j) wys += aa_w<T>(j, ymin, ycen, yinv);
  for (int j = 0; j < xsz; ++j) wxs += aa_w<T>(j, xmin, xcen, xinv);
  const T* p = in + nc * Hi * Wi;
  // separable, horizontal pass first (aten runs the W pass, then the H pass on the intermediate)
  T accv = T(0);
  for (int jy = 0; jy < ysz; ++jy) {
    const T wy = aa_w<T>(jy, ymin, ycen, yinv) / wys;
    T row = T(0);
    for (int jx = 0; jx < xsz; ++jx) row += (aa_w<T>(jx, xmin, xcen, xinv) / wxs) * p[(long)(ymin + jy) * Wi + xmin + jx];
    accv += wy * row;
  }
  out[i] = accv;
}

}  // namespace como

namespace como {
template <int KS, int MT, int WV, bool PRO>
static void launch_conv(dim3 grid, hipStream_t s, const float* in, const float* wt, const float* bias, float* out, int Cin, int CinP,
                        int Cout, int H, int W, int out_ctot, int out_coff, double* gn_sums, int gn_groups, const float2* pro_scsh,
                        const float* res, const float2* res_scsh, float slope, GnFin fin) {
  hipLaunchKernelGGL((conv_mfma_kernel<KS, MT, WV, PRO>), grid, dim3(64 * WV), 0, s, in, wt, bias, out, Cin, CinP, Cout, H, W,
                     out_ctot, out_coff, gn_sums, gn_groups, pro_scsh, res, res_scsh, slope, fin);
}

static int conv2d_impl(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout, int H,
                       int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups, const float2* pro_scsh,
                       const float* res, const float2* res_scsh, float slope, hipStream_t s,
                       GnFin fin = GnFin{nullptr, nullptr, nullptr, 0.f}) {
  if (fin.scsh && (!gn_sums || !fin.gamma || !fin.beta || res)) return COMO_ERR_ARG;
  if (!in || !wt || !out || N <= 0 || Cin <= 0 || CinP < Cin || (CinP & 3) || Cout <= 0 || H <= 0 || W <= 0 ||
      (ks != 1 && ks != 3) || out_ctot < out_coff + Cout || (gn_sums && (gn_groups <= 0 || Cout % gn_groups)) ||
      ((res != nullptr) != (res_scsh != nullptr)) || (pro_scsh && ks != 3))
    return COMO_ERR_ARG;
  const int HW = H * W;
  static const bool use_tile = [] { const char* e = getenv("COMO_NN_TILE"); return !e || e[0] != '0'; }();   // (=0: the generic kernel, A/B)
  if (ks == 3 && !res && use_tile && (W % 32) == 0 && HW >= 3072) {
    // LDS-tiled form: 4 rows per tile at 192x256 (384 workgroups), 2 below (384 / 96 x Cout / 16)
    const int th = HW >= 49152 ? 4 : 2;                     // (measured: 4 / 2 / 2 beats 8 / 4 / 2 by 2 %: twice the workgroups per level)
    if ((H % th) == 0) {
      const int R = th + 2, LW = th == 2 ? 44 : 40;
      const size_t lds = ((size_t)CinP * R * LW + (size_t)9 * CinP * 16) * sizeof(float);      // halo tile + 16 weight columns
      const size_t red = (size_t)(8 / th - 1) * (th / 2) * 16 * 64 * sizeof(float);
      const size_t bytes = lds > red ? lds : red;
      if (bytes <= 158 * 1024) {
        const dim3 g2((unsigned)(W / 32), (unsigned)(H / th), (unsigned)(N * ((Cout + 15) / 16)));
#define COMO_TILE(TH_, PRO_)                                                                                                     \
  do {                                                                                                                           \
    static bool attr = false;                                                                                                    \
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3_tile_kernel<TH_, PRO_>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024); attr = true; } \
    hipLaunchKernelGGL((conv3_tile_kernel<TH_, PRO_>), g2, dim3(256), bytes, s, in, wt, bias, out, Cin, CinP, Cout, H, W, out_ctot, \
                       out_coff, gn_sums, gn_groups, pro_scsh, slope, fin);                                                      \
  } while (0)
        if (th == 4) { if (pro_scsh) COMO_TILE(4, true); else COMO_TILE(4, false); }
        else { if (pro_scsh) COMO_TILE(2, true); else COMO_TILE(2, false); }
#undef COMO_TILE
        COMO_CHECK_LAUNCH();
        return COMO_OK;
      }
    }
  }
  const int tiles_px = (HW + 63) / 64;
  // channel-tile height and split-K width: enough waves to fill the chip, at least ~16 reduction steps per wave
  int mt = (Cout >= 32) ? 2 : 1;
  long waves = (long)tiles_px * ((Cout + 16 * mt - 1) / (16 * mt)) * N;
  if (mt == 2 && waves < 2048) { mt = 1; waves = (long)tiles_px * ((Cout + 15) / 16) * N; }
  int wvs = 1;
  while (wvs < 16 && waves * wvs < 2048 && (CinP / 4) / (wvs * 2) >= 4) wvs *= 2;
  const dim3 grid((unsigned)tiles_px, (unsigned)((Cout + 16 * mt - 1) / (16 * mt)), (unsigned)N);
#define COMO_CONV(KS_, MT_, WV_, PRO_) launch_conv<KS_, MT_, WV_, PRO_>(grid, s, in, wt, bias, out, Cin, CinP, Cout, H, W, out_ctot, out_coff, gn_sums, gn_groups, pro_scsh, res, res_scsh, slope, fin)
#define COMO_CONV_W(KS_, MT_, PRO_)                                                                               \
  switch (wvs) { case 1: COMO_CONV(KS_, MT_, 1, PRO_); break; case 2: COMO_CONV(KS_, MT_, 2, PRO_); break;        \
                 case 4: COMO_CONV(KS_, MT_, 4, PRO_); break; case 8: COMO_CONV(KS_, MT_, 8, PRO_); break;        \
                 default: COMO_CONV(KS_, MT_, 16, PRO_); break; }
  if (ks == 3 && pro_scsh) { if (mt == 2) { COMO_CONV_W(3, 2, true) } else { COMO_CONV_W(3, 1, true) } }
  else if (ks == 3 && mt == 2) { COMO_CONV_W(3, 2, false) }
  else if (ks == 3) { COMO_CONV_W(3, 1, false) }
  else if (mt == 2) { COMO_CONV_W(1, 2, false) }
  else { COMO_CONV_W(1, 1, false) }
#undef COMO_CONV_W
#undef COMO_CONV
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como

extern "C" {

int como_nn_conv2d_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                       int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups,
                       como_stream_t stream) {
  return como::conv2d_impl(in, wt, bias, out, N, Cin, CinP, Cout, H, W, ks, out_ctot, out_coff, gn_sums, gn_groups, nullptr, nullptr,
                           nullptr, 0.f, (hipStream_t)stream);
}

int como_nn_conv2d_fused_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                             int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups,
                             const float* pro_scsh, const float* res, const float* res_scsh, float slope, como_stream_t stream) {
  return como::conv2d_impl(in, wt, bias, out, N, Cin, CinP, Cout, H, W, ks, out_ctot, out_coff, gn_sums, gn_groups,
                           (const float2*)pro_scsh, res, (const float2*)res_scsh, slope, (hipStream_t)stream);
}

/* como_nn_conv2d_fused_f32 (no residual) + the finalisation of the GroupNorm that follows, inside the same launch: gn_sums holds
 * 32 * N * gn_groups * 2 doubles AND one more 8-byte word behind them (the arrival counter), all zeroed by the caller; scsh (N, Cout, 2)
 * receives what como_nn_gn_finalize_f32 would write, value for value. */
int como_nn_conv2d_gn_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                          int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups, const float* pro_scsh,
                          float slope, const float* gamma, const float* beta, float eps, float* scsh, como_stream_t stream) {
  if (!gn_sums || !gamma || !beta || !scsh) return COMO_ERR_ARG;
  return como::conv2d_impl(in, wt, bias, out, N, Cin, CinP, Cout, H, W, ks, out_ctot, out_coff, gn_sums, gn_groups,
                           (const float2*)pro_scsh, nullptr, nullptr, slope, (hipStream_t)stream,
                           como::GnFin{gamma, beta, (float2*)scsh, eps});
}

int como_nn_gn_finalize_f32(const double* sums, const float* gamma, const float* beta, int N, int C, int G, int HW, float eps,
                            float* scsh, como_stream_t stream) {
  if (!sums || !gamma || !beta || !scsh || N <= 0 || C <= 0 || G <= 0 || (C % G) || HW <= 0 || N * G > 256) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::gn_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, N, C, G, HW, eps,
                     (float2*)scsh);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

// channel slices of a reduction-split layer: >= 384 workgroups, >= 8 reduction steps per wave -- counted on the REAL input channels
// (rounded up to 4), whatever padding CinP carries, so that the scratch size below and the launcher always agree
static int deep_slices(int N, int Cin, int Cout, int HW) {
  const int tiles = ((HW + 63) / 64) * ((Cout + 15) / 16) * N;
  int ns = 1;
  while (ns < 16 && tiles * ns < 384 && (((Cin + 3) / 4) / (ns * 2)) >= 8) ns *= 2;
  return ns;
}

long como_nn_deep_part_floats(int N, int Cin, int Cout, int H, int W) {
  return (long)deep_slices(N, Cin, Cout, H * W) * N * Cout * H * W;
}

int como_nn_conv3x3_deep_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                             int H, int W, int out_ctot, int out_coff, const float* pro_scsh, float slope, float* part,
                             long part_floats, int G, const float* gamma, const float* beta, float eps, float* scsh,
                             como_stream_t stream) {
  using namespace como;
  if (!in || !wt || !out || !part || N <= 0 || Cin <= 0 || CinP < Cin || (CinP & 3) || Cout <= 0 || H <= 0 || W <= 0 ||
      out_ctot < out_coff + Cout || G <= 0 || (Cout % G) || ((gamma != nullptr) != (beta != nullptr)) || (scsh && !gamma))
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  const int ns = deep_slices(N, Cin, Cout, HW);
  if ((long)ns * N * Cout * HW > part_floats) return COMO_ERR_ARG;
  const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((Cout + 15) / 16), (unsigned)(N * ns));
  if (pro_scsh)
    hipLaunchKernelGGL(conv3_deep_kernel<true>, grid, dim3(576), 0, s, in, wt, part, Cin, CinP, Cout, H, W, ns, (const float2*)pro_scsh, slope);
  else
    hipLaunchKernelGGL(conv3_deep_kernel<false>, grid, dim3(576), 0, s, in, wt, part, Cin, CinP, Cout, H, W, ns, (const float2*)nullptr, slope);
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(deep_reduce_kernel, dim3((unsigned)(N * G)), dim3(1024), 0, s, (const float*)part, ns, bias, out, N, Cout, HW,
                     out_ctot, out_coff, G, gamma, beta, eps, (float2*)scsh);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                          float* stats, const double* sums, int N, int C, int G, int HW, float eps, float slope, int act,
                          como_stream_t stream) {
  using namespace como;
  if (!x || !gamma || !beta || !out || (!stats && !sums) || N <= 0 || C <= 0 || G <= 0 || (C % G) || HW <= 0 || act < 0 ||
      act > 2 || (act == 2 && !residual) || (sums && N * G > 256))
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!sums) {
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(N * G)), dim3(256), 0, s, x, C, G, HW, eps, stats);
    COMO_CHECK_LAUNCH();
  }
  const long total = (long)N * C * HW;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, stats, sums, gamma, beta,
                     residual, out, N, C, G, HW, eps, slope, act, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_maxpool2_f32(const float* in, float* out, int NC, int H, int W, como_stream_t stream) {
  if (!in || !out || NC <= 0 || H < 2 || W < 2) return COMO_ERR_ARG;
  const long total = (long)NC * (H / 2) * (W / 2);
  hipLaunchKernelGGL(como::maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     H, W, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_upsample2x_f32(const float* in, float* out, int NC, int H, int W, como_stream_t stream) {
  if (!in || !out || NC <= 0 || H <= 0 || W <= 0) return COMO_ERR_ARG;
  const long total = (long)NC * H * W * 4;
  hipLaunchKernelGGL(como::upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, H, W, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_normalize_f32(const float* in, float* out, int N, int HW, const float* mean3, const float* std3,
                          como_stream_t stream) {
  if (!in || !out || !mean3 || !std3 || N <= 0 || HW <= 0) return COMO_ERR_ARG;
  como::Norm3 nm;
  for (int k = 0; k < 3; ++k) { nm.mean[k] = mean3[k]; nm.sd[k] = std3[k]; }
  const long total = (long)N * 3 * HW;
  hipLaunchKernelGGL(como::normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     HW, nm, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_cov_act_f32(const float* in, float* out, int N, int HW, como_stream_t stream) {
  if (!in || !out || N <= 0 || HW <= 0) return COMO_ERR_ARG;
  const long total = (long)N * HW;
  hipLaunchKernelGGL(como::cov_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     HW, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_nn_resize_aa_f32(const float* in, float* out, int NC, int Hi, int Wi, int Ho, int Wo, como_stream_t stream) {
  if (!in || !out || NC <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return COMO_ERR_ARG;
  const long total = (long)NC * Ho * Wo;
  hipLaunchKernelGGL(como::resize_aa_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, Hi, Wi, Ho, Wo, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_nn_resize_aa_f64(const double* in, double* out, int NC, int Hi, int Wi, int Ho, int Wo, como_stream_t stream) {
  if (!in || !out || NC <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return COMO_ERR_ARG;
  const long total = (long)NC * Ho * Wo;
  hipLaunchKernelGGL(como::resize_aa_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, Hi, Wi, Ho, Wo, total);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
