// One 4-pixel step of the depth x depth Gram of csrc/ba.hip (role 0 of ba_blocks_pair2_f64): 10 tiles of 16x16, as
//   P0: 10 x v_mfma_f64_16x16x4                      (round 3)
//   P1: 40 x v_mfma_f64_4x4x4_4b, operands reused     (no rotations: the matrix pipe alone)
//   P2: P1 + the 12 rotated operands (24 v_mov_b32 row_ror)
//   P3: P2 + 4 scalings + 4 gradient FMAs             (the whole VALU side of the step)
//   P4: 40 x 4x4x4 with the rotated operands fetched by ds_bpermute (LDS crossbar) instead of DPP
// every CU busy, 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f64_gram_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int D> __device__ __forceinline__ double rot(double v) {
  constexpr int ctrl = 0x120 + (16 - 4 * D);
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int D> __device__ __forceinline__ double rotp(double v, int lane) {
  const int src = ((lane & 48) | ((lane + 4 * D) & 15)) * 4;
  int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double m4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters) {
  d4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = d4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  double q[4], g[4] = {0, 0, 0, 0};
  for (int e = 0; e < 4; ++e) q[e] = in[threadIdx.x * 4 + e];
  double w = in[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    double z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = (MODE >= 3 && MODE != 4) ? w * q[e] : q[e];
    if (MODE >= 3 && MODE != 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += z[e] * w;
    }
    if (MODE == 0) {
      int t = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) { acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[i], z[j], acc[t], 0, 0, 0); ++t; }
    } else {
      double r[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 1) { r[0][e] = z[(e + 1) & 3]; r[1][e] = z[(e + 2) & 3]; r[2][e] = z[(e + 3) & 3]; }
        else if (MODE == 4) { r[0][e] = rotp<1>(z[e], lane); r[1][e] = rotp<2>(z[e], lane); r[2][e] = rotp<3>(z[e], lane); }
        else { r[0][e] = rot<1>(z[e]); r[1][e] = rot<2>(z[e]); r[2][e] = rot<3>(z[e]); }
      }
      int t = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
          acc[t][0] = m4(z[i], z[j], acc[t][0]);
          acc[t][1] = m4(z[i], r[0][j], acc[t][1]);
          acc[t][2] = m4(z[i], r[1][j], acc[t][2]);
          acc[t][3] = m4(z[i], r[2][j], acc[t][3]);
          ++t;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] += 1e-9;            // fresh operands every step (as the K~ ring delivers them)
  }
  double s = 0;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int e = 0; e < 4; ++e) s += g[e];
  out[(blockIdx.x * 256 + threadIdx.x) & 0xfffff] = s;
}
template <int MODE> void run(const char* nm, double* d, double* in, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, in, 10);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_step = ms * 1e6 / iters / (blocks / 256.0);
  printf("%-58s %d wave(s)/SIMD  %.1f ns = %.0f cycles (2.39 GHz) per 4-pixel step and SIMD   %.1f TFLOP/s of Gram\n", nm, blocks / 256, ns_step,
         ns_step * 2.39, 10 * 2048.0 * iters * blocks * 4.0 / (ms * 1e-3) / 1e12);
}
int main() {
  double *d, *in; (void)hipMalloc(&d, 8 << 20); (void)hipMalloc(&in, 8 * 1024); (void)hipMemset(in, 0, 8 * 1024);
  for (int blocks : {256, 512}) {
    run<0>("P0 10 x mfma 16x16x4", d, in, blocks);
    run<1>("P1 40 x mfma 4x4x4 (no rotations)", d, in, blocks);
    run<2>("P2 40 x mfma 4x4x4 + 24 dpp row_ror", d, in, blocks);
    run<3>("P3 P2 + 4 scalings + 4 gradient fma", d, in, blocks);
    run<4>("P4 40 x mfma 4x4x4 + 24 ds_bpermute", d, in, blocks);
  }
  return 0;
}
