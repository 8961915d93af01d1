// Does a SIMD overlap MFMA with VALU work of the SAME wave / of another wave?  f32 16x16x4 vs bf16 16x16x32 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: f32 mfma only, 1: bf16 mfma only, 2: valu only, 3: f32 mfma + valu, 4: bf16 mfma + valu
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(b + i); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0 || MODE == 3) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
      if (MODE == 1 || MODE == 4) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[u], 0, 0, 0);
      if (MODE >= 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = v[j] * 1.000001f + 0.5f;       // 16 independent FMAs per MFMA slot
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per wave: iters * 8 slots; report cycles per slot at 2.39 GHz (one wave per SIMD when blocks == 256)
  printf("%-34s blocks=%4d  %.3f ms  -> %.1f cycles per (MFMA + 16 FMA) slot\n", name, blocks, ms, ms * 1e-3 * 2.39e9 / (iters * 8.0) / (blocks > 256 ? blocks / 256.0 : 1.0));
}

int main() {
  float* d; hipMalloc(&d, 4 * 256 * 2048);
  for (int blocks : {256, 512}) {
    run<0>("f32 mfma 16x16x4 only", d, blocks);
    run<1>("bf16 mfma 16x16x32 only", d, blocks);
    run<2>("16 valu fma only", d, blocks);
    run<3>("f32 mfma + 16 fma (same wave)", d, blocks);
    run<4>("bf16 mfma + 16 fma (same wave)", d, blocks);
  }
  return 0;
}
