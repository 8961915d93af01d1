// What does v_mfma_f64_16x16x4_f64 really sustain with every CU busy, and does a SIMD overlap it with float64 VALU work of the
// same wave / of a second wave?  (Prices the float64 block kernel of csrc/ba.hip: 320 MFMAs + ~870 other instructions per
// 64-pixel tile and wave pair.)   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: mfma only, 1: valu only (8 independent f64 FMAs per slot), 2: both in the same wave
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0001;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (MODE == 0 || MODE == 2) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
      if (MODE >= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.000001 + 0.5;
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double slots = (double)iters * 10.0 * blocks * 4.0;                       // (MFMA | 8 FMA) slots over all waves
  const double tf_mfma = (MODE != 1) ? slots * 2048.0 / (ms * 1e-3) / 1e12 : 0.0;
  const double tf_valu = (MODE >= 1) ? slots * 8.0 * 128.0 / (ms * 1e-3) / 1e12 : 0.0;
  printf("%-44s blocks=%4d  %.3f ms  %.0f ns per slot and SIMD  MFMA %.1f TFLOP/s  VALU %.1f TFLOP/s\n", name, blocks, ms,
         ms * 1e6 / ((double)iters * 10.0) / (blocks > 256 ? blocks / 256.0 : 1.0), tf_mfma, tf_valu);
}

int main() {
  double* d; hipMalloc(&d, 8 * 256 * 2048);
  for (int blocks : {256, 512}) {                                                  // one / two waves per SIMD
    run<0>("f64 mfma 16x16x4 only", d, blocks);
    run<1>("8 f64 valu fma only", d, blocks);
    run<2>("f64 mfma + 8 f64 fma (same wave)", d, blocks);
  }
  return 0;
}
