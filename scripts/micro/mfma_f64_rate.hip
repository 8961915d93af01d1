// What does the float64 matrix pipe really sustain with every CU busy -- by occupancy (1, 2, 4, 8 waves per SIMD), by shape
// (v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64) and by the number of independent accumulators per wave -- and does a SIMD
// overlap it with float64 VALU work of the same wave?  (Prices the float64 block kernel of csrc/ba.hip: 320 MFMAs + ~870 other
// instructions per 64-pixel tile and wave pair.)  Round 4: the long mode (argv[1] = seconds per configuration) holds every
// configuration long enough for scripts/micro/mfma_f64_rate.sh to sample sclk / power with rocm-smi beside it.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x4 only, 1: valu only (8 independent f64 FMAs per slot), 2: both in the same wave, 3: 4x4x4 (4 blocks) only
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  double acc1[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = d4{0, 0, 0, 0}; acc1[i] = 0.0; }
  double a = threadIdx.x * 1e-3, b = 1.0001;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) {
      if (MODE == 0 || MODE == 2) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
      if (MODE == 3) acc1[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[u], 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.000001 + 0.5;
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3] + acc1[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[(blockIdx.x * 256 + threadIdx.x) & (2048 * 256 - 1)] = s;
}

static double g_seconds = 0.0;

template <int MODE, int NACC>
void run(const char* name, double* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000 * 10 / NACC;
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (g_seconds <= 0.0 || pass == 1) break;
    iters = (int)(iters * (g_seconds * 1e3 / ms));          // second pass: hold the configuration for g_seconds
    printf("[hold] %s blocks=%d for %.1f s\n", name, blocks, g_seconds); fflush(stdout);
  }
  const double slots = (double)iters * NACC * blocks * 4.0;                       // (MFMA | 8 FMA) slots over all waves
  const double flop_mfma = MODE == 3 ? 512.0 : 2048.0;
  const double tf_mfma = (MODE != 1) ? slots * flop_mfma / (ms * 1e-3) / 1e12 : 0.0;
  const double tf_valu = (MODE == 1 || MODE == 2) ? slots * 8.0 * 128.0 / (ms * 1e-3) / 1e12 : 0.0;
  printf("%-40s acc=%2d blocks=%4d (%d waves/SIMD)  %.3f ms  %.1f ns per slot and SIMD  MFMA %.1f TFLOP/s  VALU %.1f TFLOP/s\n", name, NACC,
         blocks, blocks / 256, ms, ms * 1e6 / ((double)iters * NACC) / (blocks > 256 ? blocks / 256.0 : 1.0), tf_mfma, tf_valu);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc > 1) g_seconds = atof(argv[1]);
  double* d; hipMalloc(&d, 8 * 256 * 2048);
  for (int blocks : {256, 512, 1024, 2048}) {                                      // 1 / 2 / 4 / 8 waves per SIMD
    if (blocks <= 1024) run<0, 10>("f64 mfma 16x16x4 only", d, blocks);           // 10 accumulators = 80 registers: up to 4 waves / SIMD
    run<0, 4>("f64 mfma 16x16x4 only", d, blocks);
    run<3, 10>("f64 mfma 4x4x4 (4 blocks) only", d, blocks);
    if (blocks <= 512) {
      run<1, 10>("8 f64 valu fma only", d, blocks);
      run<2, 10>("f64 mfma + 8 f64 fma (same wave)", d, blocks);
    }
  }
  return 0;
}
