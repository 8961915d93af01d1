// Shader clock probe: s_sleep counts core cycles (64 per unit); wall_clock64 is a constant 100 MHz counter.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void clk(long* out, int reps) {
  long w0 = wall_clock64(); long m0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
  long w1 = wall_clock64(); long m1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = m1 - m0; }
}
// dependent v_fma_f32 chain: 4 cycles issue each on a 16-lane SIMD
__global__ void fchain(long* out, float* sink, int reps) {
  float a = threadIdx.x * 1e-3f, b = 1.000001f, c = 1e-7f;
  long w0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = a * b + c;
  }
  long w1 = wall_clock64();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0) out[0] = w1 - w0;
}
__global__ void dchain(long* out, double* sink, int reps) {
  double a = threadIdx.x * 1e-3, b = 1.000001, c = 1e-7;
  long w0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = a * b + c;
  }
  long w1 = wall_clock64();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0) out[0] = w1 - w0;
}
int main() {
  long* d; hipMalloc(&d, 16); float* s; hipMalloc(&s, 4096); double* sd; hipMalloc(&sd, 8192);
  long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, 0, d, 200); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_sleep: %ld ticks(10ns) for %d x 127 x 64 cycles -> %.0f MHz ; memtime rate %.0f MHz\n", h[0], 200, 200.0 * 127 * 64 / (h[0] * 10e-3) , h[1] / (h[0] * 10e-3));
  }
  hipLaunchKernelGGL(fchain, dim3(1), dim3(64), 0, 0, d, s, 1000); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("dependent f32 fma: %.2f ns each\n", h[0] * 10.0 / 64000);
  hipLaunchKernelGGL(dchain, dim3(1), dim3(64), 0, 0, d, sd, 1000); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("dependent f64 fma: %.2f ns each\n", h[0] * 10.0 / 64000);
  return 0;
}
