// Micro-benchmarks for the serial pivot chain of the Cholesky tile kernel (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void acc_kernel(const double* x, double* out_rcp, double* out_rsq, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out_rcp[i] = __builtin_amdgcn_rcp(x[i]); out_rsq[i] = __builtin_amdgcn_rsq(x[i]); }
}

// dependent chain of N f64 FMAs in one wave; reports core cycles (s_memtime) and 100 MHz ticks
template <int MODE>
__global__ void chain_kernel(double* out, long* cyc, int iters, double seed) {
  double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9;
  __shared__ double sh[64];
  long t0 = __builtin_readcyclecounter();
  long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { a = a * b + c; }                                     // f64 fma latency
    if (MODE == 1) { a = __builtin_amdgcn_rcp(a) + 1.5; }                  // rcp + add
    if (MODE == 2) { sh[threadIdx.x & 63] = a; __syncthreads(); a = sh[(threadIdx.x + 1) & 63] + c; }   // lds roundtrip + barrier
    if (MODE == 3) { float f = (float)a; f = f * 1.0000001f + 1e-9f; a = f; }
    if (MODE == 4) { sh[threadIdx.x & 63] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); a = sh[(threadIdx.x + 1) & 63] + c; }
  }
  long t1 = __builtin_readcyclecounter();
  long w1 = wall_clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int MODE>
int run_chain(const char* name, int threads) {
  double* out; long* cyc; CHK(hipMalloc(&out, 8 * 1024)); CHK(hipMalloc(&cyc, 16));
  const int iters = 4096;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain_kernel<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.25);
  CHK(hipDeviceSynchronize());
  long h[2]; CHK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  printf("%-28s threads=%4d  %.1f memtime-ticks/iter  %.1f ns/iter\n", name, threads, (double)h[0] / iters, (double)h[1] * 10.0 / iters);
  return 0;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r(n), q(n);
  for (int i = 0; i < n; ++i) x[i] = std::exp(((double)rand() / RAND_MAX - 0.5) * 60.0);
  double *dx, *dr, *dq; CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&dr, n * 8)); CHK(hipMalloc(&dq, n * 8));
  CHK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dr, dq, n);
  CHK(hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost));
  double er = 0, eq = 0;
  for (int i = 0; i < n; ++i) { er = std::fmax(er, std::fabs(r[i] * x[i] - 1.0)); eq = std::fmax(eq, std::fabs(q[i] * q[i] * x[i] - 1.0) * 0.5); }
  printf("v_rcp_f64 max rel err %.3e   v_rsq_f64 max rel err %.3e\n", er, eq);
  run_chain<0>("f64 fma chain", 64);
  run_chain<1>("f64 rcp+add chain", 64);
  run_chain<3>("f32 cvt+fma+cvt chain", 64);
  run_chain<2>("lds store+barrier+load", 64);
  run_chain<2>("lds store+barrier+load", 256);
  run_chain<2>("lds store+barrier+load", 1024);
  run_chain<4>("lds store+wavebar+load", 64);
  return 0;
}
