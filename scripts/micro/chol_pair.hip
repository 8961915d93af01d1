// Micro-benchmark + numerical check of the diagonal 2x2-tile factorisation of csrc/chol.hip (development aid):
//   round 3: factor_pair_tail  (8-wave tile factor -> product -> update -> tile factor)
//   round 4: factor_pair_lean  (one continuous 64x64 factorisation, five lean waves, LDS-only barriers)
// One workgroup, the rest of the chip idle -- the situation of the chain workgroup at the end of a panel launch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics scripts/micro/chol_pair.hip -o scripts/micro/bin/chol_pair
#include "../../como_amd/csrc/chol.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using namespace como;

template <int MODE>
__global__ __launch_bounds__(512) void pair_kernel(const double* __restrict__ A, double* __restrict__ Lw, double* __restrict__ Iw,
                                                   int has1, int* info, long* ticks) {
  __shared__ __attribute__((aligned(16))) double sm[NT2 * TSZ];
  const int tid = threadIdx.x;
  for (int e = tid; e < CB * CB; e += 512) {
    const int r = e / CB, c = e % CB, o = r * CLD + c;
    sm[1 * TSZ + o] = A[(long)r * 64 + c];
    sm[o] = A[(long)(CB + r) * 64 + c];
    sm[2 * TSZ + o] = A[(long)(CB + r) * 64 + CB + c];
  }
  __syncthreads();
  const long t0 = wall_clock64();
  const long c0 = __builtin_readcyclecounter();
  if (MODE == 0) factor_pair_tail(sm, has1 != 0, 0, Lw, Iw, 64, 64, info);
  else factor_pair_lean(sm, has1 != 0, 0, Lw, Iw, 64, 64, info);
  __syncthreads();
  const long t1 = wall_clock64();
  const long c1 = __builtin_readcyclecounter();
  if (tid == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}

static void host_chol(std::vector<double>& a, int n) {       // in place, lower
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = a[i * n + j];
      for (int k = 0; k < j; ++k) v -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = v / d;
    }
    for (int i = 0; i < j; ++i) a[i * n + j] = 0.0;
  }
}

int main() {
  const int n = 64;
  std::vector<double> B(n * (n + 8)), A(n * n);
  srand(7);
  for (auto& v : B) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = (i == j) ? 1e-3 : 0.0;
      for (int k = 0; k < n + 8; ++k) s += B[i * (n + 8) + k] * B[j * (n + 8) + k];
      A[i * n + j] = s;
    }
  A[0] += 1e6;
  double *dA, *dL, *dI; int* dinfo; long* dt;
  CHK(hipMalloc(&dA, n * n * 8)); CHK(hipMalloc(&dL, n * n * 8)); CHK(hipMalloc(&dI, 2 * 32 * 32 * 8)); CHK(hipMalloc(&dinfo, 4)); CHK(hipMalloc(&dt, 16));
  CHK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
  for (int has1 = 1; has1 >= 0; --has1) {
    const int m = has1 ? 64 : 32;
    std::vector<double> R(m * m);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) R[i * m + j] = A[i * n + j];
    host_chol(R, m);
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<long> tk, cy;
      std::vector<double> L(n * n), I(2 * 1024);
      for (int rep = 0; rep < 12; ++rep) {
        CHK(hipMemset(dL, 0, n * n * 8)); CHK(hipMemset(dI, 0, 2 * 1024 * 8)); CHK(hipMemset(dinfo, 0, 4));
        if (mode == 0) hipLaunchKernelGGL(pair_kernel<0>, dim3(1), dim3(512), 0, 0, dA, dL, dI, has1, dinfo, dt);
        else hipLaunchKernelGGL(pair_kernel<1>, dim3(1), dim3(512), 0, 0, dA, dL, dI, has1, dinfo, dt);
        CHK(hipDeviceSynchronize());
        long h[2]; CHK(hipMemcpy(h, dt, 16, hipMemcpyDeviceToHost));
        tk.push_back(h[0]); cy.push_back(h[1]);
      }
      CHK(hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(I.data(), dI, 2 * 1024 * 8, hipMemcpyDeviceToHost));
      int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
      double eL = 0, eI = 0, mx = 0;
      for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) { eL = std::fmax(eL, std::fabs(L[i * n + j] - R[i * m + j])); mx = std::fmax(mx, std::fabs(R[i * m + j])); }
      for (int t = 0; t < (has1 ? 2 : 1); ++t) {             // V_t L_tt = I
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double s = 0; for (int k = 0; k < 32; ++k) s += I[t * 1024 + i * 32 + k] * R[(32 * t + k) * m + 32 * t + j];
          eI = std::fmax(eI, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
      }
      std::sort(tk.begin(), tk.end()); std::sort(cy.begin(), cy.end());
      printf("%s has1=%d : min %.2f us  median %.2f us  (memtime ticks median %ld)   |L - ref| %.2e (max |L| %.1e)   |V L - I| %.2e   info %d\n",
             mode == 0 ? "round-3 factor_pair_tail" : "round-4 factor_pair_lean", has1, tk[0] * 0.01, tk[tk.size() / 2] * 0.01, cy[cy.size() / 2], eL, mx, eI, info);
    }
  }
#ifdef COMO_FP_PROFILE
  {   // per-wave barrier stamps of one lean factorisation: [wave][step][0 = left the barrier, 1 = reached the NEXT barrier's wait]
    long* dp; CHK(hipMalloc(&dp, 8 * 18 * 2 * 8)); CHK(hipMemset(dp, 0, 8 * 18 * 2 * 8));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(como::fp_prof), &dp, sizeof(dp)));
    CHK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(pair_kernel<1>, dim3(1), dim3(512), 0, 0, dA, dL, dI, 1, dinfo, dt); CHK(hipDeviceSynchronize()); }
    std::vector<long> P(8 * 18 * 2); CHK(hipMemcpy(P.data(), dp, P.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[6] = {"U0 (T00)", "U1 (T10)", "U2 (T11)", "micro", "inverse 0", "inverse 1"};
    const long t00 = P[(3 * 18 + 0) * 2 + 0];
    printf("cycle stamps relative to the micro wave leaving barrier 0; per step: leave-barrier -> arrive-at-next-barrier (busy cycles)\n");
    for (int w = 0; w < 6; ++w) {
      printf("%-10s", nm[w]);
      for (int st = 0; st < 17; ++st) {
        const long a = P[(w * 18 + st) * 2 + 0], b = P[(w * 18 + st + 1) * 2 + 1];
        printf(" %5ld+%-4ld", a - t00, b - a);
      }
      printf("\n");
    }
    long dnull = 0; CHK(hipMemcpyToSymbol(HIP_SYMBOL(como::fp_prof), &dnull, sizeof(dnull)));
  }
#endif
  // a non-positive-definite block: the first failing pivot must be reported (1-based) by both
  for (int where : {1, 7, 33, 50}) {
    std::vector<double> A2 = A;
    A2[(where - 1) * n + where - 1] = -1.0;
    CHK(hipMemcpy(dA, A2.data(), n * n * 8, hipMemcpyHostToDevice));
    int i0, i1;
    CHK(hipMemset(dinfo, 0, 4));
    hipLaunchKernelGGL(pair_kernel<0>, dim3(1), dim3(512), 0, 0, dA, dL, dI, 1, dinfo, dt); CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&i0, dinfo, 4, hipMemcpyDeviceToHost));
    CHK(hipMemset(dinfo, 0, 4));
    hipLaunchKernelGGL(pair_kernel<1>, dim3(1), dim3(512), 0, 0, dA, dL, dI, 1, dinfo, dt); CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&i1, dinfo, 4, hipMemcpyDeviceToHost));
    printf("negative pivot at %d: info round-3 %d, round-4 %d\n", where, i0, i1);
  }
  return 0;
}
