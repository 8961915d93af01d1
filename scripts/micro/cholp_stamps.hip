// Development aid (round 5): the persistent dense solve (csrc/cholp.hip) alone -- numerical check against a host Cholesky and the
// in-kernel wall-clock stamps of its chain workgroup (where does a pair's period go: waiting for its inputs, loading them, the two
// product stages, the factorisation, the inverse's off-diagonal block, publishing).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DCOMO_CP_PROFILE scripts/micro/cholp_stamps.hip -o scripts/micro/bin/cholp_stamps
#include "../../como_amd/csrc/cholp.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace como;

int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 760;
  if (!cholp_size_ok(D)) { printf("D = %d is outside the persistent solver's range\n", D); return 1; }
  const int np = chol_np(D);
  const long Dp = 64L * np;
  std::vector<double> B((size_t)D * (D + 8)), H((size_t)D * D), g(D), W((size_t)Dp * Dp, 0.0);
  srand(7);
  for (auto& v : B) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = (i == j) ? 1e-3 : 0.0;
      for (int k = 0; k < D + 8; ++k) s += B[(size_t)i * (D + 8) + k] * B[(size_t)j * (D + 8) + k];
      H[(size_t)i * D + j] = s;
    }
  H[0] += 1e12;
  for (auto& v : g) v = (double)rand() / RAND_MAX - 0.5;
  for (long i = 0; i < Dp; ++i)
    for (long j = 0; j < Dp; ++j) {
      double v = 0.0;
      if (i < D && j < D) v = j <= i ? H[(size_t)i * D + j] : 0.0;
      else if (i == D && j < D) v = g[j];
      else if (i == D && j == D) v = 1e300;
      else if (i == j) v = 1.0;
      W[(size_t)i * Dp + j] = v;
    }
  // host reference: L L^T = H, delta = H^-1 g
  std::vector<double> L(H), x(g);
  for (int j = 0; j < D; ++j) {
    double d = L[(size_t)j * D + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * D + k] * L[(size_t)j * D + k];
    d = std::sqrt(d);
    L[(size_t)j * D + j] = d;
    for (int i = j + 1; i < D; ++i) {
      double v = L[(size_t)i * D + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * D + k] * L[(size_t)j * D + k];
      L[(size_t)i * D + j] = v / d;
    }
  }
  for (int i = 0; i < D; ++i) { double v = x[i]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * D + k] * x[k]; x[i] = v / L[(size_t)i * D + i]; }
  for (int i = D - 1; i >= 0; --i) { double v = x[i]; for (int k = i + 1; k < D; ++k) v -= L[(size_t)k * D + i] * x[k]; x[i] = v / L[(size_t)i * D + i]; }

  const long ws_doubles = cholp_sync_offset(Dp, np) + cholp_sync_words(np) / 2 + 64;
  double *ws, *delta;
  int* info;
  long long* stamps;
  CHK(hipMalloc(&ws, ws_doubles * sizeof(double)));
  CHK(hipMalloc(&delta, D * sizeof(double)));
  CHK(hipMalloc(&info, sizeof(int)));
  CHK(hipMalloc(&stamps, 48 * 8 * sizeof(long long)));
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(cp_stamps), &stamps, sizeof(stamps)));
  if (cholp_init() == 0) { printf("persistent solver disabled on this device\n"); return 1; }
  std::vector<long long> st(48 * 8);
  std::vector<double> got(D);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int reps = D > 1100 ? 6 : 20;
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CHK(hipMemcpy(ws, W.data(), (size_t)Dp * Dp * sizeof(double), hipMemcpyHostToDevice));
    CHK(hipMemset(ws + cholp_sync_offset(Dp, np), 0, cholp_sync_words(np) * 4));
    CHK(hipMemset(info, 0, sizeof(int)));
    CHK(hipMemset(stamps, 0, 48 * 8 * sizeof(long long)));
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    if (cholp_solve(delta, ws, D, info, 0) != COMO_OK) { printf("cholp_solve refused\n"); return 1; }
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      CHK(hipMemcpy(st.data(), stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
    }
  }
  int hinfo;
  CHK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
  CHK(hipMemcpy(got.data(), delta, D * sizeof(double), hipMemcpyDeviceToHost));
  double num = 0.0, den = 0.0;
  for (int i = 0; i < D; ++i) { num += (got[i] - x[i]) * (got[i] - x[i]); den += x[i] * x[i]; }
  printf("D = %d, np = %d, %d workgroups; info %d; |delta - host| / |host| = %.3e; launch-to-end (HIP events, best of %d): %.1f us\n", D, np,
         1 + (cholp_tiles(np) < cholp_init() - 1 ? cholp_tiles(np) : cholp_init() - 1), hinfo, std::sqrt(num / den), reps, best * 1e3);
  printf("chain workgroup, per pair p (us): factor | then for the next pair: inputs-ready  V10+assemble  X-products(+publish ack)  X-store  T-stage | period\n");
  auto us = [](long long a, long long b) { return (a && b) ? (double)(b - a) / 100.0 : 0.0; };
  for (int p = 0; p < np; ++p) {
    const long long* s = &st[p * 8];
    const long long* n = &st[(p + 1) * 8];
    if (p + 1 < np)
      printf("  pair %2d: %6.2f | %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f\n", p, us(s[4], s[5]), us(s[5], n[1]), us(n[1], s[6]), us(s[6], n[2]),
             us(n[2], n[3]), us(n[3], n[4]), us(s[4], n[4]));
    else
      printf("  pair %2d: %6.2f | last pair: V10 + publish + delta %6.2f\n", p, us(s[4], s[5]), us(s[5], s[7]));
  }
  printf("chain total (first stamp -> last): %.1f us\n", (double)(st[(np - 1) * 8 + 7] - st[4]) / 100.0);
  return 0;
}
