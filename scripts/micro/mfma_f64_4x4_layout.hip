// Operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 products per instruction), probed with one-hot
// operands: A = e_a, B = e_b for every lane pair (a, b); the lane where the product lands tells (block, i, k) of a, (block, k, j) of b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
  const int a = blockIdx.x >> 6, b = blockIdx.x & 63, l = threadIdx.x;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == a ? 1.0 : 0.0, l == b ? 1.0 : 0.0, 0.0, 0, 0, 0);
  out[blockIdx.x * 64 + l] = d;
}
// one wave alone on the chip (the situation of the Cholesky chain workgroup): cycles per instruction, dependent (one accumulator)
// and independent (four accumulators) issue, both float64 shapes, and the plain f64 FMA for scale
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void lat(long* out, double* sink, int iters) {
  d4 a16[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a4[4] = {0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = 1.0001;
  const long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) a16[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a16[0], 0, 0, 0);
      if (MODE == 1) a16[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a16[u], 0, 0, 0);
      if (MODE == 2) a4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a4[0], 0, 0, 0);
      if (MODE == 3) a4[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a4[u], 0, 0, 0);
      if (MODE == 4) a4[0] = __builtin_fma(a4[0], y, x);
      if (MODE == 5) a4[u] = __builtin_fma(a4[u], y, x);
    }
  }
  const long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = a16[0][0] + a16[1][1] + a16[2][2] + a16[3][3] + a4[0] + a4[1] + a4[2] + a4[3];
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE> void runlat(const char* nm, long* d, double* sink) {
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(lat<MODE>, dim3(1), dim3(64), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("one wave alone: %-44s %.1f cycles per instruction\n", nm, (double)h / (iters * 4.0));
}

int main(int argc, char** argv) {
  {
    long* dl; double* sk; hipMalloc(&dl, 8); hipMalloc(&sk, 512);
    runlat<0>("v_mfma_f64_16x16x4 dependent (one accumulator)", dl, sk);
    runlat<1>("v_mfma_f64_16x16x4 four accumulators", dl, sk);
    runlat<2>("v_mfma_f64_4x4x4_4b dependent", dl, sk);
    runlat<3>("v_mfma_f64_4x4x4_4b four accumulators", dl, sk);
    runlat<4>("v_fma_f64 dependent", dl, sk);
    runlat<5>("v_fma_f64 four accumulators", dl, sk);
  }
  double* d; hipMalloc(&d, 4096 * 64 * 8);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
  std::vector<double> h(4096 * 64); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  if (argc > 1) {                                          // any argument: the whole 64 x 64 table
    printf("hit lane of (a, b), -1 = no product (rows a = 0..63, columns b = 0..63):\n");
    for (int a = 0; a < 64; ++a) {
      for (int b = 0; b < 64; ++b) {
        int hit = -1;
        for (int l = 0; l < 64; ++l) if (h[(a * 64 + b) * 64 + l] != 0.0) hit = hit < 0 ? l : 99;
        printf("%3d", hit);
      }
      printf("\n");
    }
  }
  // layout (found with the table above, then checked here for all 4096 lane pairs):
  //   A_blk[i][k] at lane 16 k + 4 blk + i ;  B_blk[k][j] at lane 16 k + 4 blk + j ;  D_blk[i][j] at lane 16 i + 4 blk + j
  int ok = 0, total = 0, other = 0;
  for (int a = 0; a < 64; ++a) for (int b = 0; b < 64; ++b) {
    int hit = -1, nh = 0;
    for (int l = 0; l < 64; ++l) if (h[(a * 64 + b) * 64 + l] != 0.0) { hit = l; ++nh; }
    const int ka = a >> 4, ba = (a >> 2) & 3, ia = a & 3, kb = b >> 4, bb = (b >> 2) & 3, jb = b & 3;
    const bool expect = ba == bb && ka == kb;
    if (expect) { ++total; if (nh == 1 && hit == 16 * ia + 4 * ba + jb) ++ok; else ++other; }
    else if (nh) ++other;
  }
  printf("v_mfma_f64_4x4x4_4b: A_blk[i][k] @ lane 16k+4blk+i, B_blk[k][j] @ lane 16k+4blk+j, D_blk[i][j] @ lane 16i+4blk+j : %d of %d products as predicted, %d contradictions\n",
         ok, total, other);
  return 0;
}
