// Which SIMD does the dispatcher put the waves of a 192-thread (three-wave) workgroup on?  The wave-specialised block kernel
// (csrc/ba.hip ba_blocks_ws_f64_kernel) wants every SIMD to host a mix of producer / consumer waves.  Prints, for a launch with
// the same footprint (39.9 KB LDS, 168 registers -> four workgroups per CU), a histogram of (wave index -> SIMD id) and per-CU
// the multiset of roles per SIMD under the kernel's role rotation.   hipcc --offload-arch=gfx950 -O2 simd_place.hip -o simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

__global__ __launch_bounds__(192, 3) void probe(unsigned* out, int spin) {
  __shared__ double pad[4992];
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, bits [31:0] via two reads below
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  pad[threadIdx.x] = (double)id;
  __syncthreads();
  // keep the workgroup resident for a while so that the whole first round is co-resident
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) { pad[threadIdx.x] += 1.0; }
  if ((threadIdx.x & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 3 + (threadIdx.x >> 6)] = id;
  if (pad[threadIdx.x] == 12345.678) out[0] = hw;
}

int main() {
  const int gx = 128, gy = 8;
  unsigned* d;
  hipMalloc(&d, gx * gy * 3 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(192), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(gx * gy * 3);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se_id [16:13]?)
  int hist[3][4] = {};
  std::map<unsigned, std::vector<int>> cu_roles[4];           // (cu key) -> roles on each SIMD
  for (int by = 0; by < gy; ++by)
    for (int bx = 0; bx < gx; ++bx)
      for (int w = 0; w < 3; ++w) {
        const unsigned id = h[(by * gx + bx) * 3 + w];
        const int simd = (id >> 4) & 3;
        hist[w][simd]++;
        const unsigned cu = id >> 8;                              // everything above the SIMD / wave bits identifies the CU
        const int role = (w + bx + by) % 3;
        cu_roles[simd][cu].push_back(role);
      }
  for (int w = 0; w < 3; ++w) printf("wave %d -> SIMD0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // per SIMD: how many of each role (rotated roles), summed over CUs, and the worst single SIMD (max consumers on one SIMD)
  int worst = 0, total_simds = 0; long cons_total = 0;
  int per_simd_hist[16] = {};
  for (int s = 0; s < 4; ++s)
    for (auto& kv : cu_roles[s]) {
      int cons = 0;
      for (int r : kv.second) cons += (r != 0);
      per_simd_hist[cons < 15 ? cons : 15]++;
      worst = cons > worst ? cons : worst;
      cons_total += cons;
      total_simds++;
    }
  printf("SIMDs seen %d, consumer waves %ld, max consumers on one SIMD %d\nconsumers-per-SIMD histogram:", total_simds, cons_total, worst);
  for (int k = 0; k < 10; ++k) printf(" %d:%d", k, per_simd_hist[k]);
  printf("\nfirst 8 workgroups (wave0 wave1 wave2 raw HW_ID): ");
  for (int k = 0; k < 24; ++k) printf("%08x ", h[k]);
  printf("\n");
  return 0;
}
