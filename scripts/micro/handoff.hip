// Micro-benchmark (development aid, round 5): what does it cost to hand a 64x64 float64 super-tile (32 KB) from one workgroup to
// a workgroup on ANOTHER XCD inside one kernel, and which store / load / fence combination is actually safe?  Decides the memory
// protocol of the persistent dense solve (csrc/cholp.hip).
//   writer protocols W: 0 plain stores + s_waitcnt vmcnt(0) + flag atomic     1 plain stores + agent release fence + flag
//                       2 agent-scope (sc1) stores + vmcnt(0) + flag           3 returning 64-bit atomic exchanges + flag
//   reader protocols R: 0 poll + plain loads     1 poll + agent acquire fence + plain loads     2 poll + agent-scope (sc1) loads
//   memory            : cached (hipMalloc) / uncached (hipExtMallocWithFlags(hipDeviceMallocUncached))
// Two workgroups of 1024 threads ping-pong the tile ROUNDS times (the value written = the round number; the reader counts
// elements that are not the expected value = stale reads); one round trip = two hand-offs.  Other workgroups keep the remaining
// XCDs' slots occupied so that block 0 and block 1 sit on different XCDs (round-robin dispatch), and `xcc` is recorded.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/handoff.hip -o scripts/micro/bin/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int TILE = 4096;           // doubles
constexpr int ROUNDS = 400;

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void publish(double* tile, double v, int wmode, unsigned* flag, unsigned fv) {
  const int tid = threadIdx.x;
  if (wmode == 3) {
    unsigned long long sink = 0;
    for (int e = tid; e < TILE; e += 1024)
      sink += __hip_atomic_exchange((unsigned long long*)&tile[e], (unsigned long long)__double_as_longlong(v + e), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(sink));
  } else if (wmode == 2) {
    for (int e = tid; e < TILE; e += 1024) __hip_atomic_store(&tile[e], v + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    for (int e = tid; e < TILE; e += 1024) tile[e] = v + e;
  }
  if (wmode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();                                   // (s_waitcnt vmcnt(0) of every wave, then the barrier)
  if (tid == 0) __hip_atomic_store(flag, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int consume(const double* tile, double v, int rmode, const unsigned* flag, unsigned fv, int* timeout) {
  const int tid = threadIdx.x;
  __shared__ int ok_s;
  if (tid == 0) {
    int ok = 1;
    long spin = 0;
    while (ld_flag(flag) < fv) { if (++spin > 4000000) { ok = 0; break; } }
    ok_s = ok;
  }
  __syncthreads();
  if (!ok_s) { *timeout = 1; return 0; }
  if (rmode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int bad = 0;
  for (int e = tid; e < TILE; e += 1024) {
    const double x = rmode == 2 ? __hip_atomic_load(&tile[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tile[e];
    bad += (x != v + e);
  }
  return bad;
}

__global__ __launch_bounds__(1024) void pingpong(double* tiles, unsigned* flags, int wmode, int rmode, long* out, int* xcc) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b] = (int)(id & 0xf);
  }
  if (b > 1) return;
  double* mine = tiles + (long)b * TILE;
  double* theirs = tiles + (long)(1 - b) * TILE;
  unsigned* myflag = flags + 64 * b;
  unsigned* theirflag = flags + 64 * (1 - b);
  int bad = 0, timeout = 0;
  __syncthreads();
  const long t0 = wall_clock64();
  for (int r = 1; r <= ROUNDS && !timeout; ++r) {
    if (b == 0) {
      publish(mine, (double)r, wmode, myflag, (unsigned)r);
      bad += consume(theirs, (double)r + 0.5, rmode, theirflag, (unsigned)r, &timeout);
    } else {
      bad += consume(theirs, (double)r, rmode, theirflag, (unsigned)r, &timeout);
      publish(mine, (double)r + 0.5, wmode, myflag, (unsigned)r);
    }
  }
  const long t1 = wall_clock64();
  for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o, 64);
  __shared__ int tot;
  if (tid == 0) tot = 0;
  __syncthreads();
  if ((tid & 63) == 0) atomicAdd(&tot, bad);
  __syncthreads();
  if (tid == 0) { out[3 * b] = t1 - t0; out[3 * b + 1] = tot; out[3 * b + 2] = timeout; }
}

int main() {
  double* tiles[2];
  unsigned* flags[2];
  long* out;
  int* xcc;
  CHK(hipMalloc(&tiles[0], 2 * TILE * sizeof(double)));
  CHK(hipMalloc(&flags[0], 1024));
  CHK(hipExtMallocWithFlags((void**)&tiles[1], 2 * TILE * sizeof(double), hipDeviceMallocUncached));
  CHK(hipExtMallocWithFlags((void**)&flags[1], 1024, hipDeviceMallocUncached));
  CHK(hipMalloc(&out, 64));
  CHK(hipMalloc(&xcc, 64 * sizeof(int)));
  printf("hand-off of a 32 KB tile between two workgroups (1024 threads), %d round trips; wall clock 100 MHz\n", ROUNDS);
  printf("%-9s %-28s %-22s %10s %8s %8s %s\n", "memory", "writer", "reader", "us/handoff", "stale", "timeout", "xcc(b0,b1)");
  const char* wn[4] = {"plain+vmcnt0", "plain+release fence", "sc1 stores+vmcnt0", "returning atomic xchg"};
  const char* rn[3] = {"plain loads", "acquire fence+plain", "sc1 loads"};
  for (int mem = 0; mem < 2; ++mem)
    for (int w = 0; w < 4; ++w)
      for (int r = 0; r < 3; ++r) {
        CHK(hipMemset(tiles[mem], 0, 2 * TILE * sizeof(double)));
        CHK(hipMemset(flags[mem], 0, 1024));
        CHK(hipMemset(out, 0, 64));
        hipLaunchKernelGGL(pingpong, dim3(16), dim3(1024), 0, 0, tiles[mem], flags[mem], w, r, out, xcc);
        CHK(hipDeviceSynchronize());
        long h[6];
        int x[16];
        CHK(hipMemcpy(h, out, 48, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(x, xcc, 64, hipMemcpyDeviceToHost));
        printf("%-9s %-28s %-22s %10.2f %8ld %8ld (%d,%d)\n", mem ? "uncached" : "cached", wn[w], rn[r],
               (double)h[0] / 100.0 / (2.0 * ROUNDS), h[1] + h[4], h[2] + h[5], x[0], x[1]);
      }
  return 0;
}
