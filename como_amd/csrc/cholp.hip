// Dense SPD solve delta = H^-1 g in ONE persistent launch (float64, 2 .. 34 column pairs: D <= 2175; one super-tile per
// workgroup up to 16 pairs, then two, four, eight -- see cp_tile / cp_workers_body).
//
// Reference: como/odom/backend/linear_system.py:101-112 (`solve_system`: cholesky_ex + cholesky_solve).
//
// csrc/chol.hip eliminates one PAIR of 32-wide block columns per launch; its chain workgroup spends 17.6 us per launch of
// which only 9.4 us are the factorisation of the next 64 x 64 diagonal block -- the rest is re-loading tiles that were on chip
// a moment ago, three dependent panel products, and the launch boundary (dense solve).  Here the launch boundaries
// are replaced by counters in memory and every piece of the matrix has ONE owner for the whole solve:
//
//   * the matrix is cut into 64 x 64 SUPER-TILES (I, J) (pair units; np = Dp / 64 pairs).  One workgroup (512 threads, one
//     per compute unit, all co-resident: (np - 1) + np (np - 2) super-tiles <= 255 up to 16 pairs; beyond that the super-tiles
//     are dealt out column-major and cyclically, several per workgroup) owns a super-tile and keeps it in the accumulator
//     registers of its 8 waves (two 16 x 16 quadrants each) from the first to the last update -- no read-modify-write of the
//     working copy between steps.  Step s (columns of pair s): X_I = A(I, s) Vp_s^T, X_J = A(J, s) Vp_s^T with the published
//     INVERSE of the factored diagonal pair (Vp_s = L_ss^-1, 64 x 64 lower triangular: ONE product stage instead of three
//     dependent ones), S -= X_I X_J^T.  When column pair J is next, the owner publishes its super-tile once (it is the panel
//     A(I, J) of every later step) and retires.
//   * the CHAIN workgroup factors the diagonal pairs one after the other (factor_pair_lean, csrc/chol_tile.cuh), keeping the
//     factor, its inverse and the panel of the next pair on chip: X = A(p, p-1) Vp^T, T(p, p) -= X X^T, factor, invert,
//     publish Vp_p, next.  Its inputs A(p, p-1) and T(p, p) come from their owners one step earlier (counter chainin[p]).
//   * the back-substitution rides along as in csrc/chol.hip: an identity block appended under H turns, super-tile by
//     super-tile, into L^-T; the owners of the appended super-tiles (R, np-1) accumulate x_R += (L^-T)_{R,s} y_s in registers
//     and write delta at the end -- no pass over the factor, no second launch.
//
// Memory protocol (measured with scripts/micro/handoff.hip, profiles/r5_handoff.txt): data that crosses workgroups is written
// ONCE with agent-scope (sc1, write-through) stores, followed by s_waitcnt vmcnt(0), a workgroup barrier and one agent-scope
// atomic on a counter; it is read -- after polling the counter -- with plain loads, for the first time in this launch by that
// XCD (nothing is ever re-read after an update by another workgroup), so no stale line can sit in an XCD-private L2 and no
// release / acquire fence (2.7 - 4.3 us / 1.5 us each, measured) is needed.  Plain stores instead of sc1 stores DO produce
// stale reads (same measurement).  Every wait is bounded: a time-out (~2 s; workgroups not co-resident, device wedged) sets
// the error word, every workgroup leaves, and info becomes -1.
#include "chol_tile.cuh"
#include "../../include/como_hip.h"
#include <cstdlib>

namespace como {

constexpr int PB = 64;                 // pair / super-tile width
constexpr int PLD = 65;                // LDS leading dimension of a super-tile (same bank pattern as CLD = 33)
constexpr int PSZ = PB * PLD;          // one super-tile in LDS (doubles)
constexpr int CP_THREADS = 512;           // 8 waves: wave w owns the quadrants (w >> 1, 2 (w & 1)) and (w >> 1, 2 (w & 1) + 1) of a super-tile
constexpr int CP_LPT = PB * PB / CP_THREADS;       // elements per thread of a super-tile copy
constexpr int CP_LDS_DOUBLES = NT2 * TSZ + 2 * PSZ + TSZ + 64;     // chain: factor tiles | Vp | A | Y | flags;  workers: Vp | A_I | A_J | misc

struct CholpArgs {
  double* W;            // working copy (2 Dp x Dp): H lower + g row + identity pad | appended identity rows (never initialised)
  double* Vpg;          // published pair inverses, np x [64][64]
  double* ylast;        // z = Vp_last^T y_last (64): the last pair's contribution to every delta_R is S_R z
  unsigned* sync;       // counters (cholp_sync_words)
  double* delta;
  int* info;
  int Dp, D, np;
  int dbg_stall;        // test switch (como_chol_debug_stall): the chain workgroup leaves at once -> every other workgroup's wait times out
};

#ifdef COMO_CP_PROFILE                         // scripts/micro/cholp_stamps.hip: wall-clock stamps (100 MHz) of the chain workgroup, [pair][8]
__device__ long long* cp_stamps = nullptr;
#define CP_STAMP(p, k) do { if (threadIdx.x == 0 && cp_stamps) cp_stamps[(p) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define CP_STAMP(p, k) do { } while (0)
#endif

__device__ __forceinline__ unsigned cp_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// agent-scope (write-through) store: visible to the other XCDs once acknowledged
__device__ __forceinline__ void cp_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned* cp_pairflag(const CholpArgs& a) { return a.sync; }
__device__ __forceinline__ unsigned* cp_colcnt(const CholpArgs& a, int c) { return a.sync + (1 + c) * CHOLP_SYNC_STRIDE; }
__device__ __forceinline__ unsigned* cp_chainin(const CholpArgs& a, int p) { return a.sync + (1 + a.np + p) * CHOLP_SYNC_STRIDE; }
__device__ __forceinline__ unsigned* cp_err(const CholpArgs& a) { return a.sync + (1 + 2 * a.np) * CHOLP_SYNC_STRIDE; }

// Wait until *p >= target (thread 0 polls, everybody gets the verdict).  false: timed out / another workgroup did.
__device__ __forceinline__ bool cp_wait(const unsigned* p, unsigned target, const CholpArgs& a, volatile int* ok_s) {
  if (threadIdx.x == 0) {
    unsigned* errf = cp_err(a);
    int ok = 1;
    for (long spin = 0; cp_ld(p) < target; ++spin) {
      if ((spin & 63) == 63 && cp_ld(errf)) { ok = 0; break; }
      if (spin > 1500000) { atomicExch(errf, 1u); ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) atomicCAS(a.info, 0, -1);
    *ok_s = ok;
  }
  __syncthreads();
  const bool r = *ok_s != 0;
  __syncthreads();
  return r;
}

// every thread has issued its sc1 stores: wait for their acknowledgement, then one arrival on each counter
__device__ __forceinline__ void cp_signal(unsigned* c0, unsigned* c1 = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c1) __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// acc[j] (+/-)= sum over the 32-wide k blocks kb0 .. kb1-1 of X[16 qr + r][k] Y[16 (qc0 + j) + r][k], j = 0, 1  (super-tiles in LDS,
// leading dimension PLD): the NT product of csrc/chol_tile.cuh's tile_nt_mfma on 64-wide operands, two quadrants of one quadrant
// row per wave (the X operand is read once); lane l feeds row r = l & 15, k group l >> 4.
template <bool NEG>
__device__ __forceinline__ void st_nt(const double* X, const double* Y, int qr, int qc0, int l, int kb0, int kb1, d4_t (&acc)[2],
                                      int nq = 2) {
  const int r = l & 15, q = l >> 4, ko = 16 * (q & 1) + 8 * (q >> 1);
  const double* x = X + (16 * qr + r) * PLD + ko;
  const double* y0 = Y + (16 * qc0 + r) * PLD + ko;
  const double* y1 = y0 + (nq > 1 ? 16 * PLD : 0);
  for (int kb = kb0; kb < kb1; ++kb) {
    double xa[8], ya[8], yb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { xa[s] = x[32 * kb + s]; ya[s] = y0[32 * kb + s]; yb[s] = y1[32 * kb + s]; }
    if (nq > 1) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double xs = NEG ? -xa[s] : xa[s];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, ya[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, yb[s], acc[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -xa[s] : xa[s], ya[s], acc[0], 0, 0, 0);
    }
  }
}
// The LAST k block of X = A Vp^T for the quadrant pair (qr, qc0), (qr, qc0 + 1) (qc0 even): Vp is lower triangular, so quadrant
// qc0 + 1 needs the whole 32-wide block kb = qc0 / 2 but quadrant qc0 only its first 16 columns -- four matrix instructions with the
// k mapping 4 s + q instead of eight (the 64-bit reads of this mapping are 2-4-way bank conflicted: eight extra reads per wave).
__device__ __forceinline__ void st_nt_tail(const double* X, const double* Y, int qr, int qc0, int l, d4_t (&acc)[2]) {
  const int r = l & 15, q = l >> 4, ko = 16 * (q & 1) + 8 * (q >> 1), kb = qc0 >> 1;
  const double* x = X + (16 * qr + r) * PLD + 32 * kb;
  const double* y0 = Y + (16 * qc0 + r) * PLD + 32 * kb;
  const double* y1 = y0 + 16 * PLD;
  double xa[8], yb[8], xh[4], yh[4];
#pragma unroll
  for (int s = 0; s < 8; ++s) { xa[s] = x[ko + s]; yb[s] = y1[ko + s]; }
#pragma unroll
  for (int s = 0; s < 4; ++s) { xh[s] = x[4 * s + q]; yh[s] = y0[4 * s + q]; }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s], yb[s], acc[1], 0, 0, 0);
    if (s < 4) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xh[s], yh[s], acc[0], 0, 0, 0);
  }
}

// element (row, col) inside the super-tile of accumulator register i of lane l, quadrant (qr, qc)
__device__ __forceinline__ int srow(int qr, int l, int i) { return 16 * qr + (l >> 4) + 4 * i; }
__device__ __forceinline__ int scol(int qc, int l) { return 16 * qc + (l & 15); }

__device__ __forceinline__ void st_store(double* dst, int qr, int qc0, int l, const d4_t (&a)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[srow(qr, l, i) * PLD + scol(qc0 + j, l)] = a[j][i];
}

// 64 x 64 block of a row-major matrix (leading dimension ld) -> LDS super-tile
__device__ __forceinline__ void st_load(double* dst, const double* __restrict__ src, long ld) {
  double v[CP_LPT];
#pragma unroll
  for (int u = 0; u < CP_LPT; ++u) { const int e = threadIdx.x + CP_THREADS * u; v[u] = src[(long)(e >> 6) * ld + (e & 63)]; }
#pragma unroll
  for (int u = 0; u < CP_LPT; ++u) { const int e = threadIdx.x + CP_THREADS * u; dst[(e >> 6) * PLD + (e & 63)] = v[u]; }
}

// ------------------------------------------------------------------------------------------------------------------------
// chain workgroup
//
// What the two spare waves of the chain workgroup do DURING a factorisation (factor_pair_lean's Idle hook; one call per barrier
// step, must not block): wave 5 forms Y = L10 V0 (one quadrant per step from step 10 on: L10 and V0 are complete after barrier 9),
// so that the off-diagonal block of the pair's inverse, V10 = -V1 Y, is ONE product after the factorisation; both waves poll the
// counter of the next pair's inputs -- three polls in flight, each consumed three steps (1.6 us) after its issue, i.e. when it has
// long returned -- and, once the inputs are published, start the direct-to-LDS loads of A(q, q-1); three steps later they have landed.
#ifndef COMO_CP_PFW
#define COMO_CP_PFW 2          // waves that prefetch (2: waves 5 and 7, rows interleaved; 1: wave 5 alone, half the rows per step)
#endif
struct CpIdle {
  const double* L10; const double* V0; double* Yb; double* Vp;
  const double* Asrc; long ld; double* Ab;           // the next pair's A(q, q-1) (nullptr: there is no next pair)
  const unsigned* flag;                              // chainin[q] (nullptr: original data, nothing to wait for)
  int* done;                                         // LDS: done[i] = 1 once wave i has put its half of A into Ab
  unsigned pv[3] = {0u, 0u, 0u};                     // polls in flight: slot s % 3 holds the one issued at step s - 3
  bool ready = false, written = false;
  int issued = 1 << 20;
  // A(q, q-1) goes from memory STRAIGHT into LDS (global_load_lds_dword: no registers, nothing for the wave to do when the data
  // arrives): one instruction moves 256 contiguous bytes = half a row of the super-tile (the LDS rows are padded to 65 doubles, so a
  // wider form cannot span two rows); wave i takes rows i, i + 2, ...  `store` = the data has landed: tell the workgroup.
  __device__ __forceinline__ void load(int i) {              // rows i, i + 2, ... (32 rows, 64 instructions)
    const char* src = (const char*)(Asrc + (long)i * ld) + 4 * (threadIdx.x & 63);
    double* dst = Ab + i * PLD;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 4, 256, 0);
      src += 16 * ld;
      dst += 2 * PLD;
    }
  }
  __device__ __forceinline__ void store(int i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) done[i] = 1;
    written = true;
  }
  __device__ __forceinline__ void operator()(int i, int s) {
    if (i == 0 && s >= 10 && s < 14) {
      d4_t y = {0.0, 0.0, 0.0, 0.0};
      tile_mm_mfma<false, false>(L10, V0, s - 10, threadIdx.x & 63, y);
      tile_store_mfma(Yb, s - 10, threadIdx.x & 63, y);
    }
    if (i == 1 && s >= 10 && s < 14) {               // rows 8 (s - 10) .. + 7 of [V0 | 0] into the 64-wide inverse (zeros above the diagonal)
      const int lane = threadIdx.x & 63, c = lane & 31;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = 8 * (s - 10) + 2 * u + (lane >> 5);
        Vp[r * PLD + c] = c <= r ? V0[r * CLD + c] : 0.0;
        Vp[r * PLD + 32 + c] = 0.0;
      }
    }
    if (!Asrc || (COMO_CP_PFW == 1 && i == 1)) return;
    if (!ready) {
      if (flag) { const unsigned v = pv[s % 3]; pv[s % 3] = cp_ld(flag); ready = v >= 2u; }
      else ready = true;
      if (ready) { load(COMO_CP_PFW == 1 ? 0 : i); issued = s; }
    } else if (COMO_CP_PFW == 1 && s == issued + 1) {
      load(1);
    } else if (!written && s >= issued + 4) {
      store(i);
      if (COMO_CP_PFW == 1 && (threadIdx.x & 63) == 0) done[1] = 1;
    }
  }
  __device__ __forceinline__ void finish(int i) {
    if (!Asrc || (COMO_CP_PFW == 1 && i == 1)) return;
    if (ready && COMO_CP_PFW == 1 && issued == 16) load(1);          // (detected at the last step: the second half was never issued)
    if (ready && !written) { store(i); if (COMO_CP_PFW == 1 && (threadIdx.x & 63) == 0) done[1] = 1; }
  }
};

__device__ __forceinline__ void cp_chain(const CholpArgs& a, double* dsm) {
  double* sm = dsm;                               // factor tiles 0 .. 6 (CLD layout): T10 / L10, T00 / L00, T11 / L11, V0, V1, scratch
  double* Vp = dsm + NT2 * TSZ;                   // inverse of the last factored pair, [64][PLD]
  double* Ab = Vp + PSZ;                          // A(q, q-1) -> X
  double* Yb = Ab + PSZ;                          // L10 V0 (one tile)
  volatile int* ok_s = (volatile int*)(Yb + TSZ);
  int* done = (int*)(Yb + TSZ) + 2;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const long Dp = a.Dp;
  const int np = a.np, D = a.D;
  // Quadrants per wave, balanced over the SIMDs (waves w and w + 4 share one; a float64 matrix instruction holds its pipe for 64
  // cycles, so a stage is bound by the quadrant-products of its fullest SIMD).  X stage: 16 quadrants, those of the right half
  // cost two k blocks; T stage: the 10 lower quadrants.
  const int xqr = wv >> 1, xqc0 = 2 * ((wv ^ (wv >> 2)) & 1);
  const int tqr = (0x02101233 >> (4 * wv)) & 15, tqc0 = (0x02100020 >> (4 * wv)) & 15, tnq = (0x01111222 >> (4 * wv)) & 15;
  constexpr int PUBW = 7;                         // the publishing wave: 16 matrix instructions in the X stage, none in the T stage
  // pair 0 straight from the working copy
  for (int e = tid; e < 3 * CB * CB; e += CP_THREADS) {
    const int t = e >> 10, r = (e >> 5) & 31, c = e & 31;
    const long gr = (t == 1 ? 0 : CB) + r, gc = (t == 2 ? CB : 0) + c;     // tile 1 = (0,0), tile 0 = (1,0), tile 2 = (1,1)
    sm[t * TSZ + r * CLD + c] = a.W[gr * Dp + gc];
  }
  if (tid < 2) done[tid] = 0;
  __syncthreads();
  for (int p = 0; p < np; ++p) {
    const int q = p + 1;                          // the pair whose inputs are fetched while this one is factored
    CP_STAMP(p, 4);
    CpIdle idle;
    idle.L10 = sm; idle.V0 = sm + 3 * TSZ; idle.Yb = Yb; idle.Vp = Vp;
    idle.Asrc = q < np ? a.W + (long)q * PB * Dp + (long)p * PB : nullptr;
    idle.ld = Dp; idle.Ab = Ab;
    idle.flag = q >= 2 ? cp_chainin(a, q) : nullptr;
    idle.done = done;
    factor_pair_lean(sm, true, 2 * p, (double*)nullptr, (double*)nullptr, (int)Dp, D, a.info, sm, sm + 3 * TSZ, idle);
    if (wv == 5 || wv == 7) idle.finish(wv == 7 ? 1 : 0);
    __syncthreads();
    CP_STAMP(p, 5);
    const double* L10 = sm;
    const double* L00 = sm + 1 * TSZ;
    const double* L11 = sm + 2 * TSZ;
    const double* V0 = sm + 3 * TSZ;
    const double* V1 = sm + 4 * TSZ;
    d4_t tq[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (q < np) {
      if (!(done[0] && done[1])) {                // the inputs were not there in time: wait for them now (every wave: uniform verdict)
        if (q >= 2 && !cp_wait(cp_chainin(a, q), 2u, a, ok_s)) return;
        st_load(Ab, a.W + (long)q * PB * Dp + (long)p * PB, Dp);
      }
      CP_STAMP(q, 1);
      // the lower quadrants of T(q, q), updated through pair p - 1 by their owner: in flight during the next two stages
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j < tnq) {
#pragma unroll
          for (int i = 0; i < 4; ++i) tq[j][i] = a.W[((long)q * PB + srow(tqr, l, i)) * Dp + (long)q * PB + scol(tqc0 + j, l)];
        }
    }
    // V10 = -V1 Y, [V0 | 0] and V1 into the 64-wide layout (zeros above the diagonals)
    if (wv < 4) {
      d4_t v = {0.0, 0.0, 0.0, 0.0};
      tile_mm_mfma<false, false>(V1, Yb, wv, l, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) Vp[(32 + mrow(wv, l, i)) * PLD + mcol(wv, l)] = -v[i];
    } else {                                                         // ([V0 | 0] was put there during the factorisation: 1.3 -> 1.0 us)
      for (int e = tid - 256; e < CB * CB; e += 256) {
        const int r = e >> 5, c = e & 31;
        Vp[(32 + r) * PLD + 32 + c] = c <= r ? V1[r * CLD + c] : 0.0;
      }
    }
    if (tid < 2) done[tid] = 0;
    lds_only_barrier();                                              // (not __syncthreads: the loads of T(q, q) stay in flight)
    CP_STAMP(p, 6);
    // Publish Vp_p: ONE wave issues all the agent-scope stores and goes on; it collects their acknowledgement (~2 us) when it is
    // idle during the T stage and raises the pair counter itself -- nobody else ever waits for these stores.
    if (q < np && wv == PUBW) {                                      // (nobody reads the last pair's inverse: its product with y is published)
      double* dst = a.Vpg + (long)p * PB * PB + l;
      const double* src = Vp + l;
#pragma unroll 1
      for (int b = 0; b < 4; ++b) {                                  // 16 rows per batch: all LDS reads, then all stores
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(16 * b + u) * PLD];
#pragma unroll
        for (int u = 0; u < 16; ++u) cp_st(&dst[(16 * b + u) * PB], v[u]);
      }
    }
    if (q == np) {
      // y of the last pair = row D of L (the appended right-hand side), columns below it; delta of the last pair's own rows:
      // its appended super-tile is still the identity, (L^-T)_{last,last} = Vp^T
      const int gl = D - PB * (np - 1);
      double* yl = Ab;
      if (tid < PB) {
        double v = 0.0;
        if (tid < gl) v = gl < CB ? (tid < CB ? L00[gl * CLD + tid] : 0.0)
                                  : (tid < CB ? L10[(gl - CB) * CLD + tid] : L11[(gl - CB) * CLD + tid - CB]);
        yl[tid] = v;
      }
      __syncthreads();
      if (tid < PB) {                                                // z = Vp^T y: delta of the last pair's rows (zero from row gl on), and
        double s = 0.0;                                              // what every appended row block still has to add: delta_R = x_R + S_R z
        for (int k = 0; k < PB; ++k) s = __builtin_fma(Vp[k * PLD + tid], yl[k], s);
        cp_st(&a.ylast[tid], s);
        if (tid < gl) a.delta[(long)PB * (np - 1) + tid] = s;
      }
      cp_signal(cp_pairflag(a));
      CP_STAMP(p, 7);
      return;
    }
    // X = A Vp^T (Vp lower triangular: columns >= 32 of its rows < 32 are zero).  The publishing wave is busy issuing its stores:
    // its two quadrants go to the wave it shares a SIMD with (the matrix pipe of that SIMD sees the same 48 instructions).
    d4_t x[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, x2[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    constexpr int PQR = PUBW >> 1, PQC0 = 2 * ((PUBW ^ (PUBW >> 2)) & 1);
    if (wv != PUBW) { st_nt<false>(Ab, Vp, xqr, xqc0, l, 0, xqc0 >> 1, x); st_nt_tail(Ab, Vp, xqr, xqc0, l, x); }
    if (wv == PUBW - 4) { st_nt<false>(Ab, Vp, PQR, PQC0, l, 0, PQC0 >> 1, x2); st_nt_tail(Ab, Vp, PQR, PQC0, l, x2); }
    lds_only_barrier();                                              // (LDS only: the published stores and the loads of T stay in flight)
    CP_STAMP(q, 2);
    if (wv != PUBW) st_store(Ab, xqr, xqc0, l, x);
    if (wv == PUBW - 4) st_store(Ab, PQR, PQC0, l, x2);
    lds_only_barrier();
    CP_STAMP(q, 3);
    if (wv == PUBW) {                                                // (the publishing wave owns no quadrant of the T stage)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (l == 0) __hip_atomic_fetch_add(cp_pairflag(a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tnq > 0) {                                                   // T -= X X^T (lower quadrants), into the factor's tiles
      st_nt<true>(Ab, Ab, tqr, tqc0, l, 0, 2, tq, tnq);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j < tnq) {
          const int t = tqr < 2 ? 1 : (tqc0 + j < 2 ? 0 : 2);
#pragma unroll
          for (int i = 0; i < 4; ++i) sm[t * TSZ + (srow(tqr, l, i) & 31) * CLD + (scol(tqc0 + j, l) & 31)] = tq[j][i];
        }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The owners.  Super-tiles are enumerated COLUMN by column (column c = 1 .. np-1: the regular (I, c), I > c; the diagonal (c, c)
// from c = 2 on; the appended (R, c), R < c) and dealt round-robin to the worker workgroups, so that a workgroup's tiles are sorted by
// the column pair at which they retire -- the tile the chain or the next step waits for is always the first one it works on.  Up to
// 16 column pairs every super-tile has its own workgroup (MAXT = 1, the metric window); larger systems (the full sliding windows of
// the odometry loop at D ~ 1300; up to config 4 at D = 2680: 1722 super-tiles) give each of the 255 workers up to MAXT = 8 of them, all
// register-resident, processed one after the other within a step (the pair inverse is fetched once per step).  By default the
// launcher stops at CHOLP_RUN_MAX_NP = 34 pairs: at 42 the owners' three products per super-tile step lose to the multi-launch
// solver's shared panels (1634 vs 1504 us, profiles/r5_chol_time.txt); COMO_CHOLP_MAX_NP moves the cross-over.
struct CpTile {
  int I, J, s_first, s_last;
  bool app, diag, lastcol;
  long row0;
};
__device__ __forceinline__ int cp_ntiles(int np) { return (np - 1) + np * (np - 2); }
__device__ __forceinline__ CpTile cp_tile(int t, int np, long Dp) {
  CpTile T;
  int c, o;
  if (t < np - 1) { c = 1; o = t; } else { const int u = t - (np - 1); c = 2 + u / np; o = u - (c - 2) * np; }
  const int nreg = np - 1 - c, hasdiag = c >= 2 ? 1 : 0;
  if (o < nreg) { T.app = false; T.I = c + 1 + o; }
  else if (hasdiag && o == nreg) { T.app = false; T.I = c; }
  else { T.app = true; T.I = o - nreg - hasdiag; }
  T.J = c;
  T.diag = !T.app && T.I == T.J;
  T.s_first = T.app ? T.I : 0;
  T.s_last = T.diag ? T.J - 2 : T.J - 1;
  T.lastcol = T.app && T.J == np - 1;
  T.row0 = (T.app ? Dp : 0) + (long)T.I * PB;                       // first row of the super-tile in the working copy
  return T;
}

template <int MAXT>
__device__ __forceinline__ void cp_workers_body(const CholpArgs& a, double* dsm, int w, int nworkers) {
  double* Vp = dsm;
  double* AI = dsm + PSZ;
  double* AJ = dsm + 2 * PSZ;
  double* misc = dsm + 3 * PSZ;                   // z [64] | verdict word
  volatile int* ok_s = (volatile int*)(misc + 64);
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, qr = wv >> 1, qc0 = 2 * (wv & 1);
  const long Dp = a.Dp;
  const int np = a.np, D = a.D;
  const int gl = D - PB * (np - 1);
  const int ntot = cp_ntiles(np);
  CpTile tl[MAXT];
  bool have[MAXT];
  d4_t S[MAXT][2];
  double xr[MAXT];                                                   // thread t < 64 of an owner (R, np-1): x_R[t]
  int s_lo = np, s_hi = -1;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int t = w + k * nworkers;
    have[k] = t < ntot;
    tl[k] = cp_tile(have[k] ? t : 0, np, Dp);
    xr[k] = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        S[k][j][i] = (have[k] && !tl[k].app) ? a.W[(tl[k].row0 + srow(qr, l, i)) * Dp + (long)tl[k].J * PB + scol(qc0 + j, l)] : 0.0;
    if (have[k]) { s_lo = min(s_lo, tl[k].s_first); s_hi = max(s_hi, tl[k].s_last); }
  }
  for (int s = s_lo; s <= s_hi; ++s) {
    bool waited = false, have_vp = false;
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
      const CpTile& T = tl[k];
      if (!have[k] || s < T.s_first || s > T.s_last) continue;      // (uniform over the workgroup)
      if (!waited) {                                                  // every super-tile of column pair s is published
        if (s >= 1 && !cp_wait(cp_colcnt(a, s), (unsigned)(np - 1), a, ok_s)) return;
        waited = true;
      }
      // panels of this step: A(I, s) (the identity for an appended row block that meets its own column pair) and A(J, s)
      if (T.app && s == T.I) {
#pragma unroll
        for (int u = 0; u < CP_LPT; ++u) { const int e = tid + CP_THREADS * u; AI[(e >> 6) * PLD + (e & 63)] = (e >> 6) == (e & 63) ? 1.0 : 0.0; }
      } else {
        st_load(AI, a.W + T.row0 * Dp + (long)s * PB, Dp);
      }
      if (!T.diag) st_load(AJ, a.W + (long)T.J * PB * Dp + (long)s * PB, Dp);
      if (!have_vp) {                                                 // the pair inverse: once per step
        if (!cp_wait(cp_pairflag(a), (unsigned)(s + 1), a, ok_s)) return;
        st_load(Vp, a.Vpg + (long)s * PB * PB, PB);
        have_vp = true;
      }
      __syncthreads();
      d4_t xi[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, xj[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
      st_nt<false>(AI, Vp, qr, qc0, l, 0, qc0 >> 1, xi);
      st_nt_tail(AI, Vp, qr, qc0, l, xi);
      if (!T.diag) { st_nt<false>(AJ, Vp, qr, qc0, l, 0, qc0 >> 1, xj); st_nt_tail(AJ, Vp, qr, qc0, l, xj); }
      __syncthreads();
      st_store(AI, qr, qc0, l, xi);
      if (!T.diag) st_store(AJ, qr, qc0, l, xj);
      __syncthreads();
      st_nt<true>(AI, T.diag ? AI : AJ, qr, qc0, l, 0, 2, S[k]);    // S -= X_I X_J^T
      if (T.lastcol && tid < PB) {                                   // x_R += (L^-T)_{R,s} y_s;  y_s = row gl of X_J (J = np - 1)
        double acc = 0.0;
        for (int q = 0; q < PB; ++q) acc = __builtin_fma(AI[tid * PLD + q], AJ[gl * PLD + q], acc);
        xr[k] += acc;
      }
      if (s == T.s_last && !T.lastcol) {
        // publish: this super-tile is the panel A(I, J) of every later step (a diagonal one goes to the chain only)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) cp_st(&a.W[(T.row0 + srow(qr, l, i)) * Dp + (long)T.J * PB + scol(qc0 + j, l)], S[k][j][i]);
        if (T.diag) cp_signal(cp_chainin(a, T.I));
        else if (!T.app && T.I == T.J + 1) cp_signal(cp_colcnt(a, T.J), cp_chainin(a, T.I));
        else cp_signal(cp_colcnt(a, T.J));
      }
      __syncthreads();
    }
  }
  // the last pair: delta_R = x_R + (L^-T)_{R,last} y_last = x_R + S (Vp_last^T y_last); the chain publishes z = Vp_last^T y_last
  bool got = false;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    if (!have[k] || !tl[k].lastcol) continue;
    st_store(AI, qr, qc0, l, S[k]);
    if (!got) {
      if (!cp_wait(cp_pairflag(a), (unsigned)np, a, ok_s)) return;
      if (tid < PB) misc[tid] = a.ylast[tid];
      got = true;
    }
    __syncthreads();
    if (tid < PB) {
      double acc = 0.0;
      for (int q = 0; q < PB; ++q) acc = __builtin_fma(AI[tid * PLD + q], misc[q], acc);
      a.delta[(long)tl[k].I * PB + tid] = xr[k] + acc;
    }
    __syncthreads();
  }
}

// (many super-tiles per workgroup: the owners' accumulator arrays must not take registers from the chain's waves -- the owners are
// a function of their own there; with one or two super-tiles everything is one body, as measured fastest)
template <int MAXT>
__device__ __attribute__((noinline)) void cp_workers_call(CholpArgs a, double* dsm, int w, int nworkers) {
  cp_workers_body<MAXT>(a, dsm, w, nworkers);
}

template <int MAXT>
__global__ __launch_bounds__(CP_THREADS) void cholp_kernel(CholpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  if (blockIdx.x == 0) { if (!a.dbg_stall) cp_chain(a, dsm); }
  else if (MAXT <= 2) cp_workers_body<MAXT>(a, dsm, (int)blockIdx.x - 1, (int)gridDim.x - 1);
  else cp_workers_call<MAXT>(a, dsm, (int)blockIdx.x - 1, (int)gridDim.x - 1);
}

int cholp_tiles(int np) { return (np - 1) + np * (np - 2); }

// compute units of the device if the kernel's LDS request was accepted, else 0 (called from como_chol_workspace_bytes, i.e.
// outside any stream capture, before the first solve)
int cholp_init() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    const int bytes = CP_LDS_DOUBLES * (int)sizeof(double);
    if (hipFuncSetAttribute((const void*)cholp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)cholp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)cholp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)cholp_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      n = 0;
    // every workgroup of the launch (at most one per compute unit) must be resident at once: the kernel spin-waits on its peers
    int occ = 0;
    if (n > 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)cholp_kernel<1>, CP_THREADS, (size_t)bytes) != hipSuccess || occ < 1 ||
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)cholp_kernel<2>, CP_THREADS, (size_t)bytes) != hipSuccess || occ < 1 ||
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)cholp_kernel<4>, CP_THREADS, (size_t)bytes) != hipSuccess || occ < 1 ||
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)cholp_kernel<8>, CP_THREADS, (size_t)bytes) != hipSuccess || occ < 1))
      n = 0;
    (void)hipGetLastError();
    const char* e = getenv("COMO_CHOLP");                  // COMO_CHOLP=0: the multi-launch solver (the fallback), for A/B runs
    if (e && atoi(e) == 0) n = 0;
    return n;
  }();
  return cus;
}

// The persistent solve of a system that is already packed into `workspace` (chol_pack_kernel / como_sys_finalize_pack: they also
// reset info and the counters).  COMO_ERR_ARG when the size or the device does not allow it (the caller falls back to the
// multi-launch solver).
static int g_cholp_enabled = 1;        // como_chol_set_persistent
static int g_cholp_stall = 0;          // como_chol_debug_stall
void cholp_set_stall(int on) { g_cholp_stall = on ? 1 : 0; }
int cholp_set_enabled(int e) { const int p = g_cholp_enabled; g_cholp_enabled = e ? 1 : 0; return p; }
int cholp_enabled() { return g_cholp_enabled && cholp_init() >= 2; }

int cholp_solve(double* delta, void* workspace, int D, int* info, hipStream_t s) {
  if (!g_cholp_enabled || !cholp_size_ok(D)) return COMO_ERR_ARG;
  const int np = chol_np(D);
  const int cus = cholp_init();
  if (cus < 2) return COMO_ERR_ARG;
  const int T = cholp_tiles(np);
  const int workers = T < cus - 1 ? T : cus - 1;           // one workgroup per compute unit, all co-resident: chain + workers
  const int per = (T + workers - 1) / workers;
  if (per > 8) return COMO_ERR_ARG;
  static const int maxnp = [] { const char* e = getenv("COMO_CHOLP_MAX_NP"); return e ? atoi(e) : CHOLP_RUN_MAX_NP; }();
  if (np > maxnp) return COMO_ERR_ARG;
  CholpArgs a;
  a.Dp = 64 * np; a.D = D; a.np = np;
  a.W = (double*)workspace;
  a.Vpg = a.W + 2L * a.Dp * a.Dp;
  a.ylast = a.Vpg + (long)np * PB * PB;
  a.sync = (unsigned*)(a.W + cholp_sync_offset(a.Dp, np));
  a.delta = delta;
  a.info = info;
  a.dbg_stall = g_cholp_stall;
  const dim3 grid(1 + workers), blk(CP_THREADS);
  const size_t lds = CP_LDS_DOUBLES * sizeof(double);
  if (per <= 1) hipLaunchKernelGGL(cholp_kernel<1>, grid, blk, lds, s, a);
  else if (per <= 2) hipLaunchKernelGGL(cholp_kernel<2>, grid, blk, lds, s, a);
  else if (per <= 4) hipLaunchKernelGGL(cholp_kernel<4>, grid, blk, lds, s, a);
  else hipLaunchKernelGGL(cholp_kernel<8>, grid, blk, lds, s, a);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // namespace como
