// Fused O(B*m) bookkeeping of one window-BA Gauss-Newton iteration.
//
// In the reference these are ~400 tiny PyTorch launches per iteration (landmark projection, prior factors, variable
// update); here they are three kernels, so the whole iteration is ~45 launches and stays graph-capturable:
//
//   win_scaffold : per keyframe b, lane j = inducing point j: landmark -> camera frame (exact op order), re-init test,
//                  z, log z, pixel p and the Jacobians the priors and the dense-reference kernel need.
//                  reference: como/odom/backend/sparse_map.py:18-60 (project_landmarks), Mapping.py:603-659.
//   win_priors   : all prior factors of Mapping.iterate (Mapping.py:809-917) accumulated into H, g:
//                  GP marginal-likelihood prior (factors/gp_priors.py:7-81), log-depth prior mode "first_mean"
//                  (depth_prior.py:7-141), pixel prior mode "first" (pixel_prior.py:6-130), pose anchor
//                  (pose_prior_factors.py:5-19), affine anchors and fixed-landmark anchors / mean-log-depth prior
//                  (scalar_prior_factors.py:4-34, gp_priors.py:84-150).
//                  The log-depth-space priors share the residual r0 = logz_m - log(median), so they collapse to
//                  M = K_mm^-1/s_gp^2 + diag(first/s_ld^2), w = M r0, chained once through dlogz/dT, dlogz/dP.
//   win_update   : T <- T Exp(delta), affine += delta, P += delta (linear_system.py:115-152).
#include "common.cuh"
#include "../../include/como_hip.h"

namespace como {

struct ScaffoldOut {
  double* pm;         // (B,m,2)
  double* logzm;      // (B,m)
  double* invz;       // (B,m)      dlogz/dz = 1/z
  double* dzdP;       // (B,3)      R_cw[2,:]
  double* dlogz_dT;   // (B,m,6)
  double* dlogz_dP;   // (B,m,3)
  double* dp_dP;      // (B,m,6)    2x3
  double* dp_dT;      // (B,m,12)   2x6
  double* init_Pm;    // (L,3)      re-initialisation point of every landmark
  int* reinit_flag;   // (L)        1 if the landmark's first observer re-initialised it
};

// pix-dtype mirrors for the per-pixel kernels
template <typename TP>
struct ScaffoldPix {
  TP* logzm;      // (B,m)
  TP* invz;       // (B,m)
  TP* dzdP;       // (B,3)
  TP* dlogz_dT;   // (B,m,6)
  TP* poses;      // (F,16)
  TP* aff;        // (F,2)
};

template <typename TP>
__global__ __launch_bounds__(64) void win_scaffold_kernel(
    const double* __restrict__ poses, const double* __restrict__ aff, int F, const double* __restrict__ P_m,
    const int* __restrict__ lm_ids, const int* __restrict__ first_frame, const int* __restrict__ first_slot,
    const double* __restrict__ Kmat, const double* __restrict__ median, const double* __restrict__ pm_first, int B, int m,
    ScaffoldOut o, ScaffoldPix<TP> px, uint4* __restrict__ za, long za16, uint4* __restrict__ zb, long zb16,
    double* __restrict__ err8, uint4* __restrict__ zc, long zc16) {
  const int b = blockIdx.x, j = threadIdx.x;
  // this iteration's accumulators that would otherwise each need a fill launch (~4.5 us apiece)
  for (long e = (long)b * 64 + j; e < za16; e += (long)gridDim.x * 64) za[e] = uint4{0u, 0u, 0u, 0u};
  for (long e = (long)b * 64 + j; e < zb16; e += (long)gridDim.x * 64) zb[e] = uint4{0u, 0u, 0u, 0u};
  for (long e = (long)b * 64 + j; e < zc16; e += (long)gridDim.x * 64) zc[e] = uint4{0u, 0u, 0u, 0u};
  if (err8 && b == 0 && j < 8) err8[j] = 0.0;
  // pix-dtype copies of ALL frame poses / affine params (keyframes then recent frames)
  for (int e = b * 64 + j; e < F * 16; e += gridDim.x * 64) px.poses[e] = (TP)poses[e];
  for (int e = b * 64 + j; e < F * 2; e += gridDim.x * 64) px.aff[e] = (TP)aff[e];
  if (b >= B || j >= m) return;
  double Tcw[12];
  invert_pose34(poses + 16 * (long)b, Tcw);                         // sparse_map.py:20, lie_algebra.py:83-95
  const int l = lm_ids[(long)b * m + j];
  // re-initialisation point: back-projection from the landmark's FIRST observer at that frame's median depth
  const int fb = first_frame[l], fj = first_slot[l];
  const double fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
  double iw[3];
  {
    const double* pf = pm_first + ((long)fb * m + fj) * 2;
    const double zi = median[fb];
    const double rx = (pf[0] - cx) / fx, ry = (pf[1] - cy) / fy;       // camera.py:43-54
    double Mw[12];
    const double* Tw = poses + 16 * (long)fb;
#pragma unroll
    for (int k = 0; k < 12; ++k) Mw[k] = Tw[k];
    rigid_apply(Mw, zi * rx, zi * ry, zi, iw[0], iw[1], iw[2]);
  }
  double X, Y, Z;
  rigid_apply(Tcw, P_m[3 * (long)l], P_m[3 * (long)l + 1], P_m[3 * (long)l + 2], X, Y, Z);
  const bool zbad = Z < 0.1 * median[b];                             // sparse_map.py:26-28
  if (zbad) rigid_apply(Tcw, iw[0], iw[1], iw[2], X, Y, Z);
  if (fb == b && fj == j) {                                          // this thread IS the first observer
    o.init_Pm[3 * (long)l] = iw[0]; o.init_Pm[3 * (long)l + 1] = iw[1]; o.init_Pm[3 * (long)l + 2] = iw[2];
    o.reinit_flag[l] = zbad ? 1 : 0;
  }
  const long bj = (long)b * m + j;
  const double iz = 1.0 / Z;
  const double logz = log(Z);
  o.logzm[bj] = logz;
  o.invz[bj] = iz;
  px.logzm[bj] = (TP)logz;
  px.invz[bj] = (TP)iz;
  double u, v;
  u = project1(fx, X, Z, cx);
  v = project1(fy, Y, Z, cy);
  o.pm[2 * bj] = u;
  o.pm[2 * bj + 1] = v;
  // dz/dT_wc = row 2 of [[P_c]x, -I] ; dlogz = dz / z      (sparse_map.py:46-55)
  const double dzT[6] = {-Y, X, 0.0, 0.0, 0.0, -1.0};
#pragma unroll
  for (int a = 0; a < 6; ++a) { o.dlogz_dT[6 * bj + a] = dzT[a] * iz; px.dlogz_dT[6 * bj + a] = (TP)(dzT[a] * iz); }
  const double r20 = Tcw[8], r21 = Tcw[9], r22 = Tcw[10];
  o.dlogz_dP[3 * bj] = r20 * iz; o.dlogz_dP[3 * bj + 1] = r21 * iz; o.dlogz_dP[3 * bj + 2] = r22 * iz;
  if (j == 0) {
    o.dzdP[3 * b] = r20; o.dzdP[3 * b + 1] = r21; o.dzdP[3 * b + 2] = r22;
    px.dzdP[3 * b] = (TP)r20; px.dzdP[3 * b + 1] = (TP)r21; px.dzdP[3 * b + 2] = (TP)r22;
  }
  // dp/dP_c (camera.py:28-35), then dp/dP_w = dp/dP_c R_cw, dp/dT_wc = dp/dP_c [[P_c]x, -I]
  const double a00 = fx * iz, a02 = -(fx * X * iz) * iz, a11 = fy * iz, a12 = -(fy * Y * iz) * iz;
  double* dP = o.dp_dP + 6 * bj;
  dP[0] = a00 * Tcw[0] + a02 * Tcw[8]; dP[1] = a00 * Tcw[1] + a02 * Tcw[9]; dP[2] = a00 * Tcw[2] + a02 * Tcw[10];
  dP[3] = a11 * Tcw[4] + a12 * Tcw[8]; dP[4] = a11 * Tcw[5] + a12 * Tcw[9]; dP[5] = a11 * Tcw[6] + a12 * Tcw[10];
  double* dT = o.dp_dT + 12 * bj;
  // [P]x = [[0,-Z,Y],[Z,0,-X],[-Y,X,0]]
  dT[0] = a02 * (-Y);            dT[1] = a00 * (-Z) + a02 * X;  dT[2] = a00 * Y;
  dT[3] = -a00;                  dT[4] = 0.0;                   dT[5] = -a02;
  dT[6] = a11 * Z + a12 * (-Y);  dT[7] = a12 * X;               dT[8] = a11 * (-X);
  dT[9] = 0.0;                   dT[10] = -a11;                 dT[11] = -a12;
}

__global__ void win_apply_reinit_kernel(double* __restrict__ P_m, const double* __restrict__ init_Pm,
                                        const int* __restrict__ flag, int L) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < L && flag[l]) {                                             // Mapping.py:645-648
    P_m[3 * (long)l] = init_Pm[3 * (long)l]; P_m[3 * (long)l + 1] = init_Pm[3 * (long)l + 1]; P_m[3 * (long)l + 2] = init_Pm[3 * (long)l + 2];
  }
}

struct PriorArgs {
  const double* logzm; const double* dlogz_dT; const double* dlogz_dP; const double* pm; const double* pm_first;
  const double* dp_dP; const double* dp_dT; const uint8_t* first_mask; const double* Kmm_inv;
  const void* median; int median_is_f32; int median_stride;
  const long* pose_inds;      // (B,8)
  const long* landmark_inds;  // (B,3m)
  const double* poses; const double* aff; const double* pose_anchor; const double* aff_anchor;
  const double* P_m; const double* P_anchor; const long* fix_inds;  // (nfix*3) rows of H ; P_anchor (nfix,3); fix_lm (nfix)
  const int* fix_lm; int nfix;
  double s_gp, s_ld, s_px, s_pose, s_aff, s_lm;   // sigmas
  double* H; double* g; long D; double* err;      // err: 8 doubles
  double* median_out;
  long long* fix; long plane;                     // order-independent mode: fixed-point system buffer (common.cuh fix_add)
  const double* mld_J; const double* mld_anchor; double s_mld;   // filling window: mean-log-depth scale prior on keyframe 0
};

__global__ __launch_bounds__(256) void win_priors_kernel(PriorArgs A, int B, int m) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int MS = m + 1;           // padded row stride: column walks of M are conflict-free
  double* M = sm;                 // m x (m + 1)
  double* MG = M + m * MS;        // m x 6
  double* w = MG + m * 6;         // m
  double* r0 = w + m;             // m
  double* G = r0 + m;             // m x 6
  double* dP = G + m * 6;         // m x 3
  double* red = dP + m * 3;       // 64 scratch (+ m x 21 pixel-prior rows behind it, then 6 + 3m mean-log-depth row)
  // grid (B, S): every slice rebuilds the small m x m prologue in LDS and scatters 1/S of H_TP / H_PP (the 9 m^2
  // fp64 atomics per keyframe were the whole cost with one workgroup per keyframe); slice 0 adds everything else
  const int b = blockIdx.x, tid = threadIdx.x, slice = blockIdx.y, nsl = gridDim.y;
  const bool lead = slice == 0;
  const double med = A.median_is_f32 ? (double)((const float*)A.median)[(long)b * A.median_stride]
                                     : ((const double*)A.median)[(long)b * A.median_stride];
  if (A.median_out && lead && tid == 0) A.median_out[b] = med;      // the next iteration's re-initialisation depth
  const double logmed = log(med);
  const double i_gp = 1.0 / (A.s_gp * A.s_gp), i_ld = 1.0 / (A.s_ld * A.s_ld), i_px = 1.0 / (A.s_px * A.s_px);
  {
    double kin[16];                                  // m <= 64: at most 16 elements per thread, all loads in flight at once
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u;
      kin[u] = (e < m * m) ? A.Kmm_inv[(long)b * m * m + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u;
      if (e < m * m) {
        const int i = e / m, j = e % m;
        double v = kin[u] * i_gp;
        if (i == j && A.first_mask[(long)b * m + i]) v += i_ld;
        M[i * MS + j] = v;
      }
    }
  }
  for (int e = tid; e < m; e += 256) r0[e] = A.logzm[(long)b * m + e] - logmed;
  for (int e = tid; e < m * 6; e += 256) G[e] = A.dlogz_dT[(long)b * m * 6 + e];
  for (int e = tid; e < m * 3; e += 256) dP[e] = A.dlogz_dP[(long)b * m * 3 + e];
  __syncthreads();
  for (int e = tid; e < m; e += 256) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += M[e * MS + k] * r0[k];
    w[e] = s;
  }
  for (int e = tid; e < m * 6; e += 256) {
    const int i = e / 6, a = e % 6;
    double s = 0;
    for (int k = 0; k < m; ++k) s += M[i * MS + k] * G[k * 6 + a];
    MG[e] = s;
  }
  __syncthreads();
  const long* pi = A.pose_inds + 8 * (long)b;
  const long* li = A.landmark_inds + 3 * (long)m * b;
  const long D = A.D;
  // Accumulation.  Floating-point mode: fp64 atomics into H (both triangles), g, err.  Fixed-point mode (A.fix): exact integer
  // atomics, LOWER triangle only (como_sys_finalize mirrors it) -- every symmetric entry pair is added exactly once.
  const bool FX = A.fix != nullptr;
  long long* poison = FX ? A.fix + D * D + D + FIX_POISON : nullptr;
  auto addH = [&](long r, long c, double v) {                 // floating mode: entry (r, c); fixed mode: entry (max, min)
    if (FX) { const long rr = r > c ? r : c, cc = r > c ? c : r; fix_add(A.fix, A.plane, rr * D + cc, v, poison); }
    else atomicAdd(&A.H[r * D + c], v);
  };
  auto addG = [&](long i, double v) {
    if (FX) fix_add(A.fix, A.plane, D * D + i, v, poison);
    else atomicAdd(&A.g[i], v);
  };
  auto addE = [&](int k, double v) {                          // prior errors: slots 1..6 of the fixed-point err block
    if (FX) fix_add(A.fix, A.plane, D * D + D + 1 + k, v, poison);
    else atomicAdd(&A.err[k], v);
  };
  // H_TT, g_T  (log-depth-space priors)
  if (lead && tid < 36) {
    const int a = tid / 6, c = tid % 6;
    double s = 0;
    for (int k = 0; k < m; ++k) s += G[k * 6 + a] * MG[k * 6 + c];
    if (!FX || a >= c) addH(pi[a], pi[c], s);
  } else if (lead && tid < 42) {
    const int a = tid - 36;
    double s = 0;
    for (int k = 0; k < m; ++k) s += G[k * 6 + a] * w[k];
    addG(pi[a], -s);
  }
  // H_TP (both triangles), g_P
  for (int e = tid + 256 * slice; e < 6 * m * 3; e += 256 * nsl) {
    const int a = e / (3 * m), q = e % (3 * m), j = q / 3, d = q % 3;
    const double v = MG[j * 6 + a] * dP[j * 3 + d];
    addH(li[q], pi[a], v);
    if (!FX) addH(pi[a], li[q], v);
  }
  if (lead)
    for (int q = tid; q < 3 * m; q += 256) addG(li[q], -w[q / 3] * dP[q]);
  // H_PP
  for (int e = tid + 256 * slice; e < 9 * m * m; e += 256 * nsl) {
    const int q1 = e / (3 * m), q2 = e % (3 * m);
    const double v = M[(q1 / 3) * MS + (q2 / 3)] * dP[q1] * dP[q2];
    if (!FX || li[q1] >= li[q2]) addH(li[q1], li[q2], v);
  }
  // errors: gp = r0^T (Kinv/s^2) r0, ld = sum first r0^2 / s^2
  if (lead && tid < 64) {
    double egp = 0, eld = 0;
    for (int i = tid; i < m; i += 64) {
      const double fi = A.first_mask[(long)b * m + i] ? i_ld : 0.0;
      eld += fi * r0[i] * r0[i];
      egp += r0[i] * (w[i] - fi * r0[i]);
    }
    egp = wave_sum(egp); eld = wave_sum(eld);
    if (tid == 0) { addE(0, egp); addE(1, eld); }
  }
  // pixel prior, mode "first".  A wave-level fp64 atomic instruction costs ~180 ns whether 2 or 64 lanes are active, so
  // every scatter below is laid out with one ITEM per lane (never a per-lane serial loop of atomics): the 48 landmark-side
  // entries of each first-observed landmark are items q = 48 j + t spread over threads and slices; the 36 + 6 pose-side
  // entries shared by all landmarks are summed by 42 threads (one output each) and added with a single instruction.
  double* px = red + 64;                           // m x 21 {JP 6, JT 12, r 2, act}
  for (int j = tid; j < m; j += 256) {
    const long bj = (long)b * m + j;
    const bool act = A.first_mask[bj];
    double* o = px + 21 * j;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = A.dp_dP[6 * bj + k];
#pragma unroll
    for (int k = 0; k < 12; ++k) o[6 + k] = A.dp_dT[12 * bj + k];
    o[18] = A.pm[2 * bj] - A.pm_first[2 * bj];
    o[19] = A.pm[2 * bj + 1] - A.pm_first[2 * bj + 1];
    o[20] = act ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int q = tid + 256 * slice; q < 48 * m; q += 256 * nsl) {
    const int j = q / 48, t = q % 48;
    const double* o = px + 21 * j;
    if (o[20] == 0.0) continue;
    const double* jp = o;
    const double* jt = o + 6;
    if (t < 9) {
      const int d = t / 3, d2 = t % 3;
      if (!FX || d >= d2) addH(li[3 * j + d], li[3 * j + d2], i_px * (jp[d] * jp[d2] + jp[3 + d] * jp[3 + d2]));
    } else if (t < 12) {
      const int d = t - 9;
      addG(li[3 * j + d], -i_px * (jp[d] * o[18] + jp[3 + d] * o[19]));
    } else {
      const int u = t - 12, d = u / 12, a = (u % 12) >> 1;
      const double v = i_px * (jt[a] * jp[d] + jt[6 + a] * jp[3 + d]);
      if (u & 1) addH(li[3 * j + d], pi[a], v);
      else if (!FX) addH(pi[a], li[3 * j + d], v);
    }
  }
  // mean-log-depth scale prior on keyframe 0 while the window fills (Mapping.py:900-917, gp_priors.py:84-150): ONE residual
  // r = J . logz_m - anchor with the row v = [J G (6) | J_k dlogz_k/dP (3m)]: H += v v^T / s^2 -- (6 + 3m)^2 / 2 entries, one
  // item per lane over all slices of keyframe 0's workgroups
  if (b == 0 && A.mld_J && A.nfix == 0) {
    double* vrow = red + 64 + 21 * m;               // 6 + 3m
    const double im = 1.0 / (A.s_mld * A.s_mld);
    if (tid < 6) {
      double s = 0;
      for (int k = 0; k < m; ++k) s += A.mld_J[k] * G[k * 6 + tid];
      vrow[tid] = s;
    }
    for (int q = tid; q < 3 * m; q += 256) vrow[6 + q] = A.mld_J[q / 3] * dP[q];
    __syncthreads();
    double rr = 0;
    for (int k = 0; k < m; ++k) rr += A.mld_J[k] * A.logzm[k];
    rr -= A.mld_anchor[0];
    const int nv = 6 + 3 * m;
    auto vidx = [&](int i) -> long { return i < 6 ? pi[i] : li[i - 6]; };
    for (int e = tid + 256 * slice; e < nv * nv; e += 256 * nsl) {
      const int i = e / nv, j = e % nv;
      const long ri = vidx(i), rj = vidx(j);
      // (distinct variables of one keyframe have distinct rows, so ri >= rj keeps each symmetric pair exactly once)
      if (!FX || ri >= rj) addH(ri, rj, im * vrow[i] * vrow[j]);
    }
    if (lead) {
      for (int i = tid; i < nv; i += 256) addG(vidx(i), -im * rr * vrow[i]);
      if (tid == 0) addE(5, im * rr * rr);
    }
  }
  if (!lead) return;
  if (tid < 42) {
    double acc = 0.0;
    if (tid < 36) {
      const int a = tid / 6, c = tid % 6;
      for (int j = 0; j < m; ++j) {
        const double* o = px + 21 * j;
        acc += o[20] * (o[6 + a] * o[6 + c] + o[12 + a] * o[12 + c]);
      }
      if (!FX || a >= c) addH(pi[a], pi[c], i_px * acc);
    } else {
      const int a = tid - 36;
      for (int j = 0; j < m; ++j) {
        const double* o = px + 21 * j;
        acc += o[20] * (o[6 + a] * o[18] + o[12 + a] * o[19]);
      }
      addG(pi[a], -i_px * acc);
    }
  } else if (tid >= 64 && tid < 128) {
    const int j = tid - 64;
    double e = 0.0;
    if (j < m) { const double* o = px + 21 * j; e = o[20] * (o[18] * o[18] + o[19] * o[19]); }
    for (int jj = j + 64; jj < m; jj += 64) { const double* o = px + 21 * jj; e += o[20] * (o[18] * o[18] + o[19] * o[19]); }
    e = wave_sum(e);
    if (j == 0) addE(2, i_px * e);
  }
  // anchors on keyframe 0 (Mapping.py:855-900)
  if (b == 0 && tid == 0) {
    // pose prior: xi = -Log(T0^-1 anchor) with the reference's SE3_logmap (lie_algebra.py:127-176)
    double Ti[12];
    invert_pose34(A.poses, Ti);
    double T[12];
    const double* An = A.pose_anchor;
    for (int i = 0; i < 3; ++i)
      for (int j2 = 0; j2 < 4; ++j2)
        T[i * 4 + j2] = Ti[i * 4] * An[j2] + Ti[i * 4 + 1] * An[4 + j2] + Ti[i * 4 + 2] * An[8 + j2] + (j2 == 3 ? Ti[i * 4 + 3] : 0.0);
    const double tr = T[0] + T[5] + T[10];
    const double tr3 = tr - 3.0;
    const double theta = acos(0.5 * (tr - 1.0));
    const double mag = (tr3 < -1e-6) ? theta / (2.0 * sin(theta)) : 0.5 - tr3 / 12.0 + tr3 * tr3 / 60.0;
    const double wv[3] = {mag * (T[9] - T[6]), mag * (T[2] - T[8]), mag * (T[4] - T[1])};
    double th = sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]);
    th = fmax(th, 1e-6);
    const double wn[3] = {wv[0] / th, wv[1] / th, wv[2] / th};
    const double tt[3] = {T[3], T[7], T[11]};
    const double wxt[3] = {wn[1] * tt[2] - wn[2] * tt[1], wn[2] * tt[0] - wn[0] * tt[2], wn[0] * tt[1] - wn[1] * tt[0]};
    const double wxwxt[3] = {wn[1] * wxt[2] - wn[2] * wxt[1], wn[2] * wxt[0] - wn[0] * wxt[2], wn[0] * wxt[1] - wn[1] * wxt[0]};
    const double cf = 1.0 - th / (2.0 * tan(0.5 * th));
    for (int i = 0; i < 3; ++i) { red[i] = -wv[i]; red[3 + i] = -(tt[i] - (0.5 * tt[i]) * wxt[i] + cf * wxwxt[i]); }
  }
  __syncthreads();
  if (b == 0 && tid < 8) {                           // 6 pose + 2 affine anchor entries: one lane each
    const bool isp = tid < 6;
    const int q = isp ? 0 : tid - 6;
    const double isq = 1.0 / A.s_pose;
    const double jtj = (double)((float)isq * (float)isq);          // float32 J^T J of the reference (torch.eye default dtype)
    const double ia = (1.0 / A.s_aff) * (1.0 / A.s_aff);
    const double xi = red[isp ? tid : 0];
    const double r = A.aff[q] - A.aff_anchor[q];
    addH(pi[tid], pi[tid], isp ? jtj : ia);
    addG(pi[tid], isp ? -(isq * (isq * xi)) : -ia * r);
    double ep = isp ? (isq * xi) * (isq * xi) : 0.0;
    double ea = isp ? 0.0 : ia * r * r;
    for (int o = 1; o < 8; o <<= 1) { ep += __shfl_xor(ep, o, 8); ea += __shfl_xor(ea, o, 8); }
    if (tid == 0) { addE(3, ep); addE(4, ea); }
  }
  if (b == 0 && A.nfix > 0) {
    const double il = (1.0 / A.s_lm) * (1.0 / A.s_lm);
    double el = 0;
    for (int e = tid; e < 3 * A.nfix; e += 256) {
      const int f = e / 3, d = e % 3;
      const double r = A.P_m[3 * (long)A.fix_lm[f] + d] - A.P_anchor[e];
      const long ii = A.fix_inds[e];
      addH(ii, ii, il);
      addG(ii, -il * r);
      el += il * r * r;
    }
    red[tid & 63] = 0;
    el = wave_sum(el);
    if ((tid & 63) == 0) addE(5, el);
  }
}

__global__ __launch_bounds__(256) void win_update_kernel(const double* __restrict__ delta, double* __restrict__ poses,
                                                         double* __restrict__ aff, const long* __restrict__ frame_inds,
                                                         int F, double* __restrict__ P_m, int L, long lm_start,
                                                         const int* __restrict__ info) {
  // a solve that did not complete (info != 0: non-positive pivot, or -1 = the persistent solver timed out and left `delta`
  // unwritten) must not move the state: the update is skipped and the caller reads `info` at its next synchronisation
  if (info && *info != 0) return;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid < F) {
    const long* ix = frame_inds + 8 * (long)tid;
    double xi[6], E[16], Tn[16];
    for (int a = 0; a < 6; ++a) xi[a] = delta[ix[a]];
    se3_exp_f64(xi, E);
    double* T = poses + 16 * (long)tid;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double t = 0;
        for (int k = 0; k < 4; ++k) t += T[i * 4 + k] * E[k * 4 + j];
        Tn[i * 4 + j] = t;
      }
    for (int e = 0; e < 16; ++e) T[e] = Tn[e];
    aff[2 * tid] += delta[ix[6]];
    aff[2 * tid + 1] += delta[ix[7]];
  }
  for (int e = tid; e < 3 * L; e += gridDim.x * 256) P_m[e] += delta[lm_start + e];
}

// invertSE3 (lie_algebra.py:83-95 without the Jacobian; also the inverse inside get_T_w_curr / get_rel_pose, transforms.py:6-13):
// Ti = [R^T | -(R^T t); 0 0 0 1] for n poses, one thread each -- the mirror's torch form is six launches (zeros_like, three slice
// assignments, a batched matmul, a negation: ~90 us of host time) and runs several times per frame in the tracker / mapper glue.
// The products are the plain sums a (3x3)(3x1) matmul forms, left to right, without contraction.
template <typename T>
__global__ __launch_bounds__(64) void se3_inverse_kernel(const T* __restrict__ in, T* __restrict__ out, int n) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const T* a = in + 16 * (long)i;
  T* o = out + 16 * (long)i;
  T R[9], t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R[3 * r + c] = a[4 * r + c];
    t[r] = a[4 * r + 3];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    T s = R[r] * t[0];                 // row r of R^T = column r of R
    s = s + R[3 + r] * t[1];
    s = s + R[6 + r] * t[2];
    o[4 * r + 0] = R[r]; o[4 * r + 1] = R[3 + r]; o[4 * r + 2] = R[6 + r]; o[4 * r + 3] = -s;
  }
  o[12] = T(0); o[13] = T(0); o[14] = T(0); o[15] = T(1);
}

// out_i = op(A_i) op(B_i) for n pose pairs, one thread each; mode 1: inv(A) B (relative pose, transforms.py:11-13), mode 2: A inv(B)
// (world pose of the current frame, transforms.py:6-8), mode 0: A B.  na / nb = 1 broadcasts that operand.  The inverse is formed
// exactly as se3_inverse_kernel does, the 4x4 product as plain left-to-right sums -- one launch instead of the inverse kernel + a
// library GEMM (4x4 products were the last library calls of a plain frame of the sequential loop).
template <typename T>
__global__ __launch_bounds__(64) void se3_compose_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ out, int n,
                                                         int na, int nb, int mode) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  T a[16], b[16];
  auto load = [](const T* p, T* m, bool inv) {
    if (!inv) {
#pragma unroll
      for (int e = 0; e < 16; ++e) m[e] = p[e];
      return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      T s = p[r] * p[3];
      s = s + p[4 + r] * p[7];
      s = s + p[8 + r] * p[11];
      m[4 * r + 0] = p[r]; m[4 * r + 1] = p[4 + r]; m[4 * r + 2] = p[8 + r]; m[4 * r + 3] = -s;
    }
    m[12] = T(0); m[13] = T(0); m[14] = T(0); m[15] = T(1);
  };
  load(A + 16 * (long)(na == 1 ? 0 : i), a, mode == 1);
  load(B + 16 * (long)(nb == 1 ? 0 : i), b, mode == 2);
  T* o = out + 16 * (long)i;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      T s = a[4 * r] * b[c];
      s = s + a[4 * r + 1] * b[4 + c];
      s = s + a[4 * r + 2] * b[8 + c];
      s = s + a[4 * r + 3] * b[12 + c];
      o[4 * r + c] = s;
    }
  }
}

// normalizeSE3_inplace (como/geometry/lie_algebra.py:98-101: R <- U V^T of R's SVD), for rotation blocks that are rotations up to
// rounding (a float32 tracked pose composed in float64): U V^T is the orthogonal polar factor of R, which Newton's iteration
// X <- (X + X^-T) / 2 reaches quadratically from such a start (error e -> e^2 / 2: three steps from 1e-7 are below 1e-16; eight are
// run).  The reference's device SVD is a dozen solver launches with their own synchronisations (~1 ms), the host LAPACK form a
// device -> host -> device round trip in the middle of a keyframe insertion; this is one thread per pose.  Differs from an SVD's
// U V^T by rounding only (|dR| ~ 2e-16).
template <typename T>
__global__ void se3_normalize_kernel(T* __restrict__ poses, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  T* P = poses + 16 * (long)i;
  double X[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) X[3 * r + c] = (double)P[4 * r + c];
  for (int it = 0; it < 8; ++it) {
    double C[9];                                             // cofactors: X^-T = C / det
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    const double inv = 1.0 / det;
    for (int e = 0; e < 9; ++e) X[e] = 0.5 * (X[e] + C[e] * inv);
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) P[4 * r + c] = (T)X[3 * r + c];
}

// Everything the host reads and keeps of a tracked frame (Tracking.handle_frame, Tracking.py:315-379) in ONE record:
// [|t| of T_curr_kf | median depth | pixels seen | status word of every pyramid level | T_curr_kf (16) | aff_curr_kf (2) | T_w_curr (16)],
// T_w_curr = T_w_kf inv(T_curr_kf) with se3_compose_kernel mode 2's arithmetic.  One thread: it replaces the pose composition, a
// clone, a norm reduction, two casts and a concatenation -- five dependent launches of >= 4.5 us each behind every tracked frame.
__global__ void track_frame_record_kernel(const float* __restrict__ T, const float* __restrict__ aff, const float* __restrict__ T_w_kf,
                                          const float* __restrict__ median, const int* __restrict__ nseen,
                                          const float* __restrict__ recs, int levels, int stride, float* __restrict__ out) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float a[16], p[16], b[16];
  for (int e = 0; e < 16; ++e) { a[e] = T_w_kf[e]; p[e] = T[e]; }
  for (int r = 0; r < 3; ++r) {
    float s = p[r] * p[3];
    s = s + p[4 + r] * p[7];
    s = s + p[8 + r] * p[11];
    b[4 * r + 0] = p[r]; b[4 * r + 1] = p[4 + r]; b[4 * r + 2] = p[8 + r]; b[4 * r + 3] = -s;
  }
  b[12] = 0.f; b[13] = 0.f; b[14] = 0.f; b[15] = 1.f;
  out[0] = sqrtf((p[3] * p[3] + p[7] * p[7]) + p[11] * p[11]);
  out[1] = median[0];
  out[2] = (float)nseen[0];
  for (int l = 0; l < levels; ++l) out[3 + l] = recs[(long)l * stride + 104];
  float* o = out + 3 + levels;
  for (int e = 0; e < 16; ++e) o[e] = p[e];
  o[16] = aff[0]; o[17] = aff[1];
  o += 18;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = a[4 * r] * b[c];
      s = s + a[4 * r + 1] * b[4 + c];
      s = s + a[4 * r + 2] * b[8 + c];
      s = s + a[4 * r + 3] * b[12 + c];
      o[4 * r + c] = s;
    }
}

// A tracked frame's state in the world frame, as the mapper needs it when the tracker hands a frame over (Mapping.handle_tracking_data,
// Mapping.py:580-598): T_w_curr = T_w_kf inv(T_curr_kf) (get_T_w_curr, transforms.py:6-8: se3_compose_kernel mode 2's arithmetic) and
// aff_w_curr = (a_kf + a_cur, b_kf + b_cur exp(a_cur)) (get_aff_w_curr, affine_brightness.py:5-10: product and sums rounded on
// their own, the device library's exp).  The tracker's float32 values are widened first, as `.to(float64)` does.  One thread,
// one launch, for the ~10 launches of the torch forms.
template <typename TIN>
__global__ void frame_world_kernel(const double* __restrict__ T_w_kf, const TIN* __restrict__ T_curr_kf, const double* __restrict__ aff_w_kf,
                                   const TIN* __restrict__ aff_curr_kf, double* __restrict__ T_out, double* __restrict__ aff_out) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[16], p[16], b[16];
  for (int e = 0; e < 16; ++e) { a[e] = T_w_kf[e]; p[e] = (double)T_curr_kf[e]; }
  for (int r = 0; r < 3; ++r) {
    double s = p[r] * p[3];
    s = s + p[4 + r] * p[7];
    s = s + p[8 + r] * p[11];
    b[4 * r + 0] = p[r]; b[4 * r + 1] = p[4 + r]; b[4 * r + 2] = p[8 + r]; b[4 * r + 3] = -s;
  }
  b[12] = 0.0; b[13] = 0.0; b[14] = 0.0; b[15] = 1.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = a[4 * r] * b[c];
      s = s + a[4 * r + 1] * b[4 + c];
      s = s + a[4 * r + 2] * b[8 + c];
      s = s + a[4 * r + 3] * b[12 + c];
      T_out[4 * r + c] = s;
    }
  const double a_cur = (double)aff_curr_kf[0], b_cur = (double)aff_curr_kf[1];
  aff_out[0] = aff_w_kf[0] + a_cur;
  const double t = b_cur * exp(a_cur);
  aff_out[1] = aff_w_kf[1] + t;
}

}  // namespace como

extern "C" {

int como_se3_normalize_f64(double* poses, int n, como_stream_t stream) {
  if (!poses || n <= 0) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::se3_normalize_kernel<double>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, poses, n);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
int como_se3_normalize_f32(float* poses, int n, como_stream_t stream) {
  if (!poses || n <= 0) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::se3_normalize_kernel<float>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, poses, n);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_track_frame_record_f32(const float* T_curr_kf, const float* aff_curr_kf, const float* T_w_kf, const float* median, const int* nseen,
                                const float* level_records, int levels, int record_stride, float* out, como_stream_t stream) {
  if (!T_curr_kf || !aff_curr_kf || !T_w_kf || !median || !nseen || !level_records || levels < 1 || levels > 16 || record_stride < 105 ||
      !out)
    return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::track_frame_record_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, T_curr_kf, aff_curr_kf, T_w_kf, median, nseen,
                     level_records, levels, record_stride, out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_frame_world_f64(const double* T_w_kf, const void* T_curr_kf, const double* aff_w_kf, const void* aff_curr_kf, int cur_is_f32,
                         double* T_out, double* aff_out, como_stream_t stream) {
  if (!T_w_kf || !T_curr_kf || !aff_w_kf || !aff_curr_kf || !T_out || !aff_out) return COMO_ERR_ARG;
  if (cur_is_f32)
    hipLaunchKernelGGL(como::frame_world_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, T_w_kf, (const float*)T_curr_kf, aff_w_kf,
                       (const float*)aff_curr_kf, T_out, aff_out);
  else
    hipLaunchKernelGGL(como::frame_world_kernel<double>, dim3(1), dim3(64), 0, (hipStream_t)stream, T_w_kf, (const double*)T_curr_kf, aff_w_kf,
                       (const double*)aff_curr_kf, T_out, aff_out);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_win_scaffold(const como_win_args* a, como_stream_t stream) {
  using namespace como;
  if (!a || a->B <= 0 || a->m <= 0 || a->m > 64 || a->F < a->B) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  ScaffoldOut o{a->pm, a->logzm, a->invz, a->dzdP, a->dlogz_dT, a->dlogz_dP, a->dp_dP, a->dp_dT, a->init_Pm, a->reinit_flag};
  int grid = a->B > (a->F * 16 + 63) / 64 ? a->B : (a->F * 16 + 63) / 64;
  const long za16 = a->zero_a ? a->zero_a_bytes / 16 : 0, zb16 = a->zero_b ? a->zero_b_bytes / 16 : 0;
  const long zc16 = a->zero_c ? a->zero_c_bytes / 16 : 0;
  if ((a->zero_a && (a->zero_a_bytes & 15)) || (a->zero_b && (a->zero_b_bytes & 15)) || (a->zero_c && (a->zero_c_bytes & 15)))
    return COMO_ERR_ARG;
  if (za16 + zb16 + zc16 > 0 && grid < 64) grid = 64;      // enough threads for the clears
  if (zc16 > 0) {                                          // megabytes (the fixed-point system): ~4 stores per thread
    const long want = (zc16 + 64 * 4 - 1) / (64 * 4);
    if (want > grid) grid = (int)(want > 8192 ? 8192 : want);
  }
  double* zerr = (a->zero_a || a->zero_b) ? a->err : nullptr;
  if (a->pix_is_f64) {
    ScaffoldPix<double> px{(double*)a->px_logzm, (double*)a->px_invz, (double*)a->px_dzdP, (double*)a->px_dlogz_dT,
                           (double*)a->px_poses, (double*)a->px_aff};
    hipLaunchKernelGGL(win_scaffold_kernel<double>, dim3(grid), dim3(64), 0, s, a->poses, a->aff, a->F, a->P_m, a->lm_ids,
                       a->first_frame, a->first_slot, a->K, a->median, a->pm_first, a->B, a->m, o, px, (uint4*)a->zero_a, za16,
                       (uint4*)a->zero_b, zb16, zerr, (uint4*)a->zero_c, zc16);
  } else {
    ScaffoldPix<float> px{(float*)a->px_logzm, (float*)a->px_invz, (float*)a->px_dzdP, (float*)a->px_dlogz_dT,
                          (float*)a->px_poses, (float*)a->px_aff};
    hipLaunchKernelGGL(win_scaffold_kernel<float>, dim3(grid), dim3(64), 0, s, a->poses, a->aff, a->F, a->P_m, a->lm_ids,
                       a->first_frame, a->first_slot, a->K, a->median, a->pm_first, a->B, a->m, o, px, (uint4*)a->zero_a, za16,
                       (uint4*)a->zero_b, zb16, zerr, (uint4*)a->zero_c, zc16);
  }
  COMO_CHECK_LAUNCH();
  hipLaunchKernelGGL(win_apply_reinit_kernel, dim3((a->L + 255) / 256), dim3(256), 0, s, a->P_m, a->init_Pm, a->reinit_flag, a->L);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

/* The log-depths the NEXT iteration's scaffold will compute, ahead of it: the SAME kernel (bit-identical arithmetic) over the state as it
 * stands now, with every output redirected to `scratch` except the pix-dtype log-depths (-> px_logzm_out); nothing of the window is
 * written, no landmark is re-initialised.  `zero` (optional, a multiple of 16 bytes) is cleared in the same launch (the radix-select
 * histograms of the median pass that follows).  Between the end of an iteration and the next one's scaffold nothing changes the
 * keyframes' poses / landmarks / medians in the sequential loop's plain and one-way frames (Mapping.py:760-807 rebuilds its tensors from the
 * same values), so the full-image median of the next iteration (Mapping.store_vars, Mapping.py:749-758) can be streamed while the
 * tracker runs instead of beside the block kernel. */
long como_win_logz_ahead_scratch_bytes(int B, int m, int L, int F) {
  const long bm = (long)B * m;
  return 8 * (31 * bm + 3L * B + 3L * L + 8) + 4 * ((long)L + 4) + 8 * (7 * bm + 3L * B + 18L * F + 8);
}

int como_win_logz_ahead(const como_win_args* a, void* scratch, long scratch_bytes, void* px_logzm_out, void* zero, long zero_bytes,
                        como_stream_t stream) {
  using namespace como;
  if (!a || !scratch || !px_logzm_out || a->B <= 0 || a->m <= 0 || a->m > 64 || a->F < a->B || (zero && (zero_bytes & 15)) ||
      scratch_bytes < como_win_logz_ahead_scratch_bytes(a->B, a->m, a->L, a->F) || ((uintptr_t)scratch & 15))
    return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long bm = (long)a->B * a->m;
  double* d = (double*)scratch;
  ScaffoldOut o;
  o.pm = d; d += 2 * bm;
  o.logzm = d; d += bm;
  o.invz = d; d += bm;
  o.dzdP = d; d += 3L * a->B;
  o.dlogz_dT = d; d += 6 * bm;
  o.dlogz_dP = d; d += 3 * bm;
  o.dp_dP = d; d += 6 * bm;
  o.dp_dT = d; d += 12 * bm;
  o.init_Pm = d; d += 3L * a->L;
  d += ((uintptr_t)d & 8) ? 1 : 0;
  o.reinit_flag = (int*)d;
  d += ((long)a->L + 3) / 2 + 1;
  int grid = a->B > (a->F * 16 + 63) / 64 ? a->B : (a->F * 16 + 63) / 64;
  const long za16 = zero ? zero_bytes / 16 : 0;
  if (za16 > 0 && grid < 64) grid = 64;
  if (a->pix_is_f64) {
    double* q = d;
    ScaffoldPix<double> px;
    px.logzm = (double*)px_logzm_out;
    px.invz = q; q += bm;
    px.dzdP = q; q += 3L * a->B;
    px.dlogz_dT = q; q += 6 * bm;
    px.poses = q; q += 16L * a->F;
    px.aff = q;
    hipLaunchKernelGGL(win_scaffold_kernel<double>, dim3(grid), dim3(64), 0, s, a->poses, a->aff, a->F, a->P_m, a->lm_ids,
                       a->first_frame, a->first_slot, a->K, a->median, a->pm_first, a->B, a->m, o, px, (uint4*)zero, za16,
                       (uint4*)nullptr, 0L, (double*)nullptr, (uint4*)nullptr, 0L);
  } else {
    float* q = (float*)d;
    ScaffoldPix<float> px;
    px.logzm = (float*)px_logzm_out;
    px.invz = q; q += bm;
    px.dzdP = q; q += 3L * a->B;
    px.dlogz_dT = q; q += 6 * bm;
    px.poses = q; q += 16L * a->F;
    px.aff = q;
    hipLaunchKernelGGL(win_scaffold_kernel<float>, dim3(grid), dim3(64), 0, s, a->poses, a->aff, a->F, a->P_m, a->lm_ids,
                       a->first_frame, a->first_slot, a->K, a->median, a->pm_first, a->B, a->m, o, px, (uint4*)zero, za16,
                       (uint4*)nullptr, 0L, (double*)nullptr, (uint4*)nullptr, 0L);
  }
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_win_priors(const como_win_args* a, como_stream_t stream) {
  using namespace como;
  if (!a || a->B <= 0 || a->m <= 0 || a->m > 64 || (!a->sysfix && (!a->H || !a->g || !a->err))) return COMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  PriorArgs A;
  A.logzm = a->logzm; A.dlogz_dT = a->dlogz_dT; A.dlogz_dP = a->dlogz_dP; A.pm = a->pm; A.pm_first = a->pm_first;
  A.dp_dP = a->dp_dP; A.dp_dT = a->dp_dT; A.first_mask = a->first_mask; A.Kmm_inv = a->Kmm_inv;
  A.median = a->median_new; A.median_is_f32 = a->median_new_is_f32; A.median_stride = a->median_new_stride;
  A.pose_inds = a->pose_inds; A.landmark_inds = a->landmark_inds; A.poses = a->poses; A.aff = a->aff;
  A.pose_anchor = a->pose_anchor; A.aff_anchor = a->aff_anchor; A.P_m = a->P_m; A.P_anchor = a->P_anchor;
  A.fix_inds = a->fix_inds; A.fix_lm = a->fix_lm; A.nfix = a->nfix;
  A.s_gp = a->s_gp; A.s_ld = a->s_ld; A.s_px = a->s_px; A.s_pose = a->s_pose; A.s_aff = a->s_aff; A.s_lm = a->s_lm;
  A.H = a->H; A.g = a->g; A.D = a->D; A.err = a->err;
  A.median_out = a->median_out;
  A.fix = (long long*)a->sysfix; A.plane = a->fix_plane;
  A.mld_J = a->mld_J; A.mld_anchor = a->mld_anchor; A.s_mld = a->s_mld;
  if (A.mld_J && (!A.mld_anchor || !(A.s_mld > 0.0) || a->nfix != 0)) return COMO_ERR_ARG;
  if (A.fix && A.plane < a->D * a->D + a->D + FIX_ERR_SLOTS) return COMO_ERR_ARG;
  const int m = a->m;
  const size_t lds = (size_t)(m * (m + 1) + m * 6 + m + m + m * 6 + m * 3 + 64 + m * 21 + 6 + 3 * m) * sizeof(double);
  hipLaunchKernelGGL(win_priors_kernel, dim3(a->B, 32), dim3(256), lds, s, A, a->B, m);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_win_update_checked(const double* delta, double* poses, double* aff, const long* frame_inds, int F, double* P_m, int L,
                            long lm_start, const int* info, como_stream_t stream) {
  if (!delta || !poses || !aff || !frame_inds || !P_m || F <= 0 || L <= 0) return COMO_ERR_ARG;
  int blocks = (3 * L + 255) / 256;
  if (blocks < (F + 255) / 256) blocks = (F + 255) / 256;
  hipLaunchKernelGGL(como::win_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, delta, poses, aff, frame_inds, F,
                     P_m, L, lm_start, info);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

int como_win_update(const double* delta, double* poses, double* aff, const long* frame_inds, int F, double* P_m, int L,
                    long lm_start, como_stream_t stream) {
  return como_win_update_checked(delta, poses, aff, frame_inds, F, P_m, L, lm_start, nullptr, stream);
}

int como_se3_inverse_f32(const float* T, float* out, int n, como_stream_t stream) {
  if (!T || !out || n <= 0) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::se3_inverse_kernel<float>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, T, out, n);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}
#define COMO_DEF_SE3_COMPOSE(SFX, T)                                                                                            \
  int como_se3_compose_##SFX(const T* A, const T* B, T* out, int n, int na, int nb, int mode, como_stream_t stream) {             \
    if (!A || !B || !out || n <= 0 || (na != 1 && na != n) || (nb != 1 && nb != n) || mode < 0 || mode > 2) return COMO_ERR_ARG;  \
    hipLaunchKernelGGL(como::se3_compose_kernel<T>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, A, B, out, n, na, nb,    \
                       mode);                                                                                                   \
    COMO_CHECK_LAUNCH();                                                                                                        \
    return COMO_OK;                                                                                                             \
  }
COMO_DEF_SE3_COMPOSE(f32, float)
COMO_DEF_SE3_COMPOSE(f64, double)

int como_se3_inverse_f64(const double* T, double* out, int n, como_stream_t stream) {
  if (!T || !out || n <= 0) return COMO_ERR_ARG;
  hipLaunchKernelGGL(como::se3_inverse_kernel<double>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, T, out, n);
  COMO_CHECK_LAUNCH();
  return COMO_OK;
}

}  // extern "C"
