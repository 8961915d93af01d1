/* como_hip.h -- C ABI of libcomo_hip.so: the MI355X (gfx950) hot path of COMO's dense photometric
 * Gauss-Newton backend and DepthCov inference path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - tensors are dense row-major in the layouts the reference's Python passes around
 *     (cited per function as reference file:line under /root/reference);
 *   - no allocation, no exceptions, no host synchronisation inside: the caller owns every buffer
 *     (outputs and workspaces) and the `stream`; calls only enqueue work on `stream`;
 *   - return value: 0 = ok, 1 = bad argument, 2 = launch failure.
 *   - como_stream_t is a hipStream_t (opaque pointer; NULL = default stream).
 *
 * Reference interfaces replaced
 *   como_backends.cross_covariance      como/backend/include/cov.h:10, src/cov.cpp:5-32, src/cov_gpu.cu:17-84
 *   como_backends.get_new_chol_obs_info como/backend/include/cov.h:18-20, src/cov.cpp:34-65, src/cov_gpu.cu:132-215
 *   (the functions below marked "python path" replace pure-PyTorch code of the reference, so their
 *    "FFI" is the Python operator boundary listed in INTEGRATION.md)
 */
#ifndef COMO_HIP_H
#define COMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* como_stream_t;

int como_abi_version(void);
/* Read and clear the HIP runtime's per-thread last error (returns its code, 0 = none).  Every como_* call checks the last error
 * after its launches, so a caller whose own runtime call failed (e.g. an aborted stream capture) clears it before going on. */
int como_clear_last_error(void);
/* End a stream capture that was invalidated (the stream stays in capture mode, and every launch on it fails, until
 * hipStreamEndCapture is called); destroys the partial graph and clears the last error.  Returns 1 if `stream` was capturing. */
int como_abort_capture(como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Exact k-th select (lower median of |r| over valid entries; torch.median semantics).
 * python path: photo.py:124-128, photo_tracking.py:131-134, two_frame_sfm.py:258-261.
 * `hists` = workspace of como_select_workspace_bytes() bytes.  Protocol:
 *   como_select_begin; como_select_hist_*(pass = 0..P-1) [P = 3 for f32, 6 for f64; multi-GPU: sum the
 *   ranks' hists of pass p with an all-reduce before pass p+1]; como_select_finish_* -> out3 =
 *   {median, 1.4826*median, nvalid}.
 * Segments: `nseg` independent selects over consecutive length-n slices of r / valid (one median per keyframe,
 * sparse_map.py:220); hists holds nseg * como_select_workspace_bytes() bytes, out3 is (nseg,3).  valid may be NULL
 * (all entries valid). */
int como_select_workspace_bytes(void);
int como_select_begin(void* hists, int nseg, como_stream_t stream);
int como_select_hist_f32(const float* r, const uint8_t* valid, long n, int nseg, void* hists, int pass, como_stream_t stream);
int como_select_hist_f64(const double* r, const uint8_t* valid, long n, int nseg, void* hists, int pass, como_stream_t stream);
int como_select_finish_f32(const void* hists, int nseg, float* out3, como_stream_t stream);
int como_select_finish_f64(const void* hists, int nseg, double* out3, como_stream_t stream);
/* Multi-GPU double select with ONE exchange for digits 3..5 (the reference has no distributed code; this serves the same
 * global median of photo.py:124-128 across ranks).  After digits 0..2 are all-reduced, run como_select_hist_f64 with
 * pass = 3 | 0x100 | 0x200 (collect the keys matching the 33-bit prefix, no tail) on every rank, then
 *   como_select_cand_pack : hists (nseg workspaces) -> out (nseg x como_select_cand_words() uint32: count | 0 | 512 keys), scratch
 *                           and the rank-local digit-3 histogram cleared;
 *   all-gather the records over the ranks -> gathered (world, nseg_total, words);
 *   como_select_cand_merge: writes the digit 3, 4, 5 histograms of the UNION into hists (segments seg0 .. seg0 + nseg of the
 *                           gathered records) -- consumers and como_select_finish_f64 resolve as after six plain passes.
 * A rank with more than 512 candidates (> 512 keys sharing 33 leading bits: sigma = 0 inputs) makes the select report zero
 * valid keys (median NaN, poisoned system). */
int como_select_cand_words(void);
int como_select_cand_pack(void* hists, int nseg, void* out, como_stream_t stream);
int como_select_cand_merge(void* hists, int nseg, const void* gathered, int world, int nseg_total, int seg0, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Tracking: one inverse-compositional GN iteration (python path: frontend/photo_tracking.py:117-143).
 *   Tji (4,4)  K (3,3)  aff (2)  P (N,3)  vals_i (N)  img (H,W)  J8 (N,8) [column 6 is overwritten with
 *   -e^{-a} I_j as the reference does in place]  r_ws (N) workspace  valid_out (N) u8
 *   pj_out (N,2) / depth_out (N) optional (NULL to skip)  hists: select workspace
 *   partials: como_track_partials_bytes() bytes
 *   out (105): [0:64) H | [64:72) g | [72:80) delta | [80:96) T_new | [96:98) aff_new | 98 mean_sq_err |
 *              99 grad_norm | 100 total_err | 101 sigma | 102 nvalid | 103 |delta| | 104 cholesky info */
long como_track_partials_bytes(void);
int como_track_iter_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                        const float* img, int H, int W, long N, float* J8, float* r_ws, uint8_t* valid_out,
                        float* pj_out, float* depth_out, void* hists, void* partials, float* out, como_stream_t stream);
int como_track_iter_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                        const double* img, int H, int W, long N, double* J8, double* r_ws, uint8_t* valid_out,
                        double* pj_out, double* depth_out, void* hists, void* partials, double* out,
                        como_stream_t stream);
/* Same iteration over a FIXED-size reference set with a selection mask instead of a gathered subset (the reference gathers
 * vals / P / dI_dT with the keyframe's validity mask on every frame, photo_tracking.py:20-26): in_mask (N) u8, 0 = the point
 * is ignored exactly like one that projects outside the image (it may carry non-finite P / J8).  Buffer sizes never change
 * between keyframes, so one captured hipGraph per pyramid level serves the whole run. */
int como_track_iter_masked_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                               const float* img, int H, int W, long N, float* J8, float* r_ws, uint8_t* valid_out,
                               float* pj_out, float* depth_out, void* hists, void* partials, float* out,
                               const uint8_t* in_mask, como_stream_t stream);
int como_track_iter_masked_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                               const double* img, int H, int W, long N, double* J8, double* r_ws, uint8_t* valid_out,
                               double* pj_out, double* depth_out, void* hists, void* partials, double* out,
                               const uint8_t* in_mask, como_stream_t stream);

/* One whole pyramid LEVEL -- photo_level_tracking, como/odom/frontend/photo_tracking.py:147-185 -- in one launch: the
 * Gauss-Newton loop (tracking_iter :117-143 per iteration) with the stop test (:166-180: iter >= max_iter or |delta| <
 * delta_norm or |(mse_prev - mse) / mse_prev| < rel_tol or |g| < grad_norm, float32 arithmetic) evaluated on the device.
 * A persistent kernel: reference pixels stay in registers across iterations, device-wide barriers replace the kernel
 * boundaries of the como_track_iter_* chain, no host round trip.  float32 (the reference's tracking dtype); gray (colour: como_track_level_channels_f32).
 *   J8 (N,8) is read only (column 6 is recomputed every iteration, not written back); in_mask (N) u8 or NULL.
 *   workspace: como_track_level_workspace_bytes() bytes (cleared by the call); workspace_uncached = 1 if it came from
 *   como_track_level_workspace_create() (uncached device memory: the device-wide barriers then need no L2 invalidate,
 *   40 instead of 50 us per iteration at 640x480), 0 for ordinary device memory.
 *   out (106): the como_track_iter_* record of the LAST iteration run (T at [80:96), aff at [96:98)), [104] = cholesky
 *   info, -1 if a barrier timed out, -2 if an XCD-local level did not sit on one XCD; [105] = number of iterations run.
 * Returns COMO_ERR_ARG when N exceeds 5 x 256 x (number of compute units) pixels (use the chain then). */
long como_track_level_workspace_bytes(void);
/* allocator of such a workspace in UNCACHED device memory (hipDeviceMallocUncached; call outside stream capture); NULL if the
 * runtime refuses -- pass ordinary device memory and workspace_uncached = 0 then */
void* como_track_level_workspace_create(void);
void como_track_level_workspace_destroy(void* ws);
int como_track_level_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P, const float* vals_i,
                         const float* img, int H, int W, long N, const float* J8, const uint8_t* in_mask, int max_iter,
                         float delta_norm, float rel_tol, float grad_norm, void* workspace, int workspace_uncached, float* out,
                         como_stream_t stream);

/* Per-frame glue of the tracker, fused (csrc/trackref.hip).
 * como_track_reference_*: a keyframe's reference arrays of ONE pyramid level from its depth map (Tracking.py:255-300
 *   update_kf_reference): depth (b,h*w), rel (b,4,4) = T_lastkf^-1 T_kf, K (3,3) of the level, dI_dw (b,n,1,2) image gradients,
 *   vals (b,n,1) intensities (n = h*w, gray) -> P_out (b,n,3) points in the newest keyframe's frame, mask_out (b,n) u8 =
 *   projects into the image grown by `border` with depth > depth_thresh (closed interval, :265-281), J_out (b,n,1,8)
 *   inverse-compositional Jacobians (photo_tracking.py:46-74).
 * como_reproject_depth_*: the newest keyframe's finest-level points P (n,3) seen from the current frame Tck (4,4) = T_curr_kf as a
 *   depth image img (h*w) (NaN where nothing lands; several points on a pixel: the LAST one wins, utils/coords.py:50-56), seen
 *   (h*w) u8, *nseen = pixels hit (Tracking.py:163-185 get_reproj_last_kf + :341-344).  order_ws: h*w int64, ZERO on entry, left
 *   zero; zbuf: n elements of scratch. */
int como_track_reference_f32(const float* depth, const float* rel, const float* K, const float* dI_dw, const float* vals, int b,
                             int h, int w, float border, float depth_thresh, float* P_out, uint8_t* mask_out, float* J_out,
                             como_stream_t stream);
int como_track_reference_f64(const double* depth, const double* rel, const double* K, const double* dI_dw, const double* vals,
                             int b, int h, int w, double border, double depth_thresh, double* P_out, uint8_t* mask_out,
                             double* J_out, como_stream_t stream);
/* The tracker's whole reference pyramid in one launch (Tracking.update_kf_reference, como/odom/Tracking.py:187-313, with
 * pyr.depth_interp_mode nearest_neighbor, como/data/depth_resize.py:6-36): depth0 (nk,H0,W0) the finest depth images, kf_poses
 * (nk,4,4) world poses of the reference keyframes (the LAST one is the frame the points are expressed in: rel_b = T_last^-1 T_b);
 * levels <= 4 pyramid levels COARSE -> FINE, hw[2 l], hw[2 l + 1] their sizes (= the finest size pooled levels - 1 - l times:
 * checked), per level (host arrays of device pointers) K (3,3), dI_dw (nk,n,1,2), vals (nk,n,1) and the outputs P (nk,n,3),
 * mask (nk,n), J (nk,n,1,8) -- per pixel what como_track_reference_f32 writes. */
int como_track_reference_pyr_f32(const float* depth0, int H0, int W0, const float* kf_poses, int nk, int levels, const int* hw,
                                 const float* const* K, const float* const* dI_dw, const float* const* vals, float* const* P_out,
                                 uint8_t* const* mask_out, float* const* J_out, float border, float depth_thresh, como_stream_t stream);
int como_reproject_depth_f32(const float* Tck, const float* K, const float* P, long n, int h, int w, void* order_ws, float* zbuf,
                             float* img, uint8_t* seen, int* nseen, como_stream_t stream);
int como_reproject_depth_f64(const double* Tck, const double* K, const double* P, long n, int h, int w, void* order_ws,
                             double* zbuf, double* img, uint8_t* seen, int* nseen, como_stream_t stream);
/* como_reproject_points_*: n points of frame i -- row/col coordinates `coords` (n,2) (NULL: the pixel grid of width `wgrid`) and
 * depths z (n) -- seen from frame j (Tji (4,4), K (3,3)): rc_out (n,2) row/col there, P_out (n,3) camera points, keep (n) (may be
 * NULL) = at least one pixel inside the (h, w) image and deeper than min_depth.  Replaces como/odom/frontend/corr.py:17-43
 * (filter_reproj_coords + reproject_points: backprojection camera.py:43-54, transform_points transforms.py:17-23, projection
 * camera.py:20-26) of the new keyframe's correspondence search. */
int como_reproject_points_f32(const float* coords, const float* z, const float* Tji, const float* K, long n, int wgrid, int h, int w,
                              float min_depth, float* rc_out, float* P_out, uint8_t* keep, como_stream_t stream);
int como_reproject_points_f64(const double* coords, const double* z, const double* Tji, const double* K, long n, int wgrid, int h, int w,
                              double min_depth, double* rc_out, double* P_out, uint8_t* keep, como_stream_t stream);

/* Colour images (`color: rgb`, config/como.yml:7; photo_tracking.py works on (1,N,c) values and (1,N,c,8) Jacobians): the
 * same iteration / level with `channels` = c image channels.  vals_i (N,c), J8 (N,c,8), img (c,H,W) planes, r_ws (N,c),
 * valid_out (N,c) u8 workspace (every channel of a pixel carries the pixel's mask; the reference's (1,N) mask is column 0),
 * pj_out (N,2), depth_out (N), in_mask (N).  Every (pixel, channel) residual is one entry of the median and of the sums;
 * mean_sq_err divides by the number of valid PIXELS (photo_tracking.py:83-85); out[102] = valid pixels.  channels in 1..4;
 * como_track_level_channels_f32 returns COMO_ERR_ARG when N * channels exceeds the persistent kernel's capacity. */
int como_track_iter_channels_f32(const float* Tji, const float* K, const float* aff, const float* P, const float* vals_i,
                                 const float* img, int H, int W, long N, int channels, float* J8, float* r_ws,
                                 uint8_t* valid_out, float* pj_out, float* depth_out, void* hists, void* partials, float* out,
                                 const uint8_t* in_mask, como_stream_t stream);
int como_track_iter_channels_f64(const double* Tji, const double* K, const double* aff, const double* P, const double* vals_i,
                                 const double* img, int H, int W, long N, int channels, double* J8, double* r_ws,
                                 uint8_t* valid_out, double* pj_out, double* depth_out, void* hists, void* partials,
                                 double* out, const uint8_t* in_mask, como_stream_t stream);
int como_track_level_channels_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                                  const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                                  const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                                  void* workspace, int workspace_uncached, float* out, como_stream_t stream);
/* Round 5: the same with a second workspace in ORDINARY (L2-cacheable) device memory (may be NULL).  Levels of at most
 * 32 x 256 x 5 elements (160x120 and below) then run XCD-LOCAL: the grid is restricted to the workgroups of one XCD, whose
 * three barriers per iteration are read-modify-writes in the one L2 they share instead of round trips to the memory side.
 * como_track_level_probe: 1 when the dispatcher places workgroup b on XCD b % 8 on this device (probed once per process). */
int como_track_level_local_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                               const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                               const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                               void* workspace, int workspace_uncached, void* local_workspace, float* out, como_stream_t stream);
int como_track_level_probe(void);
/* Round 6: the XCD-local form verifies its own placement INSIDE the launch: every participating workgroup records the XCD it runs
 * on (HW_REG_XCC_ID) and out[104] = -2 is reported unless all of them share one (the result must then be discarded: the caller
 * tracks the frame again and calls como_track_level_set_local(0), after which every level runs in the device-wide form).
 * como_track_level_set_local returns the previous setting; como_track_level_local_state = 1 while coarse levels run XCD-local;
 * como_track_level_debug_mismatch(1) makes workgroup 1 report a neighbouring XCD (the test of the -2 path). */
/* como_track_level_local_f32 without the clear of the barrier workspace (the caller cleared the first
 * como_track_level_zero_bytes() bytes of BOTH workspaces on this stream since their last use). */
int como_track_level_prezeroed_f32(const float* Tji_init, const float* K, const float* aff_init, const float* P,
                                   const float* vals_i, const float* img, int H, int W, long N, int channels, const float* J8,
                                   const uint8_t* in_mask, int max_iter, float delta_norm, float rel_tol, float grad_norm,
                                   void* workspace, int workspace_uncached, void* local_workspace, float* out, como_stream_t stream);
long como_track_level_zero_bytes(void);
int como_track_level_set_local(int enable);
int como_track_level_local_state(void);
void como_track_level_debug_mismatch(int on);
/* Band-split sums of the level kernel (csrc/track.hip): with 22 of the 32 key bits of the median |r| known, the Huber class of all but
 * a handful of pixels is known for every robust scale the median can still take (photo_tracking.py:77-93: weight 1, or 1.345 sigma / |r|,
 * linear in sigma), so the last digit's histogram pass also accumulates the inlier / outlier sums and lists the few undecided pixels: an
 * iteration is two device-wide synchronisations instead of three.  OFF by default (measured slower than the exact form -- scale first,
 * sums afterwards -- on MI355X); como_track_level_set_split(1) / COMO_TRACK_SPLIT=1 selects it, the call returns the previous setting; como_track_level_debug_amb_cap(c) shrinks the list
 * (0 .. 128, negative = default) so that the overflow path -- exact form after the split pass -- runs on ordinary data. */
int como_track_level_set_split(int enable);
void como_track_level_debug_amb_cap(int cap);
/* A level of at most 4800 elements (N * channels; 80x60 gray, the coarsest level of a 640x480 frame) can run in ONE workgroup
 * (csrc/track.hip track_level_one_kernel: pixel waves with their elements in LDS planes + registers and one solver wave; the median's
 * digit histograms in LDS, the 46 sums through packed accumulators, a wave tree and LDS; every dependency point of
 * photo_level_tracking's loop, photo_tracking.py:147-185, is a workgroup barrier; the workspaces are not touched).
 * como_track_level_set_one(512 / 768 / 1024) selects that form and its workgroup size, 0 the multi-workgroup forms -- the DEFAULT:
 * on MI355X the one-workgroup form measured 16.6 - 17.7 us per iteration against 17.4 (one compute unit is instruction-bound).
 * COMO_TRACK_ONE sets the initial value; the call returns the previous setting. */
int como_track_level_set_one(int threads);

/* ------------------------------------------------------------------------------------------------
 * Window BA linearisation (python path: backend/photo.py:83-233 batch_photo_cost).
 * All scalar tensors have the element type of the entry point (f32 / f64) except H, g (see h_is_f64).
 * "slot" = row of the per-reference-keyframe arrays a pair reads (several pairs may share a slot). */
typedef struct como_ba_args {
  int b;                     /* pairs in this batch */
  int n;                     /* reference pixels per slot */
  int m;                     /* inducing points per keyframe: multiple of 4, <= 64 */
  int H, W;                  /* target image size */
  int zmode;                 /* 0: zjac = dPwn_dzm (slots,n,3,m) as photo.py:92 receives it
                                1: factored: zjac = K~ rows (slot, row, m); dPwn_dzm[n,:,k] = uvec[n,:] K~[pix(n),k] invz[k]
                                2: factored AND compact (the tuned kernels): as 1, but dPwn_dTwc holds only the six planes
                                   dlogz_n/dT_wc = K~[n,:] dlogz_m/dT_wc (como_dense_ref_* flag 16) and uvec is not read: the
                                   kernels rebuild dP_w/dT_wc = [-[u]x R, R] + u (x) dlogz_n/dT_wc with u = P_w - t_wc from the
                                   reference keyframe's pose poses_all[ref_pose[p]] (sparse_map.py:184-230) */
  int chunks;                /* pixel chunks per pair (grid.x of the block kernel); partial records = b*chunks */
  int phase;                 /* bit mask: 1 setup+residual(+hist pass 0), 2<<(p-1) hist pass p>=1, 64 blocks, 128 reduce+assemble,
                                256 = ws_hists is already zero (skip the clear), 512 = float64 hist pass 3 collects the candidate
                                keys of the multi-GPU exchange instead of finishing locally (como_select_cand_*),
                                1024 = the pair constants alone (setup without the residual pass: that pass then runs fused into
                                como_dense_ref_fused_*; continue with phase 0xFE) */
  int h_is_f64;              /* element type of Hmat / gvec: 1 = double, 0 = float */
  int variant;               /* zmode 2 only: 0 = software-pipelined block kernels (default; the two-pair kernels where
                                grp_pairs lists pairs), 1 = straightforward one, 2 = pipelined one-pair kernel only (float32),
                                3 = one wave per SIMD, 4 = float64 two-pair kernel without the software pipeline */
  int stagger;               /* pipelined kernel: start delay (x1024 cycles) of odd hardware wave slots, 0 = none */
  int pix_begin, pix_end;    /* reference-pixel range [begin,end) of every pair handled by this call (multi-GPU shard);
                                pix_end <= 0 means n.  ws_r / ws_valid / pj_out are then (b, end-begin). */
  int anorm_f32;             /* 1: the sampling normalisation 1/W, 1/H is rounded to float32 first (two_frame_sfm.py:187-190
                                builds A_norm from an integer tensor -> float32 even in a float64 run); 0: computed in T */
  const void* Pwn;           /* zmode 0: (slots,n,3) photo.py:86 ; zmode 1, 2: planes (slots,3,n) */
  const void* vals;          /* (slots,n,c)   photo.py:84 (c = channels) */
  const void* dPwn_dTwc;     /* zmode 0: (slots,n,3,6) photo.py:90 ; zmode 1: planes (slots,18,n) ; zmode 2: planes (slots,6,n) */
  const void* zjac;          /* see zmode */
  const void* uvec;          /* zmode 1: planes (slots,3,n) */
  const int* pixidx;         /* zmode 1, 2: (slots,n) row of K~ per reference pixel, NULL = identity */
  const void* invz;          /* zmode 1, 2: (slots,m) = dlogz_m/dz_m = 1/z_m */
  long kt_slot_stride;       /* zmode 1, 2: elements between slots of K~ */
  const void* poses_all;     /* (F,4,4) target poses T_wc */
  const void* aff_all;       /* (A,2) affine brightness params */
  const void* img_base;      /* base pointer of the [I,gx,gy] (3c,H,W) stacks */
  const void* K;             /* (3,3) intrinsics */
  const int* ref_slot;       /* [b] */
  const int* ref_aff;        /* [b] index into aff_all */
  const int* tgt_aff;        /* [b] index into aff_all */
  const int* tgt_pose;       /* [b] index into poses_all */
  const long* tgt_img;       /* [b] element offset of the target stack from img_base */
  const long* pose_ref_inds; /* (b,8) rows of H for the reference pose+affine   photo.py:93 */
  const long* pose_tgt_inds; /* (b,8)                                            photo.py:94 */
  const long* landmark_inds; /* (b,3m)                                           photo.py:95 */
  const void* dzdP;          /* (slots,3) = dzm_dPwm[:,0,0,:]                    photo.py:92,169-182 */
  void* Hmat;                /* (D,D) accumulated in place, both triangles */
  void* gvec;                /* (D) */
  long D;
  double* err_out;           /* scalar, accumulated (+=) */
  void* sigma_out;           /* optional (2): {sigma_r, nvalid} in the entry point's element type */
  void* pj_out;              /* optional (b,n,2) projected pixel coordinates */
  double* pair_blocks_out;   /* optional (b, 3936): reduced raw per-pair records (tests) */
  void* ws_r;                /* (b,n_local) residual workspace */
  uint8_t* ws_valid;         /* (b,n_local) validity mask (an OUTPUT as well: bit-exact vs photo.py:15-21) */
  void* ws_hists;            /* como_select_workspace_bytes() */
  void* ws_pair;             /* b*26 elements */
  void* ws_partials;         /* como_ba_partials_elems(b, chunks, m) elements */
  const int* grp_pairs;      /* optional (ngrp,2): pairs of this batch that share their reference slot, two per row (zmode 1,
                                float32): the depth x depth block and the K~ reads are shared inside a row */
  const int* single_pairs;   /* (nsingle): the pairs not listed in grp_pairs */
  int ngrp, nsingle;         /* every pair appears exactly once in grp_pairs U single_pairs; ngrp = 0 / NULL: one pair at a time */
  /* h_is_f64 == 2: ORDER-INDEPENDENT assembly.  Hmat points at a fixed-point system buffer (see como_sys_finalize): two
     planes of `fix_plane` int64 each, {H lower triangle (D*D, row-major, row >= col) | g (D) | err + spare (8)}; every
     contribution is added with exact integer atomics (associative -> the result does not depend on the order in which
     workgroups, streams or ranks deliver it).  gvec / err_out are ignored.  At most 240 pair entries (b) per call: the fraction
     word of an entry holds 256 contributions before a carry could be lost (COMO_ERR_ARG beyond). */
  long fix_plane;
  /* reduce_mode (phase 128): 0 = reduce the per-workgroup records of every pair and expand / scatter them (single GPU);
     1 = reduce only: blocks_fix (b, 3936, 2) int64 receives the per-pair sums in fixed point (all-reduce them with an
     integer SUM -- exact, so every rank ends with identical bits); 2 = expand / scatter from blocks_fix. */
  int reduce_mode;
  void* blocks_fix;
  /* Colour images (`color: rgb`, config/como.yml:7,29; photo.py:24-27): channels = c of the [I_0..I_c-1 | gx_0.. | gy_0..]
     (3c,H,W) stacks and of vals (slots,n,c).  photo.py:112-128 treats every (pixel, channel) residual alike (one global
     median, one Huber weight each, Gram summed over n AND c), so a c-channel keyframe pair is passed as c entries of the
     pair arrays that differ only in pair_chan[p] (their blocks land on the same rows of H).  Pairs listed together in
     grp_pairs must share slot and channel.  channels <= 1: gray, pair_chan ignored. */
  int channels;
  const int* pair_chan;      /* [b] */
  const int* ref_pose;       /* zmode 2: [b] index into poses_all of the REFERENCE keyframe's pose T_wc */
  /* Round 6, phase 128 with the fixed-point system (h_is_f64 = 2, reduce_mode 0): optional grouping of the pairs BY REFERENCE
     SLOT for the assembly -- asm_grp_start (n_asm_grp + 1) offsets into asm_grp_list (b pair indices, every pair once, the pairs
     of a group sharing slot, pose_ref_inds and landmark_inds).  Everything a group's pairs add to the SAME system entries (the
     64 x 64 depth blocks, the reference-pose rows, their gradient parts, the error) is summed over the group first and
     expanded / scattered once: the window of the sequential loop has 3-6 pairs per reference keyframe.  NULL: one pair at a
     time (as before; also whenever pair_blocks_out is requested). */
  const int* asm_grp_start;
  const int* asm_grp_list;
  int n_asm_grp;
} como_ba_args;

long como_ba_partials_elems(int b, int chunks, int m);
int como_ba_linearize_f32(const como_ba_args* args_host, como_stream_t stream);
int como_ba_linearize_f64(const como_ba_args* args_host, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DepthCov native ops (the reference's only FFI, pybind module `como_backends`).
 *
 * como_cross_covariance_*  replaces  cross_covariance(x1,E1,x2,E2,scale) -> K12
 *   reference: como/backend/include/cov.h:10, src/cov.cpp:5-32, src/cov_gpu.cu:17-84, src/cov_cpu.cpp:17-64
 *   x1 (B,N,2) E1 (B,N,2,2) x2 (B,M,2) E2 (B,M,2,2) may be strided views (the sampler passes slices,
 *   depth_cov/core/samplers.py:167-172,266-268): strides_host[14] = element strides of x1[3], E1[4], x2[3], E2[4]
 *   (a HOST array).  K12 (B,N,M) dense, caller-allocated.
 * como_chol_append_obs_info_f32  replaces  get_new_chol_obs_info(L,obs_info,var,k_ni,k_id,k_ii,N)
 *   reference: include/cov.h:18-20, src/cov.cpp:34-65, src/cov_gpu.cu:132-215, src/cov_cpu.cpp:66-85
 *   L (B,n,n) obs_info (B,n,d) var (B,d) k_ni (B,N,1) k_id (B,1,d), all contiguous float32; row N of L and
 *   obs_info and all of var are updated in place.  n <= 64. */
int como_cross_covariance_f32(const float* x1, const float* E1, const float* x2, const float* E2, float scale,
                              float* K12, int B, int N, int M, const long* strides_host, como_stream_t stream);
/* half precision (the reference's dispatch includes at::Half, cov_gpu.cu:73): IEEE binary16 storage and per-operation
 * rounding, scale passed as float */
int como_cross_covariance_f16(const void* x1, const void* E1, const void* x2, const void* E2, float scale, void* K12, int B,
                              int N, int M, const long* strides_host, como_stream_t stream);
int como_cross_covariance_f64(const double* x1, const double* E1, const double* x2, const double* E2, double scale,
                              double* K12, int B, int N, int M, const long* strides_host, como_stream_t stream);
int como_chol_append_obs_info_f32(float* L, float* obs_info, float* var, const float* k_ni, const float* k_id,
                                  float k_ii, int B, int n, int d, int N, como_stream_t stream);
/* como_greedy_next_f32  replaces  greedy_loop.get_next_inds (depth_cov/core/samplers.py:219-239, torch ops there):
 *   var (B,d), coords_domain (B,d,2) normalised, chosen (B,k,2) the k points added since the last call, mask (B,d)
 *   uint8 running "farther than dist_thresh from every chosen point" (initialise to 1), best_idx (B) int64 and
 *   max_stdev (B) outputs stay on the device: the sampler loop needs no host round trip. */
int como_greedy_next_f32(const float* var, const float* coords_domain, const float* chosen, int k, uint8_t* mask,
                         float dist_thresh_sq, long* best_idx, float* max_stdev, int B, int d, como_stream_t stream);
/* como_greedy_loop_f32  replaces  the whole greedy_loop (samplers.py:196-282, terminate_early = False): two launches per
 *   added point (append: k_ni, Cholesky row, k_id, obs_info row, variance downdate; pick: distance mask, argmax, gather
 *   of the chosen point), no host involvement.  State as precalc_entropy_vars leaves it: the first m slots of
 *   coords_n (B,n,2) / E_n (B,n,2,2) / L (B,n,n) / obs_info (B,n,d) filled, var (B,d) = calc_var(...), mask (B,d) ones;
 *   on return slots m..n-1 and coord_vec_inds (B,n) int64 are filled.  n <= 64; contiguous float32.
 *   sd_trace (n+1,B) optional: row i receives the largest remaining standard deviation BEFORE slot i is filled -- the value
 *   the reference's early-termination test reads (samplers.py:255-259) -- so the caller can truncate the sequence after
 *   ONE read-back instead of synchronising on every step.
 *   scratch: optional B * 16 KiB; with it the argmax over the domain runs on many workgroups (two-stage, same ordering rule:
 *   largest cost, then smallest index) -- one workgroup walking 300k candidates took 250 us per added point. */
int como_greedy_loop_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                         float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* max_stdev, float scale,
                         float k_ii, float dist_thresh_sq, int B, int n, int d, int m, float* sd_trace, void* scratch,
                         como_stream_t stream);
/* como_greedy_thin_f32  replaces  sample_sparse_coords(coords_domain = <= 1024 given points, no current points, terminate_early)
 *   (depth_cov/core/samplers.py:38-107 -> precalc_entropy_vars' m = 0 seed :149-165 + greedy_loop :196-282 + the cut :255-259; the
 *   thinning of a new keyframe's tracked points, como/odom/frontend/corr.py:166-176) as ONE launch: coords_domain (d,2) normalised,
 *   E_domain (d,2,2); work arrays coords_n (n,2), E_n (n,2,2), L (n,n), obs_info (n,d), var (d), mask (d), best_idx (1),
 *   sd_trace (n+1); coord_vec_inds (n) int64 receives the picks, *count_out how many of them are valid (cut at the first step whose
 *   largest remaining standard deviation is below stdev_thresh; 0 = K_nn not positive definite). */
int como_greedy_thin_f32(const float* coords_domain, const float* E_domain, float* coords_n, float* E_n, long* coord_vec_inds,
                         float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* sd_trace, float scale,
                         float signal_var, float fixed_var, float dist_thresh_sq, float stdev_thresh, int n, int d, long* count_out,
                         como_stream_t stream);
/* como_greedy_loop_ws_f32: the same with the scratch size stated (floats, >= 4096 B): with >= 4 B ceil(d / 256) floats the
 *   append launch of a step also does the scan of the next pick (mask of the point being added, best candidate per workgroup), so
 *   a step is two launches (append + scan, pick) instead of three; same picks. */
int como_greedy_loop_ws_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                            float* L, float* obs_info, float* var, uint8_t* mask, long* best_idx, float* max_stdev, float scale,
                            float k_ii, float dist_thresh_sq, int B, int n, int d, int m, float* sd_trace, void* scratch,
                            long scratch_floats, como_stream_t stream);
/* Round 6: the same loop for ONE image (B = 1) over a large domain as ONE persistent launch: every thread keeps the obs_info columns of
 * its domain pixels in registers / LDS for the whole loop (the launch-per-step form streams the i previous rows of every pixel for
 * the i-th point: 40-76 MB per step at a 640x480 domain), a step costs one grid-wide exchange of the workgroups' best candidates.
 * Same picks (same operations on the same values, same order rule).  obs_info / var / mask are read only (rows < m of obs_info, the
 * variance and the mask as they stand before the loop); L rows m..n-1, coords_n / E_n / coord_vec_inds slots m..n-1, best_idx,
 * max_stdev / sd_trace are written as by como_greedy_loop_ws_f32.  workspace: como_greedy_persist_workspace_bytes(n, d) bytes;
 * status (device int): 0, or -1 when a grid-wide wait timed out (workgroups not co-resident: the results are invalid).
 * COMO_ERR_ARG when the domain needs more workgroups than the device has compute units (use the launch-per-step loop then). */
long como_greedy_persist_workspace_bytes(int n, int d);
int como_greedy_persist_f32(float* coords_n, float* E_n, long* coord_vec_inds, const float* coords_domain, const float* E_domain,
                            float* L, const float* obs_info, const float* var, const uint8_t* mask, long* best_idx, float* max_stdev,
                            float scale, float k_ii, float dist_thresh_sq, int n, int d, int m, float* sd_trace, void* workspace,
                            int* status, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense reference points in factored form (python path: backend/sparse_map.py:184-230 backproject_cloud +
 * setup_test_points, Mapping.py:661-699).  Inputs: Kt (B,rows,m) dense predictor (slot stride given), pixidx (B,n)
 * rows of the selected pixels (NULL = identity), logzm (B,m), Twc (B,4,4), K (3,3), dlogzm_dTwc (B,m,6).
 * Outputs (structure-of-arrays planes): Pwn (B,3,n), dPwn_dTwc (B,18,n) [full Jacobian incl. the path through
 * logz_m(T_wc)], uvec (B,3,n) = R_wc ray z_n, zbuf (B,n) = depth, logzn_out (B,n) optional,
 * med_out3 (B,3) = {exact median depth (sparse_map.py:220), 1.4826*median, n}.  hists: B * select workspace.
 * pixcoord (B,n) optional: linear pixel index row*W+col of every reference pixel when it differs from its K~ row
 * (two-frame SfM passes K~ rows of the selected pixels only, two_frame_sfm.py:246-252); NULL = the K~ row index.
 * flags bit 0: hists is already zero (skip the clear); bit 1: points only -- the kernel and its pass-0 depth histogram,
 * no median; bit 2: median only -- the remaining select passes + finish on the zbuf / hists a bit-1 call left (the two halves
 * may run on different streams: nothing between the reference points and the priors needs the median);
 * bit 3: depth only -- z_n = exp(K~[n,:] logz_m) of every row into zbuf and its exact per-keyframe median, no planes
 * (Pwn / dPwn_dTwc / uvec may be NULL): Mapping.store_vars' full-image median depth (Mapping.py:749-758), the value
 * the priors and the landmark re-initialisation use;
 * bit 4: compact -- dPwn_dTwc receives only the six planes (B,6,n) dlogz_n/dT_wc = K~[n,:] dlogz_m/dT_wc and uvec is not
 * written (may be NULL): what como_ba_args.zmode 2 consumes (9 planes written per pixel instead of 24);
 * bit 6 (with bit 1): no median will be asked for -- the pass-0 histogram is neither cleared nor accumulated (the depth image the
 * tracker's reference is rebuilt from on every frame, Mapping.get_kf_ref_data, Mapping.py:499-512). */
int como_dense_ref_f32(const float* Kt, long kt_slot_stride, const int* pixidx, const float* logzm, const float* Twc,
                       const float* K, const float* dlogzm_dTwc, int B, int n, int m, int Wimg, float* Pwn,
                       float* dPwn_dTwc, float* uvec, float* zbuf, float* logzn_out, void* hists, float* med_out3,
                       const int* pixcoord, int flags, como_stream_t stream);
int como_dense_ref_f64(const double* Kt, long kt_slot_stride, const int* pixidx, const double* logzm, const double* Twc,
                       const double* K, const double* dlogzm_dTwc, int B, int n, int m, int Wimg, double* Pwn,
                       double* dPwn_dTwc, double* uvec, double* zbuf, double* logzn_out, void* hists, double* med_out3,
                       const int* pixcoord, int flags, como_stream_t stream);
/* Round 6: the dense reference with PASS 1 of the window's photometric system fused in (como_ba_linearize_* phase 1 =
 * batch_photo_cost's warp / sample / residual / validity, photo.py:104-121, + the first digit of the robust scale's select): the
 * reference point of a pixel is in a register when the dense reference forms it, and the pairs that use keyframe b as reference only
 * need it, the pixel's intensity and their pair constants.  Call como_ba_linearize_* with phase 1024 first (pair constants into
 * ws_pair), this with the fields below, then como_ba_linearize_* with phase 0xFE (select passes, blocks, assembly).  Gray images,
 * the whole pixel range, the matrix-core kernel (float32, or float64 points).  r_out / valid_out / rhists receive exactly what the
 * separate pass writes. */
typedef struct como_dr_fuse {
  const int* ref_pairs;      /* (B, np_max): pairs whose reference slot is keyframe b, -1 padded */
  int np_max;
  const void* pair_T;        /* como_ba_args.ws_pair: (b,12) inverse target poses ... */
  const void* pair_aff;      /* ... followed by (b,2) relative affine parameters (ws_pair + 12 b elements) */
  const void* vals;          /* (B,n) reference intensities */
  const void* img_base;      /* as como_ba_args.img_base / tgt_img */
  const long* tgt_img;
  void* r_out;               /* como_ba_args.ws_r (b,n) */
  void* valid_out;           /* como_ba_args.ws_valid (b,n) u8 */
  void* rhists;              /* como_ba_args.ws_hists (zeroed) */
  int H, W, anorm_f32;
} como_dr_fuse;
int como_dense_ref_fused_f32(const float* Kt, long kt_slot_stride, const int* pixidx, const float* logzm, const float* Twc,
                             const float* K, const float* dlogzm_dTwc, int B, int n, int m, int Wimg, float* Pwn, float* dPwn_dTwc,
                             float* uvec, float* zbuf, float* logzn_out, void* hists, float* med_out3, const int* pixcoord, int flags,
                             const como_dr_fuse* fuse, como_stream_t stream);
int como_dense_ref_fused_f64(const double* Kt, long kt_slot_stride, const int* pixidx, const double* logzm, const double* Twc,
                             const double* K, const double* dlogzm_dTwc, int B, int n, int m, int Wimg, double* Pwn, double* dPwn_dTwc,
                             double* uvec, double* zbuf, double* logzn_out, void* hists, double* med_out3, const int* pixcoord, int flags,
                             const como_dr_fuse* fuse, como_stream_t stream);

/* Mapping.store_vars' full-image median depth (Mapping.py:749-758) WITHOUT re-reading all of K~ every GN iteration: instead of
 * the depth-only pass of como_dense_ref_* (flag 8) this evaluates z_n = exp(K~[n,:] logz_m) only for the pixels whose cached
 * log-depth interval can still straddle the new median (|delta logz_n| <= ||K~[n,:]||_1 max_k |delta logz_m,k|); the others
 * write 0 / +inf according to their side, so the exact select on zbuf returns the exact median of the full image.
 * State (caller-owned, one per keyframe set): lref / err / l1 (B,rows), l1max (B) float, logzm_prev (B,m) (updated by the call),
 * med_prev3 (B,3) = the {median, ., .} the PREVIOUS call's select wrote (read before this call's select overwrites it).
 * init != 0: builds the state (every pixel evaluated: the cost of the depth-only pass).  zbuf (B,rows) and hists as for
 * como_dense_ref_* with flags 2|8 (pass-0 histogram accumulated into an already zeroed workspace); continue with
 * como_dense_ref_*(flags 4|8) for the remaining select passes + finish.  ncand (B) optional: candidates counted (+=). */
int como_depth_band_f32(const float* Kt, long kt_slot_stride, const float* logzm, float* logzm_prev, int B, int rows, int m,
                        float* lref, float* err, float* l1, float* l1max, const float* med_prev3, float* zbuf, void* hists,
                        unsigned* ncand, int init, como_stream_t stream);
int como_depth_band_f64(const double* Kt, long kt_slot_stride, const double* logzm, double* logzm_prev, int B, int rows, int m,
                        double* lref, double* err, double* l1, float* l1max, const double* med_prev3, double* zbuf, void* hists,
                        unsigned* ncand, int init, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DepthCov covariance-kernel assembly, Python-twin formula (python path: depth_cov/core/kernels.py:22-88,
 * covariance.py:10-39) and conditioning (Mapping.py:430-468 prep_predictor).
 * como_kernel_matrix_*: out (B,N,M) = scale * k(x1_i,E1_i ; x2_j,E2_j); x (B,.,2) normalised (row,col), E (B,.,2,2).
 * como_ktilde_*: out (B,Hp,Wp,m) = K_nm K_mm^-1 for every photo pixel; cov (B,4,Hc,Wc) covariance image,
 *   xm (B,m,2) / Em (B,m,2,2) inducing points, Kinv (B,m,m) = K_mm^-1.  K_nm is never written to memory. */
int como_kernel_matrix_f32(const float* x1, const float* E1, const float* x2, const float* E2, float scale, float* out,
                           int B, int N, int M, como_stream_t stream);
int como_kernel_matrix_f64(const double* x1, const double* E1, const double* x2, const double* E2, double scale,
                           double* out, int B, int N, int M, como_stream_t stream);
int como_ktilde_f32(const float* cov, int Hc, int Wc, const float* xm, const float* Em, const float* Kinv, float scale,
                    int B, int Hp, int Wp, int m, float* out, como_stream_t stream);
int como_ktilde_f64(const double* cov, int Hc, int Wc, const double* xm, const double* Em, const double* Kinv,
                    double scale, int B, int Hp, int Wp, int m, double* out, como_stream_t stream);
/* the same with a second output: out_f32 (may be NULL) receives the values rounded to float32 -- the predictor mirror of the
 * per-pixel kernels (mapping `pix_dtype: float`), written by the kernel that forms K~ instead of a conversion pass over it.
 * Both outputs may be slots of the window's predictor buffers. */
int como_ktilde_mirror_f64(const double* cov, int Hc, int Wc, const double* xm, const double* Em, const double* Kinv, double scale, int B,
                           int Hp, int Wp, int m, double* out, float* out_f32, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Keyframe-insertion glue with the arithmetic of the reference's torch chains, one launch per chain:
 * como_cov_params_at  replaces  normalize_coordinates (como/utils/coords.py:12-15) + interpolate_kernel_params
 *   (depth_cov/core/gaussian_kernel.py:52-79: grid_sample bilinear / border / align_corners=False): coords (B,N,2) row/col pixels
 *   (float64 or float32) -> cn (B,N,2) normalised and E (B,N,2,2), both in the covariance image's type (float32 or float64;
 *   float64 coordinates are normalised in float64 and then cast, as samplers.py:64-78 does).
 * como_diag_cov_*     replaces  DiagonalCovarianceModule.forward (depth_cov/core/covariance.py:42-50, kernels.py:69-88 at Q = 0).
 * como_kernel_matrices_*  replaces  calc_kernel_matrices (depth_cov/core/distill_depth.py:8-27): cm, Em, cn, En, K_mm (B,m,m),
 *   K_nm (B,n,m), diag K_nn (B,n) of pixel coordinates coords_m (B,m,2), coords_n (B,n,2).
 * como_backproject_*  replaces  backprojection (como/geometry/camera.py:43-54, values only): P (n,3) = z (n) * ray(p (n,2) x/y; K (3,3)). */
int como_cov_params_at(const void* cov, int Hc, int Wc, const void* coords, int coords_is_f64, int out_is_f64, int N, void* cn,
                       void* E, int B, como_stream_t stream);
int como_diag_cov_f32(const float* E, long total, float scale, float* out, como_stream_t stream);
int como_diag_cov_f64(const double* E, long total, double scale, double* out, como_stream_t stream);
int como_kernel_matrices_f32(const float* cov, int Hc, int Wc, const float* coords_m, int m, const float* coords_n, int n,
                             float scale, float* cm, float* Em, float* cn, float* En, float* Kmm, float* Knm, float* Kdiag, int B,
                             como_stream_t stream);
int como_kernel_matrices_f64(const double* cov, int Hc, int Wc, const double* coords_m, int m, const double* coords_n, int n,
                             double scale, double* cm, double* Em, double* cn, double* En, double* Kmm, double* Knm, double* Kdiag,
                             int B, como_stream_t stream);
int como_backproject_f32(const float* K, const float* p, const float* z, long n, float* P, como_stream_t stream);
int como_backproject_f64(const double* K, const double* p, const double* z, long n, double* P, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weighted normal equations of the depth distillation (python path: como/utils/lin_alg.py:82-87 lstsq_chol's A^T A / A^T b as
 * used by depth_cov/core/distill_depth.py:52-84, 122-148), float64, the rows read in place:
 *   r_i = y_i - sum_k A[i][k] c[k] (c may be NULL);  AtA (m,m) = sum_i w_i A_i A_i^T (both triangles);  Atb (m) = sum_i w_i A_i r_i;
 *   stats (4, may be NULL) = {sum w, sum w r, sum w r^2, number of rows with w != 0}.
 * A (n rows, row_stride elements apart, m <= 64 columns, m % 4 == 0, 16-byte aligned), w (n) or NULL (= 1): rows with w == 0
 * contribute exact zeros whatever they hold; workspace of como_gram_workspace_bytes() bytes.  Deterministic (fixed summation order). */
long como_gram_workspace_bytes(void);
int como_gram_f64(const double* A, long row_stride, int n, int m, const double* w, const double* y, const double* c, double* AtA,
                  double* Atb, double* stats, void* workspace, como_stream_t stream);
/* como_predictor_f64: the GP predictor of the distillation (como/depth_cov/core/distill_depth.py:30-48 get_predictor) in one pass:
 * Kt (n x m, row stride ldo >= m, ldo <= 64; columns m .. ldo-1 are written as zeros) = Knm (n x m) inv (m x m),
 * var (n) = diag (n) - rowsum(Knm o Kt).  float64, m <= 64. */
int como_predictor_f64(const double* Knm, const double* inv, const double* diag, int n, int m, int ldo, double* Kt, double* var,
                       como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense SPD solve delta = H^-1 g, float64 (python path: backend/linear_system.py:101-112 solve_system).
 * H (D,D) row-major (lower triangle read), g (D), delta (D) out, workspace of como_chol_workspace_bytes(D) bytes,
 * info (1 int, device): 0 = ok, i > 0 = leading minor i not positive definite (cholesky_ex's info, reported instead
 * of being ignored as the reference does).
 * Workspace layout (Dp = D + 1 padded to whole 32-wide block columns): working copy incl. the appended identity rows of the
 * ride-along back-substitution (2 Dp x Dp) | factor (Dp x Dp) | inverted diagonal blocks (Dp x 32) | x accumulator (Dp);
 * up to 40 block columns (D < 1280) delta comes out of the factorisation launches themselves, above that from separate
 * substitution kernels (csrc/chol.hip). */
long como_chol_workspace_bytes(int D);
int como_chol_solve_f64(const double* H, const double* g, double* delta, void* workspace, int D, int* info,
                        como_stream_t stream);
/* the factorisation + substitutions of a system como_sys_finalize_pack already packed into `workspace` */
int como_chol_solve_packed_f64(double* delta, void* workspace, int D, int* info, como_stream_t stream);
/* The persistent one-launch solver (csrc/cholp.hip) spin-waits on counters and needs all its workgroups co-resident (checked once
 * against hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units); when they are not -- another process or kernel holds
 * compute units -- its bounded waits time out, *info = -1 and delta is left unwritten.  como_chol_set_persistent(0) selects the
 * multi-launch solver for the rest of the process (what a caller does after reading info = -1, before solving again; several
 * processes sharing one device should start with it off: COMO_CHOLP=0).  Returns the previous setting. */
int como_chol_set_persistent(int enable);
int como_chol_persistent_state(void);
/* test switch: the persistent solver's chain workgroup leaves at once, so that the other workgroups' waits run into their
 * time-out (~2 s) and the solve ends with *info = -1 -- the path a lost co-residency takes */
void como_chol_debug_stall(int on);

/* ------------------------------------------------------------------------------------------------
 * Conditioning of the small SPD systems of the DepthCov path (n <= 80), batched: ONE launch, one workgroup per matrix, float32
 * or float64.  Replaces torch.linalg.cholesky(_ex) + torch.cholesky_solve where the reference conditions m x m systems:
 * K_mm + 1e-6 I -> L_mm, K_mm^-1 (como/odom/Mapping.py:450-458); get_predictor and the distillation's normal equations
 * (depth_cov/core/distill_depth.py:30-48, utils/lin_alg.py:82-87); the sampler's initial factor (depth_cov/core/samplers.py:
 * 110-165); the (6 + m) two-frame solve (odom/frontend/two_frame_sfm.py:288-293).
 * A (B,n,n) row-major, lower triangle read.  Optional outputs (NULL = not wanted): L (B,n,n) lower factor (upper triangle
 * zero), Ainv (B,n,n) = A^-1 (symmetric, both triangles), X (B,n,k) = A^-1 rhs for rhs (B,n,k) (rhs and X go together);
 * info (B) = 0 or the order of the first non-positive leading minor (torch.linalg.cholesky_ex's info; nothing is raised). */
int como_chol_small_f32(const float* A, int B, int n, float* L, float* Ainv, const float* rhs, int k, float* X, int* info,
                        como_stream_t stream);
int como_chol_small_f64(const double* A, int B, int n, double* L, double* Ainv, const double* rhs, int k, double* X, int* info,
                        como_stream_t stream);
/* X (B,n,d) = L^-1 Bm: forward substitution with a lower-triangular L (B,n,n), n <= 64, for MANY right-hand sides Bm (B,n,d)
 * (torch.linalg.solve_triangular(L, K, upper=False): samplers.py:117-118 get_obs_info with d = the whole pixel domain;
 * two_frame_sfm.py:115-125 with Bm = I).  trans != 0: X = L^-T Bm (backward substitution) -- the two calls in sequence are
 * torch.cholesky_solve(Bm, L) for many columns (two_frame_sfm.py:363-366: K_mm^-1 K_mn over every pixel).  X may alias Bm. */
int como_trsm_lower_f32(const float* L, const float* Bm, float* X, int B, int n, long d, int trans, como_stream_t stream);
int como_trsm_lower_f64(const double* L, const double* Bm, double* X, int B, int n, long d, int trans, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused O(B*m) bookkeeping of one window GN iteration (python path: Mapping.prep_geometry_scaffold
 * Mapping.py:603-659 + sparse_map.py:18-60; the prior factors of Mapping.iterate :809-917 = the files of odom/factors/;
 * linear_system.update_vars :115-152).  All state is float64; px_* are mirrors in the per-pixel dtype. */
typedef struct como_win_args {
  int B, F, m, L, nfix;          /* keyframes, frames (keyframes + recent), inducing points per KF, landmarks, anchored landmarks */
  int pix_is_f64;                /* element type of the px_* outputs */
  int median_new_is_f32;         /* element type / stride of median_new (the dense-reference kernel's med_out3) */
  int median_new_stride;
  long D;
  const double* poses;           /* (F,4,4) T_wc, keyframes first */
  const double* aff;             /* (F,2) */
  const double* K;               /* (3,3) */
  const double* median;          /* (B) median depths of the PREVIOUS iteration (re-init test, sparse_map.py:26) */
  const double* pm_first;        /* (B,m,2) first-observation pixel (x,y) */
  const double* Kmm_inv;         /* (B,m,m) */
  const double* pose_anchor;     /* (4,4) */
  const double* aff_anchor;      /* (2) */
  const double* P_anchor;        /* (nfix,3) */
  double* P_m;                   /* (L,3) landmarks, re-initialised in place */
  const int* lm_ids;             /* (B,m) landmark of every (keyframe, slot) */
  const int* first_frame;        /* (L) first observer keyframe of a landmark */
  const int* first_slot;         /* (L) its slot there */
  const int* fix_lm;             /* (nfix) anchored landmark ids */
  const uint8_t* first_mask;     /* (B,m) obs_ref_mask */
  const long* pose_inds;         /* (B,8) rows of H */
  const long* landmark_inds;     /* (B,3m) */
  const long* fix_inds;          /* (nfix*3) rows of H of the anchored landmark coordinates */
  const void* median_new;        /* (B) median depths of THIS iteration's dense reference (priors use log of it) */
  double* pm; double* logzm; double* invz; double* dzdP; double* dlogz_dT; double* dlogz_dP; double* dp_dP; double* dp_dT;
  double* init_Pm; int* reinit_flag;
  void* px_logzm; void* px_invz; void* px_dzdP; void* px_dlogz_dT; void* px_poses; void* px_aff;
  double s_gp, s_ld, s_px, s_pose, s_aff, s_lm;   /* sigmas: 1, 1, 1e-2, cfg pose_prior, cfg scale_prior, cfg scale_prior */
  double* H; double* g; double* err;              /* err: 8 doubles {gp, log-depth, pixel, pose, affine, landmarks, -, -} */
  void* zero_a; long zero_a_bytes;                /* optional: two buffers (multiples of 16 bytes) that como_win_scaffold clears */
  void* zero_b; long zero_b_bytes;                /*   together with err -- the radix-select histograms of this iteration */
  double* median_out;                             /* optional (B): como_win_priors stores median_new here (next iteration's `median`) */
  void* sysfix; long fix_plane;                   /* optional: fixed-point system buffer (como_sys_finalize); then the priors are added
                                                     there (lower triangle, exact integer atomics) and H / g / err are not touched;
                                                     the 6 prior errors go to slots 1..6 of the err block */
  /* While the window is filling (fewer keyframes than the graph holds) the reference replaces the landmark anchors of
     keyframe 0 by a scale prior on its MEAN predicted log-depth (Mapping.py:900-917, factors/gp_priors.py:84-150):
     r = mld_J . logz_m[0] - *mld_anchor, information 1 / s_mld^2, with mld_J (m) = the column means of keyframe 0's K~
     (constant for a topology).  mld_J != NULL (and nfix == 0) selects it; its error goes to slot 6 of the err block like the
     anchors' (the two are alternatives). */
  const double* mld_J; const double* mld_anchor; double s_mld;
  void* zero_c; long zero_c_bytes;                /* optional third buffer cleared by como_win_scaffold (the fixed-point system buffer) */
} como_win_args;

int como_win_scaffold(const como_win_args* args_host, como_stream_t stream);
int como_win_priors(const como_win_args* args_host, como_stream_t stream);
/* The log-depths the NEXT iteration's como_win_scaffold will compute (same kernel, same arithmetic) over the state as it stands, every
 * other output into `scratch` (como_win_logz_ahead_scratch_bytes, 16-byte aligned); the window is not written.  px_logzm_out (B,m) in
 * the window's pix dtype; zero / zero_bytes: an optional buffer (multiple of 16 bytes) cleared in the same launch.  Lets the
 * full-image median of the next iteration (Mapping.store_vars, /root/reference/como/odom/Mapping.py:749-758) be streamed between two
 * iterations of the sequential loop instead of beside the block kernel. */
long como_win_logz_ahead_scratch_bytes(int B, int m, int L, int F);
int como_win_logz_ahead(const como_win_args* args_host, void* scratch, long scratch_bytes, void* px_logzm_out, void* zero,
                        long zero_bytes, como_stream_t stream);

/* Order-independent normal equations.  sysfix = 2 planes x fix_plane int64 (fix_plane >= D*D + D + 8): plane 0 holds
 * floor(v) sums, plane 1 the fractional parts in units of 2^-56, of {H[i][j] for i >= j at i*D+j | g at D*D | errors at
 * D*D+D (slot 0 photometric, 1..6 priors, 7 = count of non-finite contributions)}.  como_sys_finalize converts to float64:
 * H (D,D) BOTH triangles (exactly symmetric), g (D), err8 (8); a non-finite contribution anywhere poisons H[0][0] with NaN
 * so that the factorisation reports it.  The buffer must be zeroed before the first contribution of an iteration. */
long como_sys_fix_plane_elems(long D);
int como_sys_finalize(const void* sysfix, long fix_plane, long D, double* H, double* g, double* err8, como_stream_t stream);
/* The same conversion fused with the first step of como_chol_solve_f64: also writes the solver's packed working copy into
 * chol_workspace (como_chol_workspace_bytes(D) bytes) and resets *info; continue with como_chol_solve_packed_f64 (one launch
 * less on the critical path of every GN iteration). */
int como_sys_finalize_pack(const void* sysfix, long fix_plane, long D, double* H, double* g, double* err8, void* chol_workspace,
                           int* info, como_stream_t stream);
int como_win_update(const double* delta, double* poses, double* aff, const long* frame_inds, int F, double* P_m, int L,
                    long lm_start, como_stream_t stream);
/* The same, guarded by the solver's status word (device int, como_chol_solve_*'s `info`): with *info != 0 -- a non-positive pivot,
 * or -1 = the persistent solver's time-out, which leaves delta unwritten -- nothing is updated (the reference swallows the error,
 * linear_system.py:109, and applies whatever came out; SURVEY.md section 5 asks for the status to be acted on).  The caller reads
 * `info` at its next synchronisation point and decides (re-solve on the multi-launch solver / raise). */
/* normalizeSE3_inplace (como/geometry/lie_algebra.py:98-101): the rotation block of n contiguous (4,4) poses is replaced by its
 * nearest rotation (the orthogonal polar factor = U V^T of the SVD the reference takes), by Newton's iteration on the device; the
 * input must be a rotation up to rounding / float32 noise.  In place. */
int como_se3_normalize_f32(float* poses, int n, como_stream_t stream);
int como_se3_normalize_f64(double* poses, int n, como_stream_t stream);
/* A tracked frame's pose and affine brightness in the world frame (Mapping.handle_tracking_data, Mapping.py:580-598): T_out (4,4) =
 * T_w_kf inv(T_curr_kf) (get_T_w_curr, como/geometry/transforms.py:6-8), aff_out (2) = (a_kf + a_cur, b_kf + b_cur exp(a_cur))
 * (get_aff_w_curr, como/geometry/affine_brightness.py:5-10).  T_curr_kf / aff_curr_kf: float32 (cur_is_f32, widened first) or float64. */
int como_frame_world_f64(const double* T_w_kf, const void* T_curr_kf, const double* aff_w_kf, const void* aff_curr_kf, int cur_is_f32,
                         double* T_out, double* aff_out, como_stream_t stream);
/* The record the host reads back of a tracked frame (Tracking.handle_frame, como/odom/Tracking.py:315-379: |t| of T_curr_kf and the
 * reprojection statistics feed check_keyframe / check_one_way_frame, :117-161; poses as the reference keeps them), one launch:
 * out[3 + levels + 34] = [|t| | median[0] | (float) nseen[0] | level_records[l * record_stride + 104] for every level |
 * T_curr_kf (16) | aff_curr_kf (2) | T_w_curr = T_w_kf inv(T_curr_kf) (16), get_T_w_curr, como/geometry/transforms.py:6-8]. */
int como_track_frame_record_f32(const float* T_curr_kf, const float* aff_curr_kf, const float* T_w_kf, const float* median, const int* nseen,
                                const float* level_records, int levels, int record_stride, float* out, como_stream_t stream);
int como_win_update_checked(const double* delta, double* poses, double* aff, const long* frame_inds, int F, double* P_m, int L,
                            long lm_start, const int* info, como_stream_t stream);
/* invertSE3 (como/geometry/lie_algebra.py:83-95 without the Jacobian; the inverse inside get_T_w_curr / get_rel_pose,
 * transforms.py:6-13): out[i] = [R^T | -(R^T t); 0 0 0 1] for n row-major 4x4 poses (in and out may not alias). */
int como_se3_inverse_f32(const float* T, float* out, int n, como_stream_t stream);
int como_se3_inverse_f64(const double* T, double* out, int n, como_stream_t stream);
/* como_se3_compose_*: out_i = op(A_i) op(B_i) for n pose pairs (4x4 row-major; na / nb = 1: that operand is one pose for all i);
 * mode 0: A B, 1: inv(A) B (get_rel_pose, como/geometry/transforms.py:11-13), 2: A inv(B) (get_T_w_curr, transforms.py:6-8). */
int como_se3_compose_f32(const float* A, const float* B, float* out, int n, int na, int nb, int mode, como_stream_t stream);
int como_se3_compose_f64(const double* A, const double* B, double* out, int n, int na, int nb, int mode, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DepthCov covariance network, float32 inference (python path: como/depth_cov/nn/UNet.py:57-78 UNet.forward,
 * como/depth_cov/nn/layers.py:5-75 ResidualConv / DownConv / UpConv, DepthCovModule.py:80-87, the output
 * activation gaussian_kernel.py:6-49 and Mapping.run_model's antialiased resizes Mapping.py:409-428).
 * All tensors NCHW contiguous.
 *  conv2d   : stride 1, zero "same" padding, ks in {1,3}; wt is the torch weight (Cout,Cin,ks,ks) re-laid as
 *             [ks*ks][CinP][Cout] with CinP = Cin rounded up to 4 (zero rows); bias may be NULL; the result is written
 *             to channels [out_coff, out_coff+Cout) of an (N,out_ctot,H,W) tensor (torch.cat of UpConv, layers.py:72);
 *             gn_sums (32,N,gn_groups,2) doubles (32 contention slots), optional, pre-zeroed: per-group sum / sum of
 *             squares of the outputs are ADDED (the statistics of the GroupNorm that follows, without another pass over the tensor).
 *  groupnorm: nn.GroupNorm(G, C) (biased variance, eps) followed by act: 0 none, 1 LeakyReLU(slope),
 *             2 LeakyReLU(residual + y) (ResidualConv.forward, layers.py:23-27); statistics either from `sums` (as
 *             accumulated by conv2d) or, when sums is NULL, computed here into stats (N*G*2 float scratch).
 *  maxpool2 : nn.MaxPool2d(2); upsample2x: nn.Upsample(scale 2, bilinear, align_corners=False);
 *  normalize: torchvision Normalize(mean, std) on 3 channels (mean3/std3 are HOST pointers);
 *  cov_act  : normalize_params_cov + kernel_params_to_covariance, (N,3,HW) -> (N,4,HW) = [x, s, s, z];
 *  resize_aa: F.interpolate(mode="bilinear", antialias=True, align_corners=False). */
int como_nn_conv2d_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                       int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups,
                       como_stream_t stream);
/* Round 5, fused layers (csrc/nn.hip): a ResidualConv (layers.py:5-27) is conv1 -> conv2 (reading its input THROUGH the GroupNorm +
 * LeakyReLU in between) -> conv3 (adding the normalised main branch and the activation in its epilogue).
 *  conv2d_fused : conv2d with   pro_scsh (N,Cin,2) floats, optional, ks = 3: the input is lrelu(in * sc[c] + sh[c]);
 *                               res (N,Cout,H,W) + res_scsh (N,Cout,2), optional: out = lrelu(conv + bias + res * sc[c] + sh[c]).
 *  gn_finalize  : the (32,N,G,2) sums a convolution accumulated -> scsh (N,C,2): sc = rstd_g gamma_c, sh = beta_c - mean_g sc.
 *  conv3x3_deep : a 3x3 layer of the deep levels (<= 768 pixels) split along its reduction dimension, deterministic: partial sums
 *                 in `part` (como_nn_deep_part_floats floats), a second launch adds them + bias, writes `out`, and -- gamma / beta
 *                 given -- scsh (N,Cout,2) of the GroupNorm(G, Cout) that follows. */
int como_nn_conv2d_fused_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                             int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups,
                             const float* pro_scsh, const float* res, const float* res_scsh, float slope, como_stream_t stream);
int como_nn_gn_finalize_f32(const double* sums, const float* gamma, const float* beta, int N, int C, int G, int HW, float eps,
                            float* scsh, como_stream_t stream);
/* Round 6: conv2d_fused (no residual) + gn_finalize in ONE launch -- the convolution's last wave (an arrival counter behind the
 * statistics) forms scsh (N,Cout,2) of the GroupNorm(gn_groups, Cout) that follows (layers.py:21-24), value for value what
 * como_nn_gn_finalize_f32 writes.  gn_sums: 32 * N * gn_groups * 2 doubles + ONE more 8-byte word (the counter), zeroed by the caller.
 * Measured slower than the two launches on MI355X (a tiled layer 14-17 -> 20-29 us against 4.7 us for the launch it replaces): the
 * product uses it only with COMO_NN_GN_IN_CONV=1. */
int como_nn_conv2d_gn_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                          int H, int W, int ks, int out_ctot, int out_coff, double* gn_sums, int gn_groups, const float* pro_scsh,
                          float slope, const float* gamma, const float* beta, float eps, float* scsh, como_stream_t stream);
long como_nn_deep_part_floats(int N, int Cin, int Cout, int H, int W);
int como_nn_conv3x3_deep_f32(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int CinP, int Cout,
                             int H, int W, int out_ctot, int out_coff, const float* pro_scsh, float slope, float* part,
                             long part_floats, int G, const float* gamma, const float* beta, float eps, float* scsh,
                             como_stream_t stream);
int como_nn_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                          float* stats, const double* sums, int N, int C, int G, int HW, float eps, float slope, int act,
                          como_stream_t stream);
int como_nn_maxpool2_f32(const float* in, float* out, int NC, int H, int W, como_stream_t stream);
int como_nn_upsample2x_f32(const float* in, float* out, int NC, int H, int W, como_stream_t stream);
int como_nn_normalize_f32(const float* in, float* out, int N, int HW, const float* mean3, const float* std3,
                          como_stream_t stream);
int como_nn_cov_act_f32(const float* in, float* out, int N, int HW, como_stream_t stream);
int como_nn_resize_aa_f32(const float* in, float* out, int NC, int Hi, int Wi, int Ho, int Wo, como_stream_t stream);
int como_nn_resize_aa_f64(const double* in, double* out, int NC, int Hi, int Wi, int Ho, int Wo, como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-frame image operators (python paths: como/utils/image_processing.py:8-44 ImageGradientModule,
 * :47-87 GaussianBlurModule + ImagePyramidModule level step, como/odom/backend/sparse_map.py:116-142
 * subselect_pixels, como/odom/frontend/photo_tracking.py:46-74 precalc_jacobians).  NCHW contiguous.
 *  img_grads       : img (N,C,H,W) -> out (N,3C,H,W) = cat(img, Scharr_x/32, Scharr_y/32), reflect padding
 *                    (the layout Mapping.get_img_and_grads / Tracking build with torch.cat).
 *  img_blur_down   : [1 2 1]^2/16 blur (reflect) followed by [0::2, 0::2]: (NC,H,W) -> (NC,ceil(H/2),ceil(W/2)).
 *  subselect_pixels: gray img_and_grads (B,3,H,W); per window x window cell the FIRST maximum of sqrt(gx^2+gy^2)
 *                    (max_pool2d(return_indices) semantics); coords (B,n,2) int64 (row,col), n = (H/w)(W/w);
 *                    pixidx (B,n) int32 = row*W+col or NULL (the K~ row index of como_dense_ref_*).
 *  track_precalc_jac: dI_dw (N,2), P (N,3), vals (N), K (3,3) -> J (N,8), c = 1. */
int como_img_grads_f32(const float* img, float* out, int N, int C, int H, int W, como_stream_t stream);
int como_img_grads_f64(const double* img, double* out, int N, int C, int H, int W, como_stream_t stream);
/* rgb_to_gray: (N,3,H,W) -> (N,1,H,W), ITU-R 601-2 luma (0.2989 r + 0.587 g) + 0.114 b, each operation rounded on its own
 * (torchvision's rgb_to_grayscale as the reference calls it in Mapping.get_img_and_grads / Tracking.prep_tracking_img). */
int como_rgb_to_gray_f32(const float* rgb, float* out, int N, int H, int W, como_stream_t stream);
int como_rgb_to_gray_f64(const double* rgb, double* out, int N, int H, int W, como_stream_t stream);
/* frame_stack: Mapping.get_img_and_grads (Mapping.py:369-379) of one gray-mode frame in ONE launch: rgb (3,H,W) float32
 * (rgb_is_f32: widened, as .to(float64)) or float64 -> stack (3,H,W) float64 = [luma | Scharr_x/32 | Scharr_y/32] (the values of
 * rgb_to_gray_f64 followed by img_grads_f64), and, when stack_pix != NULL, the same rounded to float32 (the per-pixel kernels'
 * mirror).  Both destinations may be slots of the window's image buffers. */
int como_frame_stack_f64(const void* rgb, int rgb_is_f32, int H, int W, double* stack, float* stack_pix, como_stream_t stream);
int como_img_blur_down_f32(const float* img, float* out, int NC, int H, int W, como_stream_t stream);
/* The tracker's three-level image pyramid of one colour frame (3,H,W) in one launch (Tracking.prep_tracking_img,
 * /root/reference/como/odom/Tracking.py:103-107): gray (H,W), l1 = blur_down(gray), l2 = blur_down(l1) -- bit-identical to
 * como_rgb_to_gray_f32 + 2 x como_img_blur_down_f32; up to eight small buffers (16-byte aligned, multiples of 16 bytes) are cleared in
 * the same launch. */
int como_track_frame_pyramid3_f32(const float* rgb, float* gray, float* l1, float* l2, int H, int W, void* const* zero_ptrs,
                                  const long* zero_bytes, int n_zero, como_stream_t stream);
int como_img_blur_down_f64(const double* img, double* out, int NC, int H, int W, como_stream_t stream);
/* img_blur: the same blur without decimation (GaussianBlurModule); depth_pool2: pyr_depth (como/data/depth_resize.py:6-36)
 * with kernel_size 2, mode 0 bilinear, 1 nearest_neighbor, 2 max, 3 min, 4 masked_bilinear. */
int como_img_blur_f32(const float* img, float* out, int NC, int H, int W, como_stream_t stream);
int como_img_blur_f64(const double* img, double* out, int NC, int H, int W, como_stream_t stream);
int como_depth_pool2_f32(const float* in, float* out, int NC, int H, int W, int mode, como_stream_t stream);
int como_depth_pool2_f64(const double* in, double* out, int NC, int H, int W, int mode, como_stream_t stream);
int como_subselect_pixels_f32(const float* img_and_grads, int B, int H, int W, int window, long* coords, int* pixidx,
                              como_stream_t stream);
int como_subselect_pixels_f64(const double* img_and_grads, int B, int H, int W, int window, long* coords, int* pixidx,
                              como_stream_t stream);
int como_track_precalc_jac_f32(const float* dI_dw, const float* P, const float* vals, const float* K, float* J, long N,
                               como_stream_t stream);
int como_track_precalc_jac_f64(const double* dI_dw, const double* P, const double* vals, const double* K, double* J, long N,
                               como_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise glue of a keyframe insertion, fused (csrc/kfglue.hip): each entry replaces a chain of 4 .. 12 tiny torch launches
 * with the same operations in the same order (one rounding per torch op, no contraction).
 *  predictor_sinv   : como/depth_cov/core/distill_depth.py:42-46 -- var += min over the rows that count (row_mask bytes, optional)
 *                     + 1e-8; sinv = 1 / sqrt(var).  partial: scratch of >= 64 doubles.
 *  distill_prep     : distill_depth.py:96-111, 152-166 -- z = z_obs[i * z_stride] (e.g. the third component of points (n,3));
 *                     ok = z > min_depth [& obs_mask]; zs = ok ? z : 1 (optional output);
 *                     y = log(zs); w = ok ? s^2 : 0 with s = sinv[i] (sinv NULL: 1 / stdev_dev[0] -- a device scalar, no read-back --
 *                     or, stdev_dev NULL too, sinv_scalar) for weight_mode 1,
 *                     w = ok ? 1 : 0 for weight_mode 0.
 *  corr_good        : como/odom/frontend/corr.py:47-59, 113-118 (modes logz / logr) -- the z components (element 2 of rows of
 *                     `stride` doubles: pass the address of the first z) of four point sets and the depth-gradient measure:
 *                     good = max(|log a - log b|, |log c - log d|) < corr_thresh & grad < grad_thresh.
 *  normalize_coords : como/utils/coords.py:12-15 -- out = A2[k] x + A[k] - 1, k = coordinate index (n2 = 2 x points). */
int como_kf_predictor_sinv_f64(const double* var_n, const uint8_t* row_mask, long n, double* partial, double* sinv,
                               como_stream_t stream);
int como_kf_distill_prep_f64(const double* z_obs, long z_stride, const uint8_t* obs_mask, long n, double min_depth, const double* sinv,
                             double sinv_scalar, const double* stdev_dev, int weight_mode, uint8_t* okm, double* zs, double* y,
                             double* w, como_stream_t stream);
int como_kf_corr_good_f64(const double* a, const double* b, const double* c, const double* d, int stride, const double* grad, long m,
                          double corr_thresh, double grad_thresh, uint8_t* good, como_stream_t stream);
int como_kf_normalize_coords_f32(const float* x, long n2, const float* A, const float* A2, float* out, como_stream_t stream);
int como_kf_normalize_coords_f64(const double* x, long n2, const double* A, const double* A2, double* out, como_stream_t stream);
/* normalize_coords_swap: the same values written with x / y exchanged (swap_coords_xy of the result, coords.py:5-6; out != x);
 * grad_mag: sqrt(gx^2 + gy^2) (corr.py:95-96); aff: affine brightness parameters (B,2) -- mode 0 get_aff_w_curr(p, q) =
 * (p0 + q0, p1 + q1 exp(q0)), mode 1 get_rel_aff(p, q) = (p0 - q0, exp(-(p0 - q0)) (p1 - q1)) (geometry/affine_brightness.py:5-16). */
int como_kf_normalize_coords_swap_f32(const float* x, long n2, const float* A, const float* A2, float* out, como_stream_t stream);
int como_kf_normalize_coords_swap_f64(const double* x, long n2, const double* A, const double* A2, double* out, como_stream_t stream);
int como_kf_grad_mag_f32(const float* gx, const float* gy, long n, float* out, como_stream_t stream);
int como_kf_grad_mag_f64(const double* gx, const double* gy, long n, double* out, como_stream_t stream);
int como_kf_aff_f32(const float* p, const float* q, int B, int mode, float* out, como_stream_t stream);
int como_kf_aff_f64(const double* p, const double* q, int B, int mode, double* out, como_stream_t stream);
/* The small system of the conditional distillation (distill_depth.py:122-148, normal-equation form): cond_c: c (mp) = [log z1 (m1) ; 0];
 * cond_system: A22 (m2,m2) = AtA[m1:m1+m2, m1:m1+m2] + sp2 I (AtA row stride ld), b2 (m2) = Atb[m1:m1+m2] + sp2 * s_med[0]. */
int como_kf_cond_c_f64(const double* z1, int m1, int mp, double* c, como_stream_t stream);
/* masked_std: torch.std (unbiased) of the entries of res (n) whose okm byte is set -- the spread of a distillation's valid residuals
 * (distill_depth.py:113-116) -- into out[0], one launch, fixed summation order (not torch's: equal up to that rounding). */
int como_kf_masked_std_f64(const double* res, const uint8_t* okm, long n, double* out, como_stream_t stream);
int como_kf_cond_system_f64(const double* AtA, const double* Atb, int ld, int m1, int m2, double sp2, const double* s_med,
                            double* A22, double* b2, como_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COMO_HIP_H */
