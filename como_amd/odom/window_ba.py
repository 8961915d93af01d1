"""Headless keyframe-window bundle adjustment: the call sequence of the reference's `Mapping.iterate`
(como/odom/Mapping.py:760-968) without the GUI / process plumbing around it.

    scaffold (project landmarks)          Mapping.prep_geometry_scaffold  :603-659   small torch ops
    dense reference points                Mapping.prep_dense_ref          :661-699   HIP (factored, no (B,n,3,m) tensor)
    photometric normal equations          create_photo_system             backend/photo.py:236-353   HIP (csrc/ba.hip)
    priors                                Mapping.iterate                 :809-917   small torch ops
    solve + update                        lin_sys.solve_system/update_vars  linear_system.py:101-152

State tensors live on one GPU.  `pix_dtype` is the element type of the per-pixel path (images, K~, dense
points: float32 for the mixed-precision configuration, float64 to mirror config/como.yml:28); the system
(H, g, poses, landmarks, priors, solve) is always float64.
"""
import torch

import como_amd.odom.backend.linear_system as lin_sys
import como_amd.odom.backend.photo as photo
import como_amd.odom.backend.sparse_map as smap
from como_amd.odom.backend.graph_pair_construction import setup_photometric_pairs
from como_amd.odom.factors.depth_prior import log_depth_prior
from como_amd.odom.factors.gp_priors import gp_ml_cost, mean_log_depth_cost
from como_amd.odom.factors.pixel_prior import pixel_prior_cost
from como_amd.odom.factors.pose_prior_factors import linearize_pose_prior
from como_amd.odom.factors.scalar_prior_factors import linearize_multi_scalar_prior, linearize_scalar_prior
from como_amd.geometry.camera import backprojection

DEFAULT_CFG = {
    "photo_construction": {"nonmax_suppression_window": 4, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                           "degrees_thresh": 0.0},
    "sigmas": {"photo": 1e-1, "mean_depth_prior": 1e-2, "scale_prior": 1e-4, "pose_prior": 1e-6},
}


class WindowBA:
    def __init__(self, state, cfg=None, pix_dtype=torch.float32, window_full=True, dense_ref="hip", shard=None):
        """state: dict as produced by como_amd.synth.make_window (plus K_mm_inv / L_mm / Knm_Kmminv).
        shard: como_amd.dist.Shard for the one-process-per-GPU data-parallel mode (None = single GPU)."""
        self.shard = shard
        self.events = None
        self.cfg = cfg or DEFAULT_CFG
        self.dev = state["kf_poses"].device
        self.dt = torch.float64
        self.pix_dtype = pix_dtype
        f64 = lambda t: t.to(self.dt).contiguous()
        self.intrinsics = f64(state["intrinsics"])
        self.kf_poses = f64(state["kf_poses"])
        self.kf_aff_params = f64(state["kf_aff_params"])
        self.P_m = f64(state["P_m"])
        self.correspondence_mask = state["correspondence_mask"]
        self.obs_ref_mask = state["obs_ref_mask"]
        self.pm_first_obs = f64(state["pm_first_obs"])
        self.L_mm = f64(state["L_mm"])
        self.kf_timestamps = state["kf_timestamps"]
        self.recent_poses = torch.empty((0, 4, 4), device=self.dev, dtype=self.dt)
        self.recent_aff_params = torch.empty((0, 2, 1), device=self.dev, dtype=self.dt)
        self.recent_timestamps = torch.empty((0,), device=self.dev, dtype=self.dt)
        B, _, self.Himg, self.Wimg = state["kf_img_and_grads"].shape
        self.B = B
        self.m = state["coords_m"].shape[1]
        # per-pixel data in pix_dtype
        self.img = state["kf_img_and_grads"].to(pix_dtype).contiguous()
        self.Kt = state["Knm_Kmminv"].to(pix_dtype).reshape(B, self.Himg * self.Wimg, self.m).contiguous()
        self.K_pix = self.intrinsics[0].to(pix_dtype).contiguous()
        self.median_depths = f64(state["median_depth_init"]) if "median_depth_init" in state else torch.full((B,), 1.0, device=self.dev, dtype=self.dt)
        self.window_full = window_full
        self.pose_anchor = self.kf_poses[0:1].clone() if "pose_anchor" not in state else f64(state["pose_anchor"])
        self.aff_anchor = torch.zeros((1, 2, 1), device=self.dev, dtype=self.dt)
        self.P_m_anchors = f64(state["P_anchor"]) if "P_anchor" in state else self.P_m[self.correspondence_mask[0]].clone()
        self.init_scale_anchor = state.get("init_scale_anchor")
        self.dense_ref = dense_ref
        self._prepare_topology()

    # ---- things that change only when the keyframe set changes ---------------------------------------------------
    def _prepare_topology(self):
        B, dev = self.B, self.dev
        w = self.cfg["photo_construction"]["nonmax_suppression_window"]
        coords_n, _ = smap.subselect_pixels(self.img, w)                       # Mapping.py:665-668
        self.coords_n = coords_n
        self.n = coords_n.shape[1]
        self.pixidx = (coords_n[..., 0] * self.Wimg + coords_n[..., 1]).to(torch.int32).contiguous()
        self.vals_n = torch.gather(self.img[:, 0].reshape(B, -1), 1, self.pixidx.long()).contiguous()
        self.remap, paired = smap.get_batch_remap_function(self.correspondence_mask)
        landmark_ids, _ = paired
        self.point_inds = lin_sys.landmark_to_batched_3d_point_inds(landmark_ids, B)
        L = self.P_m.shape[0]
        nrec = self.recent_poses.shape[0]
        self.dim = 8 * B + 8 * nrec + 3 * L
        self.kf_inds = torch.arange(8 * B, device=dev).reshape(B, 8)
        self.recent_inds = (torch.arange(8 * nrec, device=dev).reshape(nrec, 8) + 8 * B) if nrec else \
            torch.empty((0), device=dev, dtype=torch.long)
        self.lm_start = 8 * B + 8 * nrec
        self.landmark_inds = self.point_inds + self.lm_start
        self.landmark_inds_flat = torch.arange(3 * L, device=dev).reshape(L, 3) + self.lm_start
        # index lists of the oldest keyframe's landmarks (their anchors, Mapping.py:884-898): precomputed -- boolean-mask
        # indexing inside the iteration would synchronise with the host (and cannot be captured in a hipGraph)
        self.fix_idx = torch.nonzero(self.correspondence_mask[0])[:, 0]
        self.fix_inds_flat = self.landmark_inds_flat[self.fix_idx].flatten().contiguous()
        first_obs = torch.argmax(self.correspondence_mask.int(), dim=0)
        fom = torch.zeros_like(self.correspondence_mask)
        fom[first_obs, torch.arange(L, device=dev)] = True
        self.first_obs_mask = self.remap(fom, False)
        ref, tgt, ow_kf, ow_t = setup_photometric_pairs(self.kf_poses, self.recent_poses, self.kf_timestamps,
                                                        self.recent_timestamps, self.median_depths,
                                                        self.cfg["photo_construction"])
        self.kf_pairs, self.one_way_pairs = [ref, tgt], [ow_kf, ow_t]
        self.table = photo.PairTable(ref + ow_kf, tgt + ow_t, [False] * len(ref) + [True] * len(ow_kf), B, self.kf_inds,
                                     self.recent_inds, self.landmark_inds, 3 * self.Himg * self.Wimg, 0, dev)
        # H | g | err packed in ONE buffer: the multi-GPU exchange of the normal equations is a single all-reduce
        D = self.dim
        self.sys = torch.zeros((D * D + D + 1,), device=dev, dtype=self.dt)
        self.H = self.sys[:D * D].view(D, D)
        self.g = self.sys[D * D:D * D + D]
        self.err = self.sys[D * D + D:].view(())
        self.pix_range = self.shard.pixel_range(self.n) if self.shard is not None else None
        self.sigma = torch.zeros(2, device=dev, dtype=self.pix_dtype)

    # ---- one Gauss-Newton iteration ------------------------------------------------------------------------------
    def scaffold(self):
        """Mapping.prep_geometry_scaffold (:603-659) without host synchronisation."""
        K = self.intrinsics
        depth_init = self.median_depths[:, None, None].expand(-1, self.m, 1)
        init_Pc, _ = backprojection(K[0], self.pm_first_obs, depth_init)
        init_Pw = smap.rigid_apply_exact(self.kf_poses, init_Pc)                  # (B,m,3)
        # re-initialisation point of a landmark = back-projection from its FIRST observer at the median depth (:625-634)
        lm_ids = self.point_inds[:, ::3] // 3 - 0                                 # (B,m) landmark id of every batched slot
        lm_ids = (self.point_inds[:, ::3]) // 3
        fom = self.first_obs_mask
        init_Pm = torch.zeros_like(self.P_m)
        init_Pm.index_add_(0, lm_ids.reshape(-1), (init_Pw * fom[..., None]).reshape(-1, 3))
        reinit_b = init_Pm[lm_ids]                                                # (B,m,3), same remap as P_m
        Pwm = self.P_m[lm_ids]
        out = smap.project_landmarks(self.kf_poses, Pwm, K[0], reinit_b, self.median_depths)
        z_mask = out[2]
        # landmarks re-initialised in their first-observation frame are moved for good (Mapping.py:645-648)
        flag = torch.zeros((self.P_m.shape[0], 1), device=self.dev, dtype=self.dt)
        flag.index_add_(0, lm_ids.reshape(-1), (fom & z_mask).reshape(-1, 1).to(self.dt))
        self.P_m.copy_(torch.where(flag > 0, init_Pm, self.P_m))
        return out

    def dense_reference(self, logzm, dlogzm_dTwc):
        if self.dense_ref == "hip":
            from como_amd.odom.backend.dense_ref import dense_reference_factored
            return dense_reference_factored(logzm.to(self.pix_dtype), self.kf_poses.to(self.pix_dtype), self.Kt, self.pixidx,
                                            self.K_pix, dlogzm_dTwc.to(self.pix_dtype), self.Wimg)
        p = self.pix_dtype
        return smap.dense_reference_factored_torch(logzm.to(p), self.kf_poses.to(p), self.Kt, self.pixidx, self.coords_n,
                                                  self.K_pix, dlogzm_dTwc.to(p))

    def linearize(self):
        pm, logzm, z_mask, dlogzm_dzm, dzm_dPwm, dzm_dTwc, dpm_dPwm, dpm_dTwc = self.scaffold()
        dlogzm_dTwc = dlogzm_dzm @ dzm_dTwc
        dlogzm_dPwm = dlogzm_dzm @ dzm_dPwm
        Pwn, dPwn_dTwc, uvec, med, logzn = self.dense_reference(logzm, dlogzm_dTwc)
        self.median_depths.copy_(med)
        self.pm, self.logzm = pm, logzm
        H, g = self.H, self.g
        self.sys.zero_()
        p = self.pix_dtype
        poses_all = torch.cat((self.kf_poses, self.recent_poses)).to(p).contiguous()
        aff_all = torch.cat((self.kf_aff_params, self.recent_aff_params)).reshape(-1, 2).to(p).contiguous()
        photo.photo_system_factored(self.table, poses_all=poses_all, aff_all=aff_all, Pwn=Pwn, vals=self.vals_n,
                                    dPwn_dTwc=dPwn_dTwc, uvec=uvec, Kt=self.Kt, pixidx=self.pixidx,
                                    invz=dlogzm_dzm[:, :, 0, 0].to(p).contiguous(), dzdP=dzm_dPwm[:, 0, 0, :].to(p).contiguous(),
                                    img_base=self.img, K=self.K_pix, H_img=self.Himg, W_img=self.Wimg, H=H, g=g,
                                    err_out=self.err, sigma_out=self.sigma, pix_range=self.pix_range,
                                    reduce_hists=(self.shard.all_reduce_sum if self.shard is not None else None),
                                    events=self.events)
        if self.shard is not None:
            self.shard.all_reduce_sum(self.sys)          # normal equations of all shards: H | g | err in one collective
        kf_pose_inds, kf_aff_inds = self.kf_inds[:, :6], self.kf_inds[:, 6:]
        log_med = torch.log(self.median_depths[:, None, None])
        sg = self.cfg["sigmas"]
        e = [self.err.clone()]
        e.append(gp_ml_cost(logzm, log_med, self.L_mm, dlogzm_dPwm, dlogzm_dTwc, self.landmark_inds, kf_pose_inds, H, g, sigma=1e0))
        e.append(log_depth_prior(logzm, log_med, dlogzm_dPwm, dlogzm_dTwc, self.obs_ref_mask, self.landmark_inds, kf_pose_inds,
                                 H, g, mode="first_mean", sigma_first=1e0, sigma_all=1e-0))
        e.append(pixel_prior_cost(pm, self.pm_first_obs, dpm_dPwm, dpm_dTwc, self.obs_ref_mask, self.landmark_inds, kf_pose_inds,
                                  H, g, mode="first", pixel_sigma_first=1e-2, pixel_sigma_all=3.33e-1))
        e.append(linearize_pose_prior(self.kf_poses[0:1], self.pose_anchor[0:1], H, g, [0, 6], sigma=sg["pose_prior"]))
        e.append(linearize_scalar_prior(self.kf_aff_params[0, 0:1, :], self.aff_anchor[0, 0:1, :], H, g, [6, 7], sigma=sg["scale_prior"]))
        e.append(linearize_scalar_prior(self.kf_aff_params[0, 1:2, :], self.aff_anchor[0, 1:2, :], H, g, [7, 8], sigma=sg["scale_prior"]))
        if self.window_full:
            e.append(linearize_multi_scalar_prior(self.P_m[self.fix_idx].flatten(), self.P_m_anchors.flatten(), H, g,
                                                  self.fix_inds_flat, sigma=sg["scale_prior"]))
        else:
            e.append(mean_log_depth_cost(logzm[0:1], self.Kt[0:1].to(self.dt), self.init_scale_anchor, dlogzm_dPwm[0:1],
                                         dlogzm_dTwc[0:1], self.landmark_inds[0:1], kf_pose_inds[0:1], H, g,
                                         sg["mean_depth_prior"]))
        self.total_err = sum(x.reshape(()).to(torch.float64) for x in e)
        return H, g

    def iterate(self):
        """One GN iteration, eager.  State tensors are updated IN PLACE (fixed addresses -> capturable)."""
        H, g = self.linearize()
        delta = self.solve(H, g)
        kp, ka, rp, ra, Pn = lin_sys.update_vars(delta, self.kf_poses, self.kf_aff_params, self.kf_inds, self.recent_poses,
                                                 self.recent_aff_params, self.recent_inds, self.P_m, self.lm_start)
        self.kf_poses.copy_(kp)
        self.kf_aff_params.copy_(ka)
        self.P_m.copy_(Pn)
        if self.recent_poses.shape[0] > 0:
            self.recent_poses.copy_(rp)
            self.recent_aff_params.copy_(ra)
        self.delta = delta
        return delta

    def solve(self, H, g):
        return lin_sys.solve_system(H, g)

    # ---- hipGraph: the iteration is ~200 small launches; replaying a captured graph removes the host from the loop
    def capture(self, warmup=3):
        """Capture one GN iteration into a hipGraph (torch.cuda.CUDAGraph).  Returns True on success; on failure the
        object stays usable in eager mode."""
        if self.shard is not None:
            return False                      # collectives between kernels: keep the multi-GPU path eager
        self.graph = None
        try:
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.iterate()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            ev = self.events
            self.events = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.iterate()
            self.events = ev
            self.graph = g
            return True
        except Exception as e:   # noqa: BLE001
            self.graph = None
            import traceback
            self.capture_error = traceback.format_exc()[-1500:]
            try:
                torch.cuda.synchronize(self.dev)
            except Exception:   # noqa: BLE001
                pass
            return False

    def step(self):
        """One GN iteration: graph replay when captured, eager otherwise."""
        if getattr(self, "graph", None) is not None:
            self.graph.replay()
            return self.delta
        return self.iterate()
