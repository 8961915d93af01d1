"""Oracle: one full window-BA Gauss-Newton iteration = the call sequence of reference Mapping.iterate
(como/odom/Mapping.py:760-968) assembled from the oracle pieces.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import torch

from . import dense_ref as odr
from . import photo_ba as oba
from . import priors as opr


class OracleWindow:
    def __init__(self, st, window=4, window_full=True):
        c = {k: (v.detach().cpu().double() if torch.is_tensor(v) and v.is_floating_point() else (v.cpu() if torch.is_tensor(v) else v))
             for k, v in st.items()}
        self.c = c
        self.K = c["intrinsics"][0]
        self.B = c["kf_poses"].shape[0]
        self.m = c["coords_m"].shape[1]
        self.poses, self.aff, self.P = c["kf_poses"].clone(), c["kf_aff_params"].clone(), c["P_m"].clone()
        self.corr = c["correspondence_mask"]
        self.ids = torch.stack([torch.nonzero(self.corr[k])[:, 0] for k in range(self.B)])
        L = self.P.shape[0]
        self.D = 8 * self.B + 3 * L
        self.kf_inds = torch.arange(8 * self.B).reshape(self.B, 8)
        self.lm_start = 8 * self.B
        self.lm = (3 * self.ids.repeat_interleave(3, dim=1) + torch.arange(3).repeat(self.m)[None]) + self.lm_start
        self.cn = odr.subselect_pixels(c["kf_img_and_grads"], window)
        bi = torch.arange(self.B)[:, None].expand(-1, self.cn.shape[1])
        self.Kt_rows = c["Knm_Kmminv"][bi, self.cn[..., 0], self.cn[..., 1], :]
        nch = c["kf_img_and_grads"].shape[1] // 3                                 # image channels (gray 1, rgb 3)
        self.vals = c["kf_img_and_grads"][:, :nch].permute(0, 2, 3, 1)[bi, self.cn[..., 0], self.cn[..., 1]]   # (B,n,c), Mapping.py:677-681
        self.med = c["median_depth_init"].clone() if "median_depth_init" in c else torch.full((self.B,), 1.0, dtype=torch.float64)
        self.pose_anchor = c.get("pose_anchor", self.poses[0:1].clone())
        self.P_anchor = c.get("P_anchor", self.P[self.corr[0]].clone())
        self.window_full = window_full

    def iterate(self):
        c, K, B = self.c, self.K, self.B
        Pb = self.P[self.ids]
        # re-initialisation point of every landmark: its first-observation pixel back-projected from its FIRST observer at
        # that keyframe's (full-image) median depth (Mapping.py:612-634); used where z < 0.1 median (sparse_map.py:26-41)
        L = self.P.shape[0]
        first = torch.argmax(self.corr.int(), dim=0)
        slot = (self.ids[first] == torch.arange(L)[:, None]).int().argmax(dim=1)
        px = c["pm_first_obs"][first, slot]                                       # (L,2) (x, y)
        ray = torch.stack(((px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1], torch.ones(L, dtype=torch.float64)), -1)
        Tf = self.poses[first]
        init_P = torch.einsum("lij,lj->li", Tf[:, :3, :3], ray * self.med[first][:, None]) + Tf[:, :3, 3]
        pm, logz, zmask, dlogz_dz, dz_dPw, dz_dTwc, dp_dPw, dp_dTwc = odr.project_landmarks(self.poses, Pb, K, init_P[self.ids],
                                                                                            self.med)
        moved = zmask[first, slot]                                                # re-initialised by the first observer: for good
        self.P = torch.where(moved[:, None], init_P, self.P)                      # Mapping.py:645-648
        self.zmask, self.moved = zmask, moved
        dlogz_dT = dlogz_dz @ dz_dTwc
        dlogz_dP = dlogz_dz @ dz_dPw
        Pw, dT, dz, med_sub, _ = odr.dense_reference(logz, self.poses, self.Kt_rows, self.cn, K, dlogz_dT, dlogz_dz)
        self.med_subset = med_sub                                                 # setup_test_points: pair graph only
        # Mapping.store_vars (Mapping.py:749-758): median of the full depth image -> priors, next iteration's re-init
        Kt = c["Knm_Kmminv"].reshape(B, -1, self.m)
        med = torch.median(torch.exp(Kt @ logz.reshape(B, self.m, 1))[..., 0], dim=1).values
        self.med = med
        H = torch.zeros((self.D, self.D), dtype=torch.float64)
        g = torch.zeros(self.D, dtype=torch.float64)
        ref, tgt = oba.consecutive_pairs(B)
        rid, tid = torch.tensor(ref), torch.tensor(tgt)
        err, aux = oba.batch_photo_cost(self.vals[rid], self.aff[rid], Pw[rid], self.poses[tid], self.aff[tid],
                                        c["kf_img_and_grads"][tid], dT[rid], dz[rid], dz_dPw[rid], self.kf_inds[rid],
                                        self.kf_inds[tid], self.lm[rid], K, H, g, return_aux=True)
        self.aux = aux
        # the arguments batch_photo_cost just received (photo.py:83-99), for tests that drive another implementation with them
        self.last_photo_args = (self.vals[rid], self.aff[rid], Pw[rid], self.poses[tid], self.aff[tid], c["kf_img_and_grads"][tid],
                                dT[rid], dz[rid], dz_dPw[rid], self.kf_inds[rid], self.kf_inds[tid], self.lm[rid], K)
        kpi = self.kf_inds[:, :6]
        log_med = torch.log(med)[:, None, None]
        opr.gp_ml_cost(logz, log_med, c["L_mm"], dlogz_dP, dlogz_dT, self.lm, kpi, H, g, 1.0)
        opr.log_depth_prior_first_mean(logz, log_med, dlogz_dP, dlogz_dT, c["obs_ref_mask"], self.lm, kpi, H, g, 1.0)
        opr.pixel_prior_first(pm, c["pm_first_obs"], dp_dPw, dp_dTwc, c["obs_ref_mask"], self.lm, kpi, H, g, 1e-2)
        opr.pose_prior(self.poses[0:1], self.pose_anchor, H, g, 0, 1e-6)
        opr.scalar_prior(self.aff[0, 0], torch.zeros(1, dtype=torch.float64), H, g, torch.tensor([6]), 1e-4)
        opr.scalar_prior(self.aff[0, 1], torch.zeros(1, dtype=torch.float64), H, g, torch.tensor([7]), 1e-4)
        if self.window_full:
            fix = self.corr[0]
            L = self.P.shape[0]
            inds = (torch.arange(3 * L).reshape(L, 3) + self.lm_start)[fix].reshape(-1)
            opr.scalar_prior(self.P[fix].reshape(-1), self.P_anchor.reshape(-1), H, g, inds, 1e-4)
        self.H, self.g = H, g
        delta, info = oba.solve_system(H, g)
        self.info = int(info)
        self.poses, self.aff, _, _, self.P = oba.update_vars(delta, self.poses, self.aff, self.kf_inds, torch.empty(0, 4, 4),
                                                             torch.empty(0, 2, 1), torch.empty(0, dtype=torch.long), self.P,
                                                             self.lm_start)
        return delta
