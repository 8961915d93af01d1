"""Headless frame tracker: the state machine of the reference's `Tracking` (como/odom/Tracking.py:21-379) without the process
/ queue plumbing.  Same method names, arguments and return tuples, so the caller code of the reference (Odometry / the
multiprocessing wrappers) can drive it unchanged.

Per frame (`handle_frame`): image pyramid -> `photo_tracking_pyr` (HIP, csrc/track.hip, one captured graph per pyramid level)
-> reprojection of the last keyframe's points -> keyframe / one-way-frame decision.  Per new keyframe
(`update_kf_reference`): pyramids of the keyframe image(s) and depth(s), back-projection into the last keyframe's frame and
the inverse-compositional Jacobians (`precalc_jacobians`, HIP, csrc/image.hip).  The O(N) glue between the kernels is torch
elementwise work on the device.
"""
import torch

from como_amd.geometry.affine_brightness import get_aff_w_curr, get_rel_aff
from como_amd.geometry.camera import backprojection
from como_amd.geometry.lie_algebra import composeSE3, invertSE3  # noqa: F401
from como_amd.geometry.transforms import get_rel_pose, get_T_w_curr
import os

import numpy as np

import como_amd.odom.frontend.photo_tracking as _pt
from como_amd.odom.frontend.photo_tracking import photo_tracking_pyr, precalc_jacobians
from como_amd.utils.coords import fill_image, get_test_coords, swap_coords_xy
from como_amd import _lib
from como_amd.utils.select import masked_median
from como_amd.utils.image_processing import (DepthPyramidModule, ImageGradientModule, ImagePyramidModule,
                                             IntrinsicsPyramidModule, rgb_to_grayscale)

_DTYPES = {"float": torch.float32, "double": torch.float64, "half": torch.float16}


def _rigid(T, P):
    """R P + t for poses (B,4,4) and points (b,n,3), b in {1, B} -> (B,n,3), as ONE small GEMM per pose.  (transform_points
    also builds the (n,3,6) Jacobian through a batched product over n tiny matrices: milliseconds at 640x480, unused here.)"""
    return P @ T[:, :3, :3].transpose(-1, -2) + T[:, None, :3, 3]


def _project(K, P):
    """Pinhole projection, values only, in the reference's operation order (camera.py:20-26: f X / Z + c)."""
    return torch.stack((K[0, 0] * P[..., 0] / P[..., 2] + K[0, 2], K[1, 1] * P[..., 1] / P[..., 2] + K[1, 2]), dim=-1)


def _in_image(p, depth, img_size, border, depth_thresh, strict):
    """Projected pixel p (.., 2) as (x, y) inside the image grown by `border`, and depth above the threshold."""
    x, y = p[..., 0], p[..., 1]
    if strict:                      # Tracking.py:173-178 (open interval)
        ok = (x > -border) & (x < img_size[-1] - 1 + border) & (y > -border) & (y < img_size[-2] - 1 + border)
    else:                           # Tracking.py:265-281 (closed interval)
        ok = (x >= -border) & (x <= img_size[-1] - 1 + border) & (y >= -border) & (y <= img_size[-2] - 1 + border)
    return ok & (depth[..., 0] > depth_thresh)


_PYR_REFERENCE = os.environ.get("COMO_TRACK_PYR_REFERENCE", "1") != "0"    # 0: pose composition + depth pyramid + one reference launch per level (A/B)
_FRAME_RECORD = os.environ.get("COMO_TRACK_FRAME_RECORD", "1") != "0"      # 0: the frame's record by pose composition + norm + casts + cat (A/B)
_FOREACH_INPUTS = os.environ.get("COMO_TRACK_FOREACH_INPUTS", "1") != "0"   # 0: one copy launch per input of the frame graph (A/B)
_LEAN_HEAD = os.environ.get("COMO_TRACK_LEAN_HEAD", "1") != "0"      # 0: gray + 2 x blur_down + one workspace clear per level / per median (A/B)
_DIRECT_REF = os.environ.get("COMO_TRACK_DIRECT_REF", "1") != "0"     # (read once: an os.environ look-up costs ~25 us, this ran per frame)


class Tracking:
    def __init__(self, cfg, intrinsics, img_size):
        self.cfg = cfg
        self.device = cfg["device"]
        self.dtype = _DTYPES[cfg["dtype"]] if isinstance(cfg["dtype"], str) else cfg["dtype"]
        self.intrinsics = intrinsics
        self.img_size = img_size
        self.mapping_init = False

    def track(self, data):
        raise NotImplementedError

    def setup(self):
        self.init_basic_vars()
        self.init_kf_vars()
        self.reset_one_way_vars()
        self.T_w_rec_last = None

    # ---- construction ----------------------------------------------------------------------------------------------
    def init_basic_vars(self):
        """Tracking.py:46-72"""
        self.intrinsics = self.intrinsics.to(device=self.device, dtype=self.dtype)
        pyr = self.cfg["pyr"]
        s, e = pyr["start_level"], pyr["end_level"]
        self.intrinsics_pyr = IntrinsicsPyramidModule(s, e, self.device)(self.intrinsics, [1.0, 1.0])
        c = {"gray": 1, "rgb": 3}[self.cfg["color"]]
        self.gradient_module = ImageGradientModule(channels=c, device=self.device, dtype=self.dtype)
        self.img_pyr_module = ImagePyramidModule(c, s, e, self.device, dtype=self.dtype)
        self.depth_pyr_module = DepthPyramidModule(s, e, pyr["depth_interp_mode"], self.device)

    def reset_one_way_vars(self):
        self.num_one_way_since_kf = 0
        self.last_one_way_empty_pixels = 0
        self.last_flow_rmse = 0.0
        self.last_flow_wo_rot_rmse = 0.0

    def init_kf_vars(self):
        self.T_curr_kf = torch.eye(4, device=self.device, dtype=self.dtype).unsqueeze(0)
        self.aff_curr_kf = torch.zeros((1, 2, 1), device=self.device, dtype=self.dtype)
        self.last_one_way_num_pixels = self.img_size[-1] * self.img_size[-2]
        self.last_kf_sent_ts = torch.zeros(1, device=self.device, dtype=self.dtype)
        self.kf_received_ts = torch.zeros(1, device=self.device, dtype=self.dtype)

    # ---- small accessors -------------------------------------------------------------------------------------------
    def get_curr_world_pose(self):
        return get_T_w_curr(self.T_w_kf, self.T_curr_kf)

    def get_curr_world_aff(self):
        return get_aff_w_curr(self.aff_w_kf, self.aff_curr_kf)

    def prep_tracking_img(self, rgb):
        img = rgb_to_grayscale(rgb) if self.cfg["color"] == "gray" else rgb.clone()
        return self.img_pyr_module(img)

    def get_img_gradients(self, img_pyr):
        out = []
        for lvl in img_pyr:
            gx, gy = self.gradient_module(lvl)
            out.append(torch.cat((lvl, gx, gy), dim=1))
        return out

    # ---- keyframe / one-way decisions (Tracking.py:110-161) ----------------------------------------------------------
    def check_keyframe(self, median_depth, num_reproj_depth, T_curr_kf):
        if not bool(self.last_kf_sent_ts <= self.kf_received_ts):
            return False                                   # a keyframe request is still in flight
        kfg = self.cfg["keyframing"]
        n_px = self.vals_pyr[-1].shape[1]
        if torch.linalg.norm(T_curr_kf[:, :3, 3]) > kfg["kf_depth_motion_ratio"] * median_depth:
            return True
        return bool(kfg["kf_num_pixels_frac"] > num_reproj_depth / n_px)

    def check_one_way_frame(self, median_depth, num_reproj_depth, T_curr_kf, T_w_curr):
        kfg = self.cfg["keyframing"]
        pending = 1 if bool(self.last_kf_sent_ts > self.kf_received_ts) else 0
        scale = (1.0 + self.num_one_way_since_kf + pending) / (1.0 + kfg["one_way_freq"])
        n_px = self.vals_pyr[-1].shape[1]
        empty = n_px - num_reproj_depth
        moved = torch.linalg.norm(T_curr_kf[:, :3, 3]) > scale * kfg["kf_depth_motion_ratio"] * median_depth
        new = bool(moved) or bool(empty > scale * (1 - kfg["kf_num_pixels_frac"]) * n_px)
        if new:
            self.last_one_way_empty_pixels = empty
            self.T_w_rec_last = T_w_curr
        return new

    def reproj_stats_last_kf(self, T_curr_kf, P=None, clone_count=True, hist_prezeroed=False):
        """(reprojected depth image (1,h,w), seen mask, number of pixels seen, their exact median depth) of the newest keyframe's
        finest-level points in the current frame (Tracking.py:163-185 get_reproj_last_kf + :341-345): two launches
        (csrc/trackref.hip `como_reproject_depth_*`) + the device select -- no boolean-mask gathers, no host synchronisation."""
        if P is None:
            P = self.P_pyr[-1][-1]                                               # (n,3) of the newest keyframe
        if not P.is_cuda:
            reproj = self.get_reproj_last_kf(T_curr_kf)
            seen = ~torch.isnan(reproj)
            return reproj, seen, torch.count_nonzero(seen), torch.median(reproj[seen])
        h, w = int(self.img_size[-2]), int(self.img_size[-1])
        n, dt, dev = P.shape[0], P.dtype, P.device
        ws = getattr(self, "_reproj_ws", None)
        if ws is None or ws["key"] != (n, h, w, dt):
            ws = self._reproj_ws = {"key": (n, h, w, dt), "order": torch.zeros(h * w, dtype=torch.int64, device=dev),
                                    "z": torch.empty(n, dtype=dt, device=dev), "img": torch.empty(h * w, dtype=dt, device=dev),
                                    "seen": torch.empty(h * w, dtype=torch.uint8, device=dev),
                                    "nseen": torch.zeros(1, dtype=torch.int32, device=dev)}
        fn = getattr(_lib.lib(), "como_reproject_depth_" + _lib.suffix(dt))
        _lib.check(fn(T_curr_kf.reshape(4, 4).contiguous().data_ptr(), self.intrinsics_pyr[-1].contiguous().data_ptr(),
                      P.contiguous().data_ptr(), n, h, w, ws["order"].data_ptr(), ws["z"].data_ptr(), ws["img"].data_ptr(),
                      ws["seen"].data_ptr(), ws["nseen"].data_ptr(), _lib.stream_ptr(dev)), "como_reproject_depth")
        med = masked_median(ws["img"], ws["seen"], prezeroed=hist_prezeroed)
        return ws["img"].view(1, h, w), ws["seen"].view(1, h, w), (ws["nseen"][0].clone() if clone_count else ws["nseen"][0]), med

    def get_reproj_last_kf(self, T_curr_kf):
        """Depth image of the newest keyframe's finest-level points seen from the current frame, NaN where nothing lands
        (Tracking.py:163-185)."""
        P_curr = _rigid(T_curr_kf, self.P_pyr[-1][None, -1, :, :])
        p = _project(self.intrinsics_pyr[-1], P_curr)
        depth = P_curr[:, :, 2:3]
        ok = _in_image(p, depth, self.img_size, 0, 0.0, strict=True)
        return fill_image(swap_coords_xy(p)[ok, :], depth[ok, :], self.img_size)

    def _kf_images(self, kf_rgb):
        """Intensities, gradients and pixel coordinates of the keyframe image(s) at every pyramid level (Tracking.py:203-222)."""
        coords_pyr, vals_pyr, img_grads_pyr = [], [], []
        for lvl in self.prep_tracking_img(kf_rgb):
            gx, gy = self.gradient_module(lvl)
            b, c, h, w = lvl.shape
            flat = lambda t: t.reshape(b, c, h * w).permute(0, 2, 1)       # (B,N,C), row-major pixel order
            vals_pyr.append(flat(lvl).contiguous())
            img_grads_pyr.append(torch.stack((flat(gx), flat(gy)), dim=-1).contiguous())
            coords_pyr.append(get_test_coords((h, w), device=self.device, batch_size=b))
        return coords_pyr, vals_pyr, img_grads_pyr

    def prepare_kf_images(self, timestamp, rgb):
        """The image half of `update_kf_reference` for the frame that was just sent to the mapper as a keyframe (one reference
        keyframe: `track_ref.num_keyframes: 1`), AHEAD of the mapper's answer: it only depends on the frame itself.  The sequential
        loop calls it while the covariance network of the insertion runs (`Mapping.while_network_runs`): ~30 torch calls that
        otherwise sit in the host-bound tail of a keyframe frame, behind the window's first iteration, with the device idle.
        `update_kf_reference` adopts the result when timestamp, shape and element type are those of the image it receives."""
        rgb = rgb.to(self.dtype) if rgb.dtype != self.dtype else rgb
        self._kf_img_ahead = (timestamp, tuple(rgb.shape), rgb.dtype, self._kf_images(rgb))

    # ---- new reference keyframe(s) from mapping (Tracking.py:187-313) ------------------------------------------------
    def update_kf_reference(self, kf_data):
        timestamps, kf_rgb, kf_pose, kf_aff, depth = kf_data
        nk = kf_pose.shape[0]
        if timestamps[-1] > self.kf_received_ts and self.mapping_init:
            # re-base the current frame's relative pose / affine parameters on the new last keyframe
            self.T_w_f = get_T_w_curr(self.T_w_kf, self.T_curr_kf)
            self.T_curr_kf = get_rel_pose(self.T_w_f, kf_pose[nk - 1:nk])
            self.aff_w_f = get_aff_w_curr(self.aff_w_kf, self.aff_curr_kf)
            self.aff_curr_kf = get_rel_aff(self.aff_w_f, kf_aff[nk - 1:nk])
            self.reset_one_way_vars()
        elif not self.mapping_init:
            self.mapping_init = True
            self.last_kf_sent_ts = timestamps[-1]

        if timestamps[-1] != self.kf_received_ts:          # new image(s): intensities and gradients at every pixel
            ah, self._kf_img_ahead = getattr(self, "_kf_img_ahead", None), None
            if (ah is not None and ah[0] == timestamps[-1] and nk == 1 and kf_rgb is not None and tuple(kf_rgb.shape) == ah[1] and
                    kf_rgb.dtype == ah[2]):
                self.coords_pyr, self.vals_pyr, self.img_grads_pyr = ah[3]     # (`prepare_kf_images`: the same calls on the same frame)
            else:
                self.coords_pyr, self.vals_pyr, self.img_grads_pyr = self._kf_images(kf_rgb)

        self.P_pyr, self.dI_dT_pyr, self.mask_pyr = [], [], []
        pb = None
        if (depth.is_cuda and self.vals_pyr[0].shape[2] == 1 and depth.dtype == self.vals_pyr[0].dtype and
                _DIRECT_REF):
            # this tracker's own persistent reference buffers (the level kernels and the captured frame graph read them): the
            # reference kernels below write straight into them -- no per-update allocations, no copies before the next frame
            sizes = self._depth_pyramid_sizes(tuple(depth.shape[-2:]))
            key = (nk, depth.dtype, sizes)
            if getattr(self, "_pb_key", None) != key:
                self._pb = _pt._PyrBuffers.from_shapes(nk, 1, list(sizes), depth.device, depth.dtype)
                self._pb_key, self._pb_vals_ts = key, None
            pb = self._pb
            if self._pb_vals_ts != timestamps[-1]:          # the keyframe image(s) changed
                for i, c in enumerate(pb.levels):
                    c["vals"].copy_(self.vals_pyr[i].reshape(c["vals"].shape))
                    self.vals_pyr[i] = c["vals"].view(self.vals_pyr[i].shape)
                self._pb_vals_ts = timestamps[-1]
        if self._reference_pyramid_in_one_launch(kf_pose, depth, pb):
            self.kf_received_ts = timestamps[-1]
            self.T_w_kf = kf_pose[nk - 1:nk]
            self.aff_w_kf = kf_aff[nk - 1:nk]
            return
        rel = composeSE3(kf_pose[nk - 1:nk], kf_pose, 1)    # every keyframe -> the last keyframe's frame
        depth_pyr = self.depth_pyr_module(depth)
        for i, d in enumerate(depth_pyr):
            coords = self.coords_pyr[i]
            b, _, h, w = d.shape
            if d.is_cuda and self.vals_pyr[i].shape[2] == 1:
                # one launch per level (csrc/trackref.hip): back-projection, transform, projection mask, Jacobians
                dt, dev = d.dtype, d.device
                if pb is not None:
                    c = pb.levels[i]
                    P_all, mask, J = c["P"].view(b, h * w, 3), c["mask"].view(b, h * w), c["dI"].view(b, h * w, 1, 8)
                else:
                    P_all = torch.empty((b, h * w, 3), dtype=dt, device=dev)
                    mask = torch.empty((b, h * w), dtype=torch.uint8, device=dev)
                    J = torch.empty((b, h * w, 1, 8), dtype=dt, device=dev)
                fn = getattr(_lib.lib(), "como_track_reference_" + _lib.suffix(dt))
                _lib.check(fn(d.contiguous().data_ptr(), rel.to(dt).contiguous().data_ptr(), self.intrinsics_pyr[i].to(dt).contiguous().data_ptr(),
                              self.img_grads_pyr[i].to(dt).contiguous().data_ptr(), self.vals_pyr[i].to(dt).contiguous().data_ptr(), b, h, w,
                              50.0, 1e-4, P_all.data_ptr(), mask.data_ptr(), J.data_ptr(), _lib.stream_ptr(dev)), "como_track_reference")
                self.mask_pyr.append(mask.view(torch.bool))
                self.dI_dT_pyr.append(J)
                self.P_pyr.append(P_all)
                continue
            z = d[:, 0].reshape(b, h * w, 1)
            P, _ = backprojection(self.intrinsics_pyr[i], swap_coords_xy(coords), z)
            P_all = _rigid(rel, P)
            p_all = _project(self.intrinsics_pyr[i], P_all)
            self.mask_pyr.append(_in_image(p_all, P_all[:, :, 2:3], (h, w), 50, 1e-4, strict=False))
            self.dI_dT_pyr.append(precalc_jacobians(self.img_grads_pyr[i], P_all, self.vals_pyr[i], self.intrinsics_pyr[i]))
            self.P_pyr.append(P_all)

        self.kf_received_ts = timestamps[-1]
        self.T_w_kf = kf_pose[nk - 1:nk]
        self.aff_w_kf = kf_aff[nk - 1:nk]

    def _depth_pyramid_sizes(self, hw):
        """Sizes (h, w) of `depth_pyr_module`'s levels, coarse -> fine, for a finest depth image of size hw (pyr_depth with
        kernel_size 2: nearest_neighbor keeps (n + 1) // 2 samples, the pooling modes n // 2)."""
        lo, hi = self.cfg["pyr"]["start_level"], self.cfg["pyr"]["end_level"]
        nn = self.cfg["pyr"]["depth_interp_mode"] == "nearest_neighbor"
        out, (h, w) = [], hw
        for i in range(hi - 1):
            if i >= lo:
                out.insert(0, (h, w))
            h, w = ((h + 1) // 2, (w + 1) // 2) if nn else (h // 2, w // 2)
        out.insert(0, (h, w))
        return tuple(out)

    def _reference_pyramid_in_one_launch(self, kf_pose, depth, pb):
        """The reference arrays of every pyramid level -- points in the last keyframe's frame, projection masks, Jacobians --
        straight from the finest depth image and the keyframe poses in ONE launch (csrc/trackref.hip track_reference_pyr_kernel:
        pose composition, nearest-neighbour depth pyramid and the per-level reference kernels of the loop below).  Applies to the
        tracker's own persistent buffers, gray float32 frames, `depth_interp_mode: nearest_neighbor`, start_level 0, <= 4 levels."""
        import ctypes
        pyr = self.cfg["pyr"]
        nl = len(self.vals_pyr)
        if not (_PYR_REFERENCE and pb is not None and depth.is_cuda and depth.dtype == torch.float32 and kf_pose.dtype == torch.float32 and
                pyr["depth_interp_mode"] == "nearest_neighbor" and pyr["start_level"] == 0 and 1 <= nl <= 4 and len(pb.levels) == nl and
                depth.dim() == 4 and depth.shape[1] == 1 and all(v.shape[2] == 1 for v in self.vals_pyr)):
            return False
        nk, _, H0, W0 = depth.shape
        sizes = self._depth_pyramid_sizes((H0, W0))
        if len(sizes) != nl or any(self.vals_pyr[i].shape[1] != h * w for i, (h, w) in enumerate(sizes)):
            return False
        dt, dev = depth.dtype, depth.device
        grads = [g if (g.dtype == dt and g.is_contiguous()) else g.to(dt).contiguous() for g in self.img_grads_pyr]
        Ks = [k if (k.dtype == dt and k.is_contiguous()) else k.to(dt).contiguous() for k in self.intrinsics_pyr]
        vals = [pb.levels[i]["vals"] for i in range(nl)]
        if any(self.vals_pyr[i].data_ptr() != vals[i].data_ptr() for i in range(nl)):
            return False
        arr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
        hw = (ctypes.c_int * (2 * nl))(*[x for s in sizes for x in s])
        d0, kp = depth.contiguous(), kf_pose.contiguous()
        _lib.check(_lib.lib().como_track_reference_pyr_f32(
            d0.data_ptr(), H0, W0, kp.data_ptr(), nk, nl, hw, arr(Ks), arr(grads), arr(vals), arr([c["P"] for c in pb.levels]),
            arr([c["mask"] for c in pb.levels]), arr([c["dI"] for c in pb.levels]), 50.0, 1e-4, _lib.stream_ptr(dev)),
            "como_track_reference_pyr_f32")
        self.P_pyr, self.dI_dT_pyr, self.mask_pyr = [], [], []
        for i, (h, w) in enumerate(sizes):
            c = pb.levels[i]
            self.P_pyr.append(c["P"].view(nk, h * w, 3))
            self.dI_dT_pyr.append(c["dI"].view(nk, h * w, 1, 8))
            self.mask_pyr.append(c["mask"].view(nk, h * w).view(torch.bool))
        return True

    # ---- the same two tests on host scalars (one read-back per frame instead of four synchronising bool()s) ----------------
    def decide_frame(self, norm_t, median_depth, num_reproj_depth, T_w_curr):
        """check_keyframe / check_one_way_frame (above) evaluated on the three scalars they look at -- |t| of T_curr_kf, the
        median reprojected depth, the number of pixels seen -- in the arithmetic the tensor expressions use: float32 products /
        quotients of float32 operands (a Python scalar meeting a float32 / integer tensor is cast to float32), comparisons in
        float32.  Returns "keyframe", "one-way" or None; the one-way bookkeeping is updated like check_one_way_frame does."""
        f32 = np.float32
        kfg = self.cfg["keyframing"]
        n_px = self.vals_pyr[-1].shape[1]
        norm, md, nseen = f32(norm_t), f32(median_depth), int(num_reproj_depth)
        if nseen == 0:
            # no pixel of the keyframe is seen: there is no median depth (masked_median reports NaN; the reference's
            # torch.median of an empty tensor raises).  The motion tests are skipped explicitly -- the pixel-count tests below
            # then request a keyframe (or, while one is pending, a one-way frame) on their own.
            md = f32(np.inf)
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            if bool(self.last_kf_sent_ts <= self.kf_received_ts):
                if norm > f32(kfg["kf_depth_motion_ratio"]) * md:
                    return "keyframe"
                if f32(kfg["kf_num_pixels_frac"]) > f32(nseen) / f32(n_px):
                    return "keyframe"
            pending = 1 if bool(self.last_kf_sent_ts > self.kf_received_ts) else 0
            scale = (1.0 + self.num_one_way_since_kf + pending) / (1.0 + kfg["one_way_freq"])
            empty = n_px - nseen
            moved = norm > f32(scale * kfg["kf_depth_motion_ratio"]) * md
            new = bool(moved) or bool(f32(empty) > f32(scale * (1 - kfg["kf_num_pixels_frac"]) * n_px))
        if new:
            self.last_one_way_empty_pixels = empty
            self.T_w_rec_last = T_w_curr
            return "one-way"
        return None

    # ---- one frame as ONE hipGraph replay -----------------------------------------------------------------------------------
    # A tracked frame is a fixed sequence of ~100 small launches (gray image + pyramid, three persistent level kernels, world
    # pose, reprojection of the keyframe's points, its exact median, the scalars of the keyframe tests): shapes never change, so
    # the sequence is captured once and replayed on static buffers; the frame's only host synchronisation is the read-back of
    # eight scalars.  Reference arrays go through the persistent pyramid buffers (copied when the mapper sends new ones).
    FRAME_GRAPH = os.environ.get("COMO_TRACK_FRAME_GRAPH", "1") != "0"

    def _frame_graph_applies(self, rgb):
        return (self.FRAME_GRAPH and _pt.FUSED_LEVEL and rgb.is_cuda and self.dtype == torch.float32 and
                self.cfg["color"] == "gray" and len(self.P_pyr) > 0 and self.P_pyr[-1].shape[0] == 1 and
                not getattr(self, "_fg_disabled", False))

    def copies_its_input(self, rgb):
        """True when handle_frame will not keep a reference to `rgb` itself (the frame graph works on its own static copy and a
        frame handed to the mapper is cloned): the caller need not clone the frame for the tracker."""
        return self._frame_graph_applies(rgb) and getattr(self, "_fg", None) is not None and self._fg.get("graph") is not None

    def _lean_head_applies(self, rgb):
        """The frame graph's head as ONE launch (csrc/image.hip frame_pyramid3_kernel: luma + both blur_down levels + the clears of
        the three levels' barrier workspaces and of the median's histograms): gray float32 frames, three pyramid levels from 0."""
        pyr = self.cfg["pyr"]
        return (_LEAN_HEAD and rgb.dtype == torch.float32 and rgb.dim() == 4 and rgb.shape[0] == 1 and rgb.shape[1] == 3 and
                rgb.shape[2] >= 4 and rgb.shape[3] >= 4 and pyr["start_level"] == 0 and pyr["end_level"] == 3 and len(self.P_pyr) == 3)

    def _frame_head_fused(self, fg):
        import ctypes
        rgb = fg["rgb"]
        _, _, H, W = rgb.shape
        H1, W1 = (H + 1) // 2, (W + 1) // 2
        dev = rgb.device
        gray = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        l1 = torch.empty((1, 1, H1, W1), dtype=torch.float32, device=dev)
        l2 = torch.empty((1, 1, (H1 + 1) // 2, (W1 + 1) // 2), dtype=torch.float32, device=dev)
        zl = fg.get("zero_list")
        if zl is None:
            L = _lib.lib()
            nb = int(L.como_track_level_zero_bytes())
            ptrs, nbytes = [], []
            for ws, wsp in fg["ws"]:
                ptrs.append(ws.data_ptr()); nbytes.append(nb)
                if wsp:
                    ptrs.append(int(wsp)); nbytes.append(nb)
            from como_amd.utils.select import median_workspace
            h = median_workspace(dev, 1)
            ptrs.append(h.data_ptr()); nbytes.append(h.numel() * 4)
            zl = fg["zero_list"] = ((ctypes.c_void_p * len(ptrs))(*ptrs), (ctypes.c_long * len(ptrs))(*nbytes), len(ptrs), h)
        _lib.check(_lib.lib().como_track_frame_pyramid3_f32(rgb.data_ptr(), gray.data_ptr(), l1.data_ptr(), l2.data_ptr(), H, W, zl[0], zl[1],
                                                            zl[2], _lib.stream_ptr(dev)), "como_track_frame_pyramid3_f32")
        return [l2, l1, gray]

    def _frame_body(self, fg):
        lean = bool(fg.get("lean"))
        img_pyr = self._frame_head_fused(fg) if lean else self.prep_tracking_img(fg["rgb"])
        res = _pt.photo_tracking_levels_static(fg["T"], fg["aff"], fg["pb"], img_pyr, self.intrinsics_pyr, self.cfg["term_criteria"],
                                               fg["ws"], prezeroed=lean)
        if res is None:
            return None
        T, aff, recs = res
        dt = T.dtype
        if (_FRAME_RECORD and dt == torch.float32 and aff.dtype == dt and recs.dtype == dt and recs.dim() == 2 and recs.is_contiguous() and
                fg["T_w_kf"].dtype == dt):
            # the frame's record in ONE launch (csrc/window.hip track_frame_record_kernel): world pose, |t|, casts, concatenation
            _, _, n_seen, med = self.reproj_stats_last_kf(T, P=fg["pb"].levels[-1]["P"][0], clone_count=False, hist_prezeroed=lean)
            nl = int(recs.shape[0])
            sc = torch.empty((3 + nl + 34,), dtype=dt, device=T.device)
            if med.dtype == dt and n_seen.dtype == torch.int32:
                _lib.check(_lib.lib().como_track_frame_record_f32(
                    T.contiguous().data_ptr(), aff.contiguous().data_ptr(), fg["T_w_kf"].contiguous().data_ptr(), med.data_ptr(),
                    n_seen.data_ptr(), recs.data_ptr(), nl, int(recs.stride(0)), sc.data_ptr(), _lib.stream_ptr(T.device)),
                    "como_track_frame_record_f32")
                return T, aff, sc[3 + nl + 18:].view(1, 4, 4), sc
        T_w = get_T_w_curr(fg["T_w_kf"], T)
        _, _, n_seen, med = self.reproj_stats_last_kf(T, P=fg["pb"].levels[-1]["P"][0], hist_prezeroed=lean)
        # everything the host reads and keeps of a frame in ONE buffer: [|t|, median depth, pixels seen, per-level barrier status |
        # T_curr_kf (16) | aff_curr_kf (2) | T_w_curr (16)] -- one read-back and one copy per frame instead of one + three
        sc = torch.cat((torch.linalg.norm(T[:, :3, 3]).reshape(1), med.reshape(1).to(dt), n_seen.reshape(1).to(dt), recs[:, 104],
                        T.reshape(-1), aff.reshape(-1).to(dt), T_w.reshape(-1).to(dt)))
        return T, aff, T_w, sc

    def _track_frame_graph(self, rgb):
        """(T_curr_kf, aff_curr_kf, T_w_curr, |t|, median depth, pixels seen) or None (the caller then runs the eager path)."""
        fg = getattr(self, "_fg", None)
        if fg is None or fg["rgb"].shape != rgb.shape or (getattr(self, "_pb", None) is not None and fg["pb"] is not self._pb):
            pb = getattr(self, "_pb", None)
            if pb is None:
                pb = _pt.pyr_buffers(self.vals_pyr, self.P_pyr, self.dI_dT_pyr, self.prep_tracking_img(rgb), self.intrinsics_pyr)
            # ONE barrier workspace pair per tracker (the uncached allocation has no destroy entry point): a rebuilt frame graph
            # -- new pyramid buffers, another image size -- reuses it; the old graph is dropped with the old dict
            lean = self._lean_head_applies(rgb)
            wsp = getattr(self, "_fg_ws", None)
            if wsp is None or (wsp[0][0] if isinstance(wsp, list) else wsp[0]).device != rgb.device or isinstance(wsp, list) != lean:
                # (lean head: one pair PER level -- all cleared inside the graph's first launch instead of one clear launch per level)
                wsp = self._fg_ws = ([_pt.level_workspace_pair(rgb.device) for _ in range(3)] if lean else
                                     _pt.level_workspace_pair(rgb.device))
            fg = self._fg = {"rgb": torch.empty_like(rgb), "T": torch.empty_like(self.T_curr_kf), "aff": torch.empty_like(self.aff_curr_kf),
                             "T_w_kf": torch.empty_like(self.T_w_kf), "pb": pb, "ws": wsp, "lean": lean,
                             "graph": None, "out": None, "warm": 0}
        srcs = [rgb, self.T_curr_kf.reshape(1, 4, 4), self.aff_curr_kf.reshape(1, 2, 1), self.T_w_kf]
        dsts = [fg["rgb"], fg["T"], fg["aff"], fg["T_w_kf"]]
        if _FOREACH_INPUTS and all(a.dtype == b.dtype and a.device == b.device and a.shape == b.shape for a, b in zip(srcs, dsts)):
            torch._foreach_copy_(dsts, srcs)            # the frame graph's four inputs in ONE launch (a dependent launch is >= 4.5 us)
        else:
            for b, a in zip(dsts, srcs):
                b.copy_(a)
        fg["pb"].load_reference(self.vals_pyr, self.P_pyr, self.dI_dT_pyr, self.mask_pyr)
        if fg["graph"] is None and fg["warm"] >= 2:
            torch.cuda.synchronize(rgb.device)
            g, out = _lib.capture_graph(lambda: self._frame_body(fg), rgb.device)
            if g is None or out is None:
                self._fg_disabled = True               # (capture refused: stay on the eager path for the rest of the run)
                return None
            fg["graph"], fg["out"] = g, out
        if fg["graph"] is not None:
            fg["graph"].replay()
            out = fg["out"]
        else:
            out = self._frame_body(fg)
            fg["warm"] += 1
            if out is None:
                self._fg_disabled = True
                return None
        T, aff, T_w, sc = out
        hook = getattr(self, "while_waiting", None)
        if hook is not None:                            # (host work that fits under the tracking launches: sequential.py sets it)
            hook(getattr(self, "_cur_timestamp", None))
        v = sc.tolist()                                 # the frame's one host synchronisation
        nl = len(v) - 34 - 3                            # pyramid levels
        if min(v[3:3 + nl]) < 0:                        # a level kernel's barrier timed out / XCD census failed: track eagerly
            if _pt.level_kernel_failed():               # (the XCD-local form is off from now on: the graph holds the old launches)
                self._fg = None
            return None
        keep = sc[3 + nl:].clone()                      # (the graph's buffers are overwritten by the next replay)
        return (keep[:16].view(1, 4, 4).to(T.dtype), keep[16:18].view(1, 2, 1).to(aff.dtype), keep[18:34].view(1, 4, 4).to(T_w.dtype),
                v[0], v[1], int(v[2]))

    # ---- one frame (Tracking.py:315-379) -----------------------------------------------------------------------------
    def handle_frame(self, data):
        timestamp, rgb = data
        self._cur_timestamp = timestamp
        if self._frame_graph_applies(rgb):
            res = self._track_frame_graph(rgb)
            if res is not None:
                self.T_curr_kf, self.aff_curr_kf, T_w_curr, norm_t, median_depth, n_seen = res
                self.last_reproj_stats = (n_seen, median_depth)
                track_data_viz = (timestamp, T_w_curr)
                track_data_map = None
                kind = self.decide_frame(norm_t, median_depth, n_seen, T_w_curr)
                hook = getattr(self, "after_decision", None)
                if hook is not None:                    # (device work that fits into the host-bound hand-over: sequential.py sets it)
                    hook(kind)
                if kind == "keyframe":
                    track_data_map = ("keyframe", rgb.clone(), self.T_curr_kf, self.aff_curr_kf, self.kf_received_ts, timestamp)
                    self.last_kf_sent_ts = timestamp
                elif kind == "one-way":
                    track_data_map = ("one-way", rgb.clone(), self.T_curr_kf, self.aff_curr_kf, self.kf_received_ts, timestamp)
                    self.last_rec_sent_ts = timestamp
                    self.num_one_way_since_kf += 1
                return track_data_viz, track_data_map
        img_pyr = self.prep_tracking_img(rgb)
        self.T_curr_kf, self.aff_curr_kf = photo_tracking_pyr(self.T_curr_kf, self.aff_curr_kf, self.vals_pyr, self.P_pyr,
                                                              self.dI_dT_pyr, self.mask_pyr, self.intrinsics_pyr, img_pyr,
                                                              self.cfg["sigmas"]["photo"], self.cfg["term_criteria"])
        T_w_curr = self.get_curr_world_pose()
        track_data_viz = (timestamp, T_w_curr.clone())
        track_data_map = None

        reproj, seen, n_seen, median_depth = self.reproj_stats_last_kf(self.T_curr_kf)
        self.last_reproj_stats = (n_seen, median_depth)

        if self.check_keyframe(median_depth, n_seen, self.T_curr_kf):
            track_data_map = ("keyframe", rgb.clone(), self.T_curr_kf, self.aff_curr_kf, self.kf_received_ts, timestamp)
            self.last_kf_sent_ts = timestamp
        elif self.check_one_way_frame(median_depth, n_seen, self.T_curr_kf, T_w_curr):
            track_data_map = ("one-way", rgb.clone(), self.T_curr_kf, self.aff_curr_kf, self.kf_received_ts, timestamp)
            self.last_rec_sent_ts = timestamp
            self.num_one_way_since_kf += 1
        return track_data_viz, track_data_map
