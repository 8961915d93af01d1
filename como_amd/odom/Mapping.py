"""Headless keyframe-window mapper: the state machine of the reference's `Mapping` (como/odom/Mapping.py:42-968) without the
GUI / process / queue plumbing.  Same method names, arguments and state attributes (`kf_poses`, `P_m`,
`correspondence_mask`, `recent_poses`, ...), so the reference's caller code can drive it unchanged.

What runs where:
  * `iterate()`                 -> `WindowBA` (odom/window_ba.py): the fused HIP chain (scaffold, dense reference, photometric
                                   system, priors, Cholesky, update) once the window is full; the
                                   reference-signature path while it fills.  The WindowBA object is rebuilt whenever the
                                   window's topology changes (new keyframe / one-way frame) and owns the iteration state in
                                   between; this class's attributes are refreshed from it after every iteration.
  * `add_keyframe()`            -> Scharr gradients (csrc/image.hip), DepthCov network (csrc/nn.hip), `track_and_init`
                                   (frontend/corr.py: HIP sampler + covariance kernels), K~ predictor (csrc/densify.hip).
  * window bookkeeping (correspondence mask, anchors, pruning): small torch ops on the device, as in the reference.

Deviations, documented: (1) no checkpoint loader -- `setup(model)` takes a `DepthCovModule` mirror (or a state dict);
(2) `depth_imgs` is evaluated on demand from the stored log-depths instead of on every iteration (the reference recomputes
B x H x W x m products per GN iteration only to keep this cache warm; the per-keyframe median of that image, which the
priors and the landmark re-initialisation read, IS computed every iteration -- WindowBA's full-image median pass).
"""
import os

import torch

_PIX_MIRRORS = os.environ.get("COMO_PIX_MIRRORS", "1") != "0"       # 0: every window rebuild converts the whole K~ / image window (A/B)
_KT_DIRECT = os.environ.get("COMO_KF_KT_DIRECT", "1") != "0"        # 0: K~ of a new keyframe is formed in a temporary and copied / converted into the window (A/B)
_HOST_CORR = os.environ.get("COMO_KF_HOST_CORR", "1") != "0"        # 0: the correspondence-mask bookkeeping of a keyframe insertion on the device (A/B)
_SE3_NORMALIZE_KERNEL = os.environ.get("COMO_SE3_NORMALIZE_KERNEL", "1") != "0"   # 0: LAPACK SVD on the host (A/B)
_FUSED_FRAME = os.environ.get("COMO_FUSED_FRAME", "1") != "0"       # 0: the torch chains of a frame hand-over (world pose / affine, gray + gradients + cat + copies) (A/B)
_KEPT_MEDIANS = os.environ.get("COMO_KF_KEPT_MEDIANS", "1") != "0"  # 0: a keyframe insertion re-evaluates every keyframe's depth image (A/B)
_TRACK_REF_PIX = os.environ.get("COMO_TRACK_REF_PIX", "1") != "0"    # 0: the tracker's reference depth image from the float64 K~ whatever pix_dtype is (A/B)
_ASYNC_NETWORK = os.environ.get("COMO_KF_ASYNC_NETWORK", "1") != "0"   # 0: the covariance network of a keyframe insertion in line on the main stream (A/B)
_RETARGET = os.environ.get("COMO_BA_RETARGET", "1") != "0"          # 0: a one-way frame builds a new window object, as round 5 (A/B)

from como_amd.depth_cov.core.covariance import prep_predictor as _prep_predictor
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule, run_model as _run_model
from como_amd.geometry.affine_brightness import get_aff_w_curr
from como_amd.geometry.camera import backprojection, backprojection_points
from como_amd.geometry.transforms import get_T_w_curr, transform_points, transform_points_values
from como_amd.odom.frontend.corr import prepare_track_and_init, track_and_init
from como_amd.odom.frontend.TwoFrameSfm import TwoFrameSfm
from como_amd.odom.backend.dense_ref import depth_image
from como_amd.odom.window_ba import WindowBA
from como_amd.utils.coords import swap_coords_xy
from como_amd.utils.select import masked_median
from como_amd.depth_cov.nn.UNet import resize_aa
from como_amd.utils.image_processing import ImageGradientModule, rgb_to_grayscale

_DTYPES = {"float": torch.float32, "double": torch.float64}


def normalizeSE3_inplace(T):
    """Project the rotation block onto SO(3) (reference geometry/lie_algebra.py:98-101: U V^T of the SVD).  On the GPU: one thread
    per pose running Newton's polar iteration (csrc/window.hip se3_normalize_kernel) -- the pose comes from a tracked float32
    transform, i.e. its rotation block is a rotation up to 1e-7; the device SVD is a dozen solver launches with their own
    synchronisations (~1 ms) and the host LAPACK form a device -> host -> device round trip in the middle of a keyframe insertion."""
    if T.is_cuda and T.dtype in (torch.float32, torch.float64) and T.is_contiguous() and T.dim() == 3 and _SE3_NORMALIZE_KERNEL:
        from como_amd import _lib
        _lib.check(getattr(_lib.lib(), "como_se3_normalize_" + _lib.suffix(T.dtype))(T.data_ptr(), T.shape[0], _lib.stream_ptr(T.device)),
                   "como_se3_normalize")
        return
    R = T[..., :3, :3]
    U, _, Vh = torch.linalg.svd(R.cpu())
    T[..., :3, :3] = (U @ Vh).to(R.device)


class Mapping:
    def __init__(self, cfg, intrinsics):
        self.cfg = cfg
        self.device = cfg["device"]
        self.dtype = _DTYPES[cfg["dtype"]] if isinstance(cfg["dtype"], str) else cfg["dtype"]
        self.pix_dtype = _DTYPES[cfg.get("pix_dtype", "float")]      # element type of the per-pixel kernels
        self.intrinsics = intrinsics.unsqueeze(0)
        self.is_init = False
        self._ba = None

    def setup(self, model=None):
        self.init_basic_vars()
        self.load_model(model)
        self.init_keyframe_vars()
        self.init_prior_vals()
        self.reset_iteration_vars(new_kf=True, converged=True)
        self.two_frame_sfm = TwoFrameSfm(self.cfg, self.intrinsics[0, :, :], self.model, self.cov_level, self.network_size)

    # ---- construction (Mapping.py:70-136) -----------------------------------------------------------------------------
    def init_basic_vars(self):
        c = {"gray": 1, "rgb": 3}[self.cfg["color"]]
        self.intrinsics = self.intrinsics.to(device=self.device, dtype=self.dtype)
        self.gradient_module = ImageGradientModule(channels=c, device=self.device, dtype=self.dtype)
        self.last_kf_send_time = 0.0

    def init_keyframe_vars(self):
        e = lambda dt=None: torch.empty((0), device=self.device, dtype=dt or self.dtype)
        self.kf_timestamps = []
        self.rgb, self.kf_img_and_grads, self.cov_params_img = e(), e(), e()
        self.kf_poses, self.kf_aff_params = e(), e()
        self.depth_dims = []
        self.pm_first_obs, self.pm, self.logzm = e(), e(), e()
        self.L_mm, self.Kmm_inv, self.Knm_Kmminv = e(), e(), e()
        self.correspondence_mask = e()
        self.P_m = e()
        self.obs_ref_mask = e(torch.bool)
        self.recent_timestamps = []
        self.recent_img_and_grads, self.recent_poses, self.recent_aff_params = e(), e(), e()
        self.Knm_Kmminv_pix = self.kf_img_and_grads_pix = self.recent_img_and_grads_pix = None
        self._depth_cache = None

    def init_prior_vals(self):
        self.window_full = False
        self.pose_anchor = torch.empty((0), device=self.device, dtype=self.dtype)
        self.aff_anchor = torch.empty((0), device=self.device, dtype=self.dtype)
        self.sparse_log_depth_anchor = torch.empty((0), device=self.device, dtype=self.dtype)
        self.init_scale_anchor = None

    def load_model(self, model=None):
        """Mapping.py:397-407.  model: DepthCovModule mirror, a state dict, or None (then cfg['model_path'] must point to a
        torch-saved state dict)."""
        self.cov_level = -1
        self.network_size_list = list(self.cfg.get("network_size", [192, 256]))
        self.network_size = torch.tensor(self.network_size_list, device=self.device)
        if model is None:
            model = torch.load(self.cfg["model_path"], map_location=self.device)
        if isinstance(model, dict):
            sd = model.get("state_dict", model)
            model = DepthCovModule({k: v.to(self.device) for k, v in sd.items()})
        self.model = model

    # ---- window helpers (Mapping.py:470-497) ----------------------------------------------------------------------------
    # Per-pixel data the window solver reads in ITS element type (cfg pix_dtype): with a float32 pixel path and the reference's
    # float64 mapping state, every rebuild of the window (each keyframe AND each one-way frame) converted the whole K~ window
    # (1.26 GB at 9 x 640x480) and all image stacks again.  The mirrors below follow the same window operations, so only the
    # NEW keyframe / frame is converted, once.
    _PIX_MIRRORED = ("Knm_Kmminv", "kf_img_and_grads", "recent_img_and_grads")
    # the large per-frame stacks live in sliding double-capacity buffers (`_cat_sliding`): a window operation writes the NEW frame
    # only -- `torch.cat` of the kept frames copied 59 MB (images of 8 keyframes) to 177 MB (24 one-way frames) per stack, on every
    # keyframe / one-way frame
    _SLIDING = {"Knm_Kmminv": "num_keyframes", "kf_img_and_grads": "num_keyframes", "rgb": "num_keyframes",
                "cov_params_img": "num_keyframes", "recent_img_and_grads": "num_one_way_frames"}

    def _cat(self, name, new_var, i):
        old = getattr(self, name)
        if _PIX_MIRRORS and name in self._PIX_MIRRORED and new_var.is_cuda and self.pix_dtype != new_var.dtype:
            pix = name + "_pix"
            if getattr(self, pix, None) is None or (old.numel() == 0 and old.dim() == 1):
                setattr(self, pix, torch.empty((0), device=new_var.device, dtype=self.pix_dtype))
            # (converted by the copy into the window buffer itself, not into a temporary first: 157 MB per keyframe for K~)
            self._cat_sliding(pix, getattr(self, pix), new_var, i, dtype=self.pix_dtype, cap=self.cfg["graph"][self._SLIDING[name]])
        base = name[:-4] if name.endswith("_pix") else name
        if base in self._SLIDING and new_var.is_cuda and new_var.dim() > 1:
            return self._cat_sliding(name, old, new_var, i, cap=self.cfg["graph"][self._SLIDING[base]])
        setattr(self, name, new_var.clone() if old.numel() == 0 and old.dim() == 1 else torch.cat((old[i:, ...], new_var), dim=0))

    def _cat_sliding(self, name, old, new_var, i, dtype=None, cap=None):
        """The dense predictors K~ are 157 MB per keyframe at 640x480 (float64): `torch.cat` of a growing window asks the allocator
        for a new, larger block on every keyframe (a hipMalloc of > 1 GB: ~12 ms each while the window fills), and copying the kept
        keyframes into a second buffer moved 1.3 GB (+ the pixel-type mirror) per keyframe (0.55 ms).  ONE buffer of twice the
        window's capacity instead: the window is a view [start, start + count) that slides -- the kept keyframes stay where they
        are, the new one is written behind them -- and is moved back to the front when it reaches the end (once per
        `num_keyframes` insertions)."""
        dst = self._slide_reserve(name, old, new_var.shape[0], tuple(new_var.shape[1:]), i, dtype or new_var.dtype, new_var.device, cap)
        dst.copy_(new_var)                                          # (the window's element type: the copy converts)

    def _slide_reserve(self, name, old, n_new, tail, i, dtype, device, cap=None):
        """The sliding-window step of `_cat_sliding` without the copy: makes room for n_new frames of shape `tail` behind the kept
        ones, points the attribute `name` at the new window and returns the view of the NEW frames' slots for the producer to
        write into (csrc/image.hip frame_stack_kernel writes a frame's image stack there directly)."""
        cap = cap or self.cfg["graph"]["num_keyframes"]
        store = self.__dict__.setdefault("_kt_pp", {})              # per window tensor (e.g. K~ and its pixel-type mirror)
        st = store.get(name)
        if (st is None or tuple(st["buf"].shape[1:]) != tuple(tail) or st["buf"].dtype != dtype or st["buf"].shape[0] != 2 * cap or
                st["buf"].device != torch.device(device)):
            st = store[name] = {"buf": torch.empty((2 * cap,) + tuple(tail), dtype=dtype, device=device),
                                "start": 0, "count": 0}
        buf = st["buf"]
        empty = old.numel() == 0 and old.dim() == 1
        keep = old[0:0] if empty else old[i:, ...]
        k = keep.shape[0]
        if k + n_new > cap:
            raise RuntimeError("como_amd Mapping: more keyframes than graph.num_keyframes")
        # `old` is the view handed out last time, or a suffix / sub-range of it (prune_one_way slices the one-way stacks): any view
        # of whole slots of this buffer stays where it is
        in_place = False
        if (not empty and old.dtype == buf.dtype and old.stride() == buf.stride() and old.device == buf.device and
                old.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()):
            slot = buf.stride(0) * buf.element_size()
            rel = old.data_ptr() - buf.data_ptr()
            in_place = slot > 0 and rel >= 0 and rel % slot == 0 and rel // slot + old.shape[0] <= 2 * cap
        if in_place:
            s0 = rel // slot + old.shape[0] - k                     # the kept frames stay where they are
            if s0 + k + n_new > 2 * cap:                            # (then s0 > cap >= k: source and destination are disjoint)
                if k:
                    buf[:k].copy_(buf[s0:s0 + k])
                s0 = 0
        else:                                                       # a window tensor that is not the view handed out last time
            s0 = 0
            if k:
                same = keep.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()
                buf[:k].copy_(keep.clone() if same else keep)
        st["start"], st["count"] = s0, k + n_new
        setattr(self, name, buf[s0:s0 + k + n_new])
        return buf[s0 + k:s0 + k + n_new]

    def _slide_peek(self, name, old, n_new, i, cap):
        """Device address the window tensor `name` WOULD start at after `_slide_reserve(name, old, n_new, ..., i, ...)`, without
        doing it (None when that cannot be told: no buffer yet, a view that is not the one handed out last time)."""
        st = self.__dict__.get("_kt_pp", {}).get(name)
        if st is None or old is None:
            return None
        buf = st["buf"]
        if buf.shape[0] != 2 * cap:
            return None
        slot = buf.stride(0) * buf.element_size()
        if old.numel() == 0 and old.dim() == 1:
            return buf.data_ptr()
        if not (old.dtype == buf.dtype and old.stride() == buf.stride() and old.device == buf.device and
                old.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()) or slot <= 0:
            return None
        rel = old.data_ptr() - buf.data_ptr()
        if rel < 0 or rel % slot or rel // slot + old.shape[0] > 2 * cap:
            return None
        k = old[i:, ...].shape[0]
        s0 = rel // slot + old.shape[0] - k
        if s0 + k + n_new > 2 * cap:
            s0 = 0
        return buf.data_ptr() + s0 * slot

    def speculate_one_way(self, timestamp):
        """Called by the tracker between the launch of a frame's tracking and the wait for its result (`Tracking.while_waiting`):
        if this frame turns out to be a one-way frame, the window the next iteration runs on is the current keyframe set + the
        one-way frames below -- its pair table is built NOW (`WindowBA.speculate`), while the host has nothing else to do.
        Nothing of the mapper's state is touched; a wrong guess (plain frame, keyframe) costs idle host time only."""
        ba = self._ba
        if ba is None or not self.is_init or not hasattr(ba, "speculate"):
            return
        i = self.get_recent_start_window_ind()
        rec = list(self.recent_timestamps[i:]) + [timestamp]
        cap = self.cfg["graph"]["num_one_way_frames"]
        name = "recent_img_and_grads_pix" if (_PIX_MIRRORS and self.pix_dtype == torch.float32 and self.dtype != self.pix_dtype) else \
            "recent_img_and_grads"
        old = getattr(self, name, None)
        if name.endswith("_pix") and (old is None or (self.recent_img_and_grads.numel() == 0 and self.recent_img_and_grads.dim() == 1)):
            old = torch.empty((0), device=self.kf_poses.device, dtype=torch.float32)     # (add_one_way_frame starts the mirror over)
        ba.speculate(self.kf_timestamps, rec, self._slide_peek(name, old, 1, i, cap))

    def median_ahead_now(self, kind):
        """`median_ahead: gap` -- called by the tracker the moment a frame's request is known (`Tracking.after_decision`): a one-way
        frame leaves the device idle for ~0.25 ms while the host hands the frame over and re-targets the window; the full-image
        median of the iteration that follows (0.7 GB of K~: ~100 us of streaming) only depends on the keyframes' state, which that
        hand-over does not touch -- it is streamed into the gap (`WindowBA._issue_median_ahead`) and the iteration's side branch
        is left with the priors.  Not for a keyframe (another window follows) nor a plain frame (no gap: its iteration is queued at once)."""
        ba = self._ba
        if (kind == "one-way" and ba is not None and self.is_init and not self.converged and getattr(ba, "median_ahead", False) == "gap"):
            ba._issue_median_ahead()

    def window_cat_helper_list(self, var, new_var, i):
        del var[:i]
        var.append(new_var)

    def get_kf_start_window_ind(self):
        return -self.cfg["graph"]["num_keyframes"] + 1

    def get_recent_start_window_ind(self):
        return -self.cfg["graph"]["num_one_way_frames"] + 1

    def _retire_ba(self):
        """The window solver is rebuilt at the next iterate; the old one is kept until then so that the new one can take over
        what only depends on the keyframe set (WindowBA(prev=...))."""
        if self._ba is not None:
            self._ba_prev = self._ba
        self._ba = None

    def reset_iteration_vars(self, new_kf, converged=False):
        self.converged = converged
        if new_kf:
            self._state_owner = None                          # keyframe state changed outside an iteration
        self._retire_ba()                                 # topology (or a frame's initial values) changed
        if new_kf:
            self.iter = 0
            self.total_err_prev = float("inf")

    def get_safe_ind(self, ind):
        n = self.kf_poses.shape[0]
        return n - 1 if ind == -1 or ind >= n else ind

    # ---- images / network / predictor ------------------------------------------------------------------------------------
    def get_img_and_grads(self, rgb):
        img = rgb_to_grayscale(rgb) if self.cfg["color"] == "gray" else rgb
        if img.is_cuda:
            from como_amd.utils.image_processing import img_and_grads
            return img_and_grads(img)                         # the kernel writes [img | gx | gy] itself (csrc/image.hip): no cat
        gx, gy = self.gradient_module(img)
        return torch.cat((img, gx, gy), dim=1)

    def run_model(self, rgb):
        """Mapping.py:409-428 on the HIP network."""
        return _run_model(self.model, rgb, network_size=self.network_size_list, dtype=self.dtype,
                          graphed=self.cfg.get("graph_network", True))

    def start_model(self, rgb):
        """`run_model` on a side stream, started as soon as the tracker has asked for a keyframe: the network only reads the frame,
        while the ~45 small launches and two read-backs that precede its first use in `add_keyframe` (world pose, image stack,
        depth image and reprojection of the last keyframe) are host-bound -- 0.65 ms during which the device sat idle and the
        network (0.7 ms: a chain of ~60 dependent launches on a fraction of the compute units) had not even been submitted."""
        dev = rgb.device
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_net_stream", None)
        if side is None:
            side = self._net_stream = torch.cuda.Stream(device=dev)
        side.wait_stream(main)                                # (the frame is ready; an unclaimed earlier evaluation ran on `side` too)
        with torch.cuda.stream(side):
            cov = self.run_model(rgb)
        cov.record_stream(main)
        ev = torch.cuda.Event()
        ev.record(side)
        self._net_pending = (rgb.data_ptr(), tuple(rgb.shape), cov, ev)

    def take_model(self, rgb):
        """The covariance image of `rgb`: the side-stream evaluation `start_model` began for this frame, or `run_model` in line."""
        pend, self._net_pending = getattr(self, "_net_pending", None), None
        if pend is not None:
            # (also when the result is not the one asked for: the graphed network owns ONE set of static buffers per input shape,
            # an evaluation in line must not start while the side stream still runs in them)
            torch.cuda.current_stream(rgb.device).wait_event(pend[3])
            if pend[1] == tuple(rgb.shape):
                return pend[2]
        return self.run_model(rgb)

    def prep_predictor(self, cov_params_img, coords_m, into_window=False):
        """Mapping.py:430-468 -> (K_mm_inv, L_mm, Knm_Kmminv (b,H,W,m)); K_nm is never materialised (csrc/densify.hip).
        into_window (add_keyframe): K~ is written STRAIGHT into the new keyframe's slot of the window's sliding predictor buffer (and,
        with float32 per-pixel kernels, rounded into the mirror's slot by the same launch) -- `self.Knm_Kmminv` then already is the
        new window and initialize_sparse_pixel_vars must not append it again (pass Knm_Kmminv=None): saves the 157 MB copy and the
        conversion pass of every keyframe insertion at 640x480."""
        size = tuple(self.kf_img_and_grads.shape[-2:]) if self.kf_img_and_grads.numel() else tuple(cov_params_img.shape[-2:])
        cm = coords_m.to(cov_params_img.dtype)
        need_mirror = _PIX_MIRRORS and self.pix_dtype != cov_params_img.dtype
        mirror_ok = (not need_mirror) or (self.pix_dtype == torch.float32 and cov_params_img.dtype == torch.float64 and
                                          getattr(self, "Knm_Kmminv_pix", None) is not None and self.Knm_Kmminv_pix.dim() == 4)
        if (into_window and _KT_DIRECT and mirror_ok and cov_params_img.is_cuda and cov_params_img.shape[0] == 1 and
                self.Knm_Kmminv.dim() == 4 and cov_params_img.dtype == self.Knm_Kmminv.dtype and
                tuple(self.Knm_Kmminv.shape[1:]) == size + (cm.shape[1],)):
            i = self.get_kf_start_window_ind()
            cap = self.cfg["graph"]["num_keyframes"]
            tail = size + (cm.shape[1],)
            dst = self._slide_reserve("Knm_Kmminv", self.Knm_Kmminv, 1, tail, i, cov_params_img.dtype, cov_params_img.device, cap)
            dpix = None
            if need_mirror:
                dpix = self._slide_reserve("Knm_Kmminv_pix", self.Knm_Kmminv_pix, 1, tail, i, torch.float32, cov_params_img.device, cap)
            Kinv, L_mm, _ = _prep_predictor(cov_params_img, cm, self.model.get_scale(-1), photo_img_size=size, out=dst, out_pix=dpix)
            return Kinv, L_mm, None
        return _prep_predictor(cov_params_img, cm, self.model.get_scale(-1), photo_img_size=size)

    # ---- tracker-facing accessors ----------------------------------------------------------------------------------------
    def find_kf_from_timestamp(self, kf_timestamp):
        for i in range(len(self.kf_timestamps) - 1, -1, -1):
            if kf_timestamp == self.kf_timestamps[i]:
                return i
        return None

    def get_curr_world_pose(self, pose_curr_kf, kf_ind):
        return get_T_w_curr(self.kf_poses[kf_ind:kf_ind + 1, ...], pose_curr_kf)

    def get_curr_world_aff(self, aff_curr_kf, kf_ind):
        return get_aff_w_curr(self.kf_aff_params[kf_ind:kf_ind + 1, ...], aff_curr_kf)

    def get_curr_world_state(self, pose_curr_kf, aff_curr_kf, kf_ind):
        """(get_curr_world_pose, get_curr_world_aff) of a tracked frame in ONE launch (csrc/window.hip frame_world_kernel: the same
        arithmetic, the tracker's float32 values widened first) -> ((1,4,4), (1,2,1)) in the mapping dtype."""
        if (_FUSED_FRAME and pose_curr_kf.is_cuda and self.dtype == torch.float64 and pose_curr_kf.dtype == aff_curr_kf.dtype and
                pose_curr_kf.dtype in (torch.float32, torch.float64) and pose_curr_kf.numel() == 16 and aff_curr_kf.numel() == 2 and
                self.kf_poses.dtype == torch.float64 and self.kf_aff_params.dtype == torch.float64):
            from como_amd import _lib
            dev = pose_curr_kf.device
            out = torch.empty(18, dtype=torch.float64, device=dev)
            Tk = self.kf_poses[kf_ind:kf_ind + 1].contiguous()
            ak = self.kf_aff_params[kf_ind:kf_ind + 1].contiguous()
            _lib.check(_lib.lib().como_frame_world_f64(Tk.data_ptr(), pose_curr_kf.contiguous().data_ptr(), ak.data_ptr(),
                                                       aff_curr_kf.contiguous().data_ptr(), 1 if pose_curr_kf.dtype == torch.float32 else 0,
                                                       out.data_ptr(), out[16:].data_ptr(), _lib.stream_ptr(dev)), "como_frame_world_f64")
            return out[:16].view(1, 4, 4), out[16:].view(1, 2, 1)
        return (self.get_curr_world_pose(pose_curr_kf.to(self.dtype), kf_ind), self.get_curr_world_aff(aff_curr_kf.to(self.dtype), kf_ind))

    @property
    def depth_imgs(self):
        """(B,1,H,W) depth images exp(K~ logz_m) of the stored log-depths (Mapping.store_vars :753-755), cached until the
        log-depths change."""
        if self._depth_cache is None and self.logzm.numel():
            b, h, w, m = self.Knm_Kmminv.shape
            self._depth_cache = depth_image(self.Knm_Kmminv.reshape(b, h * w, m), self.logzm).reshape(b, 1, h, w)
        return self._depth_cache

    def depth_imgs_of(self, lo, hi, for_tracker=False):
        """Depth images of keyframes lo..hi-1 only (one 157 MB predictor each at 640x480: the tracker asks for the newest one
        on every frame, `depth_imgs` would evaluate all nine).  for_tracker: with float32 pixel kernels (`pix_dtype: float`) the
        image is formed from the float32 mirror of K~ that the window's own dense reference reads (half the bytes, matrix-core
        kernel: 41 -> 18 us per frame) and handed over in float32 -- the tracker's element type, which it was cast to anyway."""
        if self._depth_cache is not None:
            return self._depth_cache[lo:hi]
        mirror = getattr(self, "Knm_Kmminv_pix", None)
        if (for_tracker and _TRACK_REF_PIX and mirror is not None and mirror.dtype == torch.float32 and mirror.dim() == 4 and
                mirror.shape == self.Knm_Kmminv.shape):
            Kt = mirror[lo:hi]
            b, h, w, m = Kt.shape
            lp = getattr(self, "_logzm_pix", None)
            lz = lp[0][lo:hi] if (lp is not None and lp[1] is self.logzm and lp[0].shape[0] == self.logzm.shape[0]) else self.logzm[lo:hi]
            return depth_image(Kt.reshape(b, h * w, m), lz).reshape(b, 1, h, w)
        Kt = self.Knm_Kmminv[lo:hi]
        b, h, w, m = Kt.shape
        return depth_image(Kt.reshape(b, h * w, m), self.logzm[lo:hi]).reshape(b, 1, h, w)   # one pass over K~ (csrc/densify.hip)

    def get_kf_ref_data(self, ind=-1):
        """Mapping.py:499-512: the newest `track_ref.num_keyframes` keyframes for the tracker."""
        end = self.kf_poses.shape[0]
        ind = max(0, end - self.cfg["track_ref"]["num_keyframes"])
        return (self.kf_timestamps[ind:end], self.rgb[ind:end], self.kf_poses[ind:end], self.kf_aff_params[ind:end],
                self.depth_imgs_of(ind, end, for_tracker=True))

    def get_kf_viz_data(self, ind=-1):
        """Mapping.py:514-544: cloned snapshot for a viewer -- the reference's 10-tuple, returned by default as the reference does.
        The GUI is out of scope: a headless loop that has no consumer for it sets `viewer_snapshots: false`
        in the mapping config (bench.py, como_amd.run, scripts/ate_sequence.py do) and gets None -- 2 x all depth images + clones of
        the colour images and the window state per keyframe, ~40 launches saved; the send time is kept either way (MappingSeq.map's
        1 s rule)."""
        import time
        self.last_kf_send_time = time.time()
        if not self.cfg.get("viewer_snapshots", True):
            return None
        return (self.kf_timestamps.copy(), self.rgb.clone(), self.kf_poses.clone(), self.depth_imgs.clone(),
                swap_coords_xy(self.pm).clone(), self.P_m.clone(), self.obs_ref_mask.clone(), self.recent_poses.clone(),
                list(getattr(self, "kf_pairs", [])), list(getattr(self, "one_way_pairs", [])))

    def store_vars(self, pm, logzm, Knm_Kmminv, kept_depths=None, kept_medians=None):
        """Mapping.py:749-758.  kept_depths: depth images of keyframes 0..B-2 that are known to be current (add_keyframe: the
        log-depths of the keyframes that stay are untouched by the insertion) -- only the new keyframe's image is evaluated.
        kept_medians (B-1,): the median depths of those keyframes as the last iteration stored them -- the exact medians of
        exp(K~ logz_m) at the very log-depths `logzm` holds for them (Mapping.iterate publishes the scaffold's log-depths and the
        medians of their depth images together): with them only the NEW keyframe's predictor is streamed (one 157 MB pass at
        640x480 instead of nine + a nine-segment select: ~0.4 ms per keyframe insertion)."""
        self.pm, self.logzm = pm, logzm
        self._depth_cache = None
        B = self.logzm.shape[0]
        if kept_depths is not None and B > 1 and kept_depths.shape[0] == B - 1:
            self._depth_cache = torch.cat((kept_depths, self.depth_imgs_of(B - 1, B)), dim=0)
        elif kept_medians is not None and B > 1 and kept_medians.dim() == 1 and kept_medians.shape[0] == B - 1 and _KEPT_MEDIANS:
            d_new = self.depth_imgs_of(B - 1, B)
            med_new = masked_median(d_new.reshape(1, -1)).to(d_new.dtype)
            self.median_depths = torch.cat((kept_medians.to(d_new.dtype), med_new))
            return
        d = self.depth_imgs
        d2 = d.reshape(d.shape[0], -1)
        # per-keyframe exact median of the full depth image: one segmented device select instead of B sorts
        self.median_depths = masked_median(d2).to(d2.dtype)     # (device tensors only: dense_ref.depth_image has no CPU form)

    # ---- keyframe insertion (Mapping.py:138-229) -------------------------------------------------------------------------
    def init_keyframe(self, rgb, cov_params_img, coords_m, pose_init, logz_m, aff_init, timestamp):
        """First keyframe with given inducing points and log-depths (from the two-frame initialisation)."""
        img_and_grads = self.get_img_and_grads(rgb)
        cov_params_img = resize_aa(cov_params_img, rgb.shape[-2:])
        self.initialize_pose_vars(pose_init, aff_init)
        self.initialize_kf_img_vars_vars(rgb, img_and_grads, cov_params_img)
        Kmm_inv, L_mm, Knm_Kmminv = self.prep_predictor(cov_params_img, coords_m)
        depth_dim = coords_m.shape[1]
        pm = swap_coords_xy(coords_m)
        z_m = torch.exp(logz_m)
        self.initialize_sparse_pixel_vars(pm, z_m, depth_dim, Kmm_inv, L_mm, Knm_Kmminv)
        Pc_m, _ = backprojection(self.intrinsics[0], pm, z_m)
        Pw_m, _, _ = transform_points(pose_init, Pc_m)
        self.initialize_sparse_landmark_vars(torch.ones((1, depth_dim), device=self.device, dtype=torch.bool), Pw_m.squeeze(0))
        self.kf_timestamps = [timestamp]
        self.store_vars(pm, logz_m, Knm_Kmminv)

    def add_keyframe(self, rgb, kf_pose_init, kf_aff_init, timestamp):
        from como_amd.depth_cov.core.samplers import check_pending_info
        check_pending_info()                                  # (status words of the previous insertion's sampler: arrived long ago)
        img_and_grads = self.get_img_and_grads(rgb)
        coords_m_last = swap_coords_xy(self.pm[-1:, ...])
        zm_last = torch.exp(self.logzm[-1:, ...])
        z_img_last = self.depth_imgs_of(self.kf_poses.shape[0] - 1, self.kf_poses.shape[0])
        # the reprojection of the last keyframe into the new frame does not need the network's output (the covariance image has the
        # frame's size): issued first, its host synchronisation does not wait for the network, which then runs under the host work
        pre = prepare_track_and_init(self.kf_poses[-1:, ...], kf_pose_init, coords_m_last, zm_last, z_img_last, self.intrinsics,
                                     tuple(rgb.shape[-2:]), self.cfg["corr"])
        cov_params_img = self.take_model(rgb)
        coords_m_new, z_m_new, corr_mask, coords_m, zm_first_obs = track_and_init(
            self.kf_poses[-1:, ...], kf_pose_init, coords_m_last, zm_last, z_img_last, cov_params_img,
            self.intrinsics, self.model, self.cfg["corr"], self.cfg["sampling"], self.kf_img_and_grads.shape[-2:], prepared=pre)
        # depth images of the current window, if a snapshot just evaluated them (handle_tracking_data): still valid for the keyframes
        # that stay -- the insertion below does not touch their log-depths
        kept_depths = self._depth_cache[self.get_kf_start_window_ind():] if self._depth_cache is not None else None
        md = getattr(self, "median_depths", None)
        kept_medians = (md[self.get_kf_start_window_ind():] if (torch.is_tensor(md) and md.dim() == 1 and
                                                                 md.shape[0] == self.kf_poses.shape[0]) else None)
        p_m_new = swap_coords_xy(coords_m_new).to(dtype=z_m_new.dtype)
        Pc_new = backprojection_points(self.intrinsics[0], p_m_new, z_m_new)
        Pw_new = transform_points_values(kf_pose_init, Pc_new)
        Kmm_inv, L_mm, Knm_Kmminv = self.prep_predictor(cov_params_img, coords_m, into_window=True)
        pm_first_obs = swap_coords_xy(coords_m)
        self.window_cat_helper_list(self.kf_timestamps, timestamp, self.get_kf_start_window_ind())
        self.initialize_pose_vars(kf_pose_init, kf_aff_init)
        self.initialize_kf_img_vars_vars(rgb, img_and_grads, cov_params_img)
        self.initialize_sparse_pixel_vars(pm_first_obs, zm_first_obs, coords_m_new.shape[1], Kmm_inv, L_mm, Knm_Kmminv)
        self.initialize_sparse_landmark_vars(corr_mask, Pw_new.squeeze(0), corr_host=getattr(track_and_init, "corr_host", None))
        self.reset_iteration_vars(new_kf=True)
        self.store_vars(self.pm, self.logzm, self.Knm_Kmminv, kept_depths=kept_depths, kept_medians=kept_medians)
        self.prune_one_way()

    def prune_one_way(self):
        """Drop one-way frames older than the oldest keyframe (Mapping.py:231-245)."""
        oldest = self.kf_timestamps[0]
        r = 0
        for i, ts in enumerate(self.recent_timestamps):
            if ts < oldest:
                r = i + 1
        if r:
            self.recent_timestamps = self.recent_timestamps[r:]
            self.recent_img_and_grads = self.recent_img_and_grads[r:]
            if getattr(self, "recent_img_and_grads_pix", None) is not None:
                self.recent_img_and_grads_pix = self.recent_img_and_grads_pix[r:]
            self.recent_poses = self.recent_poses[r:]
            self.recent_aff_params = self.recent_aff_params[r:]
            self._retire_ba()

    def add_one_way_frame(self, rgb, pose_init, aff_init, timestamp):
        i = self.get_recent_start_window_ind()
        self.window_cat_helper_list(self.recent_timestamps, timestamp, i)
        if (_FUSED_FRAME and rgb.is_cuda and self.cfg["color"] == "gray" and self.dtype == torch.float64 and rgb.dim() == 4 and
                rgb.shape[:2] == (1, 3) and rgb.dtype in (torch.float32, torch.float64) and rgb.is_contiguous()):
            # gray + Scharr gradients + the [I | gx | gy] stack in ONE launch, written straight into the slots of the sliding
            # buffers (float64 state + the per-pixel kernels' float32 mirror): csrc/image.hip frame_stack_kernel
            from como_amd import _lib
            H, W = rgb.shape[-2:]
            cap = self.cfg["graph"]["num_one_way_frames"]
            old64 = self.recent_img_and_grads
            was_empty = old64.numel() == 0 and old64.dim() == 1
            dst = self._slide_reserve("recent_img_and_grads", old64, 1, (3, H, W), i, torch.float64, rgb.device, cap)
            dpix = None
            if _PIX_MIRRORS and self.pix_dtype == torch.float32:
                old = self.recent_img_and_grads_pix
                if old is None or was_empty:                  # (as _cat: the mirror starts over with the window tensor)
                    old = torch.empty((0), device=rgb.device, dtype=torch.float32)
                dpix = self._slide_reserve("recent_img_and_grads_pix", old, 1, (3, H, W), i, torch.float32, rgb.device, cap)
            _lib.check(_lib.lib().como_frame_stack_f64(rgb.data_ptr(), 1 if rgb.dtype == torch.float32 else 0, H, W, dst.data_ptr(),
                                                       _lib.ptr(dpix), _lib.stream_ptr(rgb.device)), "como_frame_stack_f64")
        else:
            self._cat("recent_img_and_grads", self.get_img_and_grads(rgb.to(self.dtype)), i)
        self._cat("recent_poses", pose_init, i)
        self._cat("recent_aff_params", aff_init, i)
        self.reset_iteration_vars(new_kf=False)

    # ---- per-keyframe state (Mapping.py:262-367) -------------------------------------------------------------------------
    def initialize_pose_vars(self, pose_init, aff_init):
        num_kf = self.kf_poses.shape[0] if self.kf_poses.dim() > 1 else 0
        window_empty = num_kf == 0
        self.window_full = num_kf >= self.cfg["graph"]["num_keyframes"]
        i = self.get_kf_start_window_ind()
        pose_init = pose_init.clone()
        normalizeSE3_inplace(pose_init)
        self._cat("kf_poses", pose_init, i)
        self._cat("kf_aff_params", aff_init, i)
        if window_empty or self.window_full:
            # gauge anchors: oldest keyframe's pose; affine parameters re-expressed relative to it
            self.pose_anchor = self.kf_poses[0:1, ...].clone()
            self.kf_aff_params = self.kf_aff_params - self.kf_aff_params[0:1, ...]
            self.aff_anchor = torch.zeros_like(self.kf_aff_params[0:1, ...])

    def initialize_kf_img_vars_vars(self, rgb, img_and_grads, cov_params_img):
        i = self.get_kf_start_window_ind()
        self._cat("rgb", rgb, i)
        self._cat("kf_img_and_grads", img_and_grads, i)
        self._cat("cov_params_img", cov_params_img, i)

    def initialize_sparse_pixel_vars(self, pm_first_obs, zm_first_obs, new_depth_dim, Kmm_inv, L_mm, Knm_Kmminv):
        i = self.get_kf_start_window_ind()
        self.window_cat_helper_list(self.depth_dims, new_depth_dim, i)
        self._cat("pm_first_obs", pm_first_obs, i)
        self._cat("pm", pm_first_obs, i)
        self._cat("logzm", torch.log(zm_first_obs), i)
        self._depth_cache = None
        # which of a keyframe's inducing points were first observed in it (the newly sampled ones sit at the end)
        first = torch.zeros((1, self.cfg["sampling"]["max_num_coords"]), device=self.device, dtype=torch.bool)
        first[:, -new_depth_dim:] = True       # as the reference: with 0 new points the slice [-0:] marks ALL (Mapping.py:312)
        self._cat("obs_ref_mask", first, i)
        self._cat("Kmm_inv", Kmm_inv, i)
        self._cat("L_mm", L_mm, i)
        if Knm_Kmminv is not None:                            # (None: prep_predictor(into_window=True) already placed it)
            self._cat("Knm_Kmminv", Knm_Kmminv, i)

    def initialize_sparse_landmark_vars(self, corr_mask, P, corr_host=None):
        """Correspondence mask (num_kf x num_landmarks) and landmark list after inserting a keyframe (Mapping.py:321-367).
        corr_host: `corr_mask` as a host list (frontend/corr.py reads it back with the sampler's picks).  With it the bookkeeping --
        which landmarks are still seen by a keyframe that stays, which of the last keyframe's were tracked on -- runs on a HOST
        mirror of the correspondence mask (`_corr_host`, numpy) and the device tensors are gathered with index lists built there:
        no boolean-mask indexing, i.e. no `nonzero` + host synchronisation in the middle of a keyframe insertion (two of them
        before).  Without it (or once the mirror is lost) the device form below runs."""
        import numpy as np
        from como_amd.utils.hostlist import to_device
        i = self.get_kf_start_window_ind()
        num_kf = self.correspondence_mask.shape[0] if self.correspondence_mask.dim() > 1 else 0
        self.window_full = num_kf >= self.cfg["graph"]["num_keyframes"]
        n_new = P.shape[0]
        mirror = getattr(self, "_corr_host", None)
        if num_kf == 0:
            self.correspondence_mask = corr_mask
            self.P_m = P
            # (the first keyframe: every point is its own landmark -- init_keyframe passes an all-true mask)
            self._corr_host = np.ones(tuple(corr_mask.shape), dtype=bool) if corr_host is None and n_new == corr_mask.shape[-1] else (
                np.asarray(corr_host, dtype=bool).reshape(tuple(corr_mask.shape)) if corr_host is not None else None)
        elif (corr_host is not None and mirror is not None and mirror.shape == tuple(self.correspondence_mask.shape) and _HOST_CORR
              and self.P_m.is_cuda):
            old = mirror
            alive = old[i:, :].any(axis=0)
            tracked = np.zeros(old.shape[1], dtype=bool)
            tracked[np.nonzero(old[-1, :])[0]] = np.asarray(corr_host, dtype=bool)
            ai = np.nonzero(alive)[0]
            kept = np.concatenate((old[i:, :][:, ai], tracked[ai][None]), axis=0)
            fresh = np.zeros((kept.shape[0], n_new), dtype=bool)
            fresh[-1, :] = True
            new = np.concatenate((kept, fresh), axis=1)
            self._corr_host = new
            self.correspondence_mask = to_device(new, torch.bool, self.device)
            self.P_m = torch.cat((self.P_m.index_select(0, to_device(ai, torch.int64, self.device)), P), dim=0)
            if self.window_full:
                self.P_m_anchors = self.P_m.index_select(0, to_device(np.nonzero(new[0, :])[0], torch.int64, self.device))
            return
        else:
            self._corr_host = None                                   # (the mirror cannot follow: device form from here on)
            old = self.correspondence_mask
            alive = old[i:, :].any(dim=0)                          # landmarks still seen by a keyframe that stays
            tracked = torch.zeros_like(old[0, :])
            tracked[torch.nonzero(old[-1, :])[:, 0]] = corr_mask   # the last keyframe's landmarks that were tracked on
            ai = torch.nonzero(alive)[:, 0]                        # ONE index list for the three selections (each boolean mask = a nonzero + a sync)
            kept = torch.cat((old[i:, :].index_select(1, ai), tracked.index_select(0, ai)[None]), dim=0)
            fresh = torch.zeros((kept.shape[0], n_new), device=self.device, dtype=torch.bool)
            fresh[-1, :] = True
            self.correspondence_mask = torch.cat((kept, fresh), dim=1)
            self.P_m = torch.cat((self.P_m.index_select(0, ai), P), dim=0)
        if self.window_full:
            # landmarks of the (new) oldest keyframe carry what left the window: pin them
            self.P_m_anchors = self.P_m[self.correspondence_mask[0, :], :]

    def attempt_two_frame_init(self, timestamp, rgb):
        """Mapping.py:546-578: feed frames to the two-frame initialiser; once it succeeds, its reference becomes keyframe 0
        (inducing points and log-depths from the SfM), the current frame keyframe 1, and the mean log-depth the scale anchor
        used until the window is full."""
        sfm = self.two_frame_sfm
        self.is_init, T_curr_kf, aff_curr_kf, logd_kf, _, _, mean_log_depth = sfm.handle_frame(rgb, timestamp)
        if not self.is_init:
            return False
        # init_keyframe resizes the covariance image to the frame size: the initialiser already holds it at that size
        self.init_keyframe(sfm.rgb, sfm.cov_params_img, sfm.coords_m, sfm.pose_init, logd_kf, sfm.aff_init, sfm.timestamp)
        self.add_keyframe(rgb, get_T_w_curr(sfm.pose_init, T_curr_kf), get_aff_w_curr(sfm.aff_init, aff_curr_kf), timestamp)
        self.init_scale_anchor = mean_log_depth
        sfm.delete_init_reference()
        return True

    # ---- tracker messages (Mapping.py:580-601) ---------------------------------------------------------------------------
    def handle_tracking_data(self, data):
        kf_viz_data, kf_updated = None, False
        if data[0] in ("one-way", "keyframe"):
            rgb, pose_curr_kf, aff_curr_kf, kf_timestamp, timestamp = data[1:]
            if data[0] == "keyframe" and _ASYNC_NETWORK and rgb.is_cuda:
                self.start_model(rgb)                         # (float(rgb) is what run_model reads, whatever add_keyframe gets)
                hook = getattr(self, "while_network_runs", None)
                if hook is not None:                          # (sequential.py: the tracker prepares its side of the new keyframe now)
                    hook(timestamp, rgb)
            k = self.find_kf_from_timestamp(kf_timestamp)
            pose_w, aff_w = self.get_curr_world_state(pose_curr_kf, aff_curr_kf, k)
            if data[0] == "one-way":
                self.add_one_way_frame(rgb, pose_w, aff_w, timestamp)        # (the frame in the tracker's element type is fine)
            else:
                kf_viz_data = self.get_kf_viz_data()          # snapshot BEFORE the insertion, as the reference
                self.add_keyframe(rgb.to(self.dtype), pose_w, aff_w, timestamp)
                kf_updated = True
        return kf_viz_data, kf_updated

    # ---- one Gauss-Newton iteration over the window (Mapping.py:760-968) -------------------------------------------------
    def _window_state(self):
        ts = lambda lst: torch.as_tensor([float(t) for t in lst], dtype=torch.float64)     # host: only the pair graph reads them
        st = {"intrinsics": self.intrinsics, "kf_poses": self.kf_poses, "kf_aff_params": self.kf_aff_params,
              "kf_img_and_grads": self.kf_img_and_grads, "coords_m": self.pm, "P_m": self.P_m,
              "correspondence_mask": self.correspondence_mask, "obs_ref_mask": self.obs_ref_mask,
              "pm_first_obs": self.pm_first_obs, "L_mm": self.L_mm, "K_mm_inv": self.Kmm_inv, "Knm_Kmminv": self.Knm_Kmminv,
              "kf_timestamps": ts(self.kf_timestamps), "median_depth_init": self.median_depths,
              "pose_anchor": self.pose_anchor, "init_scale_anchor": self.init_scale_anchor}
        if self.window_full:
            st["P_anchor"] = self.P_m_anchors
        if len(self.recent_timestamps):
            st.update({"recent_poses": self.recent_poses, "recent_aff_params": self.recent_aff_params,
                       "recent_img_and_grads": self.recent_img_and_grads, "recent_timestamps": ts(self.recent_timestamps)})
        # whose state these tensors were published from (iterate): a re-targeted window need not load them again
        st["_published_by"] = getattr(self, "_state_owner", None)
        mirror = getattr(self, "_corr_host", None)
        if mirror is not None and mirror.shape == tuple(self.correspondence_mask.shape):
            st["corr_host"] = mirror
        for name in self._PIX_MIRRORED:                       # already in the solver's per-pixel element type (see _cat)
            mirror = getattr(self, name + "_pix", None)
            if mirror is not None and name in st and mirror.shape == st[name].shape:
                st[name + "_pix"] = mirror
        return st

    def _check_solver(self):
        """The status word of the PREVIOUS iteration's solve, acted on without a synchronisation of its own: iterate() leaves an
        asynchronous copy of it in pinned host memory behind the iteration; by the time the next frame reaches the mapper the
        tracker has synchronised with the device (its per-frame read-back), so the event below is long complete.
        -1 (the persistent solver timed out -- csrc/cholp.hip; the device-side guard of the update kept the state): the persistent
        form is switched off for the process and the loop goes on (that frame simply had no Gauss-Newton step); > 0 (H not positive
        definite / non-finite): RuntimeError -- the reference would silently continue on garbage (linear_system.py:109)."""
        ev = getattr(self, "_info_event", None)
        if ev is None:
            return
        self._info_event = None
        ev.synchronize()
        v = int(self._info_host[0])
        if v == 0:
            return
        if v == -1:
            from como_amd import _lib
            _lib.lib().como_chol_set_persistent(0)
            self.solver_fallbacks = getattr(self, "solver_fallbacks", 0) + 1
            return
        raise RuntimeError(f"como_amd Mapping: the window's normal equations were not positive definite (Cholesky info {v}) in "
                           f"iteration {self.iter}; that update was not applied")

    def iterate(self):
        self._check_solver()
        if self._ba is None:
            cfg = {"photo_construction": self.cfg["photo_construction"], "sigmas": self.cfg["sigmas"]}
            prev = getattr(self, "_ba_prev", None)
            st = self._window_state()
            # (eager launches: a topology lives for ~2-3 iterations in the sequential loop, less than a graph capture costs)
            if prev is not None and _RETARGET and prev.retarget(st, window_full=self.window_full):
                self._ba = prev                              # same keyframes, another set of one-way frames: the object stays
            else:
                self._ba = WindowBA(st, cfg=cfg, pix_dtype=self.pix_dtype, window_full=self.window_full, prev=prev,
                                    rec_capacity=self.cfg["graph"]["num_one_way_frames"],
                                    band_median=self.cfg.get("band_median", False))
            self._ba_prev = None
            # (`median_ahead`: "gap" -- the sequential loop's default, sequential.py -- the next iteration's full-image median is
            # streamed into a one-way frame's hand-over (median_ahead_now); "end" / True: right after every iteration's update --
            # measured no gain: in the eager loop the median branch already hides beside the block kernel)
            self._ba.median_ahead = self.cfg.get("median_ahead", False)
        ba = self._ba
        ba.step()
        info = getattr(ba, "info", None)
        if info is not None and info.is_cuda:
            if getattr(self, "_info_host", None) is None:
                self._info_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._info_host.copy_(info, non_blocking=True)
            self._info_event = torch.cuda.Event()
            self._info_event.record()
        # refresh the public state from the solver's buffers (two copies: snapshot_state)
        sn = ba.snapshot_state(self.dtype)
        B = ba.B
        self.kf_poses, self.kf_aff_params = sn["poses"][:B], sn["aff"][:B].view(B, 2, 1)
        if ba.F > B:
            self.recent_poses, self.recent_aff_params = sn["poses"][B:], sn["aff"][B:].view(ba.F - B, 2, 1)
        self.P_m = sn["P_m"]
        self.kf_pairs, self.one_way_pairs = ba.kf_pairs, ba.one_way_pairs
        if "pm" in sn:
            self.pm, self.logzm = sn["pm"], sn["logzm"].unsqueeze(-1)
            # (the scaffold's float32 mirror of these very log-depths -- the window's live buffer, valid until its next scaffold: the
            # tracker's depth image is formed from it right below, without the cast launch `depth_image` would need)
            pl = ba.w.get("px_logzm") if isinstance(getattr(ba, "w", None), dict) else None
            self._logzm_pix = (pl, self.logzm) if (pl is not None and pl.dtype == torch.float32 and pl.shape[0] == B) else None
        else:
            self.pm, self.logzm = ba.pm.to(self.dtype).clone(), ba.logzm.to(self.dtype).clone()
        self._depth_cache = None
        self.median_depths = sn["median"]
        self._state_owner = ba                               # (kf poses / affine / landmarks / medians above == ba's buffer)
        self.iter += 1
        return self.converged
