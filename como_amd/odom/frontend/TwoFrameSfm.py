"""Two-frame initialisation state machine (reference como/odom/frontend/TwoFrameSfm.py:11-234): the first frame becomes the
reference (DepthCov network -> inducing points -> per-level predictors), every following frame is aligned against it by the
coarse-to-fine photometric SfM of two_frame_sfm.py (HIP BA kernels) until the baseline is large enough to start mapping.
Same method names and return tuples as the reference class."""
import torch

from como_amd.depth_cov.core.DepthCovModule import run_model
from como_amd.depth_cov.core.samplers import sample_sparse_coords
from como_amd.odom.frontend.two_frame_sfm import setup_reference, two_frame_sfm_pyr
from como_amd.utils.coords import fill_image, normalize_coordinates
from como_amd.utils.image_processing import ImageGradientModule, ImagePyramidModule, rgb_to_grayscale

_DTYPES = {"float": torch.float32, "double": torch.float64}


class TwoFrameSfm:
    def __init__(self, cfg, intrinsics, model, cov_level, network_size):
        self.cfg = cfg
        self.device = cfg["device"]
        self.dtype = _DTYPES[cfg["dtype"]] if isinstance(cfg["dtype"], str) else cfg["dtype"]
        self.intrinsics = intrinsics
        self.model = model
        self.cov_level = cov_level
        self.network_size = network_size
        self.has_reference = False
        self.is_init = False
        self.pose_init = torch.eye(4, device=self.device, dtype=self.dtype).unsqueeze(0)
        self.aff_init = torch.zeros((1, 2, 1), device=self.device, dtype=self.dtype)

    def handle_frame(self, rgb, timestamp):
        """-> (is_init, T_curr_kf, aff_curr_kf, sparse_log_depth_kf, coords_curr, depth_curr, mean_log_depth)  (:28-79)"""
        img_and_grads = self.get_img_gradient_pyr(rgb)
        if not self.has_reference:
            self.init_frame(timestamp, rgb, img_and_grads)
            return False, None, None, None, None, None, None
        T_curr_kf, logd_kf, aff_curr_kf, coords_curr, depth_curr, mean_log_depth = self.align_frame(img_and_grads)
        reproj = fill_image(coords_curr, depth_curr, img_and_grads[-1].shape[-2:])
        n_px = self.vals_pyr[-1].shape[2]
        n_seen = torch.count_nonzero(~torch.isnan(reproj))
        icfg = self.cfg["init"]
        is_init = False
        if bool(icfg["kf_num_pixels_frac"] > n_seen / n_px):
            self.has_reference = False                       # too little overlap left: start over from the next frame
        elif bool(torch.linalg.norm(T_curr_kf[:, :3, 3]) > icfg["kf_depth_motion_ratio"] * torch.median(depth_curr)):
            is_init = True
        return is_init, T_curr_kf, aff_curr_kf, logd_kf, coords_curr, depth_curr, mean_log_depth

    def get_img_gradient_pyr(self, rgb):
        gray = rgb_to_grayscale(rgb)
        icfg = self.cfg["init"]
        pyr = ImagePyramidModule(gray.shape[-3], icfg["start_level"], icfg["end_level"], self.device, self.dtype)(gray)
        grad = ImageGradientModule(channels=gray.shape[-3], device=self.device, dtype=self.dtype)
        out = []
        for lvl in pyr:
            gx, gy = grad(lvl)
            out.append(torch.cat((lvl, gx, gy), dim=1))
        return out

    def init_frame(self, timestamp, rgb_in, img_and_grads):
        """:111-176"""
        self.timestamp, self.rgb, self.img_and_grads = timestamp, rgb_in, img_and_grads
        ns = self.network_size.tolist() if torch.is_tensor(self.network_size) else list(self.network_size)
        self.cov_params_img = run_model(self.model, self.rgb, network_size=ns, dtype=self.dtype,
                                        graphed=self.cfg.get("graph_network", True))
        sp = self.cfg["sampling"]
        self.coords_m, _ = sample_sparse_coords(self.cov_params_img, sp["max_num_coords"], mode=sp["mode"],
                                                max_stdev_thresh=sp["max_stdev_thresh"], border=sp["border"],
                                                terminate_early=False, dist_thresh=sp["dist_thresh"],
                                                signal_var=self.model.get_scale(-1), fixed_var=sp["fixed_var"])
        self.coords_m = self.coords_m.to(dtype=self.dtype)
        self.sparse_coords_norm = normalize_coordinates(self.coords_m, self.cov_params_img.shape[-2:])
        (self.vals_pyr, self.test_coords_pyr, self.Knm_Kmminv_pyr, self.img_sizes_pyr, self.intrinsics_pyr, self.dr_prior_dd,
         self.H_prior_d_d) = setup_reference(self.img_and_grads, self.sparse_coords_norm, self.model, self.cov_params_img,
                                             self.intrinsics)
        self.sparse_log_depth = torch.zeros((1, self.sparse_coords_norm.shape[1], 1), device=self.device, dtype=self.dtype)
        self.T_curr_kf = torch.eye(4, device=self.device, dtype=self.dtype).unsqueeze(0)
        self.aff_curr_kf = torch.zeros((1, 2, 1), device=self.device, dtype=self.dtype)
        self.has_reference = True

    def align_frame(self, img_and_grads):
        """:178-212: always restarts from the stored (identity / zero) initial values, as the reference."""
        return two_frame_sfm_pyr(self.T_curr_kf, self.sparse_log_depth, self.aff_curr_kf, self.test_coords_pyr, self.vals_pyr,
                                 self.Knm_Kmminv_pyr, img_and_grads, self.dr_prior_dd, self.H_prior_d_d, self.intrinsics_pyr,
                                 self.cfg["sigmas"], self.cfg["term_criteria"], self.cfg["init"])

    def delete_init_reference(self):
        for name in ("timestamp", "rgb", "cov_params_img", "img_and_grads", "coords_m", "sparse_coords_norm", "vals_pyr",
                     "test_coords_pyr", "Knm_Kmminv_pyr", "img_sizes_pyr", "intrinsics_pyr", "dr_prior_dd", "H_prior_d_d",
                     "sparse_log_depth", "T_curr_kf", "aff_curr_kf"):
            if hasattr(self, name):
                delattr(self, name)
        self.has_reference = False
