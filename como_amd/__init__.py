"""como_amd -- MI355X-native hot path of COMO's dense photometric Gauss-Newton backend and DepthCov inference path.

Host code is Python on PyTorch-ROCm; the work is done by hand-written HIP kernels in csrc/ behind the C ABI of
include/como_hip.h.  See DESIGN.md / INTEGRATION.md.
"""
import importlib
import sys

__version__ = "0.1.0"

_DROPIN = {
    "como_backends": "como_amd.como_backends",
    "como.odom.backend.photo": "como_amd.odom.backend.photo",
    "como.odom.backend.linear_system": "como_amd.odom.backend.linear_system",
    "como.odom.backend.sparse_map": "como_amd.odom.backend.sparse_map",
    "como.odom.backend.graph_pair_construction": "como_amd.odom.backend.graph_pair_construction",
    "como.odom.backend.robust_loss": "como_amd.odom.backend.robust_loss",
    "como.odom.frontend.photo_tracking": "como_amd.odom.frontend.photo_tracking",
    "como.depth_cov.core.samplers": "como_amd.depth_cov.core.samplers",
    "como.depth_cov.core.distill_depth": "como_amd.depth_cov.core.distill_depth",
    "como.utils.image_processing": "como_amd.utils.image_processing",
    "como.odom.frontend.corr": "como_amd.odom.frontend.corr",
    "como.odom.frontend.two_frame_sfm": "como_amd.odom.frontend.two_frame_sfm",
    "como.odom.frontend.TwoFrameSfm": "como_amd.odom.frontend.TwoFrameSfm",
    "como.odom.Mapping": "como_amd.odom.Mapping",
    "como.odom.Tracking": "como_amd.odom.Tracking",
    "como.geometry.transforms": "como_amd.geometry.transforms",
    "como.utils.io": "como_amd.utils.io",
    "como.geometry.affine_brightness": "como_amd.geometry.affine_brightness",
    "como.odom.factors.gp_priors": "como_amd.odom.factors.gp_priors",
    "como.odom.factors.depth_prior": "como_amd.odom.factors.depth_prior",
    "como.odom.factors.pixel_prior": "como_amd.odom.factors.pixel_prior",
    "como.odom.factors.pose_prior_factors": "como_amd.odom.factors.pose_prior_factors",
    "como.odom.factors.scalar_prior_factors": "como_amd.odom.factors.scalar_prior_factors",
}


def install_dropin():
    """Register the mirrors under the reference's module names (call BEFORE importing como.odom.Mapping / Tracking):
    `import como_backends`, `from como.odom.backend.photo import create_photo_system`, ... then resolve to this package."""
    for ref_name, mine in _DROPIN.items():
        sys.modules[ref_name] = importlib.import_module(mine)
    return sorted(_DROPIN)
